"""The real-vocabulary tokenizer path (stable_ts_amd.tokenizer.TiktokenEncoding: upstream's ``.tiktoken`` rank files)
on a synthetic rank file -- no real vocabulary exists offline.  The byte-pair encoder is checked against the HuggingFace
``tokenizers`` BPE model built from the same merges (an independent implementation), and the upstream conventions the hot
path relies on (special-token layout after the text vocabulary, unicode-safe word splitting, non-speech token list)."""
import base64
import os

import pytest

from stable_ts_amd.tokenizer import TiktokenEncoding, get_tokenizer

CORPUS = ("the quick brown fox jumps over the lazy dog. the dog barks; the fox runs away! "
          "don't stop, they're here: it's 12 345 o'clock... naïve café señor 日本語のテキスト ♪♪ music (laughs) [noise] "
          "hello hello world world the the then there these those").split(" ")


def _train(n_merges: int):
    """Tiny byte-level BPE trainer: returns the token byte strings in rank order (256 bytes, then merges)."""
    words = [(" " + w).encode("utf-8") for w in CORPUS]
    seqs = [[bytes([b]) for b in w] for w in words]
    vocab = [bytes([i]) for i in range(256)]
    for _ in range(n_merges):
        counts = {}
        for s in seqs:
            for a, b in zip(s[:-1], s[1:]):
                counts[(a, b)] = counts.get((a, b), 0) + 1
        if not counts:
            break
        (a, b), _n = max(counts.items(), key=lambda kv: (kv[1], kv[0]))
        vocab.append(a + b)
        for s in seqs:
            i = 0
            while i < len(s) - 1:
                if s[i] == a and s[i + 1] == b:
                    s[i:i + 2] = [a + b]
                else:
                    i += 1
    return vocab


@pytest.fixture(scope="module")
def vocab_dir(tmp_path_factory):
    d = tmp_path_factory.mktemp("tiktoken")
    vocab = _train(150)
    for name in ("gpt2", "multilingual"):
        with open(os.path.join(d, f"{name}.tiktoken"), "wb") as f:
            for rank, tok in enumerate(vocab):
                f.write(base64.b64encode(tok) + b" " + str(rank).encode() + b"\n")
    return str(d), vocab


def _bytes_to_unicode():
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("¡"), ord("¬") + 1)) + list(range(ord("®"), ord("ÿ") + 1))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return dict(zip(bs, [chr(c) for c in cs]))


def test_bpe_matches_huggingface_tokenizers(vocab_dir):
    tokenizers = pytest.importorskip("tokenizers")
    from tokenizers import Regex, Tokenizer, models, pre_tokenizers
    d, vocab = vocab_dir
    b2u = _bytes_to_unicode()
    uni = ["".join(b2u[b] for b in tok) for tok in vocab]
    merges = []
    for tok in vocab[256:]:                     # every merge token = concatenation of two earlier tokens
        for k in range(1, len(tok)):
            if tok[:k] in vocab[:vocab.index(tok)] and tok[k:] in vocab[:vocab.index(tok)]:
                a, b = tok[:k], tok[k:]
        merges.append(("".join(b2u[x] for x in a), "".join(b2u[x] for x in b)))
    hf = Tokenizer(models.BPE(vocab={u: i for i, u in enumerate(uni)}, merges=merges))
    hf.pre_tokenizer = pre_tokenizers.Sequence([
        pre_tokenizers.Split(Regex(TiktokenEncoding.PATTERN), behavior="isolated"),
        pre_tokenizers.ByteLevel(add_prefix_space=False, use_regex=False)])
    enc = TiktokenEncoding(os.path.join(d, "gpt2.tiktoken"), "gpt2", 99)
    for text in [" the quick brown fox", "the dog barks; they're here!", " naïve café 日本語", "hello   world\n\nthe end ",
                 " ♪♪ (laughs) 12 345", "", " "]:
        assert enc.encode(text) == hf.encode(text).ids, text
        assert enc.decode(enc.encode(text)) == text


def test_tokenizer_conventions_on_real_vocabulary_path(vocab_dir):
    d, vocab = vocab_dir
    tok = get_tokenizer(True, num_languages=100, language="ja", task="transcribe", vocab_dir=d)
    n = len(vocab)
    assert isinstance(tok.encoding, TiktokenEncoding)
    assert (tok.eot, tok.sot) == (n, n + 1) and tok.sot_sequence == (n + 1, tok.to_language_token("ja"), tok.transcribe)
    assert tok.timestamp_begin == tok.no_timestamps + 1 and tok.encoding.n_vocab == tok.timestamp_begin + 1501
    assert tok.decode_with_timestamps([tok.timestamp_begin + 54]) == "<|1.08|>"
    assert tok.decode(tok.encode(" hello") + [tok.timestamp_begin + 3]) == " hello"
    # multi-byte characters split across tokens stay together in one unit (upstream split_tokens_on_unicode)
    ids = tok.encode("猫と犬 the fox")
    words, groups = tok.split_tokens_on_unicode(ids)
    assert "".join(words) == "猫と犬 the fox" and [t for g in groups for t in g] == ids
    assert all("�" not in w for w in words) and any(len(g) > 1 for g in groups)
    en = get_tokenizer(False, num_languages=99, vocab_dir=d)
    w, g = en.split_to_word_tokens(en.encode(" the quick fox, they're here"))
    assert w[0] == " the" and "".join(w) == " the quick fox, they're here"
    ns = en.non_speech_tokens
    assert len(ns) > 10 and en.encode(" -")[0] in ns and all(0 <= t < len(vocab) for t in ns)
    with pytest.raises(FileNotFoundError):
        get_tokenizer(False, num_languages=98, vocab_dir=os.path.join(d, "missing"))


def test_hf_tokenizer_json_gives_the_same_encoder(vocab_dir, tmp_path):
    """a HuggingFace ``tokenizer.json`` built from the same merges loads into the same byte-pair encoder as the
    ``.tiktoken`` rank file (ids = merge ranks, special tokens in upstream's layout after the text vocabulary)"""
    tokenizers = pytest.importorskip("tokenizers")
    import json
    from tokenizers import Tokenizer, models
    d, vocab = vocab_dir
    b2u = _bytes_to_unicode()
    uni = ["".join(b2u[b] for b in tok) for tok in vocab]
    merges = []
    for tok in vocab[256:]:
        for k in range(1, len(tok)):
            if tok[:k] in vocab[:vocab.index(tok)] and tok[k:] in vocab[:vocab.index(tok)]:
                a, b = tok[:k], tok[k:]
        merges.append(("".join(b2u[x] for x in a), "".join(b2u[x] for x in b)))
    hf = Tokenizer(models.BPE(vocab={u: i for i, u in enumerate(uni)}, merges=merges))
    hf.add_special_tokens(["<|endoftext|>", "<|startoftranscript|>"])
    hf_dir = tmp_path / "hf"
    hf_dir.mkdir()
    hf.save(str(hf_dir / "tokenizer.json"))
    a = TiktokenEncoding(os.path.join(d, "gpt2.tiktoken"), "gpt2", 99)
    b = TiktokenEncoding(str(hf_dir / "tokenizer.json"), "gpt2", 99)
    assert a.ranks == b.ranks and a.special_tokens == b.special_tokens and a.n_vocab == b.n_vocab
    for text in [" the quick brown fox", " naïve café 日本語", "they're here!  12 345", ""]:
        assert a.encode(text) == b.encode(text) and b.decode(b.encode(text)) == text
    tok = get_tokenizer(False, num_languages=97, vocab_dir=str(hf_dir))
    assert isinstance(tok.encoding, TiktokenEncoding) and tok.eot == len(vocab)
    # a vocabulary whose ids do not follow the merge order is refused instead of silently mis-tokenising
    data = json.loads((hf_dir / "tokenizer.json").read_text(encoding="utf-8"))
    data["model"]["merges"] = data["model"]["merges"][1:] + data["model"]["merges"][:1]
    bad = tmp_path / "bad.json"
    bad.write_text(json.dumps(data), encoding="utf-8")
    with pytest.raises(ValueError):
        TiktokenEncoding(str(bad), "gpt2", 99)


def test_decode_accepts_tensors_without_mutating_them():
    # found by fuzzing locate(): the reference hands 0-dim tensors to tokenizer.decode (alignment.py:1002-1003); an in-place
    # floor division inside the synthetic vocabulary's letter code used to overwrite the caller's tensor
    import torch
    tok = get_tokenizer(False, num_languages=99)
    t = torch.tensor(28054)
    ids = [t, torch.tensor(19)]
    assert tok.decode(ids) == tok.decode([28054, 19])
    assert int(t) == 28054 and int(ids[1]) == 19
