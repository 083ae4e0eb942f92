"""Tokenizer surface consumed by the hot path (reference: whisper_compatibility.py:310-335;
uses at timing.py:62,230-241,310-318 ; original_whisper.py:340-344,410-415,550-590).

Upstream (openai-whisper==20250625) wraps a tiktoken BPE whose vocabulary files
(``gpt2.tiktoken`` / ``multilingual.tiktoken``) are not available offline, so the *text*
side here is a synthetic, reversible vocabulary; the *special-token id layout* is upstream's
 because the decoding rules and the alignment code depend on it.
Host-side only: no arithmetic of the hot path lives here.

Synthetic text vocabulary (ids < eot):
  id 0..15      punctuation / symbols, one character each (see _PUNCT)
  id 16         " "      (SuppressBlank needs tokenizer.encode(" "))
  id 17         " ..."   (stable-ts gap padding, timing.py:426)
  id >= 18      a 4-letter lowercase group, base-26 of the id; ids with id % 3 != 0 carry a
                leading space (start a new word), the others continue the previous word.
Every token string is unique and `decode` is plain concatenation, so
``encode(decode(ids)) == ids`` for any id list without specials.

With a real vocabulary at hand (upstream's ``gpt2.tiktoken`` / ``multilingual.tiktoken``, or a HuggingFace
``tokenizer.json`` of a Whisper checkpoint, in a directory named by ``SWX_TIKTOKEN_DIR`` or
``get_tokenizer(..., vocab_dir=...)``) `TiktokenEncoding` takes the synthetic vocabulary's place: same interface, tiktoken's byte-pair algorithm, upstream's special-token layout (tests/test_tokenizer_cpu.py).
"""
import string
from dataclasses import dataclass, field
from functools import cached_property, lru_cache
from typing import Dict, List, Optional, Tuple, Union

LANGUAGES = {
    "en": "english", "zh": "chinese", "de": "german", "es": "spanish", "ru": "russian", "ko": "korean",
    "fr": "french", "ja": "japanese", "pt": "portuguese", "tr": "turkish", "pl": "polish", "ca": "catalan",
    "nl": "dutch", "ar": "arabic", "sv": "swedish", "it": "italian", "id": "indonesian", "hi": "hindi",
    "fi": "finnish", "vi": "vietnamese", "he": "hebrew", "uk": "ukrainian", "el": "greek", "ms": "malay",
    "cs": "czech", "ro": "romanian", "da": "danish", "hu": "hungarian", "ta": "tamil", "no": "norwegian",
    "th": "thai", "ur": "urdu", "hr": "croatian", "bg": "bulgarian", "lt": "lithuanian", "la": "latin",
    "mi": "maori", "ml": "malayalam", "cy": "welsh", "sk": "slovak", "te": "telugu", "fa": "persian",
    "lv": "latvian", "bn": "bengali", "sr": "serbian", "az": "azerbaijani", "sl": "slovenian",
    "kn": "kannada", "et": "estonian", "mk": "macedonian", "br": "breton", "eu": "basque", "is": "icelandic",
    "hy": "armenian", "ne": "nepali", "mn": "mongolian", "bs": "bosnian", "kk": "kazakh", "sq": "albanian",
    "sw": "swahili", "gl": "galician", "mr": "marathi", "pa": "punjabi", "si": "sinhala", "km": "khmer",
    "sn": "shona", "yo": "yoruba", "so": "somali", "af": "afrikaans", "oc": "occitan", "ka": "georgian",
    "be": "belarusian", "tg": "tajik", "sd": "sindhi", "gu": "gujarati", "am": "amharic", "yi": "yiddish",
    "lo": "lao", "uz": "uzbek", "fo": "faroese", "ht": "haitian creole", "ps": "pashto", "tk": "turkmen",
    "nn": "nynorsk", "mt": "maltese", "sa": "sanskrit", "lb": "luxembourgish", "my": "myanmar",
    "bo": "tibetan", "tl": "tagalog", "mg": "malagasy", "as": "assamese", "tt": "tatar", "haw": "hawaiian",
    "ln": "lingala", "ha": "hausa", "ba": "bashkir", "jw": "javanese", "su": "sundanese", "yue": "cantonese",
}
TO_LANGUAGE_CODE = {
    **{language: code for code, language in LANGUAGES.items()},
    "burmese": "my", "valencian": "ca", "flemish": "nl", "haitian": "ht", "letzeburgesch": "lb",
    "pushto": "ps", "panjabi": "pa", "moldavian": "ro", "moldovan": "ro", "sinhalese": "si",
    "castilian": "es", "mandarin": "zh",
}

_PUNCT = [".", ",", "!", "?", ":", ";", "\"", "'", "(", ")", "-", "[", "]", "{", "}", "%"]
_ID_SPACE = 16
_ID_GAP = 17
_FIRST_WORD_ID = 18


def _letters(i: int) -> str:
    s = ""
    for _ in range(4):
        s = chr(ord("a") + i % 26) + s
        i //= 26
    return s


def _unletters(s: str) -> int:
    v = 0
    for ch in s:
        v = v * 26 + (ord(ch) - ord("a"))
    return v


class SyntheticEncoding:
    """Reversible text vocabulary + upstream's special-token layout."""

    def __init__(self, name: str, num_languages: int):
        self.name = name
        self.n_text = 50256 if name == "gpt2" else 50257
        specials = [
            "<|endoftext|>", "<|startoftranscript|>",
            *[f"<|{lang}|>" for lang in list(LANGUAGES.keys())[:num_languages]],
            "<|translate|>", "<|transcribe|>", "<|startoflm|>", "<|startofprev|>",
            "<|nospeech|>", "<|notimestamps|>",
            *[f"<|{i * 0.02:.2f}|>" for i in range(1501)],
        ]
        self.special_tokens: Dict[str, int] = {}
        n = self.n_text
        for tok in specials:
            self.special_tokens[tok] = n
            n += 1
        self.n_vocab = n
        self.eot_token = self.special_tokens["<|endoftext|>"]
        self._special_by_id = {v: k for k, v in self.special_tokens.items()}
        self._str_cache: Dict[int, str] = {}

    def token_str(self, t: int) -> str:
        t = int(t)                      # callers hand over 0-dim tensors too (alignment.py:1002-1003); never mutate them
        s = self._str_cache.get(t)
        if s is None:
            s = self._str_cache[t] = self._token_str(t)
        return s

    def _token_str(self, t: int) -> str:
        if t >= self.n_text:
            return self._special_by_id[t]
        if t < len(_PUNCT):
            return _PUNCT[t]
        if t == _ID_SPACE:
            return " "
        if t == _ID_GAP:
            return " ..."
        return (" " if t % 3 != 0 else "") + _letters(t)

    def decode(self, tokens: List[int]) -> str:
        return "".join(self.token_str(t) for t in tokens)

    def encode(self, text: str) -> List[int]:
        def is_grp(g: str) -> bool:
            return len(g) == 4 and all("a" <= c <= "z" for c in g)

        out: List[int] = []
        i = 0
        n = len(text)
        while i < n:
            if text.startswith(" ...", i):
                out.append(_ID_GAP)
                i += 4
                continue
            ch = text[i]
            if ch == " ":
                grp = text[i + 1:i + 5]
                if is_grp(grp):
                    tid = _unletters(grp)
                    if _FIRST_WORD_ID <= tid < self.n_text and tid % 3 != 0:
                        out.append(tid)
                        i += 5
                        continue
                out.append(_ID_SPACE)
                i += 1
                continue
            if ch in _PUNCT:
                out.append(_PUNCT.index(ch))
                i += 1
                continue
            grp = text[i:i + 4]
            if is_grp(grp):
                tid = _unletters(grp)
                if _FIRST_WORD_ID <= tid < self.n_text and tid % 3 == 0:
                    out.append(tid)
                    i += 4
                    continue
            # not produced by decode(): hash the character into the continuation range so that
            # encode stays total (free text such as an initial_prompt)
            h = _FIRST_WORD_ID + (ord(ch) * 7919) % (self.n_text - _FIRST_WORD_ID)
            out.append(h - h % 3 if h - h % 3 >= _FIRST_WORD_ID else _FIRST_WORD_ID)
            i += 1
        return out


class TiktokenEncoding:
    """Real vocabulary: upstream's ``<name>.tiktoken`` rank file (one ``base64(token bytes) rank`` pair per line, e.g.
    whisper/assets/gpt2.tiktoken / multilingual.tiktoken) with a byte-pair encoder that follows tiktoken's algorithm:
    the text is cut by upstream's GPT-2 pattern, each piece is split into bytes and the adjacent pair with the lowest
    merge rank is merged until none is left.  Special tokens are appended after the text vocabulary in upstream's
    order (whisper/tokenizer.py::get_encoding), so every id convention of the hot path (eot, sot, timestamps) holds.
    No vocabulary file ships with this repository (none is available offline): point ``SWX_TIKTOKEN_DIR`` or
    ``get_tokenizer(..., vocab_dir=...)`` at a directory that holds the two files."""

    PATTERN = r"""'s|'t|'re|'ve|'m|'ll|'d| ?\p{L}+| ?\p{N}+| ?[^\s\p{L}\p{N}]+|\s+(?!\S)|\s+"""

    def __init__(self, path: str, name: str, num_languages: int):
        import regex
        self.name = name
        self.ranks: Dict[bytes, int] = self._read_hf_json(path) if path.endswith(".json") else self._read_tiktoken(path)
        self.n_text = len(self.ranks)
        self._bytes_by_id = {v: k for k, v in self.ranks.items()}
        specials = [
            "<|endoftext|>", "<|startoftranscript|>",
            *[f"<|{lang}|>" for lang in list(LANGUAGES.keys())[:num_languages]],
            "<|translate|>", "<|transcribe|>", "<|startoflm|>", "<|startofprev|>",
            "<|nospeech|>", "<|notimestamps|>",
            *[f"<|{i * 0.02:.2f}|>" for i in range(1501)],
        ]
        self.special_tokens: Dict[str, int] = {t: self.n_text + i for i, t in enumerate(specials)}
        self.n_vocab = self.n_text + len(specials)
        self.eot_token = self.special_tokens["<|endoftext|>"]
        self._special_by_id = {v: k for k, v in self.special_tokens.items()}
        self._pat = regex.compile(self.PATTERN)

    @staticmethod
    def _read_tiktoken(path: str) -> Dict[bytes, int]:
        import base64
        ranks: Dict[bytes, int] = {}
        with open(path, "rb") as f:
            for line in f:
                if line.strip():
                    tok, rank = line.split()
                    ranks[base64.b64decode(tok)] = int(rank)
        return ranks

    @staticmethod
    def _read_hf_json(path: str) -> Dict[bytes, int]:
        """The text vocabulary of a HuggingFace ``tokenizer.json`` (byte-level BPE as shipped with the HF Whisper
        checkpoints; whisper_word_level/hf_whisper.py loads those): token strings are written in GPT-2's printable
        byte alphabet and, for a GPT-2-derived vocabulary, the ids ARE the merge ranks (256 bytes first, then one token
        per merge in merge order), which is checked here.  Special tokens are not taken from the file: the hot path
        relies on upstream's layout after the text vocabulary, and the HF files follow it."""
        import json
        with open(path, "r", encoding="utf-8") as f:
            model = json.load(f)["model"]
        if model.get("type") != "BPE":
            raise ValueError(f"{path}: expected a BPE tokenizer, got {model.get('type')!r}")
        bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("\u00a1"), ord("\u00ac") + 1)) + list(range(ord("\u00ae"), ord("\u00ff") + 1))
        cs, extra = bs[:], 0
        for b in range(256):
            if b not in bs:
                bs.append(b)
                cs.append(256 + extra)
                extra += 1
        to_byte = {chr(c): b for b, c in zip(bs, cs)}
        vocab = {bytes(to_byte[ch] for ch in tok): int(i) for tok, i in model["vocab"].items()
                 if all(ch in to_byte for ch in tok)}
        text = {t: i for t, i in vocab.items() if i < len(vocab)}
        if sorted(text.values()) != list(range(len(text))):
            raise ValueError(f"{path}: text-vocabulary ids are not contiguous from 0")
        for k, m in enumerate(model.get("merges", [])):
            a, b = m.split(" ") if isinstance(m, str) else m
            tok = bytes(to_byte[ch] for ch in a) + bytes(to_byte[ch] for ch in b)
            if text.get(tok) != 256 + k:
                raise ValueError(f"{path}: ids do not follow the merge order (merge {k}); not a GPT-2 style vocabulary")
        return text

    def _bpe(self, piece: bytes) -> List[int]:
        parts = [piece[i:i + 1] for i in range(len(piece))]
        while len(parts) > 1:
            best, best_rank = -1, None
            for i in range(len(parts) - 1):
                r = self.ranks.get(parts[i] + parts[i + 1])
                if r is not None and (best_rank is None or r < best_rank):
                    best, best_rank = i, r
            if best < 0:
                break
            parts[best:best + 2] = [parts[best] + parts[best + 1]]
        return [self.ranks[p] for p in parts]

    def encode(self, text: str) -> List[int]:
        out: List[int] = []
        for piece in self._pat.findall(text):
            b = piece.encode("utf-8")
            r = self.ranks.get(b)
            out.extend([r] if r is not None else self._bpe(b))
        return out

    def decode_bytes(self, tokens: List[int]) -> bytes:
        tokens = [int(t) for t in tokens]          # 0-dim tensors are accepted like ints
        return b"".join(self._bytes_by_id[t] if t < self.n_text else self._special_by_id[t].encode() for t in tokens)

    def decode(self, tokens: List[int]) -> str:
        return self.decode_bytes(tokens).decode("utf-8", errors="replace")


@dataclass
class Tokenizer:
    encoding: Union[SyntheticEncoding, "TiktokenEncoding"]
    num_languages: int
    language: Optional[str] = None
    task: Optional[str] = None
    sot_sequence: Tuple[int] = ()
    special_tokens: Dict[str, int] = field(default_factory=dict)

    def __post_init__(self):
        self.special_tokens = dict(self.encoding.special_tokens)
        sot: int = self.special_tokens["<|startoftranscript|>"]
        translate: int = self.special_tokens["<|translate|>"]
        transcribe: int = self.special_tokens["<|transcribe|>"]
        langs = tuple(LANGUAGES.keys())[: self.num_languages]
        sot_sequence = [sot]
        if self.language is not None:
            sot_sequence.append(sot + 1 + langs.index(self.language))
        if self.task is not None:
            task_token: int = transcribe if self.task == "transcribe" else translate
            sot_sequence.append(task_token)
        self.sot_sequence = tuple(sot_sequence)

    def encode(self, text, **kwargs):
        return self.encoding.encode(text)

    def decode(self, token_ids: List[int], **kwargs) -> str:
        token_ids = [t for t in token_ids if t < self.timestamp_begin]
        return self.encoding.decode(token_ids)

    def decode_with_timestamps(self, token_ids: List[int], **kwargs) -> str:
        return self.encoding.decode(token_ids)

    @cached_property
    def eot(self) -> int:
        return self.encoding.eot_token

    @cached_property
    def transcribe(self) -> int:
        return self.special_tokens["<|transcribe|>"]

    @cached_property
    def translate(self) -> int:
        return self.special_tokens["<|translate|>"]

    @cached_property
    def sot(self) -> int:
        return self.special_tokens["<|startoftranscript|>"]

    @cached_property
    def sot_lm(self) -> int:
        return self.special_tokens["<|startoflm|>"]

    @cached_property
    def sot_prev(self) -> int:
        return self.special_tokens["<|startofprev|>"]

    @cached_property
    def no_speech(self) -> int:
        return self.special_tokens["<|nospeech|>"]

    @cached_property
    def no_timestamps(self) -> int:
        return self.special_tokens["<|notimestamps|>"]

    @cached_property
    def timestamp_begin(self) -> int:
        return self.special_tokens["<|0.00|>"]

    @cached_property
    def language_token(self) -> int:
        if self.language is None:
            raise ValueError("This tokenizer does not have language token configured")
        return self.to_language_token(self.language)

    def to_language_token(self, language):
        if token := self.special_tokens.get(f"<|{language}|>", None):
            return token
        raise KeyError(f"Language {language} not found in tokenizer.")

    @cached_property
    def all_language_tokens(self) -> Tuple[int]:
        result = []
        for token, token_id in self.special_tokens.items():
            if token.strip("<|>") in LANGUAGES:
                result.append(token_id)
        return tuple(result)[: self.num_languages]

    @cached_property
    def all_language_codes(self) -> Tuple[str]:
        return tuple(self.encoding._special_by_id[_l].strip("<|>") for _l in self.all_language_tokens)

    @cached_property
    def sot_sequence_including_notimestamps(self) -> Tuple[int]:
        return tuple(list(self.sot_sequence) + [self.no_timestamps])

    @cached_property
    def non_speech_tokens(self) -> Tuple[int]:
        """Upstream: symbols / brackets / music notes that are suppressed unless they are spoken.
        Synthetic vocabulary: the bracket and quote punctuation ids (a fixed subset, so that the
        SuppressTokens path is exercised with a non-trivial list)."""
        if isinstance(self.encoding, SyntheticEncoding):
            return tuple(sorted(_PUNCT.index(c) for c in ['"', "(", ")", "[", "]", "{", "}", "%"]))
        symbols = list("\"#()*+/:;<=>@[\\]^_`{|}~「」『』")
        symbols += "<< >> <<< >>> -- --- -( -[ (' (\" (( )) ((( ))) [[ ]] {{ }} ♪♪ ♪♪♪".split()
        misc = set("♩♪♫♬♭♮♯")
        result = {self.encoding.encode(" -")[0], self.encoding.encode(" '")[0]}
        for symbol in symbols + list(misc):
            for toks in (self.encoding.encode(symbol), self.encoding.encode(" " + symbol)):
                if len(toks) == 1 or symbol in misc:
                    result.add(toks[0])
        return tuple(sorted(result))

    def split_to_word_tokens(self, tokens: List[int]):
        if self.language in {"zh", "ja", "th", "lo", "my", "yue"}:
            return self.split_tokens_on_unicode(tokens)
        return self.split_tokens_on_spaces(tokens)

    def split_tokens_on_unicode(self, tokens: List[int]):
        """Upstream whisper/tokenizer.py::split_tokens_on_unicode: a unit ends where the running decoding holds no
        U+FFFD (a token boundary inside a multi-byte character) -- unless the text itself contains one there."""
        full = self.decode_with_timestamps(tokens)
        bad = "\ufffd"
        words, word_tokens, run, offset = [], [], [], 0
        for token in tokens:
            run.append(token)
            piece = self.decode_with_timestamps(run)
            if bad not in piece or full[offset + piece.index(bad)] == bad:
                words.append(piece)
                word_tokens.append(run)
                run = []
                offset += len(piece)
        return words, word_tokens

    def split_tokens_on_spaces(self, tokens: List[int]):
        subwords, subword_tokens_list = self.split_tokens_on_unicode(tokens)
        words = []
        word_tokens = []
        for subword, subword_tokens in zip(subwords, subword_tokens_list):
            special = subword_tokens[0] >= self.eot
            with_space = subword.startswith(" ")
            punctuation = subword.strip() in string.punctuation
            if special or with_space or punctuation or len(words) == 0:
                words.append(subword)
                word_tokens.append(subword_tokens)
            else:
                words[-1] = words[-1] + subword
                word_tokens[-1].extend(subword_tokens)
        return words, word_tokens


def _vocab_file(name: str, vocab_dir: Optional[str]) -> Optional[str]:
    import os
    for d in (vocab_dir, os.environ.get("SWX_TIKTOKEN_DIR")):
        if d:
            p = os.path.join(d, f"{name}.tiktoken")
            for cand in (p, os.path.join(d, f"{name}.json"), os.path.join(d, "tokenizer.json")):
                if os.path.isfile(cand):
                    return cand
            if d is vocab_dir:
                raise FileNotFoundError(p)
    return None


@lru_cache(maxsize=None)
def get_encoding(name: str = "gpt2", num_languages: int = 99, vocab_dir: Optional[str] = None):
    path = _vocab_file(name, vocab_dir)
    if path is not None:
        return TiktokenEncoding(path, name, num_languages)
    return SyntheticEncoding(name, num_languages)


@lru_cache(maxsize=None)
def get_tokenizer(multilingual: bool, *, num_languages: int = 99, language: Optional[str] = None,
                  task: Optional[str] = None, vocab_dir: Optional[str] = None) -> Tokenizer:
    if language is not None:
        language = language.lower()
        if language not in LANGUAGES:
            if language in TO_LANGUAGE_CODE:
                language = TO_LANGUAGE_CODE[language]
            else:
                raise ValueError(f"Unsupported language: {language}")
    if multilingual:
        encoding_name = "multilingual"
        language = language or "en"
        task = task or "transcribe"
    else:
        encoding_name = "gpt2"
        language = None
        task = None
    encoding = get_encoding(name=encoding_name, num_languages=num_languages, vocab_dir=vocab_dir)
    return Tokenizer(encoding=encoding, num_languages=num_languages, language=language, task=task)
