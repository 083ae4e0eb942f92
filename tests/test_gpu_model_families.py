"""The model families between base and large-v3 on hardware: small.en (d = 768, 12 heads, 12 + 12 layers), medium (d = 1024, 16 heads,
24 + 24 layers, multilingual vocabulary) and large-v3-turbo (128 mels, 32 encoder layers, FOUR decoder layers).  The other GPU files run
tiny / base (d = 384 / 512) and large-v3 (d = 1280); the decode-step GEMMs are compiled per K depth (gemm_dec_f16<MT, NKS = K / 32, ..>:
12, 16, 20, 24, 32, 40 -- swx_decstep.hip), the attention kernels per head count, so d = 768 and d = 1024 are their own instantiations.

Per family, against the f32 CPU oracle on the same seeded weights (reference: decode.py:33-65, timing.py:166-198; the model table
whisper_compatibility.py:310-335 -> upstream _MODELS / ModelDimensions):
 * strict f32: greedy and beam-5 token ids identical, |delta avg logprob| <= 1e-3, no-speech probability; the scoring pass's token
   probabilities <= 1e-3, the cost matrix <= 2e-3, the DTW index path identical;
 * fp16: the fused decode step (dec GEMMs, LayerNorm folded) against the per-op path at 11 windows x 5 beams (the default dispatch) and at one
   window (the single-wave kernels): near-tie flips allowed, drift not; the leading tokens equal the oracle's; window k of the 11-window batch
   equals the window alone bit for bit (tokens, lengths, sum logprobs)."""
import gc

import numpy as np
import pytest
import torch

from oracle import stable as ost
from oracle.whisper import model as om
from oracle.whisper.decoding import DecodingOptions
from oracle.whisper.tokenizer import get_tokenizer

pytestmark = pytest.mark.gpu

#            mels  actx  astate ahead alayer vocab  tctx tstate thead tlayer
FAMILIES = {
    "small.en": ((80, 1500, 768, 12, 12, 51864, 448, 768, 12, 12), ((6, 6), (7, 0), (7, 3), (8, 2), (9, 0), (10, 1))),
    "medium": ((80, 1500, 1024, 16, 24, 51865, 448, 1024, 16, 24), ((11, 4), (14, 1), (14, 12), (16, 13), (19, 7), (21, 9))),
    "large-v3-turbo": ((128, 1500, 1280, 20, 32, 51866, 448, 1280, 20, 4), ((2, 4), (2, 11), (3, 3), (3, 6), (3, 11), (3, 14))),
}
NAMES = tuple(FAMILIES)
_CACHE = {}


@pytest.fixture(scope="module", autouse=True)
def _release():
    yield
    _CACHE.clear()
    gc.collect()
    torch.cuda.empty_cache()


def _dims(name):
    return om.ModelDimensions(*FAMILIES[name][0])


def _product_table_agrees(name):
    import stable_ts_amd as sw
    assert tuple(sw.dims_for(name).__dict__.values()) == FAMILIES[name][0], name


def _oracle(name, ln_jitter=0.0):
    key = ("o", name, ln_jitter)
    if key not in _CACHE:
        m = om.build_model(_dims(name), seed=1234, std=0.02, embed_gain=3.0, ln_jitter=ln_jitter)
        mask = torch.zeros(m.dims.n_text_layer, m.dims.n_text_head, dtype=torch.bool)
        for l, h in FAMILIES[name][1]:
            mask[l, h] = True
        m.set_alignment_heads_mask(mask)
        _CACHE[key] = m
    return _CACHE[key]


def _engine(name, dtype, ln_jitter=0.0, max_windows=11, max_rows=55):
    from stable_ts_amd.engine import Engine, ModelDimensions
    key = ("e", name, dtype, ln_jitter)
    if key not in _CACHE:
        d = _dims(name)
        eng = Engine(ModelDimensions(**d.__dict__), dtype=dtype, max_windows=max_windows, max_rows=max_rows,
                     alignment_heads=FAMILIES[name][1])
        eng.load_state_dict(om.random_state_dict(d, 1234, 0.02, 3.0, 1.0, ln_jitter))
        _CACHE[key] = eng
    return _CACHE[key]


def _drop(name):
    """one family at a time in memory (the medium / turbo oracles are 3-6 GB of f32 on the host)"""
    for k in [k for k in _CACHE if k[1] != name]:
        del _CACHE[k]
    gc.collect()
    torch.cuda.empty_cache()


def _mel(n_mels, seed=0, B=1):
    g = torch.Generator().manual_seed(seed)
    t = torch.linspace(0, 1, 3000)
    base = torch.sin(t[None, None, :] * (5 + torch.arange(n_mels)[None, :, None] * 0.37)) * 0.5
    return (base + 0.3 * torch.randn(B, n_mels, 3000, generator=g)).float()


def _tok_cfg(tok, task):
    return dict(eot=tok.eot, sot=tok.sot, no_timestamps=tok.no_timestamps, timestamp_begin=tok.timestamp_begin,
                no_speech=tok.no_speech, blank_token=tok.encode(" ")[0], suppress_tokens=list(task._get_suppress_tokens()))


def _rank(out, w):
    scores = []
    for k in range(out["tokens"].shape[1]):
        ln = int(out["lens"][w, k])
        scores.append(out["sum_logprobs"][w, k] / ln if ln > 0 else -np.inf)
    return int(np.argmax(scores))


def _task(m, n, beam):
    return ost.DecodingTaskStable(m, DecodingOptions(fp16=False, language="en", max_initial_timestamp=None, sample_len=n,
                                                     beam_size=5 if beam else None))


def _same_prefix(a, b):
    n = 0
    for x, y in zip(a, b):
        if x != y:
            break
        n += 1
    return n


@pytest.mark.parametrize("case", ["greedy", "beam", "score"])
@pytest.mark.parametrize("name", NAMES)            # (the decorator next to the function is the OUTER loop: one model build per family)
def test_family_strict_f32_vs_oracle(name, case):
    if case == "score":
        return _score_alignment_dtw_strict(name)
    beam = case == "beam"
    _drop(name)
    _product_table_agrees(name)
    m, eng = _oracle(name), _engine(name, "f32")
    n = 14 if beam else 20
    mel = _mel(m.dims.n_mels, 21, B=1)
    options = DecodingOptions(fp16=False, language="en", max_initial_timestamp=None, sample_len=n, beam_size=5 if beam else None)
    res, _ = ost.decode_stable(m, mel[0], options, min_tokens=n)
    task = _task(m, n, beam)
    xkv = eng.cross_kv(eng.encode(mel.cuda().contiguous()))
    out = eng.decode(xkv, [list(task.initial_tokens)], n_group=task.n_group, beam=beam, patience=None, sample_len=n,
                     sot_index=task.sot_index, min_tokens=n, **_tok_cfg(task.tokenizer, task))
    sb = out["sample_begin"]
    best = _rank(out, 0)
    got = out["tokens"][0, best, sb: sb + int(out["lens"][0, best])].tolist()
    assert got == res.tokens, (got, res.tokens)
    assert abs(out["sum_logprobs"][0, best] / (len(got) + 1) - res.avg_logprob) < 1e-3
    assert abs(out["no_speech_prob"][0] - res.no_speech_prob) < 1e-4 + 1e-2 * res.no_speech_prob


def _score_alignment_dtw_strict(name):
    _drop(name)
    m, eng = _oracle(name), _engine(name, "f32")
    tok = get_tokenizer(m.is_multilingual, num_languages=m.num_languages, language="en", task="transcribe")
    mel = _mel(m.dims.n_mels, 61, B=1)
    text = torch.randint(18, 50000, (41,), generator=torch.Generator().manual_seed(5)).tolist()
    wt, cache = ost.find_alignment(m, tok, text, mel[0], 480000, return_cache=True)
    toks = [[*tok.sot_sequence, tok.no_timestamps, *text, tok.eot]]
    xkv = eng.cross_kv(eng.encode(mel.cuda().contiguous()))
    probs, neg, T = eng.score(xkv, toks, [1500], n_sot=len(tok.sot_sequence), eot=tok.eot)
    paths = eng.dtw(neg, [t + 1 for t in T], [1500])
    ref_p = np.asarray(cache["text_token_probs"])
    assert np.abs(np.asarray(probs[0]) - ref_p).max() < 1e-3 * max(1e-3, ref_p.max()) + 1e-7
    assert (neg[0, :T[0] + 1, :1500].cpu() - cache["neg_matrix"]).abs().max().item() < 2e-3
    ri, rj = cache["dtw_path"]
    ti, tj = paths[0]
    assert ti.tolist() == ri.tolist() and tj.tolist() == rj.tolist()


@pytest.mark.parametrize("windows", [11, 1])
@pytest.mark.parametrize("name", NAMES)
def test_family_decode_f16_step_kernels(name, windows):
    """weights with non-trivial LayerNorm gamma / beta (ln_jitter 0.1): the folded LayerNorm of the dec GEMMs is exercised"""
    from stable_ts_amd import _lib
    _drop(name)
    lib = _lib.load()
    m, eng = _oracle(name, 0.1), _engine(name, "f16", 0.1)
    mels = _mel(m.dims.n_mels, 71, B=windows)
    task = _task(m, 24, True)
    kw = dict(n_group=task.n_group, beam=True, sample_len=24, sot_index=task.sot_index, min_tokens=24, **_tok_cfg(task.tokenizer, task))
    xkv = eng.cross_kv(eng.encode(mels.cuda().contiguous()))
    init = [list(task.initial_tokens)] * windows
    old = lib.swx_debug_flags(-1)
    try:
        fast = eng.decode(xkv, init, **kw)
        lib.swx_debug_flags(1)                     # per-op path: LayerNorm kernel + tiled / skinny GEMMs
        slow = eng.decode(xkv, init, **kw)
    finally:
        lib.swx_debug_flags(old)
    sb = fast["sample_begin"]
    seqs = [fast["tokens"][w, _rank(fast, w), sb:sb + 24].tolist() for w in range(windows)]
    agree = sum(_same_prefix(seqs[w], slow["tokens"][w, _rank(slow, w), sb:sb + 24].tolist()) for w in range(windows))
    assert agree >= windows * 24 * 0.6, agree
    assert np.isfinite(fast["sum_logprobs"][fast["lens"] > 0]).all()
    assert np.allclose(fast["no_speech_prob"], slow["no_speech_prob"], rtol=2e-2, atol=1e-6)
    # the first window against the f32 oracle: the leading tokens agree
    options = DecodingOptions(fp16=False, language="en", max_initial_timestamp=None, sample_len=24, beam_size=5)
    res, _ = ost.decode_stable(m, mels[0], options, min_tokens=24)
    assert _same_prefix(seqs[0], res.tokens) >= 3, (seqs[0], res.tokens)
    if windows > 1:
        # batch invariance of the fp16 step: window k of the batch = the window alone, bit for bit
        for k in (0, 4, windows - 1):
            xk = eng.cross_kv(eng.encode(mels[k:k + 1].cuda().contiguous()))
            one = eng.decode(xk, init[:1], **kw)
            for key in ("tokens", "lens", "sum_logprobs"):
                assert np.array_equal(np.asarray(one[key][0]), np.asarray(fast[key][k])), (k, key)


def test_family_transcribe_end_to_end_turbo():
    """load_model() -> transcribe(word_timestamps=True) on the family whose shape differs most from the benchmark's (128 mels, 32 encoder
    layers, 4 decoder layers; the host side's model table and alignment-head table for it): 65 s of the bench's synthetic audio in
    window-parallel batches of 3 (a ragged last window), beam 5 -- segments and words come back, in order, inside the recording"""
    import bench
    import stable_ts_amd as sw
    _CACHE.clear()
    gc.collect()
    torch.cuda.empty_cache()
    model = sw.load_model("large-v3-turbo", device="cuda:0", weights="random")
    audio = bench.synth_audio(65.0, seed=3)
    res = model.transcribe(audio, language="en", temperature=0.0, beam_size=5, sample_len=24, min_tokens=24, word_timestamps=True,
                           batch_size=3, logprob_threshold=None, compression_ratio_threshold=None, no_speech_threshold=None)
    words = res.all_words()
    assert len(res.segments) >= 1 and len(words) >= 1
    assert all(0.0 <= w.start <= w.end <= 65.0 + 1e-6 for w in words)
    starts = [s.start for s in res.segments]
    assert starts == sorted(starts)


def test_family_sequential_transcribe_and_align_end_to_end_medium():
    """the reference's OWN control flow (model.transcribe(audio): one window per decode call, seek from the last timestamp, prompt carried
    over -- original_whisper.py:492-710) and model.align() (alignment.py:396-429) on the d = 1024 family: five-row launches of the
    single-wave dec GEMMs at K = 1024, the long-context self-attention behind a carried-over prompt, one window per encoder pass"""
    import bench
    import stable_ts_amd as sw
    _CACHE.clear()
    gc.collect()
    torch.cuda.empty_cache()
    model = sw.load_model("medium", device="cuda:0", weights="random")
    audio = bench.synth_audio(75.0, seed=4)
    res = model.transcribe(audio, language="en", temperature=0.0, beam_size=5, sample_len=32, min_tokens=32, word_timestamps=True,
                           logprob_threshold=None, compression_ratio_threshold=None, no_speech_threshold=None)
    assert all(0.0 <= w.start <= w.end <= 75.0 + 1e-6 for w in res.all_words())
    assert [s.start for s in res.segments] == sorted(s.start for s in res.segments)
    text = torch.randint(18, 50000, (60,), generator=torch.Generator().manual_seed(9)).tolist()
    al = model.align(audio, text, language="en", token_step=40)
    words = al.all_words()
    assert len(words) >= 10 and all(0.0 <= w.start <= w.end <= 75.0 + 1e-6 for w in words)
    assert sum(len(w.tokens) for w in words) == len(text)
