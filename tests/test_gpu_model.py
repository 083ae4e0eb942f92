"""GPU parity of the model-level entry points of libswx (encoder, teacher-forced logits, decoding loop, scoring +
alignment matrix + DTW) against the CPU oracle with the SAME seeded random weights at real architecture dims.

Strict mode (dtype f32) is held to the north-star bar: identical token ids, bit-exact DTW paths, logprobs within 1e-3.
The fp16 mode (what the reference itself runs on a GPU) is held to fp16 tolerances and reported as agreement.
"""
import os

import numpy as np
import pytest
import torch

from oracle import stable as ost
from oracle.whisper import model as om
from oracle.whisper.decoding import DecodingOptions
from oracle.whisper.tokenizer import get_tokenizer

pytestmark = pytest.mark.gpu

_CACHE = {}


def _dims(name):
    return om.dims_for(name)


def _oracle(name, seed=1234, std=0.02, gain=3.0, heads=None):
    key = ("o", name, seed, std, gain, heads)
    if key not in _CACHE:
        m = om.build_model(name, seed=seed, std=std, embed_gain=gain)
        if heads is not None:
            mask = torch.zeros(m.dims.n_text_layer, m.dims.n_text_head, dtype=torch.bool)
            for l, h in heads:
                mask[l, h] = True
            m.set_alignment_heads_mask(mask)
        _CACHE[key] = m
    return _CACHE[key]


def _engine(name, dtype, seed=1234, std=0.02, gain=3.0, heads=None, max_windows=2, max_rows=10):
    from stable_ts_amd.engine import Engine, ModelDimensions
    key = ("e", name, dtype, seed, std, gain, heads)
    if key not in _CACHE:
        d = _dims(name)
        eng = Engine(ModelDimensions(**d.__dict__), dtype=dtype, max_windows=max_windows, max_rows=max_rows,
                     alignment_heads=heads)
        eng.load_state_dict(om.random_state_dict(d, seed, std, gain))
        _CACHE[key] = eng
    return _CACHE[key]


def _mel(n_mels, seed=0, B=1):
    g = torch.Generator().manual_seed(seed)
    t = torch.linspace(0, 1, 3000)
    base = torch.sin(t[None, None, :] * (5 + torch.arange(n_mels)[None, :, None] * 0.37)) * 0.5
    return (base + 0.3 * torch.randn(B, n_mels, 3000, generator=g)).float()


HEADS_TINY = ((2, 1), (2, 4), (3, 0), (3, 2), (3, 5))


@pytest.mark.parametrize("name", ["tiny.en", "base.en"])
def test_encoder_f32(name):
    m, eng = _oracle(name), _engine(name, "f32")
    mel = _mel(m.dims.n_mels, 1, B=2)
    with torch.no_grad():
        ref = m.encoder(mel)
    got = eng.encode(mel.cuda().contiguous()).cpu()
    err = (got - ref).abs().max().item()
    assert err < 2e-4 * max(1.0, ref.abs().max().item()), err


def test_encoder_f16_tiny():
    m, eng = _oracle("tiny.en"), _engine("tiny.en", "f16")
    mel = _mel(80, 2, B=2)
    with torch.no_grad():
        ref = m.encoder(mel)
    got = eng.encode(mel.cuda().contiguous()).float().cpu()
    # fp16 storage of activations through 4 layers: 3e-2 abs on LN-normalised outputs of O(1)
    err = (got - ref).abs()
    assert err.max().item() < 6e-2 and err.mean().item() < 6e-3, (err.max().item(), err.mean().item())


@pytest.mark.parametrize("name,dtype,tol", [("tiny.en", "f32", 1e-3), ("base.en", "f32", 1e-3), ("tiny.en", "f16", 0.25)])
def test_forward_logits(name, dtype, tol):
    m, eng = _oracle(name), _engine(name, dtype)
    mel = _mel(m.dims.n_mels, 3, B=2)
    g = torch.Generator().manual_seed(11)
    toks = [torch.randint(0, 50000, (n,), generator=g).tolist() for n in (37, 20)]
    with torch.no_grad():
        xa = m.encoder(mel)
        refs = [m.decoder(torch.tensor([t]), xa[i:i + 1])[0] for i, t in enumerate(toks)]
    xkv = eng.cross_kv(eng.encode(mel.cuda().contiguous()))
    got = eng.forward_logits(xkv, toks).cpu()
    for i, t in enumerate(toks):
        err = (got[i, :len(t)] - refs[i]).abs().max().item()
        assert err < tol, (i, err)
        lp_err = (got[i, :len(t)].log_softmax(-1) - refs[i].log_softmax(-1)).abs().max().item()
        assert lp_err < tol, (i, lp_err)


def _tok_cfg(tok, task):
    return dict(eot=tok.eot, sot=tok.sot, no_timestamps=tok.no_timestamps, timestamp_begin=tok.timestamp_begin,
                no_speech=tok.no_speech, blank_token=tok.encode(" ")[0], suppress_tokens=list(task._get_suppress_tokens()))


def _rank(out, w):
    """upstream MaximumLikelihoodRanker with length_penalty=None: argmax of sum_logprob / length"""
    scores = []
    for k in range(out["tokens"].shape[1]):
        ln = int(out["lens"][w, k])
        scores.append(-np.inf if ln < 0 else (out["sum_logprobs"][w, k] / ln if ln > 0 else -np.inf))
    return int(np.argmax(scores))


def _oracle_decode(m, mel, **opt):
    min_tokens = opt.pop("min_tokens", 0)
    ts_mask = opt.pop("ts_token_mask", None)
    options = DecodingOptions(fp16=False, language="en", max_initial_timestamp=None, **opt)
    return ost.decode_stable(m, mel, options, ts_token_mask=ts_mask, min_tokens=min_tokens)


@pytest.mark.parametrize("name,gain,opts", [
    ("tiny.en", 3.0, dict(sample_len=40, min_tokens=40)),
    ("tiny.en", 9.0, dict(sample_len=40, min_tokens=40)),
    ("tiny.en", 3.0, dict(sample_len=24, min_tokens=0)),
    ("base.en", 3.0, dict(sample_len=30, min_tokens=30, prompt=[1000, 2000, 3001, 40000, 7])),
    ("tiny.en", 3.0, dict(sample_len=32, min_tokens=32, beam_size=5)),
    ("tiny.en", 2.0, dict(sample_len=32, min_tokens=32, beam_size=5)),
    ("tiny.en", 3.0, dict(sample_len=40, min_tokens=6, beam_size=3, patience=2.0)),
    ("base.en", 3.0, dict(sample_len=28, min_tokens=28, beam_size=5, prompt=[555, 666])),
])
def test_decode_strict_f32_identical_tokens(name, gain, opts):
    m, eng = _oracle(name, gain=gain), _engine(name, "f32", gain=gain)
    mels = _mel(m.dims.n_mels, 21, B=2)
    opts = dict(opts)
    for w in range(2):
        res, _ = _oracle_decode(m, mels[w], **dict(opts))
        task = ost.DecodingTaskStable(m, DecodingOptions(fp16=False, language="en", max_initial_timestamp=None,
                                                         **{k: v for k, v in opts.items() if k != "min_tokens"}))
        tok = task.tokenizer
        xkv = eng.cross_kv(eng.encode(mels[w:w + 1].cuda().contiguous()))
        out = eng.decode(xkv, [list(task.initial_tokens)], n_group=task.n_group, beam=opts.get("beam_size") is not None,
                         patience=opts.get("patience"), sample_len=task.sample_len, sot_index=task.sot_index,
                         min_tokens=opts.get("min_tokens", 0), **_tok_cfg(tok, task))
        sb = out["sample_begin"]
        best = _rank(out, 0)
        got_tokens = out["tokens"][0, best, sb: sb + int(out["lens"][0, best])].tolist()
        assert got_tokens == res.tokens, (w, got_tokens, res.tokens)
        got_avg = out["sum_logprobs"][0, best] / (len(got_tokens) + 1)
        assert abs(got_avg - res.avg_logprob) < 1e-3
        assert abs(out["no_speech_prob"][0] - res.no_speech_prob) < 1e-4 + 1e-2 * res.no_speech_prob


def test_decode_batched_windows_equal_single():
    # W windows decoded in lockstep == each window decoded alone (the sharding / batching contract)
    name = "tiny.en"
    m, eng = _oracle(name), _engine(name, "f32")
    mels = _mel(80, 31, B=2)
    task = ost.DecodingTaskStable(m, DecodingOptions(fp16=False, language="en", max_initial_timestamp=None, sample_len=20, beam_size=5))
    tok = task.tokenizer
    kw = dict(n_group=5, beam=True, sample_len=20, sot_index=task.sot_index, min_tokens=20, **_tok_cfg(tok, task))
    xkv2 = eng.cross_kv(eng.encode(mels.cuda().contiguous()))
    both = eng.decode(xkv2, [list(task.initial_tokens)] * 2, **kw)
    for w in range(2):
        xkv1 = eng.cross_kv(eng.encode(mels[w:w + 1].cuda().contiguous()))
        one = eng.decode(xkv1, [list(task.initial_tokens)], **kw)
        assert np.array_equal(one["tokens"][0], both["tokens"][w])
        assert np.allclose(one["sum_logprobs"][0], both["sum_logprobs"][w], atol=1e-4)


def test_decode_ts_mask_and_greedy_eot():
    name = "tiny.en"
    m, eng = _oracle(name), _engine(name, "f32")
    mel = _mel(80, 41, B=1)
    mask = torch.zeros(1501, dtype=torch.bool)
    mask[::3] = True
    mask[100:400] = True
    res, _ = _oracle_decode(m, mel[0], sample_len=36, min_tokens=30, ts_token_mask=mask)
    task = ost.DecodingTaskStable(m, DecodingOptions(fp16=False, language="en", max_initial_timestamp=None, sample_len=36))
    xkv = eng.cross_kv(eng.encode(mel.cuda().contiguous()))
    out = eng.decode(xkv, [list(task.initial_tokens)], sample_len=36, sot_index=task.sot_index, min_tokens=30,
                     ts_mask=mask[None], **_tok_cfg(task.tokenizer, task))
    sb = out["sample_begin"]
    got = out["tokens"][0, 0, sb: sb + int(out["lens"][0, 0])].tolist()
    assert got == res.tokens


def test_decode_f16_agreement_tiny():
    # fp16 weights/activations vs the fp32 oracle on random weights: report agreement of the greedy token stream;
    # the first tokens must agree (divergence later is the expected near-tie effect of random logits)
    name = "tiny.en"
    m, eng = _oracle(name), _engine(name, "f16")
    mel = _mel(80, 51, B=1)
    res, _ = _oracle_decode(m, mel[0], sample_len=24, min_tokens=24)
    task = ost.DecodingTaskStable(m, DecodingOptions(fp16=False, language="en", max_initial_timestamp=None, sample_len=24))
    xkv = eng.cross_kv(eng.encode(mel.cuda().contiguous()))
    out = eng.decode(xkv, [list(task.initial_tokens)], sample_len=24, sot_index=task.sot_index, min_tokens=24,
                     **_tok_cfg(task.tokenizer, task))
    sb = out["sample_begin"]
    got = out["tokens"][0, 0, sb: sb + int(out["lens"][0, 0])].tolist()
    n_same = 0
    for a, b in zip(got, res.tokens):
        if a != b:
            break
        n_same += 1
    assert n_same >= 3, (got, res.tokens)


@pytest.mark.parametrize("name,beam", [("tiny.en", False), ("base.en", True)])
def test_decode_f16_fast_step_equals_general_path(name, beam):
    # the fused decode step (un-split "dec" GEMMs, LayerNorm folded; f16) vs the per-op path (f16, flag 1): same arithmetic
    # up to f32 summation order
    from stable_ts_amd import _lib
    lib = _lib.load()
    m, eng = _oracle(name), _engine(name, "f16")
    mels = _mel(m.dims.n_mels, 71, B=3)
    task = ost.DecodingTaskStable(m, DecodingOptions(fp16=False, language="en", max_initial_timestamp=None, sample_len=24,
                                                     beam_size=5 if beam else None))
    kw = dict(n_group=task.n_group, beam=beam, sample_len=24, sot_index=task.sot_index, min_tokens=24,
              **_tok_cfg(task.tokenizer, task))
    xkv = eng.cross_kv(eng.encode(mels.cuda().contiguous()))
    fast = eng.decode(xkv, [list(task.initial_tokens)] * 3, **kw)
    old = lib.swx_debug_flags(1)
    try:
        slow = eng.decode(xkv, [list(task.initial_tokens)] * 3, **kw)
    finally:
        lib.swx_debug_flags(old)
    sb = fast["sample_begin"]
    agree = 0
    for w in range(3):
        a = fast["tokens"][w, _rank(fast, w), sb:sb + 24].tolist()
        b = slow["tokens"][w, _rank(slow, w), sb:sb + 24].tolist()
        n = 0
        for x, y in zip(a, b):
            if x != y:
                break
            n += 1
        agree += n
    assert agree >= 3 * 24 * 0.6, agree      # near-tie flips under different f32 summation order are allowed, drift is not
    assert np.allclose(fast["no_speech_prob"], slow["no_speech_prob"], rtol=2e-2, atol=1e-6)


@pytest.mark.parametrize("name,beam,windows", [("tiny.en", False, 3), ("base.en", True, 3), ("base.en", True, 11)])
def test_decode_f16_dec_step_equals_general_path(name, beam, windows):
    # third-generation decode step (un-split "dec" GEMMs with the LayerNorm folded into their epilogues, swx_decstep.hip) vs
    # the per-op path (LayerNorm kernel + tiled / skinny GEMMs), both f16, on weights with NON-trivial LayerNorm gamma / beta:
    # same mathematics, different rounding points -> near-tie flips allowed, drift not.  11 windows x 5 beams = 55 rows crosses
    # the row threshold at which the step is used by default; the 3-window cases force it.
    from stable_ts_amd import _lib
    from stable_ts_amd.engine import Engine, ModelDimensions
    lib = _lib.load()
    d = _dims(name)
    key = ("ej", name)
    if key not in _CACHE:
        eng = Engine(ModelDimensions(**d.__dict__), dtype="f16", max_windows=11, max_rows=55)
        eng.load_state_dict(om.random_state_dict(d, 1234, 0.02, 3.0, 1.0, 0.1))
        _CACHE[key] = eng
        _CACHE[("oj", name)] = om.build_model(name, seed=1234, std=0.02, embed_gain=3.0, ln_jitter=0.1)
    eng, m = _CACHE[key], _CACHE[("oj", name)]
    mels = _mel(m.dims.n_mels, 71, B=windows)
    task = ost.DecodingTaskStable(m, DecodingOptions(fp16=False, language="en", max_initial_timestamp=None, sample_len=24,
                                                     beam_size=5 if beam else None))
    kw = dict(n_group=task.n_group, beam=beam, sample_len=24, sot_index=task.sot_index, min_tokens=24,
              **_tok_cfg(task.tokenizer, task))
    xkv = eng.cross_kv(eng.encode(mels.cuda().contiguous()))
    old = lib.swx_debug_flags(-1)
    try:
        fast = eng.decode(xkv, [list(task.initial_tokens)] * windows, **kw)
        lib.swx_debug_flags(1)
        slow = eng.decode(xkv, [list(task.initial_tokens)] * windows, **kw)
    finally:
        lib.swx_debug_flags(old)
    sb = fast["sample_begin"]
    agree = 0
    for w in range(windows):
        a = fast["tokens"][w, _rank(fast, w), sb:sb + 24].tolist()
        b = slow["tokens"][w, _rank(slow, w), sb:sb + 24].tolist()
        n = 0
        for x, y in zip(a, b):
            if x != y:
                break
            n += 1
        agree += n
    assert agree >= windows * 24 * 0.6, agree
    assert np.allclose(fast["no_speech_prob"], slow["no_speech_prob"], rtol=2e-2, atol=1e-6)
    # and the first window against the f32 oracle: the leading tokens agree
    res, _ = _oracle_decode(m, mels[0], sample_len=24, min_tokens=24, **({"beam_size": 5} if beam else {}))
    a = fast["tokens"][0, _rank(fast, 0), sb:sb + 24].tolist()
    n_same = 0
    for x, y in zip(a, res.tokens):
        if x != y:
            break
        n_same += 1
    assert n_same >= 3, (a, res.tokens)


@pytest.mark.parametrize("name,beam", [("tiny.en", False), ("base.en", True)])
def test_decode_f16_packed_cross_kv_is_bit_identical(name, beam):
    # the decode-step cross-attention reading the fragment-ordered copy of K / V^T (swx_xkv_pack) vs the row layout (flag
    # 2048): only the load addresses differ, so tokens and sums of log-probabilities must be IDENTICAL
    from stable_ts_amd import _lib
    lib = _lib.load()
    m, eng = _oracle(name), _engine(name, "f16")
    mels = _mel(m.dims.n_mels, 91, B=3)
    task = ost.DecodingTaskStable(m, DecodingOptions(fp16=False, language="en", max_initial_timestamp=None, sample_len=32,
                                                     beam_size=5 if beam else None))
    kw = dict(n_group=task.n_group, beam=beam, sample_len=32, sot_index=task.sot_index, min_tokens=32,
              **_tok_cfg(task.tokenizer, task))
    xkv = eng.cross_kv(eng.encode(mels.cuda().contiguous()))
    old = lib.swx_debug_flags(-1)
    try:
        packed = eng.decode(xkv, [list(task.initial_tokens)] * 3, **kw)
        lib.swx_debug_flags(old | 2048)
        rows = eng.decode(xkv, [list(task.initial_tokens)] * 3, **kw)
    finally:
        lib.swx_debug_flags(old)
    assert np.array_equal(np.asarray(packed["lens"]), np.asarray(rows["lens"]))
    assert np.array_equal(np.asarray(packed["tokens"]), np.asarray(rows["tokens"]))
    assert np.array_equal(np.asarray(packed["sum_logprobs"]), np.asarray(rows["sum_logprobs"]))


@pytest.mark.parametrize("name,beam,windows", [("tiny.en", False, 3), ("base.en", True, 3), ("base.en", True, 11)])
def test_decode_f16_l2_prefetch_has_no_functional_effect(name, beam, windows):
    # every kernel of the decode step issues a few loads whose results nobody reads: the 128-byte lines of the NEXT projection's
    # weights, into the L2 of the XCD that will run them (DecPrefetch, swx_kernels.h).  Flag 32768 points them at the kernel's own
    # weights instead.  Tokens, lengths and sums of log-probabilities must be IDENTICAL (55 rows: three row groups per panel).
    from stable_ts_amd import _lib
    lib = _lib.load()
    m, eng = _oracle(name), _engine(name, "f16")
    mels = _mel(m.dims.n_mels, 131, B=windows)
    task = ost.DecodingTaskStable(m, DecodingOptions(fp16=False, language="en", max_initial_timestamp=None, sample_len=30,
                                                     beam_size=5 if beam else None))
    kw = dict(n_group=task.n_group, beam=beam, sample_len=30, sot_index=task.sot_index, min_tokens=30,
              **_tok_cfg(task.tokenizer, task))
    xkv = eng.cross_kv(eng.encode(mels.cuda().contiguous()))
    old = lib.swx_debug_flags(-1)
    try:
        assert not (old & 32768)
        on = eng.decode(xkv, [list(task.initial_tokens)] * windows, **kw)
        lib.swx_debug_flags(old | 32768)
        off = eng.decode(xkv, [list(task.initial_tokens)] * windows, **kw)
    finally:
        lib.swx_debug_flags(old)
    assert np.array_equal(np.asarray(on["lens"]), np.asarray(off["lens"]))
    assert np.array_equal(np.asarray(on["tokens"]), np.asarray(off["tokens"]))
    assert np.array_equal(np.asarray(on["sum_logprobs"]), np.asarray(off["sum_logprobs"]))


@pytest.mark.parametrize("name,beam,windows", [("tiny.en", False, 3), ("base.en", True, 3), ("base.en", True, 11), ("tiny.en", True, 1)])
def test_decode_f16_fused_cross_query_is_bit_identical(name, beam, windows):
    # the decode step computes the cross-attention query projection INSIDE the cross-attention launch (AttnArgs::fq_*: the dec
    # GEMM's statistics, k-step order and epilogue per (window, head)); flag 1048576 runs it as the separate gemm_dec_f16 launch it
    # replaces.  Tokens, lengths and sums of log-probabilities must be IDENTICAL (1 / 5 rows per window, 1-11 windows).
    from stable_ts_amd import _lib
    lib = _lib.load()
    m, eng = _oracle(name), _engine(name, "f16")
    mels = _mel(m.dims.n_mels, 77, B=windows)
    task = ost.DecodingTaskStable(m, DecodingOptions(fp16=False, language="en", max_initial_timestamp=None, sample_len=30,
                                                     beam_size=5 if beam else None))
    kw = dict(n_group=task.n_group, beam=beam, sample_len=30, sot_index=task.sot_index, min_tokens=30,
              **_tok_cfg(task.tokenizer, task))
    xkv = eng.cross_kv(eng.encode(mels.cuda().contiguous()))
    old = lib.swx_debug_flags(-1)
    try:
        assert not (old & 1048576)
        on = eng.decode(xkv, [list(task.initial_tokens)] * windows, **kw)
        lib.swx_debug_flags(old | 1048576)
        off = eng.decode(xkv, [list(task.initial_tokens)] * windows, **kw)
    finally:
        lib.swx_debug_flags(old)
    assert np.array_equal(np.asarray(on["lens"]), np.asarray(off["lens"]))
    assert np.array_equal(np.asarray(on["tokens"]), np.asarray(off["tokens"]))
    assert np.array_equal(np.asarray(on["sum_logprobs"]), np.asarray(off["sum_logprobs"]))
    assert np.array_equal(np.asarray(on["no_speech_prob"]), np.asarray(off["no_speech_prob"]))


@pytest.mark.parametrize("name,beam,windows", [("tiny.en", False, 3), ("base.en", True, 3), ("base.en", True, 11), ("tiny.en", True, 1)])
def test_decode_f16_cross_attention_two_blocks_in_flight_is_bit_identical(name, beam, windows):
    # round 6: the fused decode-step cross-attention with TWO key blocks of a wave in flight and non-temporal K / V^T loads
    # (attn_decode_cross_xq2_f16, the default) against the loop that requests and consumes one block per iteration (flag 33554432 =
    # SWX_FLAG_XATTN_R5): the same blocks in the same order per wave, so tokens, lengths, sums of log-probabilities and the no-speech
    # probability must be IDENTICAL.
    from stable_ts_amd import _lib
    lib = _lib.load()
    m, eng = _oracle(name), _engine(name, "f16")
    mels = _mel(m.dims.n_mels, 78, B=windows)
    task = ost.DecodingTaskStable(m, DecodingOptions(fp16=False, language="en", max_initial_timestamp=None, sample_len=30,
                                                     beam_size=5 if beam else None))
    kw = dict(n_group=task.n_group, beam=beam, sample_len=30, sot_index=task.sot_index, min_tokens=30,
              **_tok_cfg(task.tokenizer, task))
    xkv = eng.cross_kv(eng.encode(mels.cuda().contiguous()))
    old = lib.swx_debug_flags(-1)
    try:
        lib.swx_debug_flags(old & ~33554432)
        a = eng.decode(xkv, [list(task.initial_tokens)] * windows, **kw)
        lib.swx_debug_flags(old | 33554432)
        b = eng.decode(xkv, [list(task.initial_tokens)] * windows, **kw)
    finally:
        lib.swx_debug_flags(old)
    for key in ("lens", "tokens", "sum_logprobs", "no_speech_prob"):
        assert np.array_equal(np.asarray(a[key]), np.asarray(b[key])), key
    # the multi-token passes stream the same way (attn_decode_cross2_f16, 1 / 2 / 4 groups of 16 rows): a scoring pass over the windows
    heads_list = [tuple(p) for p in m.alignment_heads.indices().T.tolist()]
    eng.set_alignment_heads(heads_list)
    tok = task.tokenizer
    g = torch.Generator().manual_seed(windows)
    toks = [[*tok.sot_sequence, tok.no_timestamps, *torch.randint(18, 50000, (40 + 7 * (w % 3),), generator=g).tolist(), tok.eot]
            for w in range(windows)]
    try:
        lib.swx_debug_flags(old & ~33554432)
        p1, n1, _ = eng.score(xkv, toks, [1500] * windows, n_sot=len(tok.sot_sequence), eot=tok.eot)
        n1 = n1.clone()
        lib.swx_debug_flags(old | 33554432)
        p2, n2, _ = eng.score(xkv, toks, [1500] * windows, n_sot=len(tok.sot_sequence), eot=tok.eot)
    finally:
        lib.swx_debug_flags(old)
    assert p1 == p2 and torch.equal(n1, n2)


@pytest.mark.parametrize("name,windows", [("tiny.en", 1), ("base.en", 1), ("base.en", 3), ("tiny.en", 7)])
def test_cross_kv_fragment_ordered_copy_from_the_projection_epilogue_is_the_same_bytes(name, windows):
    # round 6: the fragment-ordered copy of a layer's cross-attention K / V^T (what the decode-step cross-attention streams) is written by
    # the K | V projection's own epilogue (GemmArgs::P) instead of by a launch that reads K / V^T again (swx_xkv_pack; flag 128 =
    # SWX_FLAG_XKV_PACK_SEPARATE): the WHOLE cross-K/V buffer -- row-layout K, V^T with its zeroed key padding, the packed copy with its
    # zeroed padding keys -- must hold the same bytes, twice in a row (a second call must not depend on what the first one left behind)
    from stable_ts_amd import _lib
    lib = _lib.load()
    m, eng = _oracle(name), _engine(name, "f16")
    mels = _mel(m.dims.n_mels, 131, B=windows).cuda().contiguous()
    xa = eng.encode(mels).clone()
    old = lib.swx_debug_flags(-1)
    try:
        lib.swx_debug_flags(old | 128)
        ref = eng.cross_kv(xa).clone()
        lib.swx_debug_flags(old & ~128)
        got1 = eng.cross_kv(xa).clone()
        eng.cross_kv(eng.encode(_mel(m.dims.n_mels, 132, B=windows).cuda().contiguous()))      # other contents in between
        got2 = eng.cross_kv(xa).clone()
    finally:
        lib.swx_debug_flags(old)
    assert ref.numel() > 0 and int((ref != 0).sum()) > ref.numel() // 2
    assert torch.equal(ref, got1), int((ref != got1).sum())
    assert torch.equal(ref, got2), int((ref != got2).sum())


@pytest.mark.parametrize("name,windows", [("tiny.en", 1), ("base.en", 2), ("base.en", 5)])
def test_encoder_f16_v_transposed_by_the_qkv_epilogue_is_bit_identical(name, windows):
    # round 6: at few windows the encoder's Q | K | V projection stores V transposed per head from its own epilogue (EPI_QKV_VT, incl.
    # the zeroed key padding) instead of through swx_transpose_v (flag 268435456 = SWX_FLAG_QKV_SEPARATE_VT): the same f16 values in
    # the same places, so the encoder output must be IDENTICAL -- twice in a row (the V^T buffer is the MLP's hidden buffer and holds
    # the previous layer's activations when the epilogue writes it)
    from stable_ts_amd import _lib
    lib = _lib.load()
    m, eng = _oracle(name), _engine(name, "f16")
    mels = _mel(m.dims.n_mels, 91, B=windows).cuda().contiguous()
    old = lib.swx_debug_flags(-1)
    try:
        lib.swx_debug_flags(old | 268435456)
        ref = eng.encode(mels).clone()
        lib.swx_debug_flags(old & ~268435456)
        got1 = eng.encode(mels).clone()
        got2 = eng.encode(mels).clone()
    finally:
        lib.swx_debug_flags(old)
    assert torch.equal(ref.view(torch.int16), got1.view(torch.int16)) and torch.equal(ref.view(torch.int16), got2.view(torch.int16))


@pytest.mark.parametrize("name,beam,windows", [("tiny.en", True, 3), ("base.en", True, 4), ("base.en", False, 7)])
def test_decode_f16_self_attention_five_rows_per_workgroup_is_bit_identical(name, beam, windows):
    # round 6 experiment (flag 134217728 = SWX_FLAG_SELFATTN_WG5): the decode-step self-attention with five rows (a window's beams)
    # per workgroup instead of one wave per workgroup -- per row the same instructions, incl. a row count that is no multiple of five
    from stable_ts_amd import _lib
    lib = _lib.load()
    m, eng = _oracle(name), _engine(name, "f16")
    mels = _mel(m.dims.n_mels, 78, B=windows)
    task = ost.DecodingTaskStable(m, DecodingOptions(fp16=False, language="en", max_initial_timestamp=None, sample_len=30,
                                                     beam_size=5 if beam else None))
    kw = dict(n_group=task.n_group, beam=beam, sample_len=30, sot_index=task.sot_index, min_tokens=30,
              **_tok_cfg(task.tokenizer, task))
    xkv = eng.cross_kv(eng.encode(mels.cuda().contiguous()))
    old = lib.swx_debug_flags(-1)
    try:
        lib.swx_debug_flags(old & ~134217728)
        a = eng.decode(xkv, [list(task.initial_tokens)] * windows, **kw)
        lib.swx_debug_flags(old | 134217728)
        b = eng.decode(xkv, [list(task.initial_tokens)] * windows, **kw)
    finally:
        lib.swx_debug_flags(old)
    for key in ("lens", "tokens", "sum_logprobs", "no_speech_prob"):
        assert np.array_equal(np.asarray(a[key]), np.asarray(b[key])), key


@pytest.mark.parametrize("name,mode", [("tiny.en", "greedy"), ("tiny.en", "beam"), ("base.en", "sample"), ("base.en", "beam_masks"),
                                       ("tiny.en", "greedy_free")])
def test_decode_select_register_kernel_is_bit_identical(name, mode):
    # decode_select_reg_kernel (the logits row read once into registers) vs decode_select_kernel (11-16 walks of the row in
    # memory, flag 8192): same element-to-thread assignment and summation order by construction, so tokens, lengths, sums of
    # log-probabilities must be IDENTICAL -- greedy, beam, temperature sampling (Gumbel keyed on the window id), with the
    # timestamp rules on / off, a max_initial_timestamp, silence-masked timestamp tokens, EOT allowed early (min_tokens 0).
    from stable_ts_amd import _lib
    lib = _lib.load()
    m, eng = _oracle(name), _engine(name, "f16")
    mels = _mel(m.dims.n_mels, 97, B=3)
    beam = mode.startswith("beam")
    task = ost.DecodingTaskStable(m, DecodingOptions(fp16=False, language="en", sample_len=36,
                                                     max_initial_timestamp=1.0 if mode == "beam_masks" else None,
                                                     beam_size=5 if beam else None))
    kw = dict(n_group=task.n_group, beam=beam, sample_len=36, sot_index=task.sot_index,
              min_tokens=0 if mode == "greedy_free" else 36, **_tok_cfg(task.tokenizer, task))
    if mode == "sample":
        kw.update(temperature=0.7, seed=1234, window_uid=[7, 300, 12])
    if mode == "beam_masks":
        g = torch.Generator().manual_seed(3)
        kw.update(ts_mask=torch.rand(3, 1501, generator=g) < 0.4, max_initial_timestamp_index=50)
    xkv = eng.cross_kv(eng.encode(mels.cuda().contiguous()))
    old = lib.swx_debug_flags(-1)
    try:
        assert not (old & 8192)
        reg = eng.decode(xkv, [list(task.initial_tokens)] * 3, **kw)
        lib.swx_debug_flags(old | 8192)
        mem = eng.decode(xkv, [list(task.initial_tokens)] * 3, **kw)
    finally:
        lib.swx_debug_flags(old)
    assert np.array_equal(np.asarray(reg["lens"]), np.asarray(mem["lens"]))
    assert np.array_equal(np.asarray(reg["tokens"]), np.asarray(mem["tokens"]))
    assert np.array_equal(np.asarray(reg["sum_logprobs"]), np.asarray(mem["sum_logprobs"]))


@pytest.mark.parametrize("name,mode,sample_len", [("tiny.en", "greedy", 37), ("tiny.en", "beam", 40), ("base.en", "beam_free", 41),
                                                  ("base.en", "sample", 24), ("tiny.en", "ctx_full", 60)])
def test_decode_graph_replay_is_bit_identical(name, mode, sample_len):
    # the decode loop replays ONE captured two-step hipGraph (swx_decode: units 2k, 2k+1) instead of launching ~290 kernels per
    # step; flag 16384 launches every step eagerly.  Same kernels, same arguments, the step index is device state in both:
    # tokens, lengths, sums of log-probabilities and no-speech probabilities must be IDENTICAL -- greedy, beam, sampling, odd and
    # even budgets, EOT allowed early (the every-8-steps completion poll), and a prompt that fills the context mid-way
    # (decode.py:60).  The second graph run hits the handle's cache (same key), the third has another key (other budget).
    from stable_ts_amd import _lib
    lib = _lib.load()
    m, eng = _oracle(name), _engine(name, "f16")
    mels = _mel(m.dims.n_mels, 113, B=3)
    beam = mode.startswith("beam")
    o = dict(fp16=False, language="en", max_initial_timestamp=None, sample_len=sample_len, beam_size=5 if beam else None)
    if mode == "ctx_full":
        o["prompt"] = list(range(1000, 1000 + 223))          # n_init = 226: the context (448) fills before the budget... 
        sample_len = 224
        o["sample_len"] = sample_len
    task = ost.DecodingTaskStable(m, DecodingOptions(**o))
    kw = dict(n_group=task.n_group, beam=beam, sample_len=sample_len, sot_index=task.sot_index,
              min_tokens=0 if mode == "beam_free" else sample_len, **_tok_cfg(task.tokenizer, task))
    if mode == "sample":
        kw.update(temperature=0.6, seed=99, window_uid=[3, 1, 400])
    xkv = eng.cross_kv(eng.encode(mels.cuda().contiguous()))
    init = [list(task.initial_tokens)] * 3
    old = lib.swx_debug_flags(-1)
    st0 = eng.graph_stats()
    try:
        assert not (old & 16384)
        g1 = eng.decode(xkv, init, **kw)
        g2 = eng.decode(xkv, init, **kw)
        g3 = eng.decode(xkv, init, **dict(kw, sample_len=sample_len - 3, min_tokens=min(kw["min_tokens"], sample_len - 3)))
        st1 = eng.graph_stats()
        lib.swx_debug_flags(old | 16384)
        e1 = eng.decode(xkv, init, **kw)
        e3 = eng.decode(xkv, init, **dict(kw, sample_len=sample_len - 3, min_tokens=min(kw["min_tokens"], sample_len - 3)))
    finally:
        lib.swx_debug_flags(old)
    for a, b in ((g1, e1), (g2, e1), (g3, e3)):
        assert np.array_equal(np.asarray(a["lens"]), np.asarray(b["lens"]))
        assert np.array_equal(np.asarray(a["tokens"]), np.asarray(b["tokens"]))
        assert np.array_equal(np.asarray(a["sum_logprobs"]), np.asarray(b["sum_logprobs"]))
        assert np.array_equal(np.asarray(a["no_speech_prob"]), np.asarray(b["no_speech_prob"]))
    assert int(np.asarray(g1["lens"]).max()) > 8
    # the graph really ran: replays were counted, the second call reused the first call's graph (one capture less than calls),
    # nothing fell back to eager launches, and the eager runs added no replay
    assert not st1["fell_back"], st1
    assert st1["replays"] - st0["replays"] >= 3 * ((int(g1["steps"]) - 2) // 2) - 3, (st0, st1, g1["steps"])
    assert st1["captures"] - st0["captures"] <= 2, (st0, st1)
    assert eng.graph_stats()["replays"] == st1["replays"]


@pytest.mark.parametrize("name,beam,windows,prompt_len", [("tiny.en", False, 1, 223), ("base.en", True, 1, 223), ("base.en", True, 2, 150),
                                                          ("tiny.en", True, 3, 0), ("base.en", False, 1, 0)])
def test_decode_f16_few_workgroup_kernels_are_bit_identical(name, beam, windows, prompt_len):
    # round 6: a decode of few rows -- the reference's sequential flow, one window per call, with up to 223 prompt tokens carried over
    # (positions 226-340) -- runs its projections as SINGLE-wave workgroups (gemm_dec_f16<.., WPB = 1>, launches of <= 80 workgroups) and
    # its long-context self-attention on the kernel that requests every load of a row in two batches (self_attn_step_long_f16, <= 1 024
    # waves).  Flags 16 / 4 put the four-wave workgroups / the chunk-by-chunk kernel back: through the whole loop (prefill, captured
    # step graph, beam bookkeeping) tokens, lengths, sums of log-probabilities and no-speech probabilities must be IDENTICAL.
    from stable_ts_amd import _lib
    lib = _lib.load()
    m, eng = _oracle(name), _engine(name, "f16")
    mels = _mel(m.dims.n_mels, 117, B=windows)
    o = dict(fp16=False, language="en", max_initial_timestamp=None, sample_len=40, beam_size=5 if beam else None)
    if prompt_len:
        o["prompt"] = list(range(1000, 1000 + prompt_len))
    task = ost.DecodingTaskStable(m, DecodingOptions(**o))
    kw = dict(n_group=task.n_group, beam=beam, sample_len=40, sot_index=task.sot_index, min_tokens=40, **_tok_cfg(task.tokenizer, task))
    xkv = eng.cross_kv(eng.encode(mels.cuda().contiguous()))
    init = [list(task.initial_tokens)] * windows
    old = lib.swx_debug_flags(-1)
    try:
        lib.swx_debug_flags(old & ~(4 | 16))
        new1 = eng.decode(xkv, init, **kw)
        lib.swx_debug_flags(old | 4 | 16)
        ref = eng.decode(xkv, init, **kw)
        lib.swx_debug_flags((old & ~4) | 16)
        new2 = eng.decode(xkv, init, **kw)
    finally:
        lib.swx_debug_flags(old)
    assert int(np.asarray(ref["lens"]).max()) > 8
    for got in (new1, new2):
        for key in ("lens", "tokens", "sum_logprobs", "no_speech_prob"):
            assert np.array_equal(np.asarray(got[key]), np.asarray(ref[key])), key


@pytest.mark.parametrize("name,heads", [("tiny.en", HEADS_TINY), ("base.en", None)])
def test_score_alignment_dtw_strict(name, heads):
    m, eng = _oracle(name, heads=heads), _engine(name, "f32", heads=heads)
    if heads is None:
        heads_list = [tuple(p) for p in m.alignment_heads.indices().T.tolist()]
        eng.set_alignment_heads(heads_list)
    tok = get_tokenizer(False, num_languages=m.num_languages)
    mels = _mel(m.dims.n_mels, 61, B=2)
    g = torch.Generator().manual_seed(5)
    texts = [torch.randint(18, 50000, (n,), generator=g).tolist() for n in (57, 23)]
    num_samples = [480000, 301234]
    toks, refs = [], []
    for w in range(2):
        wt, cache = ost.find_alignment(m, tok, texts[w], mels[w], num_samples[w], return_cache=True)
        refs.append((wt, cache))
        toks.append([*tok.sot_sequence, tok.no_timestamps, *texts[w], tok.eot])
    xkv = eng.cross_kv(eng.encode(mels.cuda().contiguous()))
    n_frames = [round(n / 320) for n in num_samples]
    probs, neg, T = eng.score(xkv, toks, n_frames, n_sot=len(tok.sot_sequence), eot=tok.eot)
    paths = eng.dtw(neg, [t + 1 for t in T], n_frames)
    for w in range(2):
        wt, cache = refs[w]
        ref_p = np.asarray(cache["text_token_probs"])
        assert np.abs(np.asarray(probs[w]) - ref_p).max() < 1e-3 * max(1e-3, ref_p.max()) + 1e-7
        ref_neg = cache["neg_matrix"]
        got_neg = neg[w, :T[w] + 1, :n_frames[w]].cpu()
        assert (got_neg - ref_neg).abs().max().item() < 2e-3
        # DTW: the kernel on the ORACLE's matrix is bit-exact (tests/test_gpu_kernels.py); on the GPU's own matrix the
        # path must coincide too unless two cumulative costs tie to within f32 round-off
        ri, rj = cache["dtw_path"]
        ti, tj = paths[w]
        assert ti.tolist() == ri.tolist() and tj.tolist() == rj.tolist()


@pytest.mark.parametrize("name", ["tiny.en", "base.en"])
def test_score_f16_small_pass_on_dec_gemms_equals_general_path(name):
    # teacher-forced pass of ONE window (<= 160 rows) on the decode-step "dec" GEMMs (decoder_forward_dec: LayerNorm folded,
    # K / V scattered by the QKV epilogue at pos0 + token index) vs the per-op path (flag 1), both f16, on weights
    # with non-trivial LayerNorm gamma / beta; then both against the f32 oracle.  This is the pass align() runs per window.
    from stable_ts_amd import _lib
    from stable_ts_amd.engine import Engine, ModelDimensions
    lib = _lib.load()
    d = _dims(name)
    key = ("ej1", name)
    if key not in _CACHE:
        eng = Engine(ModelDimensions(**d.__dict__), dtype="f16", max_windows=1, max_rows=5)
        eng.load_state_dict(om.random_state_dict(d, 1234, 0.02, 3.0, 1.0, 0.1))
        _CACHE[key] = eng
    if ("oj", name) not in _CACHE:
        _CACHE[("oj", name)] = om.build_model(name, seed=1234, std=0.02, embed_gain=3.0, ln_jitter=0.1)
    eng, m = _CACHE[key], _CACHE[("oj", name)]
    heads_list = [tuple(p) for p in m.alignment_heads.indices().T.tolist()]
    eng.set_alignment_heads(heads_list)
    tok = get_tokenizer(False, num_languages=m.num_languages)
    mels = _mel(m.dims.n_mels, 67, B=1)
    g = torch.Generator().manual_seed(9)
    text = torch.randint(18, 50000, (97,), generator=g).tolist()
    wt, cache = ost.find_alignment(m, tok, text, mels[0], 480000, return_cache=True)
    toks = [[*tok.sot_sequence, tok.no_timestamps, *text, tok.eot]]
    xkv = eng.cross_kv(eng.encode(mels.cuda().contiguous()))
    old = lib.swx_debug_flags(-1)
    try:
        assert not (old & 1)
        p_fast, neg_fast, T = eng.score(xkv, toks, [1500], n_sot=len(tok.sot_sequence), eot=tok.eot)
        neg_fast = neg_fast.clone()
        lib.swx_debug_flags(old | 1)
        p_slow, neg_slow, _ = eng.score(xkv, toks, [1500], n_sot=len(tok.sot_sequence), eot=tok.eot)
        neg_slow = neg_slow.clone()
    finally:
        lib.swx_debug_flags(old)
    ref_p = np.asarray(cache["text_token_probs"])
    ref_neg = cache["neg_matrix"]
    n = T[0] + 1
    e_fast = (neg_fast[0, :n].cpu() - ref_neg).abs().max().item()
    e_slow = (neg_slow[0, :n].cpu() - ref_neg).abs().max().item()
    assert not torch.equal(neg_fast, neg_slow)          # the two paths really are different code
    assert e_fast < max(2.5 * e_slow, 0.05), (e_fast, e_slow)
    pe_fast = np.abs(np.asarray(p_fast[0]) - ref_p).max()
    pe_slow = np.abs(np.asarray(p_slow[0]) - ref_p).max()
    assert pe_fast < max(2.5 * pe_slow, 2e-2 * max(ref_p.max(), 1e-3)), (pe_fast, pe_slow)
