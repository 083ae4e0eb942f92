"""fp16 -- the dtype bench.py times and the reference itself runs on a GPU (whisper_word_level/original_whisper.py:250-259) --
pinned against the f32 CPU ORACLE at the FULL depth of the benchmarked model (large-v3: 32 + 32 layers, d = 1280, 20 heads,
128 mels, 51 866 tokens) ON THE WEIGHTS bench.py TIMES (stable_ts_amd.BENCH_WEIGHTS -- one recipe for the benchmark and for
these tests since round 4: token-embedding gain 9, cross-attention score gain 8, LayerNorm jitter 0.1, timestamp rows x0.1)
and AT THE BENCHMARK'S LENGTH (112 decode steps, beam 5: VERDICT r3 item 1).  Rounding accumulates over 64 layers and over
112 steps of beam bookkeeping; the 2-layer tests of test_gpu_largev3.py cannot see that.

ASSERTED at BASELINE.json's north-star tolerances on a synthetic spectrogram: token ids identical (greedy and beam 5; 24 tokens
and 112 tokens), |avg_logprob difference| <= 1e-3, then the word-timestamp stage (swx_score + swx_align + swx_dtw;
timing.py:202-306) on the oracle's transcript and on a 100-token random text: every word start / end within +-20 ms of the
oracle's, per-token log-probabilities at the bar fp16 storage supports (profiles/r04_f16_error_budget.json says where the error
comes from).  The benchmark's OWN audio (windows 0 / 7 / 19 of the timed recording, which tests/test_gpu_batch_invariance.py
chains to the 20-window x 5-beam launch shapes) is held to the oracle in tests/test_gpu_f16_bench_windows.py (round 5).
Every case writes its numbers to gpurun_out/f16_depth_report.json BEFORE asserting (copied to profiles/ per round).
"""
import gc
import json
import os

import numpy as np
import pytest
import torch

from oracle import stable as ost
from oracle.whisper import model as om
from oracle.whisper.decoding import DecodingOptions
from oracle.whisper.tokenizer import get_tokenizer

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADS = ((7, 0), (10, 17), (12, 18), (13, 12), (16, 1), (17, 14), (19, 11), (21, 4), (24, 1), (25, 6))   # large-v3's
import stable_ts_amd as _sw
WEIGHTS = {"sharp": dict(_sw.BENCH_WEIGHTS)}        # "sharp" = the benchmark's recipe (bench.py defaults)
_STATE = {}


def _report(name, payload):
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    path = os.path.join(ROOT, "gpurun_out", "f16_depth_report.json")
    try:
        with open(path) as f:
            data = json.load(f)
    except Exception:
        data = {}
    data[name] = payload
    with open(path, "w") as f:
        json.dump(data, f, indent=1)


def _setup(kind):
    """oracle (f32, CPU) + engine (f16, GPU) of the full-depth model on one set of weights; one set lives at a time (6 GB each)"""
    if _STATE.get("kind") == kind:
        return _STATE
    _STATE.clear()
    gc.collect()
    torch.cuda.empty_cache()
    from stable_ts_amd.engine import Engine, ModelDimensions
    import stable_ts_amd as sw
    dims = om.dims_for("large-v3")
    sd = om.random_state_dict(dims, 1234, 0.02, **WEIGHTS[kind])
    m = om.Whisper(dims)
    m.load_state_dict(sd)
    m.eval()
    mask = torch.zeros(dims.n_text_layer, dims.n_text_head, dtype=torch.bool)
    for l, h in HEADS:
        mask[l, h] = True
    m.set_alignment_heads_mask(mask)
    eng = Engine(ModelDimensions(**dims.__dict__), dtype="f16", max_windows=1, max_rows=5, alignment_heads=HEADS)
    eng.load_state_dict(sd)
    del sd
    gc.collect()
    _STATE.update(kind=kind, oracle=m, engine=eng, model=sw.Whisper.from_engine(eng), inputs={}, dims=dims,
                  tok=get_tokenizer(True, num_languages=m.num_languages, language="en", task="transcribe"))
    return _use_input(_STATE, "tone")


def _use_input(st, name):
    """the window the cases run on.  "tone": a synthetic spectrogram (sinusoid pattern + noise; rounds 2-3).  The BENCHMARK's own
    audio is tests/test_gpu_f16_bench_windows.py's subject (windows 0 / 7 / 19 of the timed recording, round 5).  Encoder output /
    cross-K/V of both sides are cached per input."""
    if name not in st["inputs"]:
        dims = st["dims"]
        assert name == "tone"
        g = torch.Generator().manual_seed(7)
        t = torch.linspace(0, 1, 3000)
        base = torch.sin(t[None, :] * (5 + torch.arange(128)[:, None] * 0.37)) * 0.5
        mel = (base + 0.3 * torch.randn(128, 3000, generator=g)).float()
        with torch.no_grad():
            xa_ref = st["oracle"].encoder(mel[None])
        xa = st["engine"].encode(mel[None].cuda().contiguous())
        st["inputs"][name] = dict(mel=mel, xa_ref=xa_ref, xa=xa, xkv=st["engine"].cross_kv(xa))
    st.update(st["inputs"][name], input=name)
    return st


def _tok_cfg(tok, task):
    return dict(eot=tok.eot, sot=tok.sot, no_timestamps=tok.no_timestamps, timestamp_begin=tok.timestamp_begin,
                no_speech=tok.no_speech, blank_token=tok.encode(" ")[0], suppress_tokens=list(task._get_suppress_tokens()))


def _decode_both(st, beam, n):
    m, eng = st["oracle"], st["engine"]
    o = dict(language="en", sample_len=n)
    if beam:
        o["beam_size"] = beam
    options = DecodingOptions(fp16=False, max_initial_timestamp=None, **o)
    ref, _ = ost.decode_stable(m, st["mel"], options, audio_features=st["xa_ref"], min_tokens=n)
    task = ost.DecodingTaskStable(m, options)
    out = eng.decode(st["xkv"], [list(task.initial_tokens)], n_group=task.n_group, beam=beam is not None, patience=None,
                     sample_len=n, sot_index=task.sot_index, min_tokens=n, **_tok_cfg(task.tokenizer, task))
    sb = out["sample_begin"]
    scores = [(-np.inf if int(out["lens"][0, k]) <= 0 else out["sum_logprobs"][0, k] / int(out["lens"][0, k]))
              for k in range(out["tokens"].shape[1])]
    best = int(np.argmax(scores))
    toks = out["tokens"][0, best, sb: sb + int(out["lens"][0, best])].tolist()
    return ref, toks, float(out["sum_logprobs"][0, best]) / (len(toks) + 1), float(out["no_speech_prob"][0])


def _words_both(st, text, num_samples=480000):
    """word-timestamp stage on a given text: oracle (find_alignment, timing.py:202-306) vs the device path"""
    from stable_ts_amd.timing import AlignmentJob, find_alignment_batch
    m, tok = st["oracle"], st["tok"]
    ref_words, cache = ost.find_alignment(m, tok, list(text), st["mel"], num_samples, audio_features=st["xa_ref"], return_cache=True)
    job = AlignmentJob(tok, list(text), num_samples)
    words = find_alignment_batch(st["model"], [job], st["xkv"], return_debug=True)[0]
    ri, rj = cache["dtw_path"]
    ti, tj = job.debug["path"]
    p_ref = np.asarray(cache["text_token_probs"], dtype=np.float64)
    p_got = np.asarray(job.debug["token_probs"], dtype=np.float64)[:len(p_ref)]
    mid = (p_ref > 1e-30) & (p_ref < 0.99)                       # unsaturated probabilities only (p ~ 1 hides any error)
    dt = np.asarray([(abs(a.start - b.start), abs(a.end - b.end)) for a, b in zip(words, ref_words)])
    # frames by which the two DTW paths differ, per text-token row (first frame of each row)
    first = lambda i, j: {int(r): int(c) for r, c in reversed(list(zip(i.tolist(), j.tolist())))}
    fa, fb = first(ti, tj), first(ri, rj)
    # how much worse the device's path is than the oracle's optimum ON THE ORACLE'S OWN cost matrix: where fp16 moves a path, it moves
    # it between alternatives of (nearly) equal cost -- a near-tie of the DTW, not a different alignment
    neg = cache["neg_matrix"].double().numpy()
    cost_ref, cost_got = float(neg[ri, rj].sum()), float(neg[ti, tj].sum())
    rep = dict(path_cost_oracle=cost_ref, path_cost_device_path_on_oracle_matrix=cost_got,
               path_cost_gap_rel=(cost_got - cost_ref) / abs(cost_ref),
               words=len(ref_words), text_tokens=len(text), same_word_split=[w.word for w in words] == [w.word for w in ref_words],
               dtw_path_identical=bool(ti.tolist() == ri.tolist() and tj.tolist() == rj.tolist()),
               dtw_row_start_max_frame_diff=int(max(abs(fa[r] - fb[r]) for r in fb)),
               within_20ms=float(((dt[:, 0] <= 0.0201) & (dt[:, 1] <= 0.0201)).mean()), max_dt=float(dt.max()),
               max_dprob=float(np.abs(p_got - p_ref).max()),
               unsaturated_tokens=int(mid.sum()),
               max_dlogprob_unsaturated=float(np.abs(np.log(p_got[mid]) - np.log(p_ref[mid])).max()) if mid.any() else None,
               max_dlogprob_over_tol=float((np.abs(np.log(p_got[mid]) - np.log(p_ref[mid])) /
                                            (2e-2 + 1e-3 * np.abs(np.log(p_ref[mid])))).max()) if mid.any() else None,
               prob_range=(float(p_ref.min()), float(p_ref.max())))
    return rep


@pytest.mark.parametrize("beam", [None, 5])
def test_full_depth_f16_decode_vs_oracle_sharp(beam):
    st = _use_input(_setup("sharp"), "tone")
    ref, toks, avg_lp, nsp = _decode_both(st, beam, 24)
    same = sum(1 for a, b in zip(toks, ref.tokens) if a == b)
    rep = dict(tokens=len(ref.tokens), same=same, d_avg_logprob=abs(avg_lp - ref.avg_logprob), avg_logprob=(avg_lp, ref.avg_logprob),
               no_speech=(nsp, ref.no_speech_prob), text_tokens=sum(1 for t in ref.tokens if t < st["tok"].eot))
    _report(f"sharp/decode[beam={beam}]", rep)
    assert toks == ref.tokens, rep
    assert rep["d_avg_logprob"] <= 1e-3, rep                     # north star: logprobs within 1e-3
    assert abs(nsp - ref.no_speech_prob) <= 1e-4 + 5e-2 * ref.no_speech_prob, rep
    _STATE[f"ref_tokens_{beam}"] = list(ref.tokens)


def test_full_depth_f16_encoder_vs_oracle_sharp():
    st = _use_input(_setup("sharp"), "tone")
    ref = st["xa_ref"][0]
    got = st["xa"][0].float().cpu()
    err = (got - ref).abs().max().item()
    rel = err / max(1.0, ref.abs().max().item())
    _report("sharp/encoder", dict(max_abs_err=err, ref_absmax=ref.abs().max().item(), rel=rel,
                                  rms_err=float((got - ref).pow(2).mean().sqrt()), rms_ref=float(ref.pow(2).mean().sqrt())))
    assert rel < 2e-2        # fp16 storage over 32 layers (strict f32: 5e-4, test_gpu_largev3.py)


def test_full_depth_f16_words_vs_oracle_sharp():
    st = _use_input(_setup("sharp"), "tone")
    tok = st["tok"]
    texts = {}
    dec = _STATE.get("ref_tokens_5") or _STATE.get("ref_tokens_None")
    if dec:
        t = [x for x in dec if x < tok.eot]
        if len(t) >= 4:
            texts["oracle's decoded tokens"] = t
    g = torch.Generator().manual_seed(5)
    texts["100 random text tokens"] = torch.randint(18, 50000, (100,), generator=g).tolist()
    reps = {}
    for name, text in texts.items():
        reps[name] = _words_both(st, text)
    _report("sharp/words", reps)
    for name, rep in reps.items():
        assert rep["same_word_split"], (name, rep)
        assert rep["within_20ms"] == 1.0 and rep["max_dt"] <= 0.0201, (name, rep)     # north star: every word within +-20 ms
        if rep["max_dlogprob_over_tol"] is not None:
            # fp16 storage of the hidden states: ~1e-3 relative per logit -> |delta log p| <= 2e-2 + 1e-3 |log p| (measured: 0.038 at
            # p = 5e-20, 0.0065 for the decoded tokens whose p > 0.05)
            assert rep["max_dlogprob_over_tol"] <= 1.0, (name, rep)


@pytest.mark.parametrize("beam", [None, 5])
def test_full_depth_f16_decode_112_steps_vs_oracle(beam):
    """the benchmark's decode length: 112 steps (greedy, and the timed beam 5) on the benchmark's weights"""
    st = _use_input(_setup("sharp"), "tone")
    ref, toks, avg_lp, nsp = _decode_both(st, beam, 112)
    n_same = 0
    for a, b in zip(toks, ref.tokens):
        if a != b:
            break
        n_same += 1
    rep = dict(tokens=len(ref.tokens), identical_prefix=n_same, d_avg_logprob=abs(avg_lp - ref.avg_logprob),
               avg_logprob=(avg_lp, ref.avg_logprob), text_tokens=sum(1 for t in ref.tokens if t < st["tok"].eot))
    _report(f"sharp/decode112[beam={beam}]", rep)
    assert len(ref.tokens) == 112 and rep["text_tokens"] >= 100, rep
    assert toks == ref.tokens, rep                               # north star: identical token ids
    assert rep["d_avg_logprob"] <= 1e-3, rep                     # north star: logprobs within 1e-3
    _STATE[f"ref_tokens112_{beam}"] = list(ref.tokens)


@pytest.fixture(scope="module", autouse=True)
def _release_models():
    yield
    _STATE.clear()
    gc.collect()
    torch.cuda.empty_cache()
