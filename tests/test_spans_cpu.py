"""Span-parallel transcription (stable_ts_amd/spans.py) on the CPU oracle stand-in.

Oracle: the reference's own ``transcribe()`` run once per span and concatenated -- exact by construction, because every
span runs the reference's sequential algorithm (seek from timestamp tokens, prompt carried over) and only the batching
on the device differs.  Needs /root/reference for the live comparison; the plan / merge helpers are tested without it.
"""
import multiprocessing as mp
import os
import sys
import warnings

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "golden"))
sys.path.insert(0, HERE)

from stable_ts_amd.spans import merge_span_results, plan_spans, transcribe_spans  # noqa: E402

HAVE_REF = os.path.isdir("/root/reference/stable_whisper")
BASE = dict(temperature=0.0, logprob_threshold=None, compression_ratio_threshold=None, no_speech_threshold=None, sample_len=36)


def _snap(res):
    out = []
    for s in res.segments:
        ws = None if not s.has_words else [(w.word, round(w.start, 3), round(w.end, 3), round(float(w.probability), 9), list(w.tokens)) for w in s.words]
        out.append((round(s.start, 3), round(s.end, 3), s.text, ws))
    return out


def test_plan_spans_cuts_in_quiet_places():
    g = torch.Generator().manual_seed(0)
    total = 16000 * 200
    audio = 0.2 * torch.randn(total, generator=g)
    quiet = [(16000 * 63, 16000 * 65), (16000 * 131, 16000 * 132 + 8000)]
    for a, b in quiet:
        audio[a:b] = 0
    plan = plan_spans(audio, 3)
    assert plan[0][0] == 0 and plan[-1][1] == total and all(a[1] == b[0] for a, b in zip(plan[:-1], plan[1:]))
    assert len(plan) == 3
    for (_, cut), (a, b) in zip(plan[:-1], quiet):
        assert a < cut < b                                      # nominal cuts at 66.7 s / 133.3 s moved into the gaps
    # no quiet place within reach: nominal cuts; more spans than windows: clipped; one span: the whole recording
    loud = 0.2 * torch.randn(total, generator=g)
    assert [c for _, c in plan_spans(loud, 4, search=2.0)][:-1] == [800000, 1600000, 2400000]
    assert len(plan_spans(loud[: 16000 * 70], 8)) == 2 and plan_spans(loud, 1) == [(0, total)]
    assert plan_spans(loud[:1000], 3) == [(0, 1000)]


def test_merge_span_results_offsets_and_orders():
    from stable_ts_amd.result import WhisperResult
    mk = lambda t0, txt: WhisperResult(dict(language="en", segments=[dict(start=t0, end=t0 + 1.0, text=txt, words=[
        dict(word=txt, start=t0, end=t0 + 1.0, probability=0.5, tokens=[1])])], nonspeech_sections=[dict(start=0.1, end=0.2)]))
    merged = merge_span_results([(32000, mk(0.5, " b")), (0, mk(0.25, " a"))])
    assert [(s.start, s.end, s.text) for s in merged.segments] == [(0.25, 1.25, " a"), (2.5, 3.5, " b")]
    assert merged.text == " a b" and merged.language == "en"
    assert [round(d["start"], 3) for d in merged.nonspeech_sections] == [0.1, 2.1]


@pytest.fixture(scope="module")
def models():
    import make_golden as G
    sw = G.import_reference()
    from oracle.whisper.model import build_model
    m = build_model("tiny.en", seed=1234, std=0.02, embed_gain=2.0, ts_gain=0.5)
    sw.modify_model(m)
    from oracle_engine import CpuWhisper
    return G, m, CpuWhisper(m)


CASES = {
    "defaults": dict(),
    "beam_no_condition": dict(beam_size=2, condition_on_previous_text=False),
    "prompt_no_silence": dict(initial_prompt=" aaat aaau", suppress_silence=False, regroup=False),
    "ts_tokens_skip": dict(suppress_ts_tokens=True, nonspeech_skip=0.4, regroup=False),
    "segment_level": dict(word_timestamps=False),
}


@pytest.mark.skipif(not HAVE_REF, reason="reference checkout not present")
@pytest.mark.parametrize("name", list(CASES))
def test_spans_equal_reference_per_span(models, monkeypatch, name):
    G, ref_model, mine = models
    from oracle_engine import install
    install(monkeypatch)
    opts = dict(BASE, **CASES[name])
    audio = torch.as_tensor(G.synth_audio(170.0, seed=31))
    plan = plan_spans(audio, 3)
    assert len(plan) == 3
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        parts = [(a, ref_model.transcribe(audio[a:b], language="en", verbose=None, ignore_compatibility=True, **opts)) for a, b in plan]
        got = transcribe_spans(mine, audio, 3, language="en", **opts)
    want = []
    for a, res in parts:
        res.offset_time(a / 16000)
        want.extend(_snap(res))
    assert _snap(got) == want and len(want) > 3
    assert got.text == "".join(r.text for _, r in parts)
    if opts.get("suppress_silence", True):
        ref_secs = [(round(d["start"] + a / 16000, 3), round(d["end"] + a / 16000, 3)) for a, r in parts for d in r.nonspeech_sections]
        assert [(round(d["start"], 3), round(d["end"], 3)) for d in got.nonspeech_sections] == ref_secs


@pytest.mark.skipif(not HAVE_REF, reason="reference checkout not present")
def test_spans_explicit_plan_and_errors(models, monkeypatch):
    G, ref_model, mine = models
    from oracle_engine import install
    install(monkeypatch)
    audio = torch.as_tensor(G.synth_audio(70.0, seed=5))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        one = transcribe_spans(mine, audio, spans=[(0, len(audio))], language="en", **BASE)
        whole = mine.transcribe(audio, language="en", **BASE)
    assert _snap(one) == _snap(whole)                            # a single span is the sequential driver itself
    with pytest.raises(NotImplementedError):
        transcribe_spans(mine, audio, 2, language="en", batch_size=2, **BASE)
    with pytest.raises(RuntimeError):
        transcribe_spans(mine, torch.zeros(0), 2, language="en", **BASE)


def _worker(rank, world, port, q, work_queue=True):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(4)
    from stable_ts_amd import parallel as par
    import stable_ts_amd.transcribe as T
    from make_golden import synth_audio
    from oracle.whisper.model import build_model
    from oracle_engine import CpuWhisper
    T._xkv_select = lambda model, xkv, idx: xkv.select(idx)          # oracle-backed stand-in for the GPU engine (tests only)
    par.init_from_env(backend="gloo")
    model = CpuWhisper(build_model("tiny.en", seed=1234, std=0.02, embed_gain=2.0, ts_gain=0.5))
    audio = torch.as_tensor(synth_audio(140.0, seed=9))
    kw = dict(language="en", sample_len=24, **{k: v for k, v in BASE.items() if k != "sample_len"})
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = par.transcribe_sharded(model, audio, mode="spans", spans_per_rank=2, work_queue=work_queue, lockstep=1, **kw)
        out = None if res is None else _snap(res)
        single = None
        if rank == 0:                                                 # the same plan on one rank
            single = _snap(transcribe_spans(model, audio, 4, **kw))
    par.barrier()
    q.put((rank, out, single))
    dist.destroy_process_group()


@pytest.mark.parametrize("work_queue", [True, False])
def test_sharded_spans_over_two_gloo_ranks(work_queue):
    """parallel.transcribe_sharded(mode='spans'), world_size 2: the ranks pull spans from the shared work queue (or take
    static runs), rank 0 gathers; equal to all four spans on one rank (and thereby to the reference per span, test above)
    whichever rank ran which span."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000) + (7 if work_queue else 0)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, work_queue)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=900) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, out0, single0), (_, out1, _) = got
    assert out1 is None and out0 is not None and len(out0) > 3
    assert out0 == single0
