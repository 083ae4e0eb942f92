"""The multi-GPU code path on the REAL backend with the hardware that is reachable: one rank.  `SWX_FORCE_DIST=1` makes
parallel.init_from_env() join a world of size 1 on the "nccl" backend (= RCCL on ROCm), so bench.py's RCCL broadcast of the
packed weight arena, the barrier, the max-over-ranks all-reduce and the result gather all execute on the device backend
instead of being skipped.  (N > 1 cannot be launched from this harness; the N = 2 logic is covered on CPU with gloo:
tests/test_parallel_cpu.py, tests/test_spans_cpu.py.)"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_single_rank_rccl_in_subprocess():
    env = dict(os.environ, SWX_FORCE_DIST="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(29600 + os.getpid() % 300), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--model", "tiny.en", "--minutes", "1",
                        "--batch", "2", "--steps", "1", "--warmup", "0", "--tokens", "16", "--beam", "5", "--no-cpu-baseline",
                        "--no-roofline", "--no-f32"], capture_output=True, text=True, timeout=400, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-2500:])
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 1 and out["n_gpus_measured"] == 1 and out["value"] > 0
    assert out["config"]["per_rank"] == [dict(rank=0, segments=out["config"]["segments"], words=out["config"]["words"])]


@pytest.mark.parametrize("name,dtype", [("base.en", "f16"), ("large-v3", "f16"), ("tiny.en", "f32")])
def test_receiving_rank_path_equals_the_rank_that_loaded(name, dtype):
    """Ranks != 0 never see a state dict: they receive rank 0's packed arena (parallel.broadcast_arena = one RCCL broadcast) and call
    Engine.mark_weights_loaded(), which takes every tensor as present and re-runs the load-time preparation (bench.py, main()).  Seven of
    the eight ranks of an 8-GPU run work that way, and no harness here can launch them -- so the path is emulated inside one process:
    model B's arena is a device copy of model A's (what the broadcast delivers), then both transcribe the same recording.  Everything a
    rank computes must be equal bit for bit: tokens, word times, probabilities."""
    import gc

    import torch

    import bench
    import stable_ts_amd as sw
    dims = sw.dims_for(name)
    sd = sw.random_state_dict(dims, seed=1234, std=0.02, **sw.BENCH_WEIGHTS)
    from stable_ts_amd.model import OFFICIAL_ALIGNMENT_HEADS
    kw_model = dict(device="cuda:0", dtype=dtype, max_windows=4, max_rows=20, alignment_heads=OFFICIAL_ALIGNMENT_HEADS[name])
    a = sw.Whisper(dims, **kw_model)
    a.load_state_dict(sd)
    del sd
    b = sw.Whisper(dims, **kw_model)
    assert b.engine.arena.shape == a.engine.arena.shape and b.engine.arena.dtype == a.engine.arena.dtype
    b.engine.arena.copy_(a.engine.arena)
    torch.cuda.synchronize()
    b.engine.mark_weights_loaded()
    audio = bench.synth_audio(100.0, seed=1).cuda()          # 4 windows, the last one ragged
    # (1) the device path stage by stage (any weights give tokens here): encoder output, every beam's tokens, scores, words
    from stable_ts_amd.decoding import DecodingOptions, DecodingPlan
    from stable_ts_amd.timing import AlignmentJob, find_alignment_batch
    import numpy as np

    def run(model):
        wins = [audio[i * 480000:(i + 1) * 480000].contiguous() for i in range(3)]
        plan = DecodingPlan(model, DecodingOptions(language="en", beam_size=5, sample_len=24, min_tokens=24, max_initial_timestamp=None))
        xa = model.encoder(model.log_mel_batch(wins, [0] * 3))
        xkv = model.cross_kv(xa)
        out = model.engine.decode(xkv, [list(plan.initial_tokens)] * 3, **plan.engine_kwargs())
        res = plan.results(out, [None] * 3, ["en"] * 3)
        tok = plan.tokenizer
        jobs = [AlignmentJob(tok, [t for t in r.tokens if t < tok.eot] or [1000, 2000], 480000) for r in res]
        words = find_alignment_batch(model, jobs, xkv, return_debug=True)
        return xa.clone(), out, [[(w.word, w.start, w.end, w.probability) for w in ws] for ws in words]
    xa_a, out_a, words_a = run(a)
    xa_b, out_b, words_b = run(b)
    assert torch.equal(xa_a, xa_b)
    for key in ("tokens", "lens", "sum_logprobs", "no_speech_prob"):
        assert np.array_equal(np.asarray(out_a[key]), np.asarray(out_b[key])), key
    assert words_a == words_b and all(len(w) >= 1 for w in words_a)
    # (2) transcribe() end to end (the random-weight recipe is shaped for large-v3: other sizes may yield no text, equal on both sides)
    kw = dict(language="en", temperature=0.0, beam_size=5, sample_len=24, min_tokens=24, word_timestamps=True, batch_size=4,
              logprob_threshold=None, compression_ratio_threshold=None, no_speech_threshold=None)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ra, rb = a.transcribe(audio, **kw), b.transcribe(audio, **kw)
    assert [s.tokens for s in ra.segments] == [s.tokens for s in rb.segments]
    wa = [(w.word, w.start, w.end, w.probability) for w in ra.all_words()]
    wb = [(w.word, w.start, w.end, w.probability) for w in rb.all_words()]
    assert wa == wb
    if name == "large-v3":
        assert len(ra.segments) >= 1 and len(wa) >= 1
    del a, b
    gc.collect()
    torch.cuda.empty_cache()
