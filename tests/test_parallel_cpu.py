"""world_size-2 gloo tests (CPU) of the multi-GPU host logic: window sharding, weight-arena broadcast, result gather."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from stable_ts_amd import parallel as par
    r, _, w = par.init_from_env(backend="gloo")
    arena = torch.arange(1000, dtype=torch.uint8) if r == 0 else torch.zeros(1000, dtype=torch.uint8)
    par.broadcast_arena(arena, src=0)
    ok_arena = bool(torch.equal(arena, torch.arange(1000, dtype=torch.uint8)))
    mine = par.shard_windows(7, r, w)
    recs = [dict(window=i, words=[dict(word=f"w{i}", start=30.0 * i, end=30.0 * i + 1)]) for i in mine]
    allr = par.gather_results(recs, dst=0)
    mx = par.max_over_ranks(float(r + 1))
    par.barrier()
    q.put((r, ok_arena, list(mine), None if allr is None else [x["window"] for x in allr], mx))
    dist.destroy_process_group()


def test_two_rank_gloo_roundtrip():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, ok0, m0, all0, mx0), (r1, ok1, m1, all1, mx1) = out
    assert ok0 and ok1
    assert m0 == [0, 1, 2, 3] and m1 == [4, 5, 6]
    assert all0 == [0, 1, 2, 3, 4, 5, 6] and all1 is None
    assert mx0 == mx1 == 2.0


def test_shard_windows_partition():
    from stable_ts_amd.parallel import shard_windows
    for n in (0, 1, 7, 8, 20, 961):
        for world in (1, 2, 4, 8):
            parts = [list(shard_windows(n, r, world)) for r in range(world)]
            flat = [i for p in parts for i in p]
            assert flat == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def _sharded_worker(rank, world, port, q, multilingual=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(4)
    from stable_ts_amd import parallel as par
    import stable_ts_amd.transcribe as T
    from make_golden import synth_audio
    from oracle.whisper.model import build_model
    from oracle_engine import CpuWhisper
    T._xkv_select = lambda model, xkv, idx: xkv.select(idx)          # oracle-backed stand-in for the GPU engine (tests only)
    par.init_from_env(backend="gloo")
    model = CpuWhisper(build_model("tiny" if multilingual else "tiny.en", seed=1234, std=0.02, embed_gain=2.0, ts_gain=0.5))
    audio = synth_audio(100.0, seed=3)                                # 4 windows: ranks take [0, 1] and [2, 3]
    kw = dict(language="en", temperature=0.0, logprob_threshold=None, compression_ratio_threshold=None,
              no_speech_threshold=None, sample_len=24, regroup=False)
    if multilingual:
        # no language given, the first 33 s silent: the language must be settled on the first window the single-rank run
        # decodes (window 1 here), on every rank alike; the default regrouping runs once on the gathered result
        audio = audio.clone()
        audio[: 33 * 16000] = 0.0
        kw.update(language=None, regroup=True)
    res = par.transcribe_sharded(model, audio, batch_size=2, **kw)
    out = None
    if res is not None:
        out = [(s.start, s.end, s.text, [(w.word, w.start, w.end) for w in s.words]) for s in res.segments]
    single = None
    if rank == 0:                                                     # the same recording on one rank, same window-parallel mode
        one = model.transcribe(audio, batch_size=2, **kw)
        single = [(s.start, s.end, s.text, [(w.word, w.start, w.end) for w in s.words]) for s in one.segments]
    par.barrier()
    q.put((rank, out, single))
    dist.destroy_process_group()


@pytest.mark.parametrize("multilingual", [False, True])
def test_sharded_transcribe_equals_single_rank(multilingual):
    """parallel.transcribe_sharded over 2 gloo ranks == the window-parallel transcribe of the whole recording on one rank
    (windows are independent in that mode); runs the real host pipeline on the oracle-backed CPU stand-in.
    multilingual: no language given + leading silence + default regrouping (the language is settled per rank on the window
    the single-rank run settles it on; regrouping runs once on rank 0 after the gather)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000) + (37 if multilingual else 0)
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, q, multilingual)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=600) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, out0, single0), (_, out1, _single1) = got
    assert out1 is None and out0 is not None and len(out0) > 0
    assert out0 == single0
