"""bench.py's OWN N > 1 plumbing on two gloo ranks (CPU), so that the first real multi-GPU lease cannot die on
host code: RANK / WORLD_SIZE from the environment, the arena broadcast from rank 0, `mark_weights_loaded` on the others, the barriers
around the timed region, `max_over_ranks`, `gather_results` into `config.per_rank`, ONE JSON line from rank 0 with the whole-job
value.  The device engine is replaced by the oracle-backed stand-in of tests/oracle_engine.py (test infrastructure); nothing else of
bench.py is patched except the three torch.cuda calls that need a GPU.  No scaling number comes out of this: DESIGN.md keeps the
sentence "no scaling curve was measured"."""
import json
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rank(rank, world, port, q, mode):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    import io
    import contextlib
    import stable_ts_amd as sw
    import stable_ts_amd.transcribe as T
    from oracle.whisper.model import build_model
    from oracle_engine import CpuWhisper
    import bench

    T._xkv_select = lambda model, xkv, idx: xkv.select(idx)
    calls = dict(mark=0, load=0)

    class Lib:
        def swx_debug_flags(self, v):
            return 0

    def make(dims, device=None, dtype="f16", alignment_heads=None, max_windows=1, max_rows=1):
        m = CpuWhisper(build_model("tiny.en", seed=1234, std=0.02, embed_gain=2.0, ts_gain=0.5))
        m.engine.lib = Lib()
        # a recognisable arena: rank 0 holds the bytes, the others must receive them through bench.py's broadcast
        m.engine.arena = torch.arange(4096, dtype=torch.uint8) if rank == 0 else torch.zeros(4096, dtype=torch.uint8)
        m.engine.mark_weights_loaded = lambda: calls.__setitem__("mark", calls["mark"] + 1)
        m.engine.graph_stats = lambda: dict(captures=0, replays=0, eager_steps=0, fell_back=False)
        m.load_state_dict = lambda sd: calls.__setitem__("load", calls["load"] + 1)
        m.transcribe_spans = None
        made.append(m)
        return m

    made = []
    sw.Whisper = make
    sw.random_state_dict = lambda *a, **k: {}                     # the stand-in carries its own weights
    torch.cuda.set_device = lambda *_: None
    torch.cuda.synchronize = lambda *_: None
    torch.cuda.empty_cache = lambda: None
    sys.argv = ["bench.py", "--gpus", str(world), "--steps", "1", "--warmup", "1", "--model", "tiny.en", "--minutes", "1", "--batch", "2",
                "--beam", "1", "--tokens", "6", "--host-audio", "--no-roofline", "--no-cpu-baseline", "--no-f32", "--mode", mode]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main()
    q.put((rank, buf.getvalue(), calls, bool(torch.equal(made[0].engine.arena, torch.arange(4096, dtype=torch.uint8)))))


@pytest.mark.parametrize("mode", ["transcribe", "sharded"])
def test_bench_two_ranks_dry_run(mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000) + (41 if mode == "sharded" else 0)
    procs = [ctx.Process(target=_rank, args=(r, 2, port, q, mode)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, out0, calls0, arena0), (_, out1, calls1, arena1) = got
    assert out1.strip() == "", out1                                 # only rank 0 prints
    lines = [l for l in out0.splitlines() if l.strip()]
    assert len(lines) == 1, out0                                    # exactly one JSON line
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 1 and rec["warmup"] == 1 and rec["scaling"] == "weak" and rec["higher_is_better"] is True
    assert rec["vs_baseline"] is None and rec["data"] == "synthetic" and rec["unit"] == "x real time"
    # whole-job value: both ranks' minutes over the slowest rank's wall time
    assert abs(rec["value"] - 2 * 1 * 60.0 / (rec["ms_per_step"] / 1000.0)) <= 0.02 * rec["value"], rec
    assert "dp2" in rec["config"]["parallelism"]
    if mode == "transcribe":
        per = rec["config"]["per_rank"]
        assert [p["rank"] for p in per] == [0, 1] and all(p["segments"] >= 1 for p in per), per
    else:
        assert rec["config"]["windows_per_gpu"] == 4                # the gathered result covers the recording of both ranks
    assert arena0 and arena1                                        # the broadcast reached rank 1
    assert calls0 == dict(mark=0, load=1) and calls1 == dict(mark=1, load=0), (calls0, calls1)
