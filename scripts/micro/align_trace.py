"""Wraps the engine entry points with a synchronise + wall clock and runs align() on 5 minutes of audio at large-v3 dims:
per-call times of every stage in the order they happen (diagnosis of the align-mode encoder stage)."""
import os, sys, time, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import stable_ts_amd as sw
import importlib
bench = importlib.import_module("bench")

def main():
    dims = sw.dims_for("large-v3")
    model = sw.Whisper(dims, device="cuda:0", dtype="f16", alignment_heads=bench.LARGE_V3_HEADS, max_windows=1, max_rows=1)
    model.load_state_dict(sw.random_state_dict(dims, seed=1234, std=0.02, embed_gain=3.0, ts_gain=0.01))
    audio = bench.synth_audio(300.0, seed=0).to("cuda:0")
    g = torch.Generator().manual_seed(7)
    toks = torch.randint(18, 50000, (750,), generator=g).tolist()
    eng = model.engine
    log = collections.defaultdict(list)
    def wrap(name):
        fn = getattr(eng, name)
        def w(*a, **k):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            if os.environ.get("TRACE_VERBOSE"):
                print("->", name, [tuple(x.shape) if torch.is_tensor(x) else (len(x) if hasattr(x, "__len__") else x) for x in a][:4], flush=True)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn(*a, **k)
            e1.record()
            t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
            log[name].append(((t1 - t0) * 1e3, (t2 - t0) * 1e3, e0.elapsed_time(e1)))
            return r
        setattr(eng, name, w)
    for n in ("log_mel", "encode", "cross_kv", "score", "dtw"):
        wrap(n)
    import gc
    for rep in range(3):
        log.clear()
        if rep == 2:
            gc.disable()
        st0 = torch.cuda.memory_stats()
        t0 = time.perf_counter()
        model.align(audio, list(toks), language="en", token_step=100)
        torch.cuda.synchronize()
        st1 = torch.cuda.memory_stats()
        print(f"pass {rep}: {time.perf_counter() - t0:.3f} s   segments allocated {st1['segment.all.allocated'] - st0['segment.all.allocated']} "
              f"freed {st1['segment.all.freed'] - st0['segment.all.freed']}  alloc retries {st1['num_alloc_retries'] - st0['num_alloc_retries']} "
              f"reserved {st1['reserved_bytes.all.current'] / 1e9:.2f} GB  gc enabled {gc.isenabled()} counts {gc.get_count()}")
    for n, v in log.items():
        host = sorted(x[0] for x in v); tot = sorted(x[1] for x in v)
        print(f"{n:10s} calls {len(v):3d}  host enqueue median {host[len(host)//2]:7.2f} ms  call+sync median {tot[len(tot)//2]:7.2f} ms  max {tot[-1]:7.2f}  sum {sum(tot):8.1f} ms")
    print("encode calls (host enqueue, call+sync wall, device by events) ms:", [(round(a, 2), round(b, 2), round(c, 2)) for a, b, c in log["encode"]])

if __name__ == "__main__":
    main()
