#!/usr/bin/env python
"""Device-level A/B of the decode loop of the headline workload (20 windows x 5 beams x 112 steps, large-v3 f16): ONE lockstep
job of 20 windows against 2 / 4 concurrent jobs of 10 / 5 windows on engine clones (shared weights, own workspace, own HIP stream,
own captured step graphs), each driven by its own host thread that sits inside swx_decode (ctypes releases the GIL; the loop is
one hipGraphLaunch per two steps).  Isolates what overlapping half-batches buys on the device -- one half's HBM-bound
cross-attention under the other half's latency-bound GEMM chain -- from everything the Python host does.
usage: python scripts/ab_decode_lanes.py [--lanes 1,2,4] [--reps 3]"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lanes", default="1,2,4")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--windows", type=int, default=20)
    ap.add_argument("--steps", type=int, default=112)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    import stable_ts_amd as sw
    from stable_ts_amd.decoding import DecodingOptions, DecodingPlan
    from stable_ts_amd.transcribe import _xkv_select
    dev = "cuda:0"
    W = args.windows
    dims = sw.dims_for("large-v3")
    model = sw.Whisper(dims, device=dev, dtype="f16", alignment_heads=bench.LARGE_V3_HEADS, max_windows=W, max_rows=W * 5)
    model.load_state_dict(sw.random_state_dict(dims, seed=1234, std=0.02, **sw.BENCH_WEIGHTS))
    audio = bench.synth_audio(30.0 * W, seed=0).to(dev)
    mel = model.log_mel_batch([audio[i * 480000:(i + 1) * 480000] for i in range(W)], [0] * W)
    xkv = model.cross_kv(model.encoder(mel))
    torch.cuda.synchronize()
    opts = DecodingOptions(language="en", beam_size=5, sample_len=args.steps, min_tokens=args.steps, max_initial_timestamp=None)
    plan = DecodingPlan(model, opts)
    kw = plan.engine_kwargs()
    init = list(plan.initial_tokens)
    lanes_list = [int(x) for x in args.lanes.split(",")]
    max_l = max(lanes_list)
    engines = [model.engine] + [model.engine.clone_shared(max_windows=(W + 1) // 2, max_rows=((W + 1) // 2) * 5) for _ in range(max_l - 1)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(max_l)]
    ref = None
    summary = {}
    for n in lanes_list:
        per = (W + n - 1) // n
        parts = [list(range(k * per, min(W, (k + 1) * per))) for k in range(n)]
        subs = [_xkv_select(model, xkv, p) for p in parts]
        torch.cuda.synchronize()
        outs = [None] * n

        def run(k):
            with torch.cuda.stream(streams[k]):
                outs[k] = engines[k].decode(subs[k], [init] * len(parts[k]), **kw)
            streams[k].synchronize()

        ts = []
        for rep in range(args.reps + 1):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if n == 1:
                outs[0] = engines[0].decode(subs[0], [init] * W, **kw)
            else:
                th = [threading.Thread(target=run, args=(k,)) for k in range(n)]
                for t in th:
                    t.start()
                for t in th:
                    t.join()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            if rep > 0:
                ts.append(dt)
        toks = np.concatenate([o["tokens"] for o in outs], 0)
        slp = np.concatenate([o["sum_logprobs"] for o in outs], 0)
        if ref is None:
            ref = (toks, slp)
        same_t = bool((toks == ref[0]).all())
        same_s = bool((slp == ref[1]).all())
        summary[str(n)] = dict(ms=[round(1000 * t, 1) for t in ts], median_ms=round(1000 * float(np.median(ts)), 1),
                               tokens_identical_to_single_job=same_t, sum_logprobs_bit_identical=same_s,
                               max_abs_dsumlp=float(np.abs(slp - ref[1]).max()))
        print(n, summary[str(n)], flush=True)
    print(json.dumps(summary))
    if args.out:
        with open(args.out, "w") as f:
            json.dump(summary, f, indent=1)


if __name__ == "__main__":
    main()
