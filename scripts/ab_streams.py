#!/usr/bin/env python
"""A/B of the headline pass (bench.py's default workload) with the batch driven as 1 / 2 / 3 / 4 stream lanes, interleaved in ONE
process on ONE box (boxes differ by +-4 %).  usage: python scripts/ab_streams.py [--lanes 1,2,3] [--rounds 3] [--kw k=v ...]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lanes", default="1,2,3")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--batch", type=int, default=20)
    ap.add_argument("--minutes", type=float, default=10.0)
    ap.add_argument("--embed-gain", type=float, default=3.0)
    ap.add_argument("--ts-gain", type=float, default=0.01)
    ap.add_argument("--mode", default="streams", help="streams (transcribe(streams=n)) | pipeline (transcribe(pipeline=n))")
    ap.add_argument("--flags", default="", help="A/B of swx_debug_flags values instead of lane counts, e.g. 0,262144")
    ap.add_argument("--phase", action="store_true", help="print transcribe.PHASE_TIMES of one extra pass per configuration")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    import stable_ts_amd as sw
    dev = "cuda:0"
    torch.cuda.set_device(0)
    dims = sw.dims_for("large-v3")
    model = sw.Whisper(dims, device=dev, dtype="f16", alignment_heads=bench.LARGE_V3_HEADS, max_windows=args.batch,
                       max_rows=args.batch * 5)
    sd = sw.random_state_dict(dims, seed=1234, std=0.02, **sw.BENCH_WEIGHTS)
    model.load_state_dict(sd)
    del sd
    audio = bench.synth_audio(args.minutes * 60.0, seed=0).to(dev)
    base = dict(language="en", temperature=0.0, logprob_threshold=None, compression_ratio_threshold=None,
                no_speech_threshold=None, beam_size=5, sample_len=112, min_tokens=112, word_timestamps=True, regroup=True,
                batch_size=args.batch, max_instant_words=1.0)
    lanes = [int(x) for x in args.lanes.split(",")]
    flags = [int(x) for x in args.flags.split(",")] if args.flags else None
    if flags is not None:
        lanes = flags
    ref_words = None
    times = {n: [] for n in lanes}
    for rnd in range(args.rounds + 1):
        for n in lanes:
            kw = dict(base)
            if flags is not None:
                model.engine.lib.swx_debug_flags(n)
            elif n > 1:
                kw["streams" if args.mode == "streams" else "pipeline"] = n
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            res = model.transcribe(audio, **kw)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            words = [(w.word, w.start, w.end) for w in res.all_words()]
            if ref_words is None:
                ref_words = words
            same = words == ref_words
            if rnd > 0:
                times[n].append(dt)
            print(f"round {rnd} lanes {n}: {1000 * dt:8.1f} ms  words {len(words)} identical_to_first {same}", flush=True)
    if args.phase:
        from stable_ts_amd import transcribe as _tr
        for n in lanes:
            if flags is not None:
                model.engine.lib.swx_debug_flags(n)
            _tr.PHASE_TIMES = {}
            model.transcribe(audio, **base)
            torch.cuda.synchronize()
            print("phase ms", n, {k: round(1000 * v, 2) for k, v in _tr.PHASE_TIMES.items()}, flush=True)
            _tr.PHASE_TIMES = None
    summary = {str(n): dict(median_ms=round(1000 * float(np.median(v)), 1), min_ms=round(1000 * min(v), 1),
                            x_real_time=round(args.minutes * 60.0 / float(np.median(v)), 1)) for n, v in times.items()}
    print(json.dumps(dict(mode=args.mode, batch=args.batch, minutes=args.minutes, lanes=summary)))
    if args.out:
        with open(args.out, "w") as f:
            json.dump(dict(mode=args.mode, batch=args.batch, minutes=args.minutes, lanes=summary), f, indent=1)


if __name__ == "__main__":
    main()
