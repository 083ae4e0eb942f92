"""A/B timing of the decode-step experiment switches (SWX_FLAG_* in csrc/swx_kernels.h) on the bench workload.

    python scripts/tune_flags.py [--flags 0,2,4,8,16,30] [--passes 2]     (one GPU; prints one line per flag value)

SWX_PG_BLOCKS (split-K workgroup target of the decode GEMM) is read once per process: set it in the environment.
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--flags", default="0,2,4,8,16,30")
    ap.add_argument("--passes", type=int, default=2)
    ap.add_argument("--minutes", type=float, default=10.0)
    args = ap.parse_args()
    import stable_ts_amd as sw
    dims = sw.dims_for("large-v3")
    model = sw.Whisper(dims, device="cuda:0", dtype="f16", alignment_heads=bench.LARGE_V3_HEADS, max_windows=20, max_rows=100)
    model.load_state_dict(sw.random_state_dict(dims, seed=1234, std=0.02, embed_gain=2.0, ts_gain=0.5))
    audio = bench.synth_audio(args.minutes * 60.0, seed=0).to("cuda:0")
    kw = dict(language="en", temperature=0.0, logprob_threshold=None, compression_ratio_threshold=None,
              no_speech_threshold=None, beam_size=5, sample_len=112, min_tokens=112, word_timestamps=True, regroup=False,
              batch_size=20)
    lib = model.engine.lib
    ref_tokens = None
    best = None
    model.transcribe(audio, **kw)                                   # warm-up (allocations, first-touch)
    for f in [int(x) for x in args.flags.split(",")]:
        lib.swx_debug_flags(f)
        res = model.transcribe(audio, **kw)
        toks = [t for s in res.segments for t in s.tokens]
        if ref_tokens is None:
            ref_tokens = toks
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.passes):
            model.transcribe(audio, **kw)
        torch.cuda.synchronize()
        ms = 1000.0 * (time.perf_counter() - t0) / args.passes
        if toks == ref_tokens and (best is None or ms < best[1]):
            best = (f, ms)
        print(f"flags={f:3d} pg_blocks={os.environ.get('SWX_PG_BLOCKS', 'default')} ms_per_pass={ms:8.2f} "
              f"rtf={args.minutes * 60000.0 / ms:7.1f} same_tokens={toks == ref_tokens} n_tokens={len(toks)}", flush=True)
    if best is not None and os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        with open(os.path.join(ROOT, "gpurun_out", "best_flags.txt"), "w") as fh:
            fh.write(str(best[0]))


if __name__ == "__main__":
    main()
