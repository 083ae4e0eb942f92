#!/usr/bin/env python
"""The encoder call INSIDE model.align() (bench.py --mode align's workload, 6 minutes): device time by events, wall time of
call + synchronise, and what the GPU was doing right before -- next to the same call in an isolated loop
(scripts/exp/encode_b1_timing.py: 5.8 ms).  Diagnostic for the 1.35x between the two.

    python scripts/exp/align_encode_probe.py          (on a GPU box)
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    import stable_ts_amd as sw
    from bench import LARGE_V3_HEADS, synth_audio
    dims = sw.dims_for("large-v3")
    model = sw.Whisper(dims, device="cuda:0", dtype="f16", alignment_heads=LARGE_V3_HEADS, max_windows=1, max_rows=1)
    model.load_state_dict(sw.random_state_dict(dims, seed=1234, std=0.02, embed_gain=3.0, ts_gain=0.01))
    minutes = 6.0
    audio = synth_audio(minutes * 60.0, seed=0).cuda()
    g = torch.Generator().manual_seed(7)
    text_tokens = torch.randint(18, 50000, (int(150 * minutes),), generator=g).tolist()
    eng = model.engine
    real_encode, real_ckv = eng.encode, eng.cross_kv
    rec = {"encode": [], "cross_kv": []}
    mels = []

    def timed(name, fn):
        def wrapped(x):
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            a.record()
            out = fn(x)
            b.record()
            torch.cuda.synchronize()
            rec[name].append((a.elapsed_time(b), 1000.0 * (time.perf_counter() - t0)))
            if name == "encode" and len(mels) < 3:
                mels.append(x.clone())
            return out
        return wrapped
    model.align(audio, list(text_tokens), language="en", token_step=100)          # warm-up, untimed
    eng.encode, eng.cross_kv = timed("encode", real_encode), timed("cross_kv", real_ckv)
    t0 = time.perf_counter()
    model.align(audio, list(text_tokens), language="en", token_step=100)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    eng.encode, eng.cross_kv = real_encode, real_ckv
    for name, v in rec.items():
        d = sorted(x[0] for x in v)
        w = sorted(x[1] for x in v)
        print(f"inside align(): {name:9s} {len(v):3d} calls: device median {d[len(d) // 2]:6.2f} ms (min {d[0]:.2f}, max {d[-1]:.2f}); "
              f"wall call+sync median {w[len(w) // 2]:6.2f} ms", flush=True)
    print(f"align pass with the probes: {1000 * wall:.0f} ms for {minutes:g} min", flush=True)
    # the very same mel tensors, replayed in an isolated loop
    for i, mel in enumerate(mels):
        dev = []
        for it in range(15):
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            real_encode(mel)
            b.record()
            torch.cuda.synchronize()
            if it >= 3:
                dev.append(a.elapsed_time(b))
        dev.sort()
        print(f"isolated loop on the mel of align window {i} (shape {tuple(mel.shape)}, dtype {mel.dtype}): device median {dev[len(dev) // 2]:6.2f} ms "
              f"(min {dev[0]:.2f}, max {dev[-1]:.2f})", flush=True)


if __name__ == "__main__":
    main()
