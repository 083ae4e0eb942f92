#!/usr/bin/env python
"""Where does a one-window encoder pass spend its time?  (align(): rocprofv3 shows 6.1 ms of kernels back to back per window,
the stage timer 8.3 ms.)  Times `Engine.encode` / `cross_kv` of ONE window, large-v3 random-init, three ways: device time by
events around the call, wall time of call + synchronise, and both again with 2 ms of host idling in front of every call (the
align loop leaves the GPU idle between windows: a clock ramp would show here).

    python scripts/exp/encode_b1_timing.py          (on a GPU box)
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    import stable_ts_amd as sw
    dims = sw.dims_for("large-v3")
    model = sw.Whisper(dims, device="cuda:0", dtype="f16", max_windows=1, max_rows=5)
    model.load_state_dict(sw.random_state_dict(dims, seed=1234, std=0.02, embed_gain=3.0, ts_gain=0.01))
    eng = model.engine
    mel = torch.randn(1, dims.n_mels, 3000, device="cuda:0")
    for idle_ms in (0.0, 2.0, 6.0):
        for name, fn in (("encode", lambda: eng.encode(mel)), ("encode + cross_kv", lambda: eng.cross_kv(eng.encode(mel)))):
            dev, wall = [], []
            for it in range(25):
                torch.cuda.synchronize()
                if idle_ms:
                    time.sleep(idle_ms / 1000.0)
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0 = time.perf_counter()
                a.record()
                fn()
                b.record()
                t1 = time.perf_counter()
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                if it >= 5:
                    dev.append(a.elapsed_time(b))
                    wall.append(1000.0 * (t2 - t0))
                    enq = 1000.0 * (t1 - t0)
            dev.sort(); wall.sort()
            print(f"idle {idle_ms:3.0f} ms before each call | {name:18s}: device (events) median {dev[len(dev) // 2]:6.2f} ms  min {dev[0]:6.2f}   "
                  f"wall call+sync median {wall[len(wall) // 2]:6.2f} ms   host enqueue (last) {enq:5.2f} ms", flush=True)


if __name__ == "__main__":
    main()
