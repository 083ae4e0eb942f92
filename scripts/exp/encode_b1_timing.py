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
    from bench import LARGE_V3_HEADS
    model = sw.Whisper(dims, device="cuda:0", dtype="f16", alignment_heads=LARGE_V3_HEADS, max_windows=1, max_rows=5)
    model.load_state_dict(sw.random_state_dict(dims, seed=1234, std=0.02, embed_gain=3.0, ts_gain=0.01))
    eng = model.engine
    mel = torch.randn(1, dims.n_mels, 3000, device="cuda:0")
    # what the align loop runs between two encoder passes: the loudness probe (ONE workgroup for ~0.1-0.4 ms), a teacher-forced
    # scoring pass (~300 short launches) and DTW.  Does the kind of work in front of the encoder change its duration?
    from stable_ts_amd.engine import loudness_probe
    audio = torch.randn(480000, device="cuda:0") * 0.1
    xkv0 = eng.cross_kv(eng.encode(mel))
    toks = [[50258, 50259, 50359, 50363] + [1000 + 7 * i for i in range(100)] + [50257]]

    def pre_probe():
        loudness_probe([audio])

    def pre_score():
        eng.score(xkv0, toks, [1500], 4, 50257)

    for label, pre in (("loudness probe", pre_probe), ("scoring pass", pre_score), ("probe + scoring pass", lambda: (pre_probe(), pre_score()))):
        dev = []
        for it in range(25):
            torch.cuda.synchronize()
            pre()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            eng.encode(mel)
            b.record()
            torch.cuda.synchronize()
            if it >= 5:
                dev.append(a.elapsed_time(b))
        dev.sort()
        print(f"after a {label:22s} | encode            : device (events) median {dev[len(dev) // 2]:6.2f} ms  min {dev[0]:6.2f}  max {dev[-1]:6.2f}", flush=True)

    # the same call on the log-mel of the bench's synthetic audio instead of N(0, 1) noise (is the duration data-dependent?)
    from bench import synth_audio
    wav = synth_audio(60.0, seed=0).cuda()
    for w0 in (0, 480000):
        seg = wav[w0:w0 + 480000]
        real = model.log_mel_batch([seg], [0])
        dev = []
        for it in range(25):
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            model.encoder(real)
            b.record()
            torch.cuda.synchronize()
            if it >= 5:
                dev.append(a.elapsed_time(b))
        dev.sort()
        print(f"log-mel of the bench audio, window at {w0 / 16000:4.0f} s | model.encoder   : device (events) median {dev[len(dev) // 2]:6.2f} ms  "
              f"min {dev[0]:6.2f}  max {dev[-1]:6.2f}   (mel min {float(real.min()):.2f} max {float(real.max()):.2f})", flush=True)

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
