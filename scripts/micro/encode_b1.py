"""Times the encoder + cross-K/V projection at batch 1 and batch 20 (large-v3 dims, random weights): device time by events
around N back-to-back calls, host time of the enqueue.  Diagnoses the align-mode gap between rocprofv3's kernel sum
(8.6 ms per window) and the wall time of the encoder stage (27 ms per window)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import stable_ts_amd as sw

def main():
    dims = sw.dims_for("large-v3")
    for B in (1, 20):
        model = sw.Whisper(dims, device="cuda:0", dtype="f16", max_windows=B, max_rows=B)
        model.load_state_dict(sw.random_state_dict(dims, seed=1, std=0.02))
        mel = torch.randn(B, dims.n_mels, 3000, device="cuda:0")
        for rep in range(2):
            xa = model.encoder(mel)
        torch.cuda.synchronize()
        for n in (1, 4):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            a.record()
            for _ in range(n):
                xa = model.encoder(mel)
            b.record()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            print(f"B={B:2d} encoder x{n}: device {a.elapsed_time(b) / n:8.2f} ms per call, host enqueue {(t1 - t0) * 1e3 / n:7.2f} ms per call, wall {(t2 - t0) * 1e3 / n:7.2f} ms")
        if B == 1:
            # the same call after the device sat idle for a while (align() alternates ~10 ms of device work with host work)
            for idle_ms in (0, 2, 5, 20):
                ts = []
                for _ in range(6):
                    torch.cuda.synchronize()
                    time.sleep(idle_ms * 1e-3)
                    t0 = time.perf_counter()
                    xa = model.encoder(mel)
                    torch.cuda.synchronize()
                    ts.append((time.perf_counter() - t0) * 1e3)
                print(f"B= 1 encoder after {idle_ms:2d} ms idle: wall {min(ts):7.2f} .. {max(ts):7.2f} ms (median {sorted(ts)[3]:.2f})")
            # with a fresh 1 GB device allocation + host tensor work in between, as the alignment loop does
            ts = []
            for _ in range(6):
                torch.cuda.synchronize()
                junk = torch.empty(492 * 1024 * 1024, dtype=torch.uint8, device="cuda:0")
                h = torch.randn(480000).abs().sort().values
                t0 = time.perf_counter()
                xa = model.encoder(mel)
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
            print(f"B= 1 encoder after alloc + host work: wall {min(ts):7.2f} .. {max(ts):7.2f} ms")
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        xkv = model.cross_kv(xa)
        b.record()
        torch.cuda.synchronize()
        print(f"B={B:2d} cross_kv: device {a.elapsed_time(b):8.2f} ms")
        del model
        torch.cuda.empty_cache()

if __name__ == "__main__":
    main()
