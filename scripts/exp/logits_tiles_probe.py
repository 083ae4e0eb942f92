"""Price check (round 6): the decode step's vocabulary projection (M rows x 51 866 x 1 280, f32 logits) on the tiled kernels the dispatch does
not pick for it: force_kernel 0 = dispatch (128-column tiles, occupancy-overlapped), 8 = 64-column tiles of the same kernel, 9 = 128 always,
10 / 11 = ring kernel with 64 / 128 columns.  HBM-cold weights: a 600 MB buffer is streamed between launches.
    python scripts/exp/logits_tiles_probe.py"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from stable_ts_amd import _lib
lib = _lib.load(); _lib.require_gpu()
p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
N, K = 51866, 1280
w = (torch.randn(N, K) * 0.05).half().cuda()
flush = torch.empty(600 * 1024 * 1024, dtype=torch.uint8, device="cuda")
for M in (100, 5):
    a = (torch.randn(M, K) * 0.5).half().cuda()
    c = torch.empty(M, N, dtype=torch.float32, device="cuda")
    ref = None
    for force in (0, 8, 9, 10, 11):
        ts = []
        for rep in range(12):
            flush.fill_(rep)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = lib.swx_test_gemm(1, p(a), K, p(w), None, None, p(c), N, M, N, K, 8, force, st)
            e1.record(); torch.cuda.synchronize()
            if rc != 0:
                break
            ts.append(e0.elapsed_time(e1) * 1000.0)
        if rc != 0:
            print(f"M={M} force={force}: rc {rc}"); continue
        same = True if ref is None else bool(torch.equal(ref.view(torch.int32), c.view(torch.int32)))
        if ref is None:
            ref = c.clone()
        ts.sort()
        print(f"M={M:3d} force={force:2d}: min {ts[0]:7.1f} median {ts[len(ts)//2]:7.1f} us (events incl. ~5 us pair)  bit-identical to dispatch {same}")
