"""Launches every decode-step "dec" GEMM shape of large-v3 (M = 100) 60 times with rotating weight copies (> 256 MB per
shape, so the weights come from HBM as in a real step); run under `rocprofv3 --kernel-trace` -- the per-kernel durations are
the measurement (the host loop is launch-bound).  Environment: SWX_DEC_ABL (bit 0 no weight loads, 1 no activation DMA,
2 no LayerNorm statistics, 3 no MFMA phase, 4 no epilogue, 5 row groups spread over the XCDs), SWX_DEC_POLICY."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from stable_ts_amd import _lib
    lib = _lib.load()
    _lib.require_gpu()
    dev = "cuda:0"
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    rnd = lambda *s: (torch.randn(*s, device=dev) * 0.05).half()
    M, d = int(os.environ.get("DEC_M", "100")), 1280
    for name, N, K, epi in [("qkv", 3840, 1280, 1 | 8 | 32), ("attn-out", 1280, 1280, 4 | 32), ("cross-q", 1280, 1280, 1 | 32),
                            ("mlp-1", 5120, 1280, 1 | 2 | 32), ("mlp-2", 1280, 5120, 4 | 16 | 32)]:
        ws = [rnd(N, K) for _ in range(max(2, int(400e6 / (N * K * 2)) + 1))]
        a = rnd(M, K)
        c = torch.empty(M, N if not (epi & 8) else d, dtype=torch.half, device=dev)
        x = rnd(M, N if not (epi & 8) else d)
        c1, c2 = torch.zeros(N, device=dev), torch.zeros(N, device=dev)
        kc = torch.zeros(M, 448, d, dtype=torch.half, device=dev)
        vc = torch.zeros(M, 448, d, dtype=torch.half, device=dev)
        pos0 = torch.full((M,), 17, dtype=torch.int32, device=dev)
        scratch = torch.empty(N * K * 2 + 8 * N + 16 * M * N * 4 + 8192, dtype=torch.uint8, device=dev)
        ldc = d if (epi & 8) else N
        hot = os.environ.get("DEC_HOT") == "1"
        touch = os.environ.get("DEC_TOUCH")
        for i in range(60):
            if hot:
                i = 0
            if touch is not None:          # read a weight copy `touch` launches ahead (warms the Infinity Cache and the TLB)
                ws[(i + int(touch)) % len(ws)].float().sum()
            rc = lib.swx_test_dec_gemm(p(a), K, p(ws[i % len(ws)]), p(c1), p(c2), p(c2), p(c), ldc, p(x), p(kc), p(vc), p(pos0), 448, d,
                                       M, N, K, epi, p(scratch), scratch.numel(), st)
            assert rc == 0, rc
        torch.cuda.synchronize()


if __name__ == "__main__":
    main()
