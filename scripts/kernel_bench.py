"""Per-kernel micro-benchmarks at the bench workload's shapes (large-v3, 20 windows x beam 5) through libswx's test hooks.

    python scripts/kernel_bench.py [--iters 200] [--only gemm|flash|cross|splitk]

One HIP-event pair brackets `iters` back-to-back launches of the same kernel, so the figure is the steady-state
launch-to-launch time (kernel + one kernel boundary), which is what a dependent chain such as the decode step pays.
Environment switches (SWX_FLAGS, SWX_PG_POLICY, SWX_PG_BLOCKS) apply as in the library.
"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

EPI_BIAS, EPI_GELU, EPI_RES = 1, 2, 4


def timed(fn, iters):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1000.0 / iters      # us per launch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--only", default="")
    ap.add_argument("--gemm-kernel", type=int, default=1, help="force_kernel of the tiled GEMM: 1 register-staged, 4 direct-to-LDS")
    args = ap.parse_args()
    from stable_ts_amd import _lib
    lib = _lib.load()
    _lib.require_gpu()
    dev = "cuda:0"
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    rnd = lambda *s: (torch.randn(*s, device=dev) * 0.05).half()
    print(f"SWX_FLAGS={lib.swx_debug_flags(-1)} SWX_PG_POLICY={os.environ.get('SWX_PG_POLICY')} SWX_PG_BLOCKS={os.environ.get('SWX_PG_BLOCKS')}")

    if args.only in ("", "gemm"):
        print(f"-- tiled MFMA GEMM (encoder / cross-KV / scoring shapes), f16, bias epilogue, force_kernel={args.gemm_kernel}")
        for M, N, K in [(30000, 1280, 1280), (30000, 3840, 1280), (30000, 5120, 1280), (30000, 1280, 5120), (30000, 2560, 1280),
                        (2240, 1280, 1280), (2240, 5120, 1280), (100, 51866, 1280)]:
            a, w, c = rnd(M, K), rnd(N, K), torch.empty(M, N, dtype=torch.half, device=dev)
            bias = torch.zeros(N, device=dev)
            us = timed(lambda: lib.swx_test_gemm(1, p(a), K, p(w), p(bias), None, p(c), N, M, N, K, EPI_BIAS, args.gemm_kernel, st), max(args.iters // 10, 5))
            print(f"  M={M:6d} N={N:6d} K={K:5d}: {us:9.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s")

    if args.only in ("", "flash"):
        print("-- encoder self-attention (flash), B=20 H=20 nq=nk=1500")
        B, H, n = 20, 20, 1500
        q, k, v = rnd(B, n, H * 64), rnd(B, n, H * 64), rnd(B, n, H * 64)
        o = torch.empty_like(q)
        us = timed(lambda: lib.swx_test_attention(1, p(q), H * 64, p(k), p(v), H * 64, p(o), H * 64, B, H, n, n, 2, 0, st), max(args.iters // 10, 5))
        print(f"  {us:9.1f} us  {4.0 * B * H * n * n * 64 / us / 1e6:7.1f} TFLOP/s")

    if args.only in ("", "cross"):
        print("-- decode-step cross-attention, B=20 H=20 nq=5 nk=1500 (transposed-V layout)")
        B, H, nq, nk, kp = 20, 20, 5, 1500, 1536
        q, k = rnd(B, nq, H * 64), rnd(B, nk, H * 64)
        vt = torch.zeros(B, H, 64, kp, dtype=torch.half, device=dev)
        vt[..., :nk] = rnd(B, H, 64, nk)
        o = torch.empty_like(q)
        us = timed(lambda: lib.swx_test_attention(1, p(q), H * 64, p(k), p(vt), H * 64, p(o), H * 64, B, H, nq, nk, 3, kp, st), args.iters)
        print(f"  {us:9.1f} us  {B * H * 64 * 2 * (2.0 * nk + 2.0 * nq) / us / 1e3:7.1f} GB/s algorithmic")

    if args.only in ("", "splitk"):
        print("-- decode-step GEMMs (split-K weight streaming + finish), M=100")
        M = 100
        for name, N, K, epi, ln in [("qkv", 3840, 1280, EPI_BIAS, False), ("attn-out + LN", 1280, 1280, EPI_BIAS | EPI_RES, True),
                                    ("cross-q", 1280, 1280, EPI_BIAS, False), ("mlp-1 (GELU)", 5120, 1280, EPI_BIAS | EPI_GELU, False),
                                    ("mlp-2 + LN", 1280, 5120, EPI_BIAS | EPI_RES, True)]:
            a, w = rnd(M, K), rnd(N, K)
            bias, lg, lb = torch.zeros(N, device=dev), torch.ones(N, device=dev), torch.zeros(N, device=dev)
            c = torch.zeros(M, N, dtype=torch.half, device=dev)
            h = torch.empty(M, N, dtype=torch.half, device=dev) if ln else None
            us = timed(lambda: lib.swx_test_gemm_splitk(p(a), K, p(w), p(bias), p(c) if epi & EPI_RES else None, p(c), N,
                                                        p(lg) if ln else None, p(lb) if ln else None, p(h), M, N, K, epi, st), args.iters)
            print(f"  {name:14s} N={N:5d} K={K:5d}: {us:7.2f} us per GEMM+finish  {2.0 * (N * K + M * K + M * N) / us / 1e3:7.1f} GB/s algorithmic")


if __name__ == "__main__":
    main()
