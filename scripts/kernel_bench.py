"""Per-kernel micro-benchmarks at the bench workload's shapes (large-v3, 20 windows x beam 5) through libswx's test hooks.

    python scripts/kernel_bench.py [--iters 200] [--only gemm|gemm_small|gemm_big|flash|flash_small|cross|dec|dtw]

One HIP-event pair brackets `iters` back-to-back launches of the same kernel, so the figure is the steady-state
launch-to-launch time (kernel + one kernel boundary), which is what a dependent chain such as the decode step pays.
"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

EPI_BIAS, EPI_GELU, EPI_RES = 1, 2, 4


def timed(fn, iters):
    """fn() or fn(i): i cycles so that a caller can rotate through operand copies (weights that must come from HBM: one copy
    would sit in the 256 MB Infinity Cache after the first launch, which the decode step's 1.6 GB per step never does)"""
    import inspect
    takes_i = len(inspect.signature(fn).parameters) == 1
    call = fn if takes_i else (lambda i: fn())
    for i in range(5):
        call(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(iters):
        call(i)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1000.0 / iters      # us per launch


def weight_copies(rnd, N, K, total_mb=600):
    n = max(2, int(total_mb * 1e6 / (N * K * 2)) + 1)
    return [rnd(N, K) for _ in range(n)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--only", default="")
    ap.add_argument("--flash-kernel", type=int, default=2, help="force_kernel of the flash attention: 2 default dispatch, 4 / 6 / 5 = 32 / 48 / 64 queries per wave")
    ap.add_argument("--gemm-kernel", type=int, default=7, help="force_kernel of the tiled GEMM: 1 register-staged, 7 direct-to-LDS (occupancy-overlapped), 8 / 9 = its 64-column tiles always / never, 10 / 11 ring kernel at 64 / 128 columns, 0 = dispatch")
    ap.add_argument("--flags", type=int, default=0, help="swx_debug_flags for the whole run (A/B switches, csrc/swx_kernels.h)")
    args = ap.parse_args()
    from stable_ts_amd import _lib
    lib = _lib.load()
    _lib.require_gpu()
    if args.flags:
        lib.swx_debug_flags(args.flags)
    dev = "cuda:0"
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    rnd = lambda *s: (torch.randn(*s, device=dev) * 0.05).half()
    print(f"debug flags = {lib.swx_debug_flags(-1)}")

    if args.only in ("", "gemm"):
        print(f"-- tiled MFMA GEMM (encoder / cross-KV / scoring shapes), f16, bias epilogue, force_kernel={args.gemm_kernel}")
        for M, N, K in [(30000, 1280, 1280), (30000, 3840, 1280), (30000, 5120, 1280), (30000, 1280, 5120), (30000, 2560, 1280),
                        (2240, 1280, 1280), (2240, 5120, 1280), (100, 51866, 1280), (1500, 1280, 1280), (1500, 3840, 1280),
                        (1500, 5120, 1280), (1500, 1280, 5120)]:
            a, w, c = rnd(M, K), rnd(N, K), torch.empty(M, N, dtype=torch.half, device=dev)
            bias = torch.zeros(N, device=dev)
            us = timed(lambda: lib.swx_test_gemm(1, p(a), K, p(w), p(bias), None, p(c), N, M, N, K, EPI_BIAS, args.gemm_kernel, st), max(args.iters // 10, 5))
            print(f"  M={M:6d} N={N:6d} K={K:5d}: {us:9.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s")

    if args.only in ("gemm_small",):
        # the encoder / cross-K/V at batch 1 (align(), sequential transcribe()): launches of 240-480 tiles.  7 = the kernel that
        # overlaps through occupancy, 10 / 11 = the ring kernel at 64 / 128 columns, 0 = the dispatch
        print("-- tiled MFMA GEMM at batch 1: us per launch by force_kernel")
        codes = [7, 10, 11, 0]
        print("  " + " " * 28 + "".join(f"{c:>9d}" for c in codes))
        for M, N, K in [(1500, 1280, 1280), (1500, 3840, 1280), (1500, 5120, 1280), (1500, 1280, 5120), (1500, 2560, 1280),
                        (3000, 1280, 384), (1500, 1280, 3840), (1500, 512, 512), (1500, 2048, 512), (1500, 512, 2048),
                        (4500, 1280, 1280), (6000, 1280, 5120)]:
            a, w, c = rnd(M, K), rnd(N, K), torch.empty(M, N, dtype=torch.half, device=dev)
            bias = torch.zeros(N, device=dev)
            row = []
            for code in codes:
                rc = lib.swx_test_gemm(1, p(a), K, p(w), p(bias), None, p(c), N, M, N, K, EPI_BIAS, code, st)
                row.append(timed(lambda: lib.swx_test_gemm(1, p(a), K, p(w), p(bias), None, p(c), N, M, N, K, EPI_BIAS, code, st),
                                 max(args.iters // 4, 5)) if rc == 0 else float("nan"))
            # the dispatch's kernel with COLD weights (rotating through 600 MB of copies: more than the 256 MB Infinity Cache holds), which
            # is how a batch-1 encoder meets them (1.27 GB of weights per window), and with a side stream touching the next copy while
            # the current launch runs (what a weight prefetch one GEMM ahead would do)
            ws = weight_copies(rnd, N, K)
            cold = timed(lambda i: lib.swx_test_gemm(1, p(a), K, p(ws[i % len(ws)]), p(bias), None, p(c), N, M, N, K, EPI_BIAS, 0, st),
                         max(args.iters // 4, 5))
            side = torch.cuda.Stream()
            sink = torch.empty(1, dtype=torch.float32, device=dev)

            def with_prefetch(i):
                nxt = ws[(i + 1) % len(ws)]
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    sink.copy_(nxt.view(-1)[::32].float().sum().reshape(1), non_blocking=True)      # touches every 64-byte line
                lib.swx_test_gemm(1, p(a), K, p(ws[i % len(ws)]), p(bias), None, p(c), N, M, N, K, EPI_BIAS, 0, st)
            pf = timed(with_prefetch, max(args.iters // 4, 5))
            print(f"  M={M:6d} N={N:6d} K={K:5d}:" + "".join(f"{u:9.1f}" for u in row) + f"   cold {cold:7.1f}   cold + side-stream touch of the next {pf:7.1f}")

    if args.only in ("gemm_big",):
        # the encoder at 20 / 8 / 4 windows: 7 = 128 x 128 tiles overlapped through occupancy, 12 = the 256 x 256 kernel (gemm_f16_big8)
        print("-- tiled MFMA GEMM at large M: us per launch (TFLOP/s) by force_kernel")
        codes = [7, 12, 0]
        print("  " + " " * 28 + "".join(f"{c:>18d}" for c in codes) + "   12, grouped tile order")
        flags0 = lib.swx_debug_flags(-1)
        for M, N, K in [(30000, 1280, 1280), (30000, 3840, 1280), (30000, 5120, 1280), (30000, 1280, 5120), (30000, 2560, 1280),
                        (12000, 1280, 1280), (12000, 3840, 1280), (12000, 5120, 1280), (12000, 1280, 5120),
                        (6000, 3840, 1280), (6000, 5120, 1280), (6000, 1280, 5120), (60000, 1280, 384), (30000, 1280, 3840)]:
            a, w, c = rnd(M, K), rnd(N, K), torch.empty(M, N, dtype=torch.half, device=dev)
            bias = torch.zeros(N, device=dev)
            row = []
            epi = EPI_BIAS | (2 if N == 5120 else 0)          # the MLP's first projection carries the GELU
            for code in codes + [-12]:
                fk = abs(code)
                lib.swx_debug_flags((flags0 | 4194304) if code < 0 else flags0)       # -12: SWX_FLAG_BIG8_GROUPED
                rc = lib.swx_test_gemm(1, p(a), K, p(w), p(bias), None, p(c), N, M, N, K, epi, fk, st)
                row.append(timed(lambda: lib.swx_test_gemm(1, p(a), K, p(w), p(bias), None, p(c), N, M, N, K, epi, fk, st),
                                 max(args.iters // 10, 5)) if rc == 0 else float("nan"))
                lib.swx_debug_flags(flags0)
            print(f"  M={M:6d} N={N:6d} K={K:5d}:" + "".join(f"{u:9.1f} ({2.0 * M * N * K / u / 1e6:6.0f})" for u in row))

    if args.only in ("flash_small",):
        print("-- encoder self-attention at batch 1 (B=1 H=20 nq=nk=1500) and batch 2 / 4: us per launch by queries per wave")
        for B in (1, 2, 4):
            H, n, kp = 20, 1500, 1536
            q, k = rnd(B, n, H * 64), rnd(B, n, H * 64)
            vt = torch.zeros(B, H, 64, kp, dtype=torch.half, device=dev)
            vt[..., :n] = rnd(B, H, 64, n)
            o = torch.empty_like(q)
            row = [timed(lambda: lib.swx_test_attention(1, p(q), H * 64, p(k), p(vt), H * 64, p(o), H * 64, B, H, n, n, fk, kp, st),
                         max(args.iters // 4, 5)) for fk in (4, 6, 5, 2)]
            print(f"  B={B}: 32 q/wave {row[0]:7.1f}   48 {row[1]:7.1f}   64 {row[2]:7.1f}   dispatch {row[3]:7.1f}")

    if args.only in ("", "flash"):
        print("-- encoder self-attention (flash, V transposed per head as the encoder hands it over), B=20 H=20 nq=nk=1500")
        B, H, n, kp = 20, 20, 1500, 1536
        q, k = rnd(B, n, H * 64), rnd(B, n, H * 64)
        vt = torch.zeros(B, H, 64, kp, dtype=torch.half, device=dev)
        vt[..., :n] = rnd(B, H, 64, n)
        o = torch.empty_like(q)
        for fk in ([args.flash_kernel] if args.flash_kernel != 2 else [4, 6, 5, 4, 6, 5]):
            us = timed(lambda: lib.swx_test_attention(1, p(q), H * 64, p(k), p(vt), H * 64, p(o), H * 64, B, H, n, n, fk, kp, st), max(args.iters // 10, 5))
            print(f"  force_kernel {fk} ({ {4: 32, 6: 48, 5: 64}.get(fk, 0) } queries per wave): {us:9.1f} us  {4.0 * B * H * n * n * 64 / us / 1e6:7.1f} TFLOP/s")

    if args.only in ("", "cross"):
        print("-- decode-step cross-attention, B=20 H=20 nq=5 nk=1500 (transposed-V layout)")
        B, H, nq, nk, kp = 20, 20, 5, 1500, 1536
        q, k = rnd(B, nq, H * 64), rnd(B, nk, H * 64)
        vt = torch.zeros(B, H, 64, kp, dtype=torch.half, device=dev)
        vt[..., :nk] = rnd(B, H, 64, nk)
        o = torch.empty_like(q)
        us = timed(lambda: lib.swx_test_attention(1, p(q), H * 64, p(k), p(vt), H * 64, p(o), H * 64, B, H, nq, nk, 3, kp, st), args.iters)
        print(f"  {us:9.1f} us  {B * H * 64 * 2 * (2.0 * nk + 2.0 * nq) / us / 1e3:7.1f} GB/s algorithmic")

    if args.only in ("", "dec"):
        print("-- decode-step GEMMs (un-split dec kernels), M=100")
        M, d = 100, 1280
        tot = 0.0
        for name, N, K, epi in [("qkv (LN fold + scatter)", 3840, 1280, 1 | 8 | 32), ("attn-out (+x)", 1280, 1280, 4 | 32), ("cross-q (LN fold)", 1280, 1280, 1 | 32),
                                ("cross-out (+x)", 1280, 1280, 4 | 32), ("mlp-1 (LN fold + GELU)", 5120, 1280, 1 | 2 | 32), ("mlp-2 (+x, K=4d)", 1280, 5120, 4 | 16 | 32)]:
            a, ws = rnd(M, K), weight_copies(rnd, N, K)
            c = torch.empty(M, N if not (epi & 8) else d, dtype=torch.half, device=dev)
            x = rnd(M, N if not (epi & 8) else d)
            c1, c2 = torch.zeros(N, device=dev), torch.zeros(N, device=dev)
            kc = torch.zeros(M, 448, d, dtype=torch.half, device=dev)
            vc = torch.zeros(M, 448, d, dtype=torch.half, device=dev)
            pos0 = torch.full((M,), 17, dtype=torch.int32, device=dev)
            scratch = torch.empty(N * K * 2 + 8 * N + 16 * M * N * 4 + 8192, dtype=torch.uint8, device=dev)
            ldc = d if (epi & 8) else N
            fn = lambda i: lib.swx_test_dec_gemm(p(a), K, p(ws[i % len(ws)]), p(c1), p(c2), p(c2), p(c), ldc, p(x), p(kc), p(vc), p(pos0), 448, d,
                                                 M, N, K, epi, p(scratch), scratch.numel(), st)
            rc = fn(0)
            assert rc == 0, rc
            us = timed(fn, args.iters)
            tot += us
            print(f"  {name:26s} N={N:5d} K={K:5d}: {us:7.2f} us  {2.0 * (N * K + M * K + M * N) / us / 1e3:7.1f} GB/s algorithmic")
        print(f"  sum of the six projections of one layer: {tot:.1f} us")

    if args.only in ("gemm_gelu",):
        # the MLP's first projection with pre-activations of REALISTIC magnitude (std ~1.5: erff's two branches both taken inside a wave;
        # the 0.05-scaled operands of the other sections keep every lane on its cheap branch): does the 256 x 256 kernel (one workgroup per
        # CU: nothing to overlap its epilogue with) still beat the 128 x 128 kernel (three workgroups per CU) when the epilogue is this heavy?
        print("-- tiled MFMA GEMM + bias + GELU at realistic activation magnitudes: us per launch by force_kernel")
        codes = [7, 12, 0]
        print("  " + " " * 34 + "".join(f"{c:>10d}" for c in codes))
        for M, N, K in [(30000, 5120, 1280), (12000, 5120, 1280), (1500, 5120, 1280)]:
            for scale, tag in ((0.05, "tiny x"), (0.2, "std ~1.4")):
                a = (torch.randn(M, K, device=dev) * scale).half(); w = (torch.randn(N, K, device=dev) * scale).half()
                c = torch.empty(M, N, dtype=torch.half, device=dev); bias = torch.zeros(N, device=dev)
                row = []
                for fk in codes:
                    rc = lib.swx_test_gemm(1, p(a), K, p(w), p(bias), None, p(c), N, M, N, K, EPI_BIAS | EPI_GELU, fk, st)
                    row.append(timed(lambda: lib.swx_test_gemm(1, p(a), K, p(w), p(bias), None, p(c), N, M, N, K, EPI_BIAS | EPI_GELU, fk, st),
                                     max(args.iters // 10, 5)) if rc == 0 else float("nan"))
                print(f"  M={M:6d} N={N:5d} K={K:5d} {tag:9s}:" + "".join(f"{u:10.1f}" for u in row))

    if args.only in ("logits",):
        # the vocabulary projection of a decode step (133 MB of weights, no bias, f32 out): the tiled kernel it runs on today against
        # the weight-streaming dec kernel at the nearest shape that kernel accepts (N = 810 x 64, residual epilogue) -- a price check
        print("-- decode-step logits projection: tiled GEMM (today) vs the dec kernel at N = 51840")
        K = 1280
        for M in (100, 5):
            N = 51866
            a, ws = rnd(M, K), weight_copies(rnd, N, K, total_mb=700)
            c = torch.empty(M, N, dtype=torch.float32, device=dev)
            fn = lambda i: lib.swx_test_gemm(1, p(a), K, p(ws[i % len(ws)]), None, None, p(c), N, M, N, K, 8, 0, st)
            assert fn(0) == 0
            us = timed(fn, args.iters)
            print(f"  tiled   M={M:4d} N={N}: {us:7.2f} us  {2.0 * N * K / us / 1e3:7.1f} GB/s of weights")
            N = 51840
            ws = weight_copies(rnd, N, K, total_mb=700)
            x = rnd(M, N); c16 = torch.empty(M, N, dtype=torch.half, device=dev)
            c1, c2 = torch.zeros(N, device=dev), torch.zeros(N, device=dev)
            kc = torch.zeros(1, 448, 1280, dtype=torch.half, device=dev); pos0 = torch.full((M,), 17, dtype=torch.int32, device=dev)
            scratch = torch.empty(N * K * 2 + 8 * N + (1 << 20), dtype=torch.uint8, device=dev)
            fn2 = lambda i: lib.swx_test_dec_gemm(p(a), K, p(ws[i % len(ws)]), p(c1), p(c2), p(c2), p(c16), N, p(x), p(kc), p(kc), p(pos0), 448, 1280,
                                                  M, N, K, 4 | 32, p(scratch), scratch.numel(), st)
            rc = fn2(0)
            if rc != 0:
                print(f"  dec     M={M:4d} N={N}: rc {rc}")
                continue
            us = timed(fn2, args.iters)
            print(f"  dec     M={M:4d} N={N}: {us:7.2f} us  {2.0 * N * K / us / 1e3:7.1f} GB/s of weights")

    if args.only in ("", "dtw"):
        print("-- DTW + backtrace (one workgroup per window)")
        from stable_ts_amd.engine import dtw as _dtw  # noqa: F401
        import numpy as np
        for W, N, Mc in [(1, 226, 1500), (20, 226, 1500), (20, 112, 1500), (20, 64, 1500), (4, 448, 1500)]:
            x = torch.randn(W, N, Mc, device=dev)
            dN = torch.full((W,), N, dtype=torch.int32, device=dev)
            dM = torch.full((W,), Mc, dtype=torch.int32, device=dev)
            cap = N + Mc
            ti = torch.empty(W, cap, dtype=torch.int32, device=dev)
            tj = torch.empty(W, cap, dtype=torch.int32, device=dev)
            ln = torch.empty(W, dtype=torch.int32, device=dev)
            wsb = torch.empty(lib.swx_dtw_workspace_bytes(W, N, Mc), dtype=torch.uint8, device=dev)
            fn = lambda: lib.swx_dtw(p(x), W, N, Mc, p(dN), p(dM), p(ti), p(tj), p(ln), p(wsb), st)
            assert fn() == 0
            us = timed(fn, 20)
            steps = Mc + (N + (N + 63) // 64 - 1) // ((N + 63) // 64) - 1
            print(f"  W={W:3d} N={N:4d} M={Mc}: {us:8.1f} us per launch   {1000.0 * us / (steps + N + Mc):7.1f} ns per dependent step (sweep {steps} + walk <= {N + Mc})")


if __name__ == "__main__":
    main()
