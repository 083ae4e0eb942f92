"""Hardware check of gemm_f16_big8 (csrc/swx_gemm.hip: the 256 x 256 tile on a ring of eight half-tile slots, force_kernel 12)
against the register-staged gemm_f16_tiled (1): the same MFMA sequence per accumulator, so both must agree BIT FOR BIT (first
hardware run, with the two-stage generation it replaced as a third party: profiles/r04_big8_check.txt).  The kernel orders its LDS-DMA by counted waits and raw barriers between two staggered groups of waves; a
misplaced wait would show as a rare wrong tile, not as a wrong kernel, so every shape is repeated REPS times on fresh NaN-filled
outputs while a second stream keeps the memory system busy (a copy loop), and compared each time.  Shapes: the encoder's
projections at 20 windows (the launches the dispatch gives this kernel), M / N tails (clamped rows), two and three K tiles
(the shortest loops: prologue + last-tile waits only), every plain epilogue.  Exit code 0 = all agree.

    python tests/hw_checks/gemm_big8_check.py [--reps 20]
"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

EPI_BIAS, EPI_GELU, EPI_RES = 1, 2, 4


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    from stable_ts_amd import _lib
    lib = _lib.load()
    _lib.require_gpu()
    dev = "cuda:0"
    g = torch.Generator(device="cpu").manual_seed(7)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    side = torch.cuda.Stream()
    junk_a, junk_b = torch.empty(256 << 20, dtype=torch.uint8, device=dev), torch.empty(256 << 20, dtype=torch.uint8, device=dev)
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    bad = 0
    shapes = [(30000, 3840, 1280, EPI_BIAS), (30000, 5120, 1280, EPI_BIAS | EPI_GELU), (30000, 1280, 5120, EPI_BIAS | EPI_RES),
              (30000, 1280, 1280, EPI_BIAS | EPI_RES), (30000, 2560, 1280, EPI_BIAS), (12000, 1280, 1280, EPI_BIAS | EPI_RES),
              (6000, 5120, 1280, EPI_BIAS | EPI_GELU), (256, 256, 128, 0), (256, 256, 192, EPI_BIAS), (257, 263, 256, EPI_BIAS | EPI_RES),
              (1000, 1152, 3840, EPI_BIAS | EPI_RES), (515, 520, 640, EPI_BIAS | EPI_GELU | EPI_RES), (77, 136, 128, EPI_BIAS | EPI_RES),
              (4500, 3840, 1280, EPI_BIAS), (3000, 1000, 1280, EPI_BIAS),
              # round 6 (the grouped tile order behind SWX_FLAG_BIG8_GROUPED: groups of 4 column tiles + a remainder group): 6 / 9 / 13 column tiles, ragged M
              (5000, 1536, 256, EPI_BIAS), (2100, 2304, 384, EPI_BIAS | EPI_RES), (1300, 3300, 128, EPI_BIAS)]
    for (M, N, K, epi) in shapes:
        a = (torch.randn(M, K, generator=g) * 0.5).half().to(dev)
        w = (torch.randn(N, K, generator=g) * 0.03).half().to(dev)
        bias = torch.randn(N, generator=g).float().to(dev)
        res = torch.randn(M, N, generator=g).half().to(dev)
        ldc = N if N % 8 == 0 else N + (8 - N % 8)                       # the 256 x 256 kernels store 16-byte rows

        def run(force):
            c = torch.full((M, ldc), float("nan"), dtype=torch.half, device=dev)
            r = None
            if epi & EPI_RES:
                r = torch.zeros(M, ldc, dtype=torch.half, device=dev)
                r[:, :N] = res
            rc = lib.swx_test_gemm(1, p(a), K, p(w), p(bias), p(r), p(c), ldc, M, N, K, epi, force, st)
            return rc, c
        rc1, c1 = run(1)
        torch.cuda.synchronize()
        wrong = 0
        with torch.cuda.stream(side):                                    # background traffic for the whole repetition loop
            for _ in range(15 * args.reps):
                junk_b.copy_(junk_a, non_blocking=True)
        flags0 = lib.swx_debug_flags(-1)
        for rep in range(args.reps):
            # even repetitions: the row-major tile order (the default), odd ones: the grouped order (SWX_FLAG_BIG8_GROUPED)
            lib.swx_debug_flags((flags0 | 4194304) if rep & 1 else (flags0 & ~4194304))
            rc13, c13 = run(12)
            lib.swx_debug_flags(flags0)
            same = rc13 == 0 and torch.equal(c13[:, :N], c1[:, :N])
            wrong += not same
        torch.cuda.synchronize()
        ok = rc1 == 0 and wrong == 0
        nan_left = bool(torch.isnan(c13[:, :N]).any()) if rc13 == 0 else True
        print(("ok   " if ok and not nan_left else "FAIL ") + f"M={M} N={N} K={K} epi={epi}: rc tiled / big8 = {rc1},{rc13}; "
              f"big8 != tiled in {wrong} of {args.reps} runs; NaN left: {nan_left}")
        bad += not (ok and not nan_left)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
