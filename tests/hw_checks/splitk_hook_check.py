"""Hardware check of swx_test_gemm_splitk (the decode-step GEMM as the fused step launches it: split-K weight streaming
+ finish kernel with bias / GELU / residual / fused LayerNorm) against a float64 reference computed on the CPU from the
same f16 inputs.  Tolerance: the f16 rounding of the output (f32 accumulation inside).  Exit code 0 = all shapes agree.

    python tests/hw_checks/splitk_hook_check.py
"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

EPI_BIAS, EPI_GELU, EPI_RES = 1, 2, 4


def main() -> int:
    from stable_ts_amd import _lib
    lib = _lib.load()
    _lib.require_gpu()
    dev = "cuda:0"
    g = torch.Generator(device="cpu").manual_seed(0)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    bad = 0
    for (M, N, K, epi, ln) in [(100, 1280, 1280, EPI_BIAS | EPI_RES, True), (100, 3840, 1280, EPI_BIAS, False),
                               (100, 5120, 1280, EPI_BIAS | EPI_GELU, False), (100, 1280, 5120, EPI_BIAS | EPI_RES, True),
                               (5, 384, 384, EPI_BIAS, False), (37, 1536, 384, EPI_BIAS | EPI_GELU, False),
                               (128, 512, 2048, EPI_BIAS | EPI_RES, True)]:
        a = (torch.randn(M, K, generator=g) * 0.5).half().to(dev)
        w = (torch.randn(N, K, generator=g) * 0.03).half().to(dev)
        bias = torch.randn(N, generator=g).float().to(dev)
        res = (torch.randn(M, N, generator=g)).half().to(dev)
        lg, lb = torch.randn(N, generator=g).float().to(dev), torch.randn(N, generator=g).float().to(dev)
        c = res.clone() if epi & EPI_RES else torch.empty(M, N, dtype=torch.half, device=dev)
        h = torch.empty(M, N, dtype=torch.half, device=dev) if ln else None
        rc = lib.swx_test_gemm_splitk(p(a), K, p(w), p(bias), p(c) if epi & EPI_RES else None, p(c), N,
                                      p(lg) if ln else None, p(lb) if ln else None, p(h), M, N, K, epi, st)
        torch.cuda.synchronize()
        ref = a.cpu().double() @ w.cpu().double().t() + bias.cpu().double()      # CPU, float64
        if epi & EPI_GELU:
            ref = torch.nn.functional.gelu(ref)
        if epi & EPI_RES:
            ref = ref + res.cpu().double()
        err = ((c.cpu().double() - ref).abs() / (ref.abs() + 1.0)).max().item()
        ok = rc == 0 and err < 4e-3
        msg = f"M={M} N={N} K={K} epi={epi} ln={ln}: rc={rc} max rel err {err:.2e}"
        if ln and rc == 0:
            x = c.cpu().double()                             # the fused LayerNorm sees the stored (f16-rounded) row
            ref_h = torch.nn.functional.layer_norm(x, (N,), lg.cpu().double(), lb.cpu().double(), 1e-5)
            err_h = ((h.cpu().double() - ref_h).abs() / (ref_h.abs() + 1.0)).max().item()
            ok = ok and err_h < 4e-3
            msg += f", LayerNorm {err_h:.2e}"
        print(("ok   " if ok else "FAIL ") + msg)
        bad += not ok
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
