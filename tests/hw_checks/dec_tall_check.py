"""Hardware check of the tall dec GEMM (csrc/swx_decstep.hip::gemm_dectall_f16: register-resident weight fragments, 16-row tiles
through two LDS buffers, LDS-DMA of the next tile from inline asm under the current tile's MFMAs): it must agree BIT FOR BIT
with the decode-step kernel gemm_dec_f16 on the same operands -- same LayerNorm statistics, same k-step order, same epilogue --
for every epilogue the decoder uses (LN+QKV scatter, residual, LN, LN+GELU, K-split slabs + finish), for row counts with and
without a 16-row tail, repeated REPS times per shape (a misplaced wait shows as a rare wrong tile).  Exit code 0 = all agree.

    python tests/hw_checks/dec_tall_check.py
"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
DEC_LN, DEC_GELU, DEC_RES, DEC_QKV, DEC_SLAB, TALL = 1, 2, 4, 8, 16, 64
REPS = 16


def main() -> int:
    from stable_ts_amd import _lib
    lib = _lib.load()
    _lib.require_gpu()
    dev = "cuda:0"
    g = torch.Generator(device="cpu").manual_seed(3)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    bad = 0
    d, n_ctx = 1280, 448
    shapes = [(2260, 3 * d, d, DEC_LN | DEC_QKV), (2260, d, d, DEC_RES), (2261, d, d, DEC_LN), (2255, 4 * d, d, DEC_LN | DEC_GELU),
              (2260, d, 4 * d, DEC_RES | DEC_SLAB), (161, d, d, DEC_RES), (700, 4 * d, d, DEC_LN | DEC_GELU), (4480, d, d, DEC_LN),
              (333, 384, 384, DEC_LN), (500, 512, 2048, DEC_RES | DEC_SLAB), (1000, 3 * 768, 768, DEC_LN | DEC_QKV)]
    for (M, N, K, epi) in shapes:
        dd = N // 3 if epi & DEC_QKV else d
        a = (torch.randn(M, K, generator=g) * 0.7).half().to(dev)
        w = (torch.randn(N, K, generator=g) * 0.03).half().to(dev)
        gamma = (1.0 + 0.1 * torch.randn(K, generator=g)).float().to(dev)
        beta = (0.1 * torch.randn(K, generator=g)).float().to(dev)
        bias = torch.randn(N, generator=g).float().to(dev)
        x0 = torch.randn(M, N, generator=g).half().to(dev)
        n_seq = (M + 6) // 7
        pos0 = torch.randint(0, n_ctx - 8, (n_seq + 1,), generator=g).int().to(dev)
        scratch = torch.empty(N * K * 2 + 8 * N + 16 * M * N * 4 + 4096, dtype=torch.uint8, device=dev)
        outs = []
        for flag in (0, TALL, "w4"):                # reference launch, tall kernel (eight waves per workgroup where the dispatch takes them), tall kernel with four waves always (flag 32)
            for rep in range(REPS if flag else 1):
                c = torch.full((M, N if not (epi & DEC_QKV) else dd), float("nan"), dtype=torch.half, device=dev)
                x = x0.clone()
                kc = torch.zeros(n_seq + 1, n_ctx, dd, dtype=torch.half, device=dev)
                vc = torch.zeros_like(kc)
                # the regular launch scatters with rps = 0 (one token per row): give it the tall launch's row -> (sequence, token) map
                # by running it tall-flagged too but below the tall threshold?  No: compare like with like -- both launches get bit 6
                # (rps = 7); the kernel choice is forced by SWX_FLAG_NO_TALL instead.
                lib.swx_debug_flags(32 if flag == "w4" else 0 if flag else 524288)
                rc = lib.swx_test_dec_gemm(p(a), K, p(w), p(gamma), p(beta), p(bias), p(c), c.shape[1], p(x), p(kc), p(vc), p(pos0),
                                           n_ctx, dd, M, N, K, epi | TALL, p(scratch), scratch.numel(), st)
                torch.cuda.synchronize()
                lib.swx_debug_flags(0)
                res = (rc, c, x, kc, vc)
                if rep == 0:
                    outs.append(res)
                elif not (rc == outs[-1][0] and all(bool(((torch.isnan(u) & torch.isnan(v)) | (u == v)).all()) for u, v in zip(res[1:], outs[-1][1:]))):
                    print(f"M={M} N={N} K={K} epi={epi}: tall kernel differs from its own previous run (rep {rep})")
                    bad += 1
                    break
        (rc0, *t0), (rc1, *t1), (rc2, *t2) = outs
        names = ("C", "X", "kcache", "vcache")
        same = rc0 == rc1 == rc2 == 0 and all(torch.equal(u, v) or (torch.isnan(u) & torch.isnan(v) | (u == v)).all() for u, v in zip(t0, t1)) \
            and all(torch.equal(u, v) or (torch.isnan(u) & torch.isnan(v) | (u == v)).all() for u, v in zip(t0, t2))
        touched = {n_: bool((~torch.isnan(u.float())).any() and (u.float().nan_to_num() != 0).any()) for n_, u in zip(names, t1)}
        print(f"M={M:5d} N={N:5d} K={K:5d} epi={epi:2d}: rc {rc0} / {rc1} / {rc2} (reference / tall / tall, four waves)  bit-identical {bool(same)}  outputs written {touched}")
        if not same:
            for tag, tt in (("tall", t1), ("tall w4", t2)):
                for n_, u, v in zip(names, t0, tt):
                    df = (u.float().nan_to_num() - v.float().nan_to_num()).abs()
                    if df.max() > 0:
                        print(f"   {tag} {n_}: max |diff| {df.max().item():.4g} at {int((df > 0).sum())} elements")
            bad += 1
    print("FAILED" if bad else "all shapes bit-identical")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
