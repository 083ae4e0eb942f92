"""Hardware check of the decode-step self-attention (csrc/swx_attn.hip::self_attn_step_f16<LONG>): for positions below 128 the
short and the long variant, and for every position up to the end of the context the long variant, must agree BIT FOR BIT with the
general cached-attention kernel (self_attn_cached) on the same cache, with and without an ancestor table (beam search), with
ragged positions per row.  Round 4 batches the loads of the positions >= 128 (a decode that started from a long prompt); the
multiply-adds keep their order; round 6 adds the few-waves kernel that requests every load of a row in two batches.  Exit code 0 = all agree.

    python tests/hw_checks/self_attn_step_check.py
"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main() -> int:
    from stable_ts_amd import _lib
    lib = _lib.load()
    _lib.require_gpu()
    dev = "cuda:0"
    g = torch.Generator(device="cpu").manual_seed(11)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    bad = 0
    for (R, H, d, n_ctx, pos_lo, pos_hi, use_anc) in [(100, 20, 1280, 448, 3, 120, True), (100, 20, 1280, 448, 120, 447, True),
                                                      (5, 20, 1280, 448, 200, 447, True), (5, 20, 1280, 448, 128, 129, False),
                                                      (40, 8, 512, 448, 0, 447, True), (7, 6, 384, 448, 250, 400, False),
                                                      (100, 20, 1280, 448, 127, 128, True), (33, 20, 1280, 448, 380, 447, True),
                                                      (15, 6, 384, 448, 0, 127, True), (600, 20, 1280, 448, 3, 120, True),
                                                      (5, 20, 1280, 448, 228, 340, True), (51, 20, 1280, 448, 300, 447, True),
                                                      (5, 20, 1280, 448, 0, 447, True), (10, 8, 512, 448, 319, 321, True)]:
        q = (torch.randn(R, d, generator=g) * 0.8).half().to(dev)
        kc = (torch.randn(R, n_ctx, d, generator=g) * 0.8).half().to(dev)
        vc = torch.randn(R, n_ctx, d, generator=g).half().to(dev)
        pos = torch.randint(pos_lo, pos_hi + 1, (R,), generator=g).int().to(dev)
        anc = None
        if use_anc:      # position j of row r lives in the cache row of some row of the same group of 5 (beam ancestry)
            grp = (torch.arange(R) // 5 * 5)[:, None]
            anc = (grp + torch.randint(0, 5, (R, n_ctx), generator=g)).clamp(max=R - 1).int()
            anc[torch.arange(R), pos.cpu().long()] = torch.arange(R).int()      # the newest position is the row's own (the beam
            anc = anc.to(dev)                                                    # update writes it so; the step kernel assumes it)
        outs = {}
        # variant 1 = the long-context step: at R x H <= 1024 waves the kernel that requests every load of a row in two batches
        # (self_attn_step_long_f16, round 6), else / with flag 4 (SWX_FLAG_SELFATTN_NO_DEEP) the chunk-by-chunk kernel: "1" and "1c"
        old_flags = lib.swx_debug_flags(-1)
        for variant in (0, 1, "1c", 2):
            if variant == 0 and pos_hi >= 128:
                continue
            o = torch.full((R, d), float("nan"), dtype=torch.half, device=dev)
            lib.swx_debug_flags((old_flags | 4) if variant == "1c" else (old_flags & ~4))
            rc = lib.swx_test_self_attn_step(p(q), p(kc), p(vc), p(anc), p(pos), R, H, n_ctx, d, 1 if variant == "1c" else variant, p(o), st)
            torch.cuda.synchronize()
            lib.swx_debug_flags(old_flags)
            outs[str(variant)] = (rc, o)
        ref = outs["2"]
        ok = all(rc == 0 for rc, _ in outs.values()) and all(torch.equal(o, ref[1]) for _, o in outs.values()) and not torch.isnan(ref[1]).any()
        print(f"R={R:3d} H={H:2d} d={d:4d} pos {pos_lo:3d}..{pos_hi:3d} anc={use_anc}: variants {sorted(outs)} bit-identical {bool(ok)}")
        if not ok:
            for v, (rc, o) in outs.items():
                df = (o.float() - ref[1].float()).abs()
                print(f"   variant {v}: rc {rc} max |diff| {df.nan_to_num(1e9).max().item():.4g} at {int((df.nan_to_num(1e9) > 0).sum())} elements")
            bad += 1
    print("FAILED" if bad else "all cases bit-identical")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
