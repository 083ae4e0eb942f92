"""The lane -> line map of the decode step's weight prefetch (csrc/swx_decstep.hip, DecPrefetch in csrc/swx_kernels.h), restated in
Python and checked without a GPU: the lanes of the issuing launch (swx_gemm_dec's grid: the workgroups that got a unit, DEC_NPF
loads per wave) enumerate the 128-byte lines of the next projection's packed weights in order without gaps or repeats, every
address lies inside that matrix (lanes past its end re-touch the last line), and at the benchmarked shape (large-v3, 100 rows)
every projection of the chain is covered completely.  The kernel's own run is checked on hardware by
test_decode_f16_l2_prefetch_has_no_functional_effect (bit-identical results with the prefetch on / off); this test is about the
index arithmetic, where a mistake would be an out-of-bounds read."""
import numpy as np
import pytest

DEC_NPF = 4
DEC_MAXMT = 3


def dec_plan(M, N, K, slab_ok):
    """swx_dec_plan (csrc/swx_decstep.hip): row tiles per workgroup, K slices"""
    depth_ok = lambda ks: ks in (384, 512, 640, 768, 1024, 1280)
    ks2 = 1
    while K % ks2 != 0 or not depth_ok(K // ks2):
        ks2 += 1
        assert ks2 <= 64
    assert ks2 == 1 or slab_ok
    kslice, panels = K // ks2, N // 64
    mt = 1
    while mt < DEC_MAXMT and (mt + 1) * 16 * kslice * 2 <= 122880 and panels * ks2 * -(-M // (mt * 16)) > 256:
        mt += 1
    return mt, ks2


def issued_lines(M, Ni, Ki, slab_i, n_lines):
    """the line index of every prefetch load of one launch, as the kernel computes it, for every workgroup that got a unit"""
    mt, ks2 = dec_plan(M, Ni, Ki, slab_i)
    n_rg = -(-M // (mt * 16))
    units = (Ni // 64) * ks2
    grid = -(-units // 8) * n_rg * 8
    out = []
    for b in range(grid):
        xcd, slot = b & 7, b >> 3
        unit, rg = (slot // n_rg) * 8 + xcd, slot % n_rg
        if unit >= units:
            continue                                            # padding block of the grid: returns before any load
        stride = units * n_rg * 256
        for wave in range(4):
            t0 = ((unit * n_rg + rg) * 4 + wave) * 64
            for i in range(DEC_NPF):
                out.append(np.minimum(t0 + np.arange(64) + i * stride, n_lines - 1))
    return np.concatenate(out), units * n_rg * 256 * DEC_NPF


DIMS = {"tiny": 384, "base": 512, "small": 768, "medium": 1024, "large": 1280}


def chain(d):
    """decoder_step_dec (swx_runtime.hip): issuer (N, K, K-split allowed) -> the projection it prefetches"""
    return [((3 * d, d, False), (d, d)), ((d, d, False), (d, d)), ((d, d, False), (d, d)), ((d, d, False), (4 * d, d)),
            ((4 * d, d, False), (d, 4 * d)), ((d, 4 * d, True), (3 * d, d))]


@pytest.mark.parametrize("name,d", sorted(DIMS.items()))
@pytest.mark.parametrize("M", [32, 55, 100, 160])
def test_prefetch_lines_are_in_bounds_and_dealt_in_order(name, d, M):
    for (Ni, Ki, si), (Nn, Kn) in chain(d):
        n_lines = Nn * Kn * 2 // 128
        lines, lanes = issued_lines(M, Ni, Ki, si, n_lines)
        assert lines.size == lanes
        assert lines.min() >= 0 and lines.max() <= n_lines - 1                 # byte offset line * 128 + 4 <= N * K * 2
        covered = min(lanes, n_lines)
        counts = np.bincount(lines, minlength=n_lines)
        assert (counts[:covered - 1] == 1).all(), (name, M, (Ni, Ki), (Nn, Kn))  # a prefix, each line once ...
        assert counts[covered:].sum() == 0 or covered == n_lines                # ... nothing beyond it
        assert counts.sum() == lanes                                            # the surplus lanes all sit on the last line


def test_benchmarked_shape_is_covered_completely():
    d, M = 1280, 100
    for (Ni, Ki, si), (Nn, Kn) in chain(d):
        n_lines = Nn * Kn * 2 // 128
        lines, lanes = issued_lines(M, Ni, Ki, si, n_lines)
        assert lanes >= n_lines, ((Ni, Ki), (Nn, Kn), lanes, n_lines)
        assert (np.bincount(lines, minlength=n_lines) >= 1).all()
