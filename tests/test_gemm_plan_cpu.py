"""Which tiled-GEMM kernel a launch gets (csrc/swx_gemm.hip::swx_gemm_plan_f16 -- the function swx_gemm itself executes),
checked without a GPU against the shapes the dispatch rules were measured on (profiles/r03_kb_gemm_ring.txt,
r03_kb_gemm_big.txt): large-v3 / base.en encoder and cross-K/V projections at 1, 4, 8 and 20 windows, the logits GEMM, the
decode-step shapes, and every `force_kernel` code the hardware checks use."""
import pytest

TILED, SKINNY, GLDS128, GLDS64, RING64, RING128, BIG = range(7)
BIAS, GELU, RES, OUT_F32 = 1, 2, 4, 8
NO_RING, NO_BIG = 65536, 131072


@pytest.fixture(scope="module")
def plan():
    from stable_ts_amd import _lib
    lib = _lib.load()
    return lambda M, N, K, epi=BIAS, force=0, flags=0: lib.swx_test_gemm_plan(M, N, K, epi, force, flags)


@pytest.mark.parametrize("M,N,K,epi,want", [
    # one window (align(), sequential transcribe()): ring up to one workgroup per CU, else the occupancy-overlapped kernel
    (1500, 1280, 1280, BIAS | RES, RING64), (1500, 1280, 5120, BIAS | RES, RING64), (1500, 3840, 1280, BIAS, GLDS128),
    (1500, 5120, 1280, BIAS | GELU, GLDS128), (1500, 2560, 1280, BIAS, RING128), (3000, 1280, 384, BIAS | GELU, RING128),
    (1500, 1280, 3840, BIAS | GELU, RING64), (1500, 512, 512, BIAS, RING64), (1500, 2048, 512, BIAS | GELU, RING64),
    # twenty windows: 256 x 256 tiles where they fill whole rounds of the 256 CUs (or nearly, with a long K)
    (30000, 3840, 1280, BIAS, BIG), (30000, 5120, 1280, BIAS | GELU, BIG), (30000, 1280, 5120, BIAS | RES, BIG),
    (30000, 1280, 1280, BIAS | RES, GLDS128), (30000, 2560, 1280, BIAS, BIG), (60000, 1280, 384, BIAS | GELU, GLDS128),
    # eight / four windows
    (12000, 1280, 1280, BIAS | RES, BIG), (12000, 1280, 5120, BIAS | RES, BIG), (6000, 3840, 1280, BIAS, GLDS128),
    (6000, 5120, 1280, BIAS | GELU, BIG), (6000, 1280, 5120, BIAS | RES, GLDS128),
    # an epilogue the 256 x 256 kernel does not have (f32 output) keeps the 128 x 128 tiles
    (30000, 3840, 1280, BIAS | OUT_F32, GLDS128),
    # logits: one row tile over the vocabulary -- never the ring; decode-step sized launches: skinny
    (100, 51866, 1280, OUT_F32, GLDS128), (200, 51866, 1280, OUT_F32, GLDS128), (100, 1280, 1280, BIAS, SKINNY), (5, 3840, 1280, BIAS, SKINNY),
    # K not a multiple of 64: the register-staged kernel
    (1500, 384, 288, BIAS, TILED),
])
def test_dispatch_of_the_benchmarked_shapes(plan, M, N, K, epi, want):
    assert plan(M, N, K, epi) == want


def test_switches_and_forced_kernels(plan):
    assert plan(1500, 1280, 1280, flags=NO_RING) == GLDS64                 # the A/B switches fall back to the overlapped kernel
    assert plan(1500, 2560, 1280, flags=NO_RING) == GLDS128
    assert plan(30000, 3840, 1280, flags=NO_BIG) == GLDS128
    assert plan(30000, 3840, 1280, flags=NO_BIG | NO_RING) == GLDS128
    for force, want in ((1, TILED), (7, GLDS128), (8, GLDS64), (9, GLDS128), (10, RING64), (11, RING128), (12, BIG)):
        assert plan(4500, 3840, 1280, force=force) == want, force
    assert plan(1500, 1280, 1280, force=7) == GLDS64 and plan(1500, 1280, 1280, force=9) == GLDS128
    assert plan(100, 1280, 1280, force=2) == SKINNY and plan(1500, 1280, 1280, force=2) == -4
    assert plan(1500, 1280, 64, force=10) == -4                            # the ring needs two K steps
    assert plan(1500, 1280, 64, force=12) == -4 and plan(1500, 1280, 128, force=12) == BIG     # two K tiles in flight at least
    assert plan(3000, 1280, 1280, BIAS | OUT_F32, force=12) == -4          # not a plain epilogue
    assert plan(1500, 384, 288, force=7) == -4 and plan(1500, 384, 100) == -4


def test_ring_needs_two_k_steps_and_rows(plan):
    assert plan(1500, 1280, 64) == GLDS64                                  # one K step: nothing to keep in flight
    assert plan(1500, 1280, 128) == RING64
    assert plan(256, 20480, 1280, BIAS) in (GLDS128, GLDS64)               # M <= 256: not the ring (one or two row tiles)
