"""GPU parity tests of the individual HIP kernels, through the C ABI, against the CPU oracle (oracle/whisper/*).

bit-exact: DTW paths, median filter.   Tolerances (stated per test) for floating-point kernels.
"""
import ctypes
import os

import numpy as np
import pytest
import torch

from oracle.whisper import audio as oa
from oracle.whisper import timing as ot

pytestmark = pytest.mark.gpu


def _lib():
    from stable_ts_amd import _lib
    return _lib.load()


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


# ----------------------------------------------------------------------------------------------------------- DTW
def _dtw_case(shape, quant=None, seed=0):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal(shape).astype(np.float32)
    if quant:
        x = (np.round(x * quant) / quant).astype(np.float32)
    return x


@pytest.mark.parametrize("shape,quant", [((1, 1), None), ((1, 9), None), ((9, 1), None), ((3, 5), 1), ((7, 31), None),
                                         ((23, 57), 4), ((64, 200), 2), ((65, 333), None), ((130, 700), 3),
                                         ((226, 1500), None), ((226, 1500), 2), ((448, 1500), None), ((446, 1377), 1)])
def test_dtw_bit_exact(shape, quant):
    from stable_ts_amd.engine import dtw
    x = _dtw_case(shape, quant, seed=shape[0] * 7 + shape[1])
    ref_i, ref_j = ot.dtw_cpu(x.astype(np.float64))
    (ti, tj), = dtw(torch.from_numpy(x)[None].cuda().contiguous(), [shape[0]], [shape[1]])
    assert ti.tolist() == ref_i.tolist()
    assert tj.tolist() == ref_j.tolist()


def test_dtw_known_answer_and_batch_ragged():
    from stable_ts_amd.engine import dtw
    # SURVEY.md 8c known-answer vector + a ragged batch inside one padded tensor
    shapes = [(3, 5), (100, 1500), (37, 911), (1, 1), (225, 1499)]
    ld_n, ld_m = 226, 1500
    X = np.full((len(shapes), ld_n, ld_m), np.nan, dtype=np.float32)
    refs = []
    for k, (n, m) in enumerate(shapes):
        x = np.zeros((n, m), np.float32) if k == 0 else _dtw_case((n, m), 3 if k % 2 else None, seed=k)
        X[k, :n, :m] = x
        refs.append(ot.dtw_cpu(x.astype(np.float64)))
    out = dtw(torch.from_numpy(X).cuda(), [s[0] for s in shapes], [s[1] for s in shapes])
    assert out[0][0].tolist() == [0, 1, 2, 2, 2, 2, 2] and out[0][1].tolist() == [0, 0, 0, 1, 2, 3, 4]
    for (ti, tj), (ri, rj) in zip(out, refs):
        assert ti.tolist() == ri.tolist() and tj.tolist() == rj.tolist()


def test_dtw_path_properties_full_size():
    # size-independent properties at the maximum size: monotone, unit steps, endpoints, length bounds
    from stable_ts_amd.engine import dtw
    n, m = 448, 1500
    x = _dtw_case((n, m), None, seed=99)
    (ti, tj), = dtw(torch.from_numpy(x)[None].cuda().contiguous(), [n], [m])
    assert (ti[0], tj[0]) == (0, 0) and (ti[-1], tj[-1]) == (n - 1, m - 1)
    di, dj = np.diff(ti), np.diff(tj)
    assert ((di == 0) | (di == 1)).all() and ((dj == 0) | (dj == 1)).all() and ((di + dj) >= 1).all()
    assert max(n, m) <= len(ti) <= n + m - 1


# --------------------------------------------------------------------------------------------------- median filter
@pytest.mark.parametrize("shape,width", [((3, 17, 211), 7), ((2, 5, 1500), 7), ((4, 9), 7), ((5, 3), 7), ((2, 64), 5), ((1, 100), 3)])
def test_median_filter_bit_exact(shape, width):
    from stable_ts_amd.engine import median_filter
    x = torch.randn(*shape, generator=torch.Generator().manual_seed(1))
    ref = ot.median_filter(x, width)
    got = median_filter(x.cuda(), width).cpu()
    assert torch.equal(got, ref)


# ------------------------------------------------------------------------------------------------- align weights
def _align_ref(qk, F, qk_scale=1.0, width=7):
    w = qk[..., :F]
    w = (w * qk_scale).softmax(dim=-1)
    std, mean = torch.std_mean(w, dim=-2, keepdim=True, unbiased=False)
    w = (w - mean) / std
    w = ot.median_filter(w, width)
    return -(w.mean(dim=0))


@pytest.mark.parametrize("H,N,F", [(5, 12, 300), (10, 101, 1500), (6, 40, 777), (3, 7, 3)])
def test_align_weights_matches_reference_formula(H, N, F):
    # timing.py:105-110 + :194-195 on the CPU vs the fused kernels; tolerance 2e-5 abs on z-scores of O(1)
    from stable_ts_amd.engine import align_weights
    g = torch.Generator().manual_seed(H * 100 + N)
    qk = torch.randn(1, H, N, 1500, generator=g) * 2.0
    ref = _align_ref(qk[0], F)
    got = align_weights(qk.cuda().contiguous(), [F]).cpu()[0, :, :F]
    torch.testing.assert_close(got, ref, rtol=0, atol=2e-5)


def test_align_then_dtw_path_equals_oracle():
    from stable_ts_amd.engine import align_weights, dtw
    g = torch.Generator().manual_seed(5)
    H, N, F = 8, 60, 1234
    # peaky, roughly monotone attention so that the path is meaningful
    base = -0.5 * ((torch.arange(1500)[None, :] - torch.linspace(30, 1200, N)[:, None]) / 25.0) ** 2
    qk = base[None, None] + 0.3 * torch.randn(1, H, N, 1500, generator=g)
    ref = _align_ref(qk[0], F)
    neg = align_weights(qk.cuda().contiguous(), [F])
    (ti, tj), = dtw(neg, [N], [F])
    ri, rj = ot.dtw_cpu(ref.double().numpy())
    # the matrices agree to ~1e-6, so the DTW paths must coincide except at exact cost ties (none here)
    assert ti.tolist() == ri.tolist() and tj.tolist() == rj.tolist()


# -------------------------------------------------------------------------------------------------------- mel
def _audio(n, seed=0):
    g = torch.Generator().manual_seed(seed)
    t = torch.arange(n) / 16000.0
    x = 0.3 * torch.sin(2 * np.pi * 440 * t) * (0.5 + 0.5 * torch.sin(2 * np.pi * 3 * t))
    x += 0.1 * torch.sin(2 * np.pi * 1234.5 * t) + 0.01 * torch.randn(n, generator=g)
    x[40000:52000] = 0
    return x.float()


def _engine_for_mel(n_mels):
    from stable_ts_amd.engine import Engine, ModelDimensions
    dims = ModelDimensions(n_mels, 1500, 64, 1, 1, 51864, 448, 64, 1, 1)
    return Engine(dims, dtype="f32", max_windows=2, max_rows=2)


@pytest.mark.parametrize("n_mels", [80, 128])
def test_log_mel_matches_oracle(n_mels):
    # tolerance: 1e-3 abs in (log10+4)/4 units (an f32 FFT on the CPU side carries ~1e-4 of its own near the clamp floor);
    # median error is required to be ~1e-6
    eng = _engine_for_mel(n_mels)
    x = torch.stack([_audio(480000, 1), oa.pad_or_trim(_audio(200000, 2) * 0.1, 480000)])
    got = eng.log_mel(x.cuda().contiguous(), per_item_max=True).cpu()
    for b in range(2):
        ref = oa.log_mel_spectrogram(x[b], n_mels)
        err = (got[b] - ref).abs()
        assert err.max().item() < 1e-3, err.max().item()
        assert err.median().item() < 5e-6, err.median().item()
    # batch-global max (upstream quirk used by refine): equals oracle on the stacked batch
    got2 = eng.log_mel(x.cuda().contiguous(), per_item_max=False).cpu()
    ref2 = oa.log_mel_spectrogram(x, n_mels)
    assert (got2 - ref2).abs().max().item() < 1e-3


def test_log_mel_silence_and_short():
    eng = _engine_for_mel(80)
    x = torch.zeros(1, 480000)
    got = eng.log_mel(x.cuda(), per_item_max=True).cpu()
    ref = oa.log_mel_spectrogram(x[0], 80)
    assert torch.allclose(got[0], ref, atol=1e-6)


# ------------------------------------------------------------------------------------------------------- gemm
def _gemm(dtype, a, w, bias=None, res=None, epi=0, force=0, out_f32=False):
    lib = _lib()
    M, K = a.shape
    N = w.shape[0]
    tdt = torch.float16 if dtype == 1 else torch.float32
    c = torch.empty(M, N, dtype=torch.float32 if out_f32 else tdt, device="cuda")
    rc = lib.swx_test_gemm(dtype, _p(a), a.stride(0), _p(w), None if bias is None else _p(bias),
                           None if res is None else _p(res), _p(c), N, M, N, K, epi, force, _stream())
    assert rc == 0, rc
    torch.cuda.synchronize()
    return c


@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (1500, 384, 384), (300, 1152, 384), (77, 130, 96), (3000, 384, 288),
                                   (16, 1280, 1280), (5, 51866, 384), (100, 512, 2048)])
def test_gemm_f16_tiled_and_skinny(M, N, K):
    # A=asymmetric random (G9): f16 inputs, f32 accumulate; reference = f64 matmul of the same f16 values.
    # tolerance: 2e-3 relative to the row/col magnitude (one f16 rounding of the output)
    g = torch.Generator().manual_seed(M + N + K)
    a = (torch.randn(M, K, generator=g) * 0.5).half().cuda()
    w = (torch.randn(N, K, generator=g) * 0.5).half().cuda()
    bias = torch.randn(N, generator=g).cuda()
    ref = (a.double() @ w.double().T + bias.double())
    kinds = [1] + ([2] if (M <= 128 and K % 128 == 0 and K // 128 <= 10 and N <= 16384) else [])
    for force in kinds:
        c = _gemm(1, a, w, bias=bias, epi=1, force=force)
        err = (c.double() - ref).abs().max().item()
        scale = ref.abs().max().item()
        assert err <= 2e-3 * scale + 1e-3, (force, err, scale)
        c32 = _gemm(1, a, w, bias=bias, epi=1 | 8, force=force, out_f32=True)
        err = (c32.double() - ref).abs().max().item()
        assert err <= 2e-5 * scale * max(1.0, (K / 256) ** 0.5) + 1e-4, (force, err, scale)


@pytest.mark.parametrize("M,N,K", [(128, 128, 16), (1500, 384, 384), (77, 130, 96), (3000, 384, 240), (5, 51864, 384)])
def test_gemm_f32_exact_mode(M, N, K):
    # exact-f32 MFMA: error vs f64 is f32 round-off only: <= 2e-6 * sum|a||b| bound, checked as 3e-6 relative
    g = torch.Generator().manual_seed(M * 3 + N + K)
    a = torch.randn(M, K, generator=g).cuda()
    w = torch.randn(N, K, generator=g).cuda()
    bias = torch.randn(N, generator=g).cuda()
    ref = a.double() @ w.double().T + bias.double()
    c = _gemm(0, a, w, bias=bias, epi=1)
    bound = (a.double().abs() @ w.double().abs().T + bias.double().abs())
    assert ((c.double() - ref).abs() <= 3e-6 * bound + 1e-6).all()


@pytest.mark.parametrize("M,N,K,epi", [(100, 1280, 1280, 1), (100, 5120, 1280, 1 | 2), (100, 1280, 5120, 1 | 4), (5, 3840, 1280, 1),
                                       (128, 256, 64, 0), (37, 2570, 384, 1), (65, 300, 192, 1), (100, 51866, 128, 1)])
def test_gemm_f32_few_rows_is_bit_identical_to_the_16_deep_generation(M, N, K, epi):
    # gemm_f32_rows64 (64-deep K chunks, 4 x 4 register transpose, 16-byte LDS reads; round 5) vs gemm_f32_tiled<64, .> (force 8): the
    # same accumulator order element for element, so EQUAL bits -- with bias / GELU / residual, ragged N and M, 1-3 row blocks
    g = torch.Generator().manual_seed(M * 7 + N + K)
    a = torch.randn(M, K, generator=g).cuda()
    w = torch.randn(N, K, generator=g).cuda()
    bias = torch.randn(N, generator=g).cuda()
    res = torch.randn(M, N, generator=g).cuda() if epi & 4 else None
    new = _gemm(0, a, w, bias=bias if epi & 1 else None, res=res, epi=epi)
    old = _gemm(0, a, w, bias=bias if epi & 1 else None, res=res, epi=epi, force=8)
    assert torch.equal(new, old), (new - old).abs().max().item()
    ref = a.double() @ w.double().T + (bias.double() if epi & 1 else 0)
    if not epi & 2:
        ref = ref + (res.double() if epi & 4 else 0)
        bound = a.double().abs() @ w.double().abs().T + bias.double().abs() + 1
        assert ((new.double() - ref).abs() <= 3e-6 * bound + 1e-6).all()


def test_gemm_epilogues():
    g = torch.Generator().manual_seed(3)
    M, N, K = 200, 256, 128
    for dtype, tdt, tol in ((0, torch.float32, 1e-5), (1, torch.float16, 4e-3)):
        a = (torch.randn(M, K, generator=g) * 0.3).to(tdt).cuda()
        w = (torch.randn(N, K, generator=g) * 0.3).to(tdt).cuda()
        bias = torch.randn(N, generator=g).cuda()
        res = torch.randn(M, N, generator=g).to(tdt).cuda()
        lin = a.double() @ w.double().T + bias.double()
        gelu = torch.nn.functional.gelu(lin)
        c = _gemm(dtype, a, w, bias=bias, epi=1 | 2)
        assert (c.double() - gelu).abs().max().item() < tol * max(1.0, gelu.abs().max().item())
        c = _gemm(dtype, a, w, bias=bias, res=res, epi=1 | 4)
        assert (c.double() - (lin + res.double())).abs().max().item() < tol * max(1.0, lin.abs().max().item())


@pytest.mark.parametrize("dtype", [0, 1])
def test_layernorm(dtype):
    lib = _lib()
    tdt = torch.float16 if dtype else torch.float32
    g = torch.Generator().manual_seed(4)
    x = (torch.randn(301, 384, generator=g) * 3 + 1).to(tdt).cuda()
    gam = torch.randn(384, generator=g).cuda()
    bet = torch.randn(384, generator=g).cuda()
    y = torch.empty_like(x)
    assert lib.swx_test_layernorm(dtype, _p(x), _p(gam), _p(bet), _p(y), 301, 384, _stream()) == 0
    ref = torch.nn.functional.layer_norm(x.float(), (384,), gam, bet, 1e-5)
    tol = 2e-3 if dtype else 2e-5
    assert (y.float() - ref).abs().max().item() < tol * ref.abs().max().item()


# --------------------------------------------------------------------------------------------------- attention
def _attn(dtype, q, k, v, force, vt_kp=0):
    lib = _lib()
    B, nq, Hd = q.shape
    nk = k.shape[1]
    H = Hd // 64
    o = torch.empty_like(q)
    if vt_kp:   # transposed cross-KV layout: [B][H][64][kp], zero padded
        vt = torch.zeros(B, H, 64, vt_kp, dtype=v.dtype, device=v.device)
        vt[..., :nk] = v.view(B, nk, H, 64).permute(0, 2, 3, 1)
        v = vt.contiguous()
    rc = lib.swx_test_attention(dtype, _p(q), Hd, _p(k), _p(v), Hd, _p(o), Hd, B, H, nq, nk, force, vt_kp, _stream())
    assert rc == 0, rc
    torch.cuda.synchronize()
    return o


def _attn_ref(q, k, v):
    B, nq, Hd = q.shape
    H = Hd // 64
    qh = q.double().view(B, nq, H, 64).permute(0, 2, 1, 3)
    kh = k.double().view(B, -1, H, 64).permute(0, 2, 1, 3)
    vh = v.double().view(B, -1, H, 64).permute(0, 2, 1, 3)
    w = (qh @ kh.transpose(-1, -2) * 0.125).softmax(-1)
    return (w @ vh).permute(0, 2, 1, 3).reshape(B, nq, Hd)


@pytest.mark.parametrize("B,H,nq,nk", [(1, 2, 64, 64), (2, 3, 1500, 1500), (1, 6, 100, 1500), (2, 2, 5, 1500), (1, 1, 17, 70)])
def test_attention_kernels(B, H, nq, nk):
    g = torch.Generator().manual_seed(B * 100 + nq)
    q = torch.randn(B, nq, H * 64, generator=g)
    k = torch.randn(B, nk, H * 64, generator=g)
    v = torch.randn(B, nk, H * 64, generator=g)   # asymmetric random V catches any d<->key transposition
    # f32 rowwise: 1e-5
    o = _attn(0, q.cuda(), k.cuda(), v.cuda(), 1)
    ref = _attn_ref(q, k, v)
    assert (o.cpu().double() - ref).abs().max().item() < 2e-5
    qh, kh, vh = q.half(), k.half(), v.half()
    refh = _attn_ref(qh, kh, vh)
    for force in (1, 2, 4, 6, 5):   # rowwise f16; flash MFMA f16: default dispatch, 32 / 48 / 64 queries per wave
        for kp in (0, 1536):
            o = _attn(1, qh.cuda(), kh.cuda(), vh.cuda(), force, kp)
            err = (o.cpu().double() - refh).abs().max().item()
            assert err < 6e-3, (force, kp, err)
    o = _attn(0, q.cuda(), k.cuda(), v.cuda(), 1, 1536)
    assert (o.cpu().double() - ref).abs().max().item() < 2e-5
    # f32 on the exact-f32 matrix instruction (round 5: the strict mode's default; 7 pins it): row-major V and the cross-K/V layout
    for force in (7, 0):
        for kp in (0, 1536):
            o = _attn(0, q.cuda(), k.cuda(), v.cuda(), force, kp)
            err = (o.cpu().double() - ref).abs().max().item()
            assert err < 2e-5, (force, kp, err)


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,nq,nk", [(2, 3, 1500, 1500), (1, 2, 64, 64), (1, 2, 300, 128), (1, 1, 17, 70), (2, 2, 1500, 1), (1, 4, 257, 1472),
                                       (1, 2, 100, 1500)])
def test_flash3_is_bit_identical_to_flash2(B, H, nq, nk):
    """Round 6: the software-pipelined f16 flash tile (attn_flash3_f16, SWX_FLAG_FLASH_PIPELINED = 67108864: last tile peeled,
    MFMAs of the neighbouring query blocks inside every softmax; measured slower and off by default) against generation 2,
    32 / 48 / 64 queries per wave, full and ragged last key tiles: the same MFMA and softmax operations per accumulator, hence
    the same bits."""
    lib = _lib()
    g = torch.Generator().manual_seed(B * 77 + nq + nk)
    q = (torch.randn(B, nq, H * 64, generator=g) * 1.5).half().cuda()
    k = torch.randn(B, nk, H * 64, generator=g).half().cuda()
    v = torch.randn(B, nk, H * 64, generator=g).half().cuda()
    kp = ((nk + 63) // 64) * 64 if nk > 1000 else 1536
    old = lib.swx_debug_flags(-1)
    try:
        for force in (4, 6, 5, 2):
            lib.swx_debug_flags(old & ~67108864)
            ref = _attn(1, q, k, v, force, kp).clone()
            lib.swx_debug_flags(old | 67108864)
            got = _attn(1, q, k, v, force, kp)
            assert torch.equal(ref.view(torch.int16), got.view(torch.int16)), (force, (ref.float() - got.float()).abs().max().item())
    finally:
        lib.swx_debug_flags(old)


@pytest.mark.parametrize("B,H,nq,nk", [(1, 20, 1500, 1500), (1, 6, 1500, 1500), (2, 4, 300, 128), (1, 2, 257, 1472), (1, 8, 1500, 1500)])
def test_flash_16_queries_per_wave_is_bit_identical(B, H, nq, nk):
    # round 6: launches that 32 queries per wave would leave at one workgroup per CU or less (the encoder of ONE window: 240 workgroups)
    # run 16 queries per wave (attn_flash2_f16<true, 1>: twice the workgroups, a second wave per SIMD); flag 64 =
    # SWX_FLAG_FLASH_NO_QB1 puts 32 back.  A query block's arithmetic does not depend on its wave's other blocks: equal bits.
    lib = _lib()
    g = torch.Generator().manual_seed(B * 31 + H + nq + nk)
    q = (torch.randn(B, nq, H * 64, generator=g) * 1.5).half().cuda()
    k = torch.randn(B, nk, H * 64, generator=g).half().cuda()
    v = torch.randn(B, nk, H * 64, generator=g).half().cuda()
    kp = ((nk + 63) // 64) * 64 if nk > 1000 else 1536
    old = lib.swx_debug_flags(-1)
    try:
        lib.swx_debug_flags(old | 64)
        ref = _attn(1, q, k, v, 0, kp).clone()
        lib.swx_debug_flags(old & ~64)
        got = _attn(1, q, k, v, 0, kp)
    finally:
        lib.swx_debug_flags(old)
    assert torch.isfinite(got.float()).all()
    assert torch.equal(ref.view(torch.int16), got.view(torch.int16)), (ref.float() - got.float()).abs().max().item()
    err = (got.cpu().double() - _attn_ref(q.cpu(), k.cpu(), v.cpu())).abs().max().item()
    assert err < 8e-3, err


@pytest.mark.parametrize("B,H,nq,nk", [(1, 1, 1, 1500), (3, 4, 5, 1500), (2, 2, 16, 1500), (1, 3, 7, 333)])
def test_attention_decode_cross_kernel(B, H, nq, nk):
    # the HBM-streaming decode-step kernel (<=16 queries, transposed V) vs f64 reference; fp16 P rounding: 6e-3
    g = torch.Generator().manual_seed(B * 10 + nq)
    q = torch.randn(B, nq, H * 64, generator=g).half()
    k = torch.randn(B, nk, H * 64, generator=g).half()
    v = torch.randn(B, nk, H * 64, generator=g).half()
    ref = _attn_ref(q, k, v)
    o = _attn(1, q.cuda(), k.cuda(), v.cuda(), 3, 1536)
    err = (o.cpu().double() - ref).abs().max().item()
    assert err < 6e-3, err
    # strict f32: <= 16 queries take the key-split form of the exact-f32 MFMA kernel (four waves, one 16-key block of a tile each)
    qf, kf, vf = q.float(), k.float(), v.float()
    for kp in (0, 1536):
        o = _attn(0, qf.cuda(), kf.cuda(), vf.cuda(), 7, kp)
        err = (o.cpu().double() - ref).abs().max().item()
        assert err < 2e-5, (kp, err)


@pytest.mark.parametrize("B,H,nq,nk", [(2, 2, 3, 7), (1, 2, 16, 70), (2, 1, 5, 130), (1, 2, 33, 9), (1, 1, 129, 200), (2, 1, 300, 64)])
def test_attention_f32_mfma_edges(B, H, nq, nk):
    # the exact-f32 MFMA kernel at the edges: fewer keys than one 16-key block per wave (key-split form: waves that see no key at
    # all), ragged last tiles, query blocks that end inside a wave / inside a workgroup -- vs the f64 reference and the VALU kernel
    g = torch.Generator().manual_seed(B * 1000 + nq * 10 + nk)
    q = torch.randn(B, nq, H * 64, generator=g) * 2
    k = torch.randn(B, nk, H * 64, generator=g) * 2
    v = torch.randn(B, nk, H * 64, generator=g)
    ref = _attn_ref(q, k, v)
    for kp in (0, 256):
        o = _attn(0, q.cuda(), k.cuda(), v.cuda(), 7, kp)
        assert torch.isfinite(o).all(), kp
        err = (o.cpu().double() - ref).abs().max().item()
        assert err < 3e-5, (kp, err)
        o1 = _attn(0, q.cuda(), k.cuda(), v.cuda(), 1, kp)
        assert (o.cpu() - o1.cpu()).abs().max().item() < 3e-5, kp


@pytest.mark.parametrize("R,H,d,n_new", [(20, 20, 1280, 113), (3, 6, 384, 8), (1, 8, 512, 33), (5, 20, 1280, 226), (2, 6, 384, 448),
                                         (7, 8, 512, 31), (120, 20, 1280, 64)])
def test_self_attention_several_tokens_per_workgroup_is_bit_identical(R, H, d, n_new):
    # round 6: the multi-token self-attention of a teacher-forced pass (scoring pass, prefill: every row starts at position 0) takes 4-8
    # consecutive tokens of a (row, head) per workgroup and stages the head's K / V rows in LDS once (self_attn_cached_mq_f16) instead of one
    # wave per (row, token, head) reading them from L2 again (self_attn_cached): the same arithmetic per token -- equal bits, for token
    # counts with and without a tail, up to the whole context
    import ctypes
    lib = _lib()
    n_ctx = 448
    g = torch.Generator().manual_seed(R * 1000 + n_new)
    q = (torch.randn(R * n_new, d, generator=g) * 0.8).half().cuda()
    kc = torch.zeros(R, n_ctx, d, dtype=torch.half)
    vc = torch.zeros(R, n_ctx, d, dtype=torch.half)
    kc[:, :n_new] = (torch.randn(R, n_new, d, generator=g) * 0.8).half()
    vc[:, :n_new] = torch.randn(R, n_new, d, generator=g).half()
    kc, vc = kc.cuda(), vc.cuda()
    outs = []
    for mq in (0, 1, 1):
        o = torch.full((R * n_new, d), float("nan"), dtype=torch.half, device="cuda")
        rc = lib.swx_test_self_attn_multi(_p(q), _p(kc), _p(vc), R, H, n_new, n_ctx, d, mq, _p(o), _stream())
        assert rc == 0, rc
        torch.cuda.synchronize()
        outs.append(o)
    assert not torch.isnan(outs[0]).any() and float(outs[0].float().abs().max()) > 0
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    # ... and it is an attention: token 0 attends to position 0 only = V[0] of its head, for every head
    assert torch.equal(outs[0].view(R, n_new, d)[:, 0], vc[:, 0])


@pytest.mark.parametrize("check", ["gemm_glds_check.py", "gemm_big8_check.py", "mel_ragged_check.py", "score_qk_check.py",
                                   "self_attn_step_check.py"])
def test_new_kernel_paths_in_subprocess(check):
    # gemm_f16_glds / gemm_f16_ring / gemm_f16_big8 (the direct-to-LDS tiled GEMMs vs the register-staged kernel: bit-identical) and
    # swx_log_mel_ragged (the un-padded spectrogram of refine / locate; index logic CPU-checked in test_mel_ragged_cpu),
    # swx_score_qk (raw per-head scores for the dynamic-heads / 'new' aligner variants; host logic CPU-checked).  Own
    # process: a first-ever hardware run of new device code must not be able to disturb this process's GPU context.
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    from conftest import subprocess_env
    r = subprocess.run([sys.executable, os.path.join(here, "hw_checks", check)], capture_output=True, text=True, timeout=600,
                       env=subprocess_env())
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-2000:])


# ------------------------------------------------------------------------------------- decode-step "dec" GEMM
def _dec_scratch_bytes(M, N, K):
    return N * K * 2 + 8 * N + 16 * M * N * 4 + 8192 + 4096 * 4


def _dec_gemm(a, w, *, gamma=None, beta=None, bias=None, x=None, epi=0, d=0, n_ctx=0, pos0=None, scratch=None):
    """calls swx_test_dec_gemm; returns dict(c=..., x=..., kcache=..., vcache=...) as CPU float32 arrays.  ``scratch``: a zeroed
    buffer the caller keeps over several calls (epi bit 128: the library then leaves the arrival counters as the last launch left them)"""
    lib = _lib()
    M, K = a.shape
    N = w.shape[0]
    dev = "cuda"
    ta = torch.from_numpy(a).to(dev).half().contiguous()
    tw = torch.from_numpy(w).to(dev).half().contiguous()
    f = lambda v: None if v is None else torch.from_numpy(np.asarray(v, np.float32)).to(dev).contiguous()
    tg, tb, tbias = f(gamma), f(beta), f(bias if bias is not None else np.zeros(N, np.float32))
    ldc = d if (epi & 8) else N
    tc = torch.zeros(M, ldc, dtype=torch.float16, device=dev)
    tx = None if x is None else torch.from_numpy(x).to(dev).half().contiguous()
    kc = vc = tp = None
    if epi & 8:
        kc = torch.zeros(M, n_ctx, d, dtype=torch.float16, device=dev)
        vc = torch.zeros(M, n_ctx, d, dtype=torch.float16, device=dev)
        tp = torch.from_numpy(np.asarray(pos0, np.int32)).to(dev)
    if scratch is None:
        scratch = torch.empty(_dec_scratch_bytes(M, N, K), dtype=torch.uint8, device=dev)
    rc = lib.swx_test_dec_gemm(_p(ta), K, _p(tw), None if tg is None else _p(tg), None if tb is None else _p(tb), _p(tbias),
                               _p(tc), ldc, None if tx is None else _p(tx), None if kc is None else _p(kc),
                               None if vc is None else _p(vc), None if tp is None else _p(tp), n_ctx, d, M, N, K, epi,
                               _p(scratch), scratch.numel(), _stream())
    assert rc == 0, rc
    torch.cuda.synchronize()
    out = dict(c=tc.float().cpu().numpy())
    if tx is not None:
        out["x"] = tx.float().cpu().numpy()
    if kc is not None:
        out["kcache"], out["vcache"] = kc.float().cpu().numpy(), vc.float().cpu().numpy()
    return out


def _h(v):
    return np.asarray(v, np.float32).astype(np.float16).astype(np.float64)


def _gelu64(v):
    from math import erf
    return 0.5 * v * (1.0 + np.vectorize(erf)(v / np.sqrt(2.0)))


@pytest.mark.parametrize("M,N,K", [(100, 1280, 1280), (5, 384, 384), (37, 512, 512), (100, 1280, 5120), (64, 768, 768),
                                   (129, 1024, 1024), (20, 384, 1536), (48, 512, 2048)])
def test_dec_gemm_residual(M, N, K):
    # x += a W^T + b (out projections; K = 4d runs K-split through f32 slabs + the slab reduction): f16 rounding of the
    # stored sum is the only error beyond f32 accumulation -> half an f16 ulp of the result
    rng = np.random.default_rng(M + N + K)
    a = rng.standard_normal((M, K)).astype(np.float32) * 0.5
    w = rng.standard_normal((N, K)).astype(np.float32) * 0.03
    b = rng.standard_normal(N).astype(np.float32) * 0.1
    x = rng.standard_normal((M, N)).astype(np.float32)
    got = _dec_gemm(a, w, bias=b, x=x, epi=4 | 16)["x"]
    ref = _h(x) + b.astype(np.float64) + _h(a) @ _h(w).T
    tol = 2e-3 * np.maximum(1.0, np.abs(ref)) + 1e-3
    assert (np.abs(got - ref) <= tol).all(), float(np.abs(got - ref).max())


@pytest.mark.parametrize("M,N,K", [(100, 1280, 5120), (5, 1280, 5120), (20, 384, 1536), (48, 512, 2048), (77, 768, 3072), (160, 1024, 4096)])
def test_dec_gemm_slab_reduction_inside_the_launch_is_bit_identical(M, N, K):
    # round 5: the K slices of a (panel, row group) draw a ticket after publishing their f32 slab, the last arriver reduces
    # (DEC_TICKET, SWX_FLAG_TICKET: measured slower than the separate dec_slab_finish launch, so off by default) -- against the
    # separate launch: equal bits, 8 repetitions on ONE scratch buffer whose counters are zeroed once, before the first launch (epi bit
    # 128: no memset inside the call) -- as in the decode path, every launch must leave the counters at zero for the next one
    # (with a fresh buffer and a memset per call the self-reset was never exercised)
    lib = _lib()
    rng = np.random.default_rng(M * 5 + N + K)
    a = rng.standard_normal((M, K)).astype(np.float32) * 0.5
    w = rng.standard_normal((N, K)).astype(np.float32) * 0.03
    b = rng.standard_normal(N).astype(np.float32) * 0.1
    x = rng.standard_normal((M, N)).astype(np.float32)
    prev = lib.swx_debug_flags(-1)
    try:
        lib.swx_debug_flags(prev & ~2097152)
        ref = _dec_gemm(a, w, bias=b, x=x, epi=4 | 16)["x"]
        lib.swx_debug_flags(prev | 2097152)
        scratch = torch.zeros(_dec_scratch_bytes(M, N, K), dtype=torch.uint8, device="cuda")
        for rep in range(8):
            got = _dec_gemm(a, w, bias=b, x=x, epi=4 | 16 | 128, scratch=scratch)["x"]
            assert np.array_equal(got, ref), (rep, float(np.abs(got - ref).max()))
    finally:
        lib.swx_debug_flags(prev)


@pytest.mark.parametrize("M,N,K", [(5, 1280, 1280), (5, 3840, 1280), (5, 5120, 1280), (5, 1280, 5120), (1, 384, 384), (16, 512, 2048),
                                   (10, 768, 768), (13, 1024, 4096), (60, 1280, 1280), (40, 1280, 1280), (33, 512, 512), (17, 384, 1536)])
def test_dec_gemm_single_wave_workgroups_are_bit_identical(M, N, K):
    # round 6: launches of at most 80 four-wave workgroups with one row tile per workgroup (the 5 rows of a sequential window's decode step;
    # several row groups: the 60 rows of a 20-window prefill at N = 1280) run
    # the same waves as SINGLE-wave workgroups (gemm_dec_f16<1, NKS, EPI, WPB = 1>: four times as many CUs share the weight stream);
    # flag 16 = SWX_FLAG_DEC_NO_W1 puts the four-wave workgroups back.  Every epilogue the decode step uses: equal bits.
    lib = _lib()
    rng = np.random.default_rng(M * 7 + N + K)
    a = (rng.standard_normal((M, K)) * rng.uniform(0.5, 2.0, (M, 1)) + rng.uniform(-1, 1, (M, 1))).astype(np.float32) * 0.5
    w = rng.standard_normal((N, K)).astype(np.float32) * 0.03
    b = rng.standard_normal(N).astype(np.float32) * 0.1
    x = rng.standard_normal((M, N)).astype(np.float32)
    gamma = (1.0 + 0.2 * rng.standard_normal(K)).astype(np.float32)
    beta = (0.1 * rng.standard_normal(K)).astype(np.float32)
    cases = [("res", dict(bias=b, x=x, epi=4 | 16), "x")]
    if K <= 1280:
        cases += [("ln", dict(gamma=gamma, beta=beta, bias=b, epi=1), "c"), ("ln+gelu", dict(gamma=gamma, beta=beta, bias=b, epi=1 | 2), "c")]
        if N % 3 == 0 and N // 3 == K:
            b3 = b.copy(); b3[K:2 * K] = 0.0
            cases += [("qkv", dict(gamma=gamma, beta=beta, bias=b3, epi=1 | 8, d=K, n_ctx=8, pos0=rng.integers(0, 8, M)), None)]
    prev = lib.swx_debug_flags(-1)
    try:
        for name, kw, key in cases:
            lib.swx_debug_flags(prev & ~16)
            got = _dec_gemm(a, w, **kw)
            lib.swx_debug_flags(prev | 16)
            ref = _dec_gemm(a, w, **kw)
            for k in ([key] if key else ["c", "kcache", "vcache"]):
                assert np.isfinite(got[k]).all() and np.abs(got[k]).max() > 0, (name, k)
                assert np.array_equal(got[k], ref[k]), (name, k, float(np.abs(got[k] - ref[k]).max()))
    finally:
        lib.swx_debug_flags(prev)


@pytest.mark.parametrize("M,N,K,gelu", [(100, 1280, 1280, False), (100, 5120, 1280, True), (7, 384, 384, False),
                                        (33, 2048, 512, True), (100, 3072, 768, True)])
def test_dec_gemm_layernorm_fold(M, N, K, gelu):
    # out = [gelu](LN(x) W^T + b) with the LayerNorm folded into the epilogue; x carries a row offset (mean != 0) and a
    # row scale so that the statistics matter; gamma / beta are non-trivial
    rng = np.random.default_rng(3 * M + N + K)
    x = (rng.standard_normal((M, K)) * rng.uniform(0.5, 4.0, (M, 1)) + rng.uniform(-2, 2, (M, 1))).astype(np.float32)
    w = rng.standard_normal((N, K)).astype(np.float32) * 0.03
    b = rng.standard_normal(N).astype(np.float32) * 0.1
    gamma = (1.0 + 0.2 * rng.standard_normal(K)).astype(np.float32)
    beta = (0.1 * rng.standard_normal(K)).astype(np.float32)
    got = _dec_gemm(x, w, gamma=gamma, beta=beta, bias=b, epi=1 | (2 if gelu else 0))["c"]
    xh = _h(x)
    mu = xh.mean(1, keepdims=True)
    ln = (xh - mu) / np.sqrt(xh.var(1, keepdims=True) + 1e-5) * gamma.astype(np.float64) + beta.astype(np.float64)
    ref = ln @ _h(w).T + b.astype(np.float64)
    if gelu:
        ref = _gelu64(ref)
    # the reference path rounds LN(x) to f16 before the GEMM; the folded form rounds W*gamma instead: both are f16-level
    err = np.abs(got - ref)
    assert err.max() < 3e-2 and err.mean() < 2e-3, (float(err.max()), float(err.mean()))


@pytest.mark.parametrize("M,d", [(100, 1280), (13, 384)])
def test_dec_gemm_qkv_scatter(M, d):
    rng = np.random.default_rng(M + d)
    n_ctx = 32
    x = rng.standard_normal((M, d)).astype(np.float32)
    w = rng.standard_normal((3 * d, d)).astype(np.float32) * 0.03
    b = rng.standard_normal(3 * d).astype(np.float32) * 0.1
    b[d:2 * d] = 0.0
    gamma = (1.0 + 0.1 * rng.standard_normal(d)).astype(np.float32)
    beta = (0.05 * rng.standard_normal(d)).astype(np.float32)
    pos0 = rng.integers(0, n_ctx, M)
    got = _dec_gemm(x, w, gamma=gamma, beta=beta, bias=b, epi=1 | 8, d=d, n_ctx=n_ctx, pos0=pos0)
    xh = _h(x)
    ln = (xh - xh.mean(1, keepdims=True)) / np.sqrt(xh.var(1, keepdims=True) + 1e-5) * gamma + beta
    ref = ln @ _h(w).T + b.astype(np.float64)
    assert np.abs(got["c"] - ref[:, :d]).max() < 3e-2
    for m in range(M):
        assert np.abs(got["kcache"][m, pos0[m]] - ref[m, d:2 * d]).max() < 3e-2
        assert np.abs(got["vcache"][m, pos0[m]] - ref[m, 2 * d:]).max() < 3e-2
        mask = np.ones(n_ctx, bool)
        mask[pos0[m]] = False
        assert (got["kcache"][m, mask] == 0).all() and (got["vcache"][m, mask] == 0).all()      # nothing else is touched


# ------------------------------------------------------------------------------------- silence analysis, device half
def test_loudness_probe_kernel():
    # swx_loudness_probe vs numpy on the same samples: the k-th largest |x| (radix select on the bit patterns) must be the
    # element np.partition returns, the gathered |x| the elements at probe_indices -- bit for bit (a selection and a gather,
    # no arithmetic); then NonSpeechPredictor on DEVICE audio (which goes through the probe) must return what it returns
    # for the same audio on the host (full-length path): timings, padded mask, is_silent.
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_host_cpu import _probe_cases
    from stable_ts_amd.engine import loudness_probe
    from stable_ts_amd.stabilization import NonSpeechPredictor, probe_indices
    cases = _probe_cases()
    g = torch.Generator().manual_seed(5)
    cases += [torch.randn(480000, generator=g) * s for s in (1.0, 1e-3, 30.0)]
    cases += [torch.full((480000,), 0.25), torch.cat([torch.zeros(479000), torch.ones(1000) * 0.5])]     # ties at the threshold
    lib = _lib()
    prev = lib.swx_debug_flags(-1)
    try:
        lib.swx_debug_flags(prev | 16777216)                                            # one workgroup per window, whatever the batch
        probes = loudness_probe([c.cuda() for c in cases])
    finally:
        lib.swx_debug_flags(prev)
    # round 6: fewer than 16 windows per call (forced alignment: one) take the selection that is spread over the chip (radix passes as
    # launches of 64 / 32 workgroups per window); the same element, the same gathered samples
    singles = [loudness_probe([c.cuda()])[0] for c in cases]
    fives = [pr for i in range(0, len(cases), 5) for pr in loudness_probe([c.cuda() for c in cases[i:i + 5]])]
    nines = [pr for i in range(0, len(cases), 9) for pr in loudness_probe([c.cuda() for c in cases[i:i + 9]])]
    for ref, *others in zip(probes, singles, fives, nines):
        for o in others:
            assert (ref is None) == (o is None)
            if ref is not None:
                assert ref[0] == o[0] and np.float32(ref[1]).tobytes() == np.float32(o[1]).tobytes() and torch.equal(ref[3], o[3])
    n_checked = 0
    for x, pr in zip(cases, probes):
        n = x.numel()
        idx = probe_indices(n)
        if idx is None:
            assert pr is None
            continue
        pn, thr, pidx, vals = pr
        assert pn == n and np.array_equal(pidx, idx)
        ax = x.abs().numpy()
        k = int(n * 0.001)
        if k:
            want = np.partition(ax, ax.size - k)[ax.size - k]
            assert np.float32(thr).tobytes() == np.float32(want).tobytes(), (n, thr, want)
        else:
            assert np.isnan(thr)
        assert np.array_equal(vals.numpy(), ax[idx]), n
        a, b = NonSpeechPredictor(get_mask=True), NonSpeechPredictor(get_mask=True)
        r1, r2 = a.predict(x.clone(), offset=7.0), b.predict(x.cuda(), offset=7.0)
        assert (r1["timings"] is None) == (r2["timings"] is None)
        assert r1["timings"] is None or np.array_equal(r1["timings"], r2["timings"])
        assert (r1["mask"] is None) == (r2["mask"] is None) and (r1["mask"] is None or torch.equal(r1["mask"], r2["mask"]))
        assert r1["is_silent"] == r2["is_silent"]
        n_checked += 1
    assert n_checked >= 12


@pytest.mark.parametrize("name,B", [("tiny.en", 1), ("base.en", 1), ("base.en", 3)])
def test_cross_kv_from_one_launch_equals_two_launches(name, B):
    # round 6: the K and V projections of a decoder layer's cross-attention as ONE GEMM launch over the fused weight rows (EPI_KV: a
    # tile is a K tile or a V^T tile) against the two launches of rounds 1-5 (swx_debug_flags 8388608): the whole cross-K/V buffer --
    # K rows, V^T blocks and the fragment-ordered copy the decode steps stream -- bit for bit
    import stable_ts_amd as sw
    lib = _lib()
    dims = sw.dims_for(name)
    model = sw.Whisper(dims, device="cuda:0", dtype="f16", max_windows=B, max_rows=B)
    model.load_state_dict(sw.random_state_dict(dims, seed=7, std=0.05))
    g = torch.Generator().manual_seed(B)
    xa = (torch.randn(B, dims.n_audio_ctx, dims.n_audio_state, generator=g) * 0.7).half().cuda()
    prev = lib.swx_debug_flags(-1)
    try:
        lib.swx_debug_flags(prev | 8388608)
        two = model.cross_kv(xa).clone()
        lib.swx_debug_flags(prev & ~8388608)
        one = model.cross_kv(xa).clone()
    finally:
        lib.swx_debug_flags(prev)
    torch.cuda.synchronize()
    assert one.numel() == two.numel() and torch.equal(one, two), int((one != two).sum())
    assert float(one.view(torch.float16).float().abs().max()) > 0.1        # not a buffer of zeros@pytest.mark.gpu
def test_gelu_pair_is_bit_identical_for_every_float():
    """csrc/swx_common.h::gelu_erf2 -- the GELU of the GEMM epilogues restated for two values on packed f32 instructions, both sides of
    the device library's erff branch evaluated and selected -- against 0.5 x (1 + erff(x / sqrt 2)) for ALL 2^32 f32 bit patterns."""
    lib = _lib()
    out = torch.zeros(3, dtype=torch.int64, device="cuda")
    assert lib.swx_test_gelu_pair(_p(out), _stream()) == 0
    torch.cuda.synchronize()
    bad, first, nan_bits = out.cpu().tolist()
    assert bad == 0, (bad, hex(first))
    assert nan_bits == 0, nan_bits


@pytest.mark.gpu
def test_lane_xor_helpers_match_shuffles():
    """csrc/swx_common.h: the wave reductions and the online-softmax row statistics exchange lanes on the VALU (gfx950's
    v_permlane16/32_swap, DPP) instead of ds_bpermute; every helper must return what __shfl_xor returns, lane for lane."""
    lib = _lib()
    n_waves = 64
    g = torch.Generator().manual_seed(11)
    x = torch.randint(0, 2 ** 31 - 1, (n_waves * 64,), generator=g, dtype=torch.int64).to(torch.int32).cuda()
    out = torch.zeros(n_waves * 13 * 64, dtype=torch.int32, device="cuda")
    assert lib.swx_test_lane_xor(_p(x), _p(out), n_waves, _stream()) == 0
    torch.cuda.synchronize()
    o = out.cpu().numpy().reshape(n_waves, 13, 64)
    xs = x.cpu().numpy().reshape(n_waves, 64)
    lanes = np.arange(64)
    for k, off in enumerate((32, 16, 8, 4, 2, 1)):
        np.testing.assert_array_equal(o[:, 6 + k], xs[:, lanes ^ off], err_msg=f"__shfl_xor {off}")
        np.testing.assert_array_equal(o[:, k], o[:, 6 + k], err_msg=f"lane_xor<{off}>")
    assert (o[:, 12] == 127).all(), np.unique(o[:, 12])



