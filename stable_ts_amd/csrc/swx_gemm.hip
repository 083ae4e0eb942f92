// swx_gemm.hip -- C = epilogue(A[M,K] * W[N,K]^T) on gfx950 MFMA.
//
// Every Linear / Conv1d of the Whisper encoder and decoder (upstream whisper/model.py, reached from
// stable_whisper/decode.py:27-30,40 and timing.py:59-61) goes through these kernels.  Weights keep the
// checkpoint's [out, in] layout, so both operands are K-contiguous ("B^T input") and each MFMA fragment is one
// 16-byte load.  Three kernels:
//   * tiled f16   128x128x32 block tile, 4 waves (2x2), 4x4 v_mfma_f32_16x16x32_f16 per wave, LDS double buffer,
//                 register-staged global->LDS copies                                   (MFMA-bound shapes)
//   * tiled f32   same tiling on v_mfma_f32_16x16x4_f32 (exact f32 fma chain)         (strict-parity mode)
//   * skinny f16  M <= 128 rows: one 16-column weight panel per workgroup, K split over its 4 waves, weights
//                 streamed straight from HBM into MFMA fragments (no LDS), LDS reduction (HBM-bound decode steps)
// Epilogue (f32): +bias, GELU(erf), +f32 residual indexed by row % res_mod (positional embedding), +T residual,
// store as T or f32.
#include "swx_common.h"
#include "swx_kernels.h"

namespace {

template <typename T>
__device__ __forceinline__ void epilogue_store(const GemmArgs &g, int row, int col, float v)
{
    if (row >= g.M || col >= g.N) return;
    if (g.epi & EPI_BIAS) v += g.bias[col];
    if (g.epi & EPI_GELU) v = gelu_erf(v);
    if (g.epi & EPI_RESF32MOD) v += g.Rf[(size_t)(row % g.res_mod) * g.N + col];
    if (g.epi & EPI_RES) v += to_f32<T>(((const T *)g.R)[(size_t)row * g.ldr + col]);
    if (g.epi & EPI_STORE_VT) {
        const int w = row / g.vt_s, sidx = row - w * g.vt_s;
        ((T *)g.C)[(size_t)w * g.vt_bs + (size_t)col * g.vt_kp + sidx] = from_f32<T>(v);   // col = h*64 + dd
    } else if (g.epi & EPI_CBATCH) {
        const int w = row / g.vt_s, sidx = row - w * g.vt_s;
        ((T *)g.C)[(size_t)w * g.vt_bs + (size_t)sidx * g.ldc + col] = from_f32<T>(v);
    } else if (g.epi & EPI_OUT_F32) ((float *)g.C)[(size_t)row * g.ldc + col] = v;
    else ((T *)g.C)[(size_t)row * g.ldc + col] = from_f32<T>(v);
}

// ------------------------------------------------------------------------------------------------ tiled f16
constexpr int BM = 128, BN = 128;
constexpr int BK16 = 32, LD16 = BK16 + 8;   // halfs; 80-byte rows keep every fragment read 16-byte aligned

__global__ __launch_bounds__(256) void gemm_f16_tiled(GemmArgs g)
{
    __shared__ __attribute__((aligned(16))) f16 As[2][BM][LD16];
    __shared__ __attribute__((aligned(16))) f16 Bs[2][BN][LD16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const f16 *A = (const f16 *)g.A;
    const f16 *W = (const f16 *)g.W;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    f16x8 ra[2], rb[2];
    auto load_regs = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = tid + 256 * i, row = c >> 2, kc = (c & 3) * 8;
            const int gm = m0 + row, gn = n0 + row;
            ra[i] = (gm < g.M) ? *(const f16x8 *)(A + (size_t)gm * g.lda + k0 + kc) : (f16x8)(f16)0;
            rb[i] = (gn < g.N) ? *(const f16x8 *)(W + (size_t)gn * g.ldw + k0 + kc) : (f16x8)(f16)0;
        }
    };
    auto store_lds = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = tid + 256 * i, row = c >> 2, kc = (c & 3) * 8;
            *(f16x8 *)&As[buf][row][kc] = ra[i];
            *(f16x8 *)&Bs[buf][row][kc] = rb[i];
        }
    };

    const int KT = g.K / BK16;
    load_regs(0);
    store_lds(0);
    __syncthreads();
    for (int kt = 0; kt < KT; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < KT) load_regs((kt + 1) * BK16);
        f16x8 a[4], b[4];
        const int fr = lane & 15, fk = (lane >> 4) * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = *(const f16x8 *)&As[cur][wm * 64 + i * 16 + fr][fk];
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = *(const f16x8 *)&Bs[cur][wn * 64 + j * 16 + fr][fk];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
        if (kt + 1 < KT) store_lds(cur ^ 1);
        __syncthreads();
    }

    const int col_l = lane & 15, row_l = (lane >> 4) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                epilogue_store<f16>(g, m0 + wm * 64 + i * 16 + row_l + r, n0 + wn * 64 + j * 16 + col_l, acc[i][j][r]);
}

// ------------------------------------------------------------------------------------------------ tiled f32
constexpr int BK32 = 16, LD32 = BK32 + 1;

__global__ __launch_bounds__(256) void gemm_f32_tiled(GemmArgs g)
{
    __shared__ float As[2][BM][LD32];
    __shared__ float Bs[2][BN][LD32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const float *A = (const float *)g.A;
    const float *W = (const float *)g.W;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    f32x4 ra[2], rb[2];
    auto load_regs = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = tid + 256 * i, row = c >> 2, kc = (c & 3) * 4;
            const int gm = m0 + row, gn = n0 + row;
            ra[i] = (gm < g.M) ? *(const f32x4 *)(A + (size_t)gm * g.lda + k0 + kc) : (f32x4){0.f, 0.f, 0.f, 0.f};
            rb[i] = (gn < g.N) ? *(const f32x4 *)(W + (size_t)gn * g.ldw + k0 + kc) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };
    auto store_lds = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = tid + 256 * i, row = c >> 2, kc = (c & 3) * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) { As[buf][row][kc + e] = ra[i][e]; Bs[buf][row][kc + e] = rb[i][e]; }
        }
    };

    const int KT = g.K / BK32;
    load_regs(0);
    store_lds(0);
    __syncthreads();
    for (int kt = 0; kt < KT; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < KT) load_regs((kt + 1) * BK32);
        const int fr = lane & 15, fk = lane >> 4;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = As[cur][wm * 64 + i * 16 + fr][kk * 4 + fk];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = Bs[cur][wn * 64 + j * 16 + fr][kk * 4 + fk];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < KT) store_lds(cur ^ 1);
        __syncthreads();
    }

    const int col_l = lane & 15, row_l = (lane >> 4) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                epilogue_store<float>(g, m0 + wm * 64 + i * 16 + row_l + r, n0 + wn * 64 + j * 16 + col_l, acc[i][j][r]);
}

// ----------------------------------------------------------------------------------------------- skinny f16
// HBM-bound weight streaming for the decode steps.  Workgroup = one 16-column weight panel, its 4 waves split K.
// Per wave the K slice is walked in chunks of CH k-steps with a two-deep register pipeline: the 16-byte weight
// fragments (HBM) and the MT activation fragments (L2-resident, re-read by every panel) of chunk c+1 are in flight
// while the MFMAs of chunk c issue -- without it each k-step exposes a full memory round trip (measured 40 us per
// launch at M=100 before, rocprof r01 v0).
template <int MT>
__global__ __launch_bounds__(256) void gemm_f16_skinny(GemmArgs g)
{
    constexpr int CH = MT <= 4 ? 4 : 2;
    __shared__ f32x4 red[3][MT][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * 16;
    const f16 *A = (const f16 *)g.A;
    const f16 *W = (const f16 *)g.W;
    const int fr = lane & 15, fk = (lane >> 4) * 8;
    const int kslice = g.K / 4;
    const int nks = kslice / 32;
    const int kb = wave * kslice;
    const int n = n0 + fr;
    const bool nok = n < g.N;
    const f16 *wp = W + (size_t)(nok ? n : 0) * g.ldw + kb + fk;

    f32x4 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const f16 *ap[MT];
    bool aok[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int m = t * 16 + fr;
        aok[t] = m < g.M;
        ap[t] = A + (size_t)(aok[t] ? m : 0) * g.lda + kb + fk;
    }
    const f16x8 zero8 = (f16x8)(f16)0;
    auto load = [&](f16x8 (&wf)[CH], f16x8 (&af)[CH][MT], int ks0) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int ks = ks0 + c;
            const bool kok = ks < nks;
            wf[c] = (nok && kok) ? *(const f16x8 *)(wp + ks * 32) : zero8;
#pragma unroll
            for (int t = 0; t < MT; ++t) af[c][t] = (aok[t] && kok) ? *(const f16x8 *)(ap[t] + ks * 32) : zero8;
        }
    };
    auto comp = [&](const f16x8 (&wf)[CH], const f16x8 (&af)[CH][MT]) {
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
            for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[c][t], wf[c], acc[t], 0, 0, 0);
    };
    f16x8 w0[CH], w1[CH], a0[CH][MT], a1[CH][MT];
    const int nch = (nks + CH - 1) / CH;
    load(w0, a0, 0);
    int c = 0;
    for (; c + 2 <= nch; c += 2) {
        load(w1, a1, (c + 1) * CH);
        comp(w0, a0);
        if (c + 2 < nch) load(w0, a0, (c + 2) * CH);
        comp(w1, a1);
    }
    if (c < nch) comp(w0, a0);

    if (wave > 0) {
#pragma unroll
        for (int t = 0; t < MT; ++t) red[wave - 1][t][lane] = acc[t];
    }
    __syncthreads();
    if (wave == 0) {
        const int col = n0 + (lane & 15), row_l = (lane >> 4) * 4;
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            f32x4 v = acc[t];
#pragma unroll
            for (int w = 0; w < 3; ++w) {
                const f32x4 o = red[w][t][lane];
                v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) epilogue_store<f16>(g, t * 16 + row_l + r, col, v[r]);
        }
    }
}

}  // namespace

int swx_gemm(int dtype, const GemmArgs &g, int force_kernel, hipStream_t s)
{
    if (g.M <= 0 || g.N <= 0) return 0;
    if (dtype == SWX_F16) {
        if (g.K % 32 != 0 || g.lda % 8 != 0 || g.ldw % 8 != 0) return -4;
        const bool skinny_ok = g.M <= 128 && g.K % 128 == 0;
        const bool use_skinny = force_kernel == 2 ? skinny_ok : (force_kernel == 1 ? false : skinny_ok);
        if (force_kernel == 2 && !skinny_ok) return -4;
        if (use_skinny) {
            SwxProfScope prof(PC_GEMM_SKINNY, 2.0 * ((double)g.N * g.K + (double)g.M * g.K) + (double)g.M * g.N * ((g.epi & EPI_OUT_F32) ? 4 : 2), s);
            dim3 grid(cdiv(g.N, 16));
            const int mt = cdiv(g.M, 16);
            switch (mt) {
                case 1: hipLaunchKernelGGL(gemm_f16_skinny<1>, grid, dim3(256), 0, s, g); break;
                case 2: hipLaunchKernelGGL(gemm_f16_skinny<2>, grid, dim3(256), 0, s, g); break;
                case 3: hipLaunchKernelGGL(gemm_f16_skinny<3>, grid, dim3(256), 0, s, g); break;
                case 4: hipLaunchKernelGGL(gemm_f16_skinny<4>, grid, dim3(256), 0, s, g); break;
                case 5: hipLaunchKernelGGL(gemm_f16_skinny<5>, grid, dim3(256), 0, s, g); break;
                case 6: hipLaunchKernelGGL(gemm_f16_skinny<6>, grid, dim3(256), 0, s, g); break;
                case 7: hipLaunchKernelGGL(gemm_f16_skinny<7>, grid, dim3(256), 0, s, g); break;
                default: hipLaunchKernelGGL(gemm_f16_skinny<8>, grid, dim3(256), 0, s, g); break;
            }
        } else {
            SwxProfScope prof(PC_GEMM_TILED, 2.0 * (double)g.M * g.N * g.K, s);
            dim3 grid(cdiv(g.N, BN), cdiv(g.M, BM));
            hipLaunchKernelGGL(gemm_f16_tiled, grid, dim3(256), 0, s, g);
        }
    } else {
        if (g.K % 16 != 0 || g.lda % 4 != 0 || g.ldw % 4 != 0) return -4;
        SwxProfScope prof(PC_GEMM_TILED, 2.0 * (double)g.M * g.N * g.K, s);
        dim3 grid(cdiv(g.N, BN), cdiv(g.M, BM));
        hipLaunchKernelGGL(gemm_f32_tiled, grid, dim3(256), 0, s, g);
    }
    SWX_CHECK_LAUNCH();
    return 0;
}
