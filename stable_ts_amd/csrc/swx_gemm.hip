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
#include <cstdlib>
#include <array>
#include <vector>
#include <cstdio>
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
constexpr int BK16 = 64, LD16 = BK16 + 8;   // halfs; 144-byte rows keep every fragment read 16-byte aligned
constexpr int CLD = BN + 4;                 // f32 staging row of the epilogue

// Shared tile epilogue of the two tiled f16 kernels: special layouts element-wise, otherwise accumulators -> f32 tile in LDS
// (two halves of 64 rows, 33.8 KB) -> 16-byte row-contiguous stores with bias / GELU / residual in f32.  The bias (the same
// 8 columns for every row group of a thread) and the residual rows of a half are requested BEFORE the accumulators are
// staged: loaded per row group after the previous group's store (a store through a f16 pointer may alias the f32 bias, so
// hipcc keeps the order) they were eight dependent L2 round trips per tile -- 43 % of a K = 1280 launch (see below).
template <int NJ>      // NJ 16-column fragments per wave: tile width BNT = 32 NJ (128 or 64 columns)
__device__ __forceinline__ void tile_epilogue_f16(const GemmArgs &g, unsigned char *smem, f32x4 (&acc)[4][NJ], int m0, int n0,
                                                     int tid, int lane, int wm, int wn)
{
    constexpr int BNT = 32 * NJ, WN = 16 * NJ;        // tile width, columns per wave
    constexpr int CLD_ = BNT + 4;
    const int col_l = lane & 15, row_l = (lane >> 4) * 4;
    // EPI_STORE_VT (the cross-attention V, stored transposed per head: element (row, col) at [col][row within the window]):
    // through LDS like the plain path, but read back COLUMN-wise -- a thread takes 4 consecutive rows of one column (8 bytes in
    // the transposed layout; window boundaries and M are multiples of 4), 16 lanes cover 64 consecutive rows of a column
    // (128 contiguous bytes).  The element-wise path (a 2-byte store, a bias load and an integer division per element) cost
    // 61-83 us against 30 us for the same shape with a plain epilogue.
    if ((g.epi & EPI_STORE_VT) && !(g.epi & (EPI_RES | EPI_GELU | EPI_OUT_F32 | EPI_RESF32MOD | EPI_CBATCH)) && g.vt_s % 4 == 0 &&
        g.vt_kp % 4 == 0 && g.vt_bs % 4 == 0 && g.M % 4 == 0) {
        constexpr int CLV = BNT + 1;                         // odd pitch: the column-wise read-back is 2-way conflicted at worst
        float (*Cv)[CLV] = (float (*)[CLV])smem;            // 64 x 129 x 4 B = 33.0 KB per half
        const int rg = tid & 15, cq = tid >> 4;
        constexpr int NKV = BNT / 16;
        float bvt[NKV];                                      // this thread's columns: requested before any store
#pragma unroll
        for (int k = 0; k < NKV; ++k) bvt[k] = ((g.epi & EPI_BIAS) && n0 + cq + 16 * k < g.N) ? g.bias[n0 + cq + 16 * k] : 0.f;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (wm == half) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) Cv[i * 16 + row_l + r][wn * WN + j * 16 + col_l] = acc[i][j][r];
            }
            __syncthreads();
            const int gm = m0 + half * 64 + rg * 4;
            if (gm < g.M) {
                const int wb = gm / g.vt_s, sidx = gm - wb * g.vt_s;
                f16 *cbase = (f16 *)g.C + (size_t)wb * g.vt_bs + sidx;
#pragma unroll
                for (int k = 0; k < NKV; ++k) {
                    const int col = cq + 16 * k, gn = n0 + col;
                    if (gn >= g.N) continue;
                    const float b = bvt[k];
                    f16x4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = (f16)(Cv[rg * 4 + r][col] + b);
                    *(f16x4 *)(cbase + (size_t)gn * g.vt_kp) = o;
                }
            }
            if (half == 0) __syncthreads();
        }
        return;
    }
    // EPI_CBATCH (rows scattered per batch item: the cross-attention K of every window behind one GEMM) keeps rows contiguous,
    // so it takes the coalesced path with a per-row base; the element-wise path cost 55-98 us against 30 us for the same
    // shape with a plain epilogue (M = 1500, N = K = 1280)
    const bool plain = !(g.epi & (EPI_STORE_VT | EPI_OUT_F32 | EPI_RESF32MOD)) && (g.ldc % 8 == 0) &&
                       (!(g.epi & EPI_RES) || g.ldr % 8 == 0) &&
                       (!(g.epi & EPI_CBATCH) || (g.vt_bs % 8 == 0 && !(g.epi & EPI_RES)));
    if (!plain) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    epilogue_store<f16>(g, m0 + wm * 64 + i * 16 + row_l + r, n0 + wn * WN + j * 16 + col_l, acc[i][j][r]);
        return;
    }
    float (*Cs)[CLD_] = (float (*)[CLD_])smem;          // 64 x 132 x 4 B = 33.8 KB per half (BNT = 128)
    constexpr int TPR = BNT / 8, RPI = 256 / TPR, NIT = 64 / RPI;    // threads per row, rows per iteration, iterations per half
    const int c8 = (tid % TPR) * 8, rb = tid / TPR, gn = n0 + c8;
    const bool col_ok = gn < g.N, full = gn + 8 <= g.N;
    float bv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bv[e] = ((g.epi & EPI_BIAS) && gn + e < g.N) ? g.bias[gn + e] : 0.f;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        f16x8 rv[NIT];
        if (g.epi & EPI_RES) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int gm = m0 + half * 64 + it * RPI + rb;
                rv[it] = (gm < g.M && full) ? *(const f16x8 *)((const f16 *)g.R + (size_t)gm * g.ldr + gn) : (f16x8)(f16)0;
            }
        }
        if (wm == half) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) Cs[i * 16 + row_l + r][wn * WN + j * 16 + col_l] = acc[i][j][r];
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int row = it * RPI + rb, gm = m0 + half * 64 + row;
            if (gm >= g.M || !col_ok) continue;
            const f32x4 lo = *(const f32x4 *)&Cs[row][c8], hi = *(const f32x4 *)&Cs[row][c8 + 4];
            float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += bv[e];
            if (g.epi & EPI_GELU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = gelu_erf(v[e]);
            }
            f16 *cp = (f16 *)g.C + (size_t)gm * g.ldc + gn;
            if (g.epi & EPI_CBATCH) {
                const int wb = gm / g.vt_s;
                cp = (f16 *)g.C + (size_t)wb * g.vt_bs + (size_t)(gm - wb * g.vt_s) * g.ldc + gn;
            }
            if (full) {
                if (g.epi & EPI_RES) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += (float)rv[it][e];
                }
                f16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (f16)v[e];
                *(f16x8 *)cp = o;
            } else {
                for (int e = 0; e < 8 && gn + e < g.N; ++e) {
                    float t = v[e];
                    if (g.epi & EPI_RES) t += (float)((const f16 *)g.R)[(size_t)gm * g.ldr + gn + e];
                    cp[e] = (f16)t;
                }
            }
        }
        if (half == 0) __syncthreads();
    }
}


__global__ __launch_bounds__(256) void gemm_f16_tiled(GemmArgs g)
{
    // one allocation: [A tiles | B tiles] during the K loop, reused as the f32 C tile of the coalesced epilogue
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 2 * BM * LD16 * 2];
    f16 (*As)[BM][LD16] = (f16 (*)[BM][LD16])smem;
    f16 (*Bs)[BN][LD16] = (f16 (*)[BN][LD16])(smem + 2 * BM * LD16 * 2);
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

    f16x8 ra[4], rb[4];
    auto load_regs = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tid + 256 * i, row = c >> 3, kc = (c & 7) * 8;
            const int gm = m0 + row, gn = n0 + row;
            const bool kok = k0 + kc < g.K;          // K may end in the middle of a 64-wide step (K % 32 == 0)
            // predicated loads here: the clamped-address form measured 10-15 % slower on this kernel (profiles r01 v5)
            ra[i] = (gm < g.M && kok) ? *(const f16x8 *)(A + (size_t)gm * g.lda + k0 + kc) : (f16x8)(f16)0;
            rb[i] = (gn < g.N && kok) ? *(const f16x8 *)(W + (size_t)gn * g.ldw + k0 + kc) : (f16x8)(f16)0;
        }
    };
    auto store_lds = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tid + 256 * i, row = c >> 3, kc = (c & 7) * 8;
            *(f16x8 *)&As[buf][row][kc] = ra[i];
            *(f16x8 *)&Bs[buf][row][kc] = rb[i];
        }
    };

    const int KT = (g.K + BK16 - 1) / BK16;
    load_regs(0);
    store_lds(0);
    __syncthreads();
    for (int kt = 0; kt < KT; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < KT) load_regs((kt + 1) * BK16);
        const int fr = lane & 15, fk = (lane >> 4) * 8;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            f16x8 a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = *(const f16x8 *)&As[cur][wm * 64 + i * 16 + fr][kk * 32 + fk];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = *(const f16x8 *)&Bs[cur][wn * 64 + j * 16 + fr][kk * 32 + fk];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < KT) store_lds(cur ^ 1);
        __syncthreads();
    }

    tile_epilogue_f16<4>(g, smem, acc, m0, n0, tid, lane, wm, wn);
}

// ------------------------------------------------------------------------------------- tiled f16, direct-to-LDS
// Same 128 x 128 x 64 tile and 4 x 4 accumulator grid per wave as gemm_f16_tiled, but the operand tiles go
// global -> LDS with `global_load_lds` (16 bytes per lane, 1 KiB per wave instruction): no staging VGPRs and no
// ds_write pass, which is what bounds the register-staged kernel (8 ds_write_b128 per thread and K tile at ~79 B/clk/CU
// are as many LDS cycles as the tile's MFMAs; cdna_hip_programming.md section 5, ladder step 2 -> 3).
// LDS image: a wave instruction writes base + lane * 16, i.e. 8 consecutive 128-byte tile rows; rows are NOT padded.
// Bank conflicts of the fragment reads (16 lanes = 16 tile rows at one 16-byte column slot) are removed by an XOR swizzle
// of the slot with (row >> 1) & 7, applied on the SOURCE address of the load (the LDS destination is fixed by the
// hardware) and on the read address.  Blocks are renumbered so that each XCD (private L2) works on neighbouring tiles.
// Requires K % 64 == 0; rows past M / N are clamped to the last valid row (their results are never stored).
//
// Generation 2 (round 2).  What the K = 1280 shapes of the encoder showed for the first kernel (double buffered, one barrier
// per K step, 67.6 KB of f32 epilogue staging -> two workgroups per CU; profiles/r02_kb_gemm_glds.txt): time = 103 us per
// 1280 of K + 79 us that do not depend on K -- at M = 30000, N = 1280 the fixed part was 43 % of the launch: the epilogue's
// dependent bias / residual loads (see tile_epilogue_f16), with only one other workgroup on the CU to hide them.
// SINGLE = one operand buffer, two barriers per K step, 33.8 KB of LDS -> three to four workgroups per CU: the occupancy, not
// a software pipeline, overlaps one workgroup's loads and epilogue with its neighbours' MFMAs.  Measured at M = 30000
// (profiles/r02_kb_gemm_gen2.txt, TFLOP/s, first kernel -> <false,2> -> <true,4> -> <true,3>): N = K = 1280: 522 -> 737 -> 783
// -> 768; N = 3840: 536 -> 704 -> 754 -> 737; N = 5120: 523 -> 748 -> 809 -> 788; K = 5120: 715 -> 783 -> 816 -> 854.  All
// variants produce bit-identical results (same MFMA order per accumulator, same f32 epilogue arithmetic;
// tests/hw_checks/gemm_glds_check.py).  Default: <true, 3>.
constexpr int GL_TILE = 128 * 128;          // bytes of one operand tile: 128 rows x 64 halfs

template <bool SINGLE, int BNT>     // BNT = tile width: 128, or 64 for shapes whose 128-wide tiles would not fill the chip
__device__ __forceinline__ void gemm_f16_glds_body(const GemmArgs &g)
{
    constexpr int NJ = BNT / 32;                       // 16-column fragments per wave (2 x 2 waves: 64 rows x BNT / 2 columns each)
    constexpr int TB = BNT * 128;                      // bytes of the B operand tile (BNT rows x 64 halfs)
    constexpr int STAGE = GL_TILE + TB;
    constexpr int MAIN = (SINGLE ? 1 : 2) * STAGE;
    constexpr int EPI_ = 64 * (BNT + 4) * 4;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[MAIN > EPI_ ? MAIN : EPI_];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int bx = blockIdx.x, by = blockIdx.y;
    {
        const int gx = gridDim.x, nwg = gx * gridDim.y, orig = by * gx + bx;
        const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
        const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
        bx = wg % gx; by = wg / gx;
    }
    const int m0 = by * BM, n0 = bx * BNT;
    const f16 *A = (const f16 *)g.A;
    const f16 *W = (const f16 *)g.W;

    f32x4 acc[4][NJ];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // per-lane source rows of the 8-row chunks this wave stages: four of A, NJ of W (chunk = wave * n + c)
    const f16 *srcA[4], *srcW[NJ];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int r = (wave * 4 + c) * 8 + (lane >> 3);
        const int slot = (lane & 7) ^ ((r >> 1) & 7);
        const int gm = m0 + r < g.M ? m0 + r : g.M - 1;
        srcA[c] = A + (size_t)gm * g.lda + slot * 8;
    }
#pragma unroll
    for (int c = 0; c < NJ; ++c) {
        const int r = (wave * NJ + c) * 8 + (lane >> 3);
        const int slot = (lane & 7) ^ ((r >> 1) & 7);
        const int gn = n0 + r < g.N ? n0 + r : g.N - 1;
        srcW[c] = W + (size_t)gn * g.ldw + slot * 8;
    }
    typedef __attribute__((address_space(3))) void lds_void;
    auto stage = [&](int kt, int buf) {
        unsigned char *ta = smem + buf * STAGE, *tb = ta + GL_TILE;
#pragma unroll
        for (int c = 0; c < 4; ++c)
            __builtin_amdgcn_global_load_lds(srcA[c] + kt * 64, (lds_void *)(ta + (wave * 4 + c) * 1024), 16, 0, 0);
#pragma unroll
        for (int c = 0; c < NJ; ++c)
            __builtin_amdgcn_global_load_lds(srcW[c] + kt * 64, (lds_void *)(tb + (wave * NJ + c) * 1024), 16, 0, 0);
    };
    const int KT = g.K / 64;
    const int fr = lane & 15, fs = lane >> 4;
    auto compute = [&](int buf) {
        const unsigned char *ta = smem + buf * STAGE, *tb = ta + GL_TILE;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            f16x8 a[4], b[NJ];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = wm * 64 + i * 16 + fr;
                a[i] = *(const f16x8 *)(ta + row * 128 + (((kk * 4 + fs) ^ ((row >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int row = wn * (BNT / 2) + j * 16 + fr;
                b[j] = *(const f16x8 *)(tb + row * 128 + (((kk * 4 + fs) ^ ((row >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    };
    if (SINGLE) {
        for (int kt = 0; kt < KT; ++kt) {
            stage(kt, 0);
            __syncthreads();
            compute(0);
            __syncthreads();
        }
    } else {
        stage(0, 0);
        __syncthreads();
        for (int kt = 0; kt < KT; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < KT) stage(kt + 1, cur ^ 1);
            compute(cur);
            __syncthreads();
        }
    }
    tile_epilogue_f16<NJ>(g, smem, acc, m0, n0, tid, lane, wm, wn);
}

// the instantiations in use (second launch bound = workgroups per CU the register allocation must allow)
__global__ __launch_bounds__(256, 2) void gemm_f16_glds_d2_128(GemmArgs g) { gemm_f16_glds_body<false, 128>(g); }
__global__ __launch_bounds__(256, 4) void gemm_f16_glds_s4_128(GemmArgs g) { gemm_f16_glds_body<true, 128>(g); }
__global__ __launch_bounds__(256, 3) void gemm_f16_glds_s3_128(GemmArgs g) { gemm_f16_glds_body<true, 128>(g); }
__global__ __launch_bounds__(256, 3) void gemm_f16_glds_s3_64(GemmArgs g) { gemm_f16_glds_body<true, 64>(g); }

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
__global__ __launch_bounds__(256) void gemm_f16_skinny(GemmArgs g, float *slabs, int64_t slab_stride, int ks2)
{
    constexpr int NKS_MAX = 10;      // k-steps (of 32) per wave: the launcher keeps K / (128 * ks2) <= 10
    __shared__ f32x4 red[3][MT][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * 16;
    const f16 *A = (const f16 *)g.A;
    const f16 *W = (const f16 *)g.W;
    const int fr = lane & 15, fk = (lane >> 4) * 8;
    // K is split over blockIdx.y (ks2 slices, partial sums go to f32 slabs) and then over the 4 waves
    const int kslice = g.K / (4 * ks2);
    const int nks = kslice / 32;
    const int kb = (blockIdx.y * 4 + wave) * kslice;
    if (ks2 > 1 || slabs) {
        g.C = slabs + (size_t)blockIdx.y * slab_stride;
        g.ldc = g.N;
        g.epi = EPI_OUT_F32;
    }
    const int n = n0 + fr;
    const bool nok = n < g.N;
    const f16 *wp = W + (size_t)(nok ? n : 0) * g.ldw + kb + fk;
    const f16x8 zero8 = (f16x8)(f16)0;

    // HBM side first: every weight fragment of this wave's K slice is put in flight at once (<= 10 x 1 KB per wave);
    // the bandwidth-delay product of the chip (~10 MB) needs tens of KB outstanding per CU, which a double-buffered
    // weight load cannot provide
    f16x8 wf[NKS_MAX];
#pragma unroll
    for (int ks = 0; ks < NKS_MAX; ++ks) {
        const f16x8 v = *(const f16x8 *)(wp + (ks < nks ? ks : nks - 1) * 32);
        wf[ks] = (nok && ks < nks) ? v : zero8;
    }

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
    // L2 side: activation fragments, two k-steps deep
    f16x8 af[2][MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) { const f16x8 v = *(const f16x8 *)(ap[t]); af[0][t] = aok[t] ? v : zero8; }
#pragma unroll
    for (int ks = 0; ks < NKS_MAX; ++ks) {
        if (ks < nks) {
            if (ks + 1 < nks) {
#pragma unroll
                for (int t = 0; t < MT; ++t) { const f16x8 v = *(const f16x8 *)(ap[t] + (ks + 1) * 32); af[(ks + 1) & 1][t] = aok[t] ? v : zero8; }
            }
#pragma unroll
            for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[ks & 1][t], wf[ks], acc[t], 0, 0, 0);
        }
    }

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

// ------------------------------------------------------------------------------------------- panel-group f16
// Decode-step GEMM, second generation.  Profile r01 v3 showed the 16-column-panel kernel above bound by L2->CU
// ACTIVATION traffic (every panel re-reads the whole [M][K] activation: 3242 panels x 256 KB for the logits), not by
// HBM.  Here a workgroup owns 64 output columns (one 16-column panel per wave) x one K slice (blockIdx.y); the
// activation chunk [MT*16][64] of each step is staged ONCE per workgroup in LDS with coalesced full-line loads and
// shared by the four waves (4x fewer, 2x wider L2 requests), while every wave streams its own weight fragments of the
// whole slice from HBM up front.  Partial sums go to f32 slabs (deterministic split-K), finished by splitk_finish_f16.
constexpr int PG_LD = 72;        // halfs per LDS row
constexpr int PG_MAXIT = 20;     // 64-wide K chunks per workgroup slice, upper bound (launcher picks the <= 5, <= 10 or <= 20 build)
template <int MT, int MAXIT, bool DIRECT>
__global__ __launch_bounds__(256) void gemm_f16_pg(GemmArgs g, float *slabs, int64_t slab_stride, int ks2, int flags)
{
    __shared__ __attribute__((aligned(16))) f16 As[2][MT * 16][PG_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fk = (lane >> 4) * 8;
    const f16 *A = (const f16 *)g.A;
    const f16 *W = (const f16 *)g.W;
    const int kslice = g.K / ks2;
    const int nit = kslice / 64;
    const int kb = blockIdx.y * kslice;
    const int n = blockIdx.x * 64 + wave * 16 + fr;
    const bool nok = n < g.N;
    const f16 *wp = W + (size_t)(nok ? n : 0) * g.ldw + kb + fk;
    const f16x8 zero8 = (f16x8)(f16)0;

    f16x8 wf[2 * MAXIT];
#pragma unroll
    for (int ks = 0; ks < 2 * MAXIT; ++ks) wf[ks] = (nok && ks < 2 * nit) ? *(const f16x8 *)(wp + ks * 32) : zero8;

    f32x4 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    constexpr int NCH = (MT * 16 * 8 + 255) / 256;      // 16-byte chunks of the activation tile per thread
    f16x8 ra[NCH];
    auto load_a = [&](int it) {
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int idx = tid + 256 * j, row = idx >> 3, c8 = (idx & 7) * 8;
            ra[j] = (row < g.M) ? *(const f16x8 *)(A + (size_t)row * g.lda + kb + it * 64 + c8) : zero8;
        }
    };
    auto store_a = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int idx = tid + 256 * j, row = idx >> 3, c8 = (idx & 7) * 8;
            if (row < MT * 16) *(f16x8 *)&As[buf][row][c8] = ra[j];
        }
    };
    load_a(0);
    store_a(0);
    __syncthreads();
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        if (it < nit) {
            const int cur = it & 1;
            if (it + 1 < nit) load_a(it + 1);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    const f16x8 a = *(const f16x8 *)&As[cur][t * 16 + fr][kk * 32 + fk];
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, wf[it * 2 + kk], acc[t], 0, 0, 0);
                }
            if (it + 1 < nit) store_a(cur ^ 1);
            __syncthreads();
        }
    }
    const int col = blockIdx.x * 64 + wave * 16 + (lane & 15), row_l = (lane >> 4) * 4;
    if constexpr (DIRECT) {
        // un-split K (ks2 == 1): the workgroup owns the finished sums, so bias / GELU are applied here and the compute
        // dtype is stored directly -- no slab, no finish launch.  Same arithmetic as splitk_finish_f16 on one slab.
        if (col < g.N) {
            const float bias = (g.epi & EPI_BIAS) ? g.bias[col] : 0.f;
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = t * 16 + row_l + r;
                    float a = 0.f + acc[t][r];
                    if (g.epi & EPI_BIAS) a += bias;
                    if (g.epi & EPI_GELU) a = gelu_erf(a);
                    if (row < g.M) ((f16 *)g.C)[(size_t)row * g.ldc + col] = (f16)a;
                }
        }
        return;
    }
    // each wave owns its 16 columns for this K slice: plain f32 partials, no cross-wave reduction
    float *out = slabs + (size_t)blockIdx.y * slab_stride;
    if (col < g.N) {
        if (flags & SWX_FLAG_SC1_SLABS) {
            // write-through: the partials reach memory while the kernel runs instead of as one dirty-L2 write-back at
            // the kernel boundary (MI355X_MICROARCH.md "boundary" / "publish-large" rows)
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = t * 16 + row_l + r;
                    if (row < g.M) __hip_atomic_store(&out[(size_t)row * g.N + col], acc[t][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
        } else {
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = t * 16 + row_l + r;
                    if (row < g.M) out[(size_t)row * g.N + col] = acc[t][r];
                }
        }
    }
}

// ------------------------------------------------------------------------------------------- split-K finish
// One workgroup per output row: v = sum_k slab[k][row][:] (+bias, GELU, +residual) -> C ; optionally the LayerNorm
// of the finished row -> ln_out (saves the separate LN launch and its extra pass), and for the fused QKV projection
// the new K / V rows go straight into the self-attention cache at the row's current position.
constexpr int FIN_MAXC = 20;   // columns per thread: N <= 256 * 20
// KS2 > 0: slab count known at compile time (every slab load of a thread is independent and in flight at once);
// KS2 == 0: generic run-time count
template <int KS2, int NC>
__global__ __launch_bounds__(256) void splitk_finish_f16(const float *__restrict__ slabs, int ks2, int64_t slab_stride, int N,
                                                         FinishArgs f)
{
    __shared__ float sh[4];
    const int row = blockIdx.x, tid = threadIdx.x + blockIdx.y * (256 * NC);   // blockIdx.y > 0 only without fused LN
    float v[NC];
    const float *base = slabs + (size_t)row * N;
    if constexpr (KS2 > 0) {
        float part[NC][KS2];
#pragma unroll
        for (int i = 0; i < NC; ++i)
#pragma unroll
            for (int k = 0; k < KS2; ++k) {
                const int col = tid + 256 * i;
                part[i][k] = base[(size_t)k * slab_stride + (col < N ? col : N - 1)];   // clamped, never predicated: a
                // predicated load becomes a branch + s_waitcnt vmcnt(0) per column group (serialised round trips)
            }
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            float a = 0.f;
#pragma unroll
            for (int k = 0; k < KS2; ++k) a += part[i][k];
            v[i] = a;
        }
    } else {
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int col = tid + 256 * i;
            float a = 0.f;
            if (col < N) for (int k = 0; k < ks2; ++k) a += base[(size_t)k * slab_stride + col];
            v[i] = a;
        }
    }
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int col = tid + 256 * i;
        float a = v[i];
        if (col < N) {
            if (f.epi & EPI_BIAS) a += f.bias[col];
            if (f.epi & EPI_GELU) a = gelu_erf(a);
            if (f.epi & EPI_RES) a += (float)((const f16 *)f.R)[(size_t)row * f.ldr + col];
            const f16 h = (f16)a;
            if (f.kcache && col >= f.d) {
                const int pos = f.pos0[row];
                if (col < 2 * f.d) ((f16 *)f.kcache)[((size_t)row * f.n_ctx + pos) * f.d + (col - f.d)] = h;
                else ((f16 *)f.vcache)[((size_t)row * f.n_ctx + pos) * f.d + (col - 2 * f.d)] = h;
            } else {
                ((f16 *)f.C)[(size_t)row * f.ldc + col] = h;
            }
            a = (float)h;      // the LayerNorm below sees the stored (rounded) activation, like a separate LN launch would
        }
        v[i] = a;
    }
    if (!f.ln_out) return;
    const int wid = threadIdx.x >> 6;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) if (tid + 256 * i < N) s += v[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sh[wid] = s;
    __syncthreads();
    const float mean = (sh[0] + sh[1] + sh[2] + sh[3]) / (float)N;
    __syncthreads();
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) if (tid + 256 * i < N) { const float t = v[i] - mean; q += t * t; }
    q = wave_sum(q);
    if ((threadIdx.x & 63) == 0) sh[wid] = q;
    __syncthreads();
    const float rstd = 1.0f / sqrtf((sh[0] + sh[1] + sh[2] + sh[3]) / (float)N + 1e-5f);
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int col = tid + 256 * i;
        if (col < N) ((f16 *)f.ln_out)[(size_t)row * f.ld_ln + col] = (f16)((v[i] - mean) * rstd * f.ln_g[col] + f.ln_b[col]);
    }
}

// Experiment hook: SWX_PG_POLICY="NxK=ks2,NxK=ks2,..." pins the K split of given GEMM shapes (e.g. "5120x1280=1" runs the
// first MLP projection un-split on 80 fat workgroups, which lets the kernel finish bias + GELU itself).
int pg_policy(int N, int K)
{
    static const std::vector<std::array<int, 3>> table = [] {
        std::vector<std::array<int, 3>> t;
        const char *e = getenv("SWX_PG_POLICY");
        while (e && *e) {
            int n = 0, k = 0, c = 0, used = 0;
            if (sscanf(e, "%dx%d=%d%n", &n, &k, &c, &used) == 3 && used > 0) { t.push_back({n, k, c}); e += used; }
            else break;
            if (*e == ',') ++e;
        }
        return t;
    }();
    for (auto &r : table) if (r[0] == N && r[1] == K) return r[2];
    return 0;
}

int pg_ks2(int N, int K)
{
    const int units = K / 64;
    const int panels = (N + 63) / 64;
    const int pinned = pg_policy(N, K);
    if (pinned > 0 && units % pinned == 0 && units / pinned <= PG_MAXIT) return pinned;
    static const int target = [] { const char *e = getenv("SWX_PG_BLOCKS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 320; }();
    int want = (target + panels - 1) / panels;     // workgroups wanted per launch (tunable for experiments)
    int ks2 = 0;
    for (int c = 1; c <= units; ++c)
        if (units % c == 0 && units / c <= 10 && (ks2 == 0 || c <= want)) ks2 = c;
    return ks2;
}

int skinny_ks2(int N, int K)
{
    const int units = K / 128;                       // a wave's K slice must stay a multiple of 32
    const int panels = (N + 15) / 16;
    int want = 640 / panels;
    if (want < 1) want = 1;
    int ks2 = 0;
    for (int c = 1; c <= units; ++c)
        if (units % c == 0 && units / c <= 10 && (ks2 == 0 || c <= want)) ks2 = c;    // smallest legal c, grown up to `want`
    return ks2 ? ks2 : units;
}

}  // namespace

size_t swx_skinny_slab_floats(int M, int N, int K)
{
    if (M <= 0 || M > 128 || K % 128 != 0 || N > 256 * FIN_MAXC) return 0;
    const int a = skinny_ks2(N, K), b = pg_ks2(N, K);
    return (size_t)(a > b ? a : b) * M * N;
}

int swx_pg_splits(int N, int K) { return (K % 128 != 0 || N <= 0) ? 0 : pg_ks2(N, K); }

namespace {
int pg_launch(const void *A, int64_t lda, const void *W, int64_t ldw, int M, int N, int K, float *slabs, SlabRef *ref,
              const FinishArgs *direct, hipStream_t s);
}

int swx_gemm_pg(const void *A, int64_t lda, const void *W, int64_t ldw, int M, int N, int K, float *slabs, SlabRef *ref,
                hipStream_t s)
{
    return pg_launch(A, lda, W, ldw, M, N, K, slabs, ref, nullptr, s);
}

namespace {
// direct != null (only legal when the shape runs un-split): bias / GELU in the kernel's own epilogue, compute dtype out
int pg_launch(const void *A, int64_t lda, const void *W, int64_t ldw, int M, int N, int K, float *slabs, SlabRef *ref,
              const FinishArgs *direct, hipStream_t s)
{
    if (M <= 0 || N <= 0) return -4;
    if (M > 128 || K % 128 != 0 || N > 256 * FIN_MAXC || lda % 8 != 0 || ldw % 8 != 0) return -4;
    const int ks2 = pg_ks2(N, K);
    if (ks2 <= 0) return -4;
    GemmArgs g{};
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.M = M; g.N = N; g.K = K; g.epi = EPI_OUT_F32; g.res_mod = 1;
    if (direct) {
        if (ks2 != 1) return -4;
        g.epi = direct->epi & (EPI_BIAS | EPI_GELU); g.bias = direct->bias; g.C = direct->C; g.ldc = direct->ldc;
    }
    const int64_t stride = (int64_t)M * N;
    const int flags = swx_flags();
    dim3 grid(cdiv(N, 64), ks2);
    SwxProfScope prof(PC_GEMM_SKINNY, 2.0 * ((double)N * K + (double)M * K) + (double)M * N * 2, s);
    const int nit = K / (64 * ks2);           // K slice per workgroup in 64-wide chunks: <= 5 (default tuning), <= 10 or <= 20
#define SWX_PG2(MT, DIR) do { if (nit > 10) hipLaunchKernelGGL((gemm_f16_pg<MT, 20, DIR>), grid, dim3(256), 0, s, g, slabs, stride, ks2, flags); \
                              else if (nit > 5) hipLaunchKernelGGL((gemm_f16_pg<MT, 10, DIR>), grid, dim3(256), 0, s, g, slabs, stride, ks2, flags); \
                              else hipLaunchKernelGGL((gemm_f16_pg<MT, 5, DIR>), grid, dim3(256), 0, s, g, slabs, stride, ks2, flags); } while (0)
#define SWX_PG(MT) do { if (direct) SWX_PG2(MT, true); else SWX_PG2(MT, false); } while (0)
    switch (cdiv(M, 16)) {
        case 1: SWX_PG(1); break;
        case 2: SWX_PG(2); break;
        case 3: SWX_PG(3); break;
        case 4: SWX_PG(4); break;
        case 5: SWX_PG(5); break;
        case 6: SWX_PG(6); break;
        case 7: SWX_PG(7); break;
        default: SWX_PG(8); break;
    }
#undef SWX_PG
#undef SWX_PG2
    if (ref) { ref->slabs = slabs; ref->ks2 = ks2; ref->stride = stride; ref->N = N; ref->bias = nullptr; }
    return 0;
}
}  // namespace

int swx_gemm_skinny_splitk(const void *A, int64_t lda, const void *W, int64_t ldw, int M, int N, int K, float *slabs,
                           const FinishArgs &f, hipStream_t s)
{
    if (M <= 0 || N <= 0) return 0;
    SlabRef ref{};
    // an un-split shape (SWX_PG_POLICY) whose finish is only bias / GELU needs no finish launch at all
    if (pg_ks2(N, K) == 1 && K % 128 == 0 && !f.ln_out && !f.kcache && !(f.epi & EPI_RES) && f.C)
        return pg_launch(A, lda, W, ldw, M, N, K, slabs, &ref, &f, s);
    const int rc = swx_gemm_pg(A, lda, W, ldw, M, N, K, slabs, &ref, s);
    if (rc < 0) return rc;
    const int ks2 = ref.ks2;
    const int64_t stride = ref.stride;
    {
        SwxProfScope prof(PC_NORM, (double)ks2 * M * N * 4 + 4.0 * M * N, s);
        const int nc = cdiv(N, 256);
        // rows without a fused LayerNorm are split over blockIdx.y in 1280-column pieces (more workgroups, fewer erf per thread)
        const bool split = !f.ln_out && nc > 5;
        dim3 fg(M, split ? cdiv(N, 1280) : 1);
#define SWX_FIN(KS, NC) hipLaunchKernelGGL((splitk_finish_f16<KS, NC>), fg, dim3(256), 0, s, slabs, ks2, stride, N, f)
        if (split || nc <= 5) {
            switch (ks2) {
                case 1: SWX_FIN(1, 5); break;
                case 2: SWX_FIN(2, 5); break;
                case 4: SWX_FIN(4, 5); break;
                case 5: SWX_FIN(5, 5); break;
                case 8: SWX_FIN(8, 5); break;
                case 10: SWX_FIN(10, 5); break;
                case 16: SWX_FIN(16, 5); break;
                default: SWX_FIN(0, 5); break;
            }
        } else {
            SWX_FIN(0, 20);
        }
#undef SWX_FIN
    }
    SWX_CHECK_LAUNCH();
    return 0;
}

int swx_gemm(int dtype, const GemmArgs &g, int force_kernel, hipStream_t s)
{
    if (g.M <= 0 || g.N <= 0) return 0;
    if (dtype == SWX_F16) {
        if (g.K % 32 != 0 || g.lda % 8 != 0 || g.ldw % 8 != 0) return -4;   // tiled: K % 32, skinny: K % 128
        const bool skinny_ok = g.M <= 128 && g.K % 128 == 0 && g.K / 128 <= 10 && g.N <= 16384;   // vocabulary-sized N: tiled
        const bool use_skinny = force_kernel == 2 ? skinny_ok : ((force_kernel == 1 || force_kernel >= 4) ? false : skinny_ok);
        if (force_kernel == 2 && !skinny_ok) return -4;
        if (use_skinny) {
            SwxProfScope prof(PC_GEMM_SKINNY, 2.0 * ((double)g.N * g.K + (double)g.M * g.K) + (double)g.M * g.N * ((g.epi & EPI_OUT_F32) ? 4 : 2), s);
            dim3 grid(cdiv(g.N, 16));
            const int mt = cdiv(g.M, 16);
            switch (mt) {
                case 1: hipLaunchKernelGGL(gemm_f16_skinny<1>, grid, dim3(256), 0, s, g, (float *)nullptr, (int64_t)0, 1); break;
                case 2: hipLaunchKernelGGL(gemm_f16_skinny<2>, grid, dim3(256), 0, s, g, (float *)nullptr, (int64_t)0, 1); break;
                case 3: hipLaunchKernelGGL(gemm_f16_skinny<3>, grid, dim3(256), 0, s, g, (float *)nullptr, (int64_t)0, 1); break;
                case 4: hipLaunchKernelGGL(gemm_f16_skinny<4>, grid, dim3(256), 0, s, g, (float *)nullptr, (int64_t)0, 1); break;
                case 5: hipLaunchKernelGGL(gemm_f16_skinny<5>, grid, dim3(256), 0, s, g, (float *)nullptr, (int64_t)0, 1); break;
                case 6: hipLaunchKernelGGL(gemm_f16_skinny<6>, grid, dim3(256), 0, s, g, (float *)nullptr, (int64_t)0, 1); break;
                case 7: hipLaunchKernelGGL(gemm_f16_skinny<7>, grid, dim3(256), 0, s, g, (float *)nullptr, (int64_t)0, 1); break;
                default: hipLaunchKernelGGL(gemm_f16_skinny<8>, grid, dim3(256), 0, s, g, (float *)nullptr, (int64_t)0, 1); break;
            }
        } else {
            SwxProfScope prof(PC_GEMM_TILED, 2.0 * (double)g.M * g.N * g.K, s);
            dim3 grid(cdiv(g.N, BN), cdiv(g.M, BM));
            const bool glds_ok = g.K % 64 == 0 && ((uintptr_t)g.A % 16 == 0) && ((uintptr_t)g.W % 16 == 0);
            if (force_kernel >= 4 && !glds_ok) return -4;
            // 64-column tiles when 128-wide ones would leave CUs idle (encoder at batch 1: M = 1500, N = 1280 is 120 tiles of
            // 128 x 128 for 256 CUs); force_kernel 8 / 9 = always / never for the A/B
            const bool narrow = force_kernel == 8 || (force_kernel != 9 && g.N % 64 == 0 && (int64_t)grid.x * grid.y < 224);
            if (glds_ok && (force_kernel == 4 || force_kernel == 5))
                hipLaunchKernelGGL(gemm_f16_glds_d2_128, grid, dim3(256), 0, s, g);
            else if (glds_ok && force_kernel == 6)
                hipLaunchKernelGGL(gemm_f16_glds_s4_128, grid, dim3(256), 0, s, g);
            else if (glds_ok && narrow && (force_kernel >= 7 || (force_kernel == 0 && (swx_flags() & SWX_FLAG_GLDS_GEMM))))
                hipLaunchKernelGGL(gemm_f16_glds_s3_64, dim3(cdiv(g.N, 64), grid.y), dim3(256), 0, s, g);
            else if (glds_ok && (force_kernel >= 7 || (force_kernel == 0 && (swx_flags() & SWX_FLAG_GLDS_GEMM))))
                hipLaunchKernelGGL(gemm_f16_glds_s3_128, grid, dim3(256), 0, s, g);
            else
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
