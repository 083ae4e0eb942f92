// swx_gemm.hip -- C = epilogue(A[M,K] * W[N,K]^T) on gfx950 MFMA.
//
// Every Linear / Conv1d of the Whisper encoder and decoder (upstream whisper/model.py, reached from
// stable_whisper/decode.py:27-30,40 and timing.py:59-61) goes through these kernels.  Weights keep the
// checkpoint's [out, in] layout, so both operands are K-contiguous ("B^T input") and each MFMA fragment is one
// 16-byte load.  The kernels (swx_gemm_plan_f16 at the end of the file says which launch gets which):
//   * tiled f16, register-staged   128x128x64 tile, 4 waves (2x2), 4x4 v_mfma_f32_16x16x32_f16 per wave: K % 64 != 0, and the
//                                  bit-identity reference of the four below
//   * gemm_f16_glds_128 / _64      the same tile filled by LDS-DMA, one operand buffer, three workgroups per CU overlap each other
//   * gemm_f16_ring<64|128, 3>     launches with <= 1 workgroup per CU: three LDS stages, LDS-DMA from inline asm, counted waits
//   * gemm_f16_big8                256x256 tile, 8 waves, LDS-DMA into a ring of eight half-tile slots, four phases per K tile, the
//                                  two row groups of waves half a phase apart: launches that fill whole rounds of the 256 CUs
//   * tiled f32                    the 128x128 tiling on v_mfma_f32_16x16x4_f32 (exact f32 fma chain)     (strict-parity mode)
//   * skinny f16                   M <= 128 rows: one 16-column weight panel per workgroup, K split over its 4 waves, weights
//                                  streamed straight from HBM into MFMA fragments (no LDS), LDS reduction
// Epilogue (f32): +bias, GELU(erf), +f32 residual indexed by row % res_mod (positional embedding), +T residual,
// store as T or f32.
#include <cstdlib>
#include <type_traits>
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
// EPI_KV with GemmArgs::P (round 6): the decode-step cross-attention's fragment-ordered copy (swx_attn.hip::xkv_pack_kernel's index maps)
// written from the projection's epilogue.  K: the 16-byte piece (key, head, dims 8 c .. 8 c + 7) is lane (g = c & 3, qn) of fragment 2 t + (c >> 2)
// of the key's 32-key block, key % 32 = (qn >> 2) * 8 + 4 t + (qn & 3).  V^T: the 8-byte piece (4 keys from key0 = a multiple of 4, head, dim) is
// half (key0 % 8) / 4 of lane (g = (key0 % 32) / 8, qn = dim % 16) of fragment dim / 16.  Keys [vt_s, p_nkpad) of a head are zeros.
__device__ __forceinline__ f16 *xkv_packed_k_piece(const GemmArgs &g, int wb, int key, int col)
{
    const int h = col >> 6, c = (col & 63) >> 3, kq = key & 31;
    const int frag = (((kq >> 2) & 1) << 1) + (c >> 2), lane_ = (c & 3) * 16 + (kq >> 3) * 4 + (kq & 3);
    return (f16 *)g.P + (size_t)wb * g.p_bs + (size_t)h * swx_xkv_packed_elems_per_head(g.vt_s) + ((size_t)(key >> 5) * 4 + frag) * 512 + lane_ * 8;
}
__device__ __forceinline__ f16 *xkv_packed_v_piece(const GemmArgs &g, int wb, int key0, int vcol)
{
    const int h = vcol >> 6, dd = vcol & 63, kq = key0 & 31;
    const int64_t per_head = swx_xkv_packed_elems_per_head(g.vt_s);
    return (f16 *)g.P + (size_t)wb * g.p_bs + (size_t)h * per_head + per_head / 2 + ((size_t)(key0 >> 5) * 4 + (dd >> 4)) * 512 +
           ((kq >> 3) * 16 + (dd & 15)) * 8 + ((kq & 7) >> 2) * 4;
}

template <int NJ>      // NJ 16-column fragments per wave: tile width BNT = 32 NJ (128 or 64 columns)
__device__ __forceinline__ void tile_epilogue_f16_one(const GemmArgs &g, unsigned char *smem, f32x4 (&acc)[4][NJ], int m0, int n0,
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
                    if (g.P) {                                          // ... and the fragment-ordered copy (with its zeroed key padding)
                        *(f16x4 *)xkv_packed_v_piece(g, wb, sidx, gn - g.p_vcol0) = o;
                        if (sidx + 4 == g.vt_s) {
                            const f16x4 z = {(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
                            for (int kz = g.vt_s; kz < g.p_nkpad; kz += 4) *(f16x4 *)xkv_packed_v_piece(g, wb, kz, gn - g.p_vcol0) = z;
                        }
                    }
                    if (g.vt_zero_pad && sidx + 4 == g.vt_s) {          // the row group that ends a batch item: its columns' key padding
                        const f16x4 z = {(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
                        for (int zp = 4; sidx + zp < g.vt_kp; zp += 4) *(f16x4 *)(cbase + (size_t)gn * g.vt_kp + zp) = z;
                    }
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
                gelu_erf_n<8>(v);
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
                if (g.P && (g.epi & EPI_CBATCH)) {                      // K of the cross-attention: ... and the fragment-ordered copy
                    const int wb = gm / g.vt_s, key = gm - wb * g.vt_s;
                    *(f16x8 *)xkv_packed_k_piece(g, wb, key, gn) = o;
                    if (key + 1 == g.vt_s)
                        for (int kz = g.vt_s; kz < g.p_nkpad; ++kz) *(f16x8 *)xkv_packed_k_piece(g, wb, kz, gn) = (f16x8)(f16)0;
                }
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

// EPI_KV (round 6): the K and the V projection of a decoder layer's cross-attention as ONE launch over the fused weight rows -- a
// tile left of N / 2 takes the K epilogue (rows grouped per window), a tile right of it the V epilogue (transposed per head into
// C2, columns counted from N / 2).  Workgroup-uniform; per element the arithmetic of the two separate launches.
template <int NJ>
__device__ __forceinline__ void tile_epilogue_f16(const GemmArgs &g, unsigned char *smem, f32x4 (&acc)[4][NJ], int m0, int n0,
                                                     int tid, int lane, int wm, int wn)
{
    if (g.epi & EPI_QKV_VT) {
        // the encoder's fused Q | K | V projection: V (the last third of the columns; the third is a multiple of the tile width)
        // leaves transposed per head -- what swx_transpose_v did in a launch of its own; per element the plain epilogue's value
        GemmArgs h = g;
        const int split = (g.N / 3) * 2;
        h.epi = g.epi & ~EPI_QKV_VT;
        if (n0 >= split) { h.epi |= EPI_STORE_VT; h.C = (f16 *)g.C2 - (size_t)split * g.vt_kp; h.bias = g.bias; }
        tile_epilogue_f16_one<NJ>(h, smem, acc, m0, n0, tid, lane, wm, wn);
        return;
    }
    if (!(g.epi & EPI_KV)) { tile_epilogue_f16_one<NJ>(g, smem, acc, m0, n0, tid, lane, wm, wn); return; }
    GemmArgs h = g;
    const int half = g.N >> 1;
    if (n0 >= half) {
        h.epi = (g.epi & ~(EPI_KV | EPI_CBATCH)) | EPI_STORE_VT;
        h.C = (f16 *)g.C2 - (size_t)half * g.vt_kp;          // the V epilogue addresses by the absolute column
    } else {
        h.epi = (g.epi & ~(EPI_KV | EPI_STORE_VT)) | EPI_CBATCH;
    }
    tile_epilogue_f16_one<NJ>(h, smem, acc, m0, n0, tid, lane, wm, wn);
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
// Now: one operand buffer, two barriers per K step, 33.8 KB of LDS -> three workgroups per CU: the occupancy, not a software
// pipeline, overlaps one workgroup's loads and epilogue with its neighbours' MFMAs.  Measured at M = 30000
// (profiles/r02_kb_gemm_gen2.txt, TFLOP/s, first kernel -> double buffer, 2 per CU -> single buffer, 4 per CU -> single buffer,
// 3 per CU = this kernel): N = K = 1280: 522 -> 737 -> 783 -> 768; N = 3840: 536 -> 704 -> 754 -> 737; N = 5120: 523 -> 748 ->
// 809 -> 788; K = 5120: 715 -> 783 -> 816 -> 854.  The variants that lost were deleted in round 3.  Bit-identical to the
// register-staged kernel (same MFMA order per accumulator, same f32 epilogue arithmetic; tests/hw_checks/gemm_glds_check.py).
constexpr int GL_TILE = 128 * 128;          // bytes of one operand tile: 128 rows x 64 halfs

template <int BNT>     // BNT = tile width: 128, or 64 for shapes whose 128-wide tiles would not fill the chip
__device__ __forceinline__ void gemm_f16_glds_body(const GemmArgs &g)
{
    constexpr int NJ = BNT / 32;                       // 16-column fragments per wave (2 x 2 waves: 64 rows x BNT / 2 columns each)
    constexpr int TB = BNT * 128;                      // bytes of the B operand tile (BNT rows x 64 halfs)
    constexpr int STAGE = GL_TILE + TB;
    constexpr int MAIN = STAGE;
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
    for (int kt = 0; kt < KT; ++kt) {       // one operand buffer, two barriers per K step: three workgroups per CU overlap each other
        stage(kt, 0);
        __syncthreads();
        compute(0);
        __syncthreads();
    }
    tile_epilogue_f16<NJ>(g, smem, acc, m0, n0, tid, lane, wm, wn);
}

// the instantiations in use (second launch bound = workgroups per CU the register allocation must allow)
__global__ __launch_bounds__(256, 3) void gemm_f16_glds_128(GemmArgs g) { gemm_f16_glds_body<128>(g); }
__global__ __launch_bounds__(256, 3) void gemm_f16_glds_64(GemmArgs g) { gemm_f16_glds_body<64>(g); }

// ------------------------------------------------------------------------------- tiled f16, direct-to-LDS ring
// The kernel above hides a K step's memory round trip behind the OTHER workgroups of its CU (three resident).  Shapes with
// fewer than ~two tiles per CU have nobody to hide behind -- the encoder at batch 1 (align(): M = 1500; N = 1280 is 240 tiles
// of 128 x 64) runs stage -> wait -> compute serially, ~0.9 us per K step for 0.1-0.2 us of MFMAs: 18.6 us at K = 1280,
// 65.8 us at K = 5120 (profiles/r02_kb_gemm_narrow_tiles.txt).  Here ONE workgroup keeps NST - 1 K steps in flight: a ring
// of NST = 3 operand stages in LDS (72 / 96 KB), filled by LDS-DMA issued from inline asm (a DMA hipcc can see makes it wait vmcnt(0) before
// the next LDS read and inside __syncthreads(); cdna_hip_programming.md "glds with >1 tile in flight"), retired by counted
// `s_waitcnt vmcnt(n)` + a raw `s_barrier`.  Per K step: [wait until stage kt has landed for this wave (the NST - 2 younger
// stages stay in flight) and this wave's LDS reads of step kt - 1 are back] -> barrier (now true for every wave) -> refill
// the stage step kt - 1 used -> MFMAs of step kt.  A DMA's data is read one barrier after the wait that retires it, a stage
// is rewritten one barrier after its last read returned.  Same tile, swizzle, MFMA order per accumulator and epilogue as
// the kernel above: bit-identical results (tests/hw_checks/gemm_glds_check.py).
__device__ __forceinline__ void glds16_asm(const void *gsrc, unsigned lds_dst)
{
    unsigned keep;     // M0 is the DMA's LDS base and belongs to hipcc: saved and restored inside the statement
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N>
__device__ __forceinline__ void ring_wait_barrier()
{
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

template <int BNT, int NST>
__device__ __forceinline__ void gemm_f16_ring_body(const GemmArgs &g)
{
    constexpr int NJ = BNT / 32;
    constexpr int TB = BNT * 128;
    constexpr int STAGE = GL_TILE + TB;
    constexpr int L = 4 + NJ;                            // DMA instructions per wave and stage
    static_assert(NST == 3 || NST == 4, "ring depth");      // depth 4 measured no better than 3 anywhere and was not kept
    static_assert((NST - 2) * L < 64, "vmcnt is a 6-bit field");
    extern __shared__ __attribute__((aligned(1024))) unsigned char ring[];     // NST stages; the f32 epilogue tile afterwards
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
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
    const unsigned ring0 = (unsigned)(uintptr_t)(lds_void *)ring;
    auto stage = [&](int kt, int buf) {
        const unsigned ta = ring0 + buf * STAGE, tb = ta + GL_TILE;
#pragma unroll
        for (int c = 0; c < 4; ++c) glds16_asm(srcA[c] + kt * 64, ta + (wave_u * 4 + c) * 1024);
#pragma unroll
        for (int c = 0; c < NJ; ++c) glds16_asm(srcW[c] + kt * 64, tb + (wave_u * NJ + c) * 1024);
    };
    const int KT = g.K / 64;                             // the launcher guarantees KT >= NST - 1
    const int fr = lane & 15, fs = lane >> 4;
    auto compute = [&](int buf) {
        const unsigned char *ta = ring + buf * STAGE, *tb = ta + GL_TILE;
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
#pragma unroll
    for (int st = 0; st < NST - 1; ++st) stage(st, st);
    int buf = 0, nbuf = NST - 1;                         // stage of step kt, stage the next refill goes to
    for (int kt = 0; kt < KT; ++kt) {
        const int ahead = KT - 1 - kt;                   // K steps after this one: min(ahead, NST - 2) stages may stay in flight
        if (ahead >= NST - 2) ring_wait_barrier<(NST - 2) * L>();
        else if (NST == 4 && ahead == 1) ring_wait_barrier<L>();
        else ring_wait_barrier<0>();
        if (kt + NST - 1 < KT) stage(kt + NST - 1, nbuf);
        compute(buf);
        buf = buf + 1 == NST ? 0 : buf + 1;
        nbuf = nbuf + 1 == NST ? 0 : nbuf + 1;
    }
    __syncthreads();                                     // every wave is done with the ring: it becomes the epilogue's f32 tile
    tile_epilogue_f16<NJ>(g, ring, acc, m0, n0, tid, lane, wm, wn);
}

template <int BNT, int NST>
__global__ __launch_bounds__(256) void gemm_f16_ring(GemmArgs g) { gemm_f16_ring_body<BNT, NST>(g); }

template <int BNT, int NST>
static void launch_ring(const GemmArgs &g, hipStream_t s)
{
    constexpr int stage_bytes = GL_TILE + BNT * 128, epi_bytes = 64 * (BNT + 4) * 4;
    constexpr int lds = NST * stage_bytes > epi_bytes ? NST * stage_bytes : epi_bytes;
    hipLaunchKernelGGL((gemm_f16_ring<BNT, NST>), dim3(cdiv(g.N, BNT), cdiv(g.M, BM)), dim3(256), lds, s, g);
}

// ------------------------------------------------------------------------------------ tiled f16, 256 x 256 tile, 8 waves
// The large-M shapes (encoder at 20 windows: M = 30 000).  The 128 x 128 tile above reads 16 KB of fragments from LDS per wave
// and K step for 32 MFMAs -- 64 KB per workgroup against an LDS port of 128 B/clk: exactly as many LDS cycles as MFMA cycles,
// which is what holds that kernel at 0.29-0.34 of the MFMA peak however it is scheduled.  Here a wave owns 128 x 64 of a
// 256 x 256 tile (8 waves as 2 x 4): 24 fragment reads feed 64 MFMAs, 0.75 LDS cycles per MFMA cycle.  One workgroup per CU
// (two waves per SIMD, 128 accumulator registers each), 128 KB of operand stages filled by the ring kernel's asm LDS-DMA.  Plain
// epilogues only (bias, GELU, f16 residual, f16 rows with 16-byte aligned leading dimensions): the encoder's four projections;
// everything else stays on the kernels above.  Same MFMA order per accumulator: bit-identical results
// (tests/hw_checks/gemm_glds_check.py, gemm_big8_check.py).
constexpr int BG = 256;                                   // tile edge
constexpr int BG_CLD = BG + 4;                            // f32 epilogue row

// Epilogue of the 256 x 256 kernels: four passes of 64 rows through a f32 tile in LDS, 16-byte row-contiguous stores
// (tile_epilogue_f16's plain path).  acc[i][j] = rows wm * 128 + i * 16 + ..., columns wn * 64 + j * 16 + ... of the tile.
__device__ __forceinline__ void big_tile_epilogue(const GemmArgs &g, unsigned char *ring, f32x4 (&acc)[8][4], int m0, int n0, int tid,
                                                  int lane, int wm, int wn)
{
    float (*Cs)[BG_CLD] = (float (*)[BG_CLD])ring;
    const int col_l = lane & 15, row_l = (lane >> 4) * 4;
    const int c8 = (tid & 31) * 8, rb = tid >> 5, gn = n0 + c8;
    const bool col_ok = gn < g.N, full = gn + 8 <= g.N;
    float bv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bv[e] = ((g.epi & EPI_BIAS) && gn + e < g.N) ? g.bias[gn + e] : 0.f;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        f16x8 rv[4];
        if (g.epi & EPI_RES) {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int gm = m0 + p * 64 + it * 16 + rb;
                rv[it] = (gm < g.M && full) ? *(const f16x8 *)((const f16 *)g.R + (size_t)gm * g.ldr + gn) : (f16x8)(f16)0;
            }
        }
        if (wm == (p >> 1)) {
#pragma unroll
            for (int ii = 0; ii < 4; ++ii)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) Cs[ii * 16 + row_l + r][wn * 64 + j * 16 + col_l] = acc[(p & 1) * 4 + ii][j][r];
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = it * 16 + rb, gm = m0 + p * 64 + row;
            if (gm >= g.M || !col_ok) continue;
            const f32x4 lo = *(const f32x4 *)&Cs[row][c8], hi = *(const f32x4 *)&Cs[row][c8 + 4];
            float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += bv[e];
            if (g.epi & EPI_GELU) {
                gelu_erf_n<8>(v);
            }
            f16 *cp = (f16 *)g.C + (size_t)gm * g.ldc + gn;
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
        if (p < 3) __syncthreads();
    }
}

// ------------------------------------------------------ tiled f16, 256 x 256 tile, 8 waves, half-tile ring, staggered wave groups
// Generation 1 of this kernel (round 3: two 64 KB K-step stages, one `vmcnt(0)` + barrier per step; deleted in round 4) left the
// MFMA pipe idle twice per K step: its one LDS-DMA stage in flight was issued at the start of the step whose MFMAs (2 048 clocks
// for 64 KB) are shorter than a loaded fabric round trip, and its eight waves read fragments and multiplied in lockstep (1 536
// clocks of LDS reads per step that no MFMA covered).  This generation keeps the tile, the swizzle, the wave -> accumulator map
// and the MFMA order per accumulator (bit-identical results) and changes the schedule (cdna_hip_programming.md, "The 256^2
// 8-phase template").  Measured against generation 1 (profiles/r04_kb_gemm_big8.txt, M = 30 000, TFLOP/s): N = 3840: 962 -> 1 039;
// N = 5120 + GELU: 840 -> 882; K = 5120: 919 -> 1 044; N = 2560: 974 -> 1 035; in the headline pass 562 / 428 / 297 -> 545 / 388 /
// 280 us per launch (profiles/r04_big8_pass_kernels.csv).  Still 0.35 - 0.42 of the MFMA peak: four to five half-tiles in flight
// per CU (16 MB over the chip) cover ~2 us of fabric latency at the 7.8 TB/s of operand traffic 1 000 TFLOP/s needs -- LDS
// capacity (128 of 160 KB), not the schedule, bounds the depth.
//  * the two 64 KB operand buffers are eight HALF-TILE slots (A0 A1 B0 B1 of an even and an odd K tile).  Half h of A = the 64
//    rows each row group of waves multiplies in its phases with that half (tile rows wm * 128 + h * 64 + ..), half h of B = the
//    32 columns of each column group (wn * 64 + h * 32 + ..): a phase needs whole halves, and a slot is refilled as soon as its
//    last fragment read has retired.  Loads are issued in consumption order S_k = A0 B0 B1 A1 of tile k / 4; phase P issues
//    S_(P+7), so four to five half-tiles (64 - 80 KB per CU) are in flight at any time instead of 0 - 64 KB;
//  * a K tile is four phases = the four quadrants of a wave's 8 x 4 accumulators, (A0,B0) (A0,B1) (A1,B1) (A1,B0): 12 / 4 / 8 / 0
//    fragment reads (B0 stays in registers) for 16 MFMAs each.  Phase P of a wave:
//        R_P  fragment reads            W_P  s_waitcnt vmcnt(8): the halves phase P + 1 reads have landed (this wave's part)
//        X_P  barrier                   s_waitcnt lgkmcnt(0)     I_P  issue S_(P+7)          M_P  16 MFMAs          Y_P  barrier
//  * the row groups wm = 0 / 1 (the two waves of every SIMD) run half a phase apart: group 1 passes one extra barrier before
//    its first phase, so its R / W section runs under group 0's MFMAs and the other way round.
// Ordering, by construction (nothing here is "seen to work"): RAW -- a half read in R_(P+1) was retired by EVERY wave's W_P
// before that wave's X_P; a reader passes its Y_P first, which completes only after all waves of the other group called X_P
// (group 0 reads) or X_(P+1) (group 1 reads).  WAR -- I_P rewrites the slot of S_(P-1), whose last fragment reads are in
// R_(P-1) or earlier (A0: R_(P-1); B0, B1, A1: R_(P-2)); every wave retires its reads (lgkmcnt(0)) before its M of the same
// phase, i.e. before its Y_(P-1), and I_P follows X_P, which completes after the other group called Y_(P-1) (group 0 issues) or
// Y_P (group 1 issues).  The last tile's waits count what is really in flight (4, 2, 0).  K >= 128.
constexpr int B8_HALF = 128 * 128;                        // bytes per half-tile: 128 rows x 64 halfs
constexpr int B8_BUF = 4 * B8_HALF;                       // A0 | A1 | B0 | B1 of one K tile

template <int N>
__device__ __forceinline__ void b8_wait_vm()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void b8_barrier()
{
    asm volatile("s_barrier" ::: "memory");
}
__device__ __forceinline__ void b8_wait_lds()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

__global__ __launch_bounds__(512) void gemm_f16_big8(GemmArgs g)
{
    extern __shared__ __attribute__((aligned(1024))) unsigned char ring[];     // 2 x 4 half-tiles = 128 KB; the epilogue's f32 tile after
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int wm = wave >> 2, wn = wave & 3;
    const int wm_u = wave_u >> 2;
    int bx = blockIdx.x, by = blockIdx.y;
    {
        // Workgroup `orig` runs on XCD orig % 8 (round-robin dispatch); XCD x takes the contiguous range [start(x), start(x + 1)) of
        // a tile LIST, its k-th workgroup the k-th entry of that range -- so the ~32 tiles an XCD runs at one time are 32 neighbours
        // of the list.  The list is row-major.  At N = 5120 (20 column tiles) those 32 neighbours span every column, i.e. the whole
        // 13.1 MB weight matrix streams through the XCD's 4 MB L2 once per 256-row band (counters, round 5: 1.49 GB per launch for
        // 397 MB algorithmic).  Round 6 built the alternative (tile_order 1, SWX_FLAG_BIG8_GROUPED): GROUPS of 4 column tiles walked
        // along M, the last group taking the remainder -- 32 neighbours = 8 row bands x 4 column bands, the group's weight bands
        // (2.6 MB at K = 1280) resident in L2 for the whole walk.  A pure renumbering, bit-identical
        // (tests/hw_checks/gemm_big8_check.py runs both orders) -- and NOT faster: 458 vs 453 us at N = 5120 + GELU, 384 vs 372 us at
        // K = 5120, 291 vs 284 us at N = 3840 (profiles/r06_c2_kb_gemm_big_tile_order.txt), 427.0 vs 425.5 ms per headline pass.  The
        // re-streamed operands come out of the 256 MB Infinity Cache, whose bandwidth the launch does not exhaust: the extra
        // fabric traffic costs nothing, so the default stays row-major.
        const int gx = gridDim.x, gy = gridDim.y, nwg = gx * gy, orig = by * gx + bx;
        const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
        const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
        if (g.tile_order != 1 || gx <= 4) { bx = wg % gx; by = wg / gx; }
        else {
            constexpr int GN = 4;
            const int nfull = gx / GN, per = GN * gy;            // tiles per full group
            const int grp = wg / per;
            if (grp < nfull) { const int rem = wg - grp * per; by = rem / GN; bx = grp * GN + (rem - by * GN); }
            else { const int gl = gx - nfull * GN, rem = wg - nfull * per; by = rem / gl; bx = nfull * GN + (rem - by * gl); }
        }
    }
    const int m0 = by * BG, n0 = bx * BG;
    const f16 *A = (const f16 *)g.A;
    const f16 *W = (const f16 *)g.W;

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // a wave stages two 8-row chunks of a half-tile (local rows lr = (wave * 2 + c) * 8 + lane / 8; rows past M / N clamped)
    const f16 *sA[2][2], *sB[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int lr = (wave * 2 + c) * 8 + (lane >> 3);
            const int slot = (lane & 7) ^ ((lr >> 1) & 7);
            const int ra = (lr >> 6) * 128 + h * 64 + (lr & 63);
            const int rb = (lr >> 5) * 64 + h * 32 + (lr & 31);
            const int gm = m0 + ra < g.M ? m0 + ra : g.M - 1;
            const int gn = n0 + rb < g.N ? n0 + rb : g.N - 1;
            sA[h][c] = A + (size_t)gm * g.lda + slot * 8;
            sB[h][c] = W + (size_t)gn * g.ldw + slot * 8;
        }
    typedef __attribute__((address_space(3))) void lds_void;
    const unsigned ring0 = (unsigned)(uintptr_t)(lds_void *)ring;
    // half-tile `which` (0 A0, 1 B0, 2 B1, 3 A1: the consumption order) of K tile kt into its slot
    auto issue = [&](int kt, auto which_c) {
        constexpr int which = decltype(which_c)::value;
        constexpr bool isA = which == 0 || which == 3;
        constexpr int h = which >= 2 ? 1 : 0;
        constexpr int off = (isA ? 0 : 2 * B8_HALF) + h * B8_HALF;
        const unsigned dst = ring0 + (kt & 1) * B8_BUF + off + wave_u * 2048;
#pragma unroll
        for (int c = 0; c < 2; ++c) glds16_asm((isA ? sA[h][c] : sB[h][c]) + kt * 64, dst + c * 1024);
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;

    const int KT = g.K / 64;                              // the launcher guarantees KT >= 2
    const int fr = lane & 15, fs = lane >> 4;
    f16x8 fa[4][2], fb0[2][2], fb1[2][2];                 // [fragment][kk]: the current A half, B0, B1
    auto read_a = [&](int buf, int h) {
        const unsigned char *t = ring + buf * B8_BUF + h * B8_HALF;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int lr = wm * 64 + i * 16 + fr;
                fa[i][kk] = *(const f16x8 *)(t + lr * 128 + (((kk * 4 + fs) ^ ((lr >> 1) & 7)) << 4));
            }
    };
    auto read_b = [&](int buf, int h, f16x8 (&fb)[2][2]) {
        const unsigned char *t = ring + buf * B8_BUF + 2 * B8_HALF + h * B8_HALF;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int lr = wn * 32 + j * 16 + fr;
                fb[j][kk] = *(const f16x8 *)(t + lr * 128 + (((kk * 4 + fs) ^ ((lr >> 1) & 7)) << 4));
            }
    };
    // 16 MFMAs of quadrant (ih, jh); per accumulator kk = 0 then 1, K tiles ascending: the order of every tiled f16 kernel
    auto quad = [&](auto ih_c, auto jh_c, const f16x8 (&fb)[2][2]) {
        constexpr int ih = decltype(ih_c)::value, jh = decltype(jh_c)::value;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[ih * 4 + i][jh * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i][kk], fb[j][kk], acc[ih * 4 + i][jh * 2 + j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    // MODE 0: a tile with at least two tiles after it; 1: the second-last tile; 2: the last tile
    auto tile = [&](auto mode_c, int t) {
        constexpr int MODE = decltype(mode_c)::value;
        const int buf = t & 1;
        // phase 1: (A0, B0)
        read_a(buf, 0);
        read_b(buf, 0, fb0);
        b8_wait_vm<MODE == 2 ? 2 : 8>();                  // B1 of this tile
        b8_barrier();
        b8_wait_lds();
        if (MODE < 2) issue(t + 1, I3());                 // A1 of tile t + 1 (slot last read in phase 3 of tile t - 1)
        quad(I0(), I0(), fb0);
        b8_barrier();
        // phase 2: (A0, B1)
        read_b(buf, 1, fb1);
        b8_wait_vm<MODE == 2 ? 0 : 8>();                  // A1 of this tile
        b8_barrier();
        b8_wait_lds();
        if (MODE == 0) issue(t + 2, I0());                // A0 of tile t + 2 (slot last read in phase 1 of this tile)
        quad(I0(), I1(), fb1);
        b8_barrier();
        // phase 3: (A1, B1)
        read_a(buf, 1);
        b8_barrier();
        b8_wait_lds();
        if (MODE == 0) issue(t + 2, I1());                // B0 of tile t + 2 (slot last read in phase 1 of this tile)
        quad(I1(), I1(), fb1);
        b8_barrier();
        // phase 4: (A1, B0)
        if (MODE == 0) b8_wait_vm<8>();                   // A0, B0 of tile t + 1
        if (MODE == 1) b8_wait_vm<4>();
        b8_barrier();
        if (MODE == 0) issue(t + 2, I2());                // B1 of tile t + 2 (slot last read in phase 2 of this tile)
        quad(I1(), I0(), fb0);
        b8_barrier();
    };

    issue(0, I0()); issue(0, I1()); issue(0, I2()); issue(0, I3());
    issue(1, I0()); issue(1, I1()); issue(1, I2());
    b8_wait_vm<10>();                                     // A0, B0 of tile 0
    b8_barrier();
    if (wm_u == 1) b8_barrier();                          // row group 1 runs half a phase behind
    int t = 0;
    for (; t < KT - 2; ++t) tile(I0(), t);
    tile(I1(), t);
    tile(I2(), t + 1);
    if (wm_u == 0) b8_barrier();                          // pairs with row group 1's last barrier: every wave is done with the ring
    big_tile_epilogue(g, ring, acc, m0, n0, tid, lane, wm, wn);
}

// ------------------------------------------------------------------------------------------------ tiled f32
constexpr int BK32 = 16, LD32 = BK32 + 1;

// BMT x BNT = tile (128 x 128; 64 x 32 / 64 x 16 for launches with few rows, where 128-wide tiles leave N / 128 = 10-40 workgroups on
// 256 CUs and one CU's exact-f32 MFMA rate bounds the launch), ST = K
// steps of operands held in REGISTERS ahead of the one being multiplied.  Round 4: with ST = 2 (one step ahead) every 16-deep K
// step waited out a global round trip -- 2.6 us per step, 210 us for a K = 1280 launch at 100 rows: the decode-step projections of
// the strict-f32 mode, 3.9 s of its 5.8-s pass.  Operands now travel ST - 1 steps ahead.  Per output element the arithmetic is
// unchanged in every variant (one accumulator, K ascending in steps of 4), so all variants are bit-identical.
template <int BMT, int BNT, int ST>
__global__ __launch_bounds__(256) void gemm_f32_tiled(GemmArgs g)
{
    constexpr int WN = BNT >= 128 ? 2 : 1, WM = 4 / WN;                             // waves across the tile's columns / rows
    constexpr int WROWS = BMT / WM, WCOLS = BNT / WN;                                // a wave's share of the BMT x BNT tile
    constexpr int NI = WROWS / 16, NJ = WCOLS / 16;
    constexpr int NA = (BMT * 4 + 255) / 256, NB = (BNT * 4 + 255) / 256;            // float4 loads of the A / B tile per thread
    static_assert(NI >= 1 && NJ >= 1, "a wave owns at least one 16 x 16 fragment");
    __shared__ float As[2][BMT][LD32];
    __shared__ float Bs[2][BNT][LD32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = WN == 2 ? wave >> 1 : wave, wn = WN == 2 ? wave & 1 : 0;
    const int m0 = blockIdx.y * BMT, n0 = blockIdx.x * BNT;
    const float *A = (const float *)g.A;
    const float *W = (const float *)g.W;

    f32x4 acc[NI][NJ];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    f32x4 ra[ST][NA], rb[ST][NB];
    auto load_regs = [&](int st, int k0) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int c = tid + 256 * i, row = c >> 2, kc = (c & 3) * 4;
            const int gm = m0 + row;
            ra[st][i] = (row < BMT && gm < g.M) ? *(const f32x4 *)(A + (size_t)gm * g.lda + k0 + kc) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int c = tid + 256 * i, row = c >> 2, kc = (c & 3) * 4;
            const int gn = n0 + row;
            rb[st][i] = (row < BNT && gn < g.N) ? *(const f32x4 *)(W + (size_t)gn * g.ldw + k0 + kc) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };
    auto store_lds = [&](int st, int buf) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int c = tid + 256 * i, row = c >> 2, kc = (c & 3) * 4;
            if (row < BMT) {
#pragma unroll
                for (int e = 0; e < 4; ++e) As[buf][row][kc + e] = ra[st][i][e];
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int c = tid + 256 * i, row = c >> 2, kc = (c & 3) * 4;
            if (row < BNT) {
#pragma unroll
                for (int e = 0; e < 4; ++e) Bs[buf][row][kc + e] = rb[st][i][e];
            }
        }
    };
    auto compute = [&](int cur) {
        const int fr = lane & 15, fk = lane >> 4;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            float a[NI], b[NJ];
#pragma unroll
            for (int i = 0; i < NI; ++i) a[i] = As[cur][wm * WROWS + i * 16 + fr][kk * 4 + fk];
#pragma unroll
            for (int j = 0; j < NJ; ++j) b[j] = Bs[cur][wn * WCOLS + j * 16 + fr][kk * 4 + fk];
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    };

    const int KT = g.K / BK32;
    // register stage of K step kt is kt % ST; steps 0 .. ST - 2 are requested up front
#pragma unroll
    for (int st = 0; st < ST - 1; ++st) if (st < KT) load_regs(st, st * BK32);
    store_lds(0, 0);
    __syncthreads();
    for (int kt0 = 0; kt0 < KT; kt0 += ST) {
#pragma unroll
        for (int u = 0; u < ST; ++u) {                       // (unrolled: the register stages are compile-time indices)
            const int kt = kt0 + u;
            if (kt < KT) {
                const int cur = kt & 1;
                if (kt + ST - 1 < KT) load_regs((u + ST - 1) % ST, (kt + ST - 1) * BK32);
                compute(cur);
                if (kt + 1 < KT) store_lds((u + 1) % ST, cur ^ 1);
                __syncthreads();
            }
        }
    }

    const int col_l = lane & 15, row_l = (lane >> 4) * 4;
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                epilogue_store<float>(g, m0 + wm * WROWS + i * 16 + row_l + r, n0 + wn * WCOLS + j * 16 + col_l, acc[i][j][r]);
}

// ------------------------------------------------------------------------------------------------ f32, few rows (round 5)
// The strict-f32 launches with at most 128 rows (decode step at 100 rows, prefill, one window's scoring pass) on 64 x 16 NJ tiles:
// the arithmetic of gemm_f32_tiled element for element (one accumulator, K ascending in steps of 4, k-slot g of step s = k 4 s + g:
// bit-identical, tests/test_gpu_kernels.py), restaged.  gemm_f32_tiled<64, 16 | 32> spent a barrier, a scalar LDS round trip and
// 17-word rows on every 16-deep K step (4 MFMAs = 128 MFMA clocks per ~1 000-clock step: 36-64 us for a K = 1280 launch, 1.2 s of
// the 3.1-s strict pass, profiles/r05_c1_f32pass_kernels.csv).  Here a K chunk is 64 deep (16 NJ MFMAs per wave and barrier), a
// thread stages 16 consecutive k of one row (four 16-byte loads, two chunks ahead in registers) and writes them TRANSPOSED 4 x 4
// in registers -- LDS position 16 t + 4 g + e holds k = 16 t + 4 e + g -- so that the four steps 4 t .. 4 t + 3 of a lane's k-slot
// are ONE 16-byte LDS read; rows of 68 words keep those reads and the 16-byte writes conflict-free.
constexpr int R64_LD = 68;
template <int NJ>
__global__ __launch_bounds__(256) void gemm_f32_rows64(GemmArgs g)
{
    constexpr int BNT = 16 * NJ;
    __shared__ __attribute__((aligned(16))) float As[2][64][R64_LD];
    __shared__ __attribute__((aligned(16))) float Bs[2][BNT][R64_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * BNT;
    const float *A = (const float *)g.A;
    const float *W = (const float *)g.W;
    // staging: A tile: thread -> (row = tid >> 2, 16-k group = tid & 3), four 16-byte loads, written 4 x 4 transposed with 16-byte
    // LDS stores; B tile (16 NJ rows): thread -> (row = tid >> 4 [+ 16], k = 4 (tid & 15) .. + 3), ONE 16-byte load per 16 rows,
    // four scalar LDS stores -- every thread does the same work: no divergent branch around a load or its wait
    const int srow = tid >> 2, sgrp = tid & 3;
    const bool a_ok = m0 + srow < g.M;
    const float *ap = A + (size_t)(a_ok ? m0 + srow : g.M - 1) * g.lda + sgrp * 16;
    const int brow = tid >> 4, bc4 = tid & 15;
    bool b_ok[NJ];
    const float *wp[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        b_ok[j] = n0 + j * 16 + brow < g.N;
        wp[j] = W + (size_t)(b_ok[j] ? n0 + j * 16 + brow : g.N - 1) * g.ldw + bc4 * 4;
    }
    const int bpos = 16 * (bc4 >> 2) + (bc4 & 3);       // k = 16 t + 4 e + i (t = bc4 >> 2, e = bc4 & 3) -> position 16 t + 4 i + e
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};

    f32x4 acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[j] = zero4;
    f32x4 ra[2][4], rb[2][NJ];
#pragma unroll
    for (int st = 0; st < 2; ++st) {
#pragma unroll
        for (int e = 0; e < 4; ++e) ra[st][e] = zero4;
#pragma unroll
        for (int j = 0; j < NJ; ++j) rb[st][j] = zero4;
    }
    // Clamped addresses, never a predicated load; the zero-select of rows past M / N only when the registers go to LDS; and the
    // loads themselves in inline asm behind a hand-counted wait.  What hipcc made of ordinary loads here, seen in the ISA one form
    // after the other: a conditional load = a branch whose join waits vmcnt(0); a select right behind the load waits for it at
    // once; a prefetch under `if (c + 2 < KC)` makes the wait for chunk c + 1 valid for the path WITHOUT new loads, i.e. drains
    // them; and with everything unconditional the loop header still got a vmcnt(0) for the stage carried around the back edge.
    // Each form drained the two-chunk prefetch once per chunk (first hardware run: 22 us per K = 1280 launch).  Loads return in
    // issue order: chunk c + 1 has landed once only the NL loads of chunk c + 2 are pending.
    constexpr int NL = 4 + NJ;                     // loads per thread and chunk
    auto load_regs = [&](int st, int k0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ra[st][e]) : "v"(ap + k0 + 4 * e) : "memory");
#pragma unroll
        for (int j = 0; j < NJ; ++j) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rb[st][j]) : "v"(wp[j] + k0) : "memory");
    };
    auto wait_stage = [&](int st) {                // stage st is the OLDER of the two in flight
        if constexpr (NJ == 1)
            asm volatile("s_waitcnt vmcnt(%5)" : "+v"(ra[st][0]), "+v"(ra[st][1]), "+v"(ra[st][2]), "+v"(ra[st][3]), "+v"(rb[st][0]) : "n"(NL) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(%6)" : "+v"(ra[st][0]), "+v"(ra[st][1]), "+v"(ra[st][2]), "+v"(ra[st][3]), "+v"(rb[st][0]), "+v"(rb[st][NJ - 1]) : "n"(NL) : "memory");
    };
    auto store_lds = [&](int st, int buf) {
        // registers hold k = 16 sgrp + 4 e + i as ra[st][e][i]; LDS position 16 sgrp + 4 i + e
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 va = (f32x4){ra[st][0][i], ra[st][1][i], ra[st][2][i], ra[st][3][i]};
            *(f32x4 *)&As[buf][srow][16 * sgrp + 4 * i] = a_ok ? va : zero4;
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) Bs[buf][j * 16 + brow][bpos + 4 * i] = b_ok[j] ? rb[st][j][i] : 0.f;
    };
    auto compute = [&](int buf) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const f32x4 a4 = *(const f32x4 *)&As[buf][wave * 16 + fr][16 * t + 4 * fg];
            f32x4 b4[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) b4[j] = *(const f32x4 *)&Bs[buf][j * 16 + fr][16 * t + 4 * fg];
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[e], b4[j][e], acc[j], 0, 0, 0);
        }
    };

    const int KC = g.K / 64;
    load_regs(0, 0);
    load_regs(1, KC > 1 ? 64 : 0);
    wait_stage(0);
    store_lds(0, 0);
    __syncthreads();
    for (int c0 = 0; c0 < KC; c0 += 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {                       // (unrolled: the register stages are compile-time indices)
            const int c = c0 + u;
            if (c < KC) {
                // stage u went to LDS at the end of step c - 1.  Issued unconditionally (past the end: the last chunk again, never
                // stored): under a condition hipcc's wait for chunk c + 1 must hold on the path WITHOUT new loads and drains them
                load_regs(u, (c + 2 < KC ? c + 2 : KC - 1) * 64);
                compute(c & 1);
                if (c + 1 < KC) { wait_stage(u ^ 1); store_lds(u ^ 1, (c + 1) & 1); }
                __syncthreads();
            }
        }
    }
    // The idle re-loads of the last chunk are still in flight, and hipcc believes their destination registers dead (no store
    // follows): without the statements below it hands them out to the epilogue's address arithmetic while the loads land in them
    // (first hardware run of this form: wrong results and a memory fault).  Every staging register stays allocated up to the wait.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int st = 0; st < 2; ++st) {
#pragma unroll
        for (int e = 0; e < 4; ++e) asm volatile("" ::"v"(ra[st][e]));
#pragma unroll
        for (int j = 0; j < NJ; ++j) asm volatile("" ::"v"(rb[st][j]));
    }
    const int col_l = lane & 15, row_l = (lane >> 4) * 4;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) epilogue_store<float>(g, m0 + wave * 16 + row_l + r, n0 + j * 16 + col_l, acc[j][r]);
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
    constexpr int NKS_MAX = 10;      // k-steps (of 32) per wave: the launcher keeps K / 128 <= 10
    __shared__ f32x4 red[3][MT][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * 16;
    const f16 *A = (const f16 *)g.A;
    const f16 *W = (const f16 *)g.W;
    const int fr = lane & 15, fk = (lane >> 4) * 8;
    // K is split over the 4 waves
    const int kslice = g.K / 4;
    const int nks = kslice / 32;
    const int kb = wave * kslice;
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

}  // namespace

// Which f16 kernel a launch gets.  One pure function (no device state) so that the rule is testable without a GPU
// (tests/test_gemm_plan_cpu.py restates the benchmarked shapes) and is what swx_gemm executes.  `ptr16` = A, W (and C / R for
// the big kernel's vector epilogue) are 16-byte aligned; `force_kernel`: 0 dispatch, 1 register-staged, 2 skinny, 7 direct-to-LDS
// with occupancy overlap (8 / 9: its 64-column tiles always / never), 10 / 11 ring at 64 / 128 columns, 12 the 256 x 256 kernel.
int swx_gemm_plan_f16(int M, int N, int K, int epi, int64_t ldc, int64_t ldr, bool ptr16, int force_kernel, int flags)
{
    if (K % 32 != 0) return -4;                                            // tiled: K % 32, skinny: K % 128
    const bool skinny_ok = M <= 128 && K % 128 == 0 && K / 128 <= 10 && N <= 16384;   // vocabulary-sized N: tiled
    if (force_kernel == 2) return skinny_ok ? SWX_GEMM_SKINNY : -4;
    if (skinny_ok && force_kernel != 1 && force_kernel < 7) return SWX_GEMM_SKINNY;
    const bool glds_ok = K % 64 == 0 && ptr16;
    if (force_kernel >= 7 && !glds_ok) return -4;
    if (force_kernel == 1 || !glds_ok) return SWX_GEMM_TILED_REG;          // K % 64 != 0, and the bit-identity reference
    const int64_t t128 = (int64_t)cdiv(N, BN) * cdiv(M, BM);
    // 64-column tiles when 128-wide ones would leave CUs idle (encoder at one window: M = 1500, N = 1280 is 120 tiles of
    // 128 x 128 for 256 CUs)
    const bool narrow = force_kernel == 8 || (force_kernel != 9 && N % 64 == 0 && t128 < 224);
    // the 256 x 256 kernel: plain epilogues, when its tiles fill whole rounds of the 256 CUs -- one workgroup per CU, so a last
    // round that is 30 % full costs a full round (M = 6000, N = 3840: 360 tiles, 769 against 885 TFLOP/s) -- or nearly so with
    // a long K to amortise prologue and epilogue (M = 30 000, N = 1280, K = 5120: 590 tiles, 979 against 892); not for K < 512
    // (profiles/r03_kb_gemm_big.txt)
    const bool big_ok = !(epi & ~(EPI_BIAS | EPI_GELU | EPI_RES)) && ldc % 8 == 0 && (!(epi & EPI_RES) || ldr % 8 == 0);
    if (force_kernel == 12) return big_ok && K >= 128 ? SWX_GEMM_BIG : -4;      // (its prologue keeps two K tiles in flight)
    const int64_t t256 = (int64_t)cdiv(M, BG) * cdiv(N, BG);
    const double fill = (double)t256 / (double)(((t256 + 255) / 256) * 256);
    if (force_kernel == 0 && big_ok && t256 >= 200 && K >= 512 && (fill >= 0.9 || (fill >= 0.75 && K >= 2560)) &&
        !(flags & SWX_FLAG_NO_BIG_TILE))
        return SWX_GEMM_BIG;
    // the ring kernel for launches of at most one workgroup per CU (the encoder / cross-K/V at one window: 64-column tiles when
    // `narrow`, else 128-column ones up to 256 tiles); from ~1.5 workgroups per CU on, gemm_f16_glds -- three resident
    // workgroups, epilogues overlapped with the neighbours' MFMAs -- is as fast or faster (N = 3840 / 5120 at M = 1500: 29.3 /
    // 29.6 us against 33.4 / 35.9; profiles/r03_kb_gemm_ring.txt).  Not the one-row-tile logits GEMM (133 MB of weights).
    const bool ring_ok = K >= 128;
    if (force_kernel == 10 || force_kernel == 11) return !ring_ok ? -4 : force_kernel == 10 ? SWX_GEMM_RING64 : SWX_GEMM_RING128;
    if (force_kernel == 0 && ring_ok && M > 256 && (narrow || t128 <= 256) && !(flags & SWX_FLAG_NO_RING))
        return narrow ? SWX_GEMM_RING64 : SWX_GEMM_RING128;
    return narrow ? SWX_GEMM_GLDS64 : SWX_GEMM_GLDS128;
}

int swx_gemm(int dtype, const GemmArgs &g, int force_kernel, hipStream_t s)
{
    if (g.M <= 0 || g.N <= 0) return 0;
    if (dtype == SWX_F16) {
        if (g.lda % 8 != 0 || g.ldw % 8 != 0) return -4;
        const bool ptr16 = (uintptr_t)g.A % 16 == 0 && (uintptr_t)g.W % 16 == 0;
        // (C / R alignment only matters to the big kernel's 16-byte epilogue: withheld from it by an unaligned leading dimension)
        const bool cr16 = (uintptr_t)g.C % 16 == 0 && (!(g.epi & EPI_RES) || (uintptr_t)g.R % 16 == 0);
        const int plan = swx_gemm_plan_f16(g.M, g.N, g.K, g.epi, cr16 ? g.ldc : 1, cr16 ? g.ldr : 1, ptr16, force_kernel, swx_flags());
        if (plan < 0) return plan;
        if (plan == SWX_GEMM_SKINNY) {
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
            const dim3 grid(cdiv(g.N, BN), cdiv(g.M, BM));
            switch (plan) {
                case SWX_GEMM_BIG: {
                    GemmArgs gb = g;
                    gb.tile_order = (swx_flags() & SWX_FLAG_BIG8_GROUPED) ? 1 : 0;
                    hipLaunchKernelGGL(gemm_f16_big8, dim3(cdiv(g.N, BG), cdiv(g.M, BG)), dim3(512), 2 * B8_BUF, s, gb); break;
                }
                case SWX_GEMM_RING64: launch_ring<64, 3>(g, s); break;
                case SWX_GEMM_RING128: launch_ring<128, 3>(g, s); break;
                case SWX_GEMM_GLDS64: hipLaunchKernelGGL(gemm_f16_glds_64, dim3(cdiv(g.N, 64), grid.y), dim3(256), 0, s, g); break;
                case SWX_GEMM_GLDS128: hipLaunchKernelGGL(gemm_f16_glds_128, grid, dim3(256), 0, s, g); break;
                default: hipLaunchKernelGGL(gemm_f16_tiled, grid, dim3(256), 0, s, g); break;
            }
        }
    } else {
        if (g.K % 16 != 0 || g.lda % 4 != 0 || g.ldw % 4 != 0) return -4;
        SwxProfScope prof(PC_GEMM_TILED, 2.0 * (double)g.M * g.N * g.K, s);
        // few rows (decode step, prefill, a single window's scoring pass): 64 x 32 or 64 x 16 tiles spread the launch over
        // 160-320 workgroups instead of N / 128 = 10-40 (one CU's exact-f32 MFMA rate is 0.6 TFLOP/s); bit-identical either way,
        // so the choice may depend on the launch
        if (g.M <= BM && g.N >= 256 && g.K % 64 == 0 && force_kernel != 8) {
            // (round 5) the same tiles on 64-deep K chunks; force_kernel 8 keeps the 16-deep generation as its bit-identity reference
            if (g.N >= 2560) hipLaunchKernelGGL((gemm_f32_rows64<2>), dim3(cdiv(g.N, 32), cdiv(g.M, 64)), dim3(256), 0, s, g);
            else hipLaunchKernelGGL((gemm_f32_rows64<1>), dim3(cdiv(g.N, 16), cdiv(g.M, 64)), dim3(256), 0, s, g);
        } else if (g.M <= BM && g.N >= 256) {
            if (g.N >= 2560) hipLaunchKernelGGL((gemm_f32_tiled<64, 32, 4>), dim3(cdiv(g.N, 32), cdiv(g.M, 64)), dim3(256), 0, s, g);
            else hipLaunchKernelGGL((gemm_f32_tiled<64, 16, 4>), dim3(cdiv(g.N, 16), cdiv(g.M, 64)), dim3(256), 0, s, g);
        } else {
            hipLaunchKernelGGL((gemm_f32_tiled<128, 128, 4>), dim3(cdiv(g.N, BN), cdiv(g.M, BM)), dim3(256), 0, s, g);
        }
    }
    SWX_CHECK_LAUNCH();
    return 0;
}
