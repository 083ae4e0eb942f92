// swx_attn.hip -- attention kernels (d_head = 64 for every Whisper size).
//
// Upstream: whisper/model.py::MultiHeadAttention.qkv_attention (q,k each scaled by d_head^-0.25, fp32 softmax),
// reached from stable_whisper/decode.py:40 (decoder steps), decode.py:27-30 (encoder) and timing.py:58-61 (the
// teacher-forced pass whose cross-attention qk the word-timestamp code reads).
//
//   attn_flash2_f16     MFMA flash attention, non-causal, for the MFMA-bound cases (encoder self-attention,
//                       1500x1500 per head; cross-attention of the scoring pass).  One wave = 32 queries,
//                       K tile [64 keys][64] and V^T tile [64][64 keys] double-buffered in LDS per 4-wave workgroup.
//   attn_decode_cross   decode-step cross-attention: K / V^T streamed straight into MFMA operands (HBM-bound).
//   attn_dense_rowwise  VALU kernel (any dtype): 8 queries that share one K/V (same window, same head) per workgroup;
//                       K and V are streamed once per workgroup.  Used for the HBM-bound decode-step cross-attention
//                       (the G beams of a window share the 246 MB/window cross-KV read) and for everything in the
//                       strict f32 mode.
//   self_attn_cached    decoder self-attention over the KV cache; beams address the cache through an ancestor table
//                       (position -> physical row) so that a beam reorder never copies K/V.
//   qk_capture          raw scaled q.k of the alignment heads only (what timing.py:50-56 hooks out of every layer).
#include <type_traits>
#include "swx_common.h"
#include "swx_kernels.h"

namespace {


constexpr int DH = 64;

// ================================================================================================ flash f16
constexpr int FL_KT = 64;            // keys per tile
constexpr int FL_LD = 72;            // halfs per LDS row (144 B: keeps b128 / b64 fragment reads aligned)

// ================================================================================================ flash f16 (generation 2)
// Swapped QK^T (S^T = K.Q^T), so a lane's accumulator registers all belong to ONE query: the online-softmax row statistics are
// lane-local + two shuffles and the exponentiated tile is already laid out as the B operand of O^T += V^T.P^T (no LDS round
// trip for P).  Against the first kernel (16 queries per wave, single-buffered tiles: 0.10 of the MFMA peak, deleted in round 3):
//   * a wave owns QB 16-query blocks (QB = 2: 32 queries; QB = 4: 64 queries for long query axes): every K / V^T fragment read
//     from LDS feeds QB MFMAs -- 24 fragment reads per 16 QB MFMAs per wave and tile -- and a workgroup covers 64 QB queries per
//     barrier;
//   * K / V^T tiles are double-buffered in LDS, the global loads of tile t+1 are issued into registers BEFORE the MFMAs of
//     tile t and written to the other buffer after them: one barrier per tile, no exposed load latency;
//   * with VT (V already transposed per head in HBM: cross-KV layout, and the encoder's V through swx_transpose_v) both
//     tiles are written with 16-byte vector stores; the row-major-V form (scalar transposing stores) is kept for the callers
//     that have no transposed copy.
template <bool VT, int QB>
__global__ __launch_bounds__(256, 2) void attn_flash2_f16(AttnArgs a)
{
    __shared__ __attribute__((aligned(16))) f16 Ks[2][FL_KT][FL_LD];   // [buf][key][d]
    __shared__ __attribute__((aligned(16))) f16 Vt[2][DH][FL_LD];      // [buf][d][key]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.z, h = blockIdx.y;
    const int q0 = blockIdx.x * (QB * 64) + wave * (QB * 16);
    const int qn = lane & 15, g = lane >> 4;
    const f16 *Q = (const f16 *)a.q;
    const f16 *K = (const f16 *)a.k + (size_t)b * a.k_bs + h * DH;
    const f16 *V = (const f16 *)a.v + (size_t)b * a.v_bs + (VT ? (size_t)h * DH * a.vt_kp : (size_t)h * DH);

    f16x8 qf[QB][2];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int qi = q0 + qb * 16 + qn;
        const f16 *qp = Q + ((size_t)b * a.q_rows_per_batch + (qi < a.nq ? qi : a.nq - 1)) * a.ldq + h * DH + g * 8;   // clamped
        qf[qb][0] = *(const f16x8 *)(qp);
        qf[qb][1] = *(const f16x8 *)(qp + 32);
    }
    f32x4 o[QB][4];
    float m_run[QB], l_run[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        m_run[qb] = -__builtin_inff(); l_run[qb] = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) o[qb][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

    // staging registers of one tile: chunk c = tid + 256 i  ->  K row c>>3, dims (c&7)*8 ; V^T row c>>3, keys (c&7)*8
    f16x8 rk[2], rv[2];
    auto load_tile = [&](int kt0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = tid + 256 * i, r = c >> 3, c8 = (c & 7) * 8;
            const int kg = kt0 + r;
            const int kgc = kg < a.nk ? kg : a.nk - 1;                      // clamped, never predicated
            rk[i] = *(const f16x8 *)(K + (size_t)kgc * a.ldkv + c8);
            if (kg >= a.nk) rk[i] = (f16x8)(f16)0;
            if constexpr (VT) {
                rv[i] = *(const f16x8 *)(V + (size_t)r * a.vt_kp + kt0 + c8);           // zero padded past nk in the source
            } else {
                rv[i] = *(const f16x8 *)(V + (size_t)kgc * a.ldkv + c8);
                if (kg >= a.nk) rv[i] = (f16x8)(f16)0;
            }
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = tid + 256 * i, r = c >> 3, c8 = (c & 7) * 8;
            *(f16x8 *)&Ks[buf][r][c8] = rk[i];
            if constexpr (VT) {
                *(f16x8 *)&Vt[buf][r][c8] = rv[i];
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) Vt[buf][c8 + e][r] = rv[i][e];
            }
        }
    };

    const int ntile = (a.nk + FL_KT - 1) / FL_KT;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int t = 0; t < ntile; ++t) {
        const int cur = t & 1, kt0 = t * FL_KT;
        if (t + 1 < ntile) load_tile(kt0 + FL_KT);
        // K fragments of this tile, shared by the QB query blocks
        f16x8 kf[4][2];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) kf[tt][kk] = *(const f16x8 *)&Ks[cur][tt * 16 + qn][kk * 32 + g * 8];
        f16x8 pb[QB][2];
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            f32x4 sc[4];
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                sc[tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                sc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[tt][0], qf[qb][0], sc[tt], 0, 0, 0);
                sc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[tt][1], qf[qb][1], sc[tt], 0, 0, 0);
            }
            // softmax bookkeeping on the RAW scores (the 1/8 scale is positive: max commutes with it) and in the exp2 domain:
            // per element one max, one fma, one v_exp_f32, one add -- the key-range mask only on the last tile.  This loop,
            // not the MFMAs, bounds the kernel (32 MFMAs = 512 cycles against ~350 VALU operations = 1400 cycles per 64-key
            // tile and wave before; ~200 now).
            constexpr float SC2 = 0.125f * 1.4426950408889634f;
            if (kt0 + FL_KT > a.nk) {
#pragma unroll
                for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (kt0 + tt * 16 + g * 4 + r >= a.nk) sc[tt][r] = -__builtin_inff();
            }
            float tmax = -__builtin_inff();
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                for (int r = 0; r < 4; ++r) tmax = fmaxf(tmax, sc[tt][r]);
            tmax = lane_xor16_max(tmax);                // (VALU row swaps, not ds_bpermute: swx_common.h)
            tmax = lane_xor32_max(tmax);
            const float m_new = fmaxf(m_run[qb], tmax);                               // raw units
            const float alpha = __builtin_amdgcn_exp2f((m_run[qb] - m_new) * SC2);    // m_run = -inf on the first tile -> 0
            const float mc = m_new * SC2;
            float psum = 0.f;
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[tt][r], SC2, -mc));
                    sc[tt][r] = p;
                    psum += p;
                }
            psum = lane_xor16_add(psum);
            psum = lane_xor32_add(psum);
            l_run[qb] = l_run[qb] * alpha + psum;
            m_run[qb] = m_new;
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) { o[qb][tt][0] *= alpha; o[qb][tt][1] *= alpha; o[qb][tt][2] *= alpha; o[qb][tt][3] *= alpha; }
            // P^T as the B operand: k-slot (g, j) of k-block c <-> key 32c + (j<4 ? g*4+j : 16 + g*4 + j-4)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) { pb[qb][c][r] = (f16)sc[2 * c][r]; pb[qb][c][4 + r] = (f16)sc[2 * c + 1][r]; }
        }
        // O^T += V^T . P^T : every V^T fragment feeds all QB query blocks
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                const f16x4 lo = *(const f16x4 *)&Vt[cur][tt * 16 + qn][32 * c + g * 4];
                const f16x4 hi = *(const f16x4 *)&Vt[cur][tt * 16 + qn][32 * c + 16 + g * 4];
                f16x8 va;
#pragma unroll
                for (int e = 0; e < 4; ++e) { va[e] = lo[e]; va[4 + e] = hi[e]; }
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) o[qb][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(va, pb[qb][c], o[qb][tt], 0, 0, 0);
            }
        if (t + 1 < ntile) store_tile(cur ^ 1);
        __syncthreads();
    }

#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int qi = q0 + qb * 16 + qn;
        if (qi < a.nq) {
            const float inv = 1.0f / l_run[qb];
            f16 *op = (f16 *)a.o + ((size_t)b * a.q_rows_per_batch + qi) * a.ldo + h * DH;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                f16x4 ov;
#pragma unroll
                for (int r = 0; r < 4; ++r) ov[r] = (f16)(o[qb][t][r] * inv);
                *(f16x4 *)(op + t * 16 + g * 4) = ov;
            }
        }
    }
}

// ================================================================================================ flash f16 (generation 3, round 6)
// MEASURED SLOWER, kept behind SWX_FLAG_FLASH_PIPELINED as the record of the experiment.
// The arithmetic of attn_flash2_f16<true, QB> operation for operation (bit-identical: tests/test_gpu_kernels.py::
// test_flash3_is_bit_identical_to_flash2), re-staged so that the matrix pipe and the VALU of ONE wave could overlap.  Generation 2's
// key tile is  for qb: { 8 QK^T MFMAs -> softmax of that block (~100 VALU + 17 v_exp on the MFMA results) }  then 32 PV MFMAs, and
// hipcc keeps that order: every wave alternates between MFMA-only and VALU-only stretches (0.25 MFMA utilisation by counters at two
// waves per SIMD).  Here the tile is ONE basic block (the key-range mask lives in a peeled instantiation for the last tile, the
// next tile's loads / stores are unconditional) laid out as a software pipeline over the query blocks:
//     QK(0) | softmax(0) + QK(1) | softmax(1) + QK(2) + PV(0) | softmax(2) + QK(3) + PV(1) | softmax(3) + PV(2) | PV(3)
// with sched_group_barrier pipelines that put one MFMA in front of every few VALU instructions of a softmax (seen in the ISA).
// Result on hardware (profiles/r06_c14_flash3_ab.json): 391-406 us against 376-387 us per encoder layer at 20 windows, 42.9 against
// 37.4 us at one window; headline pass 428.6 vs 427.5 ms, align() 916 vs 902 ms.  What MI355X_MICROARCH.md says of two waves per
// SIMD holds: VALU placed beside a wave's own dependency-paced MFMAs delays them, and the second wave of the SIMD already fills
// generation 2's MFMA-only and VALU-only stretches -- moving work between the two is zero- or negative-sum.
template <int QB>
__global__ __launch_bounds__(256, 2) void attn_flash3_f16(AttnArgs a)
{
    __shared__ __attribute__((aligned(16))) f16 Ks[2][FL_KT][FL_LD];   // [buf][key][d]
    __shared__ __attribute__((aligned(16))) f16 Vt[2][DH][FL_LD];      // [buf][d][key]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.z, h = blockIdx.y;
    const int q0 = blockIdx.x * (QB * 64) + wave * (QB * 16);
    const int qn = lane & 15, g = lane >> 4;
    const f16 *Q = (const f16 *)a.q;
    const f16 *K = (const f16 *)a.k + (size_t)b * a.k_bs + h * DH;
    const f16 *V = (const f16 *)a.v + (size_t)b * a.v_bs + (size_t)h * DH * a.vt_kp;

    f16x8 qf[QB][2];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int qi = q0 + qb * 16 + qn;
        const f16 *qp = Q + ((size_t)b * a.q_rows_per_batch + (qi < a.nq ? qi : a.nq - 1)) * a.ldq + h * DH + g * 8;   // clamped
        qf[qb][0] = *(const f16x8 *)(qp);
        qf[qb][1] = *(const f16x8 *)(qp + 32);
    }
    f32x4 o[QB][4];
    float m_run[QB], l_run[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        m_run[qb] = -__builtin_inff(); l_run[qb] = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) o[qb][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const int ntile = (a.nk + FL_KT - 1) / FL_KT;
    f16x8 rk[2], rv[2];
    auto load_tile = [&](int kt0) {           // (past the last tile: the last tile again -- never predicated, never stored to a live buffer)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = tid + 256 * i, r = c >> 3, c8 = (c & 7) * 8;
            const int kg = kt0 + r;
            const int kgc = kg < a.nk ? kg : a.nk - 1;
            rk[i] = *(const f16x8 *)(K + (size_t)kgc * a.ldkv + c8);
            if (kg >= a.nk) rk[i] = (f16x8)(f16)0;
            rv[i] = *(const f16x8 *)(V + (size_t)r * a.vt_kp + kt0 + c8);               // zero padded past nk in the source
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = tid + 256 * i, r = c >> 3, c8 = (c & 7) * 8;
            *(f16x8 *)&Ks[buf][r][c8] = rk[i];
            *(f16x8 *)&Vt[buf][r][c8] = rv[i];
        }
    };
    constexpr float SC2 = 0.125f * 1.4426950408889634f;

    auto tile = [&](auto mask_tag, int cur, int kt0) {
        constexpr bool MASK = decltype(mask_tag)::value;
        f16x8 kf[4][2], va[2][4];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) kf[tt][kk] = *(const f16x8 *)&Ks[cur][tt * 16 + qn][kk * 32 + g * 8];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                const f16x4 lo = *(const f16x4 *)&Vt[cur][tt * 16 + qn][32 * c + g * 4];
                const f16x4 hi = *(const f16x4 *)&Vt[cur][tt * 16 + qn][32 * c + 16 + g * 4];
#pragma unroll
                for (int e = 0; e < 4; ++e) { va[c][tt][e] = lo[e]; va[c][tt][4 + e] = hi[e]; }
            }
        f32x4 sc[2][4];
        f16x8 pb[2][2];
        auto qk = [&](int qb, f32x4 (&s4)[4]) {
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                s4[tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                s4[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[tt][0], qf[qb][0], s4[tt], 0, 0, 0);
                s4[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[tt][1], qf[qb][1], s4[tt], 0, 0, 0);
            }
        };
        auto pv = [&](int qb, const f16x8 (&p2)[2]) {
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) o[qb][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(va[c][tt], p2[c], o[qb][tt], 0, 0, 0);
        };
        auto softmax = [&](int qb, f32x4 (&s4)[4], f16x8 (&p2)[2]) {
            if constexpr (MASK) {
#pragma unroll
                for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (kt0 + tt * 16 + g * 4 + r >= a.nk) s4[tt][r] = -__builtin_inff();
            }
            float tmax = -__builtin_inff();
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                for (int r = 0; r < 4; ++r) tmax = fmaxf(tmax, s4[tt][r]);
            tmax = lane_xor16_max(tmax);
            tmax = lane_xor32_max(tmax);
            const float m_new = fmaxf(m_run[qb], tmax);
            const float alpha = __builtin_amdgcn_exp2f((m_run[qb] - m_new) * SC2);
            const float mc = m_new * SC2;
            float psum = 0.f;
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s4[tt][r], SC2, -mc));
                    s4[tt][r] = p;
                    psum += p;
                }
            psum = lane_xor16_add(psum);
            psum = lane_xor32_add(psum);
            l_run[qb] = l_run[qb] * alpha + psum;
            m_run[qb] = m_new;
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) { o[qb][tt][0] *= alpha; o[qb][tt][1] *= alpha; o[qb][tt][2] *= alpha; o[qb][tt][3] *= alpha; }
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) { p2[c][r] = (f16)s4[2 * c][r]; p2[c][4 + r] = (f16)s4[2 * c + 1][r]; }
        };
        qk(0, sc[0]);
        auto stage = [&](auto qb_tag) {
            constexpr int qb = decltype(qb_tag)::value;
            if constexpr (qb + 1 < QB) qk(qb + 1, sc[(qb + 1) & 1]);
            if constexpr (qb >= 1) pv(qb - 1, pb[(qb - 1) & 1]);
            softmax(qb, sc[qb & 1], pb[qb & 1]);
            // one MFMA in front of every few VALU instructions of this stage's softmax (8 or 16 MFMAs per stage, ~115 VALU)
            constexpr int NM = (qb + 1 < QB ? 8 : 0) + (qb >= 1 ? 8 : 0);
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, NM == 16 ? 7 : 14, 0);
            }
        };
        stage(std::integral_constant<int, 0>{});
        if constexpr (QB > 1) stage(std::integral_constant<int, 1>{});
        if constexpr (QB > 2) stage(std::integral_constant<int, 2>{});
        if constexpr (QB > 3) stage(std::integral_constant<int, 3>{});
        static_assert(QB <= 4, "stages are listed by hand");
        pv(QB - 1, pb[(QB - 1) & 1]);
    };

    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int t = 0; t + 1 < ntile; ++t) {
        load_tile((t + 1) * FL_KT);
        tile(std::false_type{}, t & 1, t * FL_KT);
        store_tile((t & 1) ^ 1);
        __syncthreads();
    }
    if (ntile * FL_KT > a.nk) tile(std::true_type{}, (ntile - 1) & 1, (ntile - 1) * FL_KT);
    else tile(std::false_type{}, (ntile - 1) & 1, (ntile - 1) * FL_KT);

#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int qi = q0 + qb * 16 + qn;
        if (qi < a.nq) {
            const float inv = 1.0f / l_run[qb];
            f16 *op = (f16 *)a.o + ((size_t)b * a.q_rows_per_batch + qi) * a.ldo + h * DH;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                f16x4 ov;
#pragma unroll
                for (int r = 0; r < 4; ++r) ov[r] = (f16)(o[qb][t][r] * inv);
                *(f16x4 *)(op + t * 16 + g * 4) = ov;
            }
        }
    }
}

// V [B][n][ldv] (head h at column h*64) -> V^T [B][H][64][kp] per head, keys contiguous, zero padded up to kp: LDS tile
// transpose, 16-byte accesses on both sides.  One workgroup per (64-key tile, head, batch item).
__global__ __launch_bounds__(256) void transpose_v_kernel(const f16 *__restrict__ V, int64_t ldv, int64_t v_bs, int n, f16 *__restrict__ VT,
                                                          int kp, int64_t vt_bs)
{
    __shared__ __attribute__((aligned(16))) f16 T[DH][FL_LD];      // [d][key]
    const int kt0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = tid + 256 * i, key = c >> 3, c8 = (c & 7) * 8;
        const int kg = kt0 + key;
        f16x8 v = *(const f16x8 *)(V + (size_t)b * v_bs + (size_t)(kg < n ? kg : n - 1) * ldv + h * DH + c8);
        if (kg >= n) v = (f16x8)(f16)0;
#pragma unroll
        for (int e = 0; e < 8; ++e) T[c8 + e][key] = v[e];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = tid + 256 * i, dr = c >> 3, k8 = (c & 7) * 8;
        if (kt0 + k8 < kp) *(f16x8 *)(VT + (size_t)b * vt_bs + ((size_t)h * DH + dr) * kp + kt0 + k8) = *(const f16x8 *)&T[dr][k8];
    }
}

// ============================================================================================ decode cross
// Decode-step cross-attention: <= 16 queries (the G beams of one window) against the window's 1500 encoder positions.
// HBM-bound: the only traffic that matters is ONE pass over this (window, head)'s K [1500][64] and V^T [64][kp]
// (384 KB in fp16), so nothing is staged in LDS -- every MFMA fragment is a coalesced 16-byte global load:
//   S^T[16 keys][16 q] = K-frag . Q^T        (K rows gathered so that a lane ends up holding 8 CONSECUTIVE keys)
//   O^T[64 d][16 q]   += V^T-frag . P^T      (V^T rows are key-contiguous: 16-byte loads again)
// 4 waves split the keys (32-key blocks, round-robin) with a private online softmax each and merge through LDS once.
// PACKED: K and V^T come from the fragment-ordered copy (swx_xkv_pack): every MFMA operand fragment of a 32-key block is one
// contiguous 1 KB piece, so a wave instruction reads 8 full 128-byte lines instead of 16 separate 64-byte row pieces (the
// per-CU address path, not HBM, bounds the row-layout variant at ~4.2 TB/s: same finding as for the decode GEMM weights).
// QG: groups of 16 query rows one workgroup carries through a single pass over the head's K / V^T (1 for a decode step; up to 4
// for a multi-token pass over many windows, where one workgroup per group would re-stream the head's 384 KB once per group --
// 8 times for a ~113-row scoring pass: measured +14 ms per 20-window pass).  Every group's arithmetic is the one-group
// kernel's (same key order per wave, same merge), so the result does not depend on QG.
// FQ > 0 (decode step, PACKED, QG = 1): FQ = d / 32 k-steps of the cross-attention QUERY projection run inside this launch (AttnArgs::fq_*).
// The workgroup of (window b, head h) DMAs the 16-row tile of the raw residual stream that starts at the window's first row into
// LDS, its four waves take the head's four 16-column panels of the packed LayerNorm-folded weights (all FQ fragments of a wave in
// flight at once, like gemm_dec_f16), compute the rows' LayerNorm statistics from the tile, run the FQ MFMAs and leave
// f16(rstd (acc - mean c1) + c2) in a 2 KB LDS tile the attention part reads its query fragments from -- the statistics, the
// k-step order and the epilogue expression of swx_decstep.hip::gemm_dec_f16<1, FQ, DEC_LN>, so q is bit-identical to the separate
// launch (tests: test_decode_f16_fused_cross_query_is_bit_identical).  The first K / V^T block is requested before any of it.
template <bool PACKED, int QG, int FQ, int XV = 0>     // XV: 0 one key block per wave in flight (rounds 2-5); 1: two, with nt loads (round 6)
__device__ __forceinline__ void attn_decode_cross_body(const AttnArgs &a)
{
    constexpr bool NT = XV == 1 && PACKED;
    constexpr int DEPTH = XV == 0 ? 1 : 2;
    __shared__ float sm_m[4][16], sm_l[4][16];
    __shared__ float sm_o[4][DH][17];
    extern __shared__ __attribute__((aligned(1024))) unsigned char fq_smem[];   // FQ: [16][FQ * 32] f16 | float2 stat[16] | f16 q[16][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.x, b = blockIdx.y;
    // blockIdx.z = QG groups of 16 query rows of this (window, head): a teacher-forced pass of ~100 rows over ONE window has 20
    // flash workgroups to offer 256 CUs; as groups of 16 rows on this kernel it has 100-160, each streaming the head's K / V^T
    // (384 KB, L2-resident after the first group) with its 4 waves splitting the keys
    const int q_base0 = blockIdx.z * 16 * QG;
    const int qn = lane & 15, g = lane >> 4;
    const f16 *Q = (const f16 *)a.q;
    const f16 *Kp = (const f16 *)a.k + (size_t)b * a.k_bs + h * DH;
    const f16 *Vp = (const f16 *)a.v + (size_t)b * a.v_bs + (size_t)h * DH * a.vt_kp;

    f32x4 o[QG][4];
    float m_run[QG], l_run[QG];
#pragma unroll
    for (int u = 0; u < QG; ++u) {
#pragma unroll
        for (int t = 0; t < 4; ++t) o[u][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        m_run[u] = -__builtin_inff(); l_run[u] = 0.f;
    }
    const int nblk = (a.nk + 31) >> 5;
    // A-row i of S^T tile t  <->  key k0 + (i>>2)*8 + (i&3) + 4t, so that lane (q, g) owns keys k0 + g*8 + 0..7
    const int krow = (qn >> 2) * 8 + (qn & 3);

    const int64_t per_head = swx_xkv_packed_elems_per_head(a.nk);
    const f16 *Kpk = PACKED ? (const f16 *)a.kv_packed + (size_t)b * a.k_bs + (size_t)h * per_head + lane * 8 : nullptr;
    const f16 *Vpk = PACKED ? Kpk + per_head / 2 : nullptr;
    auto load_blk = [&](int cb, f16x8 (&kf)[4], f16x8 (&vf)[4]) {
        if constexpr (PACKED) {
            const int blk = cb < nblk ? cb : nblk - 1;
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                const f16x8 *kp_ = (const f16x8 *)(Kpk + ((size_t)blk * 4 + f) * 512);
                kf[f] = NT ? __builtin_nontemporal_load(kp_) : *kp_;
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const f16x8 *vp_ = (const f16x8 *)(Vpk + ((size_t)blk * 4 + t) * 512);
                vf[t] = NT ? __builtin_nontemporal_load(vp_) : *vp_;
            }
            return;
        }
        const int k0 = (cb < nblk ? cb : nblk - 1) << 5;    // tail prefetch re-reads the last block: loads are never predicated
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int key = k0 + krow + 4 * t;
            const f16 *kr = Kp + (size_t)(key < a.nk ? key : a.nk - 1) * a.ldkv + g * 8;   // rows past nk are masked to -inf below
            kf[2 * t] = *(const f16x8 *)(kr);
            kf[2 * t + 1] = *(const f16x8 *)(kr + 32);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) vf[t] = *(const f16x8 *)(Vp + (size_t)(t * 16 + qn) * a.vt_kp + k0 + g * 8);
    };
    // the first key block of this wave is requested before q is assembled, so its latency overlaps the q loads
    f16x8 kA[4], vA[4];
    load_blk(wave, kA, vA);

    f16x8 qf[QG][2];
    if constexpr (FQ > 0) {
        static_assert(QG == 1 && FQ % 4 == 0, "fused query projection: one group of <= 16 rows");
        typedef __attribute__((address_space(3))) void lds_void;
        typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
        typedef _Float16 f16x4v __attribute__((ext_vector_type(4)));
        constexpr int kslice = FQ * 32, SPR = kslice >> 3, RS = kslice * 2;
        const int li = lane & 15, lg = lane >> 4;
        const int row0 = b * a.q_rows_per_batch;
        // residual tile -> LDS: only the instructions that hold one of the window's nq rows (5 beams = 13 of the 16-row tile's FQ = 40; round 6:
        // the tile was 40 KB of the ~590 KB a workgroup pulls through its CU).  Rows past nq stay whatever LDS held: their projections are
        // never read (qok below).  A run-time loop: no instruction under a branch of its own; a.fq_full_tile (A/B) stages all 16 rows
        {
            const int n_need = a.fq_full_tile ? FQ : (a.nq * SPR + 63) >> 6;
            for (int qi = wave; qi < n_need; qi += 4) {
                const int p = qi * 64 + lane;
                const int row = p / SPR, ps = p - row * SPR;
                const int kslot = ps ^ (row & 15);
                const int gr = row0 + row < a.fq_rows ? row0 + row : a.fq_rows - 1;
                __builtin_amdgcn_global_load_lds(a.fq_x + (size_t)gr * a.fq_ldx + kslot * 8, (lds_void *)(fq_smem + qi * 1024), 16, 0, 0);
            }
        }
        const f16 *wp = a.fq_w + ((size_t)(h * 4 + wave) * FQ) * 512 + lane * 8;
        f16x8 wf[FQ];
#pragma unroll
        for (int ks = 0; ks < FQ; ++ks) wf[ks] = *(const f16x8 *)(wp + (size_t)ks * 512);
        const int ncol = h * DH + wave * 16 + lg * 4;
        const f32x4 c2 = *(const f32x4 *)(a.fq_c2 + ncol), c1 = *(const f32x4 *)(a.fq_c1 + ncol);
        unsigned pfv = 0;
        if (a.fq_pf) {      // cache prefetch of the next projection's weights: one 128-byte line per thread, result never read
            int line = ((b * a.H + h) * 4 + wave) * 64 + lane;
            line = line < a.fq_pf_lines ? line : a.fq_pf_lines - 1;
            pfv = *(const volatile unsigned *)((const unsigned char *)a.fq_pf + (size_t)line * 128);
        }
        __syncthreads();                                    // tile landed (hipcc drains every load of the wave in front of it)
        float2 *stat = (float2 *)(fq_smem + 16 * RS);
        f16 *qt = (f16 *)(fq_smem + 16 * RS + 16 * sizeof(float2));
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        const unsigned char *abase = fq_smem + (size_t)li * RS;
        {
            constexpr int PF = 8;                           // fragment reads eight k-steps ahead of their MFMA (one MFMA per step)
            f16x8 af[PF];
#pragma unroll
            for (int ks = 0; ks < PF; ++ks) af[ks] = *(const f16x8 *)(abase + (((ks * 4 + lg) ^ li) << 4));
#pragma unroll
            for (int ks = 0; ks < FQ; ++ks) {
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ks], af[ks % PF], acc, 0, 0, 0);
                if (ks + PF < FQ) af[ks % PF] = *(const f16x8 *)(abase + ((((ks + PF) * 4 + lg) ^ li) << 4));
                __builtin_amdgcn_sched_barrier(0);          // (the scheduler sinks each read back in front of its MFMA otherwise)
            }
        }
        // (the statistics after the MFMAs: their row registers do not have to live next to the 40 weight fragments)
        {
            const f16x2 one2 = {(f16)1.f, (f16)1.f};
            const int rb = wave * 4 + lg;
            const unsigned char *rp = fq_smem + (size_t)rb * RS + li * 16;
            f16x8 v[SPR / 16];
#pragma unroll
            for (int i = 0; i < SPR / 16; ++i) v[i] = *(const f16x8 *)(rp + i * 256);
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < SPR / 16; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const f16x2 pr = {v[i][2 * e], v[i][2 * e + 1]};
                    s1 = __builtin_amdgcn_fdot2(pr, one2, s1, false);
                    s2 = __builtin_amdgcn_fdot2(pr, pr, s2, false);
                }
#pragma unroll
            for (int o2 = 0; o2 < 1; ++o2) {     // 1, 2, 4, 8 in this order (DPP exchanges: swx_common.h)
                s1 += lane_xor<1>(s1, lane); s2 += lane_xor<1>(s2, lane); s1 += lane_xor<2>(s1, lane); s2 += lane_xor<2>(s2, lane);
                s1 += lane_xor<4>(s1, lane); s2 += lane_xor<4>(s2, lane); s1 += lane_xor<8>(s1, lane); s2 += lane_xor<8>(s2, lane);
            }
            if (li == 0) {
                const float inv = 1.0f / (float)kslice;
                const float mean = s1 * inv;
                float var = s2 * inv - mean * mean;
                var = var > 0.f ? var : 0.f;
                stat[rb] = make_float2(mean, 1.0f / sqrtf(var + 1e-5f));
            }
        }
        __syncthreads();                                    // stat[] visible
        {
            const float2 st = stat[li];
            f16x4v o4;
#pragma unroll
            for (int e = 0; e < 4; ++e) o4[e] = (f16)(st.y * (acc[e] - st.x * c1[e]) + c2[e]);
            *(f16x4v *)(qt + li * DH + wave * 16 + lg * 4) = o4;
        }
        __syncthreads();
        const bool qok = qn < a.nq;
        qf[0][0] = qok ? *(const f16x8 *)(qt + qn * DH + g * 8) : (f16x8)(f16)0;
        qf[0][1] = qok ? *(const f16x8 *)(qt + qn * DH + 32 + g * 8) : (f16x8)(f16)0;
        asm volatile("" ::"v"(pfv));
    } else {
#pragma unroll
    for (int u = 0; u < QG; ++u) {
        const bool qok = q_base0 + u * 16 + qn < a.nq;
        const f16 *qp = Q + ((size_t)b * a.q_rows_per_batch + (qok ? q_base0 + u * 16 + qn : 0)) * a.ldq + h * DH + g * 8;
        qf[u][0] = qok ? *(const f16x8 *)(qp) : (f16x8)(f16)0;
        qf[u][1] = qok ? *(const f16x8 *)(qp + 32) : (f16x8)(f16)0;
    }
    }

    auto compute_blk = [&](int cb, const f16x8 (&kf)[4], const f16x8 (&vf)[4]) {
        const int k0 = cb << 5;
#pragma unroll
        for (int u = 0; u < QG; ++u) {
            f32x4 s[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                s[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
                s[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[2 * t], qf[u][0], s[t], 0, 0, 0);
                s[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[2 * t + 1], qf[u][1], s[t], 0, 0, 0);
            }
            float tmax = -__builtin_inff();
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = k0 + g * 8 + 4 * t + r;
                    const float v = (key < a.nk) ? s[t][r] * 0.125f : -__builtin_inff();
                    s[t][r] = v;
                    tmax = fmaxf(tmax, v);
                }
            tmax = lane_xor16_max_lds(tmax);            // (through the LDS crossbar on purpose in this HBM-streaming kernel: swx_common.h)
            tmax = lane_xor32_max_lds(tmax);
            const float m_new = fmaxf(m_run[u], tmax);
            const float alpha = __expf(m_run[u] - m_new);
            float psum = 0.f;
            f16x8 pb;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __expf(s[t][r] - m_new);
                    psum += p;
                    pb[4 * t + r] = (f16)p;
                }
            psum = lane_xor16_add_lds(psum);
            psum = lane_xor32_add_lds(psum);
            l_run[u] = l_run[u] * alpha + psum;
            m_run[u] = m_new;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                o[u][t][0] *= alpha; o[u][t][1] *= alpha; o[u][t][2] *= alpha; o[u][t][3] *= alpha;
                o[u][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf[t], pb, o[u][t], 0, 0, 0);
            }
        }
    };
    if constexpr (DEPTH > 1) {
        // Round 6: DEPTH key blocks in flight per wave.  The loop below issues a block's eight loads and waits for them in the same
        // iteration (the compiled loop: 8 loads, vmcnt(7) .. vmcnt(0), 8 MFMAs), so a wave pays one memory round trip per block --
        // twelve per launch -- and the launch has W x H x 4 waves x 8 KB = 12.8 MB in flight: by Little's law ~5.5 TB/s at the ~2.3 us
        // a loaded round trip takes, which is what the kernel measures.  Here block n + DEPTH of the wave is requested before block
        // n + 1 is multiplied.  Same blocks in the same order per wave: bit-identical.  Loads past the wave's last block re-read it
        // (clamped, never predicated: a predicated load costs a drained join).  The stages are unconditional inside the loop and the
        // remainder follows it: with a stage under an `if` the wait in front of the first stage has to assume the shorter path and
        // drains the younger blocks' loads too (seen in the ISA of the first form).  Measured on the headline pass, A/B in one process
        // (profiles/r06_c12_*): two blocks + nt 427.3 ms, two blocks 429.8, three blocks + nt 428.9, one block (round 5) 432.9.
        f16x8 kS[DEPTH][4], vS[DEPTH][4];
#pragma unroll
        for (int f = 0; f < 4; ++f) { kS[0][f] = kA[f]; vS[0][f] = vA[f]; }
#pragma unroll
        for (int dd = 1; dd < DEPTH; ++dd) load_blk(wave + 4 * dd, kS[dd], vS[dd]);
        int cb = wave;
        for (; cb + 4 * (DEPTH - 1) < nblk; cb += 4 * DEPTH) {
#pragma unroll
            for (int dd = 0; dd < DEPTH; ++dd) {
                compute_blk(cb + 4 * dd, kS[dd], vS[dd]);
                load_blk(cb + 4 * (DEPTH + dd), kS[dd], vS[dd]);
            }
        }
#pragma unroll
        for (int dd = 0; dd < DEPTH - 1; ++dd)
            if (cb + 4 * dd < nblk) compute_blk(cb + 4 * dd, kS[dd], vS[dd]);
    } else {
        compute_blk(wave, kA, vA);                           // nblk >= 4: every wave owns at least one block
#pragma unroll 2
        for (int cb = wave + 4; cb < nblk; cb += 4) {
            f16x8 kf[4], vf[4];
            load_blk(cb, kf, vf);
            compute_blk(cb, kf, vf);
        }
    }

    // merge the four partial softmaxes, one group at a time: o[u][t][r] is O^T[d = t*16 + g*4 + r][q = qn]
#pragma unroll
    for (int u = 0; u < QG; ++u) {
        const int q_base = q_base0 + u * 16;
        if (q_base >= a.nq) break;
        if (u > 0) __syncthreads();
        if (g == 0) { sm_m[wave][qn] = m_run[u]; sm_l[wave][qn] = l_run[u]; }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) sm_o[wave][t * 16 + g * 4 + r][qn] = o[u][t][r];
        __syncthreads();
        const int nq_here = a.nq - q_base < 16 ? a.nq - q_base : 16;
        for (int i = tid; i < nq_here * DH; i += 256) {
            const int q = i >> 6, d = i & 63;
            const float M = fmaxf(fmaxf(sm_m[0][q], sm_m[1][q]), fmaxf(sm_m[2][q], sm_m[3][q]));
            float L = 0.f, O = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const float f = __expf(sm_m[w][q] - M);
                L += sm_l[w][q] * f;
                O += sm_o[w][d][q] * f;
            }
            ((f16 *)a.o)[((size_t)b * a.q_rows_per_batch + q_base + q) * a.ldo + h * DH + d] = (f16)(O / L);
        }
    }
}

template <bool PACKED, int QG>
__global__ __launch_bounds__(256) void attn_decode_cross_f16(AttnArgs a) { attn_decode_cross_body<PACKED, QG, 0>(a); }
// with the query projection inside: the 40 weight fragments of a wave must not cost the launch its second workgroup per CU
// (W x H = 400 workgroups stream K / V^T concurrently; at one per CU they would run in two rounds) -- two blocks per CU = 256 registers
template <int FQ>
__global__ __launch_bounds__(256, 2) void attn_decode_cross_xq_f16(AttnArgs a) { attn_decode_cross_body<true, 1, FQ>(a); }
// round 6: the K / V^T stream -- 384 KB per (window, head) that ONE workgroup reads once per step and nobody reads again for 5 GB of
// other traffic -- with two key blocks of a wave in flight and the non-temporal load policy (MI355X_MICROARCH.md, row `nt-weights`);
// SWX_FLAG_XATTN_R5 keeps the one-block loop above as its bit-identity reference
template <int FQ>
__global__ __launch_bounds__(256, 2) void attn_decode_cross_xq2_f16(AttnArgs a) { attn_decode_cross_body<true, 1, FQ, 1>(a); }
// (QG = 4: two workgroups per CU -- the four-group form of a many-window scoring pass compiled to 280 registers = ONE workgroup = one wave per SIMD = 64 KB
// in flight per CU without the bound; held to 256 it has 234, no spill: 98 -> 68 us per layer of the 20-window scoring pass, profiles/r06_c34_*)
template <bool PACKED, int QG>
__global__ __launch_bounds__(256, QG == 4 ? 2 : 1) void attn_decode_cross2_f16(AttnArgs a) { attn_decode_cross_body<PACKED, QG, 0, 1>(a); }

// ================================================================================================ flash f32 (round 5)
// Strict-f32 mode on the exact-f32 matrix instruction (v_mfma_f32_16x16x4_f32: f32 operands, f32 accumulate).  Until round 5 every
// f32 attention ran on the VALU kernel below -- 23.8 ms per encoder layer at 20 windows (9.7 TFLOP/s) and 118 us per decode-step
// cross-attention (2.6 TB/s): 1.2 s of the 3.1-s strict pass (profiles/r05_c1_f32pass_kernels.csv).  Same structure as the f16
// kernel: swapped QK^T (S^T = K.Q^T: a lane's four accumulator values belong to ONE query, row statistics = lane-local + two
// shuffles, the exponentiated tile is already the B operand of O^T += V^T.P^T), K / V tiles of 64 keys double-buffered in LDS,
// the next tile's global loads in registers under the current tile's MFMAs.  The k index of a 16 x 16 x 4 instruction is a
// summation index: k-slot g of step (t, e) stands for d = 16 g + 4 t + e (QK^T: one 16-byte LDS read per lane and t) and for
// key 4 g + e of the 16-key block (PV: the S^T accumulator element e IS that key's probability).
//   SPLIT = false: a wave owns 32 queries (two 16-query blocks), the workgroup 128; every wave walks all four 16-key blocks.
//   SPLIT = true (nq <= 16: the decode step's cross-attention, HBM-bound): the four waves share ONE 16-query block and take one
//   16-key block of every tile each; the four partial (max, sum, O) states are merged through LDS at the end.
// V row-major [key][d] (encoder: the fused QKV buffer) or transposed [d][key] (VT: the cross-K/V layout, keys zero padded).
constexpr int F32_KT = 64;           // keys per tile
constexpr int F32_LD = 68;           // floats per LDS row: 16-byte fragment reads of 16 consecutive rows cover all 64 banks
constexpr size_t F32_LDS_BYTES = (size_t)4 * F32_KT * F32_LD * sizeof(float);

template <bool VT, bool SPLIT>
__global__ __launch_bounds__(256, 2) void attn_flash_f32(AttnArgs a)
{
    constexpr int QB = SPLIT ? 1 : 2;            // 16-query blocks per wave
    constexpr int KB = SPLIT ? 1 : 4;            // 16-key blocks of a tile per wave
    extern __shared__ __attribute__((aligned(16))) float f32_smem[];
    float (*Ks)[F32_KT][F32_LD] = (float (*)[F32_KT][F32_LD])f32_smem;                                   // [buf][key][d]
    float (*Vs)[F32_KT][F32_LD] = (float (*)[F32_KT][F32_LD])(f32_smem + 2 * F32_KT * F32_LD);           // [buf][key][d] or, VT, [buf][d][key]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.z, h = blockIdx.y;
    const int q0 = SPLIT ? blockIdx.x * 16 : blockIdx.x * 128 + wave * 32;
    const int qn = lane & 15, g = lane >> 4;
    const float *Q = (const float *)a.q;
    const float *K = (const float *)a.k + (size_t)b * a.k_bs + h * DH;
    const float *V = (const float *)a.v + (size_t)b * a.v_bs + (VT ? (size_t)h * DH * a.vt_kp : (size_t)h * DH);

    f32x4 qf[QB][4];                              // Q[query qn][d = 16 g + 4 t + e]
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int qi = q0 + qb * 16 + qn;
        const float *qp = Q + ((size_t)b * a.q_rows_per_batch + (qi < a.nq ? qi : a.nq - 1)) * a.ldq + h * DH + g * 16;   // clamped
#pragma unroll
        for (int t = 0; t < 4; ++t) qf[qb][t] = *(const f32x4 *)(qp + 4 * t);
    }
    f32x4 o[QB][4];
    float m_run[QB], l_run[QB];                   // l_run: this LANE's share of the row sum (its four keys per block), reduced at the end
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        m_run[qb] = -__builtin_inff(); l_run[qb] = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) o[qb][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

    // staging registers of one tile: thread -> rows (tid >> 4) + 16 i, floats (tid & 15) * 4 .. + 3
    const int sr = tid >> 4, sc4 = (tid & 15) * 4;
    f32x4 rk[4], rv[4];
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto load_tile = [&](int kt0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = sr + 16 * i;
            const int kg = kt0 + r;
            const int kgc = kg < a.nk ? kg : a.nk - 1;                      // clamped, never predicated
            rk[i] = *(const f32x4 *)(K + (size_t)kgc * a.ldkv + sc4);
            if (kg >= a.nk) rk[i] = zero4;
            if constexpr (VT) {
                rv[i] = *(const f32x4 *)(V + (size_t)r * a.vt_kp + kt0 + sc4);          // row = d; zero padded past nk in the source
            } else {
                rv[i] = *(const f32x4 *)(V + (size_t)kgc * a.ldkv + sc4);
                if (kg >= a.nk) rv[i] = zero4;
            }
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *(f32x4 *)&Ks[buf][sr + 16 * i][sc4] = rk[i];
            *(f32x4 *)&Vs[buf][sr + 16 * i][sc4] = rv[i];
        }
    };

    const int ntile = (a.nk + F32_KT - 1) / F32_KT;
    const bool active = SPLIT || q0 < a.nq;       // wave-uniform: a wave without queries only stages tiles
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int t = 0; t < ntile; ++t) {
        const int cur = t & 1, kt0 = t * F32_KT;
        if (t + 1 < ntile) load_tile(kt0 + F32_KT);
        if (active) {
            f32x4 sc[QB][KB];
#pragma unroll
            for (int kbi = 0; kbi < KB; ++kbi) {
                const int kb = SPLIT ? wave : kbi;
                f32x4 kf[4];
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) kf[tt] = *(const f32x4 *)&Ks[cur][kb * 16 + qn][g * 16 + 4 * tt];
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) {
                    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[tt][e], qf[qb][tt][e], acc, 0, 0, 0);
                    sc[qb][kbi] = acc;             // S^T[key kb*16 + 4 g + r][query qn], r = 0..3
                }
            }
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                float tmax = -__builtin_inff();
#pragma unroll
                for (int kbi = 0; kbi < KB; ++kbi) {
                    const int kb = SPLIT ? wave : kbi;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = sc[qb][kbi][r] * 0.125f;
                        if (kt0 + kb * 16 + g * 4 + r >= a.nk) v = -__builtin_inff();
                        sc[qb][kbi][r] = v;
                        tmax = fmaxf(tmax, v);
                    }
                }
                tmax = SPLIT ? lane_xor16_max_lds(tmax) : lane_xor16_max(tmax);     // (SPLIT = the HBM-streaming decode step: swx_common.h)
                tmax = SPLIT ? lane_xor32_max_lds(tmax) : lane_xor32_max(tmax);
                const float m_new = fmaxf(m_run[qb], tmax);
                const float m_use = m_new == -__builtin_inff() ? 0.f : m_new;      // a wave whose keys are all masked so far (SPLIT)
                const float alpha = expf(m_run[qb] - m_use);                        // m_run = -inf -> 0
                float psum = 0.f;
#pragma unroll
                for (int kbi = 0; kbi < KB; ++kbi)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float p = expf(sc[qb][kbi][r] - m_use);
                        sc[qb][kbi][r] = p;
                        psum += p;
                    }
                l_run[qb] = l_run[qb] * alpha + psum;
                m_run[qb] = m_new;
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) { o[qb][tt][0] *= alpha; o[qb][tt][1] *= alpha; o[qb][tt][2] *= alpha; o[qb][tt][3] *= alpha; }
            }
            // O^T[d][query] += V^T[d][key] P^T[key][query]: k-slot g of step r <-> key kb*16 + 4 g + r
#pragma unroll
            for (int kbi = 0; kbi < KB; ++kbi) {
                const int kb = SPLIT ? wave : kbi;
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                    f32x4 vf;
                    if constexpr (VT) {
                        vf = *(const f32x4 *)&Vs[cur][tt * 16 + qn][kb * 16 + g * 4];
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) vf[r] = Vs[cur][kb * 16 + g * 4 + r][tt * 16 + qn];
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int qb = 0; qb < QB; ++qb)
                            o[qb][tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[r], sc[qb][kbi][r], o[qb][tt], 0, 0, 0);
                }
            }
        }
        if (t + 1 < ntile) store_tile(cur ^ 1);
        __syncthreads();
    }

    if constexpr (!SPLIT) {
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            float l = l_run[qb];
            l = lane_xor16_add(l);
            l = lane_xor32_add(l);
            const int qi = q0 + qb * 16 + qn;
            if (active && qi < a.nq) {
                const float inv = 1.0f / l;
                float *op = (float *)a.o + ((size_t)b * a.q_rows_per_batch + qi) * a.ldo + h * DH;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    f32x4 ov;
#pragma unroll
                    for (int r = 0; r < 4; ++r) ov[r] = o[qb][t][r] * inv;
                    *(f32x4 *)(op + t * 16 + g * 4) = ov;          // O^T[d = 16 t + 4 g + r][query qn]
                }
            }
        }
    } else {
        // merge the four waves' states (the loop's last barrier has passed: the tiles are dead) -- [wave][18][64]: max, lane sum, O
        float *sm = f32_smem;
        sm[(wave * 18 + 0) * 64 + lane] = m_run[0];
        sm[(wave * 18 + 1) * 64 + lane] = l_run[0];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) sm[(wave * 18 + 2 + t * 4 + r) * 64 + lane] = o[0][t][r];
        __syncthreads();
        float mw[4], M = -__builtin_inff();
#pragma unroll
        for (int w = 0; w < 4; ++w) { mw[w] = sm[(w * 18 + 0) * 64 + lane]; M = fmaxf(M, mw[w]); }
        float l = 0.f;
        f32x4 ov = (f32x4){0.f, 0.f, 0.f, 0.f};                    // this wave finishes d-tile `wave`
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float sc_w = expf(mw[w] - M);                     // -inf (nothing seen) -> 0
            l += sm[(w * 18 + 1) * 64 + lane] * sc_w;
#pragma unroll
            for (int r = 0; r < 4; ++r) ov[r] += sm[(w * 18 + 2 + wave * 4 + r) * 64 + lane] * sc_w;
        }
        l = lane_xor16_add_lds(l);                  // (the SPLIT form keeps its LDS exchanges: swx_common.h)
        l = lane_xor32_add_lds(l);
        const int qi = q0 + qn;
        if (qi < a.nq) {
            const float inv = 1.0f / l;
#pragma unroll
            for (int r = 0; r < 4; ++r) ov[r] *= inv;
            *(f32x4 *)((float *)a.o + ((size_t)b * a.q_rows_per_batch + qi) * a.ldo + h * DH + wave * 16 + g * 4) = ov;
        }
    }
}

// ============================================================================================ dense rowwise
constexpr int RW_QB = 8;
constexpr int RW_MAXK = 1536;

template <typename T>
__global__ __launch_bounds__(256) void attn_dense_rowwise(AttnArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *qs = smem;                       // [RW_QB][64]
    float *sc = smem + RW_QB * DH;          // [RW_QB][nk_pad]
    const int nkp = (a.nk + 3) & ~3;
    float *part = sc + RW_QB * nkp;         // [4][RW_QB][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * RW_QB;
    const int nqb = min(RW_QB, a.nq - q0);
    const T *Q = (const T *)a.q;
    const T *K = (const T *)a.k + (size_t)b * a.k_bs + h * DH;
    const T *V = (const T *)a.v + (size_t)b * a.v_bs + (a.vt_kp ? (size_t)h * DH * a.vt_kp : (size_t)h * DH);

    for (int i = tid; i < RW_QB * DH; i += 256) {
        const int qi = i >> 6, d = i & 63;
        qs[i] = (qi < nqb) ? to_f32<T>(Q[((size_t)b * a.q_rows_per_batch + q0 + qi) * a.ldq + h * DH + d]) : 0.f;
    }
    __syncthreads();

    // scores
    for (int j = tid; j < a.nk; j += 256) {
        const T *kr = K + (size_t)j * a.ldkv;
        float acc[RW_QB];
#pragma unroll
        for (int qi = 0; qi < RW_QB; ++qi) acc[qi] = 0.f;
#pragma unroll 2
        for (int d0 = 0; d0 < DH; d0 += 8) {
            float kv[8];
            load8<T>(kr + d0, kv);
#pragma unroll
            for (int e = 0; e < 8; ++e)
#pragma unroll
                for (int qi = 0; qi < RW_QB; ++qi) acc[qi] = fmaf(qs[qi * DH + d0 + e], kv[e], acc[qi]);
        }
#pragma unroll
        for (int qi = 0; qi < RW_QB; ++qi) sc[qi * nkp + j] = acc[qi] * 0.125f;
    }
    __syncthreads();

    // softmax: wave handles queries wave, wave+4
    for (int qi = wave; qi < nqb; qi += 4) {
        float *row = sc + qi * nkp;
        float mx = -__builtin_inff();
        for (int j = lane; j < a.nk; j += 64) mx = fmaxf(mx, row[j]);
        mx = wave_max(mx);
        float sum = 0.f;
        for (int j = lane; j < a.nk; j += 64) { const float e = expf(row[j] - mx); row[j] = e; sum += e; }
        sum = wave_sum(sum);
        const float inv = 1.0f / sum;
        for (int j = lane; j < a.nk; j += 64) row[j] *= inv;
    }
    __syncthreads();

    // out = P V : thread (slice = wave, d = lane)
    float acc[RW_QB];
#pragma unroll
    for (int qi = 0; qi < RW_QB; ++qi) acc[qi] = 0.f;
    if (a.vt_kp) {
        // transposed V: lane = d streams its own key row, wave = contiguous quarter of the keys
        const int per = ((a.nk + 31) / 32) * 8;
        const int j_lo = wave * per, j_hi = min(a.nk, j_lo + per);
        const T *vr = V + (size_t)lane * a.vt_kp;
        for (int j0 = j_lo; j0 < j_hi; j0 += 8) {
            float vv[8];
            load8<T>(vr + j0, vv);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (j0 + e < j_hi) {
#pragma unroll
                    for (int qi = 0; qi < RW_QB; ++qi) acc[qi] = fmaf(sc[qi * nkp + j0 + e], vv[e], acc[qi]);
                }
            }
        }
    } else {
        for (int j = wave; j < a.nk; j += 4) {
            const float vv = to_f32<T>(V[(size_t)j * a.ldkv + lane]);
#pragma unroll
            for (int qi = 0; qi < RW_QB; ++qi) acc[qi] = fmaf(sc[qi * nkp + j], vv, acc[qi]);
        }
    }
#pragma unroll
    for (int qi = 0; qi < RW_QB; ++qi) part[(wave * RW_QB + qi) * DH + lane] = acc[qi];
    __syncthreads();
    for (int i = tid; i < nqb * DH; i += 256) {
        const int qi = i >> 6, d = i & 63;
        const float v = part[(0 * RW_QB + qi) * DH + d] + part[(1 * RW_QB + qi) * DH + d] +
                        part[(2 * RW_QB + qi) * DH + d] + part[(3 * RW_QB + qi) * DH + d];
        ((T *)a.o)[((size_t)b * a.q_rows_per_batch + q0 + qi) * a.ldo + h * DH + d] = from_f32<T>(v);
    }
}

// ========================================================================================== cached self-attn
template <typename T>
__global__ __launch_bounds__(256) void kv_append_kernel(SelfAttnArgs a, int row_mul)
{
    // grid (n_new, R): copy k,v of token i of row r into the cache at position pos0[r] + i
    const int i = blockIdx.x, ri = blockIdx.y;
    const int r = ri * row_mul;
    const int pos = a.pos0[r] + i;
    const T *src = (const T *)a.qkv + ((size_t)ri * a.n_new + i) * a.ldqkv;
    T *kc = (T *)a.kcache + ((size_t)r * a.n_ctx + pos) * a.d;
    T *vc = (T *)a.vcache + ((size_t)r * a.n_ctx + pos) * a.d;
    for (int c = threadIdx.x; c < a.d; c += 256) { kc[c] = src[a.d + c]; vc[c] = src[2 * a.d + c]; }
}

template <typename T>
__global__ __launch_bounds__(64) void self_attn_cached(SelfAttnArgs a, int row_mul)
{
    // grid (n_new, H, R): one wave per (row, new token, head); causal over positions [0, pos0 + i]
    __shared__ float qs[DH];
    __shared__ float ps[512];
    const int lane = threadIdx.x;
    const int i = blockIdx.x, h = blockIdx.y, ri = blockIdx.z;
    const int r = ri * row_mul;
    const int pos = a.pos0[r] + i;
    const T *qp = (const T *)a.qkv + ((size_t)ri * a.n_new + i) * a.ldqkv + h * DH;
    qs[lane] = to_f32<T>(qp[lane]);
    __syncthreads();
    const int32_t *anc = a.anc ? a.anc + (size_t)r * a.n_ctx : nullptr;
    float mx = -__builtin_inff();
    for (int j = lane; j <= pos; j += 64) {
        const int pr = anc ? anc[j] : r;
        const T *kr = (const T *)a.kcache + ((size_t)pr * a.n_ctx + j) * a.d + h * DH;
        float acc = 0.f;
#pragma unroll 2
        for (int d0 = 0; d0 < DH; d0 += 8) {
            float kv[8];
            load8<T>(kr + d0, kv);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc = fmaf(qs[d0 + e], kv[e], acc);
        }
        acc *= 0.125f;
        ps[j] = acc;
        mx = fmaxf(mx, acc);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j <= pos; j += 64) { const float e = expf(ps[j] - mx); ps[j] = e; sum += e; }
    sum = wave_sum(sum);
    __syncthreads();
    const float inv = 1.0f / sum;
    // P.V : lane = (key group kg = lane>>3, 8-wide d chunk dc = lane&7): 16-byte V loads, 8 keys per wave instruction
    const int kg = lane >> 3, dc = (lane & 7) * 8;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (int j = kg; j <= pos; j += 8) {
        const int pr = anc ? anc[j] : r;
        const T *vr = (const T *)a.vcache + ((size_t)pr * a.n_ctx + j) * a.d + h * DH + dc;
        float vv[8];
        load8<T>(vr, vv);
        const float pj = ps[j] * inv;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = fmaf(pj, vv[e], acc[e]);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        acc[e] += lane_xor<8>(acc[e], lane);
        acc[e] = lane_xor16_add(acc[e]);
        acc[e] = lane_xor32_add(acc[e]);
    }
    if (kg == 0) {
        T *op = (T *)a.o + ((size_t)ri * a.n_new + i) * a.ldo + h * DH + dc;
#pragma unroll
        for (int e = 0; e < 8; ++e) op[e] = from_f32<T>(acc[e]);
    }
}

// ------------------------------------------------------------------ multi-token self-attention, several tokens per workgroup (round 6)
// self_attn_cached gives every (row, token, head) its own wave, which reads positions 0 .. pos of the head's K and V from L2 again: a 113-token
// scoring pass moves 655 MB per layer through the L1s for 11.6 MB of cache (80 us per layer at 20 windows).  Here a workgroup of NQW waves takes NQW
// CONSECUTIVE tokens of one (row, head): the K and V rows up to its last token's position are staged in LDS ONCE (rows padded to 144 B: a quarter
// wave's 16-byte reads fall on 64 different banks) and every wave runs self_attn_cached's arithmetic on its own token from there -- per score the
// same fmaf chain over d, the same expf, per output element the same key order per lane group and the same lane reduction: bit-identical
// (tests/test_gpu_kernels.py).  f16, no ancestor table, rows that start at position 0 (teacher-forced passes and prefills: the host knows it).
template <int NQW>
__global__ __launch_bounds__(64 * NQW) void self_attn_cached_mq_f16(SelfAttnArgs a, int row_mul, int lds_rows)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char mq_smem[];   // K [lds_rows][72] | V [lds_rows][72] f16 | qs [NQW][64] | ps [NQW][lds_rows] f32
    constexpr int LD = 72;
    f16 *Ks = (f16 *)mq_smem, *Vs = Ks + (size_t)lds_rows * LD;
    float *qs_all = (float *)(Vs + (size_t)lds_rows * LD), *ps_all = qs_all + NQW * DH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i0 = blockIdx.x * NQW, h = blockIdx.y, ri = blockIdx.z;
    const int r = ri * row_mul;
    const int i_last = (i0 + NQW < a.n_new ? i0 + NQW : a.n_new) - 1;       // rows start at position 0: token i attends to positions 0 .. i
    {
        const f16 *kg_ = (const f16 *)a.kcache + (size_t)r * a.n_ctx * a.d + h * DH;
        const f16 *vg_ = (const f16 *)a.vcache + (size_t)r * a.n_ctx * a.d + h * DH;
        for (int c = tid; c < (i_last + 1) * 8; c += 64 * NQW) {
            const int j = c >> 3, c8 = (c & 7) * 8;
            *(f16x8 *)(Ks + j * LD + c8) = *(const f16x8 *)(kg_ + (size_t)j * a.d + c8);
            *(f16x8 *)(Vs + j * LD + c8) = *(const f16x8 *)(vg_ + (size_t)j * a.d + c8);
        }
    }
    const bool live = i0 + wave < a.n_new;
    const int i = live ? i0 + wave : a.n_new - 1;      // a wave past the last token repeats it (every wave reaches the barriers) and stores nothing
    const int pos = i;
    float *qs = qs_all + wave * DH, *ps = ps_all + (size_t)wave * lds_rows;
    qs[lane] = (float)((const f16 *)a.qkv)[((size_t)ri * a.n_new + i) * a.ldqkv + h * DH + lane];
    __syncthreads();
    float mx = -__builtin_inff();
    for (int j = lane; j <= pos; j += 64) {
        const f16 *kr = Ks + j * LD;
        float acc = 0.f;
#pragma unroll 2
        for (int d0 = 0; d0 < DH; d0 += 8) {
            const f16x8 t = *(const f16x8 *)(kr + d0);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc = fmaf(qs[d0 + e], (float)t[e], acc);
        }
        acc *= 0.125f;
        ps[j] = acc;
        mx = fmaxf(mx, acc);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j <= pos; j += 64) { const float e = expf(ps[j] - mx); ps[j] = e; sum += e; }
    sum = wave_sum(sum);
    __syncthreads();
    const float inv = 1.0f / sum;
    const int kg = lane >> 3, dc = (lane & 7) * 8;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (int j = kg; j <= pos; j += 8) {
        const f16x8 t = *(const f16x8 *)(Vs + j * LD + dc);
        const float pj = ps[j] * inv;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = fmaf(pj, (float)t[e], acc[e]);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        acc[e] += lane_xor<8>(acc[e], lane);
        acc[e] = lane_xor16_add(acc[e]);
        acc[e] = lane_xor32_add(acc[e]);
    }
    if (kg == 0 && live) {
        f16 *op = (f16 *)a.o + ((size_t)ri * a.n_new + i) * a.ldo + h * DH + dc;
#pragma unroll
        for (int e = 0; e < 8; ++e) op[e] = (f16)acc[e];
    }
}

// ------------------------------------------------------------------------------------------ decode-step self-attention
// One wave per (row, head) of a single-token decode step, f16: q comes from the QKV projection's output (a.qkv, row stride
// ldqkv), the new token's K / V are ALREADY in the cache at position pos0[r] of row r (the projection's epilogue scattered
// them).  The dependent-load chain is 3 round trips: {pos0} -> {ancestor ids} -> {K rows, V fragments} -> math; the V
// fragments of the first 128 positions are requested together with the K rows (ancestor ids travel between lanes by
// shuffles).  Arithmetic order is self_attn_cached's, so both paths agree bit for bit.
// HBM traffic: K and V of every cached position of every row once = R * (pos + 1) * d * 2 * 2 bytes per launch (57 MB at
// position 111 for 100 rows of large-v3: at the end of a window's decode this kernel is bandwidth-, not latency-bound).
// WPB (round 6 experiment, SWX_FLAG_SELFATTN_WG5): waves per workgroup -- WPB consecutive rows of one head (the beams of a window)
// share a workgroup, i.e. a CU's L1 and one workgroup dispatch; every wave still does exactly its row's arithmetic.
template <bool LONG, int WPB = 1>
__global__ __launch_bounds__(64 * WPB) void self_attn_step_f16(SelfAttnArgs a)
{
    constexpr int PF = 16;                   // prefetched V fragments per lane (keys kg + 8 i, i < PF  <=>  j < 128)
    __shared__ float qs_all[WPB][DH];
    __shared__ float ps_all[WPB][512];
    const int lane = threadIdx.x & 63, wave_in_wg = threadIdx.x >> 6, h = blockIdx.x;
    const int r_raw = blockIdx.y * WPB + wave_in_wg;
    const bool r_live = r_raw < a.R;
    const int r = r_live ? r_raw : a.R - 1;  // a wave past the last row repeats it (every wave reaches the barriers) and does not store
    float *qs = qs_all[wave_in_wg], *ps = ps_all[wave_in_wg];
    const int d = a.d;
    const int pos = a.pos0[r];
    const int32_t *anc = a.anc ? a.anc + (size_t)r * a.n_ctx : nullptr;
    const f16 *kc = (const f16 *)a.kcache, *vc = (const f16 *)a.vcache;
    const int j0 = lane, j1 = lane + 64;
    const bool has0 = j0 <= pos, has1 = j1 <= pos;         // positions that live in the cache (the new one included)
    // ancestor ids: clamped unconditional loads + select (a predicated load would cost its own round trip)
    const int32_t *ap = anc ? anc : a.pos0;
    const bool old0 = j0 < pos, old1 = j1 < pos;           // the ancestor table covers the older positions; the new one is row r's own
    const int t0 = ap[(anc && old0) ? j0 : 0], t1 = ap[(anc && old1) ? j1 : 0];
    const int pr0 = (anc && old0) ? t0 : r, pr1 = (anc && old1) ? t1 : r;
    qs[lane] = (float)((const f16 *)a.qkv)[(size_t)r * a.ldqkv + h * DH + lane];
    // ---- K rows of positions lane, lane + 64 and the V fragments of positions < 128: one batch of loads
    f16x8 k0[8], k1[8];
    {
        const f16 *kr0 = kc + ((size_t)(has0 ? pr0 : r) * a.n_ctx + (has0 ? j0 : 0)) * d + h * DH;   // clamped, never predicated
#pragma unroll
        for (int e = 0; e < 8; ++e) k0[e] = *(const f16x8 *)(kr0 + 8 * e);
        if (pos >= 64) {
            const f16 *kr1 = kc + ((size_t)(has1 ? pr1 : r) * a.n_ctx + (has1 ? j1 : 0)) * d + h * DH;
#pragma unroll
            for (int e = 0; e < 8; ++e) k1[e] = *(const f16x8 *)(kr1 + 8 * e);
        }
    }
    const int kg = lane >> 3, dc = (lane & 7) * 8;
    f16x8 vpre[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) {
        if (i * 8 <= pos) {                                  // uniform: some key of this group of 8 is in the cache
            const int j = kg + 8 * i;
            const int src = j & 63;
            const int pa = __shfl(pr0, src, 64), pb = __shfl(pr1, src, 64);
            const int prj = j < 64 ? pa : pb;                       // (pr0 / pr1 are r for the new position)
            const bool ok = j <= pos;
            vpre[i] = *(const f16x8 *)(vc + ((size_t)(ok ? prj : r) * a.n_ctx + (ok ? j : 0)) * d + h * DH + dc);
        }
    }
    __syncthreads();                                        // qs visible
    // ---- scores (each lane owns whole keys)
    auto dot_regs = [&](const f16x8 *kk) {
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc = fmaf(qs[c * 8 + e], (float)kk[c][e], acc);
        return acc;
    };
    float mx = -__builtin_inff();
    if (j0 <= pos) {
        const float sc = dot_regs(k0) * 0.125f;
        ps[j0] = sc; mx = fmaxf(mx, sc);
    }
    if (pos >= 64 && j1 <= pos) {
        const float sc = dot_regs(k1) * 0.125f;
        ps[j1] = sc; mx = fmaxf(mx, sc);
    }
    // positions >= 128 (a decode that started from a long prompt: sequential transcribe() carries up to 223 tokens over).  The
    // plain loop here was two dependent round trips per 64 positions (ancestor id -> K row); now the ancestor ids of every
    // remaining chunk are requested in one batch and then all K rows in one batch.  Per key the dot product is the same
    // expression, so the scores are bit-identical (the maximum is order-independent).
    if (LONG && pos >= 128) {
        constexpr int KT = 6;                               // 128 + 5 * 64 = 448 = n_text_ctx (six slots: three pairs)
        int prt[KT];
#pragma unroll
        for (int c = 0; c < KT; ++c) {
            const int j = lane + 128 + 64 * c;
            const bool old = anc && j < pos;
            const int t = ap[old ? j : 0];
            prt[c] = old ? t : r;
        }
#pragma unroll
        for (int c2 = 0; c2 < KT; c2 += 2) {                // two chunks of K rows in flight at a time (64 registers, like k0 / k1)
            if (128 + 64 * c2 <= pos) {                     // uniform
                f16x8 kt[2][8];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int c = c2 + u;
                    if (128 + 64 * c <= pos) {              // uniform: the chunk holds a cached position
                        const int j = lane + 128 + 64 * c;
                        const bool has = j <= pos;
                        const f16 *kr = kc + ((size_t)(has ? prt[c] : r) * a.n_ctx + (has ? j : 0)) * d + h * DH;
#pragma unroll
                        for (int e = 0; e < 8; ++e) kt[u][e] = *(const f16x8 *)(kr + 8 * e);
                    }
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int c = c2 + u;
                    const int j = lane + 128 + 64 * c;
                    if (128 + 64 * c <= pos && j <= pos) {
                        const float sc = dot_regs(kt[u]) * 0.125f;
                        ps[j] = sc; mx = fmaxf(mx, sc);
                    }
                }
            }
        }
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j <= pos; j += 64) { const float e = expf(ps[j] - mx); ps[j] = e; sum += e; }
    sum = wave_sum(sum);
    __syncthreads();
    const float inv = 1.0f / sum;
    // ---- P.V : lane = (key group kg, 8-wide d chunk dc); keys in ascending order per lane like the generic kernel
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
    for (int i = 0; i < PF; ++i) {
        const int j = kg + 8 * i;
        if (i * 8 <= pos && j <= pos) {
            const float pj = ps[j] * inv;
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = fmaf(pj, (float)vpre[i][e], acc[e]);
        }
    }
    // keys >= 128: batches of 8 keys per lane group -- ancestor ids in one batch, V fragments in one batch, then the
    // multiply-adds in ascending key order exactly as the one-key-at-a-time loop did them (bit-identical)
    for (int jb = kg + 8 * PF; LONG && jb <= pos; jb += 64) {
        int prv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = jb + 8 * u;
            const bool old = anc && j < pos;
            const int t = ap[old ? j : 0];
            prv[u] = old ? t : r;
        }
        f16x8 vb[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = jb + 8 * u;
            const bool ok = j <= pos;
            vb[u] = *(const f16x8 *)(vc + ((size_t)(ok ? prv[u] : r) * a.n_ctx + (ok ? j : 0)) * d + h * DH + dc);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = jb + 8 * u;
            if (j <= pos) {
                const float pj = ps[j] * inv;
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] = fmaf(pj, (float)vb[u][e], acc[e]);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        acc[e] += lane_xor<8>(acc[e], lane);
        acc[e] = lane_xor16_add(acc[e]);
        acc[e] = lane_xor32_add(acc[e]);
    }
    if (kg == 0 && r_live) {
        f16 *op = (f16 *)a.o + (size_t)r * a.ldo + h * DH + dc;
#pragma unroll
        for (int e = 0; e < 8; ++e) op[e] = (f16)acc[e];
    }
}


// ---------------------------------------------------------------------------- decode-step self-attention, long context, few waves
// Round 6.  A decode that starts from a carried-over prompt (the reference's sequential flow: `model.transcribe(audio)` decodes ONE
// window per call, positions 228-340) runs self_attn_step_f16<true> as R x H = 100 waves, each a chain of DEPENDENT round trips:
// pos0 -> ancestor ids -> K rows of positions < 128 -> K rows 128-255 -> K rows 256-383 -> [softmax] -> per 64 keys (ancestor ids ->
// V fragments): 13 trips at position 340 = the launch's 13 us.  With at most one wave per SIMD there is no neighbour to hide them
// behind and the whole register file belongs to the wave, so this variant requests EVERYTHING a row needs in two batches:
//   1. the ancestor ids of all seven 64-position chunks;
//   2. K rows of positions < 256, the V fragments of positions < 128 -- and, behind them in the queue, K rows of positions 256-447 and
//      the V fragments of positions 128-447 (their ancestor ids travel between lanes by shuffles instead of being loaded again).
// = pos0 + two round trips.  Per score and per output element the expressions and their ORDER are self_attn_step_f16's (one lane owns
// a key's whole dot product; a lane group accumulates its keys in ascending order), so the result is bit-identical
// (tests/hw_checks/self_attn_check.py, tests/test_gpu_model.py).  Dispatched for R x H <= 1024 (SWX_FLAG_SELFATTN_NO_DEEP: A/B).
__global__ __launch_bounds__(64) void self_attn_step_long_f16(SelfAttnArgs a)
{
    constexpr int PF = 16;                   // V fragments per lane of positions < 128 (keys kg + 8 i)
    constexpr int NC = 5;                    // 64-position chunks past 128: 128 + 64 * 5 = 448 = n_text_ctx
    __shared__ float qs[DH];
    __shared__ float ps[512];
    const int lane = threadIdx.x, h = blockIdx.x, r = blockIdx.y;
    const int d = a.d;
    const int pos = a.pos0[r];
    const int32_t *anc = a.anc ? a.anc + (size_t)r * a.n_ctx : nullptr;
    const f16 *kc = (const f16 *)a.kcache, *vc = (const f16 *)a.vcache;
    const int32_t *ap = anc ? anc : a.pos0;
    // ---- batch 1: ancestor ids of every chunk (clamped unconditional loads + select; the new position is row r's own)
    int pr[2 + NC];
#pragma unroll
    for (int c = 0; c < 2 + NC; ++c) {
        const int j = lane + 64 * c;
        const bool old = anc && j < pos;
        const int t = ap[old ? j : 0];
        pr[c] = old ? t : r;
    }
    qs[lane] = (float)((const f16 *)a.qkv)[(size_t)r * a.ldqkv + h * DH + lane];
    // ---- batch 2: K rows (a lane owns whole keys) and V fragments (lane = key group kg, 8-wide d chunk dc)
    auto load_k = [&](int c, f16x8 (&kk)[8]) {
        const int j = lane + 64 * c;
        const bool has = j <= pos;
        const f16 *kr = kc + ((size_t)(has ? pr[c] : r) * a.n_ctx + (has ? j : 0)) * d + h * DH;     // clamped, never predicated
#pragma unroll
        for (int e = 0; e < 8; ++e) kk[e] = *(const f16x8 *)(kr + 8 * e);
    };
    // the V side's ancestor ids (lane group kg reads keys kg + 8 u of every chunk: the id sits in lane kg + 8 u): all 56 exchanges as ONE
    // batch in front of the loads -- inside the guarded load blocks each exchange is an LDS round trip of its own in front of its load
    const int kg = lane >> 3, dc = (lane & 7) * 8;
    int pv[2 + NC][8];
#pragma unroll
    for (int c = 0; c < 2 + NC; ++c)
#pragma unroll
        for (int u = 0; u < 8; ++u) pv[c][u] = __shfl(pr[c], kg + 8 * u, 64);
    __builtin_amdgcn_sched_barrier(0);
    f16x8 kA[4][8];
#pragma unroll
    for (int c = 0; c < 4; ++c)
        if (64 * c <= pos) load_k(c, kA[c]);                // uniform: the chunk holds a cached position
    f16x8 vpre[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) {
        if (i * 8 <= pos) {                                  // uniform
            const int j = kg + 8 * i;
            const int prj = pv[i >> 3][i & 7];
            const bool ok = j <= pos;
            vpre[i] = *(const f16x8 *)(vc + ((size_t)(ok ? prj : r) * a.n_ctx + (ok ? j : 0)) * d + h * DH + dc);
        }
    }
    constexpr int NE = 3;                    // V batches / K chunk requested up front: positions < 320; the rest (320-447) after the first scores
    f16x8 kE[8];
    if (64 * 4 <= pos) load_k(4, kE);
    auto load_vb = [&](int b, f16x8 (&vv)[8]) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = kg + 128 + 64 * b + 8 * u;
            const int prj = pv[2 + b][u];
            const bool ok = j <= pos;
            vv[u] = *(const f16x8 *)(vc + ((size_t)(ok ? prj : r) * a.n_ctx + (ok ? j : 0)) * d + h * DH + dc);
        }
    };
    f16x8 vb[NC][8];
#pragma unroll
    for (int b = 0; b < NE; ++b)
        if (128 + 64 * b <= pos) load_vb(b, vb[b]);          // uniform
    __syncthreads();                                        // qs visible
    // ---- scores
    auto dot_regs = [&](const f16x8 *kk) {
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc = fmaf(qs[c * 8 + e], (float)kk[c][e], acc);
        return acc;
    };
    float mx = -__builtin_inff();
    auto score = [&](int c, const f16x8 *kk) {
        const int j = lane + 64 * c;
        if (64 * c <= pos && j <= pos) {
            const float sc = dot_regs(kk) * 0.125f;
            ps[j] = sc; mx = fmaxf(mx, sc);
        }
    };
#pragma unroll
    for (int c = 0; c < 4; ++c) score(c, kA[c]);
    // positions >= 320 (never reached by the sequential flow's 223-token prompts + 112 steps; one more round trip when they are): their
    // K rows and V fragments take over the registers of the rows just multiplied
    __builtin_amdgcn_sched_barrier(0);
    f16x8 kL[2][8];
#pragma unroll
    for (int c = 5; c < 2 + NC; ++c)
        if (64 * c <= pos) load_k(c, kL[c - 5]);
#pragma unroll
    for (int b = NE; b < NC; ++b)
        if (128 + 64 * b <= pos) load_vb(b, vb[b]);
    score(4, kE);
#pragma unroll
    for (int c = 5; c < 2 + NC; ++c) score(c, kL[c - 5]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j <= pos; j += 64) { const float e = expf(ps[j] - mx); ps[j] = e; sum += e; }
    sum = wave_sum(sum);
    __syncthreads();
    const float inv = 1.0f / sum;
    // ---- P.V : keys in ascending order per lane group
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
    for (int i = 0; i < PF; ++i) {
        const int j = kg + 8 * i;
        if (i * 8 <= pos && j <= pos) {
            const float pj = ps[j] * inv;
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = fmaf(pj, (float)vpre[i][e], acc[e]);
        }
    }
#pragma unroll
    for (int b = 0; b < NC; ++b)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = kg + 128 + 64 * b + 8 * u;
            if (128 + 64 * b <= pos && j <= pos) {
                const float pj = ps[j] * inv;
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] = fmaf(pj, (float)vb[b][u][e], acc[e]);
            }
        }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        acc[e] += lane_xor<8>(acc[e], lane);
        acc[e] = lane_xor16_add(acc[e]);
        acc[e] = lane_xor32_add(acc[e]);
    }
    if (kg == 0) {
        f16 *op = (f16 *)a.o + (size_t)r * a.ldo + h * DH + dc;
#pragma unroll
        for (int e = 0; e < 8; ++e) op[e] = (f16)acc[e];
    }
}


// =============================================================================================== qk capture
template <typename T>
__global__ __launch_bounds__(256) void qk_capture_kernel(const T *__restrict__ q, int64_t ldq, int q_rows_per_w, int row0,
                                                         const T *__restrict__ k, int64_t ldk, int64_t k_bs, int nk,
                                                         const int32_t *__restrict__ heads, int head_slot0, int slots_total,
                                                         float *__restrict__ out, int out_ld_n, int out_ld_f)
{
    // grid (n_rows, n_heads, W)
    __shared__ float qs[DH];
    const int i = blockIdx.x, hs = blockIdx.y, w = blockIdx.z;
    const int head = heads[hs];
    const T *qp = q + ((size_t)w * q_rows_per_w + row0 + i) * ldq + head * DH;
    if (threadIdx.x < DH) qs[threadIdx.x] = to_f32<T>(qp[threadIdx.x]);
    __syncthreads();
    float *orow = out + (((size_t)w * slots_total + head_slot0 + hs) * out_ld_n + i) * out_ld_f;
    for (int f = threadIdx.x; f < nk; f += 256) {
        const T *kr = k + (size_t)w * k_bs + (size_t)f * ldk + head * DH;
        float acc = 0.f;
#pragma unroll 2
        for (int d0 = 0; d0 < DH; d0 += 8) {
            float kv[8];
            load8<T>(kr + d0, kv);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc = fmaf(qs[d0 + e], kv[e], acc);
        }
        orow[f] = acc * 0.125f;
    }
}

// ================================================================================================ xkv pack
// One workgroup per (32-key block, head, window): the block's K rows (32 x 64) and V^T columns (64 x 32) are copied into the
// fragment order attn_decode_cross_f16 consumes (index maps: see load_blk / compute_blk there).
__global__ __launch_bounds__(256) void xkv_pack_kernel(const f16 *__restrict__ K, const f16 *__restrict__ VT, f16 *__restrict__ P,
                                                       int nk, int ldk, int vt_kp, int64_t bs)
{
    const int blk = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, frag = tid >> 6;           // 4 fragments x 64 lanes for K, then for V^T
    const int qn = lane & 15, g = lane >> 4;
    const int64_t per_head = swx_xkv_packed_elems_per_head(nk);
    f16 *out = P + (size_t)b * bs + (size_t)h * per_head + ((size_t)blk * 4 + frag) * 512 + lane * 8;
    {   // K fragment f = 2t + kk: lane (qn, g) holds dims 32 kk + 8 g .. + 8 of key 32 blk + (qn >> 2) * 8 + (qn & 3) + 4 t
        const int t = frag >> 1, kk = frag & 1;
        const int key = blk * 32 + (qn >> 2) * 8 + (qn & 3) + 4 * t;
        f16x8 v = (f16x8)(f16)0;
        if (key < nk) v = *(const f16x8 *)(K + (size_t)b * bs + (size_t)key * ldk + h * DH + kk * 32 + g * 8);
        *(f16x8 *)out = v;
    }
    {   // V^T fragment t: lane (qn, g) holds keys 32 blk + 8 g .. + 8 of row d = 16 t + qn (zero padded past nk in the source)
        const int t = frag;
        const f16x8 v = *(const f16x8 *)(VT + (size_t)b * bs + ((size_t)h * DH + t * 16 + qn) * vt_kp + blk * 32 + g * 8);
        *(f16x8 *)(out + per_head / 2) = v;
    }
}

}  // namespace

int swx_xkv_pack(const void *k, const void *vt, void *packed, int B, int H, int nk, int ldk, int vt_kp, int64_t batch_stride,
                 hipStream_t s)
{
    if (B <= 0) return 0;
    if (((nk + 31) / 32) * 32 > vt_kp) return -5;
    hipLaunchKernelGGL(xkv_pack_kernel, dim3((nk + 31) / 32, H, B), dim3(256), 0, s, (const f16 *)k, (const f16 *)vt, (f16 *)packed,
                       nk, ldk, vt_kp, batch_stride);
    SWX_CHECK_LAUNCH();
    return 0;
}

// The dynamic-LDS attribute of the four attn_flash_f32 instantiations, once per DEVICE (the attribute belongs to the device's code
// object: a process that drives several GPUs sets it on each).  False when the device cannot grant 68 KB: the caller then takes
// the VALU kernel instead of failing.
static bool f32_flash_ready(bool vt, bool split)
{
    constexpr int MAX_DEV = 64;
    static signed char state[MAX_DEV][4] = {};            // 0 unknown, 1 ready, -1 refused
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return false;
    const int v = (vt ? 2 : 0) + (split ? 1 : 0);
    if (state[dev][v] == 0) {
        const void *fn = vt ? (split ? (const void *)attn_flash_f32<true, true> : (const void *)attn_flash_f32<true, false>)
                            : (split ? (const void *)attn_flash_f32<false, true> : (const void *)attn_flash_f32<false, false>);
        const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)F32_LDS_BYTES);
        if (e != hipSuccess) (void)hipGetLastError();
        state[dev][v] = e == hipSuccess ? 1 : -1;
    }
    return state[dev][v] > 0;
}

int swx_attention(int dtype, const AttnArgs &a_in, int force_kernel, hipStream_t s)
{
    const AttnArgs &a = a_in;
    if (a.B <= 0 || a.nq <= 0 || a.nk <= 0) return 0;
    if (a.nk > RW_MAXK) return -5;
    const bool flash = (dtype == SWX_F16) && (force_kernel == 2 || (force_kernel >= 4 && force_kernel <= 6) || (force_kernel == 0 && a.nq >= 32));
    const size_t esz = dtype == SWX_F16 ? 2 : 4;
    // the decode kernel also takes a SMALL multi-row pass (align(): one window of ~100 rows) as groups of 16 rows, when the
    // fragment-ordered K / V^T copy exists and the flash grid would be tiny
    const int ngrp = cdiv(a.nq, 16);
    const bool by_batch = swx_flags() & SWX_FLAG_SCORE_TILED;       // round 3's size-dependent choice (A/B only)
    const bool small_pass = a.nq > 16 && a.kv_packed && !(swx_flags() & SWX_FLAG_NO_PACKED_XKV) && force_kernel == 0 &&
                            (!by_batch || (a.nq <= 160 && (int64_t)a.B * a.H * ngrp <= 1024));
    const bool dec = (dtype == SWX_F16) && a.vt_kp > 0 && (a.nq <= 16 || small_pass) && a.nk >= 128 && (force_kernel == 3 || force_kernel == 0);
    if (force_kernel == 3 && !dec) return -5;
    if (dec) {
        SwxProfScope prof(PC_ATTN_ROWWISE, (double)a.B * a.H * 64 * esz * (2.0 * a.nk + 2.0 * a.nq) +
                                               (a.fq_w ? 2.0 * a.H * 64 * a.fq_k + 2.0 * a.fq_rows * a.fq_k : 0.0), s);
        // groups of 16 query rows per workgroup: one while the launch is small (a decode step; align(): one window = 100-160
        // workgroups), up to four when many windows share the pass (the head's K / V^T is then streamed once per 64 rows)
        const int64_t wgs1 = (int64_t)a.B * a.H * ngrp;
        const int qg = (a.nq <= 16 || wgs1 <= 512) ? 1 : (ngrp >= 4 && wgs1 > 2048) ? 4 : 2;
        dim3 gd(a.H, a.B, cdiv(ngrp, qg));
        const bool packed = a.kv_packed && !(swx_flags() & SWX_FLAG_NO_PACKED_XKV);     // else row-layout K / V^T: the reference
        const bool r5_loop = (swx_flags() & SWX_FLAG_XATTN_R5) != 0;                    // one key block per wave in flight (A/B)
        if (packed && a.fq_w && qg == 1 && a.nq <= 16) {
            // fused query projection: + [16][K] residual tile, statistics, q tile in dynamic LDS
            AttnArgs a = a_in;
            a.fq_full_tile = (swx_flags() & SWX_FLAG_XQ_FULL_TILE) ? 1 : 0;
            const int fq = a.fq_k / 32;
            const size_t lds = (size_t)16 * a.fq_k * 2 + 16 * sizeof(float2) + 16 * DH * 2;

#define SWX_XQ(FQ_) do { \
            static bool attr_done = false; \
            if (!attr_done) { \
                hipError_t e_ = hipFuncSetAttribute((const void *)attn_decode_cross_xq_f16<FQ_>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024); \
                if (e_ != hipSuccess) return -100 - (int)e_; \
                attr_done = true; \
            } \
            static bool attr_done2 = false; \
            if (!r5_loop && !attr_done2) { \
                hipError_t e_ = hipFuncSetAttribute((const void *)attn_decode_cross_xq2_f16<FQ_>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024); \
                if (e_ != hipSuccess) return -100 - (int)e_; \
                attr_done2 = true; \
            } \
            if (!r5_loop) hipLaunchKernelGGL((attn_decode_cross_xq2_f16<FQ_>), gd, dim3(256), lds, s, a); \
            else hipLaunchKernelGGL((attn_decode_cross_xq_f16<FQ_>), gd, dim3(256), lds, s, a); } while (0)
            switch (fq) {
                case 12: SWX_XQ(12); break; case 16: SWX_XQ(16); break; case 20: SWX_XQ(20); break;
                case 24: SWX_XQ(24); break; case 32: SWX_XQ(32); break; case 40: SWX_XQ(40); break;
                default: return -5;
            }
#undef SWX_XQ
        } else if (packed) {
            if (a.fq_w) return -5;
            if (r5_loop) {
                if (qg == 1) hipLaunchKernelGGL((attn_decode_cross_f16<true, 1>), gd, dim3(256), 0, s, a);
                else if (qg == 2) hipLaunchKernelGGL((attn_decode_cross_f16<true, 2>), gd, dim3(256), 0, s, a);
                else hipLaunchKernelGGL((attn_decode_cross_f16<true, 4>), gd, dim3(256), 0, s, a);
            } else {
                if (qg == 1) hipLaunchKernelGGL((attn_decode_cross2_f16<true, 1>), gd, dim3(256), 0, s, a);
                else if (qg == 2) hipLaunchKernelGGL((attn_decode_cross2_f16<true, 2>), gd, dim3(256), 0, s, a);
                else hipLaunchKernelGGL((attn_decode_cross2_f16<true, 4>), gd, dim3(256), 0, s, a);
            }
        } else {
            if (a.fq_w) return -5;
            if (qg == 1) hipLaunchKernelGGL((attn_decode_cross_f16<false, 1>), gd, dim3(256), 0, s, a);
            else if (qg == 2) hipLaunchKernelGGL((attn_decode_cross_f16<false, 2>), gd, dim3(256), 0, s, a);
            else hipLaunchKernelGGL((attn_decode_cross_f16<false, 4>), gd, dim3(256), 0, s, a);
        }
    } else if (flash) {
        if (dtype != SWX_F16) return -5;
        SwxProfScope prof(PC_ATTN_FLASH, 4.0 * a.B * a.H * (double)a.nq * a.nk * 64, s);
        // queries per wave: 64 when the query axis is long (encoder self-attention: 1 500 queries) AND the launch still has a
        // workgroup for every CU that way (at one window 64 queries per wave are 6 x 20 = 120 workgroups), 32 otherwise;
        // force_kernel 4 / 6 / 5 pin 32 / 48 / 64 (scripts/kernel_bench.py --flash-kernel, --only flash_small)
        const bool wide_fills = (int64_t)a.B * a.H * cdiv(a.nq, 256) >= 256;
        // ... and 16 while 32 per wave would leave the launch at one workgroup per CU or less (the encoder of ONE window: 12 x 20 = 240 workgroups
        // = one wave per SIMD on a kernel whose softmax and MFMA stretches want a second wave to fill them; 24 x 20 = 480 that way).  A
        // query block's arithmetic does not depend on how many blocks its wave carries: bit-identical (SWX_FLAG_FLASH_NO_QB1: A/B)
        const bool narrow_fills = a.vt_kp && force_kernel == 0 && a.nq >= 256 && (int64_t)a.B * a.H * cdiv(a.nq, 128) <= 256 &&
                                  !(swx_flags() & SWX_FLAG_FLASH_NO_QB1);
        const int qb = !a.vt_kp ? 2 : narrow_fills ? 1 : force_kernel == 4 ? 2 : force_kernel == 6 ? 3 : (force_kernel == 5 || (a.nq >= 1024 && wide_fills)) ? 4 : 2;
        dim3 g(cdiv(a.nq, 64 * qb), a.H, a.B);
        const bool gen3 = a.vt_kp && (swx_flags() & SWX_FLAG_FLASH_PIPELINED);  // round 6's software-pipelined tile: bit-identical, slower (A/B only)
        if (!a.vt_kp) hipLaunchKernelGGL((attn_flash2_f16<false, 2>), g, dim3(256), 0, s, a);
        else if (gen3 && qb == 4) hipLaunchKernelGGL((attn_flash3_f16<4>), g, dim3(256), 0, s, a);
        else if (gen3 && qb == 3) hipLaunchKernelGGL((attn_flash3_f16<3>), g, dim3(256), 0, s, a);
        else if (gen3 && qb == 2) hipLaunchKernelGGL((attn_flash3_f16<2>), g, dim3(256), 0, s, a);
        else if (qb == 4) hipLaunchKernelGGL((attn_flash2_f16<true, 4>), g, dim3(256), 0, s, a);
        else if (qb == 3) hipLaunchKernelGGL((attn_flash2_f16<true, 3>), g, dim3(256), 0, s, a);
        else if (qb == 1) hipLaunchKernelGGL((attn_flash2_f16<true, 1>), g, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((attn_flash2_f16<true, 2>), g, dim3(256), 0, s, a);
    } else if (dtype == SWX_F32 && (force_kernel == 0 || force_kernel == 7) && a.vt_kp % F32_KT == 0 &&
               // a transposed V is read as V[d][kt0 .. kt0 + 63] for every key tile: its row pitch must cover the padded key axis
               (a.vt_kp == 0 || a.vt_kp >= cdiv(a.nk, F32_KT) * F32_KT) && f32_flash_ready(a.vt_kp != 0, a.nq <= 16)) {
        // strict f32 on the exact-f32 matrix instruction: queries <= 16 per batch item (decode step: HBM-bound, the four waves split
        // the keys) or blocks of 128 queries (encoder self-attention, scoring pass: MFMA-bound).  Where the 68 KB of dynamic LDS
        // cannot be granted (f32_flash_ready: not gfx950) the launch falls through to the VALU kernel below.
        const bool split = a.nq <= 16;
        SwxProfScope prof(split ? PC_ATTN_ROWWISE : PC_ATTN_FLASH,
                          split ? (double)a.B * a.H * 64 * esz * (2.0 * a.nk + 2.0 * a.nq) : 4.0 * a.B * a.H * (double)a.nq * a.nk * 64, s);
        dim3 gd(split ? 1 : cdiv(a.nq, 128), a.H, a.B);
        if (a.vt_kp) {
            if (split) hipLaunchKernelGGL((attn_flash_f32<true, true>), gd, dim3(256), F32_LDS_BYTES, s, a);
            else hipLaunchKernelGGL((attn_flash_f32<true, false>), gd, dim3(256), F32_LDS_BYTES, s, a);
        } else {
            if (split) hipLaunchKernelGGL((attn_flash_f32<false, true>), gd, dim3(256), F32_LDS_BYTES, s, a);
            else hipLaunchKernelGGL((attn_flash_f32<false, false>), gd, dim3(256), F32_LDS_BYTES, s, a);
        }
    } else {
        if (force_kernel == 7) return -5;
        // algorithmic bytes: K and V of every (window, head) once + q in + o out
        SwxProfScope prof(PC_ATTN_ROWWISE, (double)a.B * a.H * 64 * esz * (2.0 * a.nk + 2.0 * a.nq), s);
        dim3 g(cdiv(a.nq, RW_QB), a.H, a.B);
        const int nkp = (a.nk + 3) & ~3;
        const size_t smem = sizeof(float) * ((size_t)RW_QB * DH + (size_t)RW_QB * nkp + (size_t)4 * RW_QB * DH);
        if (dtype == SWX_F16) hipLaunchKernelGGL(attn_dense_rowwise<f16>, g, dim3(256), smem, s, a);
        else hipLaunchKernelGGL(attn_dense_rowwise<float>, g, dim3(256), smem, s, a);
    }
    SWX_CHECK_LAUNCH();
    return 0;
}

int swx_transpose_v(const void *v, int64_t ldv, int64_t v_bs, int n, void *vt, int kp, int64_t vt_bs, int B, int H, hipStream_t s)
{
    if (B <= 0 || n <= 0) return 0;
    if (kp % 8 != 0 || kp < n) return -5;
    hipLaunchKernelGGL(transpose_v_kernel, dim3(cdiv(kp, 64), H, B), dim3(256), 0, s, (const f16 *)v, ldv, v_bs, n, (f16 *)vt, kp, vt_bs);
    SWX_CHECK_LAUNCH();
    return 0;
}

// zeroes `pad_bytes` at byte offset `off_bytes` of each of `rows` rows (row stride `row_bytes`) of `nb` batch items (stride
// `batch_bytes`): the key padding of the transposed cross-attention V (columns n .. kp of every [64 x kp] head block), which
// is multiplied by exact-zero probabilities and therefore has to be finite.  (Filling the whole cross-K/V buffer with zeros
// first -- 9.8 GB per 20-window batch -- cost 1.9 ms per pass.)
__global__ __launch_bounds__(256) void pad_zero_kernel(unsigned char *base, int64_t row_bytes, int64_t batch_bytes, int off_bytes,
                                                       int pad_bytes, int rows)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    unsigned char *p = base + (size_t)blockIdx.y * batch_bytes + (size_t)r * row_bytes + off_bytes;
    if ((((uintptr_t)p) & 7) == 0 && (pad_bytes & 7) == 0) {
        for (int i = 0; i < pad_bytes; i += 8) *(unsigned long long *)(p + i) = 0ull;
    } else {
        for (int i = 0; i < pad_bytes; ++i) p[i] = 0;
    }
}

int swx_pad_zero(void *base, int64_t row_bytes, int64_t batch_bytes, int off_bytes, int pad_bytes, int rows, int nb, hipStream_t s)
{
    if (rows <= 0 || nb <= 0 || pad_bytes <= 0) return 0;
    hipLaunchKernelGGL(pad_zero_kernel, dim3(cdiv(rows, 256), nb), dim3(256), 0, s, (unsigned char *)base, row_bytes, batch_bytes,
                       off_bytes, pad_bytes, rows);
    SWX_CHECK_LAUNCH();
    return 0;
}

int swx_self_attention(int dtype, const SelfAttnArgs &a, int row_mul, hipStream_t s)
{
    if (a.R <= 0 || a.n_new <= 0) return 0;
    if (a.n_ctx > 512) return -5;
    // algorithmic bytes: K and V of every cached position of every row once (+ q in, o out); the positions are device state --
    // the decode loop passes the one it knows (step_pos), multi-token passes attend to n_new positions on average n_new / 2 + 1
    const double npos = a.step_pos > 0 ? a.step_pos + 1 : (a.n_new + 1) * 0.5;
    SwxProfScope prof(PC_SELF_ATTN, (double)a.R * a.n_new * a.d * (dtype == SWX_F16 ? 2 : 4) * (2.0 * npos + 2.0), s);
    if (a.step_cached) {       // single-token step, q in a.qkv, the new K / V already in the cache
        if (dtype != SWX_F16 || a.n_new != 1 || row_mul != 1 || !a.skip_append) return -5;
        // (a decode whose positions stay below 128 -- no prompt carried over -- takes the variant without the long-context code:
        // 157 instead of 211 registers, three waves per SIMD)
        const bool wg5 = (swx_flags() & SWX_FLAG_SELFATTN_WG5) != 0;       // A/B: five rows (a window's beams) per workgroup
        if (a.pos_bound > 0 && a.pos_bound <= 128) {
            if (wg5) hipLaunchKernelGGL((self_attn_step_f16<false, 5>), dim3(a.H, cdiv(a.R, 5)), dim3(320), 0, s, a);
            else hipLaunchKernelGGL((self_attn_step_f16<false, 1>), dim3(a.H, a.R), dim3(64), 0, s, a);
        } else {
            // few waves (one window of the sequential flow: 100): every load of a row in two batches (bit-identical)
            const bool deep = !wg5 && (int64_t)a.R * a.H <= 1024 && a.n_ctx <= 448 && !(swx_flags() & SWX_FLAG_SELFATTN_NO_DEEP);
            if (deep) hipLaunchKernelGGL(self_attn_step_long_f16, dim3(a.H, a.R), dim3(64), 0, s, a);
            else if (wg5) hipLaunchKernelGGL((self_attn_step_f16<true, 5>), dim3(a.H, cdiv(a.R, 5)), dim3(320), 0, s, a);
            else hipLaunchKernelGGL((self_attn_step_f16<true, 1>), dim3(a.H, a.R), dim3(64), 0, s, a);
        }
        SWX_CHECK_LAUNCH();
        return 0;
    }
    dim3 g1(a.n_new, a.R);
    dim3 g2(a.n_new, a.H, a.R);
    if (dtype == SWX_F16) {
        if (!a.skip_append) hipLaunchKernelGGL(kv_append_kernel<f16>, g1, dim3(256), 0, s, a, row_mul);
        // several tokens of a (row, head) per workgroup, K / V staged in LDS once (bit-identical; SWX_FLAG_SELFATTN_NO_MQ: A/B)
        if (a.pos0_all_zero && !a.anc && a.n_new >= 8 && a.n_new <= a.n_ctx && !(swx_flags() & SWX_FLAG_SELFATTN_NO_MQ)) {
            const int nqw = a.n_new >= 32 ? 8 : 4;
            const int lds_rows = ((a.n_new + 7) / 8) * 8;
            const size_t lds = (size_t)lds_rows * 72 * 2 * 2 + (size_t)nqw * 64 * 4 + (size_t)nqw * lds_rows * 4;
            dim3 gq(cdiv(a.n_new, nqw), a.H, a.R);
            if (nqw == 8) {
                static bool attr8 = false;
                if (!attr8) {
                    hipError_t e_ = hipFuncSetAttribute((const void *)self_attn_cached_mq_f16<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
                    if (e_ != hipSuccess) return -100 - (int)e_;
                    attr8 = true;
                }
                hipLaunchKernelGGL(self_attn_cached_mq_f16<8>, gq, dim3(512), lds, s, a, row_mul, lds_rows);
            } else {
                static bool attr4 = false;
                if (!attr4) {
                    hipError_t e_ = hipFuncSetAttribute((const void *)self_attn_cached_mq_f16<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
                    if (e_ != hipSuccess) return -100 - (int)e_;
                    attr4 = true;
                }
                hipLaunchKernelGGL(self_attn_cached_mq_f16<4>, gq, dim3(256), lds, s, a, row_mul, lds_rows);
            }
            SWX_CHECK_LAUNCH();
            return 0;
        }
        hipLaunchKernelGGL(self_attn_cached<f16>, g2, dim3(64), 0, s, a, row_mul);
    } else {
        if (!a.skip_append) hipLaunchKernelGGL(kv_append_kernel<float>, g1, dim3(256), 0, s, a, row_mul);
        hipLaunchKernelGGL(self_attn_cached<float>, g2, dim3(64), 0, s, a, row_mul);
    }
    SWX_CHECK_LAUNCH();
    return 0;
}

int swx_qk_capture(int dtype, const void *q, int64_t ldq, int q_rows_per_w, int row0, int n_rows, const void *k,
                   int64_t ldk, int64_t k_bs, int nk, const int32_t *heads, int n_heads, int head_slot0, int slots_total, int W,
                   float *out, int out_ld_n, int out_ld_f, hipStream_t s)
{
    if (n_rows <= 0 || n_heads <= 0 || W <= 0) return 0;
    dim3 g(n_rows, n_heads, W);
    if (dtype == SWX_F16)
        hipLaunchKernelGGL(qk_capture_kernel<f16>, g, dim3(256), 0, s, (const f16 *)q, ldq, q_rows_per_w, row0, (const f16 *)k, ldk, k_bs, nk, heads, head_slot0, slots_total, out, out_ld_n, out_ld_f);
    else
        hipLaunchKernelGGL(qk_capture_kernel<float>, g, dim3(256), 0, s, (const float *)q, ldq, q_rows_per_w, row0, (const float *)k, ldk, k_bs, nk, heads, head_slot0, slots_total, out, out_ld_n, out_ld_f);
    SWX_CHECK_LAUNCH();
    return 0;
}
