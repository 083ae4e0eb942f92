// swx_dtw.hip -- dynamic time warping + backtrace on gfx950.
//
// Replaces whisper.timing.dtw (dtw_cpu + backtrace) reached from stable_whisper/timing.py:195.
// Parity target is the CPU recurrence: column-major sweep, strict '<' tie-break
// (ties fall to "left"), f64 add + f32 store.  (double)x + (double)c rounded to f32 is bit-identical
// to a correctly rounded f32 add for every pair of f32 inputs (the f64 sum is exact when the exponents
// are within 29 bits, and otherwise both roundings return the larger operand), so the kernel adds in f32.
//
// The problem is bound by its serial depth (M + N - 1 dependent anti-diagonal steps), not by bandwidth:
// the cost matrix is <= 2.7 MB.  Round 1's kernel paid memory latency per step (0.83 us: a global x load per lane, a global
// trace store); round 2's first LDS wavefront scan (one sweeping wave + three loader waves, x skewed into an LDS ring,
// trace in LDS: 607 us at 226 x 1500) was bounded by ONE wave executing the whole front -- ~75 VALU operations per step for
// 4 rows per lane -- and by a backtrace of one scalar iteration per path element.  Both were deleted in round 3; what
// remains is the third generation below (264 us).
#include <type_traits>
#include <cstdlib>
#include "swx_common.h"
#include "swx_kernels.h"

namespace {

// ------------------------------------------------------------------------------------------ four sweeping waves
// One workgroup (4 waves) per window -- the LDS wavefront scan the north star names:
//   * all four waves of the workgroup sweep: lane L (0..255) owns R consecutive token rows (R = 1 up to 256 tokens, 2 up to
//     512), so a step costs ~19 VALU operations per wave.  Wave w runs one 16-step chunk behind wave w-1 and takes the
//     bottom row of the wave above from a small LDS ring (one workgroup barrier per chunk); inside a wave the neighbour is
//     one DPP move (wave_shr:1).
//   * no LDS staging of x and no loader waves: a lane needs 16 CONSECUTIVE floats of its row per chunk; they are requested
//     three chunks ahead (index clamped into the window; columns outside [0, M) are loaded but never used) and wait in
//     registers.
//   * the moves of a lane's row for the 16 steps of a chunk are one 32-bit word (2 bits per step), written once per chunk to
//     [lane][chunk] (odd pitch: conflict-free); cell (i, j) lives in word (row / R, (j + row / R) >> 4) at bits 2 ((j + row / R) & 15).
//   * backtrace on the scalar unit over RUNS: consecutive "left" moves (the bulk of a path: 1500 frames vs ~226 tokens) are
//     counted with one count-leading-zeros on the trace word, so the walk takes one iteration per run (~N + M/16) instead
//     of one per path element; the runs are expanded to the two index arrays by all 256 threads at the end.
// Same recurrence, same tie-break, same f32 adds as the CPU rule: bit-identical paths (tests/test_gpu_kernels.py DTW tests).
constexpr int D4_CH = 16, D4_BR = 64;

template <int R, bool TLDS>
__global__ __launch_bounds__(256) void swx_dtw4_kernel(const float *__restrict__ x_all, int ld_n, int ld_m,
                                                       const int *__restrict__ Nw, const int *__restrict__ Mw,
                                                       int *__restrict__ text_idx, int *__restrict__ time_idx,
                                                       int *__restrict__ out_len, unsigned char *__restrict__ ws_all,
                                                       size_t ws_stride, int PITCH)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *bnd = (float *)smem;                                   // [3][D4_BR]: bottom row of waves 0..2, ring over steps
    unsigned *runs = (unsigned *)(smem + 3 * D4_BR * 4);          // [2 * (ld_n + ld_m)]: (entry, len | kind << 11 | off << 12)
    unsigned *tr_lds = runs + 2 * (size_t)(ld_n + ld_m);          // [256][PITCH][R] (TLDS)
    __shared__ int sh_n, sh_nruns;
    const int w = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N = Nw[w], M = Mw[w];
    const int cap = ld_n + ld_m;
    unsigned *tr = TLDS ? tr_lds : (unsigned *)(ws_all + (size_t)w * ws_stride);
    int *o_t = text_idx + (size_t)w * cap;
    int *o_f = time_idx + (size_t)w * cap;
    if (N <= 0 || M <= 0) {
        if (tid == 0) out_len[w] = 0;
        return;
    }
    const int nl = (N + R - 1) / R;                    // active lanes (<= 256)
    const int steps = M + nl - 1;
    const int nch = (steps + D4_CH - 1) / D4_CH;
    const float INF = __builtin_inff();
    const int L = tid;

    // x of this window: a lane needs 16 consecutive floats of each of its rows per chunk: four 16-byte loads (4-byte aligned;
    // one float per load measured 4x slower here -- every lane is on its own cache line, so the 64 line requests of a wave
    // instruction, not the bytes, are the cost).  A chunk that would run past the END of the window's matrix (only columns
    // >= M of the last rows, never used) is fetched element by element with the index clamped to the last element, so that no
    // address leaves the caller's buffer.  The index is never negative: row L*R+r starts L columns in.
    typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
    const float *xw = x_all + (size_t)w * ld_n * ld_m;
    const unsigned last_e = (unsigned)(ld_n * ld_m - 1);
    float xc[R][D4_CH], x1[R][D4_CH], x2[R][D4_CH];
    auto issue = [&](int c, float (&dst)[R][D4_CH]) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const unsigned e0 = (unsigned)((L * R + r) * ld_m + (c * D4_CH - L));
            if (e0 + (D4_CH - 1) <= last_e) {
#pragma unroll
                for (int g = 0; g < D4_CH / 4; ++g) {
                    const f4u v = *(const f4u *)(xw + e0 + 4 * g);
                    dst[r][4 * g] = v[0]; dst[r][4 * g + 1] = v[1]; dst[r][4 * g + 2] = v[2]; dst[r][4 * g + 3] = v[3];
                }
            } else {
#pragma unroll
                for (int q = 0; q < D4_CH; ++q) {
                    const unsigned e = e0 + q;
                    dst[r][q] = xw[e < last_e ? e : last_e];
                }
            }
        }
    };
    issue(0, xc); issue(1, x1); issue(2, x2);
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int q = 0; q < D4_CH; ++q) xc[r][q] = (q - L >= 0) ? xc[r][q] : INF;      // chunk 0: columns j = q - L

    float prev[R];
#pragma unroll
    for (int r = 0; r < R; ++r) prev[r] = INF;
    float diag_in = (L == 0) ? 0.0f : INF;
    float bottom = INF;

    for (int p = 0; p < nch + 3; ++p) {
        const int c = p - wave;
        if (c >= 0 && c < nch) {
            const int t0 = c * D4_CH;
            const int jbase = t0 - L;
            float bv[D4_CH];
#pragma unroll
            for (int q = 0; q < D4_CH; ++q) bv[q] = wave > 0 ? bnd[(wave - 1) * D4_BR + ((t0 - 1 + q) & (D4_BR - 1))] : INF;
            unsigned acc[R];
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] = 0u;
            int bcol = 0;
            // No per-step "is this lane inside its column range" selects on the dependent chain:
            //  * before a lane's first column (j < 0) its x is +inf (set when the chunk's registers are rotated in), so its
            //    cells stay +inf -- the border values cost[i][0] the recurrence expects, and `diag_in` = the previous `up` is
            //    +inf likewise (the lane above starts one step earlier);
            //  * after its last column (j >= M), and for lanes / rows past N, it computes on whatever the loads returned: those
            //    cells feed only cells that are outside the matrix as well, and their trace bits are never read.
#pragma unroll
            for (int q = 0; q < D4_CH; ++q) {
                const float up = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(bv[q]), __float_as_int(bottom), 0x138, 0xf, 0xf, false));
                float c0 = diag_in, c1 = up;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const float c2 = prev[r];
                    const bool s0 = (c0 < c1) & (c0 < c2);
                    const bool s1 = (c1 < c0) & (c1 < c2);
                    const float cm = s0 ? c0 : (s1 ? c1 : c2);
                    acc[r] |= (s0 ? 0u : (s1 ? 1u : 2u)) << (2 * q);
                    const float nv = __fadd_rn(xc[r][q], cm);
                    prev[r] = nv;
                    c0 = c2;
                    c1 = nv;
                }
                bottom = c1;
                diag_in = up;
                // lane 63's bottom after this step -> lane q of bcol (handed to the wave below at the end of the chunk)
                {
                    const int sb = __builtin_amdgcn_readlane(__float_as_int(bottom), 63);
                    asm("v_writelane_b32 %0, %1, %2" : "+v"(bcol) : "s"(sb), "n"(q));
                }
            }
#pragma unroll
            for (int r = 0; r < R; ++r) tr[((size_t)L * PITCH + c) * R + r] = acc[r];
            if (wave < 3 && lane < D4_CH) bnd[wave * D4_BR + ((t0 + lane) & (D4_BR - 1))] = __int_as_float(bcol);
            const int jnext = jbase + D4_CH;                 // first column of the next chunk for this lane
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int q = 0; q < D4_CH; ++q) { xc[r][q] = (jnext + q >= 0) ? x1[r][q] : INF; x1[r][q] = x2[r][q]; }
            issue(c + 3, x2);
        }
        __syncthreads();
    }
    if (!TLDS) { __threadfence_block(); __syncthreads(); }

    // ---- backtrace over runs (wave 0, scalar), then expansion by all threads
    if (wave == 0) {
        int i = N, j = M, n = 0, k = 0;
        while (i > 0 || j > 0) {
            unsigned entry = (unsigned)((i - 1) & 0xFFFF) | ((unsigned)((j - 1) & 0xFFFF) << 16);
            int len = 1, kind = 0, di = 0, dj = 0;
            if (i == 0) { len = j; kind = 0; dj = j; }
            else if (j == 0) { len = i; kind = 1; di = i; }
            else {
                const int row = i - 1, Lr = row / R, rr = row - Lr * R;
                const int t = (j - 1) + Lr, c = t >> 4, q = t & 15;
                const unsigned word = (unsigned)__builtin_amdgcn_readfirstlane((int)tr[((size_t)Lr * PITCH + c) * R + rr]);
                const unsigned mv = (word >> (2 * q)) & 3u;
                if (mv == 2u) {
                    const unsigned u = (word ^ 0xAAAAAAAAu) << (30 - 2 * q);       // 2-bit fields == 0 where the move is "left"
                    int run = u ? (__builtin_clz(u) >> 1) : 16;
                    run = run < q + 1 ? run : q + 1;
                    run = run < j ? run : j;
                    len = run; kind = 0; dj = run;
                } else if (mv == 0u) { di = 1; dj = 1; }
                else { di = 1; }
            }
            if (lane == 0) { runs[2 * k] = entry; runs[2 * k + 1] = (unsigned)len | ((unsigned)kind << 11) | ((unsigned)n << 12); }
            ++k;
            n += len;
            i -= di;
            j -= dj;
        }
        if (lane == 0) { sh_n = n; sh_nruns = k; }
    }
    __syncthreads();
    const int n = sh_n, nruns = sh_nruns;
    for (int k = tid; k < nruns; k += 256) {
        const unsigned entry = runs[2 * k], b = runs[2 * k + 1];
        const int len = (int)(b & 0x7FFu), kind = (int)((b >> 11) & 1u), off = (int)(b >> 12);
        const int r16 = (int)(entry & 0xFFFFu), c16 = (int)(entry >> 16);
        const int r0 = r16 == 0xFFFF ? -1 : r16, c0 = c16 == 0xFFFF ? -1 : c16;
        for (int q = 0; q < len; ++q) {
            const int pidx = n - 1 - (off + q);
            o_t[pidx] = kind ? r0 - q : r0;
            o_f[pidx] = kind ? c0 : c0 - q;
        }
    }
    if (tid == 0) out_len[w] = n;
}

inline int dtw4_pitch(int ld_n, int ld_m)
{
    const int lanes = ld_n < 256 ? ld_n : 256;
    return (((ld_m + lanes - 1) + D4_CH - 1) / D4_CH + 3) | 1;
}

}  // namespace

extern "C" size_t swx_dtw_workspace_bytes(int W, int ld_n, int ld_m)
{
    // per window: the trace words of the long-token-axis variant (R = 2, trace in memory)
    size_t per = (size_t)256 * dtw4_pitch(ld_n, ld_m) * 2 * sizeof(unsigned);
    per = (per + 255) & ~(size_t)255;
    return per * (size_t)(W > 0 ? W : 1);
}

extern "C" int swx_dtw(const float *d_x, int W, int ld_n, int ld_m, const int32_t *d_N, const int32_t *d_M,
                       int32_t *d_text_idx, int32_t *d_time_idx, int32_t *d_len, void *d_trace_ws, void *stream)
{
    if (W <= 0) return 0;
    if (ld_n <= 0 || ld_m <= 0 || ld_n > 448) return -2;
    hipStream_t s = (hipStream_t)stream;
    SwxProfScope prof(PC_DTW, (double)W * ld_n * ld_m * 5.0, s);
    const size_t per = swx_dtw_workspace_bytes(1, ld_n, ld_m);
    if (ld_n + ld_m >= 65535 || ld_m >= 32768) return -2;      // run records pack (offset, length) into 32 bits
    {
        const int PITCH = dtw4_pitch(ld_n, ld_m);
        const size_t base = 3 * D4_BR * 4 + 2 * (size_t)(ld_n + ld_m) * 4;
#define SWX_DTW4(RR, TL) do { \
            const size_t lds = base + ((TL) ? (size_t)256 * PITCH * RR * 4 : 0); \
            static bool attr_done4 = false; \
            if (!attr_done4) { \
                hipError_t e_ = hipFuncSetAttribute((const void *)swx_dtw4_kernel<RR, TL>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024); \
                if (e_ != hipSuccess) return -100 - (int)e_; \
                attr_done4 = true; \
            } \
            hipLaunchKernelGGL((swx_dtw4_kernel<RR, TL>), dim3(W), dim3(256), lds, s, d_x, ld_n, ld_m, d_N, d_M, d_text_idx, d_time_idx, \
                               d_len, (unsigned char *)d_trace_ws, per, PITCH); } while (0)
        const bool fits = base + (size_t)256 * PITCH * 4 <= 160 * 1024 - 1024 - 64;
        if (ld_n <= 256) { if (fits) SWX_DTW4(1, true); else SWX_DTW4(1, false); }
        else SWX_DTW4(2, false);
#undef SWX_DTW4
    }
    SWX_CHECK_LAUNCH();
    return 0;
}
