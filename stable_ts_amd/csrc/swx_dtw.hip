// swx_dtw.hip -- dynamic time warping + backtrace on gfx950.
//
// Replaces whisper.timing.dtw (dtw_cpu + backtrace) reached from stable_whisper/timing.py:195.
// Parity target is the CPU recurrence: column-major sweep, strict '<' tie-break
// (ties fall to "left"), f64 add + f32 store.  (double)x + (double)c rounded to f32 is bit-identical
// to a correctly rounded f32 add for every pair of f32 inputs (the f64 sum is exact when the exponents
// are within 29 bits, and otherwise both roundings return the larger operand), so the kernel adds in f32.
//
// Mapping: one 64-lane wavefront per window.  Lane l owns R consecutive token rows; at step t it computes
// column j = t - l + 1, so the wave sweeps a skewed anti-diagonal front and the only cross-lane traffic is
// one shuffle per step (the bottom cell of the lane above).  The sweep is bound by its serial depth
// (M + ceil(N/R) - 1 steps), not by bandwidth.  The 2-bit moves of a lane's R rows for one column are packed
// into one u16 and written to a [M][64] trace plane; the backtrace walks it and emits the path in forward order.
#include "swx_common.h"
#include "swx_kernels.h"

template <int R>
__global__ __launch_bounds__(64) void swx_dtw_kernel(const float *__restrict__ x_all, int ld_n, int ld_m,
                                                     const int *__restrict__ Nw, const int *__restrict__ Mw,
                                                     int *__restrict__ text_idx, int *__restrict__ time_idx,
                                                     int *__restrict__ out_len, unsigned char *__restrict__ ws_all,
                                                     size_t ws_stride)
{
    const int w = blockIdx.x;
    const int lane = threadIdx.x;
    const int N = Nw[w], M = Mw[w];
    const int cap = ld_n + ld_m;
    const float *__restrict__ x = x_all + (size_t)w * ld_n * ld_m;
    unsigned short *trace = (unsigned short *)(ws_all + (size_t)w * ws_stride);
    int *tmp_t = (int *)(ws_all + (size_t)w * ws_stride + (size_t)ld_m * 64 * sizeof(unsigned short));
    int *tmp_f = tmp_t + cap;
    int *o_t = text_idx + (size_t)w * cap;
    int *o_f = time_idx + (size_t)w * cap;

    if (N <= 0 || M <= 0) {  // degenerate: path of the border only
        if (lane == 0) out_len[w] = 0;
        return;
    }
    const int nl = (N + R - 1) / R;  // active lanes
    const float INF = __builtin_inff();

    float prev[R];
#pragma unroll
    for (int r = 0; r < R; ++r) prev[r] = INF;         // cost[i][0] = inf for i >= 1
    float diag_in = (lane == 0) ? 0.0f : INF;          // cost[i0-1][0]; cost[0][0] = 0
    float bottom = INF;
    const int i0 = lane * R;                           // 0-based first row of this lane
    const int steps = M + nl - 1;

    for (int t = 0; t < steps; ++t) {
        float up = __shfl_up(bottom, 1, 64);
        if (lane == 0) up = INF;                       // cost[0][j] = inf for j >= 1
        const int j = t - lane;                        // 0-based column
        if (lane < nl && j >= 0 && j < M) {
            float c0 = diag_in, c1 = up;
            unsigned tr = 0;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (i0 + r < N) {
                    const float c2 = prev[r];
                    const float xv = x[(size_t)(i0 + r) * ld_m + j];
                    float c;
                    unsigned mv;
                    if (c0 < c1 && c0 < c2) { c = c0; mv = 0u; }
                    else if (c1 < c0 && c1 < c2) { c = c1; mv = 1u; }
                    else { c = c2; mv = 2u; }
                    const float nv = __fadd_rn(xv, c);
                    tr |= mv << (2 * r);
                    c0 = c2;       // cost[i][j-1] is the diagonal of the row below
                    c1 = nv;       // cost[i][j]   is "up" of the row below
                    prev[r] = nv;
                }
            }
            bottom = c1;
            diag_in = up;
            trace[(size_t)j * 64 + lane] = (unsigned short)tr;
        }
    }

    // make the trace plane visible to the whole wave before the walk (same workgroup, global memory)
    __threadfence_block();
    __syncthreads();

    // backtrace: every lane walks the same path (uniform loads); lane 0 records it
    int i = N, j = M, n = 0;
    while (i > 0 || j > 0) {
        if (lane == 0) { tmp_t[n] = i - 1; tmp_f[n] = j - 1; }
        ++n;
        unsigned mv;
        if (i == 0) mv = 2u;
        else if (j == 0) mv = 1u;
        else {
            const unsigned wd = trace[(size_t)(j - 1) * 64 + (i - 1) / R];
            mv = (wd >> (2 * ((i - 1) % R))) & 3u;
        }
        if (mv == 0u) { --i; --j; }
        else if (mv == 1u) { --i; }
        else { --j; }
    }
    __threadfence_block();
    __syncthreads();
    for (int p = lane; p < n; p += 64) {
        o_t[p] = tmp_t[n - 1 - p];
        o_f[p] = tmp_f[n - 1 - p];
    }
    if (lane == 0) out_len[w] = n;
}

extern "C" size_t swx_dtw_workspace_bytes(int W, int ld_n, int ld_m)
{
    size_t per = (size_t)ld_m * 64 * sizeof(unsigned short) + 2 * (size_t)(ld_n + ld_m) * sizeof(int);
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
    size_t per = swx_dtw_workspace_bytes(1, ld_n, ld_m);
    const int R = (ld_n + 63) / 64;
#define SWX_DTW_LAUNCH(RR) hipLaunchKernelGGL(swx_dtw_kernel<RR>, dim3(W), dim3(64), 0, s, d_x, ld_n, ld_m, d_N, d_M, \
                                              d_text_idx, d_time_idx, d_len, (unsigned char *)d_trace_ws, per)
    switch (R) {
        case 1: SWX_DTW_LAUNCH(1); break;
        case 2: SWX_DTW_LAUNCH(2); break;
        case 3: SWX_DTW_LAUNCH(3); break;
        case 4: SWX_DTW_LAUNCH(4); break;
        case 5: SWX_DTW_LAUNCH(5); break;
        case 6: SWX_DTW_LAUNCH(6); break;
        default: SWX_DTW_LAUNCH(7); break;
    }
#undef SWX_DTW_LAUNCH
    SWX_CHECK_LAUNCH();
    return 0;
}
