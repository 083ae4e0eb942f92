// swx_align.hip -- cross-attention weights -> DTW cost matrix (a7), and the stand-alone median filter.
//
// Replaces stable_whisper/timing.py:105-110 (_compute_atten_weights: softmax over the cropped frame range,
// z-normalisation over the token axis with the population std, median filter of width 7 along frames with
// reflect padding) and timing.py:194-195 (mean over the alignment heads, negation = the DTW input).
// HBM-bound: three streaming passes over H*(T+1)*F f32 (13.5 MB for large-v3), fused so that the only thing
// written besides the [H][F] statistics is the final (T+1)xF matrix.
#include "swx_common.h"
#include "swx_kernels.h"

// ---- pass 1: p = softmax_f(qk * scale) over f in [0, F) ; one wave per (w, h, i) row --------------------
__global__ __launch_bounds__(256) void swx_align_softmax_kernel(const float *__restrict__ qk, float *__restrict__ p,
                                                                int W, int H, int N, int ld_f,
                                                                const int *__restrict__ n_rows,
                                                                const int *__restrict__ n_frames, float scale)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + wave;     // over W*H*N
    const int i = (int)(row % N);
    const long wh = row / N;
    const int w = (int)(wh / H);
    if (w >= W) return;
    if (i >= n_rows[w]) return;
    const int F = n_frames[w];
    const float *src = qk + row * (long)ld_f;
    float *dst = p + row * (long)ld_f;
    float mx = -__builtin_inff();
    for (int f = lane; f < F; f += 64) mx = fmaxf(mx, src[f] * scale);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int f = lane; f < F; f += 64) sum += expf(src[f] * scale - mx);
    sum = wave_sum(sum);
    for (int f = lane; f < F; f += 64) dst[f] = expf(src[f] * scale - mx) / sum;
}

// ---- pass 2: mean / std over tokens for every (w, h, f) ; thread per frame, coalesced along f ------------
__global__ __launch_bounds__(256) void swx_align_colstats_kernel(const float *__restrict__ p, float *__restrict__ mean,
                                                                 float *__restrict__ sd, int H, int N, int ld_f,
                                                                 const int *__restrict__ n_rows,
                                                                 const int *__restrict__ n_frames)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    const int h = blockIdx.y, w = blockIdx.z;
    const int F = n_frames[w], n = n_rows[w];
    if (f >= F) return;
    const float *base = p + ((long)(w * H + h) * N) * ld_f + f;
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += (double)base[(long)i * ld_f];
    const double mu = s / n;
    double v = 0.0;
    for (int i = 0; i < n; ++i) { const double d = (double)base[(long)i * ld_f] - mu; v += d * d; }
    mean[(long)(w * H + h) * ld_f + f] = (float)mu;
    sd[(long)(w * H + h) * ld_f + f] = (float)sqrt(v / n);
}

__device__ __forceinline__ void cswap(float &a, float &b) { const float lo = fminf(a, b), hi = fmaxf(a, b); a = lo; b = hi; }

// median of 7 by a 16-comparator sorting network restricted to what the middle element needs
__device__ __forceinline__ float median7(float v0, float v1, float v2, float v3, float v4, float v5, float v6)
{
    cswap(v0, v6); cswap(v2, v3); cswap(v4, v5);
    cswap(v0, v2); cswap(v1, v4); cswap(v3, v6);
    cswap(v0, v1); cswap(v2, v5); cswap(v3, v4);
    cswap(v1, v2); cswap(v4, v6);
    cswap(v2, v3); cswap(v4, v5);
    cswap(v1, v2); cswap(v3, v4); cswap(v5, v6);
    return v3;
}

__device__ __forceinline__ int reflect_idx(int k, int n) { if (k < 0) k = -k; if (k >= n) k = 2 * (n - 1) - k; return k; }

// ---- pass 3: out[w][i][f] = -(1/H) * sum_h median_k( (p[h][i][f+k] - mean[h][f+k]) / sd[h][f+k] ) --------------
template <int WIDTH>
__global__ __launch_bounds__(256) void swx_align_finish_kernel(const float *__restrict__ p, const float *__restrict__ mean,
                                                               const float *__restrict__ sd, float *__restrict__ out,
                                                               int H, int N, int ld_f, int out_ld_n, int out_ld_f,
                                                               const int *__restrict__ n_rows,
                                                               const int *__restrict__ n_frames)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y, w = blockIdx.z;
    const int F = n_frames[w];
    if (i >= n_rows[w] || f >= F) return;
    constexpr int PADW = WIDTH / 2;
    float acc = 0.f;
    for (int h = 0; h < H; ++h) {
        const float *prow = p + ((long)(w * H + h) * N + i) * ld_f;
        const float *mrow = mean + (long)(w * H + h) * ld_f;
        const float *srow = sd + (long)(w * H + h) * ld_f;
        float med;
        if (F <= PADW) {   // upstream median_filter returns its input unchanged when the axis is this short
            med = (prow[f] - mrow[f]) / srow[f];
        } else {
            float v[WIDTH];
#pragma unroll
            for (int k = 0; k < WIDTH; ++k) {
                const int ff = reflect_idx(f + k - PADW, F);
                v[k] = (prow[ff] - mrow[ff]) / srow[ff];
            }
            if constexpr (WIDTH == 7) {
                med = median7(v[0], v[1], v[2], v[3], v[4], v[5], v[6]);
            } else {
#pragma unroll
                for (int a = 1; a < WIDTH; ++a)
#pragma unroll
                    for (int b = WIDTH - 1; b >= a; --b) cswap(v[b - 1], v[b]);
                med = v[PADW];
            }
        }
        acc += med;
    }
    out[((long)w * out_ld_n + i) * out_ld_f + f] = -(acc / (float)H);
}

// device-side launcher shared by swx_score (runtime) and the stand-alone C entry
int swx_align_weights_launch(const float *d_qk, float *d_p, float *d_mean, float *d_sd, int W, int H, int N, int ld_f,
                             const int *d_n_rows, const int *d_n_frames, float qk_scale, int medfilt_width,
                             float *d_neg_matrix, int out_ld_n, int out_ld_f, hipStream_t s)
{
    if (W <= 0 || H <= 0 || N <= 0) return 0;
    SwxProfScope prof(PC_ALIGN, (double)W * H * N * ld_f * 4.0 * 3, s);
    const long rows = (long)W * H * N;
    hipLaunchKernelGGL(swx_align_softmax_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, d_qk, d_p, W, H, N, ld_f,
                       d_n_rows, d_n_frames, qk_scale);
    hipLaunchKernelGGL(swx_align_colstats_kernel, dim3(cdiv(ld_f, 256), H, W), dim3(256), 0, s, d_p, d_mean, d_sd, H, N,
                       ld_f, d_n_rows, d_n_frames);
    dim3 g(cdiv(ld_f, 256), N, W);
#define SWX_FIN(WD) hipLaunchKernelGGL(swx_align_finish_kernel<WD>, g, dim3(256), 0, s, d_p, d_mean, d_sd, d_neg_matrix, H, N, \
                                       ld_f, out_ld_n, out_ld_f, d_n_rows, d_n_frames)
    switch (medfilt_width) {
        case 1: SWX_FIN(1); break;
        case 3: SWX_FIN(3); break;
        case 5: SWX_FIN(5); break;
        case 7: SWX_FIN(7); break;
        case 9: SWX_FIN(9); break;
        case 11: SWX_FIN(11); break;
        default: return -3;
    }
#undef SWX_FIN
    SWX_CHECK_LAUNCH();
    return 0;
}

// ---- stand-alone median filter (whisper.timing.median_filter) --------------------------------------------------
template <int WIDTH>
__global__ __launch_bounds__(256) void swx_median_kernel(const float *__restrict__ x, float *__restrict__ out, long rows, int n)
{
    const long row = blockIdx.y;
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= rows || f >= n) return;
    constexpr int PADW = WIDTH / 2;
    const float *src = x + row * (long)n;
    if (n <= PADW) { out[row * (long)n + f] = src[f]; return; }
    float v[WIDTH];
#pragma unroll
    for (int k = 0; k < WIDTH; ++k) v[k] = src[reflect_idx(f + k - PADW, n)];
#pragma unroll
    for (int a = 1; a < WIDTH; ++a)
#pragma unroll
        for (int b = WIDTH - 1; b >= a; --b) cswap(v[b - 1], v[b]);
    out[row * (long)n + f] = v[PADW];
}

extern "C" int swx_median_filter(const float *d_x, int64_t rows, int n, int width, float *d_out, void *stream)
{
    if (rows <= 0 || n <= 0) return 0;
    if (rows > 65535) return -2;
    hipStream_t s = (hipStream_t)stream;
    dim3 g(cdiv(n, 256), (unsigned)rows);
#define SWX_MED(WD) hipLaunchKernelGGL(swx_median_kernel<WD>, g, dim3(256), 0, s, d_x, d_out, (long)rows, n)
    switch (width) {
        case 1: SWX_MED(1); break;
        case 3: SWX_MED(3); break;
        case 5: SWX_MED(5); break;
        case 7: SWX_MED(7); break;
        case 9: SWX_MED(9); break;
        case 11: SWX_MED(11); break;
        default: return -3;
    }
#undef SWX_MED
    SWX_CHECK_LAUNCH();
    return 0;
}
