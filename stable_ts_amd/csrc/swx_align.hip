// swx_align.hip -- cross-attention weights -> DTW cost matrix (a7), and the stand-alone median filter.
//
// Replaces stable_whisper/timing.py:105-110 (_compute_atten_weights: softmax over the cropped frame range,
// z-normalisation over the token axis with the population std, median filter of width 7 along frames with
// reflect padding) and timing.py:194-195 (mean over the alignment heads, negation = the DTW input).
// Two launches: (1) softmax statistics of every (window, head, token) row -- the one reduction that runs over the WHOLE
// frame axis; (2) one fused pass per (window, 32-frame tile): softmax values of the tile (+ 3-frame halo) for all tokens in
// LDS, the column statistics over the tokens, z-normalisation, median-7 along the frames, accumulated over the heads ->
// the negated head mean.  The raw scores are read twice (+ 19 % halo); nothing but the [W][H][N] row statistics and the
// final (T+1) x F matrix is written (round 1: three kernels with a normalised copy of all heads and [H][F] statistics in
// HBM).  A single launch would need a grid-wide dependency between the row reduction and the tiles (a grid barrier costs
// more than the launch boundary here, MI355X_MICROARCH.md price list), so the seam stays a kernel boundary.
#include "swx_common.h"
#include "swx_kernels.h"

// ---- launch 1: row statistics (max, sum of exp) of softmax_f(qk * scale) over f in [0, F) ; one wave per (w, h, i) row
__global__ __launch_bounds__(256) void swx_align_rowstats_kernel(const float *__restrict__ qk, float2 *__restrict__ rstat,
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
    float mx = -__builtin_inff();
    for (int f = lane; f < F; f += 64) mx = fmaxf(mx, src[f] * scale);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int f = lane; f < F; f += 64) sum += expf(src[f] * scale - mx);
    sum = wave_sum(sum);
    if (lane == 0) rstat[row] = make_float2(mx, sum);
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

// ---- launch 2: out[w][i][f] = -(1/H) * sum_h median_k( (p[h][i][f+k] - mean[h][f+k]) / sd[h][f+k] ) ------------------------
// One workgroup per (window, AL_FT-frame tile).  Per head: the tile's softmax values p[i][c] for every token i and the tile's
// columns c (frame f0 - PADW + c, reflected at the ends of [0, F) like the reference's reflect padding, so a halo column
// holds the reflected frame's own values and statistics) are built in LDS from the raw scores and the row statistics; the
// mean / population std over the tokens is accumulated per column in f64 (4 row groups per column, then 4 partials);
// the normalised tile replaces p in place; the median runs along the columns.  The head sum stays in registers.
constexpr int AL_FT = 32;
template <int WIDTH>
__global__ __launch_bounds__(256) void swx_align_fused_kernel(const float *__restrict__ qk, const float2 *__restrict__ rstat,
                                                              float *__restrict__ out, int H, int N, int ld_f, int out_ld_n,
                                                              int out_ld_f, const int *__restrict__ n_rows,
                                                              const int *__restrict__ n_frames, float scale)
{
    constexpr int PADW = WIDTH / 2, NC = AL_FT + 2 * PADW, LDC = NC + 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_[];
    float *P = (float *)smem_;                                  // [n][LDC]
    const int w = blockIdx.y, f0 = blockIdx.x * AL_FT, tid = threadIdx.x;
    const int F = n_frames[w], n = n_rows[w];
    if (f0 >= F || n <= 0) return;
    double *part = (double *)(smem_ + (((size_t)N * LDC * 4 + 15) & ~(size_t)15));     // [4][NC][2]
    float *cmean = (float *)(part + 4 * NC * 2), *csd = cmean + NC;
    const bool passthrough = F <= PADW;        // upstream median_filter returns its input when the axis is this short
    constexpr int MAXACC = (448 * AL_FT + 255) / 256;
    float acc[MAXACC];
#pragma unroll
    for (int k = 0; k < MAXACC; ++k) acc[k] = 0.f;
    for (int h = 0; h < H; ++h) {
        const long row0 = (long)(w * H + h) * N;
        __syncthreads();
        for (int e = tid; e < n * NC; e += 256) {
            const int i = e / NC, c = e - i * NC;
            const int ff = reflect_idx(f0 - PADW + c, F);
            float v = 0.f;
            if (ff >= 0 && ff < F) {                            // (F <= PADW: the reflection can fall outside; unused then)
                const float2 st = rstat[row0 + i];
                v = expf(qk[(row0 + i) * (long)ld_f + ff] * scale - st.x) / st.y;
            }
            P[i * LDC + c] = v;
        }
        __syncthreads();
        {   // column statistics over the tokens: thread (c = tid % 64 < NC, row group rg = tid / 64)
            const int c = tid & 63, rg = tid >> 6;
            if (c < NC) {
                double s = 0.0;
                for (int i = rg; i < n; i += 4) s += (double)P[i * LDC + c];
                part[(rg * NC + c) * 2] = s;
            }
            __syncthreads();
            if (tid < NC) {
                const double mu = (part[(0 * NC + tid) * 2] + part[(1 * NC + tid) * 2] + part[(2 * NC + tid) * 2] + part[(3 * NC + tid) * 2]) / n;
                part[tid * 2 + 1] = mu;          // slot [0][c][1]
            }
            __syncthreads();
            if (c < NC) {
                const double mu = part[c * 2 + 1];
                double v = 0.0;
                for (int i = rg; i < n; i += 4) { const double d = (double)P[i * LDC + c] - mu; v += d * d; }
                part[(rg * NC + c) * 2] = v;
            }
            __syncthreads();
            if (tid < NC) {
                const double var = (part[(0 * NC + tid) * 2] + part[(1 * NC + tid) * 2] + part[(2 * NC + tid) * 2] + part[(3 * NC + tid) * 2]) / n;
                cmean[tid] = (float)part[tid * 2 + 1];
                csd[tid] = (float)sqrt(var);
            }
            __syncthreads();
        }
        for (int e = tid; e < n * NC; e += 256) {
            const int i = e / NC, c = e - i * NC;
            P[i * LDC + c] = (P[i * LDC + c] - cmean[c]) / csd[c];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < MAXACC; ++k) {
            const int e = tid + 256 * k;
            const int i = e / AL_FT, c = e - i * AL_FT;
            if (i < n && f0 + c < F) {
                const float *pr = P + i * LDC + c;              // columns c .. c + WIDTH - 1 = frames f - PADW .. f + PADW
                float med;
                if (passthrough) {
                    med = pr[PADW];
                } else {
                    float v[WIDTH];
#pragma unroll
                    for (int kk = 0; kk < WIDTH; ++kk) v[kk] = pr[kk];
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
                acc[k] += med;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < MAXACC; ++k) {
        const int e = tid + 256 * k;
        const int i = e / AL_FT, c = e - i * AL_FT;
        if (i < n && f0 + c < F) out[((long)w * out_ld_n + i) * out_ld_f + f0 + c] = -(acc[k] / (float)H);
    }
}

// device-side launcher shared by swx_score (runtime) and the stand-alone C entry.  d_p: scratch for the row statistics
// (>= W*H*N float2); d_mean / d_sd are no longer used (kept in the signature for the callers' workspace layout)
int swx_align_weights_launch(const float *d_qk, float *d_p, float *d_mean, float *d_sd, int W, int H, int N, int ld_f,
                             const int *d_n_rows, const int *d_n_frames, float qk_scale, int medfilt_width,
                             float *d_neg_matrix, int out_ld_n, int out_ld_f, hipStream_t s)
{
    (void)d_mean; (void)d_sd;
    if (W <= 0 || H <= 0 || N <= 0) return 0;
    if (N > 448) return -2;
    SwxProfScope prof(PC_ALIGN, (double)W * H * N * ld_f * 4.0 * 2, s);
    const long rows = (long)W * H * N;
    float2 *rstat = (float2 *)d_p;
    hipLaunchKernelGGL(swx_align_rowstats_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, d_qk, rstat, W, H, N, ld_f,
                       d_n_rows, d_n_frames, qk_scale);
    dim3 g(cdiv(ld_f, AL_FT), W);
#define SWX_FUSED(WD) do { \
        constexpr int NC_ = AL_FT + 2 * (WD / 2); \
        const size_t lds = (((size_t)N * (NC_ + 1) * 4 + 15) & ~(size_t)15) + (size_t)4 * NC_ * 2 * 8 + (size_t)2 * NC_ * 4; \
        static bool attr_done = false; \
        if (!attr_done) { \
            hipError_t e_ = hipFuncSetAttribute((const void *)swx_align_fused_kernel<WD>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024); \
            if (e_ != hipSuccess) return -100 - (int)e_; \
            attr_done = true; \
        } \
        hipLaunchKernelGGL(swx_align_fused_kernel<WD>, g, dim3(256), lds, s, d_qk, rstat, d_neg_matrix, H, N, ld_f, out_ld_n, out_ld_f, \
                           d_n_rows, d_n_frames, qk_scale); } while (0)
    switch (medfilt_width) {
        case 1: SWX_FUSED(1); break;
        case 3: SWX_FUSED(3); break;
        case 5: SWX_FUSED(5); break;
        case 7: SWX_FUSED(7); break;
        case 9: SWX_FUSED(9); break;
        case 11: SWX_FUSED(11); break;
        default: return -3;
    }
#undef SWX_FUSED
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
