// swx_headsel.hip -- the head-selection variants of the word-timestamp stage as kernels.
//
// Replaces the per-head arithmetic of stable_whisper/timing.py:
//   * :87-103  `dynamic_heads`: per text token, the k heads of the WHOLE decoder whose attention mass lies nearest the token's
//              expected frame (its own peak, or the previous DTW pass's jump midpoints);
//   * :115-163 `aligner='new'`: the top-k sharpest heads of the whole decoder (column / row norms of the softmaxed,
//              median-filtered maps, optional coverage penalty), column-normalised and averaged.
// The reference materialises the cross-attention scores of EVERY head ([L*H][tokens][1500] f32: 0.4-1.7 GB for large-v3) and
// runs a dozen tensor expressions over them.  Here only the cross-attention QUERIES of the teacher-forced pass are kept
// ([L][tokens][d]: <= 37 MB); a head's score row is recomputed from q and the window's cross-K (already resident: swx_cross_kv)
// wherever it is needed -- once to rank the heads, once more for the few rows of the heads that were picked.  HBM traffic:
// L*H*ceil(tokens/1) passes over a 192 KB K head from L2 instead of two passes over the 0.4-1.7 GB score tensor.
//
// Arithmetic follows the reference's expressions (f32 unless it computes in f64); reductions are block trees, so sums differ
// from ATen's by rounding (1e-7 relative) -- a head choice can differ only at a near-tie of two heads' scores.
#include "swx_common.h"
#include "swx_kernels.h"

namespace {

constexpr int DH = 64;
constexpr int HS_MAXF = 1536;     // frames per window (1500) rounded up

// raw scaled scores of one (row, head): s[f] = 0.125 * q . k[f], f in [0, F) -- the arithmetic of qk_capture_kernel
template <typename T>
__device__ __forceinline__ void score_row(const float *qs, const T *kbase, int64_t ldk, int F, float *srow)
{
    for (int f = threadIdx.x; f < F; f += blockDim.x) {
        const T *kr = kbase + (size_t)f * ldk;
        float acc = 0.f;
#pragma unroll 2
        for (int d0 = 0; d0 < DH; d0 += 8) {
            float kv[8];
            load8<T>(kr + d0, kv);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc = fmaf(qs[d0 + e], kv[e], acc);
        }
        srow[f] = acc * 0.125f;
    }
}

__device__ __forceinline__ float block_max(float v, float *sh)
{
    v = wave_max(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    v = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
    __syncthreads();
    return v;
}
__device__ __forceinline__ float block_sum(float v, float *sh)
{
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    v = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    __syncthreads();
    return v;
}
__device__ __forceinline__ double block_sum_d(double v, double *sh)
{
    v = wave_sum_d(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    v = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    __syncthreads();
    return v;
}

__device__ __forceinline__ void cswap2(float &a, float &b) { const float lo = fminf(a, b), hi = fmaxf(a, b); a = lo; b = hi; }
__device__ __forceinline__ int reflect(int k, int n) { if (k < 0) k = -k; if (k >= n) k = 2 * (n - 1) - k; return k; }

// whisper.timing.median_filter along the frames (reflect padding; returned unchanged when F <= width / 2): src -> dst
template <int WIDTH>
__device__ __forceinline__ void median_row(const float *src, float *dst, int F)
{
    constexpr int PADW = WIDTH / 2;
    for (int f = threadIdx.x; f < F; f += blockDim.x) {
        if (F <= PADW) { dst[f] = src[f]; continue; }
        float v[WIDTH];
#pragma unroll
        for (int k = 0; k < WIDTH; ++k) v[k] = src[reflect(f + k - PADW, F)];
#pragma unroll
        for (int a = 1; a < WIDTH; ++a)
#pragma unroll
            for (int b = WIDTH - 1; b >= a; --b) cswap2(v[b - 1], v[b]);
        dst[f] = v[PADW];
    }
}
// any odd width (the reference's median_filter takes any; the sorting networks above cover the usual 3..11): the median is the
// value with at most width/2 smaller and more than width/2 not-larger neighbours -- a rank count, O(width^2) per output
__device__ __forceinline__ void median_row_generic(const float *src, float *dst, int F, int width)
{
    const int P = width / 2;
    for (int f = threadIdx.x; f < F; f += blockDim.x) {
        if (F <= P) { dst[f] = src[f]; continue; }
        float med = src[f];
        for (int a = 0; a < width; ++a) {
            const float va = src[reflect(f + a - P, F)];
            int lt = 0, le = 0;
            for (int b = 0; b < width; ++b) { const float vb = src[reflect(f + b - P, F)]; lt += vb < va; le += vb <= va; }
            if (lt <= P && P < le) { med = va; break; }
        }
        dst[f] = med;
    }
}
__device__ __forceinline__ void median_row_any(const float *src, float *dst, int F, int width)
{
    switch (width) {
        case 1: for (int f = threadIdx.x; f < F; f += blockDim.x) dst[f] = src[f]; break;
        case 3: median_row<3>(src, dst, F); break;
        case 5: median_row<5>(src, dst, F); break;
        case 7: median_row<7>(src, dst, F); break;
        case 9: median_row<9>(src, dst, F); break;
        case 11: median_row<11>(src, dst, F); break;
        default: median_row_generic(src, dst, F, width); break;
    }
}

// in place: row <- softmax(row * scale) over [0, F); returns nothing (values stay in the LDS row)
__device__ __forceinline__ void softmax_row(float *row, int F, float scale, float *sh)
{
    float mx = -__builtin_inff();
    for (int f = threadIdx.x; f < F; f += blockDim.x) { const float v = row[f] * scale; row[f] = v; mx = fmaxf(mx, v); }
    mx = block_max(mx, sh);
    float sum = 0.f;
    for (int f = threadIdx.x; f < F; f += blockDim.x) { const float e = expf(row[f] - mx); row[f] = e; sum += e; }
    sum = block_sum(sum, sh);
    for (int f = threadIdx.x; f < F; f += blockDim.x) row[f] = row[f] / sum;
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------ dynamic heads
// grid (n_rows, H, L): score[(l*H + h) * n_rows + i] = sum_f |peak - f| / 1500 * softmax_f(s * qk_scale)[f]  (timing.py:93-101)
// peaks == null: peak = argmax_f of the row (lowest index on a tie), distances and the sum in f32 like the reference;
// peaks != null: peak[i] = the previous pass's jump midpoint (f64), distances and the sum in f64 like the reference.
template <typename T>
__global__ __launch_bounds__(256) void headsel_score_kernel(const T *__restrict__ qcap, int max_n, int d, int row0, int n_rows,
                                                            const T *__restrict__ xkv, int64_t layer_stride, int H, int F,
                                                            float qk_scale, const double *__restrict__ peaks,
                                                            double *__restrict__ score)
{
    __shared__ float qs[DH];
    __shared__ float srow[HS_MAXF];
    __shared__ float sh[4];
    __shared__ double shd[4];
    __shared__ int sh_arg[4];
    const int i = blockIdx.x, h = blockIdx.y, l = blockIdx.z, tid = threadIdx.x;
    const T *qp = qcap + ((size_t)l * max_n + row0 + i) * d + h * DH;
    if (tid < DH) qs[tid] = to_f32<T>(qp[tid]);
    __syncthreads();
    score_row<T>(qs, xkv + (size_t)l * layer_stride + h * DH, d, F, srow);
    __syncthreads();
    // softmax statistics + argmax of the scaled row
    float mx = -__builtin_inff();
    int arg = 0x7fffffff;
    for (int f = tid; f < F; f += 256) {
        const float v = srow[f] * qk_scale;
        srow[f] = v;
        if (v > mx) { mx = v; arg = f; }            // strided scan: ascending f per thread, '>' keeps the lowest index
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float om = __shfl_xor(mx, o, 64);
        const int oa = __shfl_xor(arg, o, 64);
        if (om > mx || (om == mx && oa < arg)) { mx = om; arg = oa; }
    }
    if ((tid & 63) == 0) { sh[tid >> 6] = mx; sh_arg[tid >> 6] = arg; }
    __syncthreads();
    mx = sh[0]; arg = sh_arg[0];
#pragma unroll
    for (int w = 1; w < 4; ++w) if (sh[w] > mx || (sh[w] == mx && sh_arg[w] < arg)) { mx = sh[w]; arg = sh_arg[w]; }
    __syncthreads();
    float sum = 0.f;
    for (int f = tid; f < F; f += 256) { const float e = expf(srow[f] - mx); srow[f] = e; sum += e; }
    sum = block_sum(sum, sh);
    double out;
    if (peaks) {
        const double pk = peaks[i];
        double acc = 0.0;
        for (int f = tid; f < F; f += 256) acc += (fabs(pk - (double)f) / 1500.0) * (double)(srow[f] / sum);
        out = block_sum_d(acc, shd);
    } else {
        float acc = 0.f;
        for (int f = tid; f < F; f += 256) {
            const int dd = arg - f;
            acc += ((float)(dd < 0 ? -dd : dd) / 1500.0f) * (srow[f] / sum);
        }
        out = (double)block_sum(acc, sh);
    }
    if (tid == 0) score[((size_t)l * H + h) * n_rows + i] = out;
}

// grid (n_rows): sel[i][k] = the k-th smallest score among the LH heads of row i (ascending, lowest head index on a tie):
// torch.topk(count, largest=False).indices
__global__ __launch_bounds__(256) void headsel_pick_kernel(const double *__restrict__ score, int LH, int n_rows, int count,
                                                           int32_t *__restrict__ sel)
{
    extern __shared__ double sc[];          // [LH]
    __shared__ double shv[4];
    __shared__ int shi[4];
    const int i = blockIdx.x, tid = threadIdx.x;
    for (int h = tid; h < LH; h += 256) sc[h] = score[(size_t)h * n_rows + i];
    __syncthreads();
    for (int k = 0; k < count; ++k) {
        double best = __builtin_inf();
        int bi = 0x7fffffff;
        for (int h = tid; h < LH; h += 256) { const double v = sc[h]; if (v < best) { best = v; bi = h; } }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double ov = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ov < best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        if ((tid & 63) == 0) { shv[tid >> 6] = best; shi[tid >> 6] = bi; }
        __syncthreads();
        best = shv[0]; bi = shi[0];
#pragma unroll
        for (int w = 1; w < 4; ++w) if (shv[w] < best || (shv[w] == best && shi[w] < bi)) { best = shv[w]; bi = shi[w]; }
        // no finite score left (NaN scores: an f16 q.K overflow gives a NaN softmax and `v < best` is never true): fall back to the
        // lowest head not picked yet so that the gather kernel never sees an index outside [0, LH) (newal_pick_kernel clamps too)
        if (tid == 0) {
            if (bi >= LH) {
                bi = 0;
                for (int h = 0; h < LH; ++h) {
                    bool used = false;
                    for (int j = 0; j < k; ++j) used |= sel[(size_t)i * count + j] == h;
                    if (!used) { bi = h; break; }
                }
            }
            sel[(size_t)i * count + k] = bi;
            sc[bi] = __builtin_inf();
        }
        __syncthreads();
    }
}

// grid (n_rows, count): out[k][i][f] = raw scaled score of head sel[i][k] for row i, f in [0, nk) -- the layout swx_align_weights
// takes ([H = count][N = n_rows][ld_f]); slot k of every row plays the part of "head k" in the z-normalisation (timing.py:102)
template <typename T>
__global__ __launch_bounds__(256) void headsel_gather_kernel(const T *__restrict__ qcap, int max_n, int d, int row0, int n_rows,
                                                             const T *__restrict__ xkv, int64_t layer_stride, int H, int nk,
                                                             const int32_t *__restrict__ sel, int count, float *__restrict__ out,
                                                             int out_ld_f)
{
    __shared__ float qs[DH];
    const int i = blockIdx.x, k = blockIdx.y, tid = threadIdx.x;
    const int head = sel[(size_t)i * count + k];
    const int l = head / H, h = head - l * H;
    const T *qp = qcap + ((size_t)l * max_n + row0 + i) * d + h * DH;
    if (tid < DH) qs[tid] = to_f32<T>(qp[tid]);
    __syncthreads();
    score_row<T>(qs, xkv + (size_t)l * layer_stride + h * DH, d, nk, out + ((size_t)k * n_rows + i) * out_ld_f);
}

// ------------------------------------------------------------------------------------------------ aligner = 'new'
// grid (H, L): one workgroup walks the n token rows of its head: raw scores -> median over frames -> * qk_scale -> softmax
// (timing.py:136-137), accumulating the per-frame sum of squares (column norms, :143), the row norms (:145) and the per-frame
// coverage (:147-151).  colnorm[(l*H+h)][f] and score[l*H+h] go to memory (3.8 MB for large-v3).
template <typename T>
__global__ __launch_bounds__(256) void newal_head_kernel(const T *__restrict__ qcap, int max_n, int d, int n,
                                                         const T *__restrict__ xkv, int64_t layer_stride, int H, int F,
                                                         float qk_scale, int medfilt_width, float w_col, float w_row, float w_cov,
                                                         float *__restrict__ colnorm, float *__restrict__ score)
{
    __shared__ float qs[DH];
    __shared__ float raw[HS_MAXF], p[HS_MAXF];
    __shared__ float colsq[HS_MAXF], cov[HS_MAXF];
    __shared__ float sh[4];
    const int h = blockIdx.x, l = blockIdx.y, tid = threadIdx.x;
    for (int f = tid; f < F; f += 256) { colsq[f] = 0.f; cov[f] = 0.f; }
    float rownorm_sum = 0.f;
    const T *kbase = xkv + (size_t)l * layer_stride + h * DH;
    for (int i = 0; i < n; ++i) {
        __syncthreads();
        if (tid < DH) qs[tid] = to_f32<T>(qcap[((size_t)l * max_n + i) * d + h * DH + tid]);
        __syncthreads();
        score_row<T>(qs, kbase, d, F, raw);
        __syncthreads();
        median_row_any(raw, p, F, medfilt_width);
        __syncthreads();
        softmax_row(p, F, qk_scale, sh);
        float rs = 0.f;
        for (int f = tid; f < F; f += 256) { const float v = p[f]; colsq[f] += v * v; cov[f] += v; rs += v * v; }
        rs = block_sum(rs, sh);
        rownorm_sum += sqrtf(rs);
    }
    __syncthreads();
    float cs = 0.f, pen = 0.f;
    for (int f = tid; f < F; f += 256) {
        const float cn = sqrtf(colsq[f]);
        colnorm[((size_t)l * H + h) * HS_MAXF + f] = cn;
        cs += cn;
        pen += fmaxf(cov[f], 0.5f);
    }
    cs = block_sum(cs, sh);
    pen = block_sum(pen, sh);
    if (tid == 0) {
        float s = 0.f;
        if (w_col > 0.f) s += w_col * cs;
        if (w_row > 0.f) s += w_row * rownorm_sum;
        if (w_cov > 0.f) s -= w_cov * (pen - (float)F * 0.5f);
        score[(size_t)l * H + h] = s;
    }
}

// one workgroup: top[k] = the k-th LARGEST score (descending, lowest index on a tie): score.flatten().topk(topk).indices
__global__ __launch_bounds__(256) void newal_pick_kernel(const float *__restrict__ score, int LH, int topk, int32_t *__restrict__ top)
{
    extern __shared__ float scf[];
    __shared__ float shv[4];
    __shared__ int shi[4];
    const int tid = threadIdx.x;
    for (int h = tid; h < LH; h += 256) scf[h] = score[h];
    __syncthreads();
    for (int k = 0; k < topk; ++k) {
        float best = -__builtin_inff();
        int bi = 0x7fffffff;
        for (int h = tid; h < LH; h += 256) { const float v = scf[h]; if (v > best) { best = v; bi = h; } }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        if ((tid & 63) == 0) { shv[tid >> 6] = best; shi[tid >> 6] = bi; }
        __syncthreads();
        best = shv[0]; bi = shi[0];
#pragma unroll
        for (int w = 1; w < 4; ++w) if (shv[w] > best || (shv[w] == best && shi[w] < bi)) { best = shv[w]; bi = shi[w]; }
        if (tid == 0) { top[k] = bi < LH ? bi : 0; if (bi < LH) scf[bi] = -__builtin_inff(); }
        __syncthreads();
    }
}

// grid (n_out): row i = row0 + blockIdx.x: out[blockIdx.x][f] = -(1/topk) sum_k p_k[i][f] / colnorm_k[f]   (timing.py:157-163;
// the negation is timing.py:194's DTW input)
template <typename T>
__global__ __launch_bounds__(256) void newal_mean_kernel(const T *__restrict__ qcap, int max_n, int d, int row0,
                                                         const T *__restrict__ xkv, int64_t layer_stride, int H, int F,
                                                         float qk_scale, int medfilt_width, const int32_t *__restrict__ top, int topk,
                                                         const float *__restrict__ colnorm, float *__restrict__ out, int out_ld_f)
{
    __shared__ float qs[DH];
    __shared__ float raw[HS_MAXF], p[HS_MAXF], acc[HS_MAXF];
    __shared__ float sh[4];
    const int i = row0 + blockIdx.x, tid = threadIdx.x;
    for (int f = tid; f < F; f += 256) acc[f] = 0.f;
    for (int k = 0; k < topk; ++k) {
        const int head = top[k];
        const int l = head / H, h = head - l * H;
        __syncthreads();
        if (tid < DH) qs[tid] = to_f32<T>(qcap[((size_t)l * max_n + i) * d + h * DH + tid]);
        __syncthreads();
        score_row<T>(qs, xkv + (size_t)l * layer_stride + h * DH, d, F, raw);
        __syncthreads();
        median_row_any(raw, p, F, medfilt_width);
        __syncthreads();
        softmax_row(p, F, qk_scale, sh);
        const float *cn = colnorm + (size_t)head * HS_MAXF;
        for (int f = tid; f < F; f += 256) acc[f] += p[f] / cn[f];
    }
    __syncthreads();
    for (int f = tid; f < F; f += 256) out[(size_t)blockIdx.x * out_ld_f + f] = -(acc[f] / (float)topk);
}

// out[e] = sum_j coef[j] * x_j[e]: the pooled matrix of several models (extra_models, timing.py:177-189): each model's
// -mean over ITS heads, weighted by its head count / the total
struct WSumArgs { const float *x[8]; float c[8]; int n_in; };
__global__ __launch_bounds__(256) void weighted_sum_kernel(WSumArgs a, float *__restrict__ out, int64_t n)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    float acc = 0.f;
    for (int j = 0; j < a.n_in; ++j) acc += a.c[j] * a.x[j][e];
    out[e] = acc;
}

}  // namespace

int swx_headsel_dynamic_launch(int dtype, const void *qcap, int max_n, int d, int row0, int n_rows, const void *xkv,
                               int64_t layer_stride, int L, int H, int F, int nk, float qk_scale, int count, const double *d_peaks,
                               double *d_score, int32_t *d_sel, float *d_out, int out_ld_f, hipStream_t s)
{
    if (n_rows <= 0) return 0;
    const int LH = L * H;
    if (F <= 0 || F > HS_MAXF || nk > HS_MAXF || count <= 0 || count > LH || LH > 4096) return -2;
    dim3 g1(n_rows, H, L), g3(n_rows, count);
    if (dtype == SWX_F16) {
        hipLaunchKernelGGL(headsel_score_kernel<f16>, g1, dim3(256), 0, s, (const f16 *)qcap, max_n, d, row0, n_rows, (const f16 *)xkv,
                           layer_stride, H, F, qk_scale, d_peaks, d_score);
    } else {
        hipLaunchKernelGGL(headsel_score_kernel<float>, g1, dim3(256), 0, s, (const float *)qcap, max_n, d, row0, n_rows,
                           (const float *)xkv, layer_stride, H, F, qk_scale, d_peaks, d_score);
    }
    hipLaunchKernelGGL(headsel_pick_kernel, dim3(n_rows), dim3(256), (size_t)LH * sizeof(double), s, d_score, LH, n_rows, count, d_sel);
    if (dtype == SWX_F16) {
        hipLaunchKernelGGL(headsel_gather_kernel<f16>, g3, dim3(256), 0, s, (const f16 *)qcap, max_n, d, row0, n_rows, (const f16 *)xkv,
                           layer_stride, H, nk, d_sel, count, d_out, out_ld_f);
    } else {
        hipLaunchKernelGGL(headsel_gather_kernel<float>, g3, dim3(256), 0, s, (const float *)qcap, max_n, d, row0, n_rows,
                           (const float *)xkv, layer_stride, H, nk, d_sel, count, d_out, out_ld_f);
    }
    SWX_CHECK_LAUNCH();
    return 0;
}

int swx_headsel_new_launch(int dtype, const void *qcap, int max_n, int d, int n, int row0, int n_out, const void *xkv,
                           int64_t layer_stride, int L, int H, int F, float qk_scale, int medfilt_width, int topk, float w_col,
                           float w_row, float w_cov, float *d_colnorm, float *d_score, int32_t *d_top, float *d_out, int out_ld_f,
                           hipStream_t s)
{
    if (n <= 0 || n_out <= 0) return 0;
    const int LH = L * H;
    if (F <= 0 || F > HS_MAXF || topk <= 0 || topk > LH || LH > 8192 || row0 < 0 || row0 + n_out > n) return -2;
    if (medfilt_width < 1 || medfilt_width > 2 * HS_MAXF || !(medfilt_width & 1)) return -3;
    dim3 g1(H, L);
    if (dtype == SWX_F16) {
        hipLaunchKernelGGL(newal_head_kernel<f16>, g1, dim3(256), 0, s, (const f16 *)qcap, max_n, d, n, (const f16 *)xkv, layer_stride, H,
                           F, qk_scale, medfilt_width, w_col, w_row, w_cov, d_colnorm, d_score);
    } else {
        hipLaunchKernelGGL(newal_head_kernel<float>, g1, dim3(256), 0, s, (const float *)qcap, max_n, d, n, (const float *)xkv,
                           layer_stride, H, F, qk_scale, medfilt_width, w_col, w_row, w_cov, d_colnorm, d_score);
    }
    hipLaunchKernelGGL(newal_pick_kernel, dim3(1), dim3(256), (size_t)LH * sizeof(float), s, d_score, LH, topk, d_top);
    if (dtype == SWX_F16) {
        hipLaunchKernelGGL(newal_mean_kernel<f16>, dim3(n_out), dim3(256), 0, s, (const f16 *)qcap, max_n, d, row0, (const f16 *)xkv,
                           layer_stride, H, F, qk_scale, medfilt_width, d_top, topk, d_colnorm, d_out, out_ld_f);
    } else {
        hipLaunchKernelGGL(newal_mean_kernel<float>, dim3(n_out), dim3(256), 0, s, (const float *)qcap, max_n, d, row0,
                           (const float *)xkv, layer_stride, H, F, qk_scale, medfilt_width, d_top, topk, d_colnorm, d_out, out_ld_f);
    }
    SWX_CHECK_LAUNCH();
    return 0;
}

int swx_weighted_sum_launch(const float *const *h_xs, const float *h_coef, int n_in, float *d_out, int64_t n, hipStream_t s)
{
    if (n <= 0 || n_in <= 0) return 0;
    if (n_in > 8) return -2;
    WSumArgs a{};
    for (int j = 0; j < n_in; ++j) { a.x[j] = h_xs[j]; a.c[j] = h_coef[j]; }
    a.n_in = n_in;
    hipLaunchKernelGGL(weighted_sum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, d_out, n);
    SWX_CHECK_LAUNCH();
    return 0;
}
