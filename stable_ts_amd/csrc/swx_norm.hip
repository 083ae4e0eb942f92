// swx_norm.hip -- LayerNorm (fp32 statistics, as upstream whisper/model.py::LayerNorm), token+position embedding,
// the channel-last re-layout of the mel for the im2col-free conv stem, and the weight re-layout copies used by
// swx_load_tensor.  All HBM-bound streaming kernels: one pass, coalesced along the feature axis.
#include "swx_common.h"
#include "swx_kernels.h"

namespace {

// one wave per row; the row is read ONCE into registers (8-element chunks, 16 B per lane for f16), statistics by wave
// shuffles, result written once.  d <= 8*64*LN_MAXC.
constexpr int LN_MAXC = 3;
template <typename T>
__global__ __launch_bounds__(256) void layernorm_kernel(const T *__restrict__ x, int64_t ldx, const float *__restrict__ gamma,
                                                        const float *__restrict__ beta, T *__restrict__ y, int64_t ldy,
                                                        int rows, int d)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + wave;
    if (row >= rows) return;
    const T *xr = x + (size_t)row * ldx;
    const int nchunk = d >> 3;
    float v[LN_MAXC][8];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < LN_MAXC; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nchunk) {
            load8<T>(xr + ch * 8, v[c]);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += v[c][e];
        }
    }
    const float mean = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < LN_MAXC; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nchunk) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float t = v[c][e] - mean; q += t * t; }
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)d + 1e-5f);
    T *yr = y + (size_t)row * ldy;
#pragma unroll
    for (int c = 0; c < LN_MAXC; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nchunk) {
            float gm[8], bt[8];
            load8<float>(gamma + ch * 8, gm);
            load8<float>(beta + ch * 8, bt);
#pragma unroll
            for (int e = 0; e < 8; ++e) yr[ch * 8 + e] = from_f32<T>((v[c][e] - mean) * rstd * gm[e] + bt[e]);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void mel_transpose_kernel(const float *__restrict__ mel, int n_mels, int Cp, T *__restrict__ melT)
{
    // melT[b][t][c] for t in [0, 3002): row 0 and 3001 are the conv zero padding
    const int b = blockIdx.y;
    const size_t total = (size_t)3002 * Cp;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int t = (int)(i / Cp), c = (int)(i % Cp);
        float v = 0.f;
        if (t >= 1 && t <= 3000 && c < n_mels) v = mel[((size_t)b * n_mels + c) * 3000 + (t - 1)];
        melT[(size_t)b * total + i] = from_f32<T>(v);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void embed_kernel(const int32_t *__restrict__ tokens, int64_t ld_tok,
                                                    const int32_t *__restrict__ pos0, int n_new, const T *__restrict__ tok_emb,
                                                    const float *__restrict__ pos_emb, int d, T *__restrict__ x)
{
    const int r = blockIdx.y, i = blockIdx.x;
    const int p = pos0[r] + i;
    const int tok = tokens[(size_t)r * ld_tok + p];
    const T *e = tok_emb + (size_t)tok * d;
    const float *pe = pos_emb + (size_t)p * d;
    T *o = x + ((size_t)r * n_new + i) * d;
    for (int c = threadIdx.x; c < d; c += 256) o[c] = from_f32<T>(to_f32<T>(e[c]) + pe[c]);
}

template <typename T>
__global__ __launch_bounds__(256) void copy_rows_kernel(const float *__restrict__ src, int64_t src_ld, T *__restrict__ dst,
                                                        int64_t dst_ld, int64_t rows, int64_t cols)
{
    const int64_t total = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / cols, c = i % cols;
        dst[r * dst_ld + c] = from_f32<T>(src[r * src_ld + c]);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void copy_conv_w_kernel(const float *__restrict__ src, int out_c, int in_c, int in_cp,
                                                          T *__restrict__ dst)
{
    // src [out_c][in_c][3] (torch Conv1d) -> dst [out_c][3][in_cp], zero for c >= in_c
    const int64_t total = (int64_t)out_c * 3 * in_cp;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % in_cp);
        const int tap = (int)((i / in_cp) % 3);
        const int o = (int)(i / ((int64_t)3 * in_cp));
        dst[i] = from_f32<T>(c < in_c ? src[((int64_t)o * in_c + c) * 3 + tap] : 0.f);
    }
}

}  // namespace

int swx_layernorm(int dtype, const void *x, int64_t ldx, const float *gamma, const float *beta, void *y, int64_t ldy,
                  int rows, int d, hipStream_t s)
{
    if (rows <= 0) return 0;
    if (d % 8 != 0 || d > 8 * 64 * LN_MAXC) return -2;
    SwxProfScope prof(PC_NORM, 2.0 * rows * (double)d * (dtype == SWX_F16 ? 2 : 4), s);
    if (dtype == SWX_F16)
        hipLaunchKernelGGL(layernorm_kernel<f16>, dim3(cdiv(rows, 4)), dim3(256), 0, s, (const f16 *)x, ldx, gamma, beta, (f16 *)y, ldy, rows, d);
    else
        hipLaunchKernelGGL(layernorm_kernel<float>, dim3(cdiv(rows, 4)), dim3(256), 0, s, (const float *)x, ldx, gamma, beta, (float *)y, ldy, rows, d);
    SWX_CHECK_LAUNCH();
    return 0;
}

int swx_mel_transpose(int dtype, const float *mel, int B, int n_mels, int Cp, void *melT, hipStream_t s)
{
    if (B <= 0) return 0;
    dim3 g(256, B);
    if (dtype == SWX_F16) hipLaunchKernelGGL(mel_transpose_kernel<f16>, g, dim3(256), 0, s, mel, n_mels, Cp, (f16 *)melT);
    else hipLaunchKernelGGL(mel_transpose_kernel<float>, g, dim3(256), 0, s, mel, n_mels, Cp, (float *)melT);
    SWX_CHECK_LAUNCH();
    return 0;
}

int swx_fill_zero(void *p, size_t bytes, hipStream_t s)
{
    hipError_t e = hipMemsetAsync(p, 0, bytes, s);
    return e == hipSuccess ? 0 : -100 - (int)e;
}

int swx_embed(int dtype, const int32_t *tokens, int64_t ld_tok, const int32_t *, const int32_t *pos0, int R, int n_new,
              const void *tok_emb, const float *pos_emb, int d, void *x, hipStream_t s)
{
    if (R <= 0 || n_new <= 0) return 0;
    dim3 g(n_new, R);
    if (dtype == SWX_F16) hipLaunchKernelGGL(embed_kernel<f16>, g, dim3(256), 0, s, tokens, ld_tok, pos0, n_new, (const f16 *)tok_emb, pos_emb, d, (f16 *)x);
    else hipLaunchKernelGGL(embed_kernel<float>, g, dim3(256), 0, s, tokens, ld_tok, pos0, n_new, (const float *)tok_emb, pos_emb, d, (float *)x);
    SWX_CHECK_LAUNCH();
    return 0;
}

int swx_convert_f32(int dtype, const float *src, void *dst, int64_t n, hipStream_t s)
{
    return swx_copy_rows(dtype, src, n, dst, n, 1, n, s);
}

int swx_copy_rows(int dtype, const float *src, int64_t src_ld, void *dst, int64_t dst_ld, int64_t rows, int64_t cols, hipStream_t s)
{
    if (rows <= 0 || cols <= 0) return 0;
    const int64_t total = rows * cols;
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    if (dtype == SWX_F16) hipLaunchKernelGGL(copy_rows_kernel<f16>, dim3(blocks), dim3(256), 0, s, src, src_ld, (f16 *)dst, dst_ld, rows, cols);
    else hipLaunchKernelGGL(copy_rows_kernel<float>, dim3(blocks), dim3(256), 0, s, src, src_ld, (float *)dst, dst_ld, rows, cols);
    SWX_CHECK_LAUNCH();
    return 0;
}

int swx_copy_conv_w(int dtype, const float *src, int out_c, int in_c, int in_cp, void *dst, hipStream_t s)
{
    const int64_t total = (int64_t)out_c * 3 * in_cp;
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    if (dtype == SWX_F16) hipLaunchKernelGGL(copy_conv_w_kernel<f16>, dim3(blocks), dim3(256), 0, s, src, out_c, in_c, in_cp, (f16 *)dst);
    else hipLaunchKernelGGL(copy_conv_w_kernel<float>, dim3(blocks), dim3(256), 0, s, src, out_c, in_c, in_cp, (float *)dst);
    SWX_CHECK_LAUNCH();
    return 0;
}
