// Device half of the non-VAD silence analysis (stable_whisper/stabilization/nonvad.py:16-39 audio2loudness).
//
// The reference computes, per 30-s window on the host: |x|, the k-th largest value of it (k = 0.1 % of the samples:
// torch.topk(x, k).values[-1]), x / min(1, 1.75 thr), and a linear down-sampling to one value per 20 ms.  Only the k-th
// largest value needs every sample; the down-sampling reads two samples per output.  This kernel produces exactly those
// two things from the PCM that is already resident for the spectrogram -- the threshold by a radix select on the bit
// patterns (a selection, no arithmetic: the value IS an element of the input, bit for bit what the host's selection
// returns) and |x| at a caller-given index list -- so that the host copies out 24 KB per window instead of 1.9 MB and
// does no O(n) work.  Everything that involves floating-point arithmetic (the division, the interpolation weights, the
// pooling and quantisation) stays in the host code that the reference's own expressions run through, on the gathered
// samples, so the mask is the reference's mask by construction (stable_ts_amd/stabilization.py::loudness_from_probe).
#include "swx_common.h"
#include "swx_kernels.h"

namespace {

constexpr int LP_T = 1024;

// grid (W).  nk[w] = {n valid samples, k}.  out[w] = { k-th largest |x| (NaN when k == 0), |x[idx[w][j]]| for j < n_idx }.
__global__ __launch_bounds__(LP_T) void loudness_probe_kernel(const float *__restrict__ pcm, int64_t pcm_stride,
                                                              const int32_t *__restrict__ nk, const int32_t *__restrict__ idx,
                                                              int n_idx, float *__restrict__ out)
{
    __shared__ unsigned hist[256];
    __shared__ unsigned sh_prefix, sh_remaining;
    const int w = blockIdx.x, tid = threadIdx.x;
    const int n = nk[2 * w], k = nk[2 * w + 1];
    const unsigned *x = (const unsigned *)(pcm + (size_t)w * pcm_stride);
    float *o = out + (size_t)w * (n_idx + 1);
    if (k > 0 && k <= n) {
        if (tid == 0) { sh_prefix = 0u; sh_remaining = (unsigned)k; }
        for (int pass = 3; pass >= 0; --pass) {
            if (tid < 256) hist[tid] = 0u;
            __syncthreads();
            const unsigned prefix = sh_prefix;
            const unsigned hi_mask = pass == 3 ? 0u : (0xFFFFFFFFu << (8 * (pass + 1)));
            // audio levels share their leading bytes: the first passes would hammer one or two bins (a same-address LDS atomic
            // per sample measured 1.2 ms per pass).  Each thread therefore counts runs of equal digits in a register and
            // touches the histogram only when the digit changes; the low bytes are spread over the bins anyway.
            unsigned run_d = 0xFFFFFFFFu, run_n = 0u;
            for (int i = tid; i < n; i += LP_T) {
                const unsigned b = x[i] & 0x7FFFFFFFu;                 // bits of |x|: ordered like the values (no NaN in PCM)
                if ((b & hi_mask) == (prefix & hi_mask)) {
                    const unsigned d = (b >> (8 * pass)) & 255u;
                    if (d == run_d) ++run_n;
                    else {
                        if (run_n) atomicAdd(&hist[run_d], run_n);
                        run_d = d; run_n = 1u;
                    }
                }
            }
            if (run_n) atomicAdd(&hist[run_d], run_n);
            __syncthreads();
            if (tid == 0) {
                unsigned cum = 0, rem = sh_remaining;
                int d = 255;
                for (; d > 0; --d) {
                    if (cum + hist[d] >= rem) break;
                    cum += hist[d];
                }
                sh_remaining = rem - cum;                               // rank of the target inside bin d
                sh_prefix = prefix | ((unsigned)d << (8 * pass));
            }
            __syncthreads();
        }
        if (tid == 0) o[0] = __uint_as_float(sh_prefix);
    } else if (tid == 0) {
        o[0] = __builtin_nanf("");
    }
    const int32_t *ix = idx + (size_t)w * n_idx;
    for (int j = tid; j < n_idx; j += LP_T) {
        const int i = ix[j];
        o[1 + j] = (i >= 0 && i < n) ? __uint_as_float(x[i] & 0x7FFFFFFFu) : 0.f;
    }
}

}  // namespace

extern "C" int swx_loudness_probe(const float *d_pcm, int64_t pcm_stride, const int32_t *d_nk, const int32_t *d_idx, int n_idx,
                                  int W, float *d_out, void *stream)
{
    if (W <= 0) return 0;
    if (!d_pcm || !d_nk || !d_out || n_idx < 0 || (n_idx > 0 && !d_idx)) return -1;
    hipLaunchKernelGGL(loudness_probe_kernel, dim3(W), dim3(LP_T), 0, (hipStream_t)stream, d_pcm, pcm_stride, d_nk, d_idx, n_idx,
                       d_out);
    SWX_CHECK_LAUNCH();
    return 0;
}
