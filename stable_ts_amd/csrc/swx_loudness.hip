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

// ---- the same selection spread over the chip (round 6).  One workgroup per window is fine for a batch of windows (20 workgroups
// side by side) but a forced-alignment pass analyses ONE window per call: a single workgroup walks 480 000 samples four times,
// 108-410 us per window = 3 % of align()'s pass (profiles/r06_c7_align_kernels.csv).  Here every radix pass is a launch of NB
// workgroups per window (LDS histogram with the run-length trick above, non-empty bins flushed to the window's global histogram
// of that pass) followed by a one-workgroup pick of the digit; the last pick also gathers the index list.  A selection: the result
// is the same element of the input whatever the order of the counting.
constexpr int LPM_T = 256;

struct LpState { unsigned prefix, remaining; };

__global__ __launch_bounds__(LPM_T) void lp_hist_kernel(const float *__restrict__ pcm, int64_t pcm_stride, const int32_t *__restrict__ nk,
                                                        const LpState *__restrict__ st, unsigned *__restrict__ hist, int pass)
{
    __shared__ unsigned h[256];
    const int w = blockIdx.y, tid = threadIdx.x;
    const int n = nk[2 * w], k = nk[2 * w + 1];
    if (k <= 0 || k > n) return;
    const unsigned *x = (const unsigned *)(pcm + (size_t)w * pcm_stride);
    h[tid] = 0u;
    __syncthreads();
    const unsigned prefix = pass == 3 ? 0u : st[w].prefix;
    const unsigned hi_mask = pass == 3 ? 0u : (0xFFFFFFFFu << (8 * (pass + 1)));
    unsigned run_d = 0xFFFFFFFFu, run_n = 0u;
    for (int i = blockIdx.x * LPM_T + tid; i < n; i += gridDim.x * LPM_T) {
        const unsigned b = x[i] & 0x7FFFFFFFu;
        if ((b & hi_mask) == (prefix & hi_mask)) {
            const unsigned d = (b >> (8 * pass)) & 255u;
            if (d == run_d) ++run_n;
            else {
                if (run_n) atomicAdd(&h[run_d], run_n);
                run_d = d; run_n = 1u;
            }
        }
    }
    if (run_n) atomicAdd(&h[run_d], run_n);
    __syncthreads();
    if (h[tid]) atomicAdd(&hist[((size_t)w * 4 + pass) * 256 + tid], h[tid]);
}

// grid (W), 256 threads: the digit of this pass from the window's complete histogram; after the last pass the threshold and the gather
__global__ __launch_bounds__(LPM_T) void lp_pick_kernel(const float *__restrict__ pcm, int64_t pcm_stride, const int32_t *__restrict__ nk,
                                                        LpState *__restrict__ st, const unsigned *__restrict__ hist, int pass,
                                                        const int32_t *__restrict__ idx, int n_idx, float *__restrict__ out)
{
    const int w = blockIdx.x, tid = threadIdx.x;
    const int n = nk[2 * w], k = nk[2 * w + 1];
    const bool valid = k > 0 && k <= n;
    float *o = out + (size_t)w * (n_idx + 1);
    if (valid && tid == 0) {
        const unsigned *h = hist + ((size_t)w * 4 + pass) * 256;
        const unsigned prefix = pass == 3 ? 0u : st[w].prefix;
        unsigned cum = 0, rem = pass == 3 ? (unsigned)k : st[w].remaining;
        int d = 255;
        for (; d > 0; --d) {
            if (cum + h[d] >= rem) break;
            cum += h[d];
        }
        st[w].remaining = rem - cum;
        st[w].prefix = prefix | ((unsigned)d << (8 * pass));
        if (pass == 0) o[0] = __uint_as_float(st[w].prefix);
    }
    if (pass != 0) return;
    if (!valid && tid == 0) o[0] = __builtin_nanf("");
    const unsigned *x = (const unsigned *)(pcm + (size_t)w * pcm_stride);
    const int32_t *ix = idx + (size_t)w * n_idx;
    for (int j = tid; j < n_idx; j += LPM_T) {
        const int i = ix[j];
        o[1 + j] = (i >= 0 && i < n) ? __uint_as_float(x[i] & 0x7FFFFFFFu) : 0.f;
    }
}

}  // namespace

extern "C" size_t swx_loudness_probe_scratch_bytes(int W)
{
    return W > 0 ? (size_t)W * (4 * 256 * sizeof(unsigned) + sizeof(LpState)) : 0;
}

extern "C" int swx_loudness_probe(const float *d_pcm, int64_t pcm_stride, const int32_t *d_nk, const int32_t *d_idx, int n_idx,
                                  int W, float *d_out, void *d_scratch, size_t scratch_bytes, void *stream)
{
    if (W <= 0) return 0;
    if (!d_pcm || !d_nk || !d_out || n_idx < 0 || (n_idx > 0 && !d_idx)) return -1;
    hipStream_t s = (hipStream_t)stream;
    // few windows (forced alignment, the sequential window loop): the selection spread over the chip, nine short launches;
    // a batch of windows: one workgroup per window, one launch
    if (W < 16 && d_scratch && !(swx_flags() & SWX_FLAG_LOUDNESS_ONE_WG)) {
        if (scratch_bytes < swx_loudness_probe_scratch_bytes(W)) return -8;
        unsigned *hist = (unsigned *)d_scratch;
        LpState *st = (LpState *)(hist + (size_t)W * 4 * 256);
        { hipError_t e = hipMemsetAsync(hist, 0, (size_t)W * 4 * 256 * sizeof(unsigned), s); if (e != hipSuccess) return -100 - (int)e; }
        const int nb = W >= 8 ? 32 : 64;
        for (int pass = 3; pass >= 0; --pass) {
            hipLaunchKernelGGL(lp_hist_kernel, dim3(nb, W), dim3(LPM_T), 0, s, d_pcm, pcm_stride, d_nk, st, hist, pass);
            hipLaunchKernelGGL(lp_pick_kernel, dim3(W), dim3(LPM_T), 0, s, d_pcm, pcm_stride, d_nk, st, hist, pass, d_idx, n_idx, d_out);
        }
        SWX_CHECK_LAUNCH();
        return 0;
    }
    hipLaunchKernelGGL(loudness_probe_kernel, dim3(W), dim3(LP_T), 0, s, d_pcm, pcm_stride, d_nk, d_idx, n_idx, d_out);
    SWX_CHECK_LAUNCH();
    return 0;
}
