// swx_mel.hip -- log-mel spectrogram (a1) on gfx950.
//
// Replaces whisper.audio.log_mel_spectrogram as called at original_whisper.py:528-530 / alignment.py:410-413:
//   reflect-pad 200, hann(400) window, 400-point DFT every 160 samples (3001 frames, last dropped), |.|^2,
//   Slaney mel filterbank (n_mels x 201), log10(clamp(.,1e-10)), max(x, x.max()-8), (x+4)/4.
// The DFT is evaluated directly (201 bins x 400 taps) with f64 accumulation against an LDS-resident f64 twiddle
// table, so the result is the correctly rounded transform of the f32 windowed frame: closer to exact than any f32
// FFT, which keeps the clamp-floor region of the spectrum (8 decades below the peak) within parity tolerance.
// 0.96 GFLOP f64 per 30-s window; LDS-bound (one 16-B twiddle read per bin-tap shared by 8 frames).
//
// RAGGED variant (swx_log_mel_ragged): the callers that take the spectrogram of a segment that is NOT zero-padded to
// 30 s -- refine's inference (alignment.py:660-661: no padding) and locate (alignment.py:924-925: 201 zeros) -- get
// upstream's exact frames: per item {n_valid samples, total length L = n_valid + zero padding}, reflection about L,
// L/160 frames, the clamp floor from the max over THOSE frames (including frames >= 3000 that pad_or_trim cuts),
// and 0.0 (pad_or_trim's fill, after normalisation) in the frames past L/160.
#include "swx_common.h"
#include "swx_kernels.h"

#define MEL_FB 8          // frames per workgroup
#define MEL_NFFT 400
#define MEL_NBIN 201
#define MEL_HOP 160
#define MEL_NSAMP 480000
#define MEL_NFRAMES 3000

template <bool RAGGED>
__global__ __launch_bounds__(256) void swx_mel_power_kernel(const float *__restrict__ pcm, const float *__restrict__ hann,
                                                            const double2 *__restrict__ twiddle,
                                                            const float *__restrict__ filters, int n_mels,
                                                            float *__restrict__ out, unsigned *__restrict__ gmax,
                                                            const int2 *__restrict__ lens)
{
    __shared__ __attribute__((aligned(16))) double2 tw[MEL_NFFT];
    __shared__ float xw[MEL_FB][MEL_NFFT];
    __shared__ float pw[MEL_FB][MEL_NBIN + 7];
    __shared__ float red[4];

    const int b = blockIdx.y;
    const int t0 = blockIdx.x * MEL_FB;
    const int tid = threadIdx.x;
    const float *x = pcm + (size_t)b * MEL_NSAMP;
    int n_valid = MEL_NSAMP, n_total = MEL_NSAMP, n_frames = MEL_NFRAMES;
    if constexpr (RAGGED) {
        n_valid = lens[b].x;
        n_total = lens[b].y;
        n_frames = n_total / MEL_HOP;
        if (t0 >= n_frames) return;                      // whole block past the last frame: finish writes the 0.0 fill
    }

    for (int i = tid; i < MEL_NFFT; i += 256) tw[i] = twiddle[i];
    for (int i = tid; i < MEL_FB * MEL_NFFT; i += 256) {
        const int f = i / MEL_NFFT, n = i - f * MEL_NFFT;
        int s = (t0 + f) * MEL_HOP + n - MEL_NFFT / 2;
        if constexpr (RAGGED) {
            if (s < 0) s = -s;
            if (s >= n_total) s = 2 * (n_total - 1) - s;
            // frames >= n_frames of a partial block are computed on clamped garbage and dropped below
            xw[f][n] = (s >= 0 && s < n_valid) ? x[s] * hann[n] : 0.0f;
        } else {
            if (s < 0) s = -s;
            if (s >= MEL_NSAMP) s = 2 * (MEL_NSAMP - 1) - s;
            xw[f][n] = x[s] * hann[n];
        }
    }
    __syncthreads();

    if (tid < MEL_NBIN) {
        double re[MEL_FB], im[MEL_FB];
#pragma unroll
        for (int f = 0; f < MEL_FB; ++f) { re[f] = 0.0; im[f] = 0.0; }
        int idx = 0;
        for (int n = 0; n < MEL_NFFT; ++n) {
            const double2 c = tw[idx];
#pragma unroll
            for (int f = 0; f < MEL_FB; ++f) {
                const double v = (double)xw[f][n];
                re[f] = fma(v, c.x, re[f]);
                im[f] = fma(v, c.y, im[f]);
            }
            idx += tid;
            if (idx >= MEL_NFFT) idx -= MEL_NFFT;
        }
#pragma unroll
        for (int f = 0; f < MEL_FB; ++f) {
            const float r = (float)re[f], q = (float)im[f];
            const float mag = sqrtf(r * r + q * q);      // complex64 abs(), then ** 2 (upstream order)
            pw[f][tid] = mag * mag;
        }
    }
    __syncthreads();

    float lmax = -__builtin_inff();
    for (int i = tid; i < n_mels * MEL_FB; i += 256) {
        const int m = i / MEL_FB, f = i - m * MEL_FB;
        const float *fr = filters + (size_t)m * MEL_NBIN;
        double acc = 0.0;
        for (int k = 0; k < MEL_NBIN; ++k) acc += (double)fr[k] * (double)pw[f][k];
        const float v = log10f(fmaxf((float)acc, 1e-10f));
        if constexpr (RAGGED) {
            if (t0 + f >= n_frames) continue;
            if (t0 + f < MEL_NFRAMES) out[((size_t)b * n_mels + m) * MEL_NFRAMES + t0 + f] = v;
        } else {
            out[((size_t)b * n_mels + m) * MEL_NFRAMES + t0 + f] = v;
        }
        lmax = fmaxf(lmax, v);
    }
    lmax = wave_max(lmax);
    if ((tid & 63) == 0) red[tid >> 6] = lmax;
    __syncthreads();
    if (tid == 0) {
        const float mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        atomicMax(&gmax[b], f32_to_ordered(mx));
    }
}

template <bool RAGGED>
__global__ __launch_bounds__(256) void swx_mel_finish_kernel(float *__restrict__ out, const unsigned *__restrict__ gmax,
                                                             int B, int per_window, size_t per_item,
                                                             const int2 *__restrict__ lens)
{
    const int b = blockIdx.y;
    float mx;
    if (per_window) mx = ordered_to_f32(gmax[b]);
    else {
        mx = -__builtin_inff();
        for (int i = 0; i < B; ++i) mx = fmaxf(mx, ordered_to_f32(gmax[i]));
    }
    const float floor_v = mx - 8.0f;
    float *o = out + (size_t)b * per_item;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < per_item; i += (size_t)gridDim.x * 256) {
        if constexpr (RAGGED) {
            if ((int)(i % MEL_NFRAMES) >= lens[b].y / MEL_HOP) { o[i] = 0.0f; continue; }
        }
        const float v = fmaxf(o[i], floor_v);
        o[i] = (v + 4.0f) / 4.0f;
    }
}

int swx_mel_launch(const float *d_pcm, int B, const float *d_hann, const double2 *d_twiddle, const float *d_filters,
                   int n_mels, float *d_mel, unsigned *d_gmax, int per_item_max, hipStream_t s)
{
    if (B <= 0) return 0;
    SwxProfScope prof(PC_MEL, (double)B * (480000.0 * 4 + (double)n_mels * 3000 * 4), s);
    hipError_t e = hipMemsetAsync(d_gmax, 0, sizeof(unsigned) * B, s);   // 0 < ordered(-inf)
    if (e != hipSuccess) return -100 - (int)e;
    hipLaunchKernelGGL(swx_mel_power_kernel<false>, dim3(MEL_NFRAMES / MEL_FB, B), dim3(256), 0, s, d_pcm, d_hann,
                       d_twiddle, d_filters, n_mels, d_mel, d_gmax, (const int2 *)nullptr);
    hipLaunchKernelGGL(swx_mel_finish_kernel<false>, dim3(64, B), dim3(256), 0, s, d_mel, d_gmax, B, per_item_max,
                       (size_t)n_mels * MEL_NFRAMES, (const int2 *)nullptr);
    SWX_CHECK_LAUNCH();
    return 0;
}

// d_lens: int32 [B][2] = {n_valid, n_total}; 200 < n_total, n_valid <= min(n_total, 480000), n_total/160 <= 3008
// (checked by the caller, swx_log_mel_ragged).  One extra block row covers the frames 3000..3007 that only feed the max.
int swx_mel_ragged_launch(const float *d_pcm, const int *d_lens, int B, const float *d_hann, const double2 *d_twiddle,
                          const float *d_filters, int n_mels, float *d_mel, unsigned *d_gmax, int per_item_max,
                          hipStream_t s)
{
    if (B <= 0) return 0;
    SwxProfScope prof(PC_MEL, (double)B * (480000.0 * 4 + (double)n_mels * 3000 * 4), s);
    hipError_t e = hipMemsetAsync(d_gmax, 0, sizeof(unsigned) * B, s);
    if (e != hipSuccess) return -100 - (int)e;
    hipLaunchKernelGGL(swx_mel_power_kernel<true>, dim3(MEL_NFRAMES / MEL_FB + 1, B), dim3(256), 0, s, d_pcm, d_hann,
                       d_twiddle, d_filters, n_mels, d_mel, d_gmax, (const int2 *)d_lens);
    hipLaunchKernelGGL(swx_mel_finish_kernel<true>, dim3(64, B), dim3(256), 0, s, d_mel, d_gmax, B, per_item_max,
                       (size_t)n_mels * MEL_NFRAMES, (const int2 *)d_lens);
    SWX_CHECK_LAUNCH();
    return 0;
}
