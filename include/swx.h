/* swx.h -- C ABI of libswx.so: the MI355X (gfx950) hot path of stable-ts.
 *
 * The reference (jianfch/stable-ts) has NO C ABI / FFI: its hot path is Python that calls the
 * un-vendored dependency openai-whisper (stable_whisper/whisper_compatibility.py:58-76).  The entry
 * points below are the native replacements for exactly the callables the reference's glue consumes
 * at that seam (seam B5); each one cites the reference call site it replaces.
 * INTEGRATION.md shows the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer named d_* is a DEVICE pointer (HBM); h_* is a host pointer
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream)
 *   - return value: 0 = ok, negative = error (swx_strerror); no call blocks the host unless it says so
 *   - no hidden device allocation: the caller binds the weight arena and the workspace
 *   - compute dtype: SWX_F16 (fp16 storage + MFMA, fp32 accumulate/LN/softmax -- what the reference runs on
 *     a GPU, original_whisper.py:251-260) or SWX_F32 (exact-f32 MFMA; the strict parity mode that matches the
 *     reference's CPU path, which is always fp32)
 */
#ifndef SWX_H
#define SWX_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SWX_F32 0
#define SWX_F16 1

#define SWX_N_SAMPLES 480000   /* whisper_compatibility.py:86  */
#define SWX_N_FRAMES 3000      /* whisper_compatibility.py:87  */
#define SWX_N_FFT 400
#define SWX_HOP 160

typedef struct swx_dims {      /* == whisper.model.ModelDimensions (read via model.dims.*, e.g. alignment.py:181) */
    int32_t n_mels, n_audio_ctx, n_audio_state, n_audio_head, n_audio_layer;
    int32_t n_vocab, n_text_ctx, n_text_state, n_text_head, n_text_layer;
} swx_dims;

typedef struct swx_model swx_model;   /* opaque */

const char *swx_strerror(int code);
int swx_version(void);

/* ---- model lifetime + weights (replaces whisper.load_model + model.to(device), original_whisper.py:995-1002) */
int swx_model_create(const swx_dims *dims, int dtype, swx_model **out);
void swx_model_destroy(swx_model *m);
/* size of the packed weight arena for this model/dtype */
size_t swx_weights_bytes(const swx_model *m);
int swx_bind_weights(swx_model *m, void *d_arena, size_t bytes);
/* a further handle (same dims and dtype) on the arena of `owner`, which stays the owner of the memory: the weights are
 * shared, nothing is cleared; each handle has its own workspace, alignment heads and stream, so host threads can drive
 * them concurrently (stream lanes) or with different head sets */
int swx_share_weights(swx_model *m, const swx_model *owner);
/* copy one checkpoint tensor (upstream state_dict key, fp32, device memory) into its packed slot,
 * converting/re-laying it out (QKV fusion, conv tap-major layout, fp16 cast) on the device */
int swx_load_tensor(swx_model *m, const char *name, const float *d_src, int64_t numel, void *stream);
/* 1 if every slot has been loaded; swx_missing_tensor writes the index-th missing name into buf (returns 0 at the end) */
int swx_weights_complete(const swx_model *m);
int swx_missing_tensor(const swx_model *m, int index, char *buf, int buflen);
/* the arena was filled from outside (rank 0's packed arena received over RCCL, parallel.py::broadcast_arena): mark every
 * tensor as present.  swx_weights_finalize: load-time preparation that depends on the complete set of tensors (the
 * LayerNorm-folded copies of the decoder projections the fused f16 decode step streams, csrc/swx_fold.hip); must be
 * called once after the last swx_load_tensor / after swx_weights_mark_loaded, before swx_decode. */
int swx_weights_mark_loaded(swx_model *m);
int swx_weights_finalize(swx_model *m, void *stream);
/* alignment heads (timing.py:105 reads model.alignment_heads.indices()): pairs (layer, head) */
int swx_set_alignment_heads(swx_model *m, const int32_t *h_layer_head_pairs, int n_pairs);
int swx_num_alignment_heads(const swx_model *m);

/* ---- workspace: max_windows = windows processed together by encode/score, max_rows = decoder sequences */
size_t swx_workspace_bytes(const swx_model *m, int max_windows, int max_rows);
int swx_bind_workspace(swx_model *m, void *d_ws, size_t bytes, int max_windows, int max_rows);

/* ---- a1: log-mel (replaces whisper.audio.log_mel_spectrogram at original_whisper.py:528-530, alignment.py:410-413)
 * d_pcm:  f32 [B][480000]  (caller zero-pads each <=30 s segment to exactly 480000 samples, as the reference does)
 * d_mel:  f32 [B][n_mels][3000]; the clamp floor uses the max over the WHOLE batch when per_item_max==0
 *         (upstream quirk inherited by refine) and per window when per_item_max!=0 */
int swx_log_mel(swx_model *m, const float *d_pcm, int B, float *d_mel, int per_item_max, void *stream);

/* log-mel of segments that are NOT zero-padded to 30 s, followed by pad_or_trim(mel, 3000): refine's inference
 * (alignment.py:660-661, padding 0) and locate (alignment.py:924-925, padding 201).
 * d_pcm f32 [B][480000] (only the first n_valid[b] samples of a row are read); n_valid / n_total: HOST int32 [B],
 * n_total = n_valid + the zero padding upstream appends (200 < n_total, n_total/160 <= 3008, n_valid <= 480000).
 * Frames t < n_total/160 are upstream's (reflection about n_total, clamp floor from the per-item max over all
 * n_total/160 frames, also those past 3000; over the whole batch when per_item_max==0, which is what upstream's
 * batched call in refine does); frames >= n_total/160 are 0.0.  Returns -2 on a length out of range. */
int swx_log_mel_ragged(swx_model *m, const float *d_pcm, const int32_t *n_valid, const int32_t *n_total, int B,
                       float *d_mel, int per_item_max, void *stream);

/* ---- a2: encoder (replaces model.encoder(mel), decode.py:27-30, timing.py:59-60)
 * d_mel f32 [B][n_mels][3000] -> d_xa [B][1500][d] in the compute dtype */
int swx_encode(swx_model *m, const float *d_mel, int B, void *d_xa, void *stream);

/* ---- cross-attention K/V of every decoder layer for B windows (upstream recomputes them lazily through the
 * kv-cache hooks; here they are produced once per window and shared by decode + scoring)
 * d_xkv: compute dtype [L][B][ K: 1500 x d row-major | V^T: d x 1536 (per head [64][1536], keys contiguous, zero padded) ]
 *        (the decode-step cross-attention streams both operands as coalesced 16-byte MFMA fragments) */
size_t swx_cross_kv_bytes(const swx_model *m, int B);
int swx_cross_kv(swx_model *m, const void *d_xa, int B, void *d_xkv, void *stream);

/* ---- a3/a4: decoding (replaces DecodingTaskStable._main_loop, decode.py:33-65, and the upstream logit filters +
 * GreedyDecoder / BeamSearchDecoder it drives).  One decode job = W windows x G sequences per window
 * (G = beam_size or best_of or 1), all advanced in lockstep on the device, no per-step host sync. */
typedef struct swx_decode_cfg {
    int32_t n_windows;            /* W */
    int32_t n_group;              /* G */
    int32_t beam;                 /* 0 = greedy/sampling (GreedyDecoder), 1 = beam search (BeamSearchDecoder) */
    float temperature;            /* 0 = argmax */
    float patience;               /* beam: max_candidates = round(G * patience); <=0 -> 1.0 */
    int32_t sample_len;           /* max sampled tokens (n_text_ctx/2 = 224 by default) */
    int32_t sample_begin;         /* len(initial_tokens) (same for every window of a job) */
    int32_t sot_index;            /* index of <|startoftranscript|> in the initial tokens */
    int32_t suppress_blank;       /* SuppressBlank */
    int32_t apply_timestamp_rules;/* ApplyTimestampRules (0 when without_timestamps) */
    int32_t max_initial_timestamp_index; /* -1 = None (stable-ts forces None, original_whisper.py:262-263) */
    int32_t eot, sot, no_timestamps, timestamp_begin, no_speech, blank_token; /* tokenizer ids (blank = encode(" ")[0]) */
    int32_t n_suppress;           /* SuppressTokens list length */
    int32_t min_tokens;           /* >0: EOT is suppressed until this many tokens were sampled (synthetic-weights
                                     benchmarking only; 0 = reference behaviour) */
    uint64_t seed;                /* sampling RNG seed (temperature > 0) */
    const int32_t *window_uid;    /* HOST array [W] or NULL: a stable identity of every window (e.g. its seek position).
                                     Sampling contract (temperature > 0): the draw of a sequence is a counter-based hash of
                                     (seed, window uid, slot within the window's group, step, token id) -- Gumbel-max over the
                                     filtered logits / T, i.e. a Categorical(logits / T) sample like upstream's GreedyDecoder,
                                     but NOT the reference's (framework Philox) random stream: a T > 0 retry is reproducible for a given (seed, uid)
                                     whatever the batch it is decoded in, and is not token-identical with the reference's.
                                     NULL: the window's index in this job is its uid. */
    const float *noise;           /* DEVICE array [sample_len][W * G][n_vocab] f32 or NULL.  Non-NULL (greedy decoder, temperature > 0):
                                     the draw of row r at step t is argmax_i(logits[i] / T - log(noise[t][r][i])): upstream's
                                     Categorical(logits / T).sample() is a multinomial draw of one sample, which the framework
                                     computes as argmax(p / q) with q ~ Exp(1).  When the caller fills the array from the
                                     framework's generator -- one exponential call per step on [W * G][n_vocab], as the
                                     reference's decoding loop makes them -- the sampled tokens are the reference's for the
                                     same seed; the counter-based hash above is not used.  HBM cost: sample_len * W * G * n_vocab
                                     * 4 bytes (232 MB for large-v3 at sample_len 224, best_of 5), read once per step. */
} swx_decode_cfg;

/* runs the whole loop; outputs (device):
 *  d_tokens_out  int32 [W][G_out][n_ctx+1]   final sequences incl. the initial tokens, eot-padded
 *  d_lens_out    int32 [W][G_out]            length up to (excluding) the first eot after sample_begin
 *  d_sumlp_out   f32   [W][G_out]            sum_logprobs of each candidate
 *  d_nospeech    f32   [W]                   softmax(logits[sot_index])[no_speech] at step 0 (decode.py:42-44)
 *  G_out = G (greedy/best-of) or max(G, max_candidates) (beam)
 * inputs: d_init_tokens int32 [W][sample_begin]; d_suppress int32 [n_suppress];
 *         d_ts_mask uint8 [W][1501] or NULL (decode.py:14-16,54); d_xkv from swx_cross_kv for these W windows.
 * Returns the number of steps executed (>=0) or a negative error.  Blocks until the loop has finished. */
int swx_decode(swx_model *m, const swx_decode_cfg *cfg, const int32_t *d_init_tokens, const int32_t *d_suppress,
               const uint8_t *d_ts_mask, const void *d_xkv, int32_t *d_tokens_out, int32_t *d_lens_out,
               float *d_sumlp_out, float *d_nospeech, void *stream);
int swx_decode_gout(const swx_decode_cfg *cfg);

/* ---- a6+a7: teacher-forced scoring pass + alignment matrix (replaces timing.py:41-67 _compute_qks and
 * timing.py:70-112 _compute_atten_weights + the head-mean/negation of timing.py:194-195)
 * For each window w: tokens d_tokens[w][0..n_tok[w]) = [*sot_sequence, no_timestamps, *text_tokens, eot].
 *  d_token_probs  f32 [W][max_n]      softmax(logits[n_sot-1+i? see DESIGN.md][:eot])[text_token_i], T values per window
 *  d_neg_matrix   f32 [W][max_n][1500] rows 0..T (T+1 rows), cols 0..n_frames[w): -(mean over alignment heads of
 *                 median7(znorm_tokens(softmax_frames(qk * qk_scale)))) -- the DTW input
 *  n_sot = len(sot_sequence); T[w] = n_tok[w] - n_sot - 2 */
int swx_score(swx_model *m, const int32_t *d_tokens, const int32_t *h_n_tok, int W, int max_n, int n_sot, int eot,
              const int32_t *h_n_frames, float qk_scale, int medfilt_width, const void *d_xkv,
              float *d_token_probs, float *d_neg_matrix, void *stream);

/* raw attention scores of the configured alignment heads from the same teacher-forced pass (what timing.py:41-67
 * _compute_qks hooks out of every cross-attention layer): a test / inspection hook since round 3 -- the head-selection variants
 * no longer materialise per-head scores (swx_score_q / swx_heads_* below).
 *  d_qk f32 [W][n_align][n_rows][1500]: (q * scale) . (k * scale) of token rows row0 .. row0+n_rows-1 (row0 + n_rows <= max_n),
 *       heads in the order of swx_set_alignment_heads sorted by (layer, head); pre-softmax, qk_scale not applied
 *  d_token_probs as in swx_score, or NULL */
int swx_score_qk(swx_model *m, const int32_t *d_tokens, const int32_t *h_n_tok, int W, int max_n, int n_sot, int eot,
                 int row0, int n_rows, const void *d_xkv, float *d_token_probs, float *d_qk, void *stream);

/* ---- head-selection variants of the alignment stage (reference: stable_whisper/timing.py:87-103 `dynamic_heads`,
 * :115-163 `aligner='new'`, :177-189 `extra_models`; csrc/swx_headsel.hip).  The reference hooks the scores of EVERY head out
 * of the pass ([L*H][tokens][1500] f32, 0.4-1.7 GB for large-v3) and runs tensor expressions over them; here the pass keeps
 * the cross-attention QUERIES of every layer and the kernels recompute a head's score row from q and the window's cross-K.
 *
 * swx_score_q: the teacher-forced pass of ONE window (timing.py:41-67).  d_q receives [n_text_layer][max_n][d] in the compute
 *   dtype (swx_qcap_bytes); d_token_probs as in swx_score (or NULL); d_xkv = the cross-K/V of that single window.
 * swx_heads_dynamic: rows row0 .. row0+n_rows-1; per row the `count` heads with the smallest sum_f |peak - f| / 1500 * p[f]
 *   (peak = the row's own argmax, or d_peaks[i] = the previous DTW pass's jump midpoint, f64); d_qk_sel f32
 *   [count][n_rows][ld_f] receives their RAW scaled scores = the input of swx_align_weights (H = count, N = n_rows).
 * swx_heads_new: over the n_tok rows: median filter -> * qk_scale -> softmax per head; score = w_colnorm * sum_f ||w[:, f]|| +
 *   w_rownorm * sum_i ||w[i, :]|| - w_coverage * penalty; the topk heads, column-normalised and averaged; d_neg_matrix f32
 *   [n_out][ld_f] receives MINUS that mean for rows row0 .. row0+n_out-1 (the DTW input).
 * swx_weighted_sum: d_out[e] = sum_j h_coef[j] * h_xs[j][e] (n_in <= 8 device arrays): pooling several models' head means.
 * d_scratch: swx_heads_scratch_bytes(m, max_n) of device memory; nothing is allocated inside. */
size_t swx_qcap_bytes(const swx_model *m, int max_n);
int swx_score_q(swx_model *m, const int32_t *d_tokens, const int32_t *h_n_tok, int max_n, int n_sot, int eot,
                const void *d_xkv, float *d_token_probs, void *d_q, void *stream);
size_t swx_heads_scratch_bytes(const swx_model *m, int max_n);
int swx_heads_dynamic(swx_model *m, const void *d_q, int max_n, int row0, int n_rows, const void *d_xkv, int n_frames,
                      float qk_scale, int count, const double *d_peaks, float *d_qk_sel, int ld_f, void *d_scratch,
                      size_t scratch_bytes, void *stream);
int swx_heads_new(swx_model *m, const void *d_q, int max_n, int n_tok, int row0, int n_out, const void *d_xkv, int n_frames,
                  float qk_scale, int medfilt_width, int topk, float w_colnorm, float w_rownorm, float w_coverage,
                  float *d_neg_matrix, int ld_f, void *d_scratch, size_t scratch_bytes, void *stream);
int swx_weighted_sum(const float *const *h_xs, const float *h_coef, int n_in, float *d_out, int64_t n, void *stream);

/* full-sequence logits of a teacher-forced pass (model(mel, tokens) as used by refine/locate; also a test hook):
 * d_logits f32 [W][max_n][n_vocab].  Language detection (model.detect_language, original_whisper.py:329) is this call
 * with tokens = [[sot]]. */
int swx_forward_logits(swx_model *m, const int32_t *d_tokens, const int32_t *h_n_tok, int W, int max_n,
                       const void *d_xkv, float *d_logits, void *stream);

/* ---- a7 stand-alone (test hook + extra_models path): weights f32 [W][H][N][ld_f] raw qk ->
 * neg_matrix f32 [W][N][1500].  d_scratch: swx_align_weights_scratch_bytes(W, H, N) bytes of device memory (no entry point of
 * this library allocates device memory) */
size_t swx_align_weights_scratch_bytes(int W, int H, int N);
int swx_align_weights(const float *d_qk, int W, int H, int N, int ld_f, const int32_t *h_n_frames, float qk_scale,
                      int medfilt_width, float *d_neg_matrix, void *d_scratch, size_t scratch_bytes, void *stream);

/* ---- whisper.timing.median_filter (timing.py:110,138): f32 [rows][n] -> [rows][n], reflect padding */
int swx_median_filter(const float *d_x, int64_t rows, int n, int width, float *d_out, void *stream);

/* ---- device half of the non-VAD silence analysis (stabilization/nonvad.py:16-39 audio2loudness), on the PCM that is resident
 * for the spectrogram.  d_pcm f32 [W] windows pcm_stride samples apart; d_nk int32 [W][2] = {valid samples n, k};
 * d_idx int32 [W][n_idx] sample indices; d_out f32 [W][1 + n_idx]: out[w][0] = the k-th largest |x| of the window (the value
 * of the reference's `topk(|x|, k).values[-1]`, nonvad.py:22, found by a radix select: an element of the input, no arithmetic; NaN when
 * k == 0), out[w][1 + j] = |x[idx[w][j]]| (0 for an index outside [0, n)).  The floating-point part of the analysis stays in
 * host code on these values (stable_ts_amd/stabilization.py::loudness_from_probe). */
int swx_loudness_probe(const float *d_pcm, int64_t pcm_stride, const int32_t *d_nk, const int32_t *d_idx, int n_idx, int W,
                       float *d_out, void *d_scratch, size_t scratch_bytes, void *stream);
/* d_scratch (>= swx_loudness_probe_scratch_bytes(W) bytes, or NULL): with it and fewer than 16 windows the selection runs spread over
 * the chip (a forced-alignment pass analyses one window per call); without it one workgroup per window.  The same element either way. */
size_t swx_loudness_probe_scratch_bytes(int W);

/* ---- f2 audio front-end: FLAC decoding on the HOST (no device work; csrc/swx_flac.hip).  The reference pipes every container
 * through an `ffmpeg -f s16le` child process (stable_whisper/audio/utils.py:63-125); offline boxes have no ffmpeg and the
 * reference's only real-speech fixture is test/jfk.flac, so native FLAC streams are decoded here: STREAMINFO + frames with
 * CONSTANT / VERBATIM / FIXED / LPC subframes, Rice / Rice2 residuals with escapes, wasted bits, left-side / side-right /
 * mid-side stereo, 4-32 bits per sample, up to 8 channels; CRC-8 and CRC-16 of every frame are verified.
 * swx_flac_probe: STREAMINFO only.  swx_flac_decode: h_out int32 [capacity_frames][channels] interleaved (sample values at the
 * stream's bit depth, not scaled), or NULL to count; returns the number of inter-channel sample frames decoded, or a negative
 * error (-20 not a FLAC stream, -21 corrupt stream / CRC mismatch, -22 unsupported stream, -23 truncated stream).  The MD5 in
 * `info` is the encoder's signature of the unencoded samples; the caller checks it (stable_ts_amd/audio_io.py::read_flac). */
typedef struct swx_flac_info {
    int32_t sample_rate, channels, bits_per_sample, min_block, max_block;
    int64_t total_samples;        /* per channel; 0 = unknown */
    uint8_t md5[16];
} swx_flac_info;
int swx_flac_probe(const uint8_t *h_data, size_t n_bytes, swx_flac_info *info);
int64_t swx_flac_decode(const uint8_t *h_data, size_t n_bytes, int32_t *h_out, int64_t capacity_frames, swx_flac_info *info);

/* ---- a8: DTW + backtrace (replaces whisper.timing.dtw at timing.py:195; CPU tie-break)
 * d_x f32 [W][ld_n][ld_m] (row-major; window w uses rows 0..N[w), cols 0..M[w));
 * outputs int32 [W][ld_n+ld_m] text/time indices in forward order and int32 [W] path lengths.
 * d_trace_ws: uint8 scratch of swx_dtw_workspace_bytes(W, ld_n, ld_m) */
size_t swx_dtw_workspace_bytes(int W, int ld_n, int ld_m);
int swx_dtw(const float *d_x, int W, int ld_n, int ld_m, const int32_t *d_N, const int32_t *d_M,
            int32_t *d_text_idx, int32_t *d_time_idx, int32_t *d_len, void *d_trace_ws, void *stream);

/* ---- measurement: per-kernel-class HIP-event timing on the launch stream (bench.py's roofline object).
 * classes: 0 gemm tiled (work=flops) 1 gemm skinny (bytes) 2 flash attention (flops) 3 rowwise attention (bytes)
 *          4 cached self-attention 5 select/beam 6 mel 7 align-weights 8 dtw 9 layernorm
 * swx_prof_collect: out[cls*3+{0,1,2}] = {launches, total ms, total algorithmic work}; returns the class count */
int swx_prof_enable(int on);
/* A/B switches for tests and scripts (SWX_FLAG_* in csrc/swx_kernels.h): 1 = decode steps and small passes through the general
 * per-op path instead of the fused "dec" step (its reference), 2048 = decode cross-attention on the row-layout K / V^T
 * (reference of the fragment-ordered copy), 8192 = memory-walking logit filters (reference of the register kernel),
 * 16384 = decode loop without the captured step graph, 32768 = decode step without the cache prefetch of the next projection's
 * weights, 65536 / 131072 = tiled GEMM never on the ring / the 256 x 256 kernel (bit-identical either way), 2097152 = the decode step's
 * K-split projection reduces its slabs inside the GEMM launch (arrival tickets; bit-identical, measured slower in round 5),
 * 67108864 = f16 flash attention on round 6's software-pipelined tile (bit-identical, measured slower).  Default 0; nothing reads an environment variable.
 * flags < 0 only queries.  Returns the previous value. */
int swx_debug_flags(int flags);
int swx_prof_collect(double *out, int n_classes);
/* how swx_decode ran on this handle so far: out[0] = two-step graphs captured, out[1] = graph replays (2 decode steps each),
 * out[2] = decode steps launched eagerly, out[3] = 1 if a capture / replay failed at least once on this handle (that job ran
 * eagerly; the handle keeps trying graphs for later jobs and switches replay off for good after the third failure) */
int swx_graph_stats(const swx_model *m, int64_t *out);

/* ---- building blocks exported for the parity tests (same kernels the calls above launch) */
int swx_test_gemm(int dtype, const void *d_a, int64_t lda, const void *d_w, const float *d_bias, const void *d_res,
                  void *d_c, int64_t ldc, int M, int N, int K, int epilogue, int force_kernel, void *stream);
/* which f16 kernel swx_test_gemm's launch would get (host-only, no GPU): 0 register-staged tiled, 1 skinny, 2 / 3 direct-to-LDS
 * with 128 / 64-column tiles, 4 / 5 the LDS-DMA ring at 64 / 128 columns, 6 the 256 x 256 two-stage kernel; < 0: not offered */
int swx_test_gemm_plan(int M, int N, int K, int epilogue, int force_kernel, int flags);
/* decode-step "dec" GEMM (csrc/swx_decstep.hip), f16: epilogue bits 1 = LayerNorm fold (gamma / beta given, A = raw rows, K = full
 * row), 2 = GELU, 4 = residual update of d_x in place, 8 = QKV scatter (columns >= d go to the caches at pos0[m]), 16 = K-split
 * allowed, 64 = a multi-token pass (from 161 rows on the launch takes the tall kernel: register-resident weights, 16-row tiles; with
 * 8, rows are 7 tokens per sequence; bit-identical to the launch without it).  d_scratch: >= N*K*2 + 8N + slab bytes + 1 KiB. */
int swx_test_dec_gemm(const void *d_a, int64_t lda, const void *d_w, const float *d_gamma, const float *d_beta,
                      const float *d_bias, void *d_c, int64_t ldc, void *d_x, void *d_kcache, void *d_vcache,
                      const int32_t *d_pos0, int n_ctx, int d, int M, int N, int K, int epilogue, void *d_scratch,
                      size_t scratch_bytes, void *stream);
int swx_test_layernorm(int dtype, const void *d_x, const float *d_g, const float *d_b, void *d_y, int rows, int d,
                       void *stream);
/* vt_kp > 0: d_v is transposed per batch item, [H][64][vt_kp] (keys contiguous, zero padded) -- the cross-KV layout */
int swx_test_attention(int dtype, const void *d_q, int64_t ldq, const void *d_k, const void *d_v, int64_t ldkv,
                       void *d_o, int64_t ldo, int B, int H, int nq, int nk, int force_kernel, int vt_kp, void *stream);

/* decode-step self-attention (f16): variant 0 = the single-token kernel without long-context code (positions < 128), 1 = the
 * single-token kernel, 2 = the general cached-attention kernel (their bit-identity reference).  d_q [R][d]; caches
 * [R_phys][n_ctx][d] with the new token's K / V already at position pos0[r] of row r; d_anc [R][n_ctx] or NULL; d_o [R][d]. */
int swx_test_self_attn_step(const void *d_q, void *d_kcache, void *d_vcache, const int32_t *d_anc, const int32_t *d_pos0,
                            int R, int H, int n_ctx, int d, int variant, void *d_o, void *stream);

/* multi-token self-attention of a teacher-forced pass (f16, no ancestor table, every row starts at position 0): mq = 0 the kernel
 * with one wave per (row, token, head) reading K / V from memory, mq = 1 several tokens of a (row, head) per workgroup with K / V
 * staged in LDS once (round 6; bit-identical).  d_q [R * n_new][d] (row stride d); caches [R][n_ctx][d] holding the tokens' K / V at
 * positions 0 .. n_new - 1; d_o [R * n_new][d].  Nothing in the reference corresponds to it. */
int swx_test_self_attn_multi(const void *d_q, void *d_kcache, void *d_vcache, int R, int H, int n_new, int n_ctx, int d, int mq,
                             void *d_o, void *stream);

/* the VALU lane-exchange helpers of csrc/swx_common.h (v_permlane16/32_swap, DPP) against __shfl_xor, on n_waves waves of 64
 * u32 values: d_out[((w * 13 + k) * 64) + lane], k = 0..5 lane_xor<32, 16, 8, 4, 2, 1>, k = 6..11 the __shfl_xor of the same
 * offsets, k = 12 a bit mask of the derived forms that agreed with their shuffle form (1 wave_sum_d, 2 / 4 lane_xor16_max / 32_max,
 * 8 / 16 lane_xor16_add / 32_add, 32 wave_max, 64 wave_sum: 127 = all).  Nothing in the reference corresponds to it. */
int swx_test_lane_xor(const uint32_t *d_in, uint32_t *d_out, int n_waves, void *stream);

/* csrc/swx_common.h::gelu_erf2 (the GELU of the tiled / dec GEMM epilogues: two values on packed f32 instructions, both sides of
 * erff's branch) against gelu_erf (the device library's erff) over ALL 2^32 f32 bit patterns: d_out[0] = values whose results differ
 * (NaN results with different payloads not counted), d_out[1] = the lowest such bit pattern (0xffffffff if none), d_out[2] = NaN
 * results whose payloads differ.  Nothing in the reference corresponds to it. */
int swx_test_gelu_pair(uint64_t *d_out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SWX_H */
