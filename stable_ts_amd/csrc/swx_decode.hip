// swx_decode.hip -- on-device token selection for the decoding loop (a4).
//
// Replaces, per step and without a host round trip, what stable_whisper/decode.py:42-58 does on the host:
//   * no-speech probability at the SOT position (decode.py:42-44)
//   * upstream logit filters SuppressBlank / SuppressTokens / ApplyTimestampRules (decode.py:50-51)
//   * stable-ts's timestamp suppression by silence mask (decode.py:14-16,54) and nan_to_num_(-inf) (decode.py:56)
//   * GreedyDecoder.update (argmax / categorical sampling, log-softmax accumulation, EOT propagation) or
//     BeamSearchDecoder.update (top-(G+1) per beam, ranking, finished-sequence bookkeeping, KV "rearrangement" --
//     here an ancestor-table gather instead of a copy of the KV cache) (decode.py:58)
// One workgroup per sequence row over the f32 logits row (207 KB for large-v3, L2 resident), a handful of streaming
// passes with block reductions.  Integer bookkeeping is bit-exact by construction.
#include "swx_common.h"
#include "swx_kernels.h"
#include "swx_decode.h"

namespace {

constexpr int SEL_T = 1024;   // threads per row
constexpr float NEG_INF = -__builtin_inff();

struct ArgMax { float v; int i; };

__device__ __forceinline__ ArgMax argmax_combine(ArgMax a, ArgMax b)
{
    // larger value wins; ties -> smaller index (torch.argmax returns the first maximal element)
    if (b.v > a.v || (b.v == a.v && b.i < a.i)) return b;
    return a;
}

// Block reductions of the 1 024-thread selection kernels (16 waves): a wave butterfly, one LDS slot per wave, then the 16 wave
// results.  Rounds 1-5 had EVERY thread walk the 16 slots one after the other (16 LDS reads + 16 combines, 11 times per row:
// a third of the kernel's instructions in beam mode); now lanes take one slot each (slot = lane % 16) and run a 4-step butterfly
// inside their row of 16 lanes (DPP).  Bit-identical: max is exact and order-free (-inf start values, NaN inputs are dropped by
// v_max_f32 whichever side they are on); the arg-max order (value descending, index ascending) is total on the values it sees
// (the filters have replaced NaN), so every combination tree has the same winner; the SUM keeps its order -- slots 0 .. 15 added
// one after the other into 0 -- and only reads them as four 16-byte LDS loads.  -DSWX_SELECT_R5_REDUCE (A/B build) = the old walk.
static_assert(SEL_T == 1024, "block reductions: 16 wave slots");

__device__ ArgMax block_argmax(ArgMax x, ArgMax *sh)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // butterfly 32 .. 1 with own-vs-partner operand order as before; the exchange runs on the VALU (swx_common.h: lane_xor)
#define SWX_ARGMAX_STEP(O) { ArgMax y; y.v = lane_xor<O>(x.v, lane); y.i = lane_xor<O>(x.i, lane); x = argmax_combine(x, y); }
    SWX_ARGMAX_STEP(32) SWX_ARGMAX_STEP(16) SWX_ARGMAX_STEP(8) SWX_ARGMAX_STEP(4) SWX_ARGMAX_STEP(2) SWX_ARGMAX_STEP(1)
    __syncthreads();
    if (lane == 0) sh[wave] = x;
    __syncthreads();
#ifdef SWX_SELECT_R5_REDUCE
    ArgMax r = sh[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r = argmax_combine(r, sh[w]);
    return r;
#else
    x = sh[lane & 15];
    SWX_ARGMAX_STEP(8) SWX_ARGMAX_STEP(4) SWX_ARGMAX_STEP(2) SWX_ARGMAX_STEP(1)
    return x;
#endif
#undef SWX_ARGMAX_STEP
}

__device__ float block_max(float x, float *sh)
{
    x = wave_max(x);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) sh[wave] = x;
    __syncthreads();
#if defined(SWX_SELECT_R5_REDUCE) || defined(SWX_SELECT_MAX_WALK)
    float r = sh[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r = fmaxf(r, sh[w]);
    return r;
#else
    // (four 16-byte LDS reads, one at a time: the lane butterfly of block_argmax costs this kernel's callers -- two maxima live
    //  at once at the 128-register cap -- a spilled register)
    const f32x4 *s4 = (const f32x4 *)sh;
    float r = -__builtin_inff();
#pragma unroll
    for (int q = 0; q < 4; ++q) { const f32x4 v = s4[q]; r = fmaxf(r, fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]))); }
    return r;
#endif
}

__device__ float block_sum(float x, float *sh)
{
    x = wave_sum(x);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) sh[wave] = x;
    __syncthreads();
    float r = 0.f;
#if defined(SWX_SELECT_R5_REDUCE) || defined(SWX_SELECT_SUM_WALK)
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) r += sh[w];
#else
    const f32x4 *s4 = (const f32x4 *)sh;         // (16-byte aligned by its declarations)
#pragma unroll
    for (int q = 0; q < 4; ++q) { const f32x4 v = s4[q]; r += v[0]; r += v[1]; r += v[2]; r += v[3]; }
#endif
    return r;
}

__device__ __forceinline__ unsigned hash32(unsigned x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

__device__ __forceinline__ float gumbel(uint64_t seed, unsigned row_uid, unsigned step, unsigned idx)
{
    unsigned h = hash32((unsigned)seed ^ hash32(idx + 0x9e3779b9u * (step + 1u)));
    h = hash32(h ^ hash32(row_uid * 0x85ebca6bu + (unsigned)(seed >> 32)));
    const float u = ((float)(h >> 8) + 0.5f) * (1.0f / 16777216.0f);   // (0,1)
    return -logf(-logf(u));
}

// ---------------------------------------------------------------------------------------------------- init
__global__ void decode_init_kernel(DecodeBufs b, const int32_t *__restrict__ init_tokens)
{
    // grid (M): tokens = [init | eot ...], pos0 = 0, sum_logprobs = 0, flags cleared
    const int r = blockIdx.x;
    const int w = r / b.G;
    int32_t *row0 = b.tokens[0] + (size_t)r * b.TS;
    int32_t *row1 = b.tokens[1] + (size_t)r * b.TS;
    for (int i = threadIdx.x; i < b.TS; i += blockDim.x) {
        const int32_t t = (i < b.n_init) ? init_tokens[(size_t)w * b.n_init + i] : b.cfg.eot;
        row0[i] = t;
        row1[i] = t;
    }
    if (threadIdx.x == 0) {
        b.pos0[r] = 0;
        b.sum_lp[r] = 0.f;
        b.row_done[r] = 0;
        if (r % b.G == 0) { b.win_done[w] = 0; b.fin_count[w] = 0; }
        if (r == 0) { *b.n_done = 0; *b.step_dev = 0; }
    }
}

__global__ void decode_after_prefill_kernel(DecodeBufs b)
{
    // grid (M): every row of window w reads the prefill K/V of physical row w*G; first new position = n_init
    const int r = blockIdx.x;
    const int w = r / b.G;
    if (b.anc[0]) {
        for (int p = threadIdx.x; p < b.n_ctx; p += blockDim.x) {
            const int v = (p < b.n_init) ? w * b.G : r;
            b.anc[0][(size_t)r * b.n_ctx + p] = v;
            b.anc[1][(size_t)r * b.n_ctx + p] = v;
        }
    }
    if (threadIdx.x == 0) b.pos0[r] = b.n_init - 1;   // the select kernel advances it to n_init when it appends
}

// replicate the W prefill logits rows to the M sequence rows; no-speech probability from the SOT-position logits
__global__ __launch_bounds__(SEL_T) void decode_prefill_logits_kernel(DecodeBufs b, const float *__restrict__ lg2,
                                                                      float *__restrict__ nospeech)
{
    // lg2: [W][2][V] (row 0: position sot_index, row 1: last initial position); grid (M)
    __shared__ __attribute__((aligned(16))) float sh[SEL_T / 64];
    const int r = blockIdx.x, w = r / b.G, V = b.V;
    const float *last = lg2 + ((size_t)w * 2 + 1) * V;
    float *dst = b.logits + (size_t)r * V;
    for (int i = threadIdx.x; i < V; i += SEL_T) dst[i] = last[i];
    if (r % b.G == 0 && nospeech) {
        const float *sot = lg2 + ((size_t)w * 2) * V;
        float mx = NEG_INF;
        for (int i = threadIdx.x; i < V; i += SEL_T) mx = fmaxf(mx, sot[i]);
        mx = block_max(mx, sh);
        float sum = 0.f;
        for (int i = threadIdx.x; i < V; i += SEL_T) sum += expf(sot[i] - mx);
        sum = block_sum(sum, sh);
        if (threadIdx.x == 0) nospeech[w] = (b.cfg.no_speech >= 0) ? expf(sot[b.cfg.no_speech] - mx) / sum : NAN;
    }
}

// -------------------------------------------------------------------------------------------- filter + select
// grid (M).  step = number of tokens sampled so far (== tokens.shape[1] - sample_begin upstream).
__global__ __launch_bounds__(SEL_T) void decode_select_kernel(DecodeBufs b, int cur)
{
    const int step = *b.step_dev;
    __shared__ __attribute__((aligned(16))) float shf[SEL_T / 64];
    __shared__ __attribute__((aligned(16))) ArgMax sha[SEL_T / 64];
    __shared__ int sh_last_ts;
    const int r = blockIdx.x, w = r / b.G, V = b.V, tid = threadIdx.x;
    const swx_decode_cfg &c = b.cfg;
    if (b.win_done[w]) return;    // frozen: upstream stopped iterating for this audio
    float *lg = b.logits + (size_t)r * V;
    int32_t *tok = b.tokens[cur] + (size_t)r * b.TS;
    const int len = b.n_init + step;            // tokens so far
    const int nsamp = len - c.sample_begin;     // sampled so far
    const int tsb = c.timestamp_begin;

    // ---- SuppressBlank / SuppressTokens / min_tokens
    if (c.suppress_blank && nsamp == 0 && tid == 0) { if (c.blank_token >= 0) lg[c.blank_token] = NEG_INF; lg[c.eot] = NEG_INF; }
    for (int i = tid; i < c.n_suppress; i += SEL_T) { const int t = b.suppress[i]; if ((unsigned)t < (unsigned)V) lg[t] = NEG_INF; }   // ids are range-checked on the host too
    if (c.min_tokens > 0 && nsamp < c.min_tokens && tid == 0) lg[c.eot] = NEG_INF;
    __syncthreads();

    // ---- ApplyTimestampRules
    if (c.apply_timestamp_rules) {
        if (tid == 0) { sh_last_ts = -1; if (c.no_timestamps >= 0) lg[c.no_timestamps] = NEG_INF; }
        __syncthreads();
        // last timestamp token among the sampled tokens (they never decrease, but take the LAST occurrence as upstream)
        int my = -1;
        for (int i = c.sample_begin + tid; i < len; i += SEL_T) if (tok[i] >= tsb) my = i;
        if (my >= 0) atomicMax(&sh_last_ts, my);
        __syncthreads();
        const int last_ts_pos = sh_last_ts;
        const bool last_was_ts = nsamp >= 1 && tok[len - 1] >= tsb;
        const bool penult_was_ts = nsamp < 2 || tok[len - 2] >= tsb;
        int lo0 = 0, hi0 = 0;      // [lo0, hi0) masked by the pairing rule
        if (last_was_ts) { if (penult_was_ts) { lo0 = tsb; hi0 = V; } else { lo0 = 0; hi0 = c.eot; } }
        int hi1 = tsb;             // [tsb, hi1) masked by monotonicity
        if (last_ts_pos >= 0) {
            const int tl = tok[last_ts_pos];
            hi1 = (last_was_ts && !penult_was_ts) ? tl : tl + 1;
        }
        int hi2 = 0;               // [0, hi2) masked at the first step
        int lo3 = V;               // [lo3, V) masked by max_initial_timestamp
        if (nsamp == 0) { hi2 = tsb; if (c.max_initial_timestamp_index >= 0) lo3 = tsb + c.max_initial_timestamp_index + 1; }
        float mx_text = NEG_INF, mx_ts = NEG_INF;
        for (int i = tid; i < V; i += SEL_T) {
            float v = lg[i];
            if ((i >= lo0 && i < hi0) || (i >= tsb && i < hi1) || (i < hi2) || (i >= lo3)) { v = NEG_INF; lg[i] = v; }
            if (i < tsb) mx_text = fmaxf(mx_text, v); else mx_ts = fmaxf(mx_ts, v);
        }
        mx_text = block_max(mx_text, shf);
        mx_ts = block_max(mx_ts, shf);
        // logsumexp over the timestamp range vs. max text logit (the common log-softmax normaliser cancels)
        float s_ts = 0.f;
        if (mx_ts > NEG_INF)
            for (int i = tsb + tid; i < V; i += SEL_T) s_ts += expf(lg[i] - mx_ts);
        s_ts = block_sum(s_ts, shf);
        const float lse_ts = (mx_ts > NEG_INF) ? mx_ts + logf(s_ts) : NEG_INF;
        if (lse_ts > mx_text)
            for (int i = tid; i < tsb; i += SEL_T) lg[i] = NEG_INF;
        __syncthreads();
    }

    // ---- stable-ts: silence-masked timestamp tokens (decode.py:54), then nan_to_num_(-inf) (decode.py:56):
    //      nan -> -inf, -inf -> lowest finite, +inf -> largest finite
    const uint8_t *tmask = b.ts_mask ? b.ts_mask + (size_t)w * 1501 : nullptr;
    float mx = NEG_INF;
    for (int i = tid; i < V; i += SEL_T) {
        float v = lg[i];
        if (tmask && i >= tsb && i - tsb < 1501 && tmask[i - tsb]) v = NEG_INF;
        if (v != v) v = NEG_INF;
        else if (v == NEG_INF) v = -3.4028234663852886e38f;
        else if (v == -NEG_INF) v = 3.4028234663852886e38f;
        lg[i] = v;
        mx = fmaxf(mx, v);
    }
    mx = block_max(mx, shf);
    float se = 0.f;
    for (int i = tid; i < V; i += SEL_T) se += expf(lg[i] - mx);
    se = block_sum(se, shf);
    const float lse = logf(se);                  // logprob(i) = (lg[i] - mx) - lse

    if (!c.beam) {
        // ---- GreedyDecoder.update
        ArgMax best; best.v = NEG_INF; best.i = 0x7fffffff;
        if (c.temperature == 0.f) {
            for (int i = tid; i < V; i += SEL_T) { ArgMax x; x.v = lg[i]; x.i = i; best = argmax_combine(best, x); }
        } else {
            const float invT = 1.0f / c.temperature;
            // keyed on the window's stable identity, not on its row in this batch (fallback retries are decoded in
            // whatever batch the pending windows form)
            const unsigned row_uid = b.win_uid ? (unsigned)b.win_uid[r / b.G] * 0x9E3779B1u + (unsigned)(r % b.G) : (unsigned)r;
            // c.noise: the caller's Exp(1) variates of this step and row (the framework generator's stream: swx.h)
            const float *qn = c.noise ? c.noise + ((size_t)step * b.M + r) * V : nullptr;
            for (int i = tid; i < V; i += SEL_T) {
                ArgMax x; x.v = lg[i] * invT + (qn ? -logf(qn[i]) : gumbel(c.seed, row_uid, (unsigned)step, (unsigned)i)); x.i = i;
                best = argmax_combine(best, x);
            }
        }
        best = block_argmax(best, sha);
        if (tid == 0) {
            int next = best.i;
            const bool prev_eot = tok[len - 1] == c.eot;
            const float lp = (lg[next] - mx) - lse;
            if (!prev_eot) b.sum_lp[r] += lp; else next = c.eot;
            tok[len] = next;
            b.pos0[r] = len;                      // position of the token the next forward pass embeds
            b.row_done[r] = (next == c.eot) ? 1 : 0;
        }
    } else {
        // ---- BeamSearchDecoder.update, step 1: top-(G+1) log-probs of this beam
        const int K = b.G + 1;
        for (int k = 0; k < K; ++k) {
            ArgMax best; best.v = NEG_INF; best.i = 0x7fffffff;
            for (int i = tid; i < V; i += SEL_T) { ArgMax x; x.v = lg[i]; x.i = i; best = argmax_combine(best, x); }
            best = block_argmax(best, sha);
            if (tid == 0) {
                b.cand_lp[(size_t)r * K + k] = (best.v - mx) - lse;
                b.cand_tok[(size_t)r * K + k] = best.i;
                lg[best.i] = NEG_INF;             // exclude from the next round (row is rewritten by the next step)
            }
            __syncthreads();
        }
    }
}

// The same filter + select with the row held in REGISTERS (51 logits per thread at V = 51 866): the kernel above walks the
// row 11-16 times through L2 (three timestamp-rule passes, the nan pass, the exp pass, G + 1 arg-max rounds) and takes 105 us
// per decode step at 100 rows; here the row is read once.  Bit-identical by construction: every thread owns the elements
// i = tid + k * SEL_T the loops above give it, sums are accumulated in ascending k, and the one sum whose loop above starts
// at timestamp_begin (a rotated element-to-thread assignment) is rotated back through LDS before the block reduction.
// Point suppressions (blank / eot / suppress list / no_timestamps) are still written to the row in memory first, because a
// register array cannot be indexed by a run-time token id; nothing reads the row after this kernel (the next step's logits
// GEMM rewrites it), so the filtered values are not written back.
template <int NV>
__global__ __launch_bounds__(SEL_T) void decode_select_reg_kernel(DecodeBufs b, int cur)
{
    const int step = *b.step_dev;
    __shared__ __attribute__((aligned(16))) float shf[SEL_T / 64];
    __shared__ __attribute__((aligned(16))) ArgMax sha[SEL_T / 64];
    __shared__ float sh_rot[SEL_T];
    __shared__ int sh_last_ts;
    __shared__ float sh_raw;
    const int r = blockIdx.x, w = r / b.G, V = b.V, tid = threadIdx.x;
    const swx_decode_cfg &c = b.cfg;
    if (b.win_done[w]) return;
    float *lg = b.logits + (size_t)r * V;
    int32_t *tok = b.tokens[cur] + (size_t)r * b.TS;
    const int len = b.n_init + step;
    const int nsamp = len - c.sample_begin;
    const int tsb = c.timestamp_begin;

    if (c.suppress_blank && nsamp == 0 && tid == 0) { if (c.blank_token >= 0) lg[c.blank_token] = NEG_INF; lg[c.eot] = NEG_INF; }
    for (int i = tid; i < c.n_suppress; i += SEL_T) { const int t = b.suppress[i]; if ((unsigned)t < (unsigned)V) lg[t] = NEG_INF; }
    if (c.min_tokens > 0 && nsamp < c.min_tokens && tid == 0) lg[c.eot] = NEG_INF;
    if (c.apply_timestamp_rules && tid == 0) { sh_last_ts = -1; if (c.no_timestamps >= 0) lg[c.no_timestamps] = NEG_INF; }
    __syncthreads();

    float v[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) { const int i = tid + k * SEL_T; v[k] = i < V ? lg[i] : NEG_INF; }

    if (c.apply_timestamp_rules) {
        int my = -1;
        for (int i = c.sample_begin + tid; i < len; i += SEL_T) if (tok[i] >= tsb) my = i;
        if (my >= 0) atomicMax(&sh_last_ts, my);
        __syncthreads();
        const int last_ts_pos = sh_last_ts;
        const bool last_was_ts = nsamp >= 1 && tok[len - 1] >= tsb;
        const bool penult_was_ts = nsamp < 2 || tok[len - 2] >= tsb;
        int lo0 = 0, hi0 = 0;
        if (last_was_ts) { if (penult_was_ts) { lo0 = tsb; hi0 = V; } else { lo0 = 0; hi0 = c.eot; } }
        int hi1 = tsb;
        if (last_ts_pos >= 0) {
            const int tl = tok[last_ts_pos];
            hi1 = (last_was_ts && !penult_was_ts) ? tl : tl + 1;
        }
        int hi2 = 0, lo3 = V;
        if (nsamp == 0) { hi2 = tsb; if (c.max_initial_timestamp_index >= 0) lo3 = tsb + c.max_initial_timestamp_index + 1; }
        float mx_text = NEG_INF, mx_ts = NEG_INF;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int i = tid + k * SEL_T;
            if (i < V) {
                if ((i >= lo0 && i < hi0) || (i >= tsb && i < hi1) || (i < hi2) || (i >= lo3)) v[k] = NEG_INF;
                if (i < tsb) mx_text = fmaxf(mx_text, v[k]); else mx_ts = fmaxf(mx_ts, v[k]);
            }
        }
        mx_text = block_max(mx_text, shf);
        mx_ts = block_max(mx_ts, shf);
        // the loop above this kernel gives element i >= tsb to thread (i - tsb) % SEL_T: accumulate per owner here (same
        // ascending order), then hand each partial sum to the thread that would have held it
        float s_own = 0.f;
        if (mx_ts > NEG_INF) {
#pragma unroll
            for (int k = 0; k < NV; ++k) { const int i = tid + k * SEL_T; if (i >= tsb && i < V) s_own += expf(v[k] - mx_ts); }
        }
        sh_rot[(tid - tsb) & (SEL_T - 1)] = s_own;
        __syncthreads();
        float s_ts = block_sum(sh_rot[tid], shf);
        const float lse_ts = (mx_ts > NEG_INF) ? mx_ts + logf(s_ts) : NEG_INF;
        if (lse_ts > mx_text) {
#pragma unroll
            for (int k = 0; k < NV; ++k) if (tid + k * SEL_T < tsb) v[k] = NEG_INF;
        }
    }

    const uint8_t *tmask = b.ts_mask ? b.ts_mask + (size_t)w * 1501 : nullptr;
    float mx = NEG_INF;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int i = tid + k * SEL_T;
        if (i < V) {
            float x = v[k];
            if (tmask && i >= tsb && i - tsb < 1501 && tmask[i - tsb]) x = NEG_INF;
            if (x != x) x = NEG_INF;
            else if (x == NEG_INF) x = -3.4028234663852886e38f;
            else if (x == -NEG_INF) x = 3.4028234663852886e38f;
            v[k] = x;
            mx = fmaxf(mx, x);
        }
    }
    mx = block_max(mx, shf);
    float se = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) if (tid + k * SEL_T < V) se += expf(v[k] - mx);
    se = block_sum(se, shf);
    const float lse = logf(se);

    if (!c.beam) {
        ArgMax best; best.v = NEG_INF; best.i = 0x7fffffff;
        float raw_own = 0.f;                     // filtered logit at this thread's own best index
        if (c.temperature == 0.f) {
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const int i = tid + k * SEL_T;
                if (i < V) { ArgMax x; x.v = v[k]; x.i = i; const ArgMax nb = argmax_combine(best, x); if (nb.i != best.i) raw_own = v[k]; best = nb; }
            }
        } else {
            const float invT = 1.0f / c.temperature;
            const unsigned row_uid = b.win_uid ? (unsigned)b.win_uid[r / b.G] * 0x9E3779B1u + (unsigned)(r % b.G) : (unsigned)r;
            const float *qn = c.noise ? c.noise + ((size_t)step * b.M + r) * V : nullptr;
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const int i = tid + k * SEL_T;
                if (i < V) {
                    ArgMax x; x.v = v[k] * invT + (qn ? -logf(qn[i]) : gumbel(c.seed, row_uid, (unsigned)step, (unsigned)i)); x.i = i;
                    const ArgMax nb = argmax_combine(best, x);
                    if (nb.i != best.i) raw_own = v[k];
                    best = nb;
                }
            }
        }
        const int own_i = best.i;
        best = block_argmax(best, sha);
        if (own_i == best.i) sh_raw = raw_own;
        __syncthreads();
        if (tid == 0) {
            int next = best.i;
            const bool prev_eot = tok[len - 1] == c.eot;
            const float lp = (sh_raw - mx) - lse;
            if (!prev_eot) b.sum_lp[r] += lp; else next = c.eot;
            tok[len] = next;
            b.pos0[r] = len;
            b.row_done[r] = (next == c.eot) ? 1 : 0;
        }
    } else {
        // top-(G + 1) of the row: every thread keeps the best of its own 51 values; a round is one block arg-max over those,
        // and only the thread that owned the winner strikes it out and rescans its 51 (the other 15 waves skip the branch).
        // Rescanning every thread's values in every round -- 6 x 51 compare / select pairs per thread -- was more than half of
        // this kernel's VALU work.  Same selections: ties go to the smaller index inside a thread and between threads.
        const int K = b.G + 1;
        auto local_best = [&]() {
            ArgMax best; best.v = NEG_INF; best.i = 0x7fffffff;
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const int i = tid + k * SEL_T;
                if (i < V) { ArgMax x; x.v = v[k]; x.i = i; best = argmax_combine(best, x); }
            }
            return best;
        };
        ArgMax mine = local_best();
        for (int kk = 0; kk < K; ++kk) {
            const ArgMax best = block_argmax(mine, sha);
            if (tid == 0) {
                b.cand_lp[(size_t)r * K + kk] = (best.v - mx) - lse;
                b.cand_tok[(size_t)r * K + kk] = best.i;
            }
            if ((best.i & (SEL_T - 1)) == tid) {
#pragma unroll
                for (int k = 0; k < NV; ++k) if (tid + k * SEL_T == best.i) v[k] = NEG_INF;
                mine = local_best();
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------- beam update
// grid (W), 256 threads.  cur = token/ancestor buffer holding the current beams; the new beams go to cur^1.
__global__ __launch_bounds__(256) void decode_beam_update_kernel(DecodeBufs b, int cur)
{
    const int step = *b.step_dev;
    constexpr int MAXC = 17 * 16;
    __shared__ float sc[MAXC];
    __shared__ short src[MAXC];
    __shared__ int tk[MAXC];
    __shared__ short order[MAXC];
    __shared__ int sel_src[16], sel_tok[16];
    __shared__ float sel_sc[16];
    const int w = blockIdx.x, G = b.G, K = G + 1, tid = threadIdx.x;
    const swx_decode_cfg &c = b.cfg;
    __shared__ int s_frozen;
    if (tid == 0) s_frozen = b.win_done[w];
    __syncthreads();
    const bool frozen = s_frozen != 0;            // upstream stopped iterating for this audio: carry its state over
    const int len = b.n_init + step;
    const int nb = (step == 0) ? 1 : G;           // at the first step every beam is the same sequence: the
                                                  // upstream dict of candidate sequences collapses to beam 0's
    const int nc = nb * K;
    // Round 5: the nc <= 272 candidates are fetched by nc threads in ONE round trip and ranked in parallel (one lane had walked
    // them one by one -- 3 nc dependent loads -- and insertion-sorted them in LDS: 29 us per step, a third of a 5-row decode step's
    // selection).  rank = the candidate's position in sorted(scores, key=scores.get, reverse=True): the number of candidates with a
    // greater score plus the equal ones in front of it -- exactly the stable insertion sort's result (it moved strictly smaller
    // entries only).
    if (!frozen) {
        for (int ci = tid; ci < nc; ci += 256) {         // nc <= 17 * 16 = 272
            const int j = ci / K, k = ci - j * K, idx = w * G + j;
            sc[ci] = b.sum_lp[idx] + b.cand_lp[(size_t)idx * K + k];
            src[ci] = (short)idx;
            tk[ci] = b.cand_tok[(size_t)idx * K + k];
        }
    }
    __syncthreads();
    if (!frozen) {
        for (int ci = tid; ci < nc; ci += 256) {
            // a TOTAL order even if a score is NaN (a numeric blow-up upstream of here): NaN ranks as -inf, so that `order` stays
            // a permutation and lane 0 below never reads an unwritten slot; finite scores rank exactly as before
            const float raw = sc[ci], mine = (raw != raw) ? -INFINITY : raw;
            int rank = 0;
            for (int q = 0; q < nc; ++q) {
                const float rq = sc[q], oq = (rq != rq) ? -INFINITY : rq;
                rank += (oq > mine || (oq == mine && q < ci)) ? 1 : 0;
            }
            order[rank] = (short)ci;
        }
    }
    __syncthreads();
    if (frozen) {
        if (tid < G) sel_src[tid] = w * G + tid;
    } else if (tid == 0) {
        int saved = 0;
        int fc = b.fin_count[w];
        // newly finished sequences arrive in descending score order, which is also the order upstream merges them
        for (int a = 0; a < nc && saved < G; ++a) {
            const int ci = order[a];
            if (tk[ci] == c.eot) {
                if (fc < b.max_cand) {
                    int32_t *ft = b.fin_tokens + ((size_t)w * b.fin_cap + fc) * b.TS;
                    const int32_t *st = b.tokens[cur] + (size_t)src[ci] * b.TS;
                    for (int i = 0; i < len; ++i) ft[i] = st[i];
                    for (int i = len; i < b.TS; ++i) ft[i] = c.eot;
                    b.fin_score[(size_t)w * b.fin_cap + fc] = sc[ci];
                    b.fin_len[(size_t)w * b.fin_cap + fc] = len + 1;
                    ++fc;
                }
            } else {
                sel_src[saved] = src[ci]; sel_tok[saved] = tk[ci]; sel_sc[saved] = sc[ci];
                ++saved;
            }
        }
        b.fin_count[w] = fc;
        if (fc >= b.max_cand) b.win_done[w] = 1;
    }
    __syncthreads();
    // gather the surviving beams into the other buffer (tokens + ancestor rows); this IS rearrange_kv_cache
    for (int s = 0; s < G; ++s) {
        const int dst = w * G + s, from = sel_src[s];
        const int32_t *st = b.tokens[cur] + (size_t)from * b.TS;
        int32_t *dt = b.tokens[cur ^ 1] + (size_t)dst * b.TS;
        const int32_t *sa = b.anc[cur] + (size_t)from * b.n_ctx;
        int32_t *da = b.anc[cur ^ 1] + (size_t)dst * b.n_ctx;
        if (frozen) {
            for (int i = tid; i < b.TS; i += 256) dt[i] = st[i];
            for (int p = tid; p < b.n_ctx; p += 256) da[p] = sa[p];
        } else {
            for (int i = tid; i < b.TS; i += 256) dt[i] = (i < len) ? st[i] : (i == len ? sel_tok[s] : c.eot);
            for (int p = tid; p < b.n_ctx; p += 256) da[p] = (p < len) ? sa[p] : dst;
            // (sum_logprobs written in place: every read of this window's rows sits in front of the barriers above and no other
            //  workgroup touches them -- the separate commit launch of rounds 1-5 cost 4.7 us per step)
            if (tid == 0) { b.sum_lp[dst] = sel_sc[s]; b.pos0[dst] = len; }
        }
    }
}

// per-window completion (greedy: every row ended with EOT) and the global "all done" counter
__global__ __launch_bounds__(64) void decode_step_finish_kernel(DecodeBufs b)
{
    // one wave, a window per lane (one lane had walked the windows one dependent load after the other: 6 us per step at 20 windows)
    if (blockIdx.x != 0) return;
    const int lane = threadIdx.x;
    int nd = 0;
    for (int w = lane; w < b.W; w += 64) {
        int done = b.win_done[w];
        if (!b.cfg.beam && !done) {
            int all = 1;
            for (int g = 0; g < b.G; ++g) all &= b.row_done[w * b.G + g];
            if (all) { b.win_done[w] = 1; done = 1; }
        }
        b.win_done_prev[w] = done;
        nd += done ? 1 : 0;
    }
    nd += lane_xor<32>(nd, lane); nd += lane_xor<16>(nd, lane); nd += lane_xor<8>(nd, lane);
    nd += lane_xor<4>(nd, lane); nd += lane_xor<2>(nd, lane); nd += lane_xor<1>(nd, lane);
    if (lane == 0) { *b.n_done = nd; *b.step_dev += 1; }
}

// ------------------------------------------------------------------------------------------------ finalize
// grid (W): writes [G_out] candidate sequences per window.
__global__ void decode_finalize_kernel(DecodeBufs b, int cur, int n_steps, int32_t *__restrict__ tokens_out,
                                       int32_t *__restrict__ lens_out, float *__restrict__ sumlp_out, int G_out)
{
    const int w = blockIdx.x, tid = threadIdx.x, TS = b.TS;
    const swx_decode_cfg &c = b.cfg;
    __shared__ int n_out;
    __shared__ int pick[16];
    if (!c.beam) {
        // GreedyDecoder.finalize: pad one EOT (the buffers are EOT-filled beyond the sampled tokens already)
        for (int g = 0; g < b.G; ++g) {
            const int32_t *st = b.tokens[cur] + (size_t)(w * b.G + g) * TS;
            int32_t *dt = tokens_out + ((size_t)w * G_out + g) * TS;
            for (int i = tid; i < TS; i += blockDim.x) dt[i] = st[i];
            if (tid == 0) sumlp_out[(size_t)w * G_out + g] = b.sum_lp[w * b.G + g];
        }
        if (tid == 0) n_out = b.G;
    } else {
        // BeamSearchDecoder.finalize: top up with the best unfinished beams (+EOT) when fewer than beam_size finished
        if (tid == 0) {
            int fc = b.fin_count[w];
            int np = 0;
            if (fc < b.G) {
                bool used[16];
                for (int g = 0; g < b.G; ++g) used[g] = false;
                while (fc + np < b.G) {
                    int bj = -1;
                    for (int g = b.G - 1; g >= 0; --g)      // descending sum_logprobs (argsort()[::-1])
                        if (!used[g] && (bj < 0 || b.sum_lp[w * b.G + g] > b.sum_lp[w * b.G + bj])) bj = g;
                    used[bj] = true;
                    pick[np++] = bj;
                }
            }
            n_out = fc + np;
        }
        __syncthreads();
        const int fc = b.fin_count[w];
        for (int k = 0; k < n_out; ++k) {
            int32_t *dt = tokens_out + ((size_t)w * G_out + k) * TS;
            if (k < fc) {
                const int32_t *st = b.fin_tokens + ((size_t)w * b.fin_cap + k) * TS;
                for (int i = tid; i < TS; i += blockDim.x) dt[i] = st[i];
                if (tid == 0) sumlp_out[(size_t)w * G_out + k] = b.fin_score[(size_t)w * b.fin_cap + k];
            } else {
                const int g = pick[k - fc];
                const int32_t *st = b.tokens[cur] + (size_t)(w * b.G + g) * TS;
                for (int i = tid; i < TS; i += blockDim.x) dt[i] = st[i];     // already EOT beyond the sampled tokens
                if (tid == 0) sumlp_out[(size_t)w * G_out + k] = b.sum_lp[w * b.G + g];
            }
        }
    }
    __syncthreads();
    // lens: tokens[sample_begin : first eot]; unused candidate slots get len = -1
    for (int k = tid; k < G_out; k += blockDim.x) {
        if (k >= n_out) { lens_out[(size_t)w * G_out + k] = -1; continue; }
    }
    __syncthreads();
    if (tid == 0) {
        for (int k = 0; k < n_out; ++k) {
            const int32_t *dt = tokens_out + ((size_t)w * G_out + k) * TS;
            int e = c.sample_begin;
            while (e < TS && dt[e] != c.eot) ++e;
            lens_out[(size_t)w * G_out + k] = e - c.sample_begin;
        }
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------ launchers
int swx_decode_init(const DecodeBufs &b, const int32_t *init_tokens, hipStream_t s)
{
    hipLaunchKernelGGL(decode_init_kernel, dim3(b.M), dim3(256), 0, s, b, init_tokens);
    SWX_CHECK_LAUNCH();
    return 0;
}
int swx_decode_after_prefill(const DecodeBufs &b, const float *lg2, float *nospeech, hipStream_t s)
{
    hipLaunchKernelGGL(decode_after_prefill_kernel, dim3(b.M), dim3(256), 0, s, b);
    hipLaunchKernelGGL(decode_prefill_logits_kernel, dim3(b.M), dim3(SEL_T), 0, s, b, lg2, nospeech);
    SWX_CHECK_LAUNCH();
    return 0;
}
int swx_decode_select(const DecodeBufs &b, int cur, hipStream_t s)
{
    SwxProfScope prof(PC_SELECT, (double)b.M * b.V * 4.0, s);
    const bool mem_kernel = swx_flags() & SWX_FLAG_SELECT_MEM;       // A/B and bit-identity reference of the register kernel
    if (!mem_kernel && b.V <= 51 * SEL_T)
        hipLaunchKernelGGL(decode_select_reg_kernel<51>, dim3(b.M), dim3(SEL_T), 0, s, b, cur);
    else
        hipLaunchKernelGGL(decode_select_kernel, dim3(b.M), dim3(SEL_T), 0, s, b, cur);
    if (b.cfg.beam) {
        hipLaunchKernelGGL(decode_beam_update_kernel, dim3(b.W), dim3(256), 0, s, b, cur);
    }
    hipLaunchKernelGGL(decode_step_finish_kernel, dim3(1), dim3(64), 0, s, b);
    SWX_CHECK_LAUNCH();
    return 0;
}
int swx_decode_finalize(const DecodeBufs &b, int cur, int n_steps, int32_t *tokens_out, int32_t *lens_out,
                        float *sumlp_out, int G_out, hipStream_t s)
{
    hipLaunchKernelGGL(decode_finalize_kernel, dim3(b.W), dim3(256), 0, s, b, cur, n_steps, tokens_out, lens_out, sumlp_out, G_out);
    SWX_CHECK_LAUNCH();
    return 0;
}
