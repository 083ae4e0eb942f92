// swx_kernels.h -- internal launcher prototypes shared by the translation units of libswx.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/swx.h"

#define EPI_BIAS 1
#define EPI_GELU 2
#define EPI_RES 4
#define EPI_OUT_F32 8
#define EPI_RESF32MOD 16
#define EPI_CBATCH 64        // C rows are grouped in batch items of vt_s rows, vt_bs elements apart (row-major inside an item)
#define EPI_STORE_VT 32     // store C transposed per head: C[((row / vt_s) * (N/64) + col/64) * 64 + col%64][row % vt_s] with row stride vt_kp
#define EPI_KV 128          // f16 tiled kernels only: ONE launch for the cross-attention K and V projections of a layer (fused weight rows
                            // [2d][d]): columns [0, N/2) are stored like EPI_CBATCH into C, columns [N/2, N) like EPI_STORE_VT into C2
                            // (column index relative to N/2); N/2 is a multiple of the tile width, so a tile is one or the other

#define EPI_QKV_VT 256      // f16 tiled kernels only (round 6): the encoder's fused Q|K|V projection with V stored TRANSPOSED per head for the
                            // flash kernel -- columns [0, 2N/3) plain into C, columns [2N/3, N) like EPI_STORE_VT into C2 (column index relative
                            // to 2N/3; vt_zero_pad: the key padding [vt_s, vt_kp) of every column is zeroed by the tile that ends a batch item)

struct GemmArgs {
    const void *A; int64_t lda;     // [M][K] compute dtype
    const void *W; int64_t ldw;     // [N][K] compute dtype
    const float *bias;              // [N] or null
    const void *R; int64_t ldr;     // residual, compute dtype (may alias C)
    const float *Rf; int res_mod;   // f32 residual [res_mod][N], row index = m % res_mod
    void *C; int64_t ldc;           // compute dtype, or f32 with EPI_OUT_F32
    void *C2;                       // EPI_KV: base of the transposed half
    void *P; int64_t p_bs;          // EPI_KV, optional (round 6): the fragment-ordered copy of K / V^T the decode-step cross-attention streams
    int p_nkpad, p_vcol0;           // (swx_xkv_packed_elems_per_head) written by the same epilogue instead of by swx_xkv_pack: base of batch item
                                    // 0, batch stride in elements, keys per head rounded up to 32 (the padding keys are zeroed), first V column
    int M, N, K;
    int epi;
    int vt_s, vt_kp; int64_t vt_bs;   // EPI_STORE_VT: rows per batch item, padded key stride, element stride between batch items
    int vt_zero_pad;                  // EPI_STORE_VT fast path: also write zeros to keys [vt_s, vt_kp) of the columns it stores
    int tile_order;                   // gemm_f16_big8 only, set by swx_gemm: 0 = row-major, 1 = an XCD's concurrent tiles share a band of
                                      // weights that fits its L2 (groups of 4 column tiles, walked along M; SWX_FLAG_BIG8_GROUPED)
};
int swx_gemm(int dtype, const GemmArgs &g, int force_kernel, hipStream_t s);
// the f16 kernel a launch gets (swx_gemm.hip; a pure function of the shape, the epilogue and the switches)
enum { SWX_GEMM_TILED_REG = 0, SWX_GEMM_SKINNY = 1, SWX_GEMM_GLDS128 = 2, SWX_GEMM_GLDS64 = 3, SWX_GEMM_RING64 = 4, SWX_GEMM_RING128 = 5,
       SWX_GEMM_BIG = 6 };
int swx_gemm_plan_f16(int M, int N, int K, int epi, int64_t ldc, int64_t ldr, bool ptr16, int force_kernel, int flags);

// ---- run-time A/B switches (swx_debug_flags(); tests and scripts only -- no environment variable reads them).  Each one
//      selects the bit-identity / parity REFERENCE of a kernel class; the superseded generations these flags used to keep
//      alive (split-K decode step, flash attention v1, DTW generation 2, the tiled-GEMM variants that lost) were deleted in
//      round 3.
#define SWX_FLAG_NO_FAST_STEP 1       // decode steps / small passes go through the generic per-op path (LayerNorm launches,
                                      // row-major weights, skinny / tiled GEMMs): the reference of the fused "dec" step
#define SWX_FLAG_NO_PACKED_XKV 2048   // decode cross-attention reads the row-layout K / V^T instead of the fragment-ordered copy
#define SWX_FLAG_SELECT_MEM 8192      // logit filters + token selection: the kernel that walks the row in memory (reference of
                                      // the register-resident kernel, and its fallback for vocabularies > 51 * 1024)
#define SWX_FLAG_NO_GRAPH 16384       // decode loop: launch every step eagerly instead of replaying the captured two-step graph
#define SWX_FLAG_NO_PREFETCH 32768    // decode step: no cache prefetch of the next projection's weights (A/B; no functional effect)
#define SWX_FLAG_NO_RING 65536        // tiled GEMM: never the ring kernel for launches with few tiles (A/B; results are bit-identical)
#define SWX_FLAG_SCORE_TILED 262144   // multi-token decoder passes above 160 rows on the tiled GEMMs + flash attention (round 3's
                                      // dispatch: faster at >= 2 windows, but a window's rounding then depends on its batch) -- A/B only
#define SWX_FLAG_NO_TALL 524288       // multi-token passes: never the tall dec GEMM (register-resident weights, 16-row tiles): A/B, bit-identical
#define SWX_FLAG_NO_FUSED_XQ 1048576  // decode step: cross-attention query projection as a launch of its own instead of inside the
                                      // cross-attention kernel (A/B; bit-identical)
#define SWX_FLAG_TICKET 2097152       // decode step: the K-split projection's slabs reduced INSIDE the GEMM launch by the last-arriving K slice
                                      // (DEC_TICKET) instead of by dec_slab_finish.  Built and measured in round 5 in two forms, both
                                      // bit-identical: plain slab stores + agent-scope release / acquire fences 451.8 vs 432.6 ms per headline
                                      // pass (the launch 17.0 us against 8.2 + 5.0); write-through `sc1` slab stores + drained ticket + `sc1`
                                      // reads (the form kept) 432.4 vs 430.1 ms -- break-even at best, so off by default; kept for A/B
                                      // (profiles/r05_c5_bench_ticket_ab.json, r05_c12_bench_ticket_sc1_ab.json).
#define SWX_FLAG_BIG8_GROUPED 4194304 // 256 x 256 GEMM: tiles listed in groups of 4 column tiles walked along M instead of row-major (A/B;
                                      // bit-identical).  Built in round 6 to cut the N = 5120 launch's 3.8x fabric traffic: measured neutral to
                                      // 3 % SLOWER at every encoder shape (profiles/r06_c2_kb_gemm_big_tile_order.txt), so off by default
#define SWX_FLAG_XKV_TWO_LAUNCHES 8388608 // cross-K/V projection as two launches (K, then V) instead of one over the fused weight rows (A/B; bit-identical)
#define SWX_FLAG_LOUDNESS_ONE_WG 16777216  // silence analysis probe: always the one-workgroup-per-window selection kernel (A/B; the same element)
#define SWX_FLAG_XATTN_R5 33554432     // decode cross-attention: one key block per wave in flight, default load policy (rounds 2-5) instead of two + nt (A/B; bit-identical)
#define SWX_FLAG_FLASH_PIPELINED 67108864 // f16 flash attention: the software-pipelined tile attn_flash3_f16 (round 6: bit-identical, measured 4 % SLOWER than generation 2 -- DESIGN.md section 7) instead of attn_flash2_f16
#define SWX_FLAG_SELFATTN_WG5 134217728  // decode-step self-attention: five rows (the beams of a window) per workgroup instead of one wave per workgroup (A/B; the same arithmetic per row)
#define SWX_FLAG_QKV_SEPARATE_VT 268435456 // encoder at few windows: V transposed by its own launch instead of by the QKV projection's epilogue (A/B; bit-identical)
#define SWX_FLAG_SELFATTN_NO_DEEP 4    // decode-step self-attention, long context at <= 1 024 waves: the chunk-by-chunk kernel instead of every load in two batches (A/B; bit-identical)
#define SWX_FLAG_DEC_NO_W1 16           // decode-step GEMM at <= 80 workgroups: four-wave workgroups instead of single-wave ones (A/B; bit-identical)
#define SWX_FLAG_TALL_NO_W8 32           // tall dec GEMM: four waves = one 64-column panel per workgroup (rounds 4-5) instead of eight waves = two panels (A/B; bit-identical)
#define SWX_FLAG_SELFATTN_NO_MQ 256       // multi-token self-attention: one wave per (row, token, head) reading K / V from L2 (rounds 1-5) instead of 4-8 tokens per workgroup from LDS (A/B; bit-identical)
#define SWX_FLAG_FLASH_NO_QB1 64           // f16 flash attention of launches with <= 256 workgroups at 32 queries per wave: 32 queries per wave (rounds 3-5) instead of 16 (A/B; bit-identical)
#define SWX_FLAG_XKV_PACK_SEPARATE 128     // cross-K/V: the fragment-ordered copy by its own launch (swx_xkv_pack, rounds 2-5) instead of by the projection's epilogue (A/B; the same bytes)
#define SWX_FLAG_DEC_W1_FULL_TILE 256     // single-wave dec GEMM workgroups: DMA all 16 rows of the activation tile (rows past M clamped) instead of the rows that exist (A/B; bit-identical)
#define SWX_FLAG_XQ_FULL_TILE 512          // decode cross-attention's fused query projection: DMA all 16 rows of the residual tile instead of the window's nq rows (A/B; bit-identical)
#define SWX_FLAG_NO_BIG_TILE 131072   // tiled GEMM: never the 256 x 256 kernel (A/B; results are bit-identical)
#define SWX_DEFAULT_FLAGS 0
int swx_flags();

// ---- decode-step GEMM (swx_decstep.hip): M split over workgroups, K whole -> the kernel finishes its own outputs (no f32
//      slabs, no finish launch), LayerNorm folded into the consumer (statistics from the staged tile)
#define DEC_LN 1        // out = rstd[m] * (acc - mean[m] * c1[n]) + c2[n]   (A = raw residual stream, W = gamma-folded weights)
#define DEC_GELU 2
#define DEC_RES 4       // X[m][n] = f16(X + c2[n] + acc)   (in place)
#define DEC_QKV 8       // columns [0,d) -> C, [d,2d) -> kcache[m][pos0[m]], [2d,3d) -> vcache[m][pos0[m]]
#define DEC_SLAB 16     // K-split allowed: f32 partials to slabs + dec_slab_finish (needs DEC_RES)
#define DEC_TICKET 32   // (launcher-internal, with DEC_SLAB | DEC_RES) the slab reduction INSIDE the launch: the K slices of a (panel, row
                        // group) draw a ticket after publishing their slab; the last arriver reduces (dec_slab_finish's arithmetic)
// The packed weights of the projection that runs NEXT in the decode step, for the cache prefetch the current kernel issues: a few
// loads per wave whose results nobody reads, one per 128-byte line, dealt over the lanes of the launch in order.  What gets warm
// is the memory-side Infinity Cache (the consumer's FETCH_SIZE does not drop, its latency does: section 5 of DESIGN.md), so
// where a line's load is issued from does not matter.  No functional effect.  base == null: nothing to prefetch.
struct DecPrefetch {
    const void *base;                    // packed weights, N * K halfs, contiguous
    int lines;                           // N * K * 2 / 128
};
DecPrefetch swx_dec_prefetch_of(const void *packed_w, int M, int N, int K, int epi);

struct DecGemmArgs {
    const _Float16 *A; int64_t lda;      // [M][K]
    const _Float16 *W; int64_t ldw;      // PACKED weights (swx_fold_ln: [N/16][K/32][64][8], MFMA fragment order); ldw unused
    int M, N, K;
    int epi;
    const float *c2;                     // [N] bias (or folded bias with DEC_LN)
    const float *c1;                     // [N] column sums of the folded weights (DEC_LN)
    _Float16 *C; int64_t ldc;
    _Float16 *X; int64_t ldx;
    float *slabs;                        // swx_dec_slab_floats(M, N, K) floats when the shape runs K-split
    int *ticket;                         // K-split shapes of the decode-step kernel: SWX_DEC_TICKETS zeroed ints (self-resetting arrival
                                         // counters per (panel, row group)) -> the slab reduction runs inside the launch; null: dec_slab_finish
    _Float16 *kcache, *vcache; const int32_t *pos0; int n_ctx, d;
    int rps;                             // DEC_QKV: rows per sequence (0 / 1: one new token per row); row m = sequence m / rps, token m % rps
    int row_mul;                         // DEC_QKV: cache row (and pos0 index) of sequence q = q * row_mul (0 / 1: q itself; the prefill writes row w * G)
    int ks2, kslice, n_rg; int64_t slab_stride;   // filled by the launcher
    int tall;                            // caller: 1 = a multi-token pass (rows may run into the thousands): from 161 rows on the launcher takes the
                                         // kernel that keeps the weights in registers and walks 16-row tiles (same arithmetic per element)
    int tps;                             // launcher (tall kernel): 16-row tiles per workgroup; n_rg = row splits
    DecPrefetch pf;                      // cache prefetch of the NEXT projection's weights (pf.base == null: none)
    int w1_full_tile;               // single-wave workgroups: stage the whole 16-row tile although fewer rows exist (A/B: SWX_FLAG_DEC_W1_FULL_TILE)
};
#define SWX_DEC_TICKETS 4096
int swx_dec_plan(int M, int N, int K, int epi, int *mt, int *ks2);    // <0: shape not supported by this generation
size_t swx_dec_slab_floats(int M, int N, int K);
int swx_gemm_dec(DecGemmArgs g, hipStream_t s);
// load-time LayerNorm fold + re-pack into MFMA fragment order: Wf = pack(f16(W * gamma)), c1[n] = sum_k Wf[n][k],
// c2[n] = bias[n] + sum_k beta[k] W[n][k];  gamma == null: plain re-pack of W (c1 / c2 untouched)
int swx_fold_ln(const void *W, const float *gamma, const float *beta, const float *bias, void *Wf, float *c1, float *c2,
                int N, int K, hipStream_t s);

// ---- in-library kernel timing (HIP events on the launch stream), used by bench.py for the roofline object
enum SwxProfClass { PC_GEMM_TILED = 0, PC_GEMM_SKINNY = 1, PC_ATTN_FLASH = 2, PC_ATTN_ROWWISE = 3, PC_SELF_ATTN = 4,
                    PC_SELECT = 5, PC_MEL = 6, PC_ALIGN = 7, PC_DTW = 8, PC_NORM = 9, PC_COUNT = 10 };
bool swx_prof_on();
void swx_prof_begin(int cls, double work, hipStream_t s);   // work = algorithmic flops (MFMA classes) or bytes (HBM classes)
void swx_prof_end(hipStream_t s);
struct SwxProfScope {
    hipStream_t s; bool on;
    SwxProfScope(int cls, double work, hipStream_t st) : s(st), on(swx_prof_on()) { if (on) swx_prof_begin(cls, work, st); }
    ~SwxProfScope() { if (on) swx_prof_end(s); }
};

// ---- swx_norm.hip
int swx_layernorm(int dtype, const void *x, int64_t ldx, const float *gamma, const float *beta, void *y, int64_t ldy,
                  int rows, int d, hipStream_t s);
// melT[b][t+1][c] (c < Cp, zero padded; rows 0 and 3001 zero) <- mel[b][c][t]
int swx_mel_transpose(int dtype, const float *mel, int B, int n_mels, int Cp, void *melT, hipStream_t s);
int swx_fill_zero(void *p, size_t bytes, hipStream_t s);
// x[r][i] = tok_emb[tokens[r*ld_tok + i]] + pos_emb[pos0[r] + i]  for i < n_new[r] (rows of d)
int swx_embed(int dtype, const int32_t *tokens, int64_t ld_tok, const int32_t *tok_off, const int32_t *pos0, int R, int n_new,
              const void *tok_emb, const float *pos_emb, int d, void *x, hipStream_t s);
int swx_convert_f32(int dtype, const float *src, void *dst, int64_t n, hipStream_t s);
// strided/re-laid-out weight copies used by swx_load_tensor
int swx_copy_rows(int dtype, const float *src, int64_t src_ld, void *dst, int64_t dst_ld, int64_t rows, int64_t cols,
                  hipStream_t s);
int swx_copy_conv_w(int dtype, const float *src, int out_c, int in_c, int in_cp, void *dst, hipStream_t s);

// ---- swx_attn.hip
struct AttnArgs {
    const void *q; int64_t ldq;      // [B*nq][...] head h at column h*64
    const void *k; const void *v; int64_t ldkv;   // row stride of K (and of V when it is row-major); head h at column h*64
    int64_t k_bs, v_bs;              // element stride between batch items of K / V
    int vt_kp;                       // 0: V row-major [nk][ldkv]; >0: V transposed per batch item [H][64][vt_kp] (keys contiguous)
    void *o; int64_t ldo;
    int B, H, nq, nk;
    int q_rows_per_batch;            // rows of q per batch item (== nq unless grouped)
    const void *kv_packed;           // decode cross-attention: K and V^T of batch item 0 in MFMA fragment order (swx_xkv_pack),
                                     // batch stride k_bs; null = read a.k / a.v
    // decode step only (nq <= 16, packed K / V^T): the cross-attention QUERY projection folded into the attention launch --
    // q = LN(x) Wcq^T + b computed per (window, head) from the raw residual rows with the decode-step GEMM's arithmetic
    // (packed LayerNorm-folded weights, swx_fold_ln), instead of a launch of its own; a.q is not read then.  fq_w == null: off.
    const _Float16 *fq_x; int64_t fq_ldx; int fq_rows;      // residual stream [fq_rows][d]; window b owns rows b * q_rows_per_batch ..
    const _Float16 *fq_w; const float *fq_c1, *fq_c2; int fq_k;   // packed folded weights [d][fq_k], column constants
    const void *fq_pf; int fq_pf_lines;                     // weights to touch for the NEXT projection (cache prefetch), or null
    int fq_full_tile;                                       // A/B (SWX_FLAG_XQ_FULL_TILE): stage all 16 rows of the residual tile, not only the nq that exist
};
// fragment-ordered copy of one layer's cross K / V^T for the decode-step cross-attention: per (window, head)
// [K: blocks of 32 keys x 4 fragments x 64 lanes x 8 halfs | V^T: the same], zero padded past nk
__host__ __device__ inline int64_t swx_xkv_packed_elems_per_head(int nk) { return (int64_t)((nk + 31) / 32) * 4 * 512 * 2; }
int swx_xkv_pack(const void *k, const void *vt, void *packed, int B, int H, int nk, int ldk, int vt_kp, int64_t batch_stride,
                 hipStream_t s);
#define SWX_VT_KP 1536               // padded key count of the transposed cross-attention V (64-key tiles never run off a row)
// dense (non-causal) attention over nk keys: encoder self-attention and cross-attention
int swx_attention(int dtype, const AttnArgs &a, int force_kernel, hipStream_t s);
// V [B][n][ldv] -> per-head transposed V^T [B][H][64][kp] (keys contiguous, zero padded): feeds the MFMA flash kernel's vector path
int swx_transpose_v(const void *v, int64_t ldv, int64_t v_bs, int n, void *vt, int kp, int64_t vt_bs, int B, int H, hipStream_t s);
int swx_pad_zero(void *base, int64_t row_bytes, int64_t batch_bytes, int off_bytes, int pad_bytes, int rows, int nb, hipStream_t s);
// decoder self-attention over the per-row KV cache with ancestor indirection
struct SelfAttnArgs {
    const void *qkv; int64_t ldqkv;  // [R*n_new][3d]: q | k | v of the new tokens
    void *kcache; void *vcache;      // [Mphys][n_ctx][d]
    int32_t *anc;                    // [rows][n_ctx] cache row holding position p of a logical row (null = identity)
    const int32_t *pos0;             // [R] position of the first new token of each row
    void *o; int64_t ldo;            // [R*n_new][d]
    int R, n_new, H, n_ctx, d;
    int skip_append;                 // K/V of the new tokens were already scattered into the cache (QKV projection's epilogue)
    int step_cached;                 // n_new == 1, f16: q at a.qkv (row stride ldqkv), the new K/V already appended -> the
                                     // latency-optimised single-token kernel of the decode step
    int step_pos;                    // profiler only: position of the new token when the host knows it (decode loop), else 0
    int pos_bound;                   // decode step: an upper bound of every row's position known to the host (initial tokens + sample
                                     // budget), or 0 = unknown.  <= 128: the kernel variant without the code for positions >= 128
    int pos0_all_zero;               // multi-token pass: the host knows that every row starts at position 0 (a.pos0 is the zeros array):
                                     // several tokens of a (row, head) per workgroup with K / V staged in LDS (self_attn_cached_mq_f16)
};
// logical row of grid index ri is ri * row_mul (prefill of beam groups computes one row per window)
int swx_self_attention(int dtype, const SelfAttnArgs &a, int row_mul, hipStream_t s);
// raw scaled qk of selected heads: out[w][hi][i][f] = 0.125 * q[w][row0+i][head] . k[w][f][head]
int swx_qk_capture(int dtype, const void *q, int64_t ldq, int q_rows_per_w, int row0, int n_rows, const void *k,
                   int64_t ldk, int64_t k_bs, int nk, const int32_t *heads, int n_heads, int head_slot0, int slots_total, int W,
                   float *out, int out_ld_n, int out_ld_f, hipStream_t s);

// ---- swx_align.hip / swx_mel.hip
int swx_align_weights_launch(const float *d_qk, float *d_p, float *d_mean, float *d_sd, int W, int H, int N, int ld_f,
                             const int *d_n_rows, const int *d_n_frames, float qk_scale, int medfilt_width,
                             float *d_neg_matrix, int out_ld_n, int out_ld_f, hipStream_t s);
int swx_mel_launch(const float *d_pcm, int B, const float *d_hann, const double2 *d_twiddle, const float *d_filters,
                   int n_mels, float *d_mel, unsigned *d_gmax, int per_item_max, hipStream_t s);
int swx_mel_ragged_launch(const float *d_pcm, const int *d_lens, int B, const float *d_hann, const double2 *d_twiddle,
                          const float *d_filters, int n_mels, float *d_mel, unsigned *d_gmax, int per_item_max,
                          hipStream_t s);

// ---- swx_headsel.hip: the head-selection variants of the word-timestamp stage from the captured cross-attention queries
//      qcap [L][max_n][d] and the window's cross-K (row layout, layer stride in elements); see the file header
constexpr int SWX_HS_MAXF = 1536;
int swx_headsel_dynamic_launch(int dtype, const void *qcap, int max_n, int d, int row0, int n_rows, const void *xkv,
                               int64_t layer_stride, int L, int H, int F, int nk, float qk_scale, int count, const double *d_peaks,
                               double *d_score, int32_t *d_sel, float *d_out, int out_ld_f, hipStream_t s);
int swx_headsel_new_launch(int dtype, const void *qcap, int max_n, int d, int n, int row0, int n_out, const void *xkv,
                           int64_t layer_stride, int L, int H, int F, float qk_scale, int medfilt_width, int topk, float w_col,
                           float w_row, float w_cov, float *d_colnorm, float *d_score, int32_t *d_top, float *d_out, int out_ld_f,
                           hipStream_t s);
int swx_weighted_sum_launch(const float *const *h_xs, const float *h_coef, int n_in, float *d_out, int64_t n, hipStream_t s);

// ---- swx_decode.hip
struct DecodeState;   // device-resident bookkeeping, defined in swx_decode.hip
