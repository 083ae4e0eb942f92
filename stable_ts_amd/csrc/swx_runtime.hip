// swx_runtime.hip -- the C ABI of libswx.so (include/swx.h): weight arena, workspace, and the host-side drivers that
// sequence the gfx950 kernels for the encoder, the decoding loop and the teacher-forced scoring pass.
// No device allocation happens here: the caller binds the arena and the workspace (PyTorch owns the memory).
#include <map>
#include <memory>
#include <string>
#include <vector>
#include <cmath>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include "swx_common.h"
#include "swx_kernels.h"
#include "swx_decode.h"

namespace {

enum SlotKind { SK_MAT = 0, SK_VEC = 1, SK_CONV = 2 };

struct Slot {
    size_t off = 0;        // byte offset in the arena
    int kind = SK_VEC;
    int64_t rows = 0, cols = 0;   // source shape (MAT: [rows][cols]; CONV: out_c=rows, in_c=cols)
    int64_t dst_ld = 0;    // MAT: destination leading dimension; CONV: padded in-channels
    bool loaded = false;
};

struct LayerW {
    size_t ln1_g, ln1_b, wqkv, bqkv, wo, bo;
    size_t lnx_g, lnx_b, wcq, bcq, wckv, bckv, wco, bco;   // decoder only
    size_t ln2_g, ln2_b, w1, b1, w2, b2;
    // decoder, f16: LayerNorm-folded copies for the third-generation decode step (swx_decstep.hip): W.gamma, c1, c2
    size_t wqkv_f, qkv_c1, qkv_c2, wcq_f, cq_c1, cq_c2, w1_f, w1_c1, w1_c2;     // folded + packed in MFMA fragment order
    size_t wo_p, wco_p, w2_p;                                                   // packed copies of the other three
};

size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

}  // namespace

struct swx_model {
    swx_dims dims;
    int dtype;
    size_t esz;                     // bytes per compute element
    int Cp;                         // conv1 input channels padded to a multiple of 32
    std::map<std::string, Slot> slots;
    std::vector<LayerW> enc, dec;
    size_t o_conv1_w, o_conv1_b, o_conv2_w, o_conv2_b, o_enc_pos, o_lnpost_g, o_lnpost_b;
    size_t o_tok_emb, o_dec_pos, o_ln_g, o_ln_b;
    size_t o_hann, o_twiddle, o_filters, o_heads, o_zeros;
    size_t arena_bytes = 0;
    unsigned char *arena = nullptr;
    bool has_fold = false;          // the layout holds the folded copies (f16, d % 128 == 0)
    // ... and swx_weights_finalize has filled them.  The state belongs to the ARENA, not to the handle: views made by
    // swx_share_weights hold the same flag, so a tensor reloaded through the owner (flag cleared until the next finalize)
    // takes every view off the folded copies as well
    std::shared_ptr<bool> fold_state = std::make_shared<bool>(false);
    bool is_folded() const { return has_fold && *fold_state; }
    // alignment heads
    std::vector<std::vector<int>> heads_by_layer;   // per decoder layer
    std::vector<int> head_slot0;                    // first capture slot of each layer
    int n_align = 0;
    std::vector<int32_t> heads_flat;      // heads of all layers, layer-major (device copy: ws + L.heads)
    // workspace
    unsigned char *ws = nullptr;
    size_t ws_bytes = 0;
    int max_windows = 0, max_rows = 0, ws_n_align = 0;
    struct WsLayout {
        size_t melT, h1, x, h, qkv, att, u, gmax, small_i32, zeros_i32, ticket;
        size_t tokens0, tokens1, anc0, anc1, pos0, sum_lp, sum_lp_next, row_done, win_done, win_done_prev, n_done;
        size_t fin_tokens, fin_score, fin_len, fin_count, cand_lp, cand_tok, logits, hid2;
        size_t kcache, vcache, sk, sv, cap, mean, sd, suppress, slabs, heads, win_uid;
        size_t total;
        int64_t rows_big, logits_rows;
        size_t slab_bytes;
    } L;

    template <typename P> P *A(size_t off) const { return (P *)(arena + off); }
    template <typename P> P *Wp(size_t off) const { return (P *)(ws + off); }

    // captured two-step decode graphs (swx_decode): keyed on everything a kernel argument of the step depends on
    struct StepGraph { std::vector<uint64_t> key; hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr; uint64_t used = 0; };
    std::vector<StepGraph> graphs;
    uint64_t graph_clock = 0;
    hipStream_t cap_stream = nullptr;   // capture happens here (the caller's stream may be the null stream, which cannot capture)
    bool graphs_off = false;            // set when capture / replay failed in three swx_decode calls of this handle: eager from then on
    int graph_failures = 0;
    int64_t n_captures = 0, n_replays = 0, n_eager_units = 0;      // swx_graph_stats
    void drop_graphs() {
        for (auto &g : graphs) { if (g.exec) (void)hipGraphExecDestroy(g.exec); if (g.graph) (void)hipGraphDestroy(g.graph); }
        graphs.clear();
    }
    ~swx_model() { drop_graphs(); if (cap_stream) (void)hipStreamDestroy(cap_stream); }
};

namespace {

constexpr int FIN_CAP = 32;
constexpr int MAX_GROUP = 16;
constexpr int MAX_SUPPRESS = 4096;
constexpr int SMALL_I32 = 8192;

size_t add_slot(swx_model *m, size_t &cur, const std::string &name, int kind, int64_t rows, int64_t cols, int64_t dst_ld,
                size_t bytes, size_t base_off = (size_t)-1, size_t sub_off = 0)
{
    Slot s;
    s.kind = kind; s.rows = rows; s.cols = cols; s.dst_ld = dst_ld;
    if (base_off == (size_t)-1) { s.off = cur; cur = align_up(cur + bytes); }
    else s.off = base_off + sub_off;
    m->slots[name] = s;
    return s.off;
}

void build_layout(swx_model *m)
{
    const swx_dims &D = m->dims;
    const size_t e = m->esz;
    size_t cur = 0;
    auto mat = [&](const std::string &n, int64_t r, int64_t c) { return add_slot(m, cur, n, SK_MAT, r, c, c, (size_t)r * c * e); };
    auto vec = [&](const std::string &n, int64_t c) { return add_slot(m, cur, n, SK_VEC, 1, c, c, (size_t)c * 4); };
    auto reserve = [&](size_t bytes) { size_t o = cur; cur = align_up(cur + bytes); return o; };

    const int da = D.n_audio_state, dt = D.n_text_state;
    m->Cp = (D.n_mels + 31) / 32 * 32;
    m->o_conv1_w = add_slot(m, cur, "encoder.conv1.weight", SK_CONV, da, D.n_mels, m->Cp, (size_t)da * 3 * m->Cp * e);
    m->o_conv1_b = vec("encoder.conv1.bias", da);
    m->o_conv2_w = add_slot(m, cur, "encoder.conv2.weight", SK_CONV, da, da, da, (size_t)da * 3 * da * e);
    m->o_conv2_b = vec("encoder.conv2.bias", da);
    m->o_enc_pos = add_slot(m, cur, "encoder.positional_embedding", SK_VEC, 1, (int64_t)D.n_audio_ctx * da, 0, (size_t)D.n_audio_ctx * da * 4);

    auto block = [&](const std::string &p, int d, bool cross) {
        LayerW w{};
        w.ln1_g = vec(p + "attn_ln.weight", d);
        w.ln1_b = vec(p + "attn_ln.bias", d);
        w.wqkv = reserve((size_t)3 * d * d * e);
        add_slot(m, cur, p + "attn.query.weight", SK_MAT, d, d, d, 0, w.wqkv, 0);
        add_slot(m, cur, p + "attn.key.weight", SK_MAT, d, d, d, 0, w.wqkv, (size_t)d * d * e);
        add_slot(m, cur, p + "attn.value.weight", SK_MAT, d, d, d, 0, w.wqkv, (size_t)2 * d * d * e);
        w.bqkv = reserve((size_t)3 * d * 4);
        add_slot(m, cur, p + "attn.query.bias", SK_VEC, 1, d, d, 0, w.bqkv, 0);
        add_slot(m, cur, p + "attn.value.bias", SK_VEC, 1, d, d, 0, w.bqkv, (size_t)2 * d * 4);
        w.wo = mat(p + "attn.out.weight", d, d);
        w.bo = vec(p + "attn.out.bias", d);
        if (cross) {
            w.lnx_g = vec(p + "cross_attn_ln.weight", d);
            w.lnx_b = vec(p + "cross_attn_ln.bias", d);
            w.wcq = mat(p + "cross_attn.query.weight", d, d);
            w.bcq = vec(p + "cross_attn.query.bias", d);
            w.wckv = reserve((size_t)2 * d * d * e);
            add_slot(m, cur, p + "cross_attn.key.weight", SK_MAT, d, d, d, 0, w.wckv, 0);
            add_slot(m, cur, p + "cross_attn.value.weight", SK_MAT, d, d, d, 0, w.wckv, (size_t)d * d * e);
            w.bckv = reserve((size_t)2 * d * 4);
            add_slot(m, cur, p + "cross_attn.value.bias", SK_VEC, 1, d, d, 0, w.bckv, (size_t)d * 4);
            w.wco = mat(p + "cross_attn.out.weight", d, d);
            w.bco = vec(p + "cross_attn.out.bias", d);
        }
        w.ln2_g = vec(p + "mlp_ln.weight", d);
        w.ln2_b = vec(p + "mlp_ln.bias", d);
        w.w1 = mat(p + "mlp.0.weight", 4 * d, d);
        w.b1 = vec(p + "mlp.0.bias", 4 * d);
        w.w2 = mat(p + "mlp.2.weight", d, 4 * d);
        w.b2 = vec(p + "mlp.2.bias", d);
        return w;
    };
    for (int l = 0; l < D.n_audio_layer; ++l) m->enc.push_back(block("encoder.blocks." + std::to_string(l) + ".", da, false));
    m->o_lnpost_g = vec("encoder.ln_post.weight", da);
    m->o_lnpost_b = vec("encoder.ln_post.bias", da);
    m->o_tok_emb = mat("decoder.token_embedding.weight", D.n_vocab, dt);
    m->o_dec_pos = add_slot(m, cur, "decoder.positional_embedding", SK_VEC, 1, (int64_t)D.n_text_ctx * dt, 0, (size_t)D.n_text_ctx * dt * 4);
    for (int l = 0; l < D.n_text_layer; ++l) m->dec.push_back(block("decoder.blocks." + std::to_string(l) + ".", dt, true));
    {
        int mt = 0, ks = 0;
        m->has_fold = m->dtype == SWX_F16 && swx_dec_plan(64, 3 * dt, dt, DEC_LN, &mt, &ks) == 0 &&
                      swx_dec_plan(64, dt, 4 * dt, DEC_RES | DEC_SLAB, &mt, &ks) == 0;
        if (m->has_fold)
            for (auto &w : m->dec) {
                w.wqkv_f = reserve((size_t)3 * dt * dt * e); w.qkv_c1 = reserve((size_t)3 * dt * 4); w.qkv_c2 = reserve((size_t)3 * dt * 4);
                w.wcq_f = reserve((size_t)dt * dt * e); w.cq_c1 = reserve((size_t)dt * 4); w.cq_c2 = reserve((size_t)dt * 4);
                w.w1_f = reserve((size_t)4 * dt * dt * e); w.w1_c1 = reserve((size_t)4 * dt * 4); w.w1_c2 = reserve((size_t)4 * dt * 4);
                w.wo_p = reserve((size_t)dt * dt * e); w.wco_p = reserve((size_t)dt * dt * e); w.w2_p = reserve((size_t)4 * dt * dt * e);
            }
    }
    m->o_ln_g = vec("decoder.ln.weight", dt);
    m->o_ln_b = vec("decoder.ln.bias", dt);
    // constants
    m->o_hann = vec("const.hann", SWX_N_FFT);
    m->o_filters = add_slot(m, cur, "const.mel_filters", SK_VEC, 1, (int64_t)D.n_mels * 201, 0, (size_t)D.n_mels * 201 * 4);
    m->o_twiddle = reserve(sizeof(double2) * SWX_N_FFT);
    m->o_heads = reserve(sizeof(int32_t) * (size_t)D.n_text_layer * D.n_text_head);
    m->o_zeros = reserve(256);
    m->arena_bytes = cur;
}

void default_alignment_heads(swx_model *m)
{
    const swx_dims &D = m->dims;
    m->heads_by_layer.assign(D.n_text_layer, {});
    for (int l = D.n_text_layer / 2; l < D.n_text_layer; ++l)
        for (int h = 0; h < D.n_text_head; ++h) m->heads_by_layer[l].push_back(h);
}

int upload_heads(swx_model *m)
{
    const swx_dims &D = m->dims;
    std::vector<int32_t> flat;
    m->head_slot0.assign(D.n_text_layer, 0);
    for (int l = 0; l < D.n_text_layer; ++l) {
        m->head_slot0[l] = (int)flat.size();
        for (int h : m->heads_by_layer[l]) flat.push_back(h);
    }
    m->n_align = (int)flat.size();
    m->heads_flat = flat;
    // same count as the bound workspace was laid out for: refresh its copy in place; a different count needs a re-bind
    // (swx_score / swx_score_qk refuse to run until then), which uploads the list
    if (m->ws && m->n_align == m->ws_n_align && !flat.empty()) {
        hipError_t e = hipMemcpy(m->ws + m->L.heads, flat.data(), flat.size() * 4, hipMemcpyHostToDevice);
        if (e != hipSuccess) return -100 - (int)e;
    }
    return 0;
}

void ws_layout(const swx_model *m, int Bmax, int Mmax, int n_align, swx_model::WsLayout &L)
{
    const swx_dims &D = m->dims;
    const size_t e = m->esz;
    const int da = D.n_audio_state, dt = D.n_text_state;
    const int dmax = da > dt ? da : dt;
    const int TS = D.n_text_ctx + 1;
    size_t cur = 0;
    auto take = [&](size_t bytes) { size_t o = cur; cur = align_up(cur + bytes); return o; };
    int64_t rows_big = (int64_t)Bmax * D.n_audio_ctx;
    if ((int64_t)Bmax * D.n_text_ctx > rows_big) rows_big = (int64_t)Bmax * D.n_text_ctx;
    if (Mmax > rows_big) rows_big = Mmax;
    L.rows_big = rows_big;
    L.melT = take((size_t)Bmax * 3002 * m->Cp * e);
    L.h1 = take((size_t)Bmax * 3002 * da * e);
    L.x = take((size_t)rows_big * dmax * e);
    L.h = take((size_t)rows_big * dmax * e);
    L.qkv = take((size_t)rows_big * 3 * dmax * e);
    L.att = take((size_t)rows_big * dmax * e);
    L.u = take((size_t)rows_big * 4 * dmax * e);
    L.gmax = take((size_t)Bmax * 4);
    L.small_i32 = take((size_t)SMALL_I32 * 4);
    L.zeros_i32 = take((size_t)(Mmax > Bmax ? Mmax : Bmax) * 4 + 256);
    L.ticket = take((size_t)SWX_DEC_TICKETS * 4);       // arrival counters of the in-launch slab reduction (zero between launches)
    L.tokens0 = take((size_t)Mmax * TS * 4);
    L.tokens1 = take((size_t)Mmax * TS * 4);
    L.anc0 = take((size_t)Mmax * D.n_text_ctx * 4);
    L.anc1 = take((size_t)Mmax * D.n_text_ctx * 4);
    L.pos0 = take((size_t)Mmax * 4);
    L.sum_lp = take((size_t)Mmax * 4);
    L.sum_lp_next = take((size_t)Mmax * 4);
    L.row_done = take((size_t)Mmax * 4);
    L.win_done = take((size_t)Bmax * 4);
    L.win_done_prev = take((size_t)Bmax * 4);
    L.n_done = take(256);
    L.fin_tokens = take((size_t)Bmax * FIN_CAP * TS * 4);
    L.fin_score = take((size_t)Bmax * FIN_CAP * 4);
    L.fin_len = take((size_t)Bmax * FIN_CAP * 4);
    L.fin_count = take((size_t)Bmax * 4);
    L.cand_lp = take((size_t)Mmax * (MAX_GROUP + 1) * 4);
    L.cand_tok = take((size_t)Mmax * (MAX_GROUP + 1) * 4);
    int64_t lrows = (int64_t)Mmax + 2 * Bmax;     // M step rows + the 2W prefill rows parked at the tail
    if (lrows < 64) lrows = 64;
    L.logits_rows = lrows;
    L.logits = take((size_t)lrows * D.n_vocab * 4);
    L.hid2 = take((size_t)2 * Bmax * dt * e);
    L.kcache = take((size_t)D.n_text_layer * Mmax * D.n_text_ctx * dt * e);
    L.vcache = take((size_t)D.n_text_layer * Mmax * D.n_text_ctx * dt * e);
    L.sk = take((size_t)Bmax * D.n_text_ctx * dt * e);
    L.sv = take((size_t)Bmax * D.n_text_ctx * dt * e);
    L.cap = take((size_t)Bmax * (n_align > 0 ? n_align : 1) * D.n_text_ctx * D.n_audio_ctx * 4);
    L.mean = take((size_t)Bmax * (n_align > 0 ? n_align : 1) * D.n_audio_ctx * 4);
    L.sd = take((size_t)Bmax * (n_align > 0 ? n_align : 1) * D.n_audio_ctx * 4);
    L.suppress = take((size_t)MAX_SUPPRESS * 4);
    {
        // f32 slabs of the one K-split projection (K = 4d): the decode step's rows, and every row of a multi-token pass
        // (Bmax windows x n_text_ctx tokens: 9.2 MB per window for large-v3)
        int64_t srows = Mmax > 160 ? Mmax : 160;
        if ((int64_t)Bmax * D.n_text_ctx > srows) srows = (int64_t)Bmax * D.n_text_ctx;
        const size_t mx = swx_dec_slab_floats((int)srows, dt, 4 * dt);
        L.slabs = take(mx * 4 + 256);
        L.slab_bytes = mx * 4;
    }
    // the alignment-head list lives with the workspace, not with the weights: views that share one weight arena
    // (swx_bind_weights on the same buffer) may be configured with different heads
    L.heads = take(sizeof(int32_t) * (size_t)D.n_text_layer * D.n_text_head);
    L.win_uid = take((size_t)Bmax * 4 + 256);
    L.total = cur;
}

inline hipStream_t S(void *s) { return (hipStream_t)s; }

// ---------------------------------------------------------------------------------------------- profiler
struct ProfRec { int cls; double work; hipEvent_t a, b; };
bool g_prof_enabled = false;
// A/B switches (SWX_FLAG_* in swx_kernels.h), set through swx_debug_flags() by tests and scripts; no environment variable
int g_debug_flags = SWX_DEFAULT_FLAGS;
std::vector<ProfRec> g_prof;
std::vector<hipEvent_t> g_pool;
size_t g_pool_next = 0;

hipEvent_t prof_event()
{
    if (g_pool_next == g_pool.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        g_pool.push_back(e);
    }
    return g_pool[g_pool_next++];
}

#define SWX_TRY(expr) do { int _r = (expr); if (_r < 0) return _r; } while (0)

GemmArgs gemm_args(const void *A, int64_t lda, const void *W, int64_t ldw, const float *bias, void *C, int64_t ldc,
                   int M, int N, int K, int epi)
{
    GemmArgs g{};
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.bias = bias; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.epi = epi;
    g.R = nullptr; g.ldr = 0; g.Rf = nullptr; g.res_mod = 1;
    return g;
}

// x += Linear(a) ; used for the attention / MLP output projections
int gemm_residual(swx_model *m, const void *A, int64_t lda, size_t w_off, size_t b_off, void *X, int64_t ldx, int M, int N,
                  int K, hipStream_t s)
{
    GemmArgs g = gemm_args(A, lda, m->arena + w_off, K, m->A<float>(b_off), X, ldx, M, N, K, EPI_BIAS | EPI_RES);
    g.R = X; g.ldr = ldx;
    return swx_gemm(m->dtype, g, 0, s);
}

inline int64_t xkv_plain_elems(const swx_dims &D)      // K [1500][d] | V^T [d][KP]
{
    return (int64_t)D.n_audio_ctx * D.n_text_state + (int64_t)D.n_text_state * SWX_VT_KP;
}
inline int64_t xkv_packed_elems(const swx_dims &D)     // f16 only: K and V^T again, in MFMA fragment order per head (swx_attn.hip)
{
    return (int64_t)D.n_text_head * swx_xkv_packed_elems_per_head(D.n_audio_ctx);
}
inline int64_t xkv_chunk_elems(const swx_model *m)
{
    return xkv_plain_elems(m->dims) + (m->dtype == SWX_F16 ? xkv_packed_elems(m->dims) : 0);
}

// ------------------------------------------------------------------------------------------------ decoder forward
struct FwdCfg {
    int W;                 // windows
    int rpw;               // logical rows per window in this pass
    int row_mul;           // logical row id = grid row * row_mul
    int n_new;             // new tokens per row
    const int32_t *tokens; int64_t ld_tok;
    const int32_t *pos0;   // device, indexed by logical row id
    unsigned char *kcache, *vcache;
    size_t layer_stride;   // bytes between layers in the cache (0 = single-layer scratch)
    int cache_rows;        // rows per layer in the cache
    int32_t *anc;
    const unsigned char *xkv;
    bool capture; int cap_row0, cap_rows, cap_ld_n;
    int step_pos;          // decode step: position of the new token when the host knows it (profiler's byte count), else 0
    int pos_bound;         // decode step: upper bound of every row's position (initial tokens + sample budget), 0 = unknown
    unsigned char *qcap;   // non-null: the cross-attention queries of every layer are copied here, [L][rows][d] (swx_score_q)
};

// One decoder step (n_new == 1) in fp16: un-split "dec" GEMMs (swx_decstep.hip) that finish their own outputs.
// 8 launches per layer (QKV+scatter, self-attention, out-proj+residual, cross-q, cross-attention, out-proj+residual, MLP-in+GELU,
// MLP-out [+ its slab reduction]); no LayerNorm launch, no f32 slabs except for the K = 4d projection.  (The split-K
// generation of rounds 1-2 -- 12 launches per layer -- was deleted in round 3; the generic per-op path below is the
// bit-identity-free reference: SWX_FLAG_NO_FAST_STEP.)
// Leaves the RAW residual stream in `x` (returns 0: the caller applies the final LayerNorm).
int decoder_step_dec(swx_model *m, const FwdCfg &f, hipStream_t s)
{
    const swx_dims &D = m->dims;
    const int d = D.n_text_state, H = D.n_text_head;
    const size_t e = m->esz;
    const int rows = f.W * f.rpw;
    f16 *x = m->Wp<f16>(m->L.x), *q = m->Wp<f16>(m->L.qkv), *att = m->Wp<f16>(m->L.att), *u = m->Wp<f16>(m->L.u);
    float *slabs = m->Wp<float>(m->L.slabs);
    SWX_TRY(swx_embed(m->dtype, f.tokens, f.ld_tok, nullptr, f.pos0, rows, 1, m->arena + m->o_tok_emb,
                      m->A<float>(m->o_dec_pos), d, x, s));
    const int64_t chunk = xkv_chunk_elems(m);
    // Prefetch chain (DecPrefetch, swx_kernels.h): each projection touches the weights of the NEXT projection.  What the
    // counters say it warms is the 256 MB Infinity Cache, not an L2 (profiles/r03_pmc_final.csv: the consumer's FETCH_SIZE does
    // not drop -- the L2s drop their clean lines at a kernel boundary -- but it is 0.8-1.2 us faster, r03_eager_pf_*_kernels.csv),
    // so a prefetch also survives the attention kernel in between (<= 57 MB / 154 MB through a 256 MB cache).  A prefetch issued
    // BY the self-attention kernel was tried and dropped: +0.36 us on it.
    // (from 32 rows on: with one 16-row tile per panel -- sequential transcribe(), 5 rows -- a launch has 20-80 workgroups, too
    // few lanes to cover a projection, and waits for its own prefetch at its end: measured 58.0 -> 56.7x there, 460.8 -> 450.4 ms
    // per pass at 100 rows, profiles/r03_dec_prefetch_ab.txt)
    const bool pf_on = !(g_debug_flags & SWX_FLAG_NO_PREFETCH) && rows >= 32;
    auto pf_of = [&](size_t w_off, int N, int K, int epi) {
        return pf_on ? swx_dec_prefetch_of(m->A<f16>(w_off), rows, N, K, epi) : DecPrefetch{};
    };
    for (int l = 0; l < D.n_text_layer; ++l) {
        const LayerW &w = m->dec[l];
        f16 *kc = (f16 *)(f.kcache + (size_t)l * f.layer_stride), *vc = (f16 *)(f.vcache + (size_t)l * f.layer_stride);
        DecGemmArgs g{};
        g.M = rows;
        // q | k | v = LN1(x) Wqkv^T + b : q -> the q buffer, k / v -> the cache at each row's position
        g.A = x; g.lda = d; g.W = m->A<f16>(w.wqkv_f); g.ldw = d; g.N = 3 * d; g.K = d; g.epi = DEC_LN | DEC_QKV;
        g.c1 = m->A<float>(w.qkv_c1); g.c2 = m->A<float>(w.qkv_c2); g.C = q; g.ldc = d;
        g.kcache = kc; g.vcache = vc; g.pos0 = f.pos0; g.n_ctx = D.n_text_ctx; g.d = d;
        g.pf = pf_of(w.wo_p, d, d, DEC_RES);                 // used after the self-attention (<= 57 MB through the cache)
        SWX_TRY(swx_gemm_dec(g, s));
        SelfAttnArgs sa{};
        sa.qkv = q; sa.ldqkv = d; sa.kcache = kc; sa.vcache = vc; sa.anc = f.anc; sa.pos0 = f.pos0; sa.o = att; sa.ldo = d;
        sa.R = rows; sa.n_new = 1; sa.H = H; sa.n_ctx = D.n_text_ctx; sa.d = d; sa.skip_append = 1; sa.step_cached = 1;
        sa.step_pos = f.step_pos; sa.pos_bound = f.pos_bound;
        SWX_TRY(swx_self_attention(m->dtype, sa, 1, s));
        // x += att Wo^T + bo
        g = DecGemmArgs{};
        g.M = rows; g.A = att; g.lda = d; g.W = m->A<f16>(w.wo_p); g.ldw = d; g.N = d; g.K = d; g.epi = DEC_RES;
        g.c2 = m->A<float>(w.bo); g.X = x; g.ldx = d;
        g.pf = pf_of(w.wcq_f, d, d, DEC_LN);
        SWX_TRY(swx_gemm_dec(g, s));
        // cross-attention query = LNx(x) Wcq^T + b: inside the cross-attention launch (AttnArgs::fq_*) unless switched off
        const bool fuse_xq = !(g_debug_flags & (SWX_FLAG_NO_FUSED_XQ | SWX_FLAG_NO_PACKED_XKV)) && f.rpw <= 16 && d % 128 == 0 &&
                             (d == 384 || d == 512 || d == 640 || d == 768 || d == 1024 || d == 1280);
        if (!fuse_xq) {
            g = DecGemmArgs{};
            g.M = rows; g.A = x; g.lda = d; g.W = m->A<f16>(w.wcq_f); g.ldw = d; g.N = d; g.K = d; g.epi = DEC_LN;
            g.c1 = m->A<float>(w.cq_c1); g.c2 = m->A<float>(w.cq_c2); g.C = q; g.ldc = d;
            g.pf = pf_of(w.wco_p, d, d, DEC_RES);                // used after the cross-attention (154 MB through the cache)
            SWX_TRY(swx_gemm_dec(g, s));
        }
        const unsigned char *kl = f.xkv + (size_t)l * f.W * chunk * e;
        AttnArgs ca{};
        ca.q = q; ca.ldq = d; ca.k = kl; ca.v = kl + (size_t)D.n_audio_ctx * d * e; ca.ldkv = d;
        ca.k_bs = chunk; ca.v_bs = chunk; ca.vt_kp = SWX_VT_KP; ca.o = att; ca.ldo = d;
        ca.B = f.W; ca.H = H; ca.nq = f.rpw; ca.nk = D.n_audio_ctx; ca.q_rows_per_batch = f.rpw;
        ca.kv_packed = kl + (size_t)xkv_plain_elems(D) * e;          // fragment-ordered copy of this layer's K / V^T
        if (fuse_xq) {
            ca.fq_x = x; ca.fq_ldx = d; ca.fq_rows = rows; ca.fq_w = m->A<f16>(w.wcq_f); ca.fq_c1 = m->A<float>(w.cq_c1);
            ca.fq_c2 = m->A<float>(w.cq_c2); ca.fq_k = d;
            if (pf_on) { ca.fq_pf = m->A<f16>(w.wco_p); ca.fq_pf_lines = (int)(((int64_t)d * d * 2) >> 7); }
        }
        SWX_TRY(swx_attention(m->dtype, ca, 0, s));
        g = DecGemmArgs{};
        g.M = rows; g.A = att; g.lda = d; g.W = m->A<f16>(w.wco_p); g.ldw = d; g.N = d; g.K = d; g.epi = DEC_RES;
        g.c2 = m->A<float>(w.bco); g.X = x; g.ldx = d;
        g.pf = pf_of(w.w1_f, 4 * d, d, DEC_LN | DEC_GELU);
        SWX_TRY(swx_gemm_dec(g, s));
        // MLP
        g = DecGemmArgs{};
        g.M = rows; g.A = x; g.lda = d; g.W = m->A<f16>(w.w1_f); g.ldw = d; g.N = 4 * d; g.K = d; g.epi = DEC_LN | DEC_GELU;
        g.c1 = m->A<float>(w.w1_c1); g.c2 = m->A<float>(w.w1_c2); g.C = u; g.ldc = 4 * d;
        g.pf = pf_of(w.w2_p, d, 4 * d, DEC_RES | DEC_SLAB);
        SWX_TRY(swx_gemm_dec(g, s));
        g = DecGemmArgs{};
        g.M = rows; g.A = u; g.lda = 4 * d; g.W = m->A<f16>(w.w2_p); g.ldw = 4 * d; g.N = d; g.K = 4 * d; g.epi = DEC_RES | DEC_SLAB;
        g.c2 = m->A<float>(w.b2); g.X = x; g.ldx = d; g.slabs = slabs; g.ticket = m->Wp<int>(m->L.ticket);
        if (l + 1 < D.n_text_layer) g.pf = pf_of(m->dec[l + 1].wqkv_f, 3 * d, d, DEC_LN | DEC_QKV);
        SWX_TRY(swx_gemm_dec(g, s));
    }
    return 0;
}

// Multi-token teacher-forced pass over a SMALL number of rows (align(): one window of ~110 tokens; refine / locate probes; the
// prefill of a decode: W windows x the initial tokens, cache rows w * G) on
// the same un-split "dec" GEMMs as the decode step: at <= 160 rows the tiled MFMA GEMM launches one or two row blocks and the
// 16-column skinny kernel 27 us per projection, while a dec launch finishes its outputs (bias / GELU / residual / K,V scatter,
// LayerNorm folded) in ~7 us.  Self-attention (causal over the new tokens), cross-attention and the alignment-head capture are
// the general kernels.  Leaves the raw residual stream in `x` (returns 0).
int decoder_forward_dec(swx_model *m, const FwdCfg &f, hipStream_t s)
{
    const swx_dims &D = m->dims;
    const int d = D.n_text_state, H = D.n_text_head;
    const size_t e = m->esz;
    const int R = f.W * f.rpw, rows = R * f.n_new;
    f16 *x = m->Wp<f16>(m->L.x), *q = m->Wp<f16>(m->L.qkv), *att = m->Wp<f16>(m->L.att), *u = m->Wp<f16>(m->L.u);
    float *slabs = m->Wp<float>(m->L.slabs);
    SWX_TRY(swx_embed(m->dtype, f.tokens, f.ld_tok, nullptr, f.pos0, R, f.n_new, m->arena + m->o_tok_emb,
                      m->A<float>(m->o_dec_pos), d, x, s));
    const int64_t chunk = xkv_chunk_elems(m);
    const bool pf_on = !(g_debug_flags & SWX_FLAG_NO_PREFETCH) && rows >= 32;       // the decode step's prefetch chain (decoder_step_dec)
    auto pf_of = [&](size_t w_off, int N, int K, int epi) {
        return pf_on ? swx_dec_prefetch_of(m->A<f16>(w_off), rows, N, K, epi) : DecPrefetch{};
    };
    for (int l = 0; l < D.n_text_layer; ++l) {
        const LayerW &w = m->dec[l];
        f16 *kc = (f16 *)(f.kcache + (size_t)l * f.layer_stride), *vc = (f16 *)(f.vcache + (size_t)l * f.layer_stride);
        DecGemmArgs g{};
        g.M = rows; g.tall = 1; g.A = x; g.lda = d; g.W = m->A<f16>(w.wqkv_f); g.ldw = d; g.N = 3 * d; g.K = d; g.epi = DEC_LN | DEC_QKV;
        g.c1 = m->A<float>(w.qkv_c1); g.c2 = m->A<float>(w.qkv_c2); g.C = q; g.ldc = d;
        g.kcache = kc; g.vcache = vc; g.pos0 = f.pos0; g.n_ctx = D.n_text_ctx; g.d = d; g.rps = f.n_new; g.row_mul = f.row_mul;
        g.pf = pf_of(w.wo_p, d, d, DEC_RES);
        SWX_TRY(swx_gemm_dec(g, s));
        SelfAttnArgs sa{};
        sa.qkv = q; sa.ldqkv = d; sa.kcache = kc; sa.vcache = vc; sa.anc = f.anc; sa.pos0 = f.pos0; sa.o = att; sa.ldo = d;
        sa.R = R; sa.n_new = f.n_new; sa.H = H; sa.n_ctx = D.n_text_ctx; sa.d = d; sa.skip_append = 1;
        sa.pos0_all_zero = f.pos0 == m->Wp<int32_t>(m->L.zeros_i32) ? 1 : 0;
        SWX_TRY(swx_self_attention(m->dtype, sa, f.row_mul, s));
        g = DecGemmArgs{};
        g.M = rows; g.tall = 1; g.A = att; g.lda = d; g.W = m->A<f16>(w.wo_p); g.ldw = d; g.N = d; g.K = d; g.epi = DEC_RES;
        g.c2 = m->A<float>(w.bo); g.X = x; g.ldx = d;
        g.pf = pf_of(w.wcq_f, d, d, DEC_LN);
        SWX_TRY(swx_gemm_dec(g, s));
        g = DecGemmArgs{};
        g.M = rows; g.tall = 1; g.A = x; g.lda = d; g.W = m->A<f16>(w.wcq_f); g.ldw = d; g.N = d; g.K = d; g.epi = DEC_LN;
        g.c1 = m->A<float>(w.cq_c1); g.c2 = m->A<float>(w.cq_c2); g.C = q; g.ldc = d;
        g.pf = pf_of(w.wco_p, d, d, DEC_RES);
        SWX_TRY(swx_gemm_dec(g, s));
        if (f.qcap) {
            hipError_t qe = hipMemcpyAsync(f.qcap + (size_t)l * rows * d * e, q, (size_t)rows * d * e, hipMemcpyDeviceToDevice, s);
            if (qe != hipSuccess) return -100 - (int)qe;
        }
        const unsigned char *kl = f.xkv + (size_t)l * f.W * chunk * e;
        AttnArgs ca{};
        ca.q = q; ca.ldq = d; ca.k = kl; ca.v = kl + (size_t)D.n_audio_ctx * d * e; ca.ldkv = d;
        ca.k_bs = chunk; ca.v_bs = chunk; ca.vt_kp = SWX_VT_KP; ca.o = att; ca.ldo = d;
        ca.B = f.W; ca.H = H; ca.nq = f.rpw * f.n_new; ca.nk = D.n_audio_ctx; ca.q_rows_per_batch = f.rpw * f.n_new;
        ca.kv_packed = kl + (size_t)xkv_plain_elems(D) * e;          // lets a small pass run as 16-row groups on the decode kernel
        SWX_TRY(swx_attention(m->dtype, ca, 0, s));
        if (f.capture && !m->heads_by_layer[l].empty()) {
            SWX_TRY(swx_qk_capture(m->dtype, q, d, f.rpw * f.n_new, f.cap_row0, f.cap_rows, kl, d, chunk, D.n_audio_ctx,
                                   m->Wp<int32_t>(m->L.heads) + m->head_slot0[l], (int)m->heads_by_layer[l].size(),
                                   m->head_slot0[l], m->n_align, f.W, m->Wp<float>(m->L.cap), f.cap_ld_n, D.n_audio_ctx, s));
        }
        g = DecGemmArgs{};
        g.M = rows; g.tall = 1; g.A = att; g.lda = d; g.W = m->A<f16>(w.wco_p); g.ldw = d; g.N = d; g.K = d; g.epi = DEC_RES;
        g.c2 = m->A<float>(w.bco); g.X = x; g.ldx = d;
        g.pf = pf_of(w.w1_f, 4 * d, d, DEC_LN | DEC_GELU);
        SWX_TRY(swx_gemm_dec(g, s));
        g = DecGemmArgs{};
        g.M = rows; g.tall = 1; g.A = x; g.lda = d; g.W = m->A<f16>(w.w1_f); g.ldw = d; g.N = 4 * d; g.K = d; g.epi = DEC_LN | DEC_GELU;
        g.c1 = m->A<float>(w.w1_c1); g.c2 = m->A<float>(w.w1_c2); g.C = u; g.ldc = 4 * d;
        g.pf = pf_of(w.w2_p, d, 4 * d, DEC_RES | DEC_SLAB);
        SWX_TRY(swx_gemm_dec(g, s));
        g = DecGemmArgs{};
        g.M = rows; g.tall = 1; g.A = u; g.lda = 4 * d; g.W = m->A<f16>(w.w2_p); g.ldw = 4 * d; g.N = d; g.K = 4 * d; g.epi = DEC_RES | DEC_SLAB;
        g.c2 = m->A<float>(w.b2); g.X = x; g.ldx = d; g.slabs = slabs; g.ticket = m->Wp<int>(m->L.ticket);
        if (l + 1 < D.n_text_layer) g.pf = pf_of(m->dec[l + 1].wqkv_f, 3 * d, d, DEC_LN | DEC_QKV);
        SWX_TRY(swx_gemm_dec(g, s));
    }
    return 0;
}

int decoder_forward(swx_model *m, const FwdCfg &f, hipStream_t s)
{
    const bool dec_ok = m->dtype == SWX_F16 && m->is_folded() && !(g_debug_flags & SWX_FLAG_NO_FAST_STEP);
    // Multi-token passes (scoring pass, prefill, refine / locate probes) take the dec GEMMs + the 16-row-group cross-attention at
    // EVERY size since round 4: which kernels compute a row must not depend on how many windows share the launch or on the longest
    // window of the batch -- window k of a 20-window batch is bit-identical to the same window alone
    // (tests/test_gpu_batch_invariance.py).  SWX_FLAG_SCORE_TILED restores round 3's dispatch (tiled GEMMs + flash attention
    // above 160 rows: ~1.5x faster per pass at 20 windows, different rounding points) for A/B runs.
    const int64_t rows_all = (int64_t)f.W * f.rpw * f.n_new;
    if (dec_ok && f.n_new > 1 && (rows_all <= 160 || !(g_debug_flags & SWX_FLAG_SCORE_TILED)) && rows_all <= m->L.rows_big &&
        swx_dec_slab_floats((int)rows_all, m->dims.n_text_state, 4 * m->dims.n_text_state) * 4 <= m->L.slab_bytes)
        return decoder_forward_dec(m, f, s);
    if (dec_ok && f.n_new == 1 && f.row_mul == 1 && !f.capture && f.rpw <= 16 && m->dims.n_audio_ctx >= 128 &&
        (size_t)f.W * f.rpw <= (size_t)m->L.rows_big)
        return decoder_step_dec(m, f, s);
    const swx_dims &D = m->dims;
    const int d = D.n_text_state, H = D.n_text_head;
    const size_t e = m->esz;
    const int R = f.W * f.rpw;
    const int rows = R * f.n_new;
    if (rows > m->L.rows_big) return -8;
    unsigned char *x = m->ws + m->L.x, *h = m->ws + m->L.h, *qkv = m->ws + m->L.qkv, *att = m->ws + m->L.att, *u = m->ws + m->L.u;
    SWX_TRY(swx_embed(m->dtype, f.tokens, f.ld_tok, nullptr, f.pos0, R, f.n_new, m->arena + m->o_tok_emb,
                      m->A<float>(m->o_dec_pos), d, x, s));
    // embed indexes rows by grid row; with row_mul > 1 the caller passes pre-strided token / position views
    for (int l = 0; l < D.n_text_layer; ++l) {
        const LayerW &w = m->dec[l];
        SWX_TRY(swx_layernorm(m->dtype, x, d, m->A<float>(w.ln1_g), m->A<float>(w.ln1_b), h, d, rows, d, s));
        SWX_TRY(swx_gemm(m->dtype, gemm_args(h, d, m->arena + w.wqkv, d, m->A<float>(w.bqkv), qkv, 3 * d, rows, 3 * d, d, EPI_BIAS), 0, s));
        SelfAttnArgs sa{};
        sa.qkv = qkv; sa.ldqkv = 3 * d;
        sa.kcache = f.kcache + (size_t)l * f.layer_stride;
        sa.vcache = f.vcache + (size_t)l * f.layer_stride;
        sa.anc = f.anc; sa.pos0 = f.pos0; sa.o = att; sa.ldo = d;
        sa.R = R; sa.n_new = f.n_new; sa.H = H; sa.n_ctx = D.n_text_ctx; sa.d = d;
        sa.pos0_all_zero = f.pos0 == m->Wp<int32_t>(m->L.zeros_i32) ? 1 : 0;
        SWX_TRY(swx_self_attention(m->dtype, sa, f.row_mul, s));
        SWX_TRY(gemm_residual(m, att, d, w.wo, w.bo, x, d, rows, d, d, s));
        // cross attention
        SWX_TRY(swx_layernorm(m->dtype, x, d, m->A<float>(w.lnx_g), m->A<float>(w.lnx_b), h, d, rows, d, s));
        SWX_TRY(swx_gemm(m->dtype, gemm_args(h, d, m->arena + w.wcq, d, m->A<float>(w.bcq), qkv, d, rows, d, d, EPI_BIAS), 0, s));
        if (f.qcap) {
            hipError_t qe = hipMemcpyAsync(f.qcap + (size_t)l * rows * d * e, qkv, (size_t)rows * d * e, hipMemcpyDeviceToDevice, s);
            if (qe != hipSuccess) return -100 - (int)qe;
        }
        // cross K/V of this layer: per window [K: 1500 x d row-major | V^T: d x SWX_VT_KP, keys contiguous]
        const int64_t chunk = xkv_chunk_elems(m);
        const unsigned char *kl = f.xkv + (size_t)l * f.W * chunk * e;
        AttnArgs ca{};
        ca.q = qkv; ca.ldq = d; ca.k = kl; ca.v = kl + (size_t)D.n_audio_ctx * d * e; ca.ldkv = d;
        ca.k_bs = chunk; ca.v_bs = chunk; ca.vt_kp = SWX_VT_KP; ca.o = att; ca.ldo = d;
        ca.B = f.W; ca.H = H; ca.nq = f.rpw * f.n_new; ca.nk = D.n_audio_ctx; ca.q_rows_per_batch = f.rpw * f.n_new;
        SWX_TRY(swx_attention(m->dtype, ca, 0, s));
        if (f.capture && !m->heads_by_layer[l].empty()) {
            SWX_TRY(swx_qk_capture(m->dtype, qkv, d, f.rpw * f.n_new, f.cap_row0, f.cap_rows, kl, d, chunk, D.n_audio_ctx,
                                   m->Wp<int32_t>(m->L.heads) + m->head_slot0[l], (int)m->heads_by_layer[l].size(),
                                   m->head_slot0[l], m->n_align, f.W, m->Wp<float>(m->L.cap), f.cap_ld_n, D.n_audio_ctx, s));
        }
        SWX_TRY(gemm_residual(m, att, d, w.wco, w.bco, x, d, rows, d, d, s));
        // MLP
        SWX_TRY(swx_layernorm(m->dtype, x, d, m->A<float>(w.ln2_g), m->A<float>(w.ln2_b), h, d, rows, d, s));
        SWX_TRY(swx_gemm(m->dtype, gemm_args(h, d, m->arena + w.w1, d, m->A<float>(w.b1), u, 4 * d, rows, 4 * d, d, EPI_BIAS | EPI_GELU), 0, s));
        SWX_TRY(gemm_residual(m, u, 4 * d, w.w2, w.b2, x, d, rows, d, 4 * d, s));
    }
    return 0;
}

int logits_gemm(swx_model *m, const void *hid, int64_t ld, int rows, float *out, hipStream_t s)
{
    const swx_dims &D = m->dims;
    return swx_gemm(m->dtype, gemm_args(hid, ld, m->arena + m->o_tok_emb, D.n_text_state, nullptr, out, D.n_vocab, rows,
                                        D.n_vocab, D.n_text_state, EPI_OUT_F32), 0, s);
}


// The captured two-step graph of a decode job, from the handle's cache or captured now.  `record` enqueues the two units on the
// stream it is given.  The key lists everything a kernel argument of the step depends on: the decode configuration, the
// buffers (workspace / arena / cross-K/V / timestamp mask), the switches.  Buffer CONTENTS are read at run time.
template <typename F>
swx_model::StepGraph *step_graph(swx_model *m, const DecodeBufs &b, const void *d_xkv, int cur, F record)
{
    const swx_decode_cfg &c = b.cfg;
    uint32_t tbits, pbits;
    memcpy(&tbits, &c.temperature, 4); memcpy(&pbits, &c.patience, 4);
    std::vector<uint64_t> key = {
        (uint64_t)c.n_windows, (uint64_t)c.n_group, (uint64_t)c.beam, tbits, pbits, (uint64_t)c.sample_len, (uint64_t)c.sample_begin,
        (uint64_t)c.sot_index, (uint64_t)c.suppress_blank, (uint64_t)c.apply_timestamp_rules, (uint64_t)(int64_t)c.max_initial_timestamp_index,
        (uint64_t)c.eot, (uint64_t)c.sot, (uint64_t)(int64_t)c.no_timestamps, (uint64_t)c.timestamp_begin, (uint64_t)(int64_t)c.no_speech,
        (uint64_t)(int64_t)c.blank_token, (uint64_t)c.n_suppress, (uint64_t)c.min_tokens, (uint64_t)c.seed,
        (uint64_t)(uintptr_t)b.ts_mask, (uint64_t)(uintptr_t)b.win_uid, (uint64_t)(uintptr_t)d_xkv, (uint64_t)(uintptr_t)m->arena,
        (uint64_t)(uintptr_t)m->ws, (uint64_t)m->ws_bytes, (uint64_t)g_debug_flags, (uint64_t)cur, (uint64_t)m->is_folded(),
        (uint64_t)(uintptr_t)c.noise};
    ++m->graph_clock;
    for (auto &g : m->graphs) if (g.key == key) { g.used = m->graph_clock; return &g; }
    if (!m->cap_stream && hipStreamCreateWithFlags(&m->cap_stream, hipStreamNonBlocking) != hipSuccess) { m->cap_stream = nullptr; return nullptr; }
    swx_model::StepGraph ng;
    if (hipStreamBeginCapture(m->cap_stream, hipStreamCaptureModeThreadLocal) != hipSuccess) return nullptr;
    const int rc = record(m->cap_stream);
    hipError_t er = hipStreamEndCapture(m->cap_stream, &ng.graph);
    if (rc < 0 || er != hipSuccess || !ng.graph) { if (ng.graph) (void)hipGraphDestroy(ng.graph); return nullptr; }
    if (hipGraphInstantiate(&ng.exec, ng.graph, nullptr, nullptr, 0) != hipSuccess) { (void)hipGraphDestroy(ng.graph); return nullptr; }
    ng.key = std::move(key); ng.used = m->graph_clock;
    ++m->n_captures;
    constexpr size_t MAX_GRAPHS = 8;
    if (m->graphs.size() >= MAX_GRAPHS) {       // least recently used entry makes room
        size_t lru = 0;
        for (size_t i = 1; i < m->graphs.size(); ++i) if (m->graphs[i].used < m->graphs[lru].used) lru = i;
        (void)hipGraphExecDestroy(m->graphs[lru].exec); (void)hipGraphDestroy(m->graphs[lru].graph);
        m->graphs[lru] = std::move(ng);
        return &m->graphs[lru];
    }
    m->graphs.push_back(std::move(ng));
    return &m->graphs.back();
}

__global__ __launch_bounds__(256) void token_prob_kernel(const float *__restrict__ logits, int V, int eot,
                                                         const int32_t *__restrict__ target_tok, float *__restrict__ out)
{
    // one block per row: softmax(logits[row][:eot])[target]
    __shared__ float sh[4];
    const int row = blockIdx.x, tid = threadIdx.x;
    const float *lg = logits + (size_t)row * V;
    float mx = -__builtin_inff();
    for (int i = tid; i < eot; i += 256) mx = fmaxf(mx, lg[i]);
    mx = wave_max(mx);
    if ((tid & 63) == 0) sh[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
    __syncthreads();
    float sum = 0.f;
    for (int i = tid; i < eot; i += 256) sum += expf(lg[i] - mx);
    sum = wave_sum(sum);
    if ((tid & 63) == 0) sh[tid >> 6] = sum;
    __syncthreads();
    sum = sh[0] + sh[1] + sh[2] + sh[3];
    if (tid == 0) {
        const int t = target_tok[row];
        out[row] = (t >= 0 && t < eot) ? expf(lg[t] - mx) / sum : 0.f;
    }
}

__global__ void score_targets_kernel(const int32_t *__restrict__ tokens, int max_n, int n_sot, int rows_per_w,
                                     int32_t *__restrict__ targets, int total)
{
    // row (w, i) of the probability pass predicts tokens[w][n_sot + 1 + i]
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int w = idx / rows_per_w, i = idx % rows_per_w;
    const int p = n_sot + 1 + i;
    targets[idx] = (p < max_n) ? tokens[(size_t)w * max_n + p] : -1;
}

}  // namespace

bool swx_prof_on() { return g_prof_enabled; }
int swx_flags() { return g_debug_flags; }
void swx_prof_begin(int cls, double work, hipStream_t s)
{
    ProfRec r{cls, work, prof_event(), prof_event()};
    if (!r.a || !r.b) return;
    (void)hipEventRecord(r.a, s);
    g_prof.push_back(r);
}
void swx_prof_end(hipStream_t s)
{
    if (!g_prof.empty()) (void)hipEventRecord(g_prof.back().b, s);
}

// ================================================================================================== C ABI
// lane-exchange helpers vs __shfl_xor (swx_test_lane_xor)
__global__ __launch_bounds__(64) void lane_xor_check_kernel(const uint32_t *__restrict__ in, uint32_t *__restrict__ out)
{
    const int lane = threadIdx.x;
    const uint32_t x = in[blockIdx.x * 64 + lane];
    uint32_t *o = out + (size_t)blockIdx.x * 13 * 64 + lane;
    o[0 * 64] = lane_xor_u32<32>(x, lane); o[1 * 64] = lane_xor_u32<16>(x, lane); o[2 * 64] = lane_xor_u32<8>(x, lane);
    o[3 * 64] = lane_xor_u32<4>(x, lane);  o[4 * 64] = lane_xor_u32<2>(x, lane);  o[5 * 64] = lane_xor_u32<1>(x, lane);
    o[6 * 64] = (uint32_t)__shfl_xor((int)x, 32, 64); o[7 * 64] = (uint32_t)__shfl_xor((int)x, 16, 64);
    o[8 * 64] = (uint32_t)__shfl_xor((int)x, 8, 64);  o[9 * 64] = (uint32_t)__shfl_xor((int)x, 4, 64);
    o[10 * 64] = (uint32_t)__shfl_xor((int)x, 2, 64); o[11 * 64] = (uint32_t)__shfl_xor((int)x, 1, 64);
    double v = (double)__uint_as_float(x);
    double ref = v;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ref += __shfl_xor(ref, off, 64);
    const double got = wave_sum_d(v);
    unsigned ok = (__double_as_longlong(got) == __double_as_longlong(ref)) ? 1u : 0u;
    // the select-free forms for commutative operations, on finite values
    const float f = (float)(int)(x >> 9) * 1.1920929e-7f - 0.37f;
    ok |= (__float_as_uint(lane_xor16_max(f)) == __float_as_uint(fmaxf(f, __shfl_xor(f, 16, 64)))) ? 2u : 0u;
    ok |= (__float_as_uint(lane_xor32_max(f)) == __float_as_uint(fmaxf(f, __shfl_xor(f, 32, 64)))) ? 4u : 0u;
    ok |= (__float_as_uint(lane_xor16_add(f)) == __float_as_uint(f + __shfl_xor(f, 16, 64))) ? 8u : 0u;
    ok |= (__float_as_uint(lane_xor32_add(f)) == __float_as_uint(f + __shfl_xor(f, 32, 64))) ? 16u : 0u;
    float wm = f, ws = f;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { wm = fmaxf(wm, __shfl_xor(wm, off, 64)); ws += __shfl_xor(ws, off, 64); }
    ok |= (__float_as_uint(wave_max(f)) == __float_as_uint(wm)) ? 32u : 0u;
    ok |= (__float_as_uint(wave_sum(f)) == __float_as_uint(ws)) ? 64u : 0u;
    o[12 * 64] = ok;
}

// gelu_erf2 (packed, both sides of erff's branch) vs gelu_erf (the device library's erff) over EVERY f32 bit pattern (swx_test_gelu_pair)
__global__ __launch_bounds__(256) void gelu_pair_check_kernel(unsigned long long *__restrict__ out)
{
    const unsigned base = (blockIdx.x * 256u + threadIdx.x) * 32u;          // 2^27 threads x 32 values = 2^32
    unsigned bad = 0, nan_bits = 0, first = 0xffffffffu;
#pragma unroll 4
    for (unsigned j = 0; j < 32; j += 2) {
        const unsigned u0 = base + j, u1 = base + j + 1;
        const f32x2 x = {__uint_as_float(u0), __uint_as_float(u1)};
        const f32x2 got = gelu_erf2(x);
        const float r0 = gelu_erf(x[0]), r1 = gelu_erf(x[1]);
        const unsigned g0 = __float_as_uint(got[0]), g1 = __float_as_uint(got[1]), e0 = __float_as_uint(r0), e1 = __float_as_uint(r1);
        if (g0 != e0) { if (r0 != r0 && got[0] != got[0]) ++nan_bits; else { ++bad; first = first < u0 ? first : u0; } }
        if (g1 != e1) { if (r1 != r1 && got[1] != got[1]) ++nan_bits; else { ++bad; first = first < u1 ? first : u1; } }
    }
    if (bad) { atomicAdd(out, (unsigned long long)bad); atomicMin(out + 1, (unsigned long long)first); }
    if (nan_bits) atomicAdd(out + 2, (unsigned long long)nan_bits);
}

extern "C" {

int swx_debug_flags(int flags)
{
    const int old = g_debug_flags;
    if (flags >= 0) g_debug_flags = flags;
    return old;
}

int swx_prof_enable(int on)
{
    g_prof_enabled = on != 0;
    if (on) { g_prof.clear(); g_pool_next = 0; }
    return 0;
}

// out[cls*3 + {0,1,2}] = {launches, total milliseconds, total work}; synchronises the device
int swx_prof_collect(double *out, int n_classes)
{
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) return -100 - (int)e;
    for (int i = 0; i < n_classes * 3; ++i) out[i] = 0.0;
    for (auto &r : g_prof) {
        if (r.cls < 0 || r.cls >= n_classes) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) continue;
        out[r.cls * 3 + 0] += 1.0;
        out[r.cls * 3 + 1] += (double)ms;
        out[r.cls * 3 + 2] += r.work;
    }
    g_prof.clear();
    g_pool_next = 0;
    (void)hipGetLastError();      // a failed elapsed-time query must not surface later as somebody else's launch error
    return PC_COUNT;
}

const char *swx_strerror(int code)
{
    switch (code) {
        case 0: return "ok";
        case -1: return "invalid argument";
        case -2: return "size out of range";
        case -3: return "unsupported filter width";
        case -4: return "gemm shape/alignment not supported";
        case -5: return "attention shape not supported";
        case -7: return "not implemented";
        case -8: return "workspace too small for this call";
        case -9: return "weights / workspace not bound";
        case -10: return "unknown tensor name";
        case -11: return "tensor size mismatch";
        case -20: return "not a FLAC stream";
        case -21: return "corrupt FLAC stream (bad header, reserved code or CRC mismatch)";
        case -22: return "unsupported FLAC stream";
        case -23: return "truncated FLAC stream";
        default: return code <= -100 ? "HIP runtime error (code = -100 - hipError_t)" : "unknown error";
    }
}

int swx_version(void) { return 1; }

int swx_model_create(const swx_dims *dims, int dtype, swx_model **out)
{
    if (!dims || !out || (dtype != SWX_F32 && dtype != SWX_F16)) return -1;
    if (dims->n_audio_state % 64 || dims->n_text_state % 64) return -1;
    if (dims->n_audio_state / dims->n_audio_head != 64 || dims->n_text_state / dims->n_text_head != 64) return -1;
    if (dims->n_text_ctx > 448 || dims->n_audio_ctx > 1500) return -2;
    swx_model *m = new swx_model();
    m->dims = *dims;
    m->dtype = dtype;
    m->esz = dtype == SWX_F16 ? 2 : 4;
    build_layout(m);
    default_alignment_heads(m);
    upload_heads(m);
    *out = m;
    return 0;
}

void swx_model_destroy(swx_model *m) { delete m; }

size_t swx_weights_bytes(const swx_model *m) { return m ? m->arena_bytes : 0; }

int swx_bind_weights(swx_model *m, void *d_arena, size_t bytes)
{
    if (m) m->drop_graphs();          // captured steps hold the old pointers
    if (!m || !d_arena || bytes < m->arena_bytes) return -1;
    m->arena = (unsigned char *)d_arena;
    hipError_t e = hipMemset(d_arena, 0, m->arena_bytes);
    if (e != hipSuccess) return -100 - (int)e;
    std::vector<double2> tw(SWX_N_FFT);
    for (int i = 0; i < SWX_N_FFT; ++i) {
        const double a = 2.0 * M_PI * (double)i / (double)SWX_N_FFT;
        tw[i].x = cos(a); tw[i].y = sin(a);
    }
    e = hipMemcpy(m->arena + m->o_twiddle, tw.data(), sizeof(double2) * SWX_N_FFT, hipMemcpyHostToDevice);
    if (e != hipSuccess) return -100 - (int)e;
    for (auto &kv : m->slots) kv.second.loaded = false;
    m->fold_state = std::make_shared<bool>(false);       // a new arena: a state of its own (views of the old one keep theirs)
    return upload_heads(m);
}

int swx_share_weights(swx_model *m, const swx_model *owner)
{
    // a second handle on an arena that is already populated: nothing is cleared or re-initialised (swx_bind_weights
    // zeroes the arena), the bookkeeping of which tensors are present is taken over from the owner
    if (!m || !owner || !owner->arena || m == owner) return -1;
    if (m->dtype != owner->dtype || m->arena_bytes != owner->arena_bytes || memcmp(&m->dims, &owner->dims, sizeof(swx_dims)) != 0)
        return -1;
    m->arena = owner->arena;
    m->fold_state = owner->fold_state;
    for (auto &kv : m->slots) {
        auto it = owner->slots.find(kv.first);
        kv.second.loaded = it != owner->slots.end() && it->second.loaded;
    }
    return upload_heads(m);
}

int swx_load_tensor(swx_model *m, const char *name, const float *d_src, int64_t numel, void *stream)
{
    if (!m || !m->arena) return -9;
    auto it = m->slots.find(name);
    if (it == m->slots.end()) return -10;
    Slot &sl = it->second;
    hipStream_t s = S(stream);
    if (sl.kind == SK_MAT) {
        if (numel != sl.rows * sl.cols) return -11;
        SWX_TRY(swx_copy_rows(m->dtype, d_src, sl.cols, m->arena + sl.off, sl.dst_ld, sl.rows, sl.cols, s));
    } else if (sl.kind == SK_CONV) {
        if (numel != sl.rows * sl.cols * 3) return -11;
        SWX_TRY(swx_copy_conv_w(m->dtype, d_src, (int)sl.rows, (int)sl.cols, (int)sl.dst_ld, m->arena + sl.off, s));
    } else {
        if (numel != sl.cols) return -11;
        hipError_t e = hipMemcpyAsync(m->arena + sl.off, d_src, (size_t)numel * 4, hipMemcpyDeviceToDevice, s);
        if (e != hipSuccess) return -100 - (int)e;
    }
    sl.loaded = true;
    *m->fold_state = false;   // the folded copies are stale until the next swx_weights_finalize (for every view of this arena)
    return 0;
}

int swx_weights_complete(const swx_model *m)
{
    if (!m) return 0;
    for (auto &kv : m->slots) if (!kv.second.loaded) return 0;
    return 1;
}

int swx_weights_mark_loaded(swx_model *m)
{
    // the arena was filled from outside (a peer's packed arena received over RCCL): every slot counts as present
    if (!m || !m->arena) return -9;
    for (auto &kv : m->slots) kv.second.loaded = true;
    return 0;
}

int swx_weights_finalize(swx_model *m, void *stream)
{
    if (!m || !m->arena) return -9;
    if (!m->has_fold) return 0;
    for (auto &kv : m->slots) if (!kv.second.loaded) return -9;
    hipStream_t s = S(stream);
    const int d = m->dims.n_text_state;
    for (auto &w : m->dec) {
        SWX_TRY(swx_fold_ln(m->arena + w.wqkv, m->A<float>(w.ln1_g), m->A<float>(w.ln1_b), m->A<float>(w.bqkv), m->arena + w.wqkv_f,
                            m->A<float>(w.qkv_c1), m->A<float>(w.qkv_c2), 3 * d, d, s));
        SWX_TRY(swx_fold_ln(m->arena + w.wcq, m->A<float>(w.lnx_g), m->A<float>(w.lnx_b), m->A<float>(w.bcq), m->arena + w.wcq_f,
                            m->A<float>(w.cq_c1), m->A<float>(w.cq_c2), d, d, s));
        SWX_TRY(swx_fold_ln(m->arena + w.w1, m->A<float>(w.ln2_g), m->A<float>(w.ln2_b), m->A<float>(w.b1), m->arena + w.w1_f,
                            m->A<float>(w.w1_c1), m->A<float>(w.w1_c2), 4 * d, d, s));
        SWX_TRY(swx_fold_ln(m->arena + w.wo, nullptr, nullptr, nullptr, m->arena + w.wo_p, nullptr, nullptr, d, d, s));
        SWX_TRY(swx_fold_ln(m->arena + w.wco, nullptr, nullptr, nullptr, m->arena + w.wco_p, nullptr, nullptr, d, d, s));
        SWX_TRY(swx_fold_ln(m->arena + w.w2, nullptr, nullptr, nullptr, m->arena + w.w2_p, nullptr, nullptr, d, 4 * d, s));
    }
    *m->fold_state = true;
    return 0;
}

int swx_missing_tensor(const swx_model *m, int index, char *buf, int buflen)
{
    int k = 0;
    for (auto &kv : m->slots)
        if (!kv.second.loaded) {
            if (k == index) { snprintf(buf, buflen, "%s", kv.first.c_str()); return 1; }
            ++k;
        }
    return 0;
}

int swx_set_alignment_heads(swx_model *m, const int32_t *pairs, int n_pairs)
{
    if (!m || (n_pairs > 0 && !pairs)) return -1;
    const swx_dims &D = m->dims;
    std::vector<std::vector<int>> by(D.n_text_layer);
    for (int i = 0; i < n_pairs; ++i) {
        const int l = pairs[2 * i], h = pairs[2 * i + 1];
        if (l < 0 || l >= D.n_text_layer || h < 0 || h >= D.n_text_head) return -1;
        by[l].push_back(h);
    }
    m->heads_by_layer = by;
    return upload_heads(m);
}

int swx_num_alignment_heads(const swx_model *m) { return m ? m->n_align : 0; }

size_t swx_workspace_bytes(const swx_model *m, int max_windows, int max_rows)
{
    if (!m || max_windows <= 0 || max_rows <= 0) return 0;
    swx_model::WsLayout L;
    ws_layout(m, max_windows, max_rows, m->n_align, L);
    return L.total;
}

int swx_bind_workspace(swx_model *m, void *d_ws, size_t bytes, int max_windows, int max_rows)
{
    if (m) m->drop_graphs();
    if (!m || !d_ws || max_windows <= 0 || max_rows <= 0) return -1;
    swx_model::WsLayout L;
    ws_layout(m, max_windows, max_rows, m->n_align, L);
    if (bytes < L.total) return -8;
    m->ws = (unsigned char *)d_ws;
    m->ws_bytes = bytes;
    m->max_windows = max_windows;
    m->max_rows = max_rows;
    m->ws_n_align = m->n_align;
    m->L = L;
    hipError_t e = hipMemset(m->ws + L.zeros_i32, 0, (size_t)(max_rows > max_windows ? max_rows : max_windows) * 4 + 256);
    if (e == hipSuccess) e = hipMemset(m->ws + L.ticket, 0, (size_t)SWX_DEC_TICKETS * 4);
    if (e != hipSuccess) return -100 - (int)e;
    if (!m->heads_flat.empty()) {
        e = hipMemcpy(m->ws + L.heads, m->heads_flat.data(), m->heads_flat.size() * 4, hipMemcpyHostToDevice);
        if (e != hipSuccess) return -100 - (int)e;
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------------- mel
int swx_log_mel(swx_model *m, const float *d_pcm, int B, float *d_mel, int per_item_max, void *stream)
{
    if (!m || !m->arena || !m->ws) return -9;
    if (B > m->max_windows) return -8;
    return swx_mel_launch(d_pcm, B, m->A<float>(m->o_hann), m->A<double2>(m->o_twiddle), m->A<float>(m->o_filters),
                          m->dims.n_mels, d_mel, m->Wp<unsigned>(m->L.gmax), per_item_max, S(stream));
}

int swx_log_mel_ragged(swx_model *m, const float *d_pcm, const int32_t *n_valid, const int32_t *n_total, int B,
                       float *d_mel, int per_item_max, void *stream)
{
    if (!m || !m->arena || !m->ws) return -9;
    if (B <= 0) return 0;
    if (B > m->max_windows) return -8;
    if (!n_valid || !n_total) return -2;
    std::vector<int32_t> lens((size_t)B * 2);
    for (int b = 0; b < B; ++b) {
        // torch.stft's reflect padding needs more than n_fft/2 samples; one extra block row of 8 frames is launched
        if (n_total[b] <= 200 || n_total[b] / 160 > 3008 || n_valid[b] < 0 || n_valid[b] > n_total[b] ||
            n_valid[b] > 480000)
            return -2;
        lens[2 * b] = n_valid[b];
        lens[2 * b + 1] = n_total[b];
    }
    hipStream_t s = S(stream);
    int32_t *d_lens = m->Wp<int32_t>(m->L.h1);          // encoder scratch, idle while the spectrogram is computed
    hipError_t e = hipMemcpyAsync(d_lens, lens.data(), lens.size() * 4, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);   // lens is a stack-lifetime staging buffer
    if (e != hipSuccess) return -100 - (int)e;
    return swx_mel_ragged_launch(d_pcm, d_lens, B, m->A<float>(m->o_hann), m->A<double2>(m->o_twiddle),
                                 m->A<float>(m->o_filters), m->dims.n_mels, d_mel, m->Wp<unsigned>(m->L.gmax), per_item_max,
                                 s);
}

// ----------------------------------------------------------------------------------------------------- encoder
int swx_encode(swx_model *m, const float *d_mel, int B, void *d_xa, void *stream)
{
    if (!m || !m->arena || !m->ws) return -9;
    if (B <= 0) return 0;
    if (B > m->max_windows) return -8;
    const swx_dims &D = m->dims;
    const int d = D.n_audio_state, H = D.n_audio_head, S_ = D.n_audio_ctx;
    const size_t e = m->esz;
    hipStream_t s = S(stream);
    unsigned char *melT = m->ws + m->L.melT, *h1 = m->ws + m->L.h1;
    unsigned char *x = m->ws + m->L.x, *h = m->ws + m->L.h, *qkv = m->ws + m->L.qkv, *att = m->ws + m->L.att, *u = m->ws + m->L.u;
    const int rows = B * S_;
    if (D.n_audio_ctx != 1500) return -2;

    SWX_TRY(swx_mel_transpose(m->dtype, d_mel, B, D.n_mels, m->Cp, melT, s));
    // conv1 (k=3, pad 1) as a GEMM over overlapping rows of the channel-last mel: A row t = melT[t .. t+2][:]
    SWX_TRY(swx_fill_zero(h1, (size_t)B * 3002 * d * e, s));
    for (int b = 0; b < B; ++b) {
        const unsigned char *a = melT + (size_t)b * 3002 * m->Cp * e;
        unsigned char *c = h1 + ((size_t)b * 3002 + 1) * d * e;
        SWX_TRY(swx_gemm(m->dtype, gemm_args(a, m->Cp, m->arena + m->o_conv1_w, 3 * m->Cp, m->A<float>(m->o_conv1_b), c, d,
                                             3000, d, 3 * m->Cp, EPI_BIAS | EPI_GELU), 0, s));
    }
    // conv2 (k=3, stride 2, pad 1): A row t' = h1[2t' .. 2t'+2][:] (padded row index), then + positional embedding
    for (int b = 0; b < B; ++b) {
        const unsigned char *a = h1 + (size_t)b * 3002 * d * e;
        GemmArgs g = gemm_args(a, 2 * d, m->arena + m->o_conv2_w, 3 * d, m->A<float>(m->o_conv2_b), x + (size_t)b * S_ * d * e, d,
                               S_, d, 3 * d, EPI_BIAS | EPI_GELU | EPI_RESF32MOD);
        g.Rf = m->A<float>(m->o_enc_pos); g.res_mod = S_;
        SWX_TRY(swx_gemm(m->dtype, g, 0, s));
    }
    for (int l = 0; l < D.n_audio_layer; ++l) {
        const LayerW &w = m->enc[l];
        SWX_TRY(swx_layernorm(m->dtype, x, d, m->A<float>(w.ln1_g), m->A<float>(w.ln1_b), h, d, rows, d, s));
        // f16 at few windows (wherever the launch does not go to the 256 x 256 kernel, whose epilogue is plain): V leaves the
        // projection's epilogue transposed per head (EPI_QKV_VT, round 6) instead of by a launch of its own -- 5 us x 32 layers per
        // window pass of align(); the `u` buffer (MLP hidden, 4d wide) is free until the MLP of this layer
        const bool fuse_vt = m->dtype == SWX_F16 && d % 128 == 0 && S_ % 4 == 0 && !(g_debug_flags & SWX_FLAG_QKV_SEPARATE_VT) &&
                             swx_gemm_plan_f16(rows, 3 * d, d, EPI_BIAS, 3 * d, 1, true, 0, g_debug_flags) != SWX_GEMM_BIG;
        {
            GemmArgs gq = gemm_args(h, d, m->arena + w.wqkv, d, m->A<float>(w.bqkv), qkv, 3 * d, rows, 3 * d, d, EPI_BIAS | (fuse_vt ? EPI_QKV_VT : 0));
            if (fuse_vt) { gq.C2 = u; gq.vt_s = S_; gq.vt_kp = SWX_VT_KP; gq.vt_bs = (int64_t)H * 64 * SWX_VT_KP; gq.vt_zero_pad = 1; }
            SWX_TRY(swx_gemm(m->dtype, gq, 0, s));
        }
        AttnArgs a{};
        a.q = qkv; a.ldq = 3 * d; a.k = qkv + (size_t)d * e; a.v = qkv + (size_t)2 * d * e; a.ldkv = 3 * d; a.o = att; a.ldo = d;
        a.k_bs = (int64_t)S_ * 3 * d; a.v_bs = a.k_bs; a.vt_kp = 0;
        a.B = B; a.H = H; a.nq = S_; a.nk = S_; a.q_rows_per_batch = S_;
        if (m->dtype == SWX_F16) {
            // V of this layer transposed per head (one streaming pass, ~35 us for 20 windows): the flash kernel then stages both
            // operands with 16-byte vector stores; the `u` buffer (MLP hidden, 4d wide) is free until the MLP of this layer
            if (!fuse_vt) SWX_TRY(swx_transpose_v(a.v, 3 * d, a.v_bs, S_, u, SWX_VT_KP, (int64_t)H * 64 * SWX_VT_KP, B, H, s));
            a.v = u; a.vt_kp = SWX_VT_KP; a.v_bs = (int64_t)H * 64 * SWX_VT_KP;
        }
        SWX_TRY(swx_attention(m->dtype, a, 0, s));
        SWX_TRY(gemm_residual(m, att, d, w.wo, w.bo, x, d, rows, d, d, s));
        SWX_TRY(swx_layernorm(m->dtype, x, d, m->A<float>(w.ln2_g), m->A<float>(w.ln2_b), h, d, rows, d, s));
        SWX_TRY(swx_gemm(m->dtype, gemm_args(h, d, m->arena + w.w1, d, m->A<float>(w.b1), u, 4 * d, rows, 4 * d, d, EPI_BIAS | EPI_GELU), 0, s));
        SWX_TRY(gemm_residual(m, u, 4 * d, w.w2, w.b2, x, d, rows, d, 4 * d, s));
    }
    SWX_TRY(swx_layernorm(m->dtype, x, d, m->A<float>(m->o_lnpost_g), m->A<float>(m->o_lnpost_b), d_xa, d, rows, d, s));
    return 0;
}

size_t swx_cross_kv_bytes(const swx_model *m, int B)
{
    if (!m) return 0;
    return (size_t)m->dims.n_text_layer * B * (size_t)xkv_chunk_elems(m) * m->esz;
}

int swx_cross_kv(swx_model *m, const void *d_xa, int B, void *d_xkv, void *stream)
{
    if (!m || !m->arena) return -9;
    const swx_dims &D = m->dims;
    const int d = D.n_text_state, S_ = D.n_audio_ctx;
    const size_t e = m->esz;
    if (D.n_audio_state != d) return -1;
    const int64_t chunk = xkv_chunk_elems(m);
    hipStream_t s = S(stream);
    // the key padding of V^T (columns S_..KP of each of the d rows of every layer's / window's V^T block) must be finite: it is
    // multiplied by exact-zero probabilities.  Everything else in the buffer is written below (K rows, V^T columns, packed copy).
    SWX_TRY(swx_pad_zero((unsigned char *)d_xkv + (size_t)S_ * d * e, (int64_t)SWX_VT_KP * e, chunk * (int64_t)e, (int)(S_ * e),
                         (int)((SWX_VT_KP - S_) * e), d, D.n_text_layer * B, s));
    for (int l = 0; l < D.n_text_layer; ++l) {
        const LayerW &w = m->dec[l];
        unsigned char *base = (unsigned char *)d_xkv + (size_t)l * B * chunk * e;
        const bool one_launch = m->dtype == SWX_F16 && d % 128 == 0 && (int64_t)B * S_ > 128 && !(g_debug_flags & SWX_FLAG_XKV_TWO_LAUNCHES);
        bool packed_by_epilogue = false;
        if (one_launch) {
            // round 6: K and V of the layer from ONE launch over the fused weight rows [2d][d] (EPI_KV: a tile left of column d stores
            // K rows per window, a tile right of it V transposed per head) -- at one window 240 tiles of 128 x 128 on the ring kernel
            // instead of two launches of 240 tiles of 128 x 64; per element the same MFMA sequence, so bit-identical
            GemmArgs gkv = gemm_args(d_xa, d, m->arena + w.wckv, d, m->A<float>(w.bckv), base, d, B * S_, 2 * d, d, EPI_BIAS | EPI_KV);
            gkv.vt_s = S_; gkv.vt_kp = SWX_VT_KP; gkv.vt_bs = chunk; gkv.C2 = base + (size_t)S_ * d * e;
            if (!(g_debug_flags & SWX_FLAG_XKV_PACK_SEPARATE)) {
                // ... and the fragment-ordered copy from the same epilogue (round 6: swx_xkv_pack read and wrote the layer's K / V^T once more)
                gkv.P = base + (size_t)xkv_plain_elems(D) * e; gkv.p_bs = chunk; gkv.p_nkpad = ((S_ + 31) / 32) * 32; gkv.p_vcol0 = d;
                packed_by_epilogue = true;
            }
            SWX_TRY(swx_gemm(m->dtype, gkv, 0, s));
        } else {
        // K (no bias upstream; the fused bias slot is zero): one GEMM over all windows, rows scattered per window chunk
        GemmArgs gk = gemm_args(d_xa, d, m->arena + w.wckv, d, m->A<float>(w.bckv), base, d, B * S_, d, d, EPI_BIAS | EPI_CBATCH);
        gk.vt_s = S_; gk.vt_kp = 0; gk.vt_bs = chunk;
        SWX_TRY(swx_gemm(m->dtype, gk, 0, s));
        // V, stored transposed per head behind each window's K
        GemmArgs gv = gemm_args(d_xa, d, m->arena + w.wckv + (size_t)d * d * e, d, m->A<float>(w.bckv) + d,
                                base + (size_t)S_ * d * e, d, B * S_, d, d, EPI_BIAS | EPI_STORE_VT);
        gv.vt_s = S_; gv.vt_kp = SWX_VT_KP; gv.vt_bs = chunk;
        SWX_TRY(swx_gemm(m->dtype, gv, 0, s));
        }
        if (m->dtype == SWX_F16 && !packed_by_epilogue)      // the decode-step cross-attention streams the fragment-ordered copy
            SWX_TRY(swx_xkv_pack(base, base + (size_t)S_ * d * e, base + (size_t)xkv_plain_elems(D) * e, B, D.n_text_head, S_, d,
                                 SWX_VT_KP, chunk, s));
    }
    return 0;
}

// ----------------------------------------------------------------------------------------------------- decode
int swx_decode_gout(const swx_decode_cfg *cfg)
{
    if (!cfg) return 0;
    if (!cfg->beam) return cfg->n_group;
    const float pat = cfg->patience > 0.f ? cfg->patience : 1.0f;
    const int mc = (int)lrintf((float)cfg->n_group * pat);   // Python round(): half-to-even, as lrintf in the default mode
    return mc > cfg->n_group ? mc : cfg->n_group;
}

int swx_decode(swx_model *m, const swx_decode_cfg *cfg, const int32_t *d_init_tokens, const int32_t *d_suppress,
               const uint8_t *d_ts_mask, const void *d_xkv, int32_t *d_tokens_out, int32_t *d_lens_out,
               float *d_sumlp_out, float *d_nospeech, void *stream)
{
    if (!m || !m->arena || !m->ws) return -9;
    if (!cfg || !d_init_tokens || !d_xkv) return -1;
    const swx_dims &D = m->dims;
    const int W = cfg->n_windows, G = cfg->n_group, M = W * G;
    if (W <= 0 || G <= 0 || G > MAX_GROUP) return -1;
    if (W > m->max_windows || M > m->max_rows) return -8;
    if (cfg->n_suppress > MAX_SUPPRESS || cfg->n_suppress < 0) return -2;
    const int n_init = cfg->sample_begin;
    if (n_init <= 0 || n_init > D.n_text_ctx) return -1;
    hipStream_t s = S(stream);
    const int d = D.n_text_state;
    const size_t e = m->esz;

    DecodeBufs b{};
    b.cfg = *cfg;
    if (cfg->beam || cfg->temperature == 0.f) b.cfg.noise = nullptr;      // only the sampling decoder draws
    b.W = W; b.G = G; b.M = M; b.V = D.n_vocab; b.TS = D.n_text_ctx + 1; b.n_ctx = D.n_text_ctx; b.n_init = n_init;
    const float pat = cfg->patience > 0.f ? cfg->patience : 1.0f;
    b.max_cand = cfg->beam ? (int)lrintf((float)G * pat) : 0;
    if (cfg->beam && (b.max_cand <= 0 || b.max_cand > FIN_CAP)) return -2;
    b.fin_cap = FIN_CAP;
    b.tokens[0] = m->Wp<int32_t>(m->L.tokens0); b.tokens[1] = m->Wp<int32_t>(m->L.tokens1);
    const bool use_anc = G > 1;
    b.anc[0] = use_anc ? m->Wp<int32_t>(m->L.anc0) : nullptr;
    b.anc[1] = use_anc ? m->Wp<int32_t>(m->L.anc1) : nullptr;
    b.pos0 = m->Wp<int32_t>(m->L.pos0);
    b.sum_lp = m->Wp<float>(m->L.sum_lp); b.sum_lp_next = m->Wp<float>(m->L.sum_lp_next);
    b.row_done = m->Wp<int32_t>(m->L.row_done);
    b.win_done = m->Wp<int32_t>(m->L.win_done); b.win_done_prev = m->Wp<int32_t>(m->L.win_done_prev);
    b.n_done = m->Wp<int32_t>(m->L.n_done);
    b.step_dev = b.n_done + 1;
    b.fin_tokens = m->Wp<int32_t>(m->L.fin_tokens); b.fin_score = m->Wp<float>(m->L.fin_score);
    b.fin_len = m->Wp<int32_t>(m->L.fin_len); b.fin_count = m->Wp<int32_t>(m->L.fin_count);
    b.cand_lp = m->Wp<float>(m->L.cand_lp); b.cand_tok = m->Wp<int32_t>(m->L.cand_tok);
    b.logits = m->Wp<float>(m->L.logits);
    b.ts_mask = d_ts_mask;
    int32_t *sup = m->Wp<int32_t>(m->L.suppress);
    if (cfg->n_suppress > 0) {
        hipError_t er = hipMemcpyAsync(sup, d_suppress, (size_t)cfg->n_suppress * 4, hipMemcpyDeviceToDevice, s);
        if (er != hipSuccess) return -100 - (int)er;
    }
    b.suppress = sup;
    // every token id the filters write at must be a vocabulary id (upstream raises IndexError; here: size out of range)
    for (int t : {cfg->eot, cfg->sot, cfg->timestamp_begin}) if (t < 0 || t >= D.n_vocab) return -2;
    for (int t : {cfg->no_timestamps, cfg->no_speech, cfg->blank_token}) if (t < -1 || t >= D.n_vocab) return -2;
    b.win_uid = nullptr;
    if (cfg->window_uid) {
        int32_t *d_uid = m->Wp<int32_t>(m->L.win_uid);
        hipError_t eu = hipMemcpyAsync(d_uid, cfg->window_uid, (size_t)W * 4, hipMemcpyHostToDevice, s);
        if (eu == hipSuccess) eu = hipStreamSynchronize(s);        // the caller's array may be freed when this call returns
        if (eu != hipSuccess) return -100 - (int)eu;
        b.win_uid = d_uid;
    }
    b.cfg.window_uid = nullptr;          // a host pointer: the kernels read the device copy (b.win_uid) only
    hipError_t er = hipMemsetAsync(b.win_done_prev, 0, (size_t)W * 4, s);
    if (er != hipSuccess) return -100 - (int)er;

    SWX_TRY(swx_decode_init(b, d_init_tokens, s));

    const size_t layer_stride = (size_t)m->max_rows * D.n_text_ctx * d * e;
    if (g_debug_flags & SWX_FLAG_TICKET) {
        // the in-launch slab reduction relies on arrival counters that every launch leaves at zero; a launch that was aborted (fault,
        // reset) would leave them non-zero and every later job would silently pick the wrong last arriver: 16 KB, once per job
        hipError_t e_ = hipMemsetAsync(m->ws + m->L.ticket, 0, (size_t)SWX_DEC_TICKETS * 4, s);
        if (e_ != hipSuccess) return -100 - (int)e_;
    }
    // ---- prefill: one row per window (row w*G), n_init tokens
    FwdCfg f{};
    f.W = W; f.rpw = 1; f.row_mul = G; f.n_new = n_init;
    f.tokens = b.tokens[0]; f.ld_tok = (int64_t)G * b.TS;
    f.pos0 = m->Wp<int32_t>(m->L.zeros_i32);      // all zeros (any stride)
    f.kcache = m->ws + m->L.kcache; f.vcache = m->ws + m->L.vcache; f.layer_stride = layer_stride; f.cache_rows = m->max_rows;
    f.anc = nullptr; f.xkv = (const unsigned char *)d_xkv; f.capture = false;
    SWX_TRY(decoder_forward(m, f, s));
    unsigned char *x = m->ws + m->L.x, *hid2 = m->ws + m->L.hid2;
    // final LN of the rows at sot_index and at the last initial position of every window -> [W][2][d]
    SWX_TRY(swx_layernorm(m->dtype, x + (size_t)cfg->sot_index * d * e, (int64_t)n_init * d, m->A<float>(m->o_ln_g),
                          m->A<float>(m->o_ln_b), hid2, 2 * d, W, d, s));
    SWX_TRY(swx_layernorm(m->dtype, x + (size_t)(n_init - 1) * d * e, (int64_t)n_init * d, m->A<float>(m->o_ln_g),
                          m->A<float>(m->o_ln_b), hid2 + (size_t)d * e, 2 * d, W, d, s));
    // the prefill logits live at the tail of the logits region so that replication to rows [0, M) never overlaps
    float *lg2 = b.logits + (size_t)(m->L.logits_rows - 2 * W) * D.n_vocab;
    SWX_TRY(logits_gemm(m, hid2, d, 2 * W, lg2, s));
    SWX_TRY(swx_decode_after_prefill(b, lg2, d_nospeech, s));

    int cur = 0, steps = 0;
    int32_t h_done = 0;
    const int CHECK = 8;
    // One unit of the loop = [forward pass of the rows' newest tokens -> final LayerNorm -> logits] + [selection of token i].
    // Token 0 is selected from the prefill logits; after every selection the loop's exit conditions are looked at:
    // context full (decode.py:60), every window done (one host sync every CHECK steps), budget used up.
    auto unit = [&](int cur_in, hipStream_t st, bool capturing = false) -> int {
        FwdCfg g{};
        g.W = W; g.rpw = G; g.row_mul = 1; g.n_new = 1;
        g.tokens = b.tokens[cur_in]; g.ld_tok = b.TS; g.pos0 = b.pos0;
        g.kcache = f.kcache; g.vcache = f.vcache; g.layer_stride = layer_stride; g.cache_rows = m->max_rows;
        g.anc = use_anc ? b.anc[cur_in] : nullptr; g.xkv = (const unsigned char *)d_xkv; g.capture = false;
        // profiler's byte count only (steps = tokens sampled so far).  Never a captured kernel argument: a replayed graph would
        // carry the position of the step it was captured at (the profiler is off under replay, and 0 = "unknown" there)
        g.step_pos = capturing ? 0 : n_init + steps;
        g.pos_bound = n_init + cfg->sample_len;        // (a property of the job, not of the step: safe to bake into the graph)
        const int fr = decoder_forward(m, g, st);
        if (fr < 0) return fr;
        unsigned char *hh = m->ws + m->L.h;
        if (fr == 0)   // the step leaves the raw residual stream in x
            SWX_TRY(swx_layernorm(m->dtype, x, d, m->A<float>(m->o_ln_g), m->A<float>(m->o_ln_b), hh, d, M, d, st));
        SWX_TRY(logits_gemm(m, hh, d, M, b.logits, st));
        return swx_decode_select(b, cur_in, st);
    };
    // returns 1 when the loop ends after the selection of token i
    auto after_select = [&](int i) -> int {
        steps = i + 1;
        if (n_init + i + 1 > D.n_text_ctx) return 1;          // tokens.shape[-1] > n_ctx (decode.py:60)
        // early-exit poll (one host sync); pointless while EOT is still suppressed by min_tokens
        if ((steps % CHECK == 0 && steps >= cfg->min_tokens) || steps == cfg->sample_len) {
            er = hipMemcpyAsync(&h_done, b.n_done, 4, hipMemcpyDeviceToHost, s);
            if (er != hipSuccess) return -100 - (int)er;
            er = hipStreamSynchronize(s);
            if (er != hipSuccess) return -100 - (int)er;
            if (h_done >= W) return 1;
        }
        return i + 1 >= cfg->sample_len ? 1 : 0;
    };
    SWX_TRY(swx_decode_select(b, cur, s));
    if (cfg->beam) cur ^= 1;
    int stop = after_select(0);
    if (stop < 0) return stop;
    // Units 2k, 2k+1 (k >= 1) are replayed from ONE captured graph: a step is ~290 kernel launches, which the host cannot issue
    // as fast as the device retires them below ~50 rows (sequential transcribe(): 5 rows).  Everything that differs between two
    // steps is device state (token buffers, positions, ancestor tables, the step counter); the token-buffer parity returns
    // to its start after two steps.  The poll above only falls after odd units, the context check is made for both units up front.
    const bool graph_ok = !g_prof_enabled && !(g_debug_flags & SWX_FLAG_NO_GRAPH) && !m->graphs_off;
    swx_model::StepGraph *sg = nullptr;
    bool graph_failed_now = false;
    for (int i = 1; !stop; ) {
        const bool pair = graph_ok && !m->graphs_off && !graph_failed_now && i >= 2 && (i & 1) == 0 && i + 1 < cfg->sample_len && n_init + i + 1 <= D.n_text_ctx;
        if (pair) {
            if (!sg) sg = step_graph(m, b, d_xkv, cur, [&](hipStream_t cs) -> int {
                SWX_TRY(unit(cur, cs, true));
                return unit(cfg->beam ? cur ^ 1 : cur, cs, true);
            });
            if (sg && hipGraphLaunch(sg->exec, s) == hipSuccess) {
                ++m->n_replays;
                stop = after_select(i + 1);
                if (stop < 0) return stop;
                i += 2;
                continue;
            }
            (void)hipGetLastError();
            // capture or replay failed: this call goes on eagerly; the next swx_decode tries again, and only the third failure
            // switches replay off for the handle (swx_graph_stats reports it)
            graph_failed_now = true;
            if (++m->graph_failures >= 3) m->graphs_off = true;
            if (m->graph_failures == 1)
                fprintf(stderr, "libswx: decode-step graph capture / replay failed; launching the steps eagerly (results are identical)\n");
        }
        SWX_TRY(unit(cur, s));
        ++m->n_eager_units;
        if (cfg->beam) cur ^= 1;
        stop = after_select(i);
        if (stop < 0) return stop;
        ++i;
    }
    const int G_out = swx_decode_gout(cfg);
    SWX_TRY(swx_decode_finalize(b, cur, steps, d_tokens_out, d_lens_out, d_sumlp_out, G_out, s));
    er = hipStreamSynchronize(s);
    if (er != hipSuccess) return -100 - (int)er;
    return steps;
}

// ------------------------------------------------------------------------------------------------------ score
// capture: raw qk of the alignment heads for token rows cap_row0 .. cap_row0 + cap_rows - 1 -> L.cap [W][n_align][cap_ld][1500]
static int score_forward(swx_model *m, const int32_t *d_tokens, const int32_t *h_n_tok, int W, int max_n, int cap_row0,
                         int cap_rows, int cap_ld, const void *d_xkv, bool capture, hipStream_t s, void *qcap = nullptr)
{
    const swx_dims &D = m->dims;
    if (W > m->max_windows) return -8;
    if (max_n <= 0 || max_n > D.n_text_ctx) return -2;
    for (int w = 0; w < W; ++w) if (h_n_tok[w] > max_n || h_n_tok[w] <= 0) return -1;
    FwdCfg f{};
    f.W = W; f.rpw = 1; f.row_mul = 1; f.n_new = max_n;
    f.tokens = d_tokens; f.ld_tok = max_n;
    f.pos0 = m->Wp<int32_t>(m->L.zeros_i32);
    f.kcache = m->ws + m->L.sk; f.vcache = m->ws + m->L.sv; f.layer_stride = 0; f.cache_rows = m->max_windows;
    f.anc = nullptr; f.xkv = (const unsigned char *)d_xkv;
    f.capture = capture;
    f.qcap = (unsigned char *)qcap;
    f.cap_row0 = cap_row0; f.cap_rows = cap_rows; f.cap_ld_n = cap_ld;
    if (capture && (cap_rows <= 0 || cap_row0 < 0 || cap_row0 + cap_rows > max_n || cap_ld < cap_rows)) return -1;
    return decoder_forward(m, f, s);
}

// token probabilities of a scoring pass: rows n_sot .. n_sot+T-1 of the final-LN'd hidden states -> logits[:, :eot] ->
// softmax -> gather at the next token (timing.py:62-64); d_token_probs f32 [W][max_n]
static int score_token_probs(swx_model *m, const int32_t *d_tokens, int W, int max_n, int n_sot, int eot, float *d_token_probs,
                             hipStream_t s)
{
    const swx_dims &D = m->dims;
    const int d = D.n_text_state;
    const size_t e = m->esz;
    unsigned char *x = m->ws + m->L.x, *hh = m->ws + m->L.h;
    const int rows = W * max_n;
    SWX_TRY(swx_layernorm(m->dtype, x, d, m->A<float>(m->o_ln_g), m->A<float>(m->o_ln_b), hh, d, rows, d, s));
    const int rpw = max_n - n_sot - 2;     // probability rows per window (T_max)
    if (rpw > 0) {
        float *lg = m->Wp<float>(m->L.logits);
        int32_t *targets = (int32_t *)(m->ws + m->L.att);     // att is free after the forward pass
        const int total = W * rpw;
        hipLaunchKernelGGL(score_targets_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, d_tokens, max_n, n_sot, rpw, targets, total);
        const int chunk = (int)m->L.logits_rows;
        for (int w = 0; w < W; ++w) {
            for (int r0 = 0; r0 < rpw; r0 += chunk) {
                const int nr = (rpw - r0) < chunk ? (rpw - r0) : chunk;
                const unsigned char *hid = hh + ((size_t)w * max_n + n_sot + r0) * d * e;
                SWX_TRY(logits_gemm(m, hid, d, nr, lg, s));
                hipLaunchKernelGGL(token_prob_kernel, dim3(nr), dim3(256), 0, s, lg, D.n_vocab, eot, targets + (size_t)w * rpw + r0,
                                   d_token_probs + (size_t)w * max_n + r0);
            }
        }
        SWX_CHECK_LAUNCH();
    }
    return 0;
}

int swx_score(swx_model *m, const int32_t *d_tokens, const int32_t *h_n_tok, int W, int max_n, int n_sot, int eot,
              const int32_t *h_n_frames, float qk_scale, int medfilt_width, const void *d_xkv, float *d_token_probs,
              float *d_neg_matrix, void *stream)
{
    if (!m || !m->arena || !m->ws) return -9;
    if (W <= 0) return 0;
    if (m->n_align != m->ws_n_align || m->n_align <= 0) return -8;
    if (2 * W + 16 > SMALL_I32) return -8;
    const swx_dims &D = m->dims;
    hipStream_t s = S(stream);
    SWX_TRY(score_forward(m, d_tokens, h_n_tok, W, max_n, n_sot, max_n - n_sot - 1, max_n, d_xkv, true, s));

    // per-window row / frame counts on the device
    std::vector<int32_t> hv(2 * W);
    for (int w = 0; w < W; ++w) {
        const int T = h_n_tok[w] - n_sot - 2;
        if (T < 0) return -1;
        hv[w] = T + 1;
        int F = h_n_frames[w];
        if (F < 1) F = 1;
        if (F > D.n_audio_ctx) F = D.n_audio_ctx;
        hv[W + w] = F;
    }
    int32_t *d_small = m->Wp<int32_t>(m->L.small_i32);
    hipError_t er = hipMemcpyAsync(d_small, hv.data(), hv.size() * 4, hipMemcpyHostToDevice, s);
    if (er != hipSuccess) return -100 - (int)er;
    er = hipStreamSynchronize(s);     // hv goes out of scope; pageable copies are staged synchronously anyway
    if (er != hipSuccess) return -100 - (int)er;

    SWX_TRY(score_token_probs(m, d_tokens, W, max_n, n_sot, eot, d_token_probs, s));
    // alignment matrix
    // scratch for the row statistics: the [W][H][1500] f32 region (>= W*H*max_n float2, max_n <= 448); the raw scores stay intact
    SWX_TRY(swx_align_weights_launch(m->Wp<float>(m->L.cap), m->Wp<float>(m->L.mean), m->Wp<float>(m->L.mean), m->Wp<float>(m->L.sd),
                                     W, m->n_align, max_n, D.n_audio_ctx, d_small, d_small + W, qk_scale, medfilt_width,
                                     d_neg_matrix, max_n, D.n_audio_ctx, s));
    return 0;
}

int swx_score_qk(swx_model *m, const int32_t *d_tokens, const int32_t *h_n_tok, int W, int max_n, int n_sot, int eot,
                 int row0, int n_rows, const void *d_xkv, float *d_token_probs, float *d_qk, void *stream)
{
    if (!m || !m->arena || !m->ws) return -9;
    if (W <= 0) return 0;
    if (m->n_align != m->ws_n_align || m->n_align <= 0) return -8;
    if (!d_qk) return -1;
    for (int w = 0; w < W; ++w) if (h_n_tok[w] - n_sot - 2 < 0) return -1;
    hipStream_t s = S(stream);
    // the capture buffer is written densely ([W][n_align][n_rows][1500]) so that one copy hands it over
    SWX_TRY(score_forward(m, d_tokens, h_n_tok, W, max_n, row0, n_rows, n_rows, d_xkv, true, s));
    if (d_token_probs) SWX_TRY(score_token_probs(m, d_tokens, W, max_n, n_sot, eot, d_token_probs, s));
    const size_t bytes = (size_t)W * m->n_align * n_rows * m->dims.n_audio_ctx * sizeof(float);
    hipError_t er = hipMemcpyAsync(d_qk, m->Wp<float>(m->L.cap), bytes, hipMemcpyDeviceToDevice, s);
    return er == hipSuccess ? 0 : -100 - (int)er;
}

int swx_forward_logits(swx_model *m, const int32_t *d_tokens, const int32_t *h_n_tok, int W, int max_n, const void *d_xkv,
                       float *d_logits, void *stream)
{
    if (!m || !m->arena || !m->ws) return -9;
    if (W <= 0) return 0;
    const swx_dims &D = m->dims;
    hipStream_t s = S(stream);
    const int d = D.n_text_state;
    SWX_TRY(score_forward(m, d_tokens, h_n_tok, W, max_n, 0, 0, max_n, d_xkv, false, s));
    unsigned char *x = m->ws + m->L.x, *hh = m->ws + m->L.h;
    const int rows = W * max_n;
    SWX_TRY(swx_layernorm(m->dtype, x, d, m->A<float>(m->o_ln_g), m->A<float>(m->o_ln_b), hh, d, rows, d, s));
    return logits_gemm(m, hh, d, rows, d_logits, s);
}

// ------------------------------------------------------------------------------- head-selection variants (f4)
// The teacher-forced pass of ONE window that keeps the cross-attention queries of every layer instead of any head's scores:
// d_q [n_text_layer][max_n][d] in the compute dtype (swx_qcap_bytes).  timing.py:41-67 with the hooks of :50-56 replaced by
// "keep q": a head's score row is q . K^T of the window's cross-K, which swx_cross_kv left in d_xkv.
size_t swx_qcap_bytes(const swx_model *m, int max_n)
{
    if (!m || max_n <= 0) return 0;
    return (size_t)m->dims.n_text_layer * max_n * m->dims.n_text_state * m->esz;
}

int swx_score_q(swx_model *m, const int32_t *d_tokens, const int32_t *h_n_tok, int max_n, int n_sot, int eot, const void *d_xkv,
                float *d_token_probs, void *d_q, void *stream)
{
    if (!m || !m->arena || !m->ws) return -9;
    if (!d_q || !d_tokens || !h_n_tok) return -1;
    if (h_n_tok[0] - n_sot - 2 < 0) return -1;
    hipStream_t s = S(stream);
    SWX_TRY(score_forward(m, d_tokens, h_n_tok, 1, max_n, 0, 0, max_n, d_xkv, false, s, d_q));
    if (d_token_probs) SWX_TRY(score_token_probs(m, d_tokens, 1, max_n, n_sot, eot, d_token_probs, s));
    return 0;
}

// scratch of the two calls below: head scores (f64) and picks of every row, column norms / scores / picks of every head
size_t swx_heads_scratch_bytes(const swx_model *m, int max_n)
{
    if (!m || max_n <= 0) return 0;
    const size_t LH = (size_t)m->dims.n_text_layer * m->dims.n_text_head;
    return align_up(LH * max_n * sizeof(double)) + align_up(LH * max_n * sizeof(int32_t)) + align_up(LH * SWX_HS_MAXF * sizeof(float)) +
           align_up(LH * sizeof(float)) + align_up(LH * sizeof(int32_t));
}

// dynamic_heads (timing.py:87-103): rows row0 .. row0 + n_rows - 1 of the pass; d_qk_sel [count][n_rows][ld_f] f32 receives the RAW
// scaled scores of the count heads picked per row -- the input layout of swx_align_weights (H = count, N = n_rows), which
// applies qk_scale / softmax / z-normalisation / median / head mean exactly as on the default path.  d_peaks: null (every row's
// own attention peak) or the n_rows jump midpoints of the previous DTW pass (f64).
int swx_heads_dynamic(swx_model *m, const void *d_q, int max_n, int row0, int n_rows, const void *d_xkv, int n_frames, float qk_scale,
                      int count, const double *d_peaks, float *d_qk_sel, int ld_f, void *d_scratch, size_t scratch_bytes, void *stream)
{
    if (!m || !m->arena) return -9;
    if (!d_q || !d_xkv || !d_qk_sel || !d_scratch) return -1;
    if (row0 < 0 || n_rows <= 0 || row0 + n_rows > max_n || ld_f < m->dims.n_audio_ctx) return -1;
    if (scratch_bytes < swx_heads_scratch_bytes(m, max_n)) return -8;
    const swx_dims &D = m->dims;
    const size_t LH = (size_t)D.n_text_layer * D.n_text_head;
    unsigned char *p = (unsigned char *)d_scratch;
    double *score = (double *)p; p += align_up(LH * max_n * sizeof(double));
    int32_t *sel = (int32_t *)p;
    int F = n_frames < 1 ? 1 : (n_frames > D.n_audio_ctx ? D.n_audio_ctx : n_frames);
    return swx_headsel_dynamic_launch(m->dtype, d_q, max_n, D.n_text_state, row0, n_rows, d_xkv, xkv_chunk_elems(m), D.n_text_layer,
                                      D.n_text_head, F, D.n_audio_ctx, qk_scale, count, d_peaks, score, sel, d_qk_sel, ld_f, S(stream));
}

// aligner = 'new' (timing.py:115-163) over the n_tok rows of the pass; d_neg_matrix [n_out][ld_f] f32 receives MINUS the
// column-normalised mean map of the topk sharpest heads for rows row0 .. row0 + n_out - 1 (= the DTW input, timing.py:194)
int swx_heads_new(swx_model *m, const void *d_q, int max_n, int n_tok, int row0, int n_out, const void *d_xkv, int n_frames,
                  float qk_scale, int medfilt_width, int topk, float w_colnorm, float w_rownorm, float w_coverage,
                  float *d_neg_matrix, int ld_f, void *d_scratch, size_t scratch_bytes, void *stream)
{
    if (!m || !m->arena) return -9;
    if (!d_q || !d_xkv || !d_neg_matrix || !d_scratch) return -1;
    if (n_tok <= 0 || n_tok > max_n) return -1;
    if (scratch_bytes < swx_heads_scratch_bytes(m, max_n)) return -8;
    const swx_dims &D = m->dims;
    const size_t LH = (size_t)D.n_text_layer * D.n_text_head;
    unsigned char *p = (unsigned char *)d_scratch;
    p += align_up(LH * max_n * sizeof(double)) + align_up(LH * max_n * sizeof(int32_t));
    float *colnorm = (float *)p; p += align_up(LH * SWX_HS_MAXF * sizeof(float));
    float *score = (float *)p; p += align_up(LH * sizeof(float));
    int32_t *top = (int32_t *)p;
    int F = n_frames < 1 ? 1 : (n_frames > D.n_audio_ctx ? D.n_audio_ctx : n_frames);
    return swx_headsel_new_launch(m->dtype, d_q, max_n, D.n_text_state, n_tok, row0, n_out, d_xkv, xkv_chunk_elems(m), D.n_text_layer,
                                  D.n_text_head, F, qk_scale, medfilt_width, topk, w_colnorm, w_rownorm, w_coverage, colnorm, score, top,
                                  d_neg_matrix, ld_f, S(stream));
}

// d_out[e] = sum_j coef[j] * d_xs[j][e], n_in <= 8 (extra_models: the pooled matrix from each model's own head mean)
int swx_weighted_sum(const float *const *h_xs, const float *h_coef, int n_in, float *d_out, int64_t n, void *stream)
{
    if (!h_xs || !h_coef || !d_out) return -1;
    return swx_weighted_sum_launch(h_xs, h_coef, n_in, d_out, n, S(stream));
}

// how the decode loops of this handle ran so far: out[0] = step graphs captured, out[1] = two-step graph replays,
// out[2] = steps launched eagerly, out[3] = 1 when capture / replay failed once and the handle fell back to eager launches
int swx_graph_stats(const swx_model *m, int64_t *out)
{
    if (!m || !out) return -1;
    out[0] = m->n_captures; out[1] = m->n_replays; out[2] = m->n_eager_units; out[3] = (m->graphs_off || m->graph_failures > 0) ? 1 : 0;
    return 0;
}

// ----------------------------------------------------------------------------- stand-alone alignment weights
size_t swx_align_weights_scratch_bytes(int W, int H, int N)
{
    if (W <= 0 || H <= 0 || N <= 0) return 0;
    return align_up((size_t)W * H * N * sizeof(float2)) + align_up((size_t)2 * W * sizeof(int32_t));
}

int swx_align_weights(const float *d_qk, int W, int H, int N, int ld_f, const int32_t *h_n_frames, float qk_scale,
                      int medfilt_width, float *d_neg_matrix, void *d_scratch, size_t scratch_bytes, void *stream)
{
    // stand-alone a7 (test hook / extra_models path).  Like every entry point it allocates nothing: the caller hands in
    // swx_align_weights_scratch_bytes(W, H, N) of device scratch (row statistics + the per-window counts)
    if (W <= 0) return 0;
    if (!d_scratch || scratch_bytes < swx_align_weights_scratch_bytes(W, H, N)) return -8;
    hipStream_t s = S(stream);
    float *rstat = (float *)d_scratch;
    int32_t *cnt = (int32_t *)((unsigned char *)d_scratch + align_up((size_t)W * H * N * sizeof(float2)));
    std::vector<int32_t> hv(2 * W);
    for (int w = 0; w < W; ++w) { hv[w] = N; hv[W + w] = h_n_frames[w]; }
    hipError_t er = hipMemcpyAsync(cnt, hv.data(), hv.size() * 4, hipMemcpyHostToDevice, s);
    if (er == hipSuccess) er = hipStreamSynchronize(s);      // hv is a stack-lifetime staging buffer
    if (er != hipSuccess) return -100 - (int)er;
    return swx_align_weights_launch(d_qk, rstat, nullptr, nullptr, W, H, N, ld_f, cnt, cnt + W, qk_scale, medfilt_width, d_neg_matrix,
                                    N, ld_f, s);
}

// ------------------------------------------------------------------------------------------------ test hooks
int swx_test_gemm(int dtype, const void *d_a, int64_t lda, const void *d_w, const float *d_bias, const void *d_res,
                  void *d_c, int64_t ldc, int M, int N, int K, int epilogue, int force_kernel, void *stream)
{
    GemmArgs g = gemm_args(d_a, lda, d_w, K, d_bias, d_c, ldc, M, N, K, epilogue);
    g.R = d_res; g.ldr = ldc;
    return swx_gemm(dtype, g, force_kernel, S(stream));
}

// the f16 kernel swx_gemm picks for a launch (contiguous, 16-byte aligned operands assumed): 0 register-staged tiled, 1 skinny,
// 2 / 3 direct-to-LDS at 128 / 64 columns, 4 / 5 ring at 64 / 128 columns, 6 the 256 x 256 kernel; < 0 = not offered.  No GPU needed.
int swx_test_gemm_plan(int M, int N, int K, int epilogue, int force_kernel, int flags)
{
    return swx_gemm_plan_f16(M, N, K, epilogue, N, N, true, force_kernel, flags);
}

int swx_test_dec_gemm(const void *d_a, int64_t lda, const void *d_w, const float *d_gamma, const float *d_beta, const float *d_bias,
                      void *d_c, int64_t ldc, void *d_x, void *d_kcache, void *d_vcache, const int32_t *d_pos0, int n_ctx, int d,
                      int M, int N, int K, int epilogue, void *d_scratch, size_t scratch_bytes, void *stream)
{
    // d_scratch: N*K halfs (folded weights) + 2N floats (c1, c2) + the slab floats of the shape, 256-byte aligned pieces
    hipStream_t s = S(stream);
    unsigned char *p = (unsigned char *)d_scratch;
    const size_t slab_b = align_up(swx_dec_slab_floats(M, N, K) * 4 + 256);
    const size_t need = align_up((size_t)N * K * 2) + 2 * align_up((size_t)N * 4) + slab_b + (size_t)SWX_DEC_TICKETS * 4;
    if (scratch_bytes < need) return -8;
    f16 *wf = (f16 *)p; p += align_up((size_t)N * K * 2);
    float *c1 = (float *)p; p += align_up((size_t)N * 4);
    float *c2 = (float *)p; p += align_up((size_t)N * 4);
    float *slabs = (float *)p; p += slab_b;
    int *ticket = (int *)p;
    // bit 7 of `epilogue`: the caller zeroed the counters ONCE and keeps one scratch buffer over its calls -- what the decode path does
    // (swx_bind_workspace zeroes, the last arriver of every launch resets); without it every call starts from fresh counters
    if (!(epilogue & 128)) { hipError_t e = hipMemsetAsync(ticket, 0, (size_t)SWX_DEC_TICKETS * 4, s); if (e != hipSuccess) return -100 - (int)e; }
    DecGemmArgs g{};
    g.A = (const f16 *)d_a; g.lda = lda; g.M = M; g.N = N; g.K = K; g.epi = epilogue; g.ldw = K;
    g.C = (f16 *)d_c; g.ldc = ldc; g.X = (f16 *)d_x; g.ldx = ldc; g.slabs = slabs; g.ticket = ticket;
    g.kcache = (f16 *)d_kcache; g.vcache = (f16 *)d_vcache; g.pos0 = d_pos0; g.n_ctx = n_ctx; g.d = d;
    g.epi = epilogue & 31;
    g.tall = (epilogue & 64) ? 1 : 0;                 // bit 6: a multi-token pass (the tall kernel from 161 rows on)
    g.rps = (epilogue & 64) && (epilogue & DEC_QKV) ? 7 : 0;      // (QKV scatter of the tall test: 7 rows per sequence)
    if ((epilogue & DEC_LN) && (epilogue & 32)) {     // bit 5: d_w is already folded, d_gamma / d_beta are c1 / c2 (timing runs)
        g.W = (const f16 *)d_w; g.c1 = d_gamma; g.c2 = d_beta;
    } else if (epilogue & DEC_LN) {
        SWX_TRY(swx_fold_ln(d_w, d_gamma, d_beta, d_bias, wf, c1, c2, N, K, s));
        g.W = wf; g.c1 = c1; g.c2 = c2;
    } else if (epilogue & 32) {                       // timing runs: d_w is taken as packed already
        g.W = (const f16 *)d_w; g.c2 = d_bias;
    } else {
        SWX_TRY(swx_fold_ln(d_w, nullptr, nullptr, nullptr, wf, nullptr, nullptr, N, K, s));      // re-pack only
        g.W = wf; g.c2 = d_bias;
    }
    return swx_gemm_dec(g, s);
}

int swx_test_self_attn_step(const void *d_q, void *d_kcache, void *d_vcache, const int32_t *d_anc, const int32_t *d_pos0,
                            int R, int H, int n_ctx, int d, int variant, void *d_o, void *stream)
{
    SelfAttnArgs sa{};
    sa.qkv = d_q; sa.ldqkv = d; sa.kcache = d_kcache; sa.vcache = d_vcache; sa.anc = (int32_t *)d_anc; sa.pos0 = d_pos0;
    sa.o = d_o; sa.ldo = d; sa.R = R; sa.n_new = 1; sa.H = H; sa.n_ctx = n_ctx; sa.d = d; sa.skip_append = 1;
    sa.step_cached = variant < 2 ? 1 : 0;
    sa.pos_bound = variant == 0 ? 128 : 0;
    return swx_self_attention(SWX_F16, sa, 1, S(stream));
}

int swx_test_self_attn_multi(const void *d_q, void *d_kcache, void *d_vcache, int R, int H, int n_new, int n_ctx, int d, int mq,
                             void *d_o, void *stream)
{
    static int32_t *d_zeros = nullptr;           // every row starts at position 0
    static int zeros_n = 0;
    if (R <= 0 || n_new <= 0 || n_new > n_ctx) return -2;
    if (zeros_n < R) {
        if (d_zeros) (void)hipFree(d_zeros);
        zeros_n = 0; d_zeros = nullptr;
        if (hipMalloc((void **)&d_zeros, (size_t)R * 4) != hipSuccess) return -3;
        zeros_n = R;
    }
    hipError_t e = hipMemsetAsync(d_zeros, 0, (size_t)R * 4, S(stream));
    if (e != hipSuccess) return -100 - (int)e;
    SelfAttnArgs sa{};
    sa.qkv = d_q; sa.ldqkv = d; sa.kcache = d_kcache; sa.vcache = d_vcache; sa.anc = nullptr; sa.pos0 = d_zeros;
    sa.o = d_o; sa.ldo = d; sa.R = R; sa.n_new = n_new; sa.H = H; sa.n_ctx = n_ctx; sa.d = d; sa.skip_append = 1;
    sa.pos0_all_zero = mq ? 1 : 0;
    return swx_self_attention(SWX_F16, sa, 1, S(stream));
}

int swx_test_layernorm(int dtype, const void *d_x, const float *d_g, const float *d_b, void *d_y, int rows, int d, void *stream)
{
    return swx_layernorm(dtype, d_x, d, d_g, d_b, d_y, d, rows, d, S(stream));
}

int swx_test_attention(int dtype, const void *d_q, int64_t ldq, const void *d_k, const void *d_v, int64_t ldkv, void *d_o,
                       int64_t ldo, int B, int H, int nq, int nk, int force_kernel, int vt_kp, void *stream)
{
    AttnArgs a{};
    a.q = d_q; a.ldq = ldq; a.k = d_k; a.v = d_v; a.ldkv = ldkv; a.o = d_o; a.ldo = ldo;
    a.k_bs = (int64_t)nk * ldkv; a.vt_kp = vt_kp;
    a.v_bs = vt_kp ? (int64_t)H * 64 * vt_kp : (int64_t)nk * ldkv;
    a.B = B; a.H = H; a.nq = nq; a.nk = nk; a.q_rows_per_batch = nq;
    return swx_attention(dtype, a, force_kernel, S(stream));
}

int swx_test_lane_xor(const uint32_t *d_in, uint32_t *d_out, int n_waves, void *stream)
{
    if (!d_in || !d_out || n_waves <= 0) return -1;
    hipLaunchKernelGGL(lane_xor_check_kernel, dim3(n_waves), dim3(64), 0, S(stream), d_in, d_out);
    SWX_CHECK_LAUNCH();
    return 0;
}

int swx_test_gelu_pair(uint64_t *d_out, void *stream)
{
    if (!d_out) return -1;
    hipStream_t s = S(stream);
    const unsigned long long init[3] = {0ull, 0xffffffffull, 0ull};
    hipError_t e = hipMemcpyAsync(d_out, init, sizeof(init), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);            // `init` is a stack-lifetime staging buffer
    if (e != hipSuccess) return -100 - (int)e;
    hipLaunchKernelGGL(gelu_pair_check_kernel, dim3(1u << 19), dim3(256), 0, s, (unsigned long long *)d_out);
    SWX_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
