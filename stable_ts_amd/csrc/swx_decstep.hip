// swx_decstep.hip -- the weight-streaming GEMM of the fused f16 decoder step, third generation ("dec" kernels).
//
// One decoder step multiplies M <= ~128 live sequences (windows x beams) with every decoder weight once: HBM-bound
// weight streaming (1.60 GB fp16 per step for large-v3), reached from stable_whisper/decode.py:40
// (`inference.logits` -> upstream TextDecoder.forward with the KV-cache hooks).
//
// What round 1's profile said about the split-K generation (swx_gemm.hip::gemm_f16_pg + splitk_finish_f16): 8.3 us per
// launch for 3-13 MB of weights (0.13 of HBM peak), as much HBM traffic for the f32 partial slabs as for the weights
// (2.05x the algorithmic bytes), and a finish launch per projection (another 8 us) only to reduce the slabs and apply
// the next LayerNorm.  This generation removes the slabs and the finish launches:
//   * split M, not K.  A workgroup owns 64 output columns (16 per wave) x MT*16 rows x the WHOLE reduction (or a
//     >= 640-deep slice of it for the one projection with K = 4d), so it finishes its outputs itself: bias / GELU /
//     residual add / K,V-cache scatter happen in the epilogue and the compute dtype is stored directly.  The row groups of
//     one column panel run on the SAME XCD (block id -> XCD is id % 8 in hardware; the mapping below is built on it for
//     speed only), so the panel's weights leave HBM once and the other row groups hit them in that XCD's L2.
//   * LayerNorm folded into the consumer:  LN(x) W^T + b  =  rstd * (x (W.gamma)^T - mean * c1) + c2  with
//     c1[n] = sum_k (W.gamma)[n][k],  c2[n] = b[n] + sum_k beta[k] W[n][k]  prepared once at load time
//     (swx_weights_finalize).  The GEMM runs on the RAW residual stream; a workgroup holds complete rows of it
//     (K = d, un-split), computes mean / rstd of its rows from its own LDS tile and applies them in the epilogue.  No
//     LayerNorm launch, no normalised copy of the activations in HBM.
//   * one-shot operand staging.  The activation tile [MT*16][kslice] goes global -> LDS by LDS-DMA (`global_load_lds`,
//     16 B per lane, no VGPRs) in ONE batch issued at kernel entry together with every weight fragment of the wave
//     (<= 40 x 1 KB in flight per wave): one memory round trip per launch instead of one per 64-deep chunk.  The LDS
//     image is XOR-swizzled on the SOURCE address (the DMA destination is lane-linear) so that the 16 rows of an MFMA
//     operand fragment fall on 16 different 16-byte bank columns.
//   * the MFMA operands are swapped (weights = A operand, activations = B operand): a lane's 4 accumulator registers are
//     4 CONSECUTIVE output columns of one row, so every store of the epilogue is 8 (f16) or 16 (f32) contiguous bytes.
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include "swx_common.h"
#include "swx_kernels.h"

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

constexpr int DEC_MAXMT = 3;
constexpr int DEC_NPF = 4;        // prefetch loads per wave (64 lines of 128 B each): the smallest launch at M = 100 (N = 1280: 140 workgroups)
                                  // x 4 waves x 4 x 8 KB = 18 MB >= any projection (13 MB)

// cache policy of the weight stream: every weight byte is read once per step by ONE XCD (the row groups of a panel share it
// through that XCD's L2).  Built with -DSWX_DEC_NT the loads carry `nt` (MI355X_MICROARCH.md "nt-weights": -5..-10 % per layer
// on a batch-1 decode layer); A/B on this step in profiles/r03_dec_nt_ab.txt.
#ifdef SWX_DEC_NT
#define SWX_DEC_W_POLICY " nt"
#else
#define SWX_DEC_W_POLICY ""
#endif

// NKS = K-slice depth / 32 and the epilogue EPI are fixed at compile time: every loop below unrolls without branches, and no
// load sits under a run-time condition (a conditional load compiles to a branch whose join waits vmcnt(0), which would drain
// the weight stream before the barrier -- seen in the ISA of the run-time-epilogue version)
// WPB = waves per workgroup: 4 (a workgroup = the four 16-column groups of a 64-column panel, sharing one activation tile in LDS), or 1
// (round 6, MT = 1, launches of few workgroups -- the 5 rows of the reference's sequential flow are 20-80 four-wave workgroups, each
// asking ONE CU for 160 KB of weights + its tile: 2-3 round trips at the ~64 KB a CU keeps in flight; as single-wave workgroups the same
// waves sit on 80-320 CUs with 40 KB + a tile each).  A wave's arithmetic -- its statistics, k-step order, epilogue -- does not depend
// on WPB: bit-identical (tests/test_gpu_kernels.py).
template <int MT, int NKS, int EPI, int WPB = 4>
__global__ __launch_bounds__(64 * WPB) void gemm_dec_f16(DecGemmArgs g)
{
    constexpr bool E_LN = (EPI & DEC_LN) != 0, E_GELU = (EPI & DEC_GELU) != 0, E_RES = (EPI & DEC_RES) != 0,
                   E_QKV = (EPI & DEC_QKV) != 0, E_SLAB = (EPI & DEC_SLAB) != 0, E_TICKET = (EPI & DEC_TICKET) != 0;
    static_assert(!E_TICKET || (E_SLAB && E_RES && !E_LN), "the ticket reduction finishes x += bias + sum of slabs");
    static_assert(WPB == 4 || (WPB == 1 && MT == 1 && !E_TICKET), "single-wave workgroups: one row tile, no in-launch reduction");
    constexpr bool E_FIN = !E_SLAB || E_TICKET;      // this launch finishes outputs itself: bias / residual operands are loaded
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];   // [MT*16][kslice] f16 | float2 stat[MT*16]
    const int tid = threadIdx.x, lane = tid & 63;
    const int li = lane & 15, lg = lane >> 4;
    // ---- block id -> (column panel, K slice, row group): all row groups of one (panel, slice) unit share an XCD
    const int b = blockIdx.x;
    const int xcd = b & 7, slot = b >> 3;
    const int unit_x = (slot / g.n_rg) * 8 + xcd, rg = slot % g.n_rg;       // WPB = 1: (unit, 16-column group) pairs
    const int unit = WPB == 4 ? unit_x : unit_x >> 2;
    const int wave = WPB == 4 ? tid >> 6 : unit_x & 3;                      // this wave's 16-column group of the panel
    const int dwave = WPB == 4 ? wave : 0;                                  // ... and its share of the workgroup's common work
    const int panels = (g.N + 63) >> 6;
    if (unit >= panels * g.ks2) return;
    const int panel = unit / g.ks2, ks_id = unit - panel * g.ks2;
    constexpr int kslice = NKS * 32;             // MFMA k-steps of 32
    constexpr int SPR = kslice >> 3;             // 16-byte slots per LDS row (a multiple of 16)
    constexpr int RS = kslice * 2;               // LDS row stride in bytes
    const int k0 = ks_id * kslice;
    const int r0 = rg * (MT * 16);

    // Issue order = arrival order (loads return in order): the activation tile first (L2-resident: the previous launch
    // wrote it), then every weight fragment of the slice (HBM, ~4 us of latency for a cold 3-13 MB panel set -- measured:
    // profiles/r02_dec_ablation.csv), then the epilogue's operands.  Only the DMA is waited for before the barrier; the
    // LayerNorm statistics and the MFMA loop then run UNDER the weight latency, each k-step waiting for its own fragment.
    // ---- activation tile -> LDS by LDS-DMA; instruction q writes LDS bytes [q*1024, q*1024 + 1024)
    {
        constexpr int n_instr = (MT * 16 * SPR) >> 6;      // = MT * NKS, a multiple of 4
        if constexpr (WPB == 1) {
            // single-wave workgroups (few rows): only the instructions that hold a row below M -- the 5 rows of a sequential window are 13 of
            // the tile's 40 KB; rows past M stay whatever LDS held (their results are never stored: an MFMA output column depends on its
            // own row only).  A run-time loop: no instruction sits under a branch of its own
            const int rows_here = g.M - r0 < MT * 16 ? g.M - r0 : MT * 16;
            const int n_need = g.w1_full_tile ? n_instr : (rows_here * SPR + 63) >> 6;
            for (int q = 0; q < n_need; ++q) {
                const int p = q * 64 + lane;
                const int row = p / SPR, ps = p - row * SPR;
                const int kslot = ps ^ (row & 15);
                const int gr = r0 + row < g.M ? r0 + row : g.M - 1;
                const f16 *src = g.A + (size_t)gr * g.lda + k0 + kslot * 8;
                __builtin_amdgcn_global_load_lds(src, (lds_void *)(smem + q * 1024), 16, 0, 0);
            }
        } else {
#pragma unroll
        for (int j = 0; j < n_instr / WPB; ++j) {
            const int q = j * WPB + dwave;
            const int p = q * 64 + lane;                    // 16-byte slot index of this lane's destination
            const int row = p / SPR, ps = p - row * SPR;
            const int kslot = ps ^ (row & 15);              // logical slot that must land there (swizzle on the source)
            const int gr = r0 + row < g.M ? r0 + row : g.M - 1;
            const f16 *src = g.A + (size_t)gr * g.lda + k0 + kslot * 8;
            __builtin_amdgcn_global_load_lds(src, (lds_void *)(smem + q * 1024), 16, 0, 0);
        }
        }
    }
    // ---- weights of this wave's 16 columns: every fragment of the slice in flight at once (HBM, or L2 behind a sibling).
    //      The weights are stored PRE-PACKED in MFMA-fragment order (swx_fold_ln at load time): the 1 KB block of
    //      (16-column group, k-step) holds lane l's 16 bytes at l*16, so a wave instruction reads 8 full 128-byte lines.
    //      (Row-major [N][K] weights make every instruction touch 16 separate 64-byte row pieces: the per-CU address /
    //      L1 path, not HBM, bounded the first version of this kernel -- hot or cold weights cost the same ~4 us.)
    const f16 *wp = g.W + ((size_t)(panel * 4 + wave) * (g.K >> 5) + (size_t)ks_id * NKS) * 512 + lane * 8;
    // The loads are inline asm so that hipcc does not count them: with an LDS-DMA in flight it waits vmcnt(0) at the first
    // use of any ordinary load result (cdna_hip_programming.md 5, trap (b)), i.e. the whole weight stream would have to land
    // before the first MFMA.  Their completion is counted by hand below: k-step ks waits until at most (NKS - 1 - ks) weight
    // loads + the epilogue loads issued after them are pending.  (Audited by tests/test_kernel_isa_cpu.py: no compiler
    // v_mov / spill of wf[] between a load and its wait -- 5.7 item 1.)
    f16x8 wf[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
        asm volatile("global_load_dwordx4 %0, %1, off offset:%2" SWX_DEC_W_POLICY : "=v"(wf[ks]) : "v"(wp + (ks >> 2) * 2048), "n"((ks & 3) * 1024) : "memory");
    // ---- epilogue operands (clamped addresses, never predicated): column constants and the residual rows
    constexpr int N_EPI = (E_FIN ? 1 : 0) + (E_LN ? 1 : 0) + ((E_RES && E_FIN) ? MT : 0);   // loads younger than the weights
    const int n = panel * 64 + wave * 16 + lg * 4;
    const int nc = n < g.N ? n : g.N - 4;
    f32x4 c2 = (f32x4){0.f, 0.f, 0.f, 0.f}, c1 = (f32x4){0.f, 0.f, 0.f, 0.f};
    if constexpr (E_FIN) c2 = *(const f32x4 *)(g.c2 + nc);
    if constexpr (E_LN) c1 = *(const f32x4 *)(g.c1 + nc);
    f16x4 xres[MT];
    if constexpr (E_RES && E_FIN) {
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            const int m = r0 + t * 16 + li;
            xres[t] = *(const f16x4 *)(g.X + (size_t)(m < g.M ? m : g.M - 1) * g.ldx + nc);
        }
    }
    // ---- cache prefetch of the NEXT projection's weights: DEC_NPF more loads per wave, the youngest in the queue, whose results
    //      nobody reads (kept in registers until the wave's last wait so that the allocator cannot hand those registers out
    //      while a load is still in flight).  The r02 ablation measured -1.0 .. -2.0 us per launch with warm instead of HBM-cold
    //      weights; here the previous kernel of the chain does the warming (measured -0.8 .. -1.2 us on the consumer).
    unsigned pfv[DEC_NPF];
    {
        // lines are dealt over the lanes of the workgroups that got a unit, in the order (unit, row group, wave, lane); a launch
        // with fewer lanes than the next projection has lines covers a prefix of it, lanes past the end re-touch the last line
        const unsigned char *pbase = g.pf.base ? (const unsigned char *)g.pf.base : (const unsigned char *)g.W;
        const int plines = g.pf.base ? g.pf.lines : (g.N >> 6) * (g.K >> 6) * 64;          // N * K * 2 / 128 (own weights: already in flight)
        const int t = ((unit * g.n_rg + rg) * 4 + wave) * 64 + lane;
        const int stride = panels * g.ks2 * g.n_rg * 256;
#pragma unroll
        for (int i = 0; i < DEC_NPF; ++i) {
            int line = t + i * stride;
            line = line < plines ? line : plines - 1;
            const unsigned char *pa = pbase + (size_t)line * 128;
            asm volatile("global_load_dword %0, %1, off" : "=v"(pfv[i]) : "v"(pa) : "memory");
        }
    }
    {
        // the DMA instructions are the oldest: they have landed once at most (weights + epilogue + prefetch loads) younger loads are pending
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NKS + N_EPI + DEC_NPF) : "memory");
        __builtin_amdgcn_s_barrier();
    }

    // ---- LayerNorm statistics of the tile's rows (complete rows: K == kslice == d), 16 lanes per row
    float2 *stat = (float2 *)(smem + (size_t)MT * 16 * RS);
    if constexpr (E_LN) {
        const f16x2 one2 = {(f16)1.f, (f16)1.f};
        for (int rb = dwave * 4 + lg; rb < MT * 16; rb += 4 * WPB) {
            const unsigned char *rp = smem + (size_t)rb * RS + li * 16;
            f16x8 v[SPR / 16];                      // the row's share of this lane: every read in flight before the first add
#pragma unroll
            for (int i = 0; i < SPR / 16; ++i) v[i] = *(const f16x8 *)(rp + i * 256);
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < SPR / 16; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const f16x2 pr = {v[i][2 * e], v[i][2 * e + 1]};
                    s1 = __builtin_amdgcn_fdot2(pr, one2, s1, false);
                    s2 = __builtin_amdgcn_fdot2(pr, pr, s2, false);
                }
#pragma unroll
            for (int o = 0; o < 1; ++o) {        // 1, 2, 4, 8 in this order (DPP exchanges: swx_common.h)
                s1 += lane_xor<1>(s1, lane); s2 += lane_xor<1>(s2, lane); s1 += lane_xor<2>(s1, lane); s2 += lane_xor<2>(s2, lane);
                s1 += lane_xor<4>(s1, lane); s2 += lane_xor<4>(s2, lane); s1 += lane_xor<8>(s1, lane); s2 += lane_xor<8>(s2, lane);
            }
            if (li == 0) {
                const float inv = 1.0f / (float)kslice;
                const float mean = s1 * inv;
                float var = s2 * inv - mean * mean;
                var = var > 0.f ? var : 0.f;
                // inline asm store: an LDS store the compiler can see is ordered behind the LDS-DMA it believes to be pending
                // (it would wait vmcnt(0) here, i.e. for the whole weight stream)
                const float2 sv = make_float2(mean, 1.0f / sqrtf(var + 1e-5f));
                const unsigned lds_addr = (unsigned)(uintptr_t)(lds_void *)(stat + rb);
                asm volatile("ds_write_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"(lds_addr), "v"(sv) : "memory");
            }
        }
    }

    // ---- MFMA: D[i = column][j = row] += W-fragment . A-fragment^T ; the activation fragments of the next PF k-steps are
    //      requested from LDS before the current step's MFMAs (the compiler does not hoist them by itself: measured in the
    //      ISA as read -> wait -> mfma per step, i.e. one exposed LDS latency per k-step)
    f32x4 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const unsigned char *abase = smem + (size_t)li * RS;
    constexpr int PF = MT == 1 ? 8 : 4;        // k-steps of fragment reads in flight: with one row tile a step is ONE MFMA (16 cycles),
    f16x8 af[PF][MT];                          // four steps ahead do not cover an LDS round trip, eight do
    auto lds_frag = [&](int ks, int t) {
        const int phys = (ks * 4 + lg) ^ li;
        return *(const f16x8 *)(abase + (size_t)t * 16 * RS + phys * 16);
    };
#pragma unroll
    for (int ks = 0; ks < PF; ++ks)
#pragma unroll
        for (int t = 0; t < MT; ++t) af[ks][t] = lds_frag(ks, t);
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        // this k-step's weight fragment has landed once no more than the younger loads are pending (counter is 6 bits wide)
        asm volatile("s_waitcnt vmcnt(%1)" : "+v"(wf[ks]) : "n"((NKS - 1 - ks) + N_EPI + DEC_NPF) : "memory");
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ks], af[ks % PF][t], acc[t], 0, 0, 0);
            if (ks + PF < NKS) af[ks % PF][t] = lds_frag(ks + PF, t);
        }
        __builtin_amdgcn_sched_barrier(0);        // keeps the prefetch PF steps ahead (the scheduler sinks it back otherwise)
    }
    if constexpr (E_LN) __syncthreads();          // stat[] visible to every wave

    // ---- epilogue: lane holds columns n .. n+3 of row m for every tile.  Every load of the wave has landed from here on (the
    //      prefetch loads were issued right behind the weights); their destination registers stay allocated up to this point.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < DEC_NPF; ++i) asm volatile("" ::"v"(pfv[i]));
    if (n >= g.N) return;                          // N % 4 == 0 is checked by the launcher
    if constexpr (E_SLAB && !E_TICKET) {
        float *out = g.slabs + (size_t)ks_id * g.slab_stride;
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            const int m = r0 + t * 16 + li;
            if (m < g.M) *(f32x4 *)(out + (size_t)m * g.N + n) = acc[t];
        }
        return;
    }
    if constexpr (E_TICKET) {
        // Round 5: the reduction of the K slices inside the launch (cdna_hip_programming.md, in-launch split-K reduction).  The K
        // slices of a (panel, row group) run on different XCDs (unit -> XCD above); nothing here depends on where they run.
        // Second form (the first -- plain slab stores, agent-scope release fence = buffer_wbl2 of the XCD's whole L2 in every one of
        // the 240 workgroups, acquire fence = buffer_inv in the last arriver -- made the launch 17.0 us against 8.2 + 5.0 us for the
        // slab GEMM and its finish launch: profiles/r05_ticket_on_kernels.csv): the slabs are PUBLISHED WRITE-THROUGH (16-byte
        // `sc1` stores: at vmcnt(0) they are in memory, no L2 write-back) -> every wave drains -> barrier -> ONE lane draws the relaxed
        // agent-scope ticket; the last arriver reads every slab with `sc1` loads (coherent reads: no invalidate) and reduces with
        // dec_slab_finish's arithmetic, operation for operation: 0 + slab 0 + slab 1 + ... + bias, then f16(sum + x) -- bit-identical
        // (tests/test_gpu_kernels.py, SWX_FLAG_TICKET).  The counter is zero before the first launch (swx_bind_workspace) and is put
        // back by the last arriver; the next launch that uses it is behind a kernel boundary.
        float *out = g.slabs + (size_t)ks_id * g.slab_stride;
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            const int m = r0 + t * 16 + li;
            const float *dstp = out + (size_t)(m < g.M ? m : g.M - 1) * g.N + n;      // rows past M: the last row's own value again
            const f32x4 val = acc[t];
            if (m < g.M) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dstp), "v"(val) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                   // every wave's slab stores are in memory; the activation tile is dead
        int *last_flag = (int *)smem;
        if (tid == 0) {
            int *cnt = g.ticket + panel * g.n_rg + rg;
            const int tk = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = tk == g.ks2 - 1 ? 1 : 0;
            if (last) __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *last_flag = last;
        }
        __syncthreads();
        if (!*last_flag) return;
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            const int m = r0 + t * 16 + li;
            const float *sp = g.slabs + (size_t)(m < g.M ? m : g.M - 1) * g.N + n;
            f32x4 part[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)                      // all in flight; coherent (sc1) reads of what the other XCDs wrote through
                asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(part[k]) : "v"(sp + (size_t)(k < g.ks2 ? k : g.ks2 - 1) * g.slab_stride) : "memory");
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(part[0]), "+v"(part[1]), "+v"(part[2]), "+v"(part[3]) : : "memory");
            f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 4; ++k) if (k < g.ks2) a += part[k];
            a += c2;
            if (m >= g.M) continue;
            f16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (f16)(a[e] + (float)xres[t][e]);
            *(f16x4 *)(g.X + (size_t)m * g.ldx + n) = o;
        }
        return;
    }
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int m = r0 + t * 16 + li;
        if (m >= g.M) continue;
        f32x4 v = acc[t];
        if constexpr (E_LN) {
            const float2 st = stat[t * 16 + li];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = st.y * (v[e] - st.x * c1[e]) + c2[e];
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += c2[e];
        }
        if constexpr (E_GELU) {
            { f32x2 g0 = {v[0], v[1]}, g1 = {v[2], v[3]}; g0 = gelu_erf2(g0); g1 = gelu_erf2(g1); v[0] = g0[0]; v[1] = g0[1]; v[2] = g1[0]; v[3] = g1[1]; }
        }
        f16 *dst;
        if constexpr (E_RES) {
            dst = g.X + (size_t)m * g.ldx + n;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += (float)xres[t][e];
        } else if (E_QKV && n >= g.d) {
            const int rps = g.rps > 1 ? g.rps : 1;
            const int seq = m / rps, crow = seq * (g.row_mul > 1 ? g.row_mul : 1), pos = g.pos0[crow] + (m - seq * rps);
            f16 *cache = n < 2 * g.d ? g.kcache : g.vcache;
            dst = cache + ((size_t)crow * g.n_ctx + pos) * g.d + (n < 2 * g.d ? n - g.d : n - 2 * g.d);
        } else {
            dst = g.C + (size_t)m * g.ldc + n;
        }
        f16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (f16)v[e];
        *(f16x4 *)dst = o;
    }
}

// ------------------------------------------------------------------------------------------------- tall variant
// The same projection for MANY rows (a multi-token pass over a whole batch: the scoring pass of 20 windows is 2 260 rows, a
// prefill with prompts several thousand).  The kernel above gives every 48 rows x 64 columns their own workgroup, which then
// re-reads its 160 KB of weights and waits out one DMA round trip for 0.3 us of MFMAs; here a workgroup keeps its four waves'
// weight fragments in REGISTERS and walks a run of 16-row tiles through three LDS buffers: the LDS-DMAs of tiles t + 1 and
// t + 2 (issued from inline asm, so that hipcc neither sees nor drains them) fly under the statistics, MFMAs and epilogue of tile t.
// Per output element the arithmetic is the kernel's above, instruction for instruction -- the LayerNorm statistics from the
// same 16-lanes-per-row dot products, the k-steps in the same order into one accumulator, the same epilogue expressions --
// so a row's result does not depend on which of the two kernels, or how many rows, the launch had
// (tests/hw_checks/dec_tall_check.py: bit-identical; tests/test_gpu_batch_invariance.py end to end).
__device__ __forceinline__ void dec_glds16_asm(const void *gsrc, unsigned lds_dst)
{
    unsigned keep;     // M0 is the DMA's LDS base and belongs to hipcc: saved and restored inside the statement
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// WPB = 8 (round 6): a workgroup = TWO neighbouring 64-column panels (eight waves, two per SIMD) walking the same run of tiles through the
// same three LDS buffers: a tile's serial chain -- barrier, statistics, 40 dependent MFMAs behind LDS reads, barrier, epilogue -- is ~90 %
// latency (0.27 us of MFMA per wave in 1.4-2.3 us per tile), so a second wave per SIMD fills it.  Per wave nothing changes: bit-identical.
template <int NKS, int EPI, int WPB = 4>
__global__ __launch_bounds__(64 * WPB) void gemm_dectall_f16(DecGemmArgs g)
{
    constexpr bool E_LN = (EPI & DEC_LN) != 0, E_GELU = (EPI & DEC_GELU) != 0, E_RES = (EPI & DEC_RES) != 0,
                   E_QKV = (EPI & DEC_QKV) != 0, E_SLAB = (EPI & DEC_SLAB) != 0;
    constexpr int kslice = NKS * 32, SPR = kslice >> 3, RS = kslice * 2, TILE = 16 * RS;
    constexpr int DMA_PER_WAVE = NKS / WPB;        // 16 rows x SPR slots / 64 lanes = NKS instructions per tile, dealt over the waves
    static_assert(WPB == 4 || WPB == 8, "one or two panels per workgroup");
    static_assert(NKS % WPB == 0, "tile = whole DMA instructions per wave");
    constexpr int NBUF = 3;                        // tile t computes while tiles t + 1 and t + 2 are on their way
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];   // NBUF x [16][kslice] f16 | float2 stat[NBUF][16]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int li = lane & 15, lg = lane >> 4;
    const int b = blockIdx.x;
    const int xcd = b & 7, slot = b >> 3;
    const int unit = (slot / g.n_rg) * 8 + xcd, rsplit = slot % g.n_rg;     // g.n_rg = row splits here
    constexpr int PW = WPB / 4;                                              // panels per workgroup
    const int panels = ((g.N + 63) >> 6) / PW;                              // (WPB = 8: the launcher guarantees an even panel count)
    if (unit >= panels * g.ks2) return;
    const int panel_u = unit / g.ks2, ks_id = unit - panel_u * g.ks2;
    const int panel = panel_u * PW + (wave >> 2);
    const int cw = wave & 3;                                                 // this wave's 16-column group of its panel
    const int k0 = ks_id * kslice;
    const int n_tiles = (g.M + 15) >> 4;
    const int t_begin = rsplit * g.tps;
    const int t_end = t_begin + g.tps < n_tiles ? t_begin + g.tps : n_tiles;
    if (t_begin >= t_end) return;
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void *)smem;

    auto stage = [&](int tile, int buf) {
#pragma unroll
        for (int j = 0; j < DMA_PER_WAVE; ++j) {
            const int q = j * WPB + wave_u;
            const int p = q * 64 + lane;
            const int row = p / SPR, ps = p - row * SPR;
            const int kslot = ps ^ (row & 15);
            const int gr = tile * 16 + row < g.M ? tile * 16 + row : g.M - 1;
            dec_glds16_asm(g.A + (size_t)gr * g.lda + k0 + kslot * 8, lds0 + buf * TILE + q * 1024);
        }
    };
    stage(t_begin, 0);
    stage(t_begin + 1 < t_end ? t_begin + 1 : t_begin, 1);
    // this wave's 16 columns of weights, resident for the whole run of tiles
    const f16 *wp = g.W + ((size_t)(panel * 4 + cw) * (g.K >> 5) + (size_t)ks_id * NKS) * 512 + lane * 8;
    f16x8 wf[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) wf[ks] = *(const f16x8 *)(wp + (size_t)ks * 512);
    const int n = panel * 64 + cw * 16 + lg * 4;
    const int nc = n < g.N ? n : g.N - 4;
    f32x4 c2 = (f32x4){0.f, 0.f, 0.f, 0.f}, c1 = (f32x4){0.f, 0.f, 0.f, 0.f};
    if constexpr (!E_SLAB) c2 = *(const f32x4 *)(g.c2 + nc);
    if constexpr (E_LN) c1 = *(const f32x4 *)(g.c1 + nc);
    float2 *stat = (float2 *)(smem + NBUF * TILE);
    const int rps = g.rps > 1 ? g.rps : 1;
    // hipcc's waits for the loads it can see (weights, column constants) belong in FRONT of the loop: a wait placed at their first
    // use inside the body would run every iteration and drain the next tile's DMA with it
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) asm volatile("" : "+v"(wf[ks]));
    asm volatile("" : "+v"(c1), "+v"(c2));

    for (int t = t_begin; t < t_end; ++t) {
        const int buf = (t - t_begin) % NBUF;
        const int m = t * 16 + li, mc = m < g.M ? m : g.M - 1;
        // tile t has landed: every wave waits for its own share of the DMA, the barrier makes it true for all.  Younger than tile
        // t's DMA in this wave's queue, and allowed to be still in flight: the previous-but-one tile's store, tile t + 1's DMA, the
        // previous tile's store (vector memory operations retire in issue order on gfx9-class hardware -- what hipcc's own
        // counted waits rely on); the previous tile's operand loads were waited for already.  (First tile: hipcc's waits in
        // front of the loop drained everything, so the count is trivially met.)
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(DMA_PER_WAVE + 2) : "memory");
        // epilogue operands of this tile, requested before the next tile's DMA so that a counted wait can tell them apart
        f16x4 xres = {(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
        int posv = 0;
        if constexpr (E_RES && !E_SLAB)
            asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(xres) : "v"(g.X + (size_t)mc * g.ldx + nc) : "memory");
        if constexpr (E_QKV) {
            const int seq = mc / rps, crow = seq * (g.row_mul > 1 ? g.row_mul : 1);
            asm volatile("global_load_dword %0, %1, off" : "=v"(posv) : "v"(g.pos0 + crow) : "memory");
        }
        stage(t + 2 < t_end ? t + 2 : t_end - 1, (t - t_begin + 2) % NBUF);    // (past the end: the last tile again, into the buffer
                                                                               //  tile t - 1 has left: never predicated, never read)
        const unsigned char *tile = smem + buf * TILE;
        if (E_LN && (WPB == 4 || wave_u < 4)) {            // (eight waves: the first four compute the 16 rows' statistics)
            const f16x2 one2 = {(f16)1.f, (f16)1.f};
            const int rb = wave * 4 + lg;                 // 16 lanes per row, one row per (wave, lane group)
            const unsigned char *rp = tile + (size_t)rb * RS + li * 16;
            // (eight waves: the row's reads in batches of four -- 256 registers per wave; the sums keep their order)
            constexpr int NV = SPR / 16, VB = WPB == 8 ? 4 : NV;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i0 = 0; i0 < NV; i0 += VB) {
                f16x8 v[VB];
#pragma unroll
                for (int i = 0; i < VB; ++i) if (i0 + i < NV) v[i] = *(const f16x8 *)(rp + (i0 + i) * 256);
#pragma unroll
                for (int i = 0; i < VB; ++i)
                    if (i0 + i < NV) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const f16x2 pr = {v[i][2 * e], v[i][2 * e + 1]};
                            s1 = __builtin_amdgcn_fdot2(pr, one2, s1, false);
                            s2 = __builtin_amdgcn_fdot2(pr, pr, s2, false);
                        }
                    }
                if (WPB == 8) __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int o = 0; o < 1; ++o) {        // 1, 2, 4, 8 in this order (DPP exchanges: swx_common.h)
                s1 += lane_xor<1>(s1, lane); s2 += lane_xor<1>(s2, lane); s1 += lane_xor<2>(s1, lane); s2 += lane_xor<2>(s2, lane);
                s1 += lane_xor<4>(s1, lane); s2 += lane_xor<4>(s2, lane); s1 += lane_xor<8>(s1, lane); s2 += lane_xor<8>(s2, lane);
            }
            if (li == 0) {
                const float inv = 1.0f / (float)kslice;
                const float mean = s1 * inv;
                float var = s2 * inv - mean * mean;
                var = var > 0.f ? var : 0.f;
                stat[buf * 16 + rb] = make_float2(mean, 1.0f / sqrtf(var + 1e-5f));
            }
        }
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        const unsigned char *abase = tile + (size_t)li * RS;
        constexpr int PF = WPB == 8 ? 4 : 8;      // one MFMA per k-step here (16 cycles): eight fragment reads ahead cover the LDS latency (four with a second wave on the SIMD: 256 registers)
        f16x8 af[PF];
#pragma unroll
        for (int ks = 0; ks < PF; ++ks) af[ks] = *(const f16x8 *)(abase + (((ks * 4 + lg) ^ li) << 4));
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ks], af[ks % PF], acc, 0, 0, 0);
            if (ks + PF < NKS) af[ks % PF] = *(const f16x8 *)(abase + ((((ks + PF) * 4 + lg) ^ li) << 4));
            __builtin_amdgcn_sched_barrier(0);    // keeps the reads PF steps ahead (the scheduler sinks each one back in front of
        }                                         // its MFMA otherwise: read -> wait -> mfma, 40 exposed LDS latencies per tile)
        if constexpr (E_LN) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // stat[] visible to every wave
        // the operand loads are older than the next tile's DMA: they have landed once only the DMA is pending
        if constexpr ((E_RES && !E_SLAB) || E_QKV)
            asm volatile("s_waitcnt vmcnt(%2)" : "+v"(xres), "+v"(posv) : "n"(DMA_PER_WAVE) : "memory");
        if (n >= g.N || m >= g.M) continue;
        if constexpr (E_SLAB) {
            *(f32x4 *)(g.slabs + (size_t)ks_id * g.slab_stride + (size_t)m * g.N + n) = acc;
            continue;
        }
        f32x4 v = acc;
        if constexpr (E_LN) {
            const float2 st = stat[buf * 16 + li];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = st.y * (v[e] - st.x * c1[e]) + c2[e];
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += c2[e];
        }
        if constexpr (E_GELU) {
            { f32x2 g0 = {v[0], v[1]}, g1 = {v[2], v[3]}; g0 = gelu_erf2(g0); g1 = gelu_erf2(g1); v[0] = g0[0]; v[1] = g0[1]; v[2] = g1[0]; v[3] = g1[1]; }
        }
        f16 *dst;
        if constexpr (E_RES) {
            dst = g.X + (size_t)m * g.ldx + n;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += (float)xres[e];
        } else if (E_QKV && n >= g.d) {
            const int seq = m / rps, crow = seq * (g.row_mul > 1 ? g.row_mul : 1), pos = posv + (m - seq * rps);
            f16 *cache = n < 2 * g.d ? g.kcache : g.vcache;
            dst = cache + ((size_t)crow * g.n_ctx + pos) * g.d + (n < 2 * g.d ? n - g.d : n - 2 * g.d);
        } else {
            dst = g.C + (size_t)m * g.ldc + n;
        }
        f16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (f16)v[e];
        *(f16x4 *)dst = o;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the idle re-stage of the last tile must not outlive the workgroup's LDS
}

// x[m][n] = f16(x + bias[n] + sum_k slab[k][m][n]) : the reduction of the one projection that stays K-split (K = 4d)
template <int KS>
__global__ __launch_bounds__(256) void dec_slab_finish(const float *__restrict__ slabs, int64_t stride, int ks2, const float *__restrict__ bias,
                                                       f16 *__restrict__ X, int64_t ldx, int M, int N)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int n4 = N >> 2;
    if (idx >= M * n4) return;
    const int m = idx / n4, n = (idx - m * n4) * 4;
    const float *sp = slabs + (size_t)m * N + n;
    f32x4 part[KS];
#pragma unroll
    for (int k = 0; k < KS; ++k) part[k] = *(const f32x4 *)(sp + (size_t)(k < ks2 ? k : ks2 - 1) * stride);   // all in flight
    f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < KS; ++k) if (k < ks2) a += part[k];
    a += *(const f32x4 *)(bias + n);
    f16 *xp = X + (size_t)m * ldx + n;
    const f16x4 xv = *(const f16x4 *)xp;
    f16x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (f16)(a[e] + (float)xv[e]);
    *(f16x4 *)xp = o;
}

// load time: Wp = pack(f16(W[n][k] * gamma[k]));  c1[n] = sum_k Wf[n][k];  c2[n] = bias[n] + sum_k beta[k] * W[n][k]
// (gamma == null: plain re-pack, c1 / c2 untouched).  Packed layout: [N/16][K/32][64 lanes][8 halfs] with
// lane = ((k % 32) / 8) * 16 + n % 16 -- the operand fragment order of v_mfma_f32_16x16x32_f16.
__global__ __launch_bounds__(256) void fold_pack_kernel(const f16 *__restrict__ W, const float *__restrict__ gamma,
                                                        const float *__restrict__ beta, const float *__restrict__ bias,
                                                        f16 *__restrict__ Wp, float *__restrict__ c1, float *__restrict__ c2, int K)
{
    __shared__ float sh[2][4];
    const int n = blockIdx.x;
    float s1 = 0.f, s2 = 0.f;
    const size_t grp = (size_t)(n >> 4) * (K >> 5);
    for (int k = threadIdx.x; k < K; k += 256) {
        const float w = (float)W[(size_t)n * K + k];
        const f16 wf = gamma ? (f16)(w * gamma[k]) : (f16)w;
        const int ks = k >> 5, lane = ((k & 31) >> 3) * 16 + (n & 15);
        Wp[((grp + ks) * 64 + lane) * 8 + (k & 7)] = wf;
        s1 += (float)wf;
        if (beta) s2 += beta[k] * w;
    }
    if (!gamma) return;
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = s1; sh[1][threadIdx.x >> 6] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        c1[n] = (sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3]);
        c2[n] = (bias ? bias[n] : 0.f) + ((sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3]));
    }
}

}  // namespace

int swx_dec_plan(int M, int N, int K, int epi, int *mt_out, int *ks2_out)
{
    if (M <= 0 || N <= 0 || N % 64 != 0 || K % 128 != 0) return -4;
    // K slice: un-split when it fits one tile row (<= 1280 deep), else the fewest slices of <= 1280 that are a multiple of 128
    int ks2 = 1;
    auto depth_ok = [](int ks) { return ks == 384 || ks == 512 || ks == 640 || ks == 768 || ks == 1024 || ks == 1280; };
    while (K % ks2 != 0 || !depth_ok(K / ks2)) { if (++ks2 > 64) return -4; }
    if ((epi & DEC_LN) && ks2 != 1) return -4;       // the statistics need complete rows
    if (ks2 > 1 && !(epi & DEC_SLAB)) return -4;
    const int kslice = K / ks2;
    const int panels = N / 64;
    // rows per workgroup: the smallest MT whose grid fits one round of 256 workgroups, within 120 KB of LDS
    int mt = 1;
    while (mt < DEC_MAXMT && (int64_t)(mt + 1) * 16 * kslice * 2 <= 122880 && panels * ks2 * cdiv(M, mt * 16) > 256) ++mt;
    *mt_out = mt; *ks2_out = ks2;
    return 0;
}

DecPrefetch swx_dec_prefetch_of(const void *packed_w, int M, int N, int K, int epi)
{
    DecPrefetch pf{};
    int mt = 1, ks2 = 1;
    if (!packed_w || swx_dec_plan(M, N, K, epi, &mt, &ks2) < 0) return pf;        // only shapes the dec kernels run have packed weights
    pf.base = packed_w; pf.lines = (int)(((int64_t)N * K * 2) >> 7);
    return pf;
}

size_t swx_dec_slab_floats(int M, int N, int K)
{
    int mt, ks2;
    if (swx_dec_plan(M, N, K, DEC_SLAB, &mt, &ks2) < 0) return 0;
    return ks2 > 1 ? (size_t)ks2 * M * N : 0;
}

int swx_gemm_dec(DecGemmArgs g, hipStream_t s)
{
    if (g.M <= 0 || g.N <= 0) return 0;
    if (g.lda % 8 != 0 || ((uintptr_t)g.A & 15) || ((uintptr_t)g.W & 15)) return -4;      // g.W: packed (swx_fold_ln)
    int mt = 1, ks2 = 1;
    { const int rc = swx_dec_plan(g.M, g.N, g.K, g.epi, &mt, &ks2); if (rc < 0) return rc; }
    if (ks2 > 1 && !g.slabs) return -4;
    if (ks2 == 1 && (g.epi & DEC_SLAB)) g.epi &= ~DEC_SLAB;      // un-split after all: the kernel finishes the output itself
    g.epi &= ~DEC_TICKET;
    g.ks2 = ks2; g.kslice = g.K / ks2; g.n_rg = cdiv(g.M, mt * 16);
    g.slab_stride = (int64_t)g.M * g.N;
    const int nks = g.kslice / 32;
    const int units = (g.N / 64) * ks2;
    int epi = g.epi & (DEC_LN | DEC_GELU | DEC_RES | DEC_QKV | DEC_SLAB);
    const bool use_tall = g.tall && g.M > 160 && (g.kslice / 32) % 4 == 0 && !(swx_flags() & SWX_FLAG_NO_TALL);
    // SWX_FLAG_TICKET: the K-split projection of the decode-step kernel reduces its slabs inside the launch (measured slower: swx_kernels.h)
    const bool ticket = ks2 > 1 && ks2 <= 4 && !use_tall && g.ticket && epi == (DEC_RES | DEC_SLAB) && g.X &&
                        (g.N / 64) * cdiv(g.M, mt * 16) <= SWX_DEC_TICKETS && (swx_flags() & SWX_FLAG_TICKET) &&
                        g.N % 64 == 0;      // the kernel's early `n >= N` exit sits in front of a block barrier: whole panels only
    if (ticket) epi |= DEC_TICKET;
    if (use_tall) {
        // tall kernel: ~two rounds of the chip's 256 CUs, every workgroup a run of `tps` 16-row tiles
        const int n_tiles = cdiv(g.M, 16);
        // eight waves = two panels per workgroup, from 40 (panel, K slice) units on (the un-split N = 1280 projections measured 27.4 vs 26.5 us
        // that way: 10 double panels leave the row splits too short); bit-identical; SWX_FLAG_TALL_NO_W8: A/B
        const bool w8 = !(swx_flags() & SWX_FLAG_TALL_NO_W8) && (g.N / 64) % 2 == 0 && nks % 8 == 0 && (g.N / 64) * ks2 >= 40;
        const int units = w8 ? (g.N / 128) * ks2 : (g.N / 64) * ks2;
        int rs = 512 / (cdiv(units, 8) * 8);
        if (rs < 1) rs = 1;
        if (rs > n_tiles) rs = n_tiles;
        g.tps = cdiv(n_tiles, rs);
        g.n_rg = cdiv(n_tiles, g.tps);
        const int grid = cdiv(units, 8) * g.n_rg * 8;
        const size_t lds = (size_t)3 * 16 * g.kslice * 2 + 3 * 16 * sizeof(float2);
        SwxProfScope prof(PC_GEMM_SKINNY, 2.0 * ((double)g.N * g.K + (double)g.M * g.K) + (double)g.M * g.N * 2, s);
#define SWX_TALL(NK_, EP_) do { \
        static bool attr_done = false; \
        if (!attr_done) { \
            hipError_t e_ = hipFuncSetAttribute((const void *)gemm_dectall_f16<NK_, EP_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024); \
            if (e_ != hipSuccess) return -100 - (int)e_; \
            attr_done = true; \
        } \
        hipLaunchKernelGGL((gemm_dectall_f16<NK_, EP_>), dim3(grid), dim3(256), lds, s, g); } while (0)
#define SWX_TALL8(NK_, EP_) do { \
        static bool attr_done8 = false; \
        if (!attr_done8) { \
            hipError_t e_ = hipFuncSetAttribute((const void *)gemm_dectall_f16<NK_, EP_, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024); \
            if (e_ != hipSuccess) return -100 - (int)e_; \
            attr_done8 = true; \
        } \
        hipLaunchKernelGGL((gemm_dectall_f16<NK_, EP_, 8>), dim3(grid), dim3(512), lds, s, g); } while (0)
#define SWX_TALL_NK(EP_) do { if (w8) { switch (nks) { \
        case 16: SWX_TALL8(16, EP_); break; case 24: SWX_TALL8(24, EP_); break; case 32: SWX_TALL8(32, EP_); break; case 40: SWX_TALL8(40, EP_); break; \
        default: return -4; } } else { switch (nks) { \
        case 12: SWX_TALL(12, EP_); break; case 16: SWX_TALL(16, EP_); break; case 20: SWX_TALL(20, EP_); break; \
        case 24: SWX_TALL(24, EP_); break; case 32: SWX_TALL(32, EP_); break; case 40: SWX_TALL(40, EP_); break; \
        default: return -4; } } } while (0)
        switch (epi) {
            case DEC_LN | DEC_QKV: SWX_TALL_NK(DEC_LN | DEC_QKV); break;
            case DEC_RES: SWX_TALL_NK(DEC_RES); break;
            case DEC_LN: SWX_TALL_NK(DEC_LN); break;
            case DEC_LN | DEC_GELU: SWX_TALL_NK(DEC_LN | DEC_GELU); break;
            case DEC_RES | DEC_SLAB: SWX_TALL_NK(DEC_RES | DEC_SLAB); break;
            default: return -4;
        }
#undef SWX_TALL_NK
#undef SWX_TALL8
#undef SWX_TALL
    } else {
    const int grid = cdiv(units, 8) * g.n_rg * 8;
    const size_t lds = (size_t)mt * 16 * g.kslice * 2 + (size_t)mt * 16 * sizeof(float2);
    // few workgroups (<= 80 of four waves: the 5 rows of a sequential window's decode step): the same waves as single-wave workgroups,
    // spread over four times as many CUs (gemm_dec_f16<.., WPB = 1>; bit-identical; SWX_FLAG_DEC_NO_W1: A/B)
    const bool w1 = mt == 1 && !ticket && units * g.n_rg <= 80 && !(swx_flags() & SWX_FLAG_DEC_NO_W1);
    const int grid1 = cdiv(units * 4, 8) * g.n_rg * 8;
    g.w1_full_tile = (swx_flags() & SWX_FLAG_DEC_W1_FULL_TILE) ? 1 : 0;
    {   // (profiler scopes must not nest: each one closes the most recent record)
    SwxProfScope prof(PC_GEMM_SKINNY, 2.0 * ((double)g.N * g.K + (double)g.M * g.K) + (double)g.M * g.N * 2, s);
    // the epilogues the decoder step uses (compile-time): QKV, out-projections, cross-q, MLP-in, MLP-out (split / un-split)
#define SWX_DEC(MT_, NK_, EP_) do { \
        static bool attr_done = false; \
        if (!attr_done) { \
            hipError_t e_ = hipFuncSetAttribute((const void *)gemm_dec_f16<MT_, NK_, EP_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024); \
            if (e_ != hipSuccess) return -100 - (int)e_; \
            attr_done = true; \
        } \
        hipLaunchKernelGGL((gemm_dec_f16<MT_, NK_, EP_>), dim3(grid), dim3(256), lds, s, g); } while (0)
#define SWX_DEC_W1(NK_, EP_) do { \
        static bool attr_done1 = false; \
        if (!attr_done1) { \
            hipError_t e_ = hipFuncSetAttribute((const void *)gemm_dec_f16<1, NK_, EP_, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024); \
            if (e_ != hipSuccess) return -100 - (int)e_; \
            attr_done1 = true; \
        } \
        hipLaunchKernelGGL((gemm_dec_f16<1, NK_, EP_, 1>), dim3(grid1), dim3(64), lds, s, g); } while (0)
#define SWX_DEC_MT4(NK_, EP_) do { if (mt == 1) SWX_DEC(1, NK_, EP_); else if (mt == 2) SWX_DEC(2, NK_, EP_); else SWX_DEC(3, NK_, EP_); } while (0)
#define SWX_DEC_MT(NK_, EP_) do { if (w1) SWX_DEC_W1(NK_, EP_); else SWX_DEC_MT4(NK_, EP_); } while (0)
#define SWX_DEC_NK_(MTM_, EP_) do { switch (nks) { \
        case 12: MTM_(12, EP_); break; case 16: MTM_(16, EP_); break; case 20: MTM_(20, EP_); break; \
        case 24: MTM_(24, EP_); break; case 32: MTM_(32, EP_); break; case 40: MTM_(40, EP_); break; \
        default: return -4; } } while (0)
#define SWX_DEC_NK(EP_) SWX_DEC_NK_(SWX_DEC_MT, EP_)
    switch (epi) {          // K-slice depths: d = 384 / 512 / 768 / 1024 / 1280 of the Whisper sizes and the pieces of 4d that fit
        case DEC_LN | DEC_QKV: SWX_DEC_NK(DEC_LN | DEC_QKV); break;
        case DEC_RES: SWX_DEC_NK(DEC_RES); break;
        case DEC_LN: SWX_DEC_NK(DEC_LN); break;
        case DEC_LN | DEC_GELU: SWX_DEC_NK(DEC_LN | DEC_GELU); break;
        case DEC_RES | DEC_SLAB: SWX_DEC_NK(DEC_RES | DEC_SLAB); break;
        case DEC_RES | DEC_SLAB | DEC_TICKET: SWX_DEC_NK_(SWX_DEC_MT4, DEC_RES | DEC_SLAB | DEC_TICKET); break;
        default: return -4;
    }
#undef SWX_DEC_NK
#undef SWX_DEC_NK_
#undef SWX_DEC_MT
#undef SWX_DEC_MT4
#undef SWX_DEC_W1
#undef SWX_DEC
    }
    }
    SWX_CHECK_LAUNCH();
    if (ks2 > 1 && !ticket) {
        // the one K-split projection: x += bias + sum of slabs
        if (!(g.epi & DEC_RES) || !g.X) return -4;
        SwxProfScope prof2(PC_NORM, (double)ks2 * g.M * g.N * 4 + 4.0 * g.M * g.N, s);
        const int blocks = cdiv((int64_t)g.M * (g.N / 4), 256);
        if (ks2 <= 4) hipLaunchKernelGGL(dec_slab_finish<4>, dim3(blocks), dim3(256), 0, s, g.slabs, g.slab_stride, ks2, g.c2, g.X, g.ldx, g.M, g.N);
        else if (ks2 <= 8) hipLaunchKernelGGL(dec_slab_finish<8>, dim3(blocks), dim3(256), 0, s, g.slabs, g.slab_stride, ks2, g.c2, g.X, g.ldx, g.M, g.N);
        else if (ks2 <= 16) hipLaunchKernelGGL(dec_slab_finish<16>, dim3(blocks), dim3(256), 0, s, g.slabs, g.slab_stride, ks2, g.c2, g.X, g.ldx, g.M, g.N);
        else return -4;
        SWX_CHECK_LAUNCH();
    }
    return 0;
}

int swx_fold_ln(const void *W, const float *gamma, const float *beta, const float *bias, void *Wf, float *c1, float *c2,
                int N, int K, hipStream_t s)
{
    if (N <= 0 || K <= 0) return 0;
    if (N % 16 != 0 || K % 32 != 0) return -4;
    hipLaunchKernelGGL(fold_pack_kernel, dim3(N), dim3(256), 0, s, (const f16 *)W, gamma, beta, bias, (f16 *)Wf, c1, c2, K);
    SWX_CHECK_LAUNCH();
    return 0;
}
