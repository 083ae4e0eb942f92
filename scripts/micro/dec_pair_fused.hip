// Microbenchmark: two chained decode-step GEMMs at M = 100, K = N = 1280 as ONE persistent launch with a
// run-ahead weight loader across the seam, against the same two GEMMs as two dependent launches.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/micro/dec_pair_fused.hip -o /tmp/dec_pair && /tmp/dec_pair
//
// Both forms run the arithmetic of csrc/swx_decstep.hip::gemm_dec_f16<1, 40, DEC_RES>: a workgroup = 64 output columns (16 per wave)
// x 16 rows x the whole reduction, fragment-packed weights (all 40 fragments of a wave in flight, inline-asm loads behind counted
// waits), the 16 x 1280 activation tile by LDS-DMA with the XOR swizzle, swapped MFMA operands.  y = f16(x + A W1^T + b1), then
// z = f16(x2 + y W2^T + b2): GEMM 2 needs COMPLETE rows of y, i.e. the 20 column panels of its row group -- a 20 -> 20 exchange
// per row group (7 row groups), the all-to-all edge of the decode chain at its smallest.
//
// Fused form (one launch, 140 resident workgroups = (panel, row group) units, one per CU):
//   phase 1  as above; y stored WRITE-THROUGH (`sc1`), then the 40 weight fragments of GEMM 2 are requested (the run-ahead
//            loader: they fly across the seam), `s_waitcnt vmcnt(40)` = the stores are in memory (vector memory operations retire
//            in issue order on this hardware: what the tall dec GEMM's counted waits rely on), barrier, ONE lane publishes
//            flag[row group][panel] = epoch (relaxed, agent scope).   [MI355X_MICROARCH.md recipe R1 / rows publish-large, handoff-flag]
//   seam     wave 0 polls the row group's 20 flags (one load instruction per pass, relaxed agent-scope, s_sleep between passes,
//            BOUNDED: 2^18 passes, then an error word is set and the kernel carries on with whatever is there);
//            barrier; the y tile comes in by LDS-DMA with `sc1` (coherent reads of what other XCDs wrote through: no invalidate).
//   phase 2  vmcnt(0) (tile + weights landed), barrier, 40 MFMAs, epilogue.
// Flags: one array per launch of the timed chain (all zeroed before it), epoch = 1.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef _Float16 f16;
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

constexpr int NKS = 40, KS = NKS * 32, SPR = KS / 8, RS = KS * 2;      // K = 1280
constexpr int N_RG = 7, PANELS = 20, M_ROWS = 100, N_COLS = 1280;

struct Gemm { const f16 *A; const f16 *W; const float *bias; f16 *X; };   // X: residual stream, updated in place (ld = N_COLS)

__device__ __forceinline__ void unit_of_block(int b, int &panel, int &rg, bool &live)
{
    const int xcd = b & 7, slot = b >> 3;                 // the row groups of a panel on one XCD (block id -> XCD is id % 8)
    const int unit = (slot / N_RG) * 8 + xcd;
    rg = slot % N_RG; panel = unit; live = unit < PANELS;
}

template <int AUX>
__device__ __forceinline__ void dma_tile(const f16 *A, int r0, unsigned char *smem, int wave, int lane)
{
#pragma unroll
    for (int j = 0; j < NKS / 4; ++j) {
        const int q = j * 4 + wave, p = q * 64 + lane;
        const int row = p / SPR, ps = p - row * SPR, kslot = ps ^ (row & 15);
        const int gr = r0 + row < M_ROWS ? r0 + row : M_ROWS - 1;
        __builtin_amdgcn_global_load_lds(A + (size_t)gr * KS + kslot * 8, (lds_void *)(smem + q * 1024), 16, 0, AUX);
    }
}

__device__ __forceinline__ void load_w(const f16 *W, int panel, int wave, int lane, f16x8 (&wf)[NKS])
{
    const f16 *wp = W + ((size_t)(panel * 4 + wave) * NKS) * 512 + lane * 8;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
        asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(wf[ks]) : "v"(wp + (ks >> 2) * 2048), "n"((ks & 3) * 1024) : "memory");
}

// MFMA loop over a landed tile; YOUNGER = loads issued after the weight fragments that may still be in flight
template <int YOUNGER, bool WAIT>
__device__ __forceinline__ f32x4 mfma_loop(const unsigned char *smem, int li, int lg, f16x8 (&wf)[NKS])
{
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const unsigned char *abase = smem + (size_t)li * RS;
    constexpr int PF = 8;
    f16x8 af[PF];
    auto frag = [&](int ks) { return *(const f16x8 *)(abase + ((((ks * 4 + lg) ^ li)) << 4)); };
#pragma unroll
    for (int ks = 0; ks < PF; ++ks) af[ks] = frag(ks);
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        if constexpr (WAIT) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(wf[ks]) : "n"((NKS - 1 - ks) + YOUNGER) : "memory");
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ks], af[ks % PF], acc, 0, 0, 0);
        if (ks + PF < NKS) af[ks % PF] = frag(ks + PF);
        __builtin_amdgcn_sched_barrier(0);
    }
    return acc;
}

// ---- the two-launch form: one GEMM per launch
__global__ __launch_bounds__(256) void k_single(Gemm g)
{
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
    int panel, rg; bool live;
    unit_of_block(blockIdx.x, panel, rg, live);
    if (!live) return;
    const int r0 = rg * 16;
    dma_tile<0>(g.A, r0, smem, wave, lane);
    f16x8 wf[NKS];
    load_w(g.W, panel, wave, lane, wf);
    const int n = panel * 64 + wave * 16 + lg * 4;
    const int m = r0 + li, mc = m < M_ROWS ? m : M_ROWS - 1;
    f32x4 c2; f16x4 xr;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(c2) : "v"(g.bias + n) : "memory");
    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(xr) : "v"(g.X + (size_t)mc * N_COLS + n) : "memory");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NKS + 2) : "memory");
    __builtin_amdgcn_s_barrier();
    f32x4 acc = mfma_loop<2, true>(smem, li, lg, wf);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(c2), "+v"(xr) : : "memory");
    if (m < M_ROWS) {
        f16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (f16)(acc[e] + c2[e] + (float)xr[e]);
        *(f16x4 *)(g.X + (size_t)m * N_COLS + n) = o;
    }
}

// ---- the fused form: GEMM 1 -> seam -> GEMM 2 in one launch.  g1.X = y (in / out), g2.A = y, g2.X = z (in / out)
__global__ __launch_bounds__(256) void k_fused(Gemm g1, Gemm g2, unsigned *flags, unsigned *err)
{
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
    int panel, rg; bool live;
    unit_of_block(blockIdx.x, panel, rg, live);
    if (!live) return;
    const int r0 = rg * 16;
    const int n = panel * 64 + wave * 16 + lg * 4;
    const int m = r0 + li, mc = m < M_ROWS ? m : M_ROWS - 1;
    // ---------------- phase 1
    dma_tile<0>(g1.A, r0, smem, wave, lane);
    f16x8 wf[NKS];
    load_w(g1.W, panel, wave, lane, wf);
    f32x4 c2; f16x4 xr;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(c2) : "v"(g1.bias + n) : "memory");
    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(xr) : "v"(g1.X + (size_t)mc * N_COLS + n) : "memory");
    // phase 2's epilogue operands do not depend on phase 1 either (z's residual input, bias 2): requested now, consumed at the end
    f32x4 c2b; f16x4 xrb;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(c2b) : "v"(g2.bias + n) : "memory");
    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(xrb) : "v"(g2.X + (size_t)mc * N_COLS + n) : "memory");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NKS + 4) : "memory");
    __builtin_amdgcn_s_barrier();
    f32x4 acc = mfma_loop<4, true>(smem, li, lg, wf);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(c2), "+v"(xr), "+v"(c2b), "+v"(xrb) : : "memory");
    {
        f16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (f16)(acc[e] + c2[e] + (float)xr[e]);
        const f16 *dst = g1.X + (size_t)mc * N_COLS + n;                       // rows past M: the last row's own value again (never predicated)
        asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(dst), "v"(o) : "memory");
    }
    // ---------------- run-ahead loader: GEMM 2's weights cross the seam in flight
    // (waves 1-3; wave 0 polls the flags below and a poll's result returns behind every older load of the wave, so it requests
    //  its fragments once the seam is crossed -- they then travel with the y tile)
    f16x8 wf2[NKS];
    if (wave != 0) {
        load_w(g2.W, panel, wave, lane, wf2);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NKS) : "memory");       // everything older than the 40 weight loads: y is in memory
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                                          // every wave's stores are in memory; the activation tile is dead
                                                                           // (raw barrier: __syncthreads() would drain the weight loads)
    if (tid == 0) __hip_atomic_store(flags + rg * 32 + panel, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // ---------------- seam: the 20 panels of this row group
    if (wave == 0) {
        const unsigned *fp = flags + rg * 32 + (lane < PANELS ? lane : 0);
        int spins = 0;
        for (;;) {
            const unsigned v = __hip_atomic_load(fp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__all(v == 1u)) break;
            if (++spins > (1 << 18)) { if (lane == 0) atomicAdd(err, 1u); break; }     // never hang the box
            __builtin_amdgcn_s_sleep(1);
        }
    }
    if (wave == 0) load_w(g2.W, panel, wave, lane, wf2);
    __builtin_amdgcn_s_barrier();
    // ---------------- phase 2
    dma_tile<16>(g2.A, r0, smem, wave, lane);                              // aux 16 = sc1: coherent reads, no invalidate
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) asm volatile("" : "+v"(wf2[ks]));
    __builtin_amdgcn_s_barrier();
    f32x4 acc2 = mfma_loop<0, false>(smem, li, lg, wf2);
    if (m < M_ROWS) {
        f16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (f16)(acc2[e] + c2b[e] + (float)xrb[e]);
        *(f16x4 *)(g2.X + (size_t)m * N_COLS + n) = o;
    }
}

static void pack_weights(const std::vector<f16> &W, std::vector<f16> &P)     // [N][K] -> [N/16][K/32][64 lanes][8]
{
    P.resize(W.size());
    for (int n = 0; n < N_COLS; ++n)
        for (int k = 0; k < KS; ++k) {
            const size_t grp = (size_t)(n >> 4) * (KS >> 5);
            const int ks = k >> 5, lane = ((k & 31) >> 3) * 16 + (n & 15);
            P[((grp + ks) * 64 + lane) * 8 + (k & 7)] = W[(size_t)n * KS + k];
        }
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 64;          // pairs per timed chain
    const int n_w = 48;                                        // weight sets cycled through: 2 x 48 x 3.3 MB = 315 MB > the 256 MB Infinity Cache
    srand(7);
    auto rnd = [](float s) { return (f16)(((rand() % 2001) - 1000) * 0.001f * s); };
    std::vector<f16> hA((size_t)M_ROWS * KS), hX((size_t)M_ROWS * N_COLS), hW((size_t)N_COLS * KS), hP;
    for (auto &v : hA) v = rnd(1.f);
    for (auto &v : hX) v = rnd(1.f);
    std::vector<float> hb(N_COLS);
    for (auto &v : hb) v = (float)rnd(0.5f);
    f16 *dA, *dX0, *dY, *dZ, *dW; float *db; unsigned *dflags, *derr;
    CK(hipMalloc(&dA, hA.size() * 2)); CK(hipMalloc(&dX0, hX.size() * 2)); CK(hipMalloc(&dY, hX.size() * 2)); CK(hipMalloc(&dZ, hX.size() * 2));
    CK(hipMalloc(&dW, (size_t)2 * n_w * hW.size() * 2)); CK(hipMalloc(&db, N_COLS * 4));
    CK(hipMalloc(&dflags, (size_t)iters * 8 * 32 * 4)); CK(hipMalloc(&derr, 4));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dX0, hX.data(), hX.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), N_COLS * 4, hipMemcpyHostToDevice));
    for (int i = 0; i < 2 * n_w; ++i) {
        for (auto &v : hW) v = rnd(0.03f);
        pack_weights(hW, hP);
        CK(hipMemcpy(dW + (size_t)i * hW.size(), hP.data(), hP.size() * 2, hipMemcpyHostToDevice));
    }
    const size_t lds = 16 * RS;
    CK(hipFuncSetAttribute((const void *)k_single, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    CK(hipFuncSetAttribute((const void *)k_fused, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    const int grid = ((PANELS + 7) / 8) * N_RG * 8;          // 168 blocks, 140 live
    hipStream_t s; CK(hipStreamCreate(&s));
    auto gemm1 = [&](int i) { return Gemm{dA, dW + (size_t)(2 * (i % n_w)) * hW.size(), db, dY}; };
    auto gemm2 = [&](int i) { return Gemm{dY, dW + (size_t)(2 * (i % n_w) + 1) * hW.size(), db, dZ}; };
    auto reset = [&]() {
        CK(hipMemcpyAsync(dY, dX0, hX.size() * 2, hipMemcpyDeviceToDevice, s)); CK(hipMemcpyAsync(dZ, dX0, hX.size() * 2, hipMemcpyDeviceToDevice, s));
        CK(hipMemsetAsync(dflags, 0, (size_t)iters * 8 * 32 * 4, s)); CK(hipMemsetAsync(derr, 0, 4, s));
    };
    // ---- correctness: one pair both ways, bit for bit
    std::vector<f16> z_pair(hX.size()), z_fused(hX.size()), y_pair(hX.size()), y_fused(hX.size());
    reset();
    hipLaunchKernelGGL(k_single, dim3(grid), dim3(256), lds, s, gemm1(0));
    hipLaunchKernelGGL(k_single, dim3(grid), dim3(256), lds, s, gemm2(0));
    CK(hipStreamSynchronize(s));
    CK(hipMemcpy(z_pair.data(), dZ, hX.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(y_pair.data(), dY, hX.size() * 2, hipMemcpyDeviceToHost));
    reset();
    hipLaunchKernelGGL(k_fused, dim3(grid), dim3(256), lds, s, gemm1(0), gemm2(0), dflags, derr);
    CK(hipStreamSynchronize(s));
    CK(hipMemcpy(z_fused.data(), dZ, hX.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(y_fused.data(), dY, hX.size() * 2, hipMemcpyDeviceToHost));
    unsigned herr = 0; CK(hipMemcpy(&herr, derr, 4, hipMemcpyDeviceToHost));
    const bool same = !memcmp(z_pair.data(), z_fused.data(), hX.size() * 2) && !memcmp(y_pair.data(), y_fused.data(), hX.size() * 2);
    double cs = 0; for (auto v : z_pair) cs += (double)(float)v;
    printf("fused vs two launches: y and z bit-identical = %s, spin timeouts = %u, checksum(z) = %.4f\n", same ? "yes" : "NO", herr, cs);
    // ---- timing: a captured chain of `iters` pairs, each form; every pair reads its own weight sets (HBM-cold) and its own flag array
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time_graph = [&](bool fused) {
        hipGraph_t gr; hipGraphExec_t ex;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int i = 0; i < iters; ++i) {
            if (fused) hipLaunchKernelGGL(k_fused, dim3(grid), dim3(256), lds, s, gemm1(i), gemm2(i), dflags + (size_t)i * 8 * 32, derr);
            else { hipLaunchKernelGGL(k_single, dim3(grid), dim3(256), lds, s, gemm1(i)); hipLaunchKernelGGL(k_single, dim3(grid), dim3(256), lds, s, gemm2(i)); }
        }
        CK(hipStreamEndCapture(s, &gr));
        CK(hipGraphInstantiate(&ex, gr, nullptr, nullptr, 0));
        float best = 1e30f, sum = 0.f; const int reps = 7;
        for (int r = 0; r < reps + 2; ++r) {
            reset(); CK(hipStreamSynchronize(s));
            CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ex, s)); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 2) { best = ms < best ? ms : best; sum += ms; }
        }
        CK(hipGraphExecDestroy(ex)); CK(hipGraphDestroy(gr));
        unsigned he = 0; CK(hipMemcpy(&he, derr, 4, hipMemcpyDeviceToHost));
        printf("  %-28s %8.2f us per pair (best of %d chains of %d pairs), mean %8.2f us, spin timeouts %u\n",
               fused ? "one persistent launch" : "two dependent launches", best * 1000.f / iters, reps, iters, sum / reps * 1000.f / iters, he);
        return best * 1000.f / iters;
    };
    for (int round = 0; round < 3; ++round) {
        const float a = time_graph(false), b = time_graph(true);
        printf("  -> fused / two launches = %.3f\n", b / a);
    }
    return 0;
}
