// Microbenchmark: cost of a device-wide barrier between phases of ONE persistent kernel on gfx950 (256 CUs, 8 XCDs),
// against the cost of a dependent kernel launch.  Used to decide whether a per-layer persistent decode-step kernel can pay.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/grid_barrier_bench.hip -o /tmp/gbar && /tmp/gbar
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ bool grid_barrier(unsigned *cnt, unsigned target, int sleep)
{
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __atomic_fetch_add(cnt, 1u, __ATOMIC_RELEASE);                     // agent scope by default for global memory
        int spins = 0;
        while (__atomic_load_n(cnt, __ATOMIC_ACQUIRE) < target) {
            if (sleep) __builtin_amdgcn_s_sleep(1);
            if (++spins > (1 << 22)) { ok = false; break; }                 // never hang the box
        }
    }
    __syncthreads();
    return ok;
}

// XCD-hierarchical barrier (MI355X_MICROARCH.md price list, row "barrier-xcd": 4.1-4.8 us at 256 workgroups): a counter per XCC
// (the physical one, read from HW_REG_XCC_ID -- the population of each XCC is counted once by a census phase, so nothing is
// assumed about the placement), the last arriver of an XCC is its leader: it writes the XCC's L2 back (one release fence
// covers every member's stores, which had reached that L2 before the member arrived), arrives on the top counter, polls it
// with RELAXED loads, acquires once and publishes the XCC's generation; the members poll the generation with relaxed loads
// and acquire once.  (The flat barrier above polls with acquire loads -- every iteration invalidates the CU's L1 -- which is
// what the guide prices at 13 us.)
struct XBar { unsigned cnt[8][32]; unsigned gen[8][32]; unsigned top[32]; unsigned pop[8][32]; unsigned census[32]; };

__device__ __forceinline__ unsigned xcc_id()
{
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 7u;
}

__device__ __forceinline__ bool spin_ge(unsigned *p, unsigned target)
{
    int spins = 0;
    while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1 << 22)) return false;
    }
    return true;
}

__device__ __forceinline__ bool xcd_barrier(XBar *xb, unsigned xcc, unsigned phase1, unsigned n_xcc)
{
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // this workgroup's stores have reached its L2
        const unsigned pop = xb->pop[xcc][0];
        const unsigned old = __hip_atomic_fetch_add(&xb->cnt[xcc][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 == pop * phase1) {                                      // last of this XCC: its leader
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(&xb->top[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = spin_ge(&xb->top[0], n_xcc * phase1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __hip_atomic_store(&xb->gen[xcc][0], phase1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            ok = spin_ge(&xb->gen[xcc][0], phase1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
    }
    __syncthreads();
    return ok;
}

__global__ __launch_bounds__(256) void phases_xcd_kernel(XBar *xb, int *buf, int n_phase, int *err, int payload)
{
    const int b = blockIdx.x, nb = gridDim.x;
    __shared__ unsigned s_xcc, s_nx;
    if (threadIdx.x == 0) {
        // census (once): population of every physical XCC, then a flat counter barrier so that everybody sees the final numbers
        const unsigned x = xcc_id();
        s_xcc = x;
        __hip_atomic_fetch_add(&xb->pop[x][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&xb->census[0], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        if (!spin_ge(&xb->census[0], (unsigned)nb)) atomicAdd(err, 1 << 20);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        unsigned nx = 0;
        for (int k = 0; k < 8; ++k) nx += __hip_atomic_load(&xb->pop[k][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 1u : 0u;
        s_nx = nx;
    }
    __syncthreads();
    const unsigned xcc = s_xcc, nx = s_nx;
    for (int p = 0; p < n_phase; ++p) {
        int *cur = buf + (size_t)(p & 1) * nb * payload;
        for (int i = threadIdx.x; i < payload; i += 256) cur[(size_t)b * payload + i] = p * 1000 + b;
        if (!xcd_barrier(xb, xcc, (unsigned)(p + 1), nx)) { if (threadIdx.x == 0) atomicAdd(err, 1 << 16); return; }
        const int src = (b + 37) % nb;                                     // a block on another XCD
        for (int i = threadIdx.x; i < payload; i += 256)
            if (__builtin_nontemporal_load(cur + (size_t)src * payload + i) != p * 1000 + src) atomicAdd(err, 1);
    }
}

__global__ __launch_bounds__(256) void phases_kernel(unsigned *cnt, int *buf, int n_phase, int sleep, int *err, int payload)
{
    const int b = blockIdx.x, nb = gridDim.x;
    for (int p = 0; p < n_phase; ++p) {
        int *cur = buf + (size_t)(p & 1) * nb * payload;
        for (int i = threadIdx.x; i < payload; i += 256) cur[(size_t)b * payload + i] = p * 1000 + b;
        if (!grid_barrier(cnt, (unsigned)(p + 1) * nb, sleep)) { if (threadIdx.x == 0) atomicAdd(err, 1 << 16); return; }
        const int src = (b + 37) % nb;                                     // a block on another XCD
        for (int i = threadIdx.x; i < payload; i += 256)
            if (__builtin_nontemporal_load(cur + (size_t)src * payload + i) != p * 1000 + src) atomicAdd(err, 1);
    }
}

__global__ __launch_bounds__(256) void one_phase_kernel(int *buf, int p, int *err, int payload)
{
    const int b = blockIdx.x, nb = gridDim.x;
    int *cur = buf + (size_t)(p & 1) * nb * payload, *prev = buf + (size_t)((p + 1) & 1) * nb * payload;
    const int src = (b + 37) % nb;
    if (p > 0)
        for (int i = threadIdx.x; i < payload; i += 256)
            if (prev[(size_t)src * payload + i] != (p - 1) * 1000 + src) atomicAdd(err, 1);
    for (int i = threadIdx.x; i < payload; i += 256) cur[(size_t)b * payload + i] = p * 1000 + b;
}

int main()
{
    const int n_phase = 2000;
    unsigned *cnt; int *buf, *err;
    CK(hipMalloc(&cnt, 4)); CK(hipMalloc(&err, 4));
    CK(hipMalloc(&buf, 2 * 1024 * 4096 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int payload : {64, 1280}) for (int nb : {256, 512}) for (int sleep : {0, 1}) {
        float best = 1e9f; int herr = 0;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemset(cnt, 0, 4)); CK(hipMemset(err, 0, 4));
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(phases_kernel, dim3(nb), dim3(256), 0, 0, cnt, buf, n_phase, sleep, err, payload);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
        }
        printf("persistent: payload %5d ints/block  blocks %3d  sleep %d : %.3f us per phase (write+barrier+check)  errors %d\n",
               payload, nb, sleep, best * 1000.f / n_phase, herr);
    }
    XBar *xb; CK(hipMalloc(&xb, sizeof(XBar)));
    for (int payload : {64, 1280}) for (int nb : {256}) {
        float best = 1e9f; int herr = 0;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemset(xb, 0, sizeof(XBar))); CK(hipMemset(err, 0, 4));
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(phases_xcd_kernel, dim3(nb), dim3(256), 0, 0, xb, buf, n_phase, err, payload);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
        }
        unsigned hpop[8][32]; CK(hipMemcpy(hpop, xb->pop, sizeof(hpop), hipMemcpyDeviceToHost));
        printf("persistent, XCD-hierarchical barrier: payload %5d ints/block  blocks %3d : %.3f us per phase (write+barrier+check)  errors %d  "
               "(workgroups per XCC: %u %u %u %u %u %u %u %u)\n", payload, nb, best * 1000.f / n_phase, herr, hpop[0][0], hpop[1][0],
               hpop[2][0], hpop[3][0], hpop[4][0], hpop[5][0], hpop[6][0], hpop[7][0]);
    }
    for (int payload : {64, 1280}) for (int nb : {256, 512}) {
        float best = 1e9f; int herr = 0;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemset(err, 0, 4));
            CK(hipEventRecord(e0, 0));
            for (int p = 0; p < n_phase; ++p) hipLaunchKernelGGL(one_phase_kernel, dim3(nb), dim3(256), 0, 0, buf, p, err, payload);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
        }
        printf("launches  : payload %5d ints/block  blocks %3d          : %.3f us per phase (one kernel per phase)      errors %d\n",
               payload, nb, best * 1000.f / n_phase, herr);
    }
    return 0;
}
