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
