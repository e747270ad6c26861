// grid_barrier.hip -- cost of a software grid barrier on MI355X (256 co-resident blocks, one per CU): flat counter vs
// per-XCD two-level counters.  Build: hipcc --offload-arch=gfx950 -O3 grid_barrier.hip -o grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef uint32_t u32;

// flat: every block adds 1 to ctr; waits until ctr >= target (monotone counter, target = (iteration + 1) * nblocks)
__global__ void __launch_bounds__(1024) flat(u32 *ctr, int iters, u32 base) {
    for (int it = 0; it < iters; it++) {
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const u32 target = base + (u32)(it + 1) * gridDim.x;
            while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
    }
}
// flat counter, relaxed atomics: one release fence before the arrive, relaxed polling, one acquire fence after
__global__ void __launch_bounds__(1024) flat_relaxed(u32 *ctr, int iters, u32 base, int fences) {
    for (int it = 0; it < iters; it++) {
        __syncthreads();
        if (threadIdx.x == 0) {
            if (fences) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const u32 target = base + (u32)(it + 1) * gridDim.x;
            while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
            if (fences) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }
}
// two-level: blocks of the same XCD (blockIdx % 8) add to their XCD counter; the last arriver of an XCD adds to the top
// counter; everyone polls the top counter
__global__ void __launch_bounds__(1024) twolevel(u32 *ctr, int iters, u32 base) {
    const u32 xcd = blockIdx.x & 7u, per = gridDim.x / 8u;
    for (int it = 0; it < iters; it++) {
        __syncthreads();
        if (threadIdx.x == 0) {
            const u32 old = __hip_atomic_fetch_add(ctr + 64u * (1u + xcd), 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if ((old + 1u) % per == 0u) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const u32 target = base + (u32)(it + 1) * 8u;
            while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
    }
}
int main() {
    u32 *ctr; CHECK(hipMalloc(&ctr, 4096 * 4)); CHECK(hipMemset(ctr, 0, 4096 * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int variant = 0; variant < 4; variant++) {
        for (int iters : {1, 101}) {
            CHECK(hipMemset(ctr, 0, 4096 * 4));
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            if (variant == 0) hipLaunchKernelGGL(flat, dim3(256), dim3(1024), 0, 0, ctr, iters, 0u);
            else if (variant == 1) hipLaunchKernelGGL(twolevel, dim3(256), dim3(1024), 0, 0, ctr, iters, 0u);
            else hipLaunchKernelGGL(flat_relaxed, dim3(256), dim3(1024), 0, 0, ctr, iters, 0u, variant == 2 ? 1 : 0);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            const char *nm[4] = {"flat acq/rel", "two-level   ", "flat relaxed + fences", "flat relaxed, no fences"};
            printf("%s: %d barriers: %.2f us total\n", nm[variant], iters, ms * 1e3);
        }
    }
    return 0;
}
