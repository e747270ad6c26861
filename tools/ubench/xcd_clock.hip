// xcd_clock.hip -- round 6: do the XCDs of one launch START 0.2 us apart (blocks b % 8 = 0..7 stamped 0.0 0.21 0.36 0.57 1.25 1.40 0.88 1.03 us
// behind the first one in every GEMV launch: tools/r6/block_ramp.py), or are the XCDs' copies of the s_memrealtime counter that far apart?
// 256 blocks (one per CU: 100 KB of LDS each), each stamps its start, adds 1 to a device counter (agent scope), spins until all 256 have
// arrived -- every block passes within the visibility latency of one atomic (~1 us, the same for all) -- and stamps again.  If the second
// stamps show the same per-XCD offsets as the first, the CLOCKS differ; if they agree across XCDs, the starts differ.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef unsigned int u32;
typedef unsigned long long u64;

__global__ void __launch_bounds__(1024) k(u32 *counter, u64 *t_start, u64 *t_after, u32 *xcc, u32 target) {
    extern __shared__ unsigned char smem[];
    const u64 t0 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        smem[0] = 1;
        u32 id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        xcc[blockIdx.x] = id;
        t_start[blockIdx.x] = t0;
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        t_after[blockIdx.x] = __builtin_amdgcn_s_memrealtime();
    }
}

int main() {
    const int NB = 256, iters = 60;
    u32 *counter, *xcc; u64 *ts, *ta;
    CHECK(hipMalloc(&counter, 4)); CHECK(hipMalloc(&xcc, NB * 4)); CHECK(hipMalloc(&ts, NB * 8)); CHECK(hipMalloc(&ta, NB * 8));
    CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    std::vector<std::vector<double>> st(8), af(8);
    std::vector<u32> hx(NB);
    for (int it = 0; it < iters; it++) {
        CHECK(hipMemset(counter, 0, 4));
        hipLaunchKernelGGL(k, dim3(NB), dim3(1024), 100 * 1024, 0, counter, ts, ta, xcc, (u32)NB);
        CHECK(hipDeviceSynchronize());
        std::vector<u64> hs(NB), ha(NB);
        CHECK(hipMemcpy(hs.data(), ts, NB * 8, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(ha.data(), ta, NB * 8, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(hx.data(), xcc, NB * 4, hipMemcpyDeviceToHost));
        if (it < 5) continue;
        const u64 s0 = *std::min_element(hs.begin(), hs.end()), a0 = *std::min_element(ha.begin(), ha.end());
        for (int b = 0; b < NB; b++) {
            st[hx[b] & 7].push_back((double)(hs[b] - s0) / 100.0);
            af[hx[b] & 7].push_back((double)(ha[b] - a0) / 100.0);
        }
    }
    auto med = [](std::vector<double> &x) { std::sort(x.begin(), x.end()); return x.empty() ? -1.0 : x[x.size() / 2]; };
    printf("block -> XCC_ID of the first 16 blocks:");
    for (int b = 0; b < 16; b++) printf(" %u", hx[b]);
    printf("\nper XCD (HW_REG_XCC_ID), median us behind the launch's earliest stamp:\n  start stamps      :");
    for (int x = 0; x < 8; x++) printf(" %5.2f", med(st[x]));
    printf("\n  stamps behind the all-blocks-arrived barrier:");
    for (int x = 0; x < 8; x++) printf(" %5.2f", med(af[x]));
    printf("\n");
    return 0;
}
