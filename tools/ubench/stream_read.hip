// stream_read.hip -- HBM read bandwidth of small one-shot kernels (the regime of a decode GEMV): each launch reads
// one `bytes`-sized buffer once; buffers rotate over a > 256 MiB working set so the Infinity Cache cannot serve them.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <bool NT, int UNROLL>
__global__ void rd(const u32x4 *p, size_t n16, uint32_t *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x * UNROLL + threadIdx.x;
    u32x4 acc = {0, 0, 0, 0};
    u32x4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
        size_t j = i + (size_t)u * blockDim.x;
        if (j < n16) v[u] = NT ? __builtin_nontemporal_load(p + j) : p[j]; else v[u] = (u32x4){0,0,0,0};
    }
#pragma unroll
    for (int u = 0; u < UNROLL; u++) acc ^= v[u];
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}

template <bool NT, int UNROLL>
float run(const std::vector<void *> &bufs, size_t bytes, uint32_t *out, int iters, int block) {
    size_t n16 = bytes / 16;
    int grid = (int)((n16 + (size_t)block * UNROLL - 1) / ((size_t)block * UNROLL));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (size_t i = 0; i < bufs.size(); i++) hipLaunchKernelGGL((rd<NT, UNROLL>), dim3(grid), dim3(block), 0, 0, (const u32x4 *)bufs[i], n16, out);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int it = 0; it < iters; it++) hipLaunchKernelGGL((rd<NT, UNROLL>), dim3(grid), dim3(block), 0, 0, (const u32x4 *)bufs[it % bufs.size()], n16, out);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / iters;
}

int main() {
    uint32_t *out; CHECK(hipMalloc(&out, 64));
    for (size_t mb : {4, 6, 15, 29, 59, 256}) {
        size_t bytes = mb << 20;
        size_t nbuf = (600u << 20) / bytes + 1; if (nbuf > 64) nbuf = 64; if (nbuf < 3) nbuf = 3;
        std::vector<void *> bufs(nbuf);
        for (auto &b : bufs) { CHECK(hipMalloc(&b, bytes)); CHECK(hipMemset(b, 1, bytes)); }
        CHECK(hipDeviceSynchronize());
        int iters = 200;
        float a = run<false, 4>(bufs, bytes, out, iters, 256);
        float b = run<true, 4>(bufs, bytes, out, iters, 256);
        float c = run<true, 8>(bufs, bytes, out, iters, 256);
        float d = run<true, 2>(bufs, bytes, out, iters, 512);
        float e = run<true, 16>(bufs, bytes, out, iters, 256);
        printf("%4zu MiB x%zu bufs: plain u4 %.2f us (%.0f GB/s) | nt u4 %.2f (%.0f) | nt u8 %.2f (%.0f) | nt u2 b512 %.2f (%.0f) | nt u16 %.2f (%.0f)\n",
               mb, nbuf, a, bytes / a / 1e3, b, bytes / b / 1e3, c, bytes / c / 1e3, d, bytes / d / 1e3, e, bytes / e / 1e3);
        for (auto &bb : bufs) CHECK(hipFree(bb));
    }
    return 0;
}
