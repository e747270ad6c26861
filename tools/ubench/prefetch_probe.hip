// prefetch_probe.hip -- does touching a buffer (one dword per cache line) in kernel A make kernel B's stream of it faster?
// (memory-side Infinity Cache: 256 MiB; the XCD L2s may be invalidated at kernel boundaries)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ void stream(const u32x4 *p, size_t n16, uint32_t *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x * 8 + threadIdx.x;
    u32x4 acc = {0, 0, 0, 0}, v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
        size_t j = i + (size_t)u * blockDim.x;
        v[u] = j < n16 ? __builtin_nontemporal_load(p + j) : (u32x4){0, 0, 0, 0};
    }
#pragma unroll
    for (int u = 0; u < 8; u++) acc ^= v[u];
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}

// one dword per `stride` bytes
__global__ void touch(const uint32_t *p, size_t bytes, uint32_t stride, uint32_t *out) {
    uint32_t acc = 0;
    for (size_t o = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * stride; o < bytes; o += (size_t)gridDim.x * blockDim.x * stride)
        acc ^= p[o / 4];
    if (acc == 0x12345678u) out[0] = 1;
}

float time_graph(hipStream_t s, hipGraphExec_t g) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipGraphLaunch(g, s)); CHECK(hipStreamSynchronize(s));
    float best = 1e9;
    for (int r = 0; r < 3; r++) {
        CHECK(hipEventRecord(e0, s)); CHECK(hipGraphLaunch(g, s)); CHECK(hipEventRecord(e1, s)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    return best * 1e3f;
}

int main() {
    uint32_t *out; CHECK(hipMalloc(&out, 64));
    hipStream_t s; CHECK(hipStreamCreate(&s));
    const int iters = 100;
    for (size_t mb : {4, 15, 29}) {
        size_t bytes = mb << 20, n16 = bytes / 16;
        size_t nbuf = (800u << 20) / bytes; if (nbuf > 100) nbuf = 100;
        std::vector<void *> bufs(nbuf);
        for (auto &b : bufs) { CHECK(hipMalloc(&b, bytes)); CHECK(hipMemset(b, 1, bytes)); }
        CHECK(hipDeviceSynchronize());
        int sgrid = (int)((n16 + 2047) / 2048);
        auto build = [&](int mode, uint32_t stride, int tgrid) {
            hipGraph_t g; hipGraphExec_t ge;
            CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
            for (int it = 0; it < iters; it++) {
                size_t i = it % nbuf, j = (it + nbuf / 2) % nbuf;
                if (mode == 1) hipLaunchKernelGGL(touch, dim3(tgrid), dim3(256), 0, s, (const uint32_t *)bufs[i], bytes, stride, out);  // warm
                if (mode == 2) hipLaunchKernelGGL(touch, dim3(tgrid), dim3(256), 0, s, (const uint32_t *)bufs[j], bytes, stride, out);  // same cost, other buffer
                if (mode != 3) hipLaunchKernelGGL(stream, dim3(sgrid), dim3(256), 0, s, (const u32x4 *)bufs[i], n16, out);
                if (mode == 3) hipLaunchKernelGGL(touch, dim3(tgrid), dim3(256), 0, s, (const uint32_t *)bufs[i], bytes, stride, out);  // touch only
            }
            CHECK(hipStreamEndCapture(s, &g)); CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            return ge;
        };
        float cold = time_graph(s, build(0, 128, 1)) / iters;
        printf("%2zu MiB: stream cold %.2f us (%.0f GB/s)\n", mb, cold, bytes / cold / 1e3);
        for (uint32_t stride : {64u, 128u})
            for (int tgrid : {256, 1024, 4096}) {
                float tonly = time_graph(s, build(3, stride, tgrid)) / iters;
                float warm = time_graph(s, build(1, stride, tgrid)) / iters;
                float other = time_graph(s, build(2, stride, tgrid)) / iters;
                printf("   touch stride %3u grid %4d: touch alone %.2f us | touch(X)+stream(X) %.2f | touch(Y)+stream(X) %.2f -> stream warm %.2f us\n",
                       stride, tgrid, tonly, warm, other, cold - (other - warm));
            }
        for (auto &bb : bufs) CHECK(hipFree(bb));
    }
    return 0;
}
