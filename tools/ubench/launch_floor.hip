// launch_floor.hip -- time per (empty) kernel inside one captured graph of back-to-back dependent launches, by grid / block shape,
// dynamic LDS size and register footprint: what a launch of the decode step costs before its first instruction does anything.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__global__ void empty_k(int *p) { if (p && threadIdx.x == 9999) p[0] = 1; }
__global__ void __launch_bounds__(1024) fat_k(int *p) {  // ~120 VGPRs live
    int v[100];
    for (int i = 0; i < 100; i++) v[i] = threadIdx.x * i;
    asm volatile("" :: "v"(v[0]), "v"(v[99]), "v"(v[50]));
    int s = 0;
    if (p && threadIdx.x == 9999) { for (int i = 0; i < 100; i++) s += v[i]; p[0] = s; }
}
template <typename K>
double run(K k, int grid, int block, size_t lds, int n = 400) {
    hipStream_t s; CHECK(hipStreamCreate(&s));
    hipGraph_t g; hipGraphExec_t ge;
    CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < n; i++) hipLaunchKernelGGL(k, dim3(grid), dim3(block), lds, s, (int *)nullptr);
    CHECK(hipStreamEndCapture(s, &g)); CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CHECK(hipGraphLaunch(ge, s)); CHECK(hipStreamSynchronize(s));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e9;
    for (int r = 0; r < 5; r++) {
        CHECK(hipEventRecord(e0, s)); CHECK(hipGraphLaunch(ge, s)); CHECK(hipEventRecord(e1, s)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    return best * 1e3 / n;
}
int main() {
    CHECK(hipFuncSetAttribute((const void *)empty_k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CHECK(hipFuncSetAttribute((const void *)fat_k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int grid : {32, 192, 256, 512})
        for (int block : {64, 256, 512, 1024})
            printf("empty  grid %4d x %4d threads, no LDS : %.2f us\n", grid, block, run(empty_k, grid, block, 0));
    for (size_t lds : {(size_t)0, (size_t)65536, (size_t)150 * 1024})
        for (int block : {512, 1024})
            printf("empty  grid  256 x %4d threads, %3zu KiB LDS: %.2f us     fat (120 VGPRs): %.2f us\n", block, lds / 1024, run(empty_k, 256, block, lds),
                   run(fat_k, 256, block, lds));
    return 0;
}
