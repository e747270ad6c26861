// icache.hip -- is straight-line one-shot code fetch-bound at kernel start?  2048 unique VALU instructions executed once.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#define R4(M) M M M M
#define A1 asm volatile("v_add_u32 %0, 0x12345, %0\n v_xor_b32 %0, 0x54321, %0\n v_add_u32 %0, 0x1111, %0\n v_xor_b32 %0, 0x77777, %0" : "+v"(v));
#define A16 R4(R4(A1))
#define A256 R4(R4(A16))
__global__ void k(unsigned long long *t, uint32_t *out, int loops) {
    uint32_t v = threadIdx.x;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < loops; i++) {
        A256 A256  // 2048 instructions, 8-byte encodings (literal) = 16 KiB of code
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 100) { t[0] = t1 - t0; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = v;
}
int main() {
    unsigned long long *t, h; uint32_t *out;
    CHECK(hipMalloc(&t, 64)); CHECK(hipMalloc(&out, 256 * 512 * 4));
    for (int wps : {1, 2}) for (int loops : {1, 2, 4}) {
        for (int rep = 0; rep < 3; rep++) {
            hipLaunchKernelGGL(k, dim3(256), dim3(256 * wps), 0, 0, t, out, loops);
            CHECK(hipDeviceSynchronize());
            CHECK(hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost));
            printf("waves/SIMD=%d loops=%d rep=%d: %llu cycles -> %.2f cycles/instr\n", wps, loops, rep, h, (double)h / (2048.0 * loops));
        }
    }
    return 0;
}
