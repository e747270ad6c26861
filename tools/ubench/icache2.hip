// icache2.hip -- what does straight-line code cost when it is in L2 but not in the instruction cache?  The decode step alternates
// five kernels with ~100 KiB of code between them (the instruction cache is 64 KiB per pair of CUs), so every launch starts on code
// that the cache has dropped.  Eight copies of a 16 KiB one-shot kernel (2048 unique 8-byte VALU instructions), launched
//   same : copy 0 again and again (warm instruction cache)           rotate : copies 0..7 round-robin (128 KiB: cold in the
//   instruction cache, warm in L2 after the first round).   Cycles from the first to the last instruction of one wave of block 100.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#define R4(M) M M M M
#define A1 asm volatile("v_add_u32 %0, 0x12345, %0\n v_xor_b32 %0, 0x54321, %0\n v_add_u32 %0, 0x1111, %0\n v_xor_b32 %0, 0x77777, %0" : "+v"(v));
#define A16 R4(R4(A1))
#define A256 R4(R4(A16))
template <int ID>
__global__ void k(unsigned long long *t, uint32_t *out) {
    uint32_t v = threadIdx.x + ID;
    unsigned long long t0 = __builtin_readcyclecounter();
    A256 A256
    unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 100) t[0] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = v;
}
typedef void (*kern_t)(unsigned long long *, uint32_t *);
int main() {
    unsigned long long *t, h; uint32_t *out;
    CHECK(hipMalloc(&t, 64)); CHECK(hipMalloc(&out, 256 * 1024 * 4));
    kern_t ks[8] = {k<0>, k<1>, k<2>, k<3>, k<4>, k<5>, k<6>, k<7>};
    for (int threads : {64, 256, 1024}) {
        for (int mode = 0; mode < 2; mode++) {
            double sum = 0; int n = 0;
            for (int it = 0; it < 48; it++) {
                hipLaunchKernelGGL(ks[mode ? it % 8 : 0], dim3(256), dim3(threads), 0, 0, t, out);
                CHECK(hipDeviceSynchronize());
                CHECK(hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost));
                if (it >= 16) sum += (double)h, n++;
            }
            printf("threads/block=%4d %-6s: %.0f cycles per 2048 instructions (16 KiB) -> %.2f cycles/instr\n", threads, mode ? "rotate" : "same", sum / n, sum / n / 2048.0);
        }
    }
    return 0;
}
