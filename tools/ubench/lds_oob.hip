// lds_oob.hip -- what does a ds_read_b128 beyond the workgroup's LDS allocation return?  (probe for zero-filled MFMA columns
// without a zero area: lanes whose B operand must be 0 could point outside the allocation)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__global__ void k(uint32_t *out, uint32_t base) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    for (uint32_t i = threadIdx.x; i < 4096; i += blockDim.x) reinterpret_cast<uint32_t *>(smem)[i] = 0xDEADBEEFu;
    __syncthreads();
    typedef __attribute__((address_space(3))) uint4 lds4;
    const uint32_t addr = base + threadIdx.x * 16u;
    uint4 v;
    asm volatile("ds_read_b128 %0, %1 offset:576\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[threadIdx.x * 4 + 0] = v.x; out[threadIdx.x * 4 + 1] = v.y; out[threadIdx.x * 4 + 2] = v.z; out[threadIdx.x * 4 + 3] = v.w;
}
int main() {
    uint32_t *d; CHECK(hipMalloc(&d, 64 * 16));
    uint32_t h[256];
    for (uint32_t base : {0u, 16384u - 576u, 16384u, 65536u, 0x28000u, 0x30000u, 0x7FFF0000u}) {
        CHECK(hipMemset(d, 0x55, 64 * 16));
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 16384, 0, d, base);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
        int nz = 0; for (int i = 0; i < 256; i++) nz += h[i] != 0;
        printf("base 0x%08x (+576): first %08x last %08x nonzero dwords %d / 256\n", base, h[0], h[255], nz);
    }
    return 0;
}
