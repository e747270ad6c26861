// lds_dma.hip -- probe of buffer_load_dwordx4 ... lds (direct-to-LDS loads) on gfx950:
//   * LDS placement (M0 base + 16 * lane), bases above 64 KiB, out-of-range lanes, visibility after vmcnt.
// Build: hipcc --offload-arch=gfx950 -O3 lds_dma.hip -o lds_dma
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
typedef uint32_t u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16(u32x4 rsrc, u32 lds_base, u32 voff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen nt lds" ::"s"(lds_base), "v"(voff), "s"(rsrc) : "memory", "m0");
}

extern "C" __global__ void __launch_bounds__(256) k(const u32 *src, u32 *dst, u32 nbytes, u32 ldsoff) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint64_t sp = (uint64_t)(uintptr_t)src;
    const u32x4 rs = {(u32)sp, (u32)(sp >> 32) & 0xFFFFu, nbytes, 0x00020000u};  // raw buffer resource: base, stride 0, num_records, flags
    const u32 w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = threadIdx.x & 63;
    u32 *all = (u32 *)(smem + ldsoff);
    for (u32 i = threadIdx.x; i < 1024; i += 256) all[i] = 0xDEADBEEFu;
    float *red = (float *)smem;
    red[threadIdx.x] = (float)threadIdx.x;
    __syncthreads();
    // lane l of wave w fetches 16 B at a permuted source position; wave 3's upper half is out of range
    const u32 srcpos = (w * 64 + (l ^ 5u)) * 16u;
    const u32 lds_base = (u32)(uintptr_t)(smem + ldsoff) + w * 1024u;
    dma16(rs, lds_base, (w == 3 && l >= 32) ? 0x80000000u : srcpos);
    float s = red[(threadIdx.x * 7) & 255];  // LDS traffic while the DMA is in flight
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (u32 i = threadIdx.x; i < 1024; i += 256) dst[i] = all[i] + (s < 0.f ? 1u : 0u);
}

int main() {
    const u32 n = 1024;
    std::vector<u32> h(n), o(n);
    for (u32 i = 0; i < n; i++) h[i] = i * 3u + 1u;
    u32 *ds, *dd;
    hipMalloc(&ds, n * 4);
    hipMalloc(&dd, n * 4);
    hipMemcpy(ds, h.data(), n * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (u32 ldsoff : {4096u, 65536u + 4096u, 150u * 1024u}) {
        hipMemset(dd, 0, n * 4);
        hipLaunchKernelGGL(k, dim3(1), dim3(256), ldsoff + 4096, 0, ds, dd, n * 4, ldsoff);
        hipError_t e = hipDeviceSynchronize();
        hipMemcpy(o.data(), dd, n * 4, hipMemcpyDeviceToHost);
        u32 bad = 0, oob_zero = 0, oob_keep = 0;
        for (u32 w = 0; w < 4; w++)
            for (u32 l = 0; l < 64; l++)
                for (u32 j = 0; j < 4; j++) {
                    const u32 got = o[(w * 64 + l) * 4 + j], want = h[(w * 64 + (l ^ 5u)) * 4 + j];
                    if (w == 3 && l >= 32) {
                        oob_zero += got == 0;
                        oob_keep += got == 0xDEADBEEFu;
                    } else
                        bad += got != want;
                }
        printf("ldsoff %6u: err=%s mismatches=%u  oob lanes: zero=%u kept=%u (of 128)\n", ldsoff, hipGetErrorString(e), bad, oob_zero, oob_keep);
    }
    return 0;
}
