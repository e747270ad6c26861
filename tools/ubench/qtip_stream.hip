// qtip_stream.hip -- the QTIP band engine's memory stream without its arithmetic: one 1024-thread block per CU (128 KiB of
// static LDS), wave w of a block takes tile blocks w, w + 16, ... of a 32-row band (256 bytes each at R = 2), block b takes
// bands b, b + grid, ...; a register queue of PF tile blocks per wave.  Variants: duplicate lanes (the two 16-row halves
// request the same 8 bytes) on / off, queue depth, rotated start per band, 16 bytes per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef uint32_t u32;

template <int PF, int DUP, int ROT, int BIGLDS, int NAT>
__global__ void __launch_bounds__(1024) stream(const u32 *comp, u32 bands, u32 nK2, u32 *out) {
    __shared__ u32 tab[BIGLDS ? 32768 : 64];
    const u32 tid = threadIdx.x, l = tid & 63u, W = 16;
    const u32 w = (u32)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    if (tid < 64) tab[tid] = tid;
    const u32 g = l >> 4, a4 = (l >> 3) & 1u, a = l & 7u, s = 4u * a + g;
    const u32 voff = NAT ? (l & 31u) * 8u : ((DUP || a4 == 0) ? s * 8u : 0xFFFFFFFFu);  // NAT: lane l loads unit l % 32 (contiguous)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)comp, 0, (int)(bands * nK2 * 256u), 0x00020000);
    u32 acc = 0;
    u32 dq[PF][2];
    u32 band = blockIdx.x;
    auto k2of = [&](u32 bnd, u32 i) {  // i-th tile block of this wave in band bnd
        u32 k = w + i * W;
        if (ROT) { k += (bnd * 37u) % nK2; if (k >= nK2) k -= nK2; }
        return k;
    };
    const u32 per = nK2 / W;  // tile blocks per wave and band (nK2 % 16 == 0 here)
    auto fetch = [&](u32 p, u32 bnd, u32 i) {
        const u32 bb = bnd < bands ? bnd : bands - 1u;
        auto v = __builtin_amdgcn_raw_buffer_load_b64(rs, bnd < bands ? voff : 0xFFFFFFFFu, bb * nK2 * 256u + k2of(bb, i) * 256u, 2);
        dq[p][0] = v[0]; dq[p][1] = v[1];
    };
    // flattened sequence of (band, i): position q -> band = blockIdx + (q / per) * grid, i = q % per
    u32 total = 0;
    for (u32 b = blockIdx.x; b < bands; b += gridDim.x) total += per;
#pragma unroll
    for (u32 p = 0; p < PF; p++) { const u32 q = p; fetch(p, blockIdx.x + (q / per) * gridDim.x, q % per); }
    __syncthreads();
    for (u32 q0 = 0; q0 < total; q0 += PF) {
#pragma unroll
        for (u32 p = 0; p < PF; p++) {
            acc ^= dq[p][0] + dq[p][1];
            const u32 q = q0 + p + PF;
            fetch(p, blockIdx.x + (q / per) * gridDim.x, q % per);
        }
    }
    if (acc == 0x12345678u) out[0] = tab[l];
}

template <int PF, int DUP, int ROT, int BIGLDS, int NAT>
float run(const std::vector<void *> &bufs, u32 bands, u32 nK2, u32 *out, int iters) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    int grid = bands < 256 ? bands : 256;
    for (size_t i = 0; i < bufs.size(); i++) hipLaunchKernelGGL((stream<PF, DUP, ROT, BIGLDS, NAT>), dim3(grid), dim3(1024), 0, 0, (const u32 *)bufs[i], bands, nK2, out);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int it = 0; it < iters; it++) hipLaunchKernelGGL((stream<PF, DUP, ROT, BIGLDS, NAT>), dim3(grid), dim3(1024), 0, 0, (const u32 *)bufs[it % bufs.size()], bands, nK2, out);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / iters;
}

// 16 bytes per lane: one wave-instruction = 1 KiB = 4 consecutive tile blocks; wave w takes chunks w, w + 16, ... of a band
template <int PF>
__global__ void __launch_bounds__(1024) stream4(const u32 *comp, u32 bands, u32 nK2, u32 *out) {
    __shared__ u32 tab[32768];
    const u32 tid = threadIdx.x, l = tid & 63u, W = 16;
    const u32 w = (u32)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    if (tid < 64) tab[tid] = tid;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)comp, 0, (int)(bands * nK2 * 256u), 0x00020000);
    typedef u32 u32x4 __attribute__((ext_vector_type(4)));
    u32 acc = 0;
    u32x4 dq[PF];
    const u32 per = nK2 / 4u / W;  // 1 KiB chunks per wave and band
    auto fetch = [&](u32 p, u32 bnd, u32 i) {
        const u32 bb = bnd < bands ? bnd : bands - 1u;
        dq[p] = __builtin_amdgcn_raw_buffer_load_b128(rs, bnd < bands ? l * 16u : 0xFFFFFFFFu, bb * nK2 * 256u + (w + i * W) * 1024u, 2);
    };
    u32 total = 0;
    for (u32 b = blockIdx.x; b < bands; b += gridDim.x) total += per;
#pragma unroll
    for (u32 p = 0; p < PF; p++) { const u32 q = p; fetch(p, blockIdx.x + (q / per) * gridDim.x, q % per); }
    __syncthreads();
    for (u32 q0 = 0; q0 < total; q0 += PF) {
#pragma unroll
        for (u32 p = 0; p < PF; p++) {
            acc ^= dq[p][0] + dq[p][1] + dq[p][2] + dq[p][3];
            const u32 q = q0 + p + PF;
            fetch(p, blockIdx.x + (q / per) * gridDim.x, q % per);
        }
    }
    if (acc == 0x12345678u) out[0] = tab[l];
}
template <int PF>
float run4(const std::vector<void *> &bufs, u32 bands, u32 nK2, u32 *out, int iters) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    int grid = bands < 256 ? bands : 256;
    for (size_t i = 0; i < bufs.size(); i++) hipLaunchKernelGGL((stream4<PF>), dim3(grid), dim3(1024), 0, 0, (const u32 *)bufs[i], bands, nK2, out);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int it = 0; it < iters; it++) hipLaunchKernelGGL((stream4<PF>), dim3(grid), dim3(1024), 0, 0, (const u32 *)bufs[it % bufs.size()], bands, nK2, out);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / iters;
}

int main() {
    u32 *out; CHECK(hipMalloc(&out, 64));
    struct S { u32 bands, nK2; const char *name; };
    for (S sh : {S{128, 128, "4096x4096"}, S{384, 128, "3x4096x4096"}, S{688, 128, "2x11008x4096"}, S{128, 352, "4096x11264"}, S{256, 256, "8192x8192"}}) {
        size_t bytes = (size_t)sh.bands * sh.nK2 * 256;
        size_t nbuf = (600u << 20) / bytes + 1; if (nbuf > 64) nbuf = 64;
        std::vector<void *> bufs(nbuf);
        for (auto &b : bufs) { CHECK(hipMalloc(&b, bytes)); CHECK(hipMemset(b, 1, bytes)); }
        CHECK(hipDeviceSynchronize());
        int it = 200;
        float a = run<8, 1, 0, 1, 0>(bufs, sh.bands, sh.nK2, out, it);
        float b = run<8, 1, 0, 1, 1>(bufs, sh.bands, sh.nK2, out, it);
        float c = run<16, 1, 0, 1, 0>(bufs, sh.bands, sh.nK2, out, it);
        float d = run<16, 1, 0, 1, 1>(bufs, sh.bands, sh.nK2, out, it);
        float e = run<8, 1, 0, 0, 1>(bufs, sh.bands, sh.nK2, out, it);
        float f = run<4, 1, 0, 1, 1>(bufs, sh.bands, sh.nK2, out, it);
        float x2 = run4<2>(bufs, sh.bands, sh.nK2, out, it), x4 = run4<4>(bufs, sh.bands, sh.nK2, out, it), x1 = run4<1>(bufs, sh.bands, sh.nK2, out, it);
        printf("%-14s 16 B per lane (1 KiB per instruction): 1 slot %.2f us | 2 slots %.2f us (%.0f GB/s) | 4 slots %.2f us (%.0f GB/s)\n", sh.name, x1, x2, bytes / x2 / 1e3, x4, bytes / x4 / 1e3);
        printf("%-14s %5.1f MB: PF8 engine order %.2f us (%.0f GB/s) | PF8 natural order %.2f (%.0f GB/s) | PF16 engine %.2f | PF16 natural %.2f | PF8 natural small-LDS %.2f | PF4 natural %.2f\n",
               sh.name, bytes / 1e6, a, bytes / a / 1e3, b, bytes / b / 1e3, c, d, e, f);
        for (auto &bb : bufs) CHECK(hipFree(bb));
    }
    return 0;
}
