// lds_masked.hip -- what does a ds_read_b128 cost the LDS pipe when only some lanes are active?  (The stream kernel's consumer
// waves read their unit's B image with 16 of 64 lanes active -- the lanes of MFMA columns 0..3 -- and all 16 waves of a block do so
// at about the same time: profiles/r04_phase_stream.txt shows ~1,300 cycles between "image built" and "B in registers".)
// One 1024-thread block per CU; every wave issues REPS x 8 reads; cycles from the first read to the last wait, max over the waves.
//   mode 0: all 64 lanes          1: lanes with (l & 15) < 4 (the kernel's pattern)      2: lanes 0..15 only
//   mode 3: 16 lanes, ds_read_b64 x 2 per 16 bytes      4: nothing (loop overhead)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int REPS = 16;
template <int MODE>
__global__ void __launch_bounds__(1024) k(unsigned long long *out, uint32_t *sink, int waves_active) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    for (uint32_t i = threadIdx.x; i < 16384; i += blockDim.x) reinterpret_cast<uint32_t *>(smem)[i] = i;
    __syncthreads();
    const uint32_t l = threadIdx.x & 63u, w = threadIdx.x >> 6;
    if ((int)w >= waves_active) return;
    const uint32_t col = l & 15u, kb = l >> 4;
    const bool act = MODE == 0 ? true : (MODE == 1 ? col < 4u : (MODE == 4 ? false : l < 16u));
    // the kernel's addresses: unit image of 2 KiB, [b][piece][k]; lane (col, kb) reads 16 B at piece * 128 + 16 kb (+ 64)
    const uint32_t addr = (uint32_t)(uintptr_t)smem + w * 2048u + (MODE == 1 ? (col & 3u) * 128u + 16u * kb : (l & 15u) * 16u + (l >> 4) * 256u);
    uint4 acc = make_uint4(0, 0, 0, 0);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < REPS; r++) {
        uint4 v[8];
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = make_uint4(0, 0, 0, 0);
        if (act) {
            if (MODE == 3) {
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    uint2 a, b;
                    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(a) : "v"(addr), "n"(0) : "memory");
                    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(b) : "v"(addr), "n"(8) : "memory");
                    v[i] = make_uint4(a.x, a.y, b.x, b.y);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("ds_read_b128 %0, %1" : "=v"(v[i]) : "v"(addr + (i & 3) * 512u + (i >> 2) * 64u) : "memory");
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 8; i++) acc.x ^= v[i].x ^ v[i].w;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (l == 0) out[blockIdx.x * 16 + w] = t1 - t0;
    if (acc.x == 0x12345u) sink[0] = acc.x;
}
template <int MODE>
void run(const char *name, unsigned long long *d, uint32_t *sink, int waves) {
    std::vector<unsigned long long> h(256 * 16);
    CHECK(hipMemset(d, 0, h.size() * 8));
    for (int it = 0; it < 3; it++) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 65536, 0, d, sink, waves);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<unsigned long long> mx;
    for (int b = 0; b < 256; b++) mx.push_back(*std::max_element(h.begin() + b * 16, h.begin() + b * 16 + waves));
    std::sort(mx.begin(), mx.end());
    printf("%-28s waves %2d: %6.1f cycles per round of 8 reads per wave (median block, slowest wave)\n", name, waves, (double)mx[128] / REPS);
}
int main() {
    unsigned long long *d; uint32_t *sink;
    CHECK(hipMalloc(&d, 256 * 16 * 8)); CHECK(hipMalloc(&sink, 64));
    for (int waves : {1, 4, 16}) {
        run<4>("no reads", d, sink, waves);
        run<0>("64 lanes b128", d, sink, waves);
        run<1>("16 lanes (col < 4) b128", d, sink, waves);
        run<2>("16 lanes (0..15) b128", d, sink, waves);
        run<3>("16 lanes (0..15) 2 x b64", d, sink, waves);
    }
    return 0;
}
