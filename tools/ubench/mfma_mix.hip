// mfma_mix.hip -- issue rate of the plane-GEMV inner pattern: 8 v_and_b32 producing the A operand + one scaled MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

// MODE 0: MFMA only.  1: 8 ANDs -> A (fresh regs reused: WAR with previous MFMA).  2: 8 ANDs into 2 alternating A sets.
// 3: like 1 but only 4 ANDs (half the VALU).  4: like 1 with B read from LDS each MFMA
template <int MODE, int NACC>
__global__ void k(float *out, const int *src, int iters) {
    __shared__ int lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i * 77;
    __syncthreads();
    int w[8];
    for (int i = 0; i < 8; i++) w[i] = src[threadIdx.x * 8 + i];
    v8i b;
    for (int i = 0; i < 8; i++) b[i] = src[threadIdx.x + i * 64];
    v4f acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = (v4f){0, 0, 0, 0};
    v8i a0 = {w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7]}, a1 = a0;
    int m = 0x04040404;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < 6; j++) {
            if (MODE == 1 || MODE == 4) {
#pragma unroll
                for (int i = 0; i < 8; i++) a0[i] = w[i] & (m << (j & 3));
            } else if (MODE == 2) {
                if (j & 1) {
#pragma unroll
                    for (int i = 0; i < 8; i++) a1[i] = w[i] & (m << (j & 3));
                } else {
#pragma unroll
                    for (int i = 0; i < 8; i++) a0[i] = w[i] & (m << (j & 3));
                }
            } else if (MODE == 3) {
#pragma unroll
                for (int i = 0; i < 4; i++) a0[i] = w[i] & (m << (j & 3));
            }
            if (MODE == 4) {
                const int4 *p = reinterpret_cast<const int4 *>(lds + ((threadIdx.x & 15) * 8 + j * 128));
                int4 x0 = p[0], x1 = p[1];
                b = (v8i){x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
            }
            if (MODE == 10 || MODE == 11) {  // the stream kernel's pattern: 4 masks of the FP4 operand per MFMA (+ 4 unrelated VALU operations)
#pragma unroll
                for (int i = 0; i < 4; i++) a0[i] = w[i] & (0x11111111 << (j & 3));
            }
            if (MODE == 11) {
#pragma unroll
                for (int i = 4; i < 8; i++) w[i] = (w[i] >> 1) ^ 0x5555;
            }
            if (MODE == 6) {
#pragma unroll
                for (int i = 0; i < 4; i++) a0[i] = w[i] & (0x11111111 << (j & 3));
            }
            v8i &aa = (MODE == 2 && (j & 1)) ? a1 : a0;
            if (MODE == 5 || MODE == 6)  // fp4 x fp4
                acc[j % NACC] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(aa, b, acc[j % NACC], 4, 4, 0, 141, 0, 127);
            else if (MODE == 7 || MODE == 10 || MODE == 11)  // fp4 A x bf8 B
                acc[j % NACC] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(aa, b, acc[j % NACC], 4, 1, 0, 141, 0, 127);
            else if (MODE == 8)  // fp6 x fp6
                acc[j % NACC] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(aa, b, acc[j % NACC], 2, 2, 0, 141, 0, 127);
            else if (MODE == 9)  // fp4 A x fp6 B
                acc[j % NACC] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(aa, b, acc[j % NACC], 4, 2, 0, 141, 0, 127);
            else
                acc[j % NACC] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(aa, b, acc[j % NACC], 1, 1, 0, 141, 0, 127);
        }
#pragma unroll
        for (int i = 0; i < 8; i++) w[i] = w[i] * 3 + 1;
    }
    float r = 0;
    for (int i = 0; i < NACC; i++) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE, int NACC>
void run(const char *name, float *out, int *src) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int wps : {1, 3, 4}) {
        int iters = 2000;
        hipLaunchKernelGGL((k<MODE, NACC>), dim3(256), dim3(256 * wps), 0, 0, out, src, 10);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<MODE, NACC>), dim3(256), dim3(256 * wps), 0, 0, out, src, iters);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        double per = ms * 1e6 / (iters * 6.0 * wps);
        printf("%-44s nacc=%d waves/SIMD=%d: %.1f ns per MFMA per SIMD (~%.0f cycles @2.2GHz)\n", name, NACC, wps, per, per * 2.2);
    }
}

int main() {
    float *out; int *src;
    CHECK(hipMalloc(&out, 256 * 1024 * 4)); CHECK(hipMalloc(&src, 1024 * 8 * 4 + 4096)); CHECK(hipMemset(src, 0x15, 1024 * 8 * 4 + 4096));
    run<0, 3>("MFMA only", out, src);
    run<0, 6>("MFMA only", out, src);
    run<1, 3>("8 v_and -> A (same regs) + MFMA", out, src);
    run<1, 6>("8 v_and -> A (same regs) + MFMA", out, src);
    run<2, 3>("8 v_and -> A (alternating reg sets) + MFMA", out, src);
    run<3, 3>("4 v_and + MFMA", out, src);
    run<4, 3>("8 v_and + B from LDS + MFMA", out, src);
    run<5, 3>("fp4 x fp4 MFMA only", out, src);
    run<6, 3>("fp4 x fp4, 4 v_and -> A + MFMA", out, src);
    run<7, 3>("fp4 A x bf8 B MFMA only", out, src);
    run<8, 3>("fp6 x fp6 MFMA only", out, src);
    run<9, 3>("fp4 A x fp6 B MFMA only", out, src);
    run<10, 3>("fp4 A x bf8 B, 4 v_and -> A + MFMA", out, src);
    run<11, 3>("fp4 A x bf8 B, 4 v_and + 8 other VALU + MFMA", out, src);
    return 0;
}
