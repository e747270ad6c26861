// scale_probe.hip -- which lane group's E8M0 scale does v_mfma_scale_f32_16x16x128_f8f6f4 (A = FP4, B = BF8) apply to B element k?
// B = one-hot (1.0) at k in column 0, A = all ones, scale_b = 127 + 4 * (lane / 16): D[0][0] = 2^(4 * group used for k).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void probe(float *D, int k, int vary_a) {
    const int l = threadIdx.x, col = l & 15, kb = l >> 4;
    v8i a, b;
    for (int i = 0; i < 8; i++) { a[i] = 0x22222222; b[i] = 0; }   // FP4 0x2 = 1.0
    // bf8 B: lane (col, kb) register r byte t holds k = 64 (r / 4) + 16 kb + 4 (r % 4) + t
    if (col == 0) {
        for (int r = 0; r < 8; r++) for (int t = 0; t < 4; t++)
            if (64 * (r / 4) + 16 * kb + 4 * (r % 4) + t == k) b[r] |= 0x3C << (8 * t);
    }
    v4f c = {0, 0, 0, 0};
    const int sa = vary_a ? 127 + 4 * kb : 127, sb = vary_a ? 127 : 127 + 4 * kb;
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 4, 1, 0, sa, 0, sb);
    if (l == 0) D[0] = c[0];
}
int main() {
    float *D; hipMalloc(&D, 16); float h;
    for (int va = 0; va < 2; va++) {
        printf("%s scale = 127 + 4 * lane_group: group whose scale multiplies B element k (k = 0..127):\n", va ? "A" : "B");
        for (int k = 0; k < 128; k++) {
            hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, D, k, va);
            hipMemcpy(&h, D, 4, hipMemcpyDeviceToHost);
            printf("%d", (int)lround(log2f(h) / 4));
            if (k % 32 == 31) printf(" ");
        }
        printf("\n");
    }
    return 0;
}
