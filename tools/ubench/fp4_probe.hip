// fp4_probe.hip -- v_mfma_scale_f32_16x16x128_f8f6f4 with A = FP4 (e2m1, cbsz = 4) and B = BF8 (e5m2, blgp = 1):
// for an A one-hot (lane la, VGPR va, nibble ia) find, by brute force over all 2048 single B bytes, the B positions
// (lane lb, VGPR vb, byte bb) with a nonzero product (A nibble 0x2 = 1.0, B byte 0x3C = 1.0).
// Result: A (lane group g = lane / 16, VGPR v, nibble i) is k = 32 g + 8 v + i; B keeps k = 64 (vb / 4) + 16 kb + 4 (vb % 4) + bb;
// patterns 0x1 / 0x2 / 0x4 / 0x8 = 0.5 / 1 / 2 / -0.   Build: hipcc --offload-arch=gfx950 -O3 fp4_probe.hip -o fp4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void k(float *out, int la, int va, int ia) {
    const int q = blockIdx.x, lb = q >> 5, vb = (q >> 2) & 7, bb = q & 3;
    const int l = threadIdx.x;
    v8i a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int v = 0; v < 8; v++) {
        if (l == la && v == va) a[v] = 2 << (4 * ia);
        if (l == lb && v == vb) b[v] = 0x3C << (8 * bb);
    }
    v4f c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 4, 1, 0, 127, 0, 127);
    float s = fabsf(c[0]) + fabsf(c[1]) + fabsf(c[2]) + fabsf(c[3]);
    for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
    if (l == 0) out[q] = s;
}
int main() {
    float *d, h[2048];
    hipMalloc(&d, 2048 * 4);
    for (int la : {0, 16, 17, 32, 48})
        for (int va : {0, 3})
            for (int ia : {0, 5}) {
                hipLaunchKernelGGL(k, dim3(2048), dim3(64), 0, 0, d, la, va, ia);
                hipMemcpy(h, d, 2048 * 4, hipMemcpyDeviceToHost);
                printf("A(lane %d, vgpr %d, nibble %d) matches B:", la, va, ia);
                int first = 1;
                for (int q = 0; q < 2048; q++)
                    if (h[q] != 0.f && ((q >> 5) & 15) == 0) {  // column 0 (the same position matches in every column)
                        printf("%s (kb %d, vgpr %d, byte %d) value %g", first ? "" : ",", q >> 9, (q >> 2) & 7, q & 3, h[q]);
                        first = 0;
                    }
                printf("\n");
            }
    return 0;
}
