#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
template <int OA, int OB>
__global__ void k(float *D, int sa, int sb) {
    v8i a, b;
    for (int i = 0; i < 8; i++) { a[i] = 0x3C3C3C3C; b[i] = 0x3C3C3C3C; }
    v4f c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 1, 1, OA, sa, OB, sb);
    if (threadIdx.x == 0) D[0] = c[0];
}
// per-lane scale: lane-dependent scale_b to learn which lanes' scales are used for which output
__global__ void lanes(float *D, const int *sa, const int *sb) {
    v8i a, b;
    for (int i = 0; i < 8; i++) { a[i] = 0x3C3C3C3C; b[i] = 0x3C3C3C3C; }
    v4f c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 1, 1, 0, sa[threadIdx.x], 0, sb[threadIdx.x]);
    for (int i = 0; i < 4; i++) D[threadIdx.x * 4 + i] = c[i];
}
int main() {
    float *D; hipMalloc(&D, 1024 * 4); float h[256];
    int sa = 127 | (128 << 8) | (129 << 16) | (130 << 24);
    int sb = 127 | (131 << 8) | (135 << 16) | (139 << 24);
#define RUN(OA, OB) hipLaunchKernelGGL((k<OA, OB>), dim3(1), dim3(64), 0, 0, D, sa, sb); hipMemcpy(h, D, 4, hipMemcpyDeviceToHost); printf("opsel_a=%d opsel_b=%d -> %g (log2 /128 = %g)\n", OA, OB, h[0], log2f(h[0] / 128));
    RUN(0,0) RUN(1,0) RUN(2,0) RUN(3,0) RUN(0,1) RUN(0,2) RUN(0,3) RUN(3,3)
    int hsa[64], hsb[64];
    for (int l = 0; l < 64; l++) { hsa[l] = 127; hsb[l] = 127 + (l % 16) + 16 * 0; }
    int *dsa, *dsb; hipMalloc(&dsa, 256); hipMalloc(&dsb, 256);
    // B scale varies with col (l%16) only
    hipMemcpy(dsa, hsa, 256, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(lanes, dim3(1), dim3(64), 0, 0, D, dsa, dsb); hipMemcpy(h, D, 1024, hipMemcpyDeviceToHost);
    printf("B scale = 127 + col: D[row0][col] log2/128: "); for (int c = 0; c < 16; c++) printf("%g ", log2f(h[c * 4] / 128)); printf("\n");
    // B scale varies with kb (l/16): each k-block contributes 32 * 2^kbscale
    for (int l = 0; l < 64; l++) hsb[l] = 127 + 4 * (l / 16);
    hipMemcpy(dsb, hsb, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(lanes, dim3(1), dim3(64), 0, 0, D, dsa, dsb); hipMemcpy(h, D, 1024, hipMemcpyDeviceToHost);
    printf("B scale = 127 + 4*kb: D[0][0] = %g (expect 32*(1+16+256+4096)=%g)\n", h[0], 32.0 * (1 + 16 + 256 + 4096));
    for (int l = 0; l < 64; l++) { hsb[l] = 127; hsa[l] = 127 + (l % 16); }
    hipMemcpy(dsa, hsa, 256, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(lanes, dim3(1), dim3(64), 0, 0, D, dsa, dsb); hipMemcpy(h, D, 1024, hipMemcpyDeviceToHost);
    printf("A scale = 127 + row: D[row][0] log2/128: "); for (int r = 0; r < 16; r++) printf("%g ", log2f(h[(16 * (r / 4)) * 4 + (r % 4)] / 128)); printf("\n");
    return 0;
}
