// fp4_probe.hip -- v_mfma_scale_f32_16x16x128_f8f6f4 with A = FP4 (e2m1, cbsz=4) and B = BF8 (e5m2, blgp=1):
//   (1) value of the single-bit nibble patterns, (2) which A position (lane k-block, VGPR, nibble) multiplies which
//   B position (lane k-block, VGPR, byte).  Build: hipcc --offload-arch=gfx950 -O3 fp4_probe.hip -o fp4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

// block p: A one-hot nibble 0b0010 (1.0) at position p = (kb, v, i) of row 0; B column 0 from Bcol[128] bytes
// (index = kb'*32 + v'*4 + byte), other columns zero.  out[p] = D[0][0].
__global__ void onehot(const unsigned char *Bcol, float *out, int pat) {
    const int p = blockIdx.x, kbA = p >> 6, vA = (p >> 3) & 7, iA = p & 7;  // 256 positions: 4 lane groups x 8 VGPRs x 8 nibbles
    const int l = threadIdx.x, n = l & 15, kb = l >> 4;
    v8i a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
    if (n == 0 && kb == kbA) a[vA] = pat << (4 * iA);  // lane (row 0, kbA)
    if (n == 0) {
        for (int v = 0; v < 8; v++) {
            unsigned w = 0;
            for (int by = 0; by < 4; by++) w |= (unsigned)Bcol[kb * 32 + v * 4 + by] << (8 * by);
            b[v] = (int)w;
        }
    }
    v4f c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 4, 1, 0, 127, 0, 127);
    if (l == 0) out[p] = c[0];
}

int main() {
    unsigned char hB[128], *dB;
    float hO[256], hO2[256], *dO;
    hipMalloc(&dB, 128);
    hipMalloc(&dO, 1024);
    // (1) pattern values: B all 1.0
    for (int i = 0; i < 128; i++) hB[i] = 0x3C;
    hipMemcpy(dB, hB, 128, hipMemcpyHostToDevice);
    for (int pat = 1; pat <= 8; pat <<= 1) {
        hipLaunchKernelGGL(onehot, dim3(128), dim3(64), 0, 0, dB, dO, pat);
        hipMemcpy(hO, dO, 512, hipMemcpyDeviceToHost);
        printf("nibble pattern 0x%x -> %g\n", pat, hO[0]);
    }
    {
        hipLaunchKernelGGL(onehot, dim3(256), dim3(64), 0, 0, dB, dO, 2);
        hipMemcpy(hO, dO, 1024, hipMemcpyDeviceToHost);
        printf("B = all ones, A one-hot 1.0 at p = (lane group, VGPR 0..7, nibble): D[0][0]:\n");
        for (int p = 0; p < 256; p++) printf("%g%s", hO[p], (p & 63) == 63 ? "\n" : ((p & 7) == 7 ? "  " : " "));
    }
    for (int pat : {3, 5, 6, 7, 0xA}) {
        hipLaunchKernelGGL(onehot, dim3(128), dim3(64), 0, 0, dB, dO, pat);
        hipMemcpy(hO, dO, 512, hipMemcpyDeviceToHost);
        printf("nibble pattern 0x%x -> %g\n", pat, hO[0]);
    }
    // (2) correspondence: B[idx] = 2^(idx%16 - 8), then 2^(idx/16 - 4)   (bf8 powers of two: byte = (e+15)<<2)
    for (int i = 0; i < 128; i++) hB[i] = (unsigned char)(((i % 16) - 8 + 15) << 2);
    hipMemcpy(dB, hB, 128, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(onehot, dim3(256), dim3(64), 0, 0, dB, dO, 2);
    hipMemcpy(hO, dO, 1024, hipMemcpyDeviceToHost);
    for (int i = 0; i < 128; i++) hB[i] = (unsigned char)(((i / 16) - 4 + 15) << 2);
    hipMemcpy(dB, hB, 128, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(onehot, dim3(256), dim3(64), 0, 0, dB, dO, 2);
    hipMemcpy(hO2, dO, 1024, hipMemcpyDeviceToHost);
    printf("A position (kb, vgpr, nibble) -> B position (kb, vgpr, byte)\n");
    for (int p = 0; p < 256; p++) {
        if (hO[p] == 0.f) continue;
        const int lo = (int)lrintf(log2f(hO[p])) + 8, hi = (int)lrintf(log2f(hO2[p])) + 4, idx = hi * 16 + lo;
        printf("  A(%d,%d,%d) -> B(%d,%d,%d)%s", p >> 6, (p >> 3) & 7, p & 7, idx >> 5, (idx >> 2) & 7, idx & 3, (p & 3) == 3 ? "\n" : "");
    }
    return 0;
}
