#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
// A = 1.0 nibbles in lanes with (lane>>4) in groupmask and VGPRs in vmask; B = 1.0 bytes in lanes in bgroupmask, VGPRs bvmask
__global__ void k(float *out, int groupmask, int vmask, int bgroupmask, int bvmask, int cb, int bl, int onelane = -1) {
    const int l = threadIdx.x, kb = l >> 4;
    v8i a, b;
    for (int v = 0; v < 8; v++) {
        a[v] = ((groupmask >> kb) & 1) && ((vmask >> v) & 1) && (onelane < 0 || l == onelane) ? (cb == 4 ? 0x22222222 : 0x3C3C3C3C) : 0;
        b[v] = ((bgroupmask >> kb) & 1) && ((bvmask >> v) & 1) ? (bl == 4 ? 0x22222222 : 0x3C3C3C3C) : 0;
    }
    v4f c = {0, 0, 0, 0};
    if (cb == 4 && bl == 1) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 4, 1, 0, 127, 0, 127);
    else if (cb == 4 && bl == 4) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 4, 4, 0, 127, 0, 127);
    else if (cb == 1 && bl == 4) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 1, 4, 0, 127, 0, 127);
    else c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 1, 1, 0, 127, 0, 127);
    for (int i = 0; i < 4; i++) out[l * 4 + i] = c[i];
}
int main() {
    float *d, h, hh[256];
    hipMalloc(&d, 1024);
    auto run = [&](const char *name, int gm, int vm, int bgm, int bvm, int cb, int bl) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, gm, vm, bgm, bvm, cb, bl);
        hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        printf("%-70s D[0][0] = %g\n", name, h);
    };
    run("bf8 x bf8, everything ones", 15, 255, 15, 255, 1, 1);
    run("fp4 A (all lanes, VGPR 0-7) x bf8 B (all)", 15, 255, 15, 255, 4, 1);
    run("fp4 A (all lanes, VGPR 0-3) x bf8 B (all)", 15, 15, 15, 255, 4, 1);
    run("fp4 A (lanes 0-15, VGPR 0-3) x bf8 B (all)", 1, 15, 15, 255, 4, 1);
    run("fp4 A (lanes 16-31, VGPR 0-3) x bf8 B (all)", 2, 15, 15, 255, 4, 1);
    run("fp4 A (lanes 32-47, VGPR 0-3) x bf8 B (all)", 4, 15, 15, 255, 4, 1);
    run("fp4 A (lanes 48-63, VGPR 0-3) x bf8 B (all)", 8, 15, 15, 255, 4, 1);
    run("fp4 A (all) x bf8 B (lanes 0-15 all VGPR)", 15, 255, 1, 255, 4, 1);
    run("fp4 A (all) x bf8 B (lanes 16-31)", 15, 255, 2, 255, 4, 1);
    run("fp4 A (all) x bf8 B (lanes 32-47)", 15, 255, 4, 255, 4, 1);
    run("fp4 A (all) x bf8 B (lanes 48-63)", 15, 255, 8, 255, 4, 1);
    run("fp4 A (all) x bf8 B (all lanes, VGPR 0-3)", 15, 255, 15, 15, 4, 1);
    run("fp4 A (all) x bf8 B (all lanes, VGPR 4-7)", 15, 255, 15, 240, 4, 1);
    run("fp4 x fp4 all", 15, 255, 15, 255, 4, 4);
    run("fp4 A lanes 16-31 x fp4 B all", 2, 15, 15, 15, 4, 4);
    run("bf8 A all x fp4 B all", 15, 255, 15, 255, 1, 4);
    for (int lane : {0, 1, 16, 17, 32, 48}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 15, 15, 15, 255, 4, 1, lane);
        hipMemcpy(hh, d, 1024, hipMemcpyDeviceToHost);
        printf("fp4 A only lane %d: nonzero outputs (lane:reg=value):", lane);
        for (int i = 0; i < 256; i++) if (hh[i] != 0.f && (i / 4) % 16 == 0) printf(" %d:%d=%g", i / 4, i % 4, hh[i]);
        printf("\n");
    }
    return 0;
}
