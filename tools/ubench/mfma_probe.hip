// mfma_probe.hip -- checks v_mfma_scale_f32_16x16x128_f8f6f4 semantics needed by the plane-MFMA GEMV:
//  (1) operand layout: lane l: row/col = l%16, k = 32*(l/16) + 4*vgpr + byte
//  (2) fp8/bf8 single-bit patterns incl. denormals are honoured
//  (3) E8M0 scale operand per lane
//  (4) issue rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int FA, int FB>
__global__ void probe(const uint32_t *A, const uint32_t *B, float *D, int scale_a, int scale_b) {
    int l = threadIdx.x;
    v8i a, b;
    for (int i = 0; i < 8; i++) { a[i] = A[l * 8 + i]; b[i] = B[l * 8 + i]; }
    v4f c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, FA, FB, 0, scale_a, 0, scale_b);
    for (int i = 0; i < 4; i++) D[l * 4 + i] = c[i];
}

__global__ void rate(float *out, int iters) {
    v8i a, b;
    for (int i = 0; i < 8; i++) { a[i] = threadIdx.x * 7 + i; b[i] = threadIdx.x * 3 + i; }
    v4f c0 = {0,0,0,0}, c1 = c0, c2 = c0, c3 = c0;
    for (int it = 0; it < iters; it++) {
        c0 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c0, 1, 1, 0, 127, 0, 127);
        c1 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c1, 1, 1, 0, 127, 0, 127);
        c2 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c2, 1, 1, 0, 127, 0, 127);
        c3 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c3, 1, 1, 0, 127, 0, 127);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

static double e5m2(uint8_t v) { int s = v >> 7, e = (v >> 2) & 31, m = v & 3; double r = e ? ldexp(1.0 + m / 4.0, e - 15) : ldexp(m / 4.0, -14); return s ? -r : r; }
static double e4m3(uint8_t v) { int s = v >> 7, e = (v >> 3) & 15, m = v & 7; if (e == 15 && m == 7) return NAN; double r = e ? ldexp(1.0 + m / 8.0, e - 7) : ldexp(m / 8.0, -6); return s ? -r : r; }

template <int FA, int FB>
void run(const char *name, int sa, int sb) {
    std::vector<uint32_t> A(64 * 8), B(64 * 8);
    std::vector<double> Am(16 * 128), Bm(128 * 16);
    srand(1);
    auto dec = [](int f, uint8_t v) { return f == 0 ? e4m3(v) : e5m2(v); };
    for (int l = 0; l < 64; l++)
        for (int v = 0; v < 8; v++) {
            uint32_t wa = 0, wb = 0;
            for (int by = 0; by < 4; by++) {
                uint8_t ba = (rand() & 1) ? (uint8_t)(1u << (rand() % 7)) : 0;  // single-bit patterns (incl. denormals)
                uint8_t bb = (uint8_t)(rand() & 0xFF);
                if (FB == 0 && (bb & 0x7F) == 0x7F) bb = 0x10;  // no NaN
                if (FB == 1 && ((bb >> 2) & 31) == 31) bb = 0x21;  // no inf/nan
                wa |= (uint32_t)ba << (8 * by);
                wb |= (uint32_t)bb << (8 * by);
                int k = 32 * (l / 16) + 4 * v + by;
                Am[(l % 16) * 128 + k] = dec(FA, ba);
                Bm[k * 16 + (l % 16)] = dec(FB, bb);
            }
            A[l * 8 + v] = wa;
            B[l * 8 + v] = wb;
        }
    uint32_t *dA, *dB; float *dD;
    CHECK(hipMalloc(&dA, A.size() * 4)); CHECK(hipMalloc(&dB, B.size() * 4)); CHECK(hipMalloc(&dD, 256 * 4));
    CHECK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL((probe<FA, FB>), dim3(1), dim3(64), 0, 0, dA, dB, dD, sa, sb);
    std::vector<float> D(256);
    CHECK(hipMemcpy(D.data(), dD, 256 * 4, hipMemcpyDeviceToHost));
    double maxerr = 0, maxref = 0;
    for (int l = 0; l < 64; l++)
        for (int r = 0; r < 4; r++) {
            int row = 4 * (l / 16) + r, col = l % 16;
            double ref = 0;
            for (int k = 0; k < 128; k++) ref += Am[row * 128 + k] * Bm[k * 16 + col];
            ref *= ldexp(1.0, (sa & 0xFF) - 127) * ldexp(1.0, (sb & 0xFF) - 127);
            maxerr = fmax(maxerr, fabs(ref - D[l * 4 + r]));
            maxref = fmax(maxref, fabs(ref));
        }
    printf("%s: scale_a=%d scale_b=%d  max|err|=%g  max|ref|=%g  %s\n", name, sa, sb, maxerr, maxref,
           maxerr <= 1e-4 * maxref ? "OK(~)" : "MISMATCH");
}


// realistic use: A = one bit position pattern or 0, B col c = bf8 piece c of x ~ N(0,1); check sum over pieces
static uint8_t to_e5m2(double v) {  // RNE, saturating, handles subnormals
    if (v == 0) return 0;
    uint8_t s = v < 0 ? 0x80 : 0; double a = fabs(v);
    if (a >= 61440.0) return s | 0x7B;
    int e; double m = frexp(a, &e);  // a = m * 2^e, m in [0.5,1)
    int E = e - 1;                    // a = (2m) * 2^E
    if (E < -14) { double q = nearbyint(a / ldexp(1.0, -16)); if (q >= 4) return s | 0x04; return s | (uint8_t)q; }
    double q = nearbyint((2 * m) * 4.0);  // 4..8
    if (q >= 8) { q = 4; E += 1; }
    if (E > 15) return s | 0x7B;
    return s | (uint8_t)(((E + 15) << 2) | ((int)q - 4));
}
void realistic(uint8_t pattern, int scale_a) {
    std::vector<uint32_t> A(64 * 8), B(64 * 8, 0);
    std::vector<double> x(128), bits(16 * 128);
    srand(7);
    for (int k = 0; k < 128; k++) { double u1 = (rand() + 1.0) / (RAND_MAX + 2.0), u2 = rand() / (double)RAND_MAX; x[k] = sqrt(-2 * log(u1)) * cos(6.283185307 * u2) * (k % 7 == 0 ? 30.0 : 1.0); }
    // pieces
    std::vector<uint8_t> P(4 * 128);
    std::vector<double> xr(128);
    for (int k = 0; k < 128; k++) { double r = (double)(float)x[k]; r = (double)__builtin_bit_cast(float, __builtin_bit_cast(uint32_t, (float)r)); xr[k] = r; for (int c = 0; c < 4; c++) { uint8_t p = to_e5m2(r); P[c * 128 + k] = p; r -= e5m2(p); } }
    for (int l = 0; l < 64; l++)
        for (int v = 0; v < 8; v++) {
            uint32_t wa = 0, wb = 0;
            for (int by = 0; by < 4; by++) {
                int k = 32 * (l / 16) + 4 * v + by;
                int bit = rand() & 1;
                bits[(l % 16) * 128 + k] = bit;
                wa |= (uint32_t)(bit ? pattern : 0) << (8 * by);
                if ((l % 16) < 4) wb |= (uint32_t)P[(l % 16) * 128 + k] << (8 * by);
            }
            A[l * 8 + v] = wa; B[l * 8 + v] = wb;
        }
    uint32_t *dA, *dB; float *dD;
    CHECK(hipMalloc(&dA, A.size() * 4)); CHECK(hipMalloc(&dB, B.size() * 4)); CHECK(hipMalloc(&dD, 256 * 4));
    CHECK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL((probe<1, 1>), dim3(1), dim3(64), 0, 0, dA, dB, dD, scale_a, 127);
    std::vector<float> D(256);
    CHECK(hipMemcpy(D.data(), dD, 256 * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int row = 0; row < 16; row++) {
        double got = 0, ref = 0, refabs = 0;
        for (int c = 0; c < 4; c++) got += D[(16 * (row / 4) + c) * 4 + (row % 4)];
        for (int k = 0; k < 128; k++) { ref += bits[row * 128 + k] * xr[k]; refabs += bits[row * 128 + k] * fabs(xr[k]); }
        worst = fmax(worst, fabs(got - ref) / refabs);
    }
    printf("realistic pattern=0x%02x scale_a=%d: worst |err|/sum|bx| = %.3g\n", pattern, scale_a, worst);
}

int main() {
    realistic(0x3C, 127);
    realistic(0x04, 127+14);
    realistic(0x01, 127+16);
    realistic(0x02, 127+15);
    realistic(0x40, 126);
    realistic(0x20, 127+7);

    run<1, 1>("A=bf8 B=bf8", 127, 127);
    run<1, 1>("A=bf8 B=bf8", 143, 120);
    run<0, 0>("A=fp8 B=fp8", 127, 127);
    run<0, 1>("A=fp8 B=bf8", 136, 127);
    run<1, 0>("A=bf8 B=fp8", 127, 130);
    // rate
    float *out; CHECK(hipMalloc(&out, 256 * 8 * 256 * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int wps : {1, 2}) {
        int iters = 4096;
        hipLaunchKernelGGL(rate, dim3(256), dim3(256 * wps), 0, 0, out, iters);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(rate, dim3(256), dim3(256 * wps), 0, 0, out, iters);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        double per = ms * 1e6 / (iters * 4.0 * wps);
        printf("mfma_scale 16x16x128 bf8: %d wave/SIMD: %.2f ns per MFMA per SIMD (%.1f cycles @2.4GHz)\n", wps, per, per * 2.4);
    }
    return 0;
}
