// cpu_twins.cpp -- host (CPU) twins of the Any-Precision entry points, part of libgq_hip.so.
//
// The reference's gpt-fast path has no CPU kernel (inference/APLinear.py:17,22,33 hard-code 'cuda'); BASELINE.json's
// configs[0] nevertheless names a "CPU reference APLinear path via generate.py (plumbing, no GPU)": the module semantics of
// inference/APLinear.py:35-60 -- seq_len > 1: dequantise + matmul, seq_len == 1: GEMV into the persistent output -- with the
// tensors in host memory.  These twins serve that path (ap_gemv.py dispatches on the tensors' device) straight from the
// packed format (any_precision/quantization/pack.py:304-321): no dequantised copy of the matrix is made for the GEMV.
//
// Arithmetic contract (documented tolerance, tests/test_cpu_twins_cpu.py): products of the fp16 operands are exact in fp32,
// accumulation is fp32 in 8 interleaved lanes per row (+ a fixed reduction), one rounding to fp16 at the end:
//     |out - exact| <= 2^-11 |exact| + 1e-5 * sum_k |w_k x_k|
// i.e. the same envelope as the GPU's fast mode, and closer to the exact product than the reference CUDA kernel's fp16
// accumulation (anyprec.cu:495-512).  This is product code: it is tested AGAINST oracle/, it never calls it.
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <cmath>
#include <vector>

#if defined(_OPENMP)
#include <omp.h>
#endif
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include "../../include/gq_hip.h"

int gq_fail(int code, const char *msg);

namespace {

// ---------------------------------------------------------------------------------------------- scalar fp16 <-> fp32
inline float h2f(uint16_t h) {
    const uint32_t s = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 1023u;
    uint32_t u;
    if (e == 0) {
        if (m == 0) {
            u = s;
        } else {  // subnormal: value = m * 2^-24, exact in fp32
            float f = (float)m * 5.9604644775390625e-08f;
            memcpy(&u, &f, 4);
            u |= s;
        }
    } else if (e == 31) {
        u = s | 0x7f800000u | (m << 13);
    } else {
        u = s | ((e + 112u) << 23) | (m << 13);
    }
    float f;
    memcpy(&f, &u, 4);
    return f;
}

inline uint16_t f2h(float f) {  // round to nearest even, subnormals, overflow to inf
    uint32_t u;
    memcpy(&u, &f, 4);
    const uint32_t s = (u >> 16) & 0x8000u;
    u &= 0x7fffffffu;
    if (u >= 0x7f800000u) return (uint16_t)(s | 0x7c00u | (u > 0x7f800000u ? 0x200u : 0u));
    if (u >= 0x477ff000u) return (uint16_t)(s | 0x7c00u);  // >= 65520 rounds to inf
    if (u < 0x38800000u) {                                   // below the smallest normal half: result is a multiple of 2^-24
        if (u < 0x33000000u) return (uint16_t)s;             // < 2^-25 rounds to zero (2^-25 itself ties to even = 0)
        float a;
        memcpy(&a, &u, 4);
        const float r = a * 16777216.0f;  // exact scaling; rintf = RNE
        return (uint16_t)(s | (uint32_t)rintf(r));
    }
    const uint32_t lsb = (u >> 13) & 1u;
    u += 0xfffu + lsb;
    return (uint16_t)(s | ((u - 0x38000000u) >> 13));
}

struct Chunk {
    uint32_t word0, tpw, e0;  // first word of the chunk in a plane row, words in it, first weight index
};

// the chunk walk of the packed format: full 1024-weight chunks of 32 words, then one tail chunk of (K % 1024) / 32 words
std::vector<Chunk> chunks_of(uint32_t K) {
    std::vector<Chunk> v;
    const uint32_t full = K / 1024u;
    for (uint32_t i = 0; i < full; i++) v.push_back({32u * i, 32u, 1024u * i});
    if (K % 1024u) v.push_back({32u * full, (K % 1024u) / 32u, 1024u * full});
    return v;
}

// ---------------------------------------------------------------------------------------------- one row, scalar
float row_dot_scalar(const uint32_t *const *planes, int bits, const float *lutf, const float *xf, const std::vector<Chunk> &ch) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (const Chunk &c : ch) {
        for (uint32_t t = 0; t < c.tpw; t++) {
            uint32_t w[8];
            for (int p = 0; p < bits; p++) w[p] = planes[p][c.word0 + t];
            for (uint32_t cc = 0; cc < 4; cc++) {
                const float *xp = xf + c.e0 + cc * 8u * c.tpw + 8u * t;
                for (uint32_t j = 0; j < 8; j++) {
                    const uint32_t bit = 31u - (8u * cc + j);
                    uint32_t code = 0;
                    for (int p = 0; p < bits; p++) code = (code << 1) | ((w[p] >> bit) & 1u);
                    acc[j] = fmaf(lutf[code], xp[j], acc[j]);
                }
            }
        }
    }
    return ((acc[0] + acc[4]) + (acc[2] + acc[6])) + ((acc[1] + acc[5]) + (acc[3] + acc[7]));
}

#if defined(__x86_64__)
// ---------------------------------------------------------------------------------------------- one row, AVX2
// 8 weights (one byte of every plane, MSB first) per step: lane j takes bit 7 - j of each plane byte, the code indexes the
// row's LUT held in one (<= 3 bits) or two (4 bits) vector registers, one fused multiply-add per 8 weights.
template <int BITS>
__attribute__((target("avx2,fma"))) float row_dot_avx2(const uint32_t *const *planes, const float *lutf, const float *xf,
                                                        const std::vector<Chunk> &ch) {
    const __m256i sh = _mm256_setr_epi32(7, 6, 5, 4, 3, 2, 1, 0), one = _mm256_set1_epi32(1);
    __m256 lut_lo, lut_hi;
    {
        float tmp[16];
        for (int i = 0; i < 16; i++) tmp[i] = i < (1 << BITS) ? lutf[i] : 0.f;
        lut_lo = _mm256_loadu_ps(tmp);
        lut_hi = _mm256_loadu_ps(tmp + 8);
    }
    __m256 acc0 = _mm256_setzero_ps(), acc1 = _mm256_setzero_ps();
    for (const Chunk &c : ch) {
        for (uint32_t t = 0; t < c.tpw; t++) {
            uint32_t w[BITS];
            for (int p = 0; p < BITS; p++) w[p] = planes[p][c.word0 + t];
            for (uint32_t cc = 0; cc < 4; cc++) {
                __m256i code = _mm256_setzero_si256();
                for (int p = 0; p < BITS; p++) {
                    const __m256i b = _mm256_set1_epi32((int)((w[p] >> (24u - 8u * cc)) & 0xffu));
                    code = _mm256_or_si256(_mm256_slli_epi32(code, 1), _mm256_and_si256(_mm256_srlv_epi32(b, sh), one));
                }
                __m256 wv = _mm256_permutevar8x32_ps(lut_lo, code);
                if (BITS == 4) {
                    const __m256 hi = _mm256_permutevar8x32_ps(lut_hi, code);
                    wv = _mm256_blendv_ps(wv, hi, _mm256_castsi256_ps(_mm256_slli_epi32(code, 28)));  // bit 3 -> sign bit
                }
                const __m256 xv = _mm256_loadu_ps(xf + c.e0 + cc * 8u * c.tpw + 8u * t);
                if (cc & 1u)
                    acc1 = _mm256_fmadd_ps(wv, xv, acc1);
                else
                    acc0 = _mm256_fmadd_ps(wv, xv, acc0);
            }
        }
    }
    float a[8];
    _mm256_storeu_ps(a, _mm256_add_ps(acc0, acc1));
    return ((a[0] + a[4]) + (a[2] + a[6])) + ((a[1] + a[5]) + (a[3] + a[7]));
}

bool have_avx2() {
    static const bool ok = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma");
    return ok;
}
#endif

float row_dot(const uint32_t *const *planes, int bits, const float *lutf, const float *xf, const std::vector<Chunk> &ch) {
#if defined(__x86_64__)
    if (have_avx2()) {
        switch (bits) {
            case 2: return row_dot_avx2<2>(planes, lutf, xf, ch);
            case 3: return row_dot_avx2<3>(planes, lutf, xf, ch);
            case 4: return row_dot_avx2<4>(planes, lutf, xf, ch);
            default: break;
        }
    }
#endif
    return row_dot_scalar(planes, bits, lutf, xf, ch);
}

int check_shape(uint32_t N, uint32_t K, int bits) {
    if (bits < 2 || bits > 8) return gq_fail(GQ_EINVAL, "Bitwidth must be between 2 and 8.");
    if (K == 0 || K % 32u) return gq_fail(GQ_EINVAL, "input_feat (K) must be a positive multiple of 32.");
    if (N == 0) return gq_fail(GQ_EINVAL, "output_feat (N) must be positive.");
    return GQ_OK;
}

}  // namespace

extern "C" int gq_anyprec_gemv_cpu(const void *x, void *out, const uint32_t *qweight, const void *lut, uint32_t M, uint32_t N,
                                   uint32_t K, int bits, int dtype, int nthreads) {
    if (dtype != GQ_DTYPE_F16) return gq_fail(GQ_ENOTSUP, "only fp16 is implemented (as in the reference, gemv.cu:46-49).");
    if (int rc = check_shape(N, K, bits)) return rc;
    if (M < 1) return gq_fail(GQ_EINVAL, "batch size M must be positive.");
    if (!x || !out || !qweight || !lut) return gq_fail(GQ_EINVAL, "null pointer argument.");
    const std::vector<Chunk> ch = chunks_of(K);
    const uint16_t *x16 = (const uint16_t *)x, *lut16 = (const uint16_t *)lut;
    uint16_t *o16 = (uint16_t *)out;
    const size_t plane_stride = (size_t)N * (K / 32u);  // anyprec.cu:446: N * K / 32 words between planes
    const int L = 1 << bits;
    std::vector<float> xf((size_t)M * K);
    for (size_t i = 0; i < xf.size(); i++) xf[i] = h2f(x16[i]);
#if defined(_OPENMP)
    const int nt = nthreads > 0 ? nthreads : omp_get_max_threads();
#pragma omp parallel for schedule(static, 8) num_threads(nt)
#endif
    for (int64_t n = 0; n < (int64_t)N; n++) {
        float lutf[256];
        for (int i = 0; i < L; i++) lutf[i] = h2f(lut16[(size_t)n * L + i]);
        const uint32_t *planes[8];
        for (int p = 0; p < bits; p++) planes[p] = qweight + (size_t)p * plane_stride + (size_t)n * (K / 32u);
        for (uint32_t m = 0; m < M; m++) o16[(size_t)m * N + n] = f2h(row_dot(planes, bits, lutf, xf.data() + (size_t)m * K, ch));
    }
    return GQ_OK;
}

extern "C" int gq_anyprec_dequant_cpu(const uint32_t *qweight, const void *lut, void *W, uint32_t N, uint32_t K, int bits,
                                      int nthreads) {
    if (int rc = check_shape(N, K, bits)) return rc;
    if (!qweight || !lut || !W) return gq_fail(GQ_EINVAL, "null pointer argument.");
    const std::vector<Chunk> ch = chunks_of(K);
    const uint16_t *lut16 = (const uint16_t *)lut;
    uint16_t *w16 = (uint16_t *)W;
    const size_t plane_stride = (size_t)N * (K / 32u);
    const int L = 1 << bits;
#if defined(_OPENMP)
    const int nt = nthreads > 0 ? nthreads : omp_get_max_threads();
#pragma omp parallel for schedule(static, 8) num_threads(nt)
#endif
    for (int64_t n = 0; n < (int64_t)N; n++) {
        const uint16_t *lr = lut16 + (size_t)n * L;
        uint16_t *wr = w16 + (size_t)n * K;
        for (const Chunk &c : ch)
            for (uint32_t t = 0; t < c.tpw; t++) {
                uint32_t w[8];
                for (int p = 0; p < bits; p++) w[p] = qweight[(size_t)p * plane_stride + (size_t)n * (K / 32u) + c.word0 + t];
                for (uint32_t cc = 0; cc < 4; cc++)
                    for (uint32_t j = 0; j < 8; j++) {
                        const uint32_t bit = 31u - (8u * cc + j);
                        uint32_t code = 0;
                        for (int p = 0; p < bits; p++) code = (code << 1) | ((w[p] >> bit) & 1u);
                        wr[c.e0 + cc * 8u * c.tpw + 8u * t + j] = lr[code];
                    }
            }
    }
    return GQ_OK;
}
