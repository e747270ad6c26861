// lutgemm.hip -- LUT-GEMM (BCQ) GEMV for gfx950.  Replaces nqmv_bias (inference/ap_gemv/lutgemm.cu:24-149).
//
// The reference launches one block per 32-activation tile and atomically adds fp16 tile results into the output, in
// an undefined order.  Here one block owns a slice of the outputs and walks the tiles in ASCENDING order (one valid
// execution of the reference's atomics, and deterministic): per group of G tiles the 4 x 256 sign-sum tables of every
// tile are built in LDS with the reference's fp16 operation order (lutgemm.cu:40-78), then every lane accumulates its
// outputs tile by tile with the same fp16 multiply / add sequence (lutgemm.cu:94-145).  Weight words are read
// coalesced (consecutive outputs of one (tile, plane) are consecutive in memory).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gq_internal.h"

namespace {
typedef uint32_t u32;
typedef _Float16 h16;
__device__ __forceinline__ h16 u2h(uint16_t h) { return __builtin_bit_cast(h16, h); }
__device__ __forceinline__ uint16_t h2u(h16 h) { return __builtin_bit_cast(uint16_t, h); }

constexpr int G = 8;  // tiles per LDS round: 8 * 4 * 256 halves = 16 KiB

__global__ void __launch_bounds__(256) lutgemm_kernel(const uint16_t *x, uint16_t *out, const u32 *W, const uint16_t *alpha,
                                                      const uint16_t *q_bias, u32 N, u32 K, int bits, u32 group_size) {
    __shared__ uint16_t lut[G][4][256];
    const u32 tid = threadIdx.x;
    const u32 m = (blockIdx.x * 256u + tid);  // one output per lane
    const bool ok = m < N;
    h16 acc = ok ? u2h(out[m]) : (h16)0;
    const u32 ntiles = K / 32u;
    for (u32 kt0 = 0; kt0 < ntiles; kt0 += G) {
        // build: thread -> (tile gi, table y, v < 64) for 4 passes, then the two doubling steps
        for (u32 e = tid; e < (u32)G * 4u * 64u; e += 256u) {
            const u32 v = e & 63u, y = (e >> 6) & 3u, gi = e >> 8;
            if (kt0 + gi < ntiles) {
                const uint16_t *xi = x + 32u * (kt0 + gi) + 8u * y;
                h16 a = (h16)0;
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const h16 sg = ((v >> i) & 1u) ? (h16)1.0f : (h16)-1.0f;
                    const h16 term = sg * u2h(xi[i]);
                    a = i == 0 ? term : a + term;
                }
                const h16 i6 = (h16)2.0f * u2h(xi[6]), i7 = (h16)2.0f * u2h(xi[7]);
                const h16 b = a + i6;        // v + 64
                lut[gi][y][v] = h2u(a);
                lut[gi][y][v + 64] = h2u(b);
                lut[gi][y][v + 128] = h2u(a + i7);
                lut[gi][y][v + 192] = h2u(b + i7);
            }
        }
        __syncthreads();
        if (ok) {
            for (u32 gi = 0; gi < (u32)G && kt0 + gi < ntiles; gi++) {
                const u32 kt = kt0 + gi;
                const u32 g = (kt * 32u) / group_size;
                h16 all = (h16)0;
#pragma unroll
                for (int y = 0; y < 4; y++) all = all + u2h(lut[gi][y][255]);
                h16 o = (h16)0 + u2h(q_bias[(size_t)g * N + m]) * all;
                h16 a = u2h(alpha[(size_t)g * bits * N + m]);
                for (int b = 0; b < bits; b++) {
                    const u32 w = W[((size_t)kt * bits + b) * N + m];
                    h16 t = (h16)0;
#pragma unroll
                    for (int y = 0; y < 4; y++) t = t + u2h(lut[gi][y][(w >> (8 * y)) & 255u]);
                    o = o + a * t;
                    a = a * (h16)2.0f;
                }
                acc = acc + o;
            }
        }
        __syncthreads();
    }
    if (ok) out[m] = h2u(acc);
}
}  // namespace

extern "C" int gq_lutgemm_gemv(const void *x, void *out, const uint32_t *qweight, const void *alpha, const void *q_bias,
                               uint32_t N, uint32_t K, int bits, int group_size, void *stream) {
    if (bits < 1 || bits > 8) return gq_fail(GQ_EINVAL, "Bitwidth must be between 1 and 8.");
    if (!x || !out || !qweight || !alpha || !q_bias) return gq_fail(GQ_EINVAL, "null pointer argument.");
    if (K == 0 || K % 32u || N == 0) return gq_fail(GQ_EINVAL, "need N > 0 and K a positive multiple of 32.");
    if (group_size <= 0 || K % (uint32_t)group_size || (uint32_t)group_size % 32u)
        return gq_fail(GQ_EINVAL, "group_size must be a multiple of 32 that divides input_feat.");
    hipLaunchKernelGGL(lutgemm_kernel, dim3((N + 255u) / 256u), dim3(256), 0, (hipStream_t)stream, (const uint16_t *)x,
                       (uint16_t *)out, qweight, (const uint16_t *)alpha, (const uint16_t *)q_bias, N, K, bits, (u32)group_size);
    GQ_HIP_CHECK(hipGetLastError());
    return GQ_OK;
}
