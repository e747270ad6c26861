// qtip.hip -- QTIP trellis-decoded matvec and the Sylvester Hadamard transform for gfx950.
//
// Replaces kernel_decompress_matvec<16,9,R,1,M,1,K> (qtip/qtip-kernels/src/inference.cu:168-425) -- an mma.sync
// program with a 32x-replicated shared-memory codebook, inline-PTX streaming loads and 73 compile-time shapes -- by
// one runtime-shaped wave-64 kernel:
//   * a wave walks 2x2 tile blocks (32 rows x 32 columns, 128*R contiguous bytes); lane (a4 = lane/32, s = lane%32)
//     owns the 4 trellis states 4s..4s+3 of the tiles of tile-row parity a4, i.e. rows a, a+8 (a = s/4) and
//     columns 2b, 2b+1, 2b+8, 2b+9 (b = s%4) -- the A-fragment slot order of the format (finetune.py:291-296);
//   * the 16-bit sliding window needs the next lane's stream unit (one ds_bpermute per tile);
//   * state -> (state*(state+1)) >> 6 & 511 -> fp16 pair from a 2 KiB LDS table, sign bit folded in with one XOR
//     (quantlut_sym, bitshift.py:72-80); v_dot2_f32_f16 accumulates in fp32;
//   * the K range of a 32-row band is split over the waves of the block and combined in LDS in a fixed order.
// Also: gq_hadamard, y = scale * x @ H_n for n a power of two (fast_hadamard_transform as used by
// inference/lib/utils/matmul_had.py:96-106).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gq_internal.h"

namespace {
typedef uint32_t u32;
typedef unsigned long long u64;
typedef _Float16 h16;
typedef h16 h16x2 __attribute__((ext_vector_type(2)));

template <int R>
__device__ __forceinline__ u32 unit_of(const u32 (&d)[R], u32 j);  // R-byte little-endian unit j of the 4R-byte s-row
template <>
__device__ __forceinline__ u32 unit_of<2>(const u32 (&d)[2], u32 j) { return (d[j >> 1] >> (16u * (j & 1u))) & 0xFFFFu; }
template <>
__device__ __forceinline__ u32 unit_of<3>(const u32 (&d)[3], u32 j) {
    switch (j) {
        case 0: return d[0] & 0xFFFFFFu;
        case 1: return (d[0] >> 24) | ((d[1] & 0xFFFFu) << 8);
        case 2: return (d[1] >> 16) | ((d[2] & 0xFFu) << 16);
        default: return d[2] >> 8;
    }
}
template <>
__device__ __forceinline__ u32 unit_of<4>(const u32 (&d)[4], u32 j) { return d[j]; }

template <int R>
__global__ void __launch_bounds__(1024) qtip_matvec_kernel(float *out, const u32 *comp, const uint16_t *x, const uint16_t *tlut,
                                                          u32 M, u32 K) {
    // the codebook sits in a STATIC LDS array (address 0, known to the compiler: the lookup address needs no base add)
    __shared__ __attribute__((aligned(16))) u32 tl[512];         // [512] half2 codebook
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t *xs = reinterpret_cast<uint16_t *>(smem);          // [K]
    float *part = reinterpret_cast<float *>(xs + K);            // [waves][32 rows]
    const u32 T = blockDim.x, tid = threadIdx.x, W = T >> 6, w = tid >> 6, l = tid & 63u;
    for (u32 i = tid; i < 512u; i += T) tl[i] = reinterpret_cast<const u32 *>(tlut)[i];
    for (u32 i = tid; i < K / 8u; i += T) reinterpret_cast<uint4 *>(xs)[i] = reinterpret_cast<const uint4 *>(x)[i];
    __syncthreads();
    const u32 M2 = blockIdx.x;           // 32-row band
    const u32 a4 = l >> 5, s = l & 31u;  // tile-row parity, stream unit
    const u32 a = s >> 2, b = s & 3u;
    const u32 nK2 = K / 32u;
    const unsigned char *band = reinterpret_cast<const unsigned char *>(comp) + (size_t)M2 * nK2 * 128u * R;
    float acc0 = 0.f, acc1 = 0.f;        // rows a and a + 8 of tile row 2*M2 + a4
    // The stream of a wave is a chain of 4R-byte-per-lane loads: with one load in flight per wave a CU has 4 KiB
    // outstanding and the kernel is latency-bound (measured 0.35 TB/s).  PF tile blocks are requested ahead.
    constexpr u32 PF = 8;
    u32 dq[PF][R];
    auto fetch = [&](u32 slot, u32 K2) {
        if (K2 < nK2) {
            const u32 *row = reinterpret_cast<const u32 *>(band + (size_t)K2 * 128u * R + (size_t)s * 4u * R);
#pragma unroll
            for (int i = 0; i < R; i++) dq[slot][i] = __builtin_nontemporal_load(row + i);
        }
    };
#pragma unroll
    for (u32 p = 0; p < PF; p++) fetch(p, w + p * W);
    for (u32 K2b = w; K2b < nK2; K2b += W * PF) {
#pragma unroll
        for (u32 p = 0; p < PF; p++) {
            const u32 K2 = K2b + p * W;
            if (K2 >= nK2) break;
            u32 d[R];
#pragma unroll
            for (int i = 0; i < R; i++) d[i] = dq[p][i];
            fetch(p, K2 + W * PF);
#pragma unroll
            for (u32 a3 = 0; a3 < 2; a3++) {
                const u32 u = unit_of<R>(d, 2u * a3 + a4);
                const u32 un = (u32)__shfl((int)u, (int)((l & 32u) | ((s + 1u) & 31u)), 64);
                const u64 comb = ((u64)u << (8 * R)) | (u64)un;
                const u32 comb32 = (u << 16) | un;  // R == 2: the whole window in one register
                const uint16_t *xk = xs + 32u * K2 + 16u * a3 + 2u * b;
                const u32 x0 = *reinterpret_cast<const u32 *>(xk), x1 = *reinterpret_cast<const u32 *>(xk + 8);
#pragma unroll
                for (u32 i = 0; i < 4; i++) {
                    // state -> st * (st + 1) (24-bit multiply-add: st < 2^16) -> codebook word at byte (idx >> 4) & 0x7FC,
                    // sign of the low half folded in with one 3-input op
                    const u32 st = R == 2 ? __builtin_amdgcn_ubfe(comb32, 16u - 4u * i, 16u)
                                          : (u32)(comb >> (16 * R - 2 * R * i - 16)) & 0xFFFFu;
                    const u32 idx = __umul24(st, st) + st;
                    const u32 w2 = *reinterpret_cast<const u32 *>(reinterpret_cast<const unsigned char *>(tl) + ((idx >> 4) & 0x7FCu)) ^ (idx & 0x8000u);
                    const u32 xv = (i & 2u) ? x1 : x0;  // cc = i / 2
                    if (i & 1u)                        // d = i % 2
                        acc1 = __builtin_amdgcn_fdot2(__builtin_bit_cast(h16x2, w2), __builtin_bit_cast(h16x2, xv), acc1, false);
                    else
                        acc0 = __builtin_amdgcn_fdot2(__builtin_bit_cast(h16x2, w2), __builtin_bit_cast(h16x2, xv), acc0, false);
                }
            }
        }
    }
    // sum over the 4 lanes b = 0..3 of a row (one quad)
    acc0 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, acc0), 0xB1, 0xF, 0xF, false));
    acc0 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, acc0), 0x4E, 0xF, 0xF, false));
    acc1 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, acc1), 0xB1, 0xF, 0xF, false));
    acc1 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, acc1), 0x4E, 0xF, 0xF, false));
    if (b == 0) {
        part[w * 32u + a4 * 16u + a] = acc0;
        part[w * 32u + a4 * 16u + a + 8u] = acc1;
    }
    __syncthreads();
    if (tid < 32u) {
        float t = 0.f;
        for (u32 i = 0; i < W; i++) t += part[i * 32u + tid];
        out[M2 * 32u + tid] = t;
    }
}

// ------------------------------------------------------------------------------------------------ Hadamard (FWHT)
__global__ void __launch_bounds__(1024) fwht_kernel(const float *x, float *y, u32 n, float scale) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *v = reinterpret_cast<float *>(smem);
    const u32 T = blockDim.x, tid = threadIdx.x;
    const float *xr = x + (size_t)blockIdx.x * n;
    float *yr = y + (size_t)blockIdx.x * n;
    for (u32 i = tid; i < n; i += T) v[i] = xr[i];
    __syncthreads();
    for (u32 h = 1; h < n; h <<= 1) {
        for (u32 p = tid; p < n / 2u; p += T) {
            const u32 j = (p / h) * 2u * h + (p % h);
            const float a = v[j], b = v[j + h];
            v[j] = a + b;
            v[j + h] = a - b;
        }
        __syncthreads();
    }
    for (u32 i = tid; i < n; i += T) yr[i] = v[i] * scale;
}
}  // namespace

extern "C" int gq_qtip_matvec(float *out, const uint32_t *compressed, const void *x, const void *codebook, uint32_t M,
                              uint32_t K, int R, void *stream) {
    if (!out || !compressed || !x || !codebook) return gq_fail(GQ_EINVAL, "null pointer argument.");
    if (R < 2 || R > 4) return gq_fail(GQ_ENOTSUP, "R (bits per weight) must be 2, 3 or 4 (kernel_check.py:1-14).");
    if (M == 0 || K == 0 || M % 32u || K % 32u) return gq_fail(GQ_EINVAL, "M and K must be positive multiples of 32.");
    if (((uintptr_t)x | (uintptr_t)compressed | (uintptr_t)codebook) & 15u) return gq_fail(GQ_EINVAL, "buffers must be 16-byte aligned.");
    const u32 nK2 = K / 32u;
    u32 waves = nK2 >= 8u ? 8u : (nK2 >= 4u ? 4u : (nK2 >= 2u ? 2u : 1u));
    // fewer 32-row bands than CUs (e.g. M = 4096: 128 blocks): 16 waves per band shorten the per-wave decode loop
    if (nK2 >= 32u && M / 32u <= 256u) waves = 16u;
    const size_t smem = (size_t)K * 2u + (size_t)waves * 32u * 4u;  // + 2 KiB static codebook
    if (smem > 160u * 1024u) return gq_fail(GQ_ENOTSUP, "K too large.");
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(M / 32u), block(waves * 64u);
#define GQ_LAUNCH_QTIP(RR)                                                                                            \
    do {                                                                                                              \
        static bool set = false;                                                                                      \
        if (smem > 48u * 1024u && !set) {                                                                             \
            GQ_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(qtip_matvec_kernel<RR>),                  \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                \
            set = true;                                                                                               \
        }                                                                                                             \
        hipLaunchKernelGGL(qtip_matvec_kernel<RR>, grid, block, smem, s, out, compressed, (const uint16_t *)x,         \
                           (const uint16_t *)codebook, M, K);                                                         \
    } while (0)
    if (R == 2) GQ_LAUNCH_QTIP(2);
    else if (R == 3) GQ_LAUNCH_QTIP(3);
    else GQ_LAUNCH_QTIP(4);
#undef GQ_LAUNCH_QTIP
    GQ_HIP_CHECK(hipGetLastError());
    return GQ_OK;
}

extern "C" int gq_hadamard(const float *x, float *y, uint32_t rows, uint32_t n, float scale, void *stream) {
    if (!x || !y) return gq_fail(GQ_EINVAL, "null pointer argument.");
    if (n == 0 || (n & (n - 1u))) return gq_fail(GQ_EINVAL, "hadamard: the last dimension must be a power of two.");
    if (rows == 0) return GQ_OK;
    const size_t smem = (size_t)n * 4u;
    if (smem > 160u * 1024u) return gq_fail(GQ_ENOTSUP, "hadamard: n too large (<= 32768).");
    static bool set = false;
    if (smem > 48u * 1024u && !set) {
        GQ_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(fwht_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         160 * 1024));
        set = true;
    }
    const u32 T = n / 2u >= 1024u ? 1024u : (n / 2u >= 64u ? n / 2u : 64u);
    hipLaunchKernelGGL(fwht_kernel, dim3(rows), dim3(T), smem, (hipStream_t)stream, x, y, n, scale);
    GQ_HIP_CHECK(hipGetLastError());
    return GQ_OK;
}
