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
#include <math.h>
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

// one 32-row band M2 of the matvec: xs (fp16 [K]) and the codebook tl are in LDS (barrier done by the caller), part is
// [waves][32] scratch; out[M2 * 32 .. + 31] is written by the first 32 threads
struct QtipNoop {
    __device__ __forceinline__ void operator()() const {}
};
// `between` runs behind the request of the first PF tile blocks (the caller's prologue: it does not depend on them)
template <int R, class F = QtipNoop>
__device__ __forceinline__ void qtip_band(float *out, const u32 *comp, const uint16_t *xs, const u32 *tl, float *part, u32 M2, u32 K,
                                          F between = F(), u32 k2lo = 0u, u32 k2hi = 0xFFFFFFFFu) {
    const u32 T = blockDim.x, tid = threadIdx.x, W = T >> 6, w = tid >> 6, l = tid & 63u;
    const u32 a4 = l >> 5, s = l & 31u;  // tile-row parity, stream unit
    const u32 a = s >> 2, b = s & 3u;
    const u32 nK2 = K / 32u;
    const unsigned char *band = reinterpret_cast<const unsigned char *>(comp) + (size_t)M2 * nK2 * 128u * R;
    float acc0 = 0.f, acc1 = 0.f;        // rows a and a + 8 of tile row 2*M2 + a4
    // The stream of a wave is a chain of 4R-byte-per-lane loads: with one load in flight per wave a CU has 4 KiB
    // outstanding and the kernel is latency-bound (measured 0.35 TB/s).  PF tile blocks are requested ahead.
    constexpr u32 PF = 8;
    u32 dq[PF][R];
    if (k2hi > nK2) k2hi = nK2;  // the K range of this block (split-K launches: a part of the band)
    auto fetch = [&](u32 slot, u32 K2) {
        if (K2 < k2hi) {
            const u32 *row = reinterpret_cast<const u32 *>(band + (size_t)K2 * 128u * R + (size_t)s * 4u * R);
#pragma unroll
            for (int i = 0; i < R; i++) dq[slot][i] = __builtin_nontemporal_load(row + i);
        }
    };
#pragma unroll
    for (u32 p = 0; p < PF; p++) fetch(p, k2lo + w + p * W);
    between();
    for (u32 K2b = k2lo + w; K2b < k2hi; K2b += W * PF) {
#pragma unroll
        for (u32 p = 0; p < PF; p++) {
            const u32 K2 = K2b + p * W;
            if (K2 >= k2hi) break;
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
                    // st * (st + 1) as ONE 24-bit multiply-add.  Written in asm: the compiler knows that only bits 6..15 of the
                    // result are used, drops the 16-bit mask of st and then needs a full-width v_mad_u64_u32 (quarter rate) for
                    // the windows that are wider than 24 bits; v_mad_u32_u24 ignores the upper bits by itself.
                    u32 idx;
                    asm("v_mad_u32_u24 %0, %1, %1, %1" : "=v"(idx) : "v"(st));
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

template <int R>
__global__ void __launch_bounds__(1024) qtip_matvec_kernel(float *out, const u32 *comp, const uint16_t *x, const uint16_t *tlut,
                                                          u32 M, u32 K) {
    // the codebook sits in a STATIC LDS array (address 0, known to the compiler: the lookup address needs no base add)
    __shared__ __attribute__((aligned(16))) u32 tl[512];         // [512] half2 codebook
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t *xs = reinterpret_cast<uint16_t *>(smem);          // [K]
    float *part = reinterpret_cast<float *>(xs + K);            // [waves][32 rows]
    const u32 T = blockDim.x, tid = threadIdx.x;
    for (u32 i = tid; i < 512u; i += T) tl[i] = reinterpret_cast<const u32 *>(tlut)[i];
    for (u32 i = tid; i < K / 8u; i += T) reinterpret_cast<uint4 *>(xs)[i] = reinterpret_cast<const uint4 *>(x)[i];
    __syncthreads();
    qtip_band<R>(out, comp, xs, tl, part, blockIdx.x, K);
}

// in-place Sylvester butterflies on n floats in LDS (n a power of two, barriers inside, one behind the last pass).
// Stages h = 1, 2, 4, .. in this order with (a + b, a - b) at (j, j + h) -- the arithmetic of a plain radix-2 loop, so every
// user (gq_hadamard, the fused linear) agrees bit for bit -- but scheduled for the LDS:
//   stages h = 1, 2, 4   : 8 consecutive elements per thread in registers (two 16-byte LDS reads);
//   stages h = 8, 16, 32 : partner elements sit in lanes t ^ 1, t ^ 2, t ^ 4: cross-lane moves, no memory;
//   stages h >= 64       : three at a time on 8 registers per work item, element stride h0 >= 64: consecutive lanes touch
//                          consecutive words (a stride of 8 words, as plain radix-8 passes have it at h0 = 1 and 8,
//                          serialises 8 ways on the 64 banks: measured 7 us for n = 8192).
// P: transform length; the n floats are n / P independent consecutive segments (P = n: one transform)
__device__ __forceinline__ void fwht_lds(float *v, u32 n, u32 P);
__device__ __forceinline__ void fwht_lds(float *v, u32 n) { fwht_lds(v, n, n); }
__device__ __forceinline__ void fwht_lds(float *v, u32 n, u32 P) {
    const u32 T = blockDim.x, tid = threadIdx.x;
    u32 h0 = 1;
    if (P >= 64u) {
        // n / 8 work items: whole waves, or the first n / 8 lanes of wave 0 (a set closed under t ^ 1, t ^ 2, t ^ 4)
        for (u32 t = tid; t < n / 8u; t += T) {
            float r[8];
            const float4 lo4 = *reinterpret_cast<const float4 *>(v + 8u * t), hi4 = *reinterpret_cast<const float4 *>(v + 8u * t + 4u);
            r[0] = lo4.x, r[1] = lo4.y, r[2] = lo4.z, r[3] = lo4.w, r[4] = hi4.x, r[5] = hi4.y, r[6] = hi4.z, r[7] = hi4.w;
#pragma unroll
            for (u32 st = 1; st < 8; st <<= 1)
#pragma unroll
                for (u32 k = 0; k < 8; k++)
                    if (!(k & st)) {
                        const float x0 = r[k], x1 = r[k | st];
                        r[k] = x0 + x1;
                        r[k | st] = x0 - x1;
                    }
#pragma unroll
            for (u32 m = 1; m < 8; m <<= 1) {  // element stride 8 m: the partner thread is t ^ m (same wave: 8 | 64)
                const bool upper = (t & m) != 0;
#pragma unroll
                for (u32 k = 0; k < 8; k++) {
                    // lane ^ 1, ^ 2: quad permutes; lane ^ 4 = half-row mirror (^ 7) of the quad reversal (^ 3): DPP moves on
                    // the VALU instead of 24 ds_bpermute per thread through the LDS crossbar
                    int o = __builtin_bit_cast(int, r[k]);
                    if (m == 1u) o = __builtin_amdgcn_update_dpp(o, o, 0xB1, 0xF, 0xF, false);       // quad_perm [1,0,3,2]
                    else if (m == 2u) o = __builtin_amdgcn_update_dpp(o, o, 0x4E, 0xF, 0xF, false);  // quad_perm [2,3,0,1]
                    else {
                        o = __builtin_amdgcn_update_dpp(o, o, 0x1B, 0xF, 0xF, false);   // quad_perm [3,2,1,0]
                        o = __builtin_amdgcn_update_dpp(o, o, 0x141, 0xF, 0xF, false);  // row_half_mirror
                    }
                    const float other = __builtin_bit_cast(float, o);
                    r[k] = upper ? other - r[k] : r[k] + other;
                }
            }
            *reinterpret_cast<float4 *>(v + 8u * t) = make_float4(r[0], r[1], r[2], r[3]);
            *reinterpret_cast<float4 *>(v + 8u * t + 4u) = make_float4(r[4], r[5], r[6], r[7]);
        }
        h0 = 64u;
        __syncthreads();
    }
    while (h0 < P) {
        const u32 left = P / h0;  // 2^(stages left)
        if (left >= 8u) {
            for (u32 t = tid; t < n / 8u; t += T) {
                const u32 lo = t & (h0 - 1u), hi = t / h0;
                float *b = v + hi * 8u * h0 + lo;
                float r[8];
#pragma unroll
                for (u32 k = 0; k < 8; k++) r[k] = b[k * h0];
#pragma unroll
                for (u32 st = 1; st < 8; st <<= 1)
#pragma unroll
                    for (u32 k = 0; k < 8; k++)
                        if (!(k & st)) {
                            const float x0 = r[k], x1 = r[k | st];
                            r[k] = x0 + x1;
                            r[k | st] = x0 - x1;
                        }
#pragma unroll
                for (u32 k = 0; k < 8; k++) b[k * h0] = r[k];
            }
            h0 *= 8u;
        } else {  // one or two stages left
            const u32 rad = left;  // 2 or 4
            for (u32 t = tid; t < n / rad; t += T) {
                const u32 lo = t & (h0 - 1u), hi = t / h0;
                float *b = v + hi * rad * h0 + lo;
                float r[4];
                r[0] = b[0], r[1] = b[h0];
                if (rad == 4u) {
                    r[2] = b[2u * h0], r[3] = b[3u * h0];
                    const float s0 = r[0] + r[1], d0 = r[0] - r[1], s1 = r[2] + r[3], d1 = r[2] - r[3];
                    b[0] = s0 + s1, b[h0] = d0 + d1, b[2u * h0] = s0 - s1, b[3u * h0] = d0 - d1;
                } else {
                    b[0] = r[0] + r[1], b[h0] = r[0] - r[1];
                }
            }
            h0 *= rad;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------ fused QTIP linear
// BitshiftLinear.forward (inference/lib/codebook/bitshift.py:415-472, eval, one row, no tensor-parallel Hadamard split):
//   x32 = x.float() * SU;  x = hadamard(x32) * n^-1/2 / 32;  y = decompress_matvec(trellis, x.half(), tlut)       [kernel A]
//   y = hadamard(y) * m^-1/2;  out = (y * (SV * 32)).half()  (+ the residual add of the block, model.py:311-313)       [kernel B]
// Kernel A fuses what produces x in the decode step -- RMSNorm (model.py:281-292) or silu(gate) * up (model.py:266) --
// and serves up to 3 linears that share the input (q/k/v, gate/up) in one launch; every 32-row band block repeats the
// K-point transform (K log K / 2 butterflies: small next to its 32 x K trellis decode).
struct QtipIn {
    const u32 *comp;
    const float *SU;
    const uint16_t *tlut;
    float *y32;
    u32 band0;  // first band (block / ksplit) of this linear
    u32 M;
};
struct QtipOut {
    const float *y32, *SV32;  // SV32 = SV * 32 (fp32)
    const uint16_t *resid;
    uint16_t *out;
    u32 M;
    float mscale;
    u32 parts;  // y32 is [parts][M] split-K partial sums, added in this order
};
struct QtipInArgs {
    const uint16_t *x, *x2, *normw;
    float eps, kscale;  // kscale = (float)K^-1/2, rounded from double like the scale argument of hadamard()
    u32 K, n;
    u32 ksplit;  // 1 / 2: blocks per band; block (band, ks) covers half of K and writes its sums to y32 + ks * M
    QtipIn lin[3];
    u32 nprev;          // 1 / 2: x (and x2) are the outputs of the linears prev[] whose transform-out is done here (M == K)
    QtipOut prev[2];
};
enum { QPRO_NONE = 0, QPRO_RMSNORM = 1, QPRO_SILUMUL = 2, QPRO_PRE = 3 };

template <int R, int PRO>
__global__ void __launch_bounds__(1024) qtip_linear_in_kernel(QtipInArgs a) {
    __shared__ __attribute__((aligned(16))) u32 tl[512];
    __shared__ float redf[17];
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const u32 K = a.K, T = blockDim.x, tid = threadIdx.x;
    float *v = reinterpret_cast<float *>(smem);                 // [K] fp32 transform buffer (none for a pre-transformed input)
    uint16_t *xs = reinterpret_cast<uint16_t *>(v + (PRO == QPRO_PRE ? 0u : K));  // [K] fp16 matvec input
    float *part = reinterpret_cast<float *>(xs + K);            // [waves][32]
    const u32 bid = blockIdx.x / a.ksplit, ks = blockIdx.x % a.ksplit;
    u32 li = 0;
    if (a.n > 1 && bid >= a.lin[1].band0) li = 1;
    if (a.n > 2 && bid >= a.lin[2].band0) li = 2;
    const QtipIn L = a.lin[li];
    // codebook words of this thread (T >= 256): requested now, stored to LDS in the prologue
    const u32 tl0 = reinterpret_cast<const u32 *>(L.tlut)[tid & 511u], tl1 = reinterpret_cast<const u32 *>(L.tlut)[(tid + 256u) & 511u];
    // Folded transform-out of the producing linear(s): what gq_qtip_linear_out would have written is rebuilt in LDS by
    // every block (same arithmetic, same order) and stored once, by block 0, for the kernels that need it later
    // (the residual stream).  One launch and one global round trip less per linear.
    const uint16_t *xg = a.x, *x2g = a.x2;
    if (a.nprev) {
        uint16_t *xp = reinterpret_cast<uint16_t *>(part + (T >> 6) * 32u);  // [nprev][K] fp16
        for (u32 r = 0; r < a.nprev; r++) {
            const QtipOut P = a.prev[r];
            for (u32 i = tid; i < K; i += T) v[i] = P.parts == 2u ? P.y32[i] + P.y32[K + i] : P.y32[i];
            __syncthreads();
            fwht_lds(v, K);
            for (u32 i = tid; i < K; i += T) {
                h16 y = (h16)gq_pin_f32((v[i] * P.mscale) * P.SV32[i]);
                if (P.resid) y = __builtin_bit_cast(h16, P.resid[i]) + y;
                xp[r * K + i] = __builtin_bit_cast(uint16_t, y);
                if (blockIdx.x == 0 && P.out) P.out[i] = __builtin_bit_cast(uint16_t, y);
            }
            __syncthreads();
        }
        xg = xp;
        x2g = xp + K;
    }
    // The input vectors are requested first (registers), the first tile blocks of the band behind them, and the prologue
    // runs while those are on their way from HBM (vector memory returns in order: requested the other way round, the
    // first use of x would wait for the tiles).
    constexpr u32 NU = 2;  // 8-element units per thread held in registers (K <= 16 T)
    const bool inreg = !a.nprev && K <= 8u * NU * T && PRO != QPRO_PRE && !(((uintptr_t)xg | (uintptr_t)x2g | (uintptr_t)a.normw | (uintptr_t)L.SU) & 15u);
    uint4 xq[NU], x2q[NU], nwq[NU];
    float4 su0[NU], su1[NU];
    if (inreg) {
#pragma unroll
        for (u32 k = 0; k < NU; k++) {
            const u32 u = tid + k * T;
            const bool ok = u < K / 8u;
            xq[k] = ok ? reinterpret_cast<const uint4 *>(xg)[u] : make_uint4(0u, 0u, 0u, 0u);
            su0[k] = ok ? reinterpret_cast<const float4 *>(L.SU)[2u * u] : make_float4(0.f, 0.f, 0.f, 0.f);
            su1[k] = ok ? reinterpret_cast<const float4 *>(L.SU)[2u * u + 1u] : make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (PRO == QPRO_RMSNORM) nwq[k] = ok ? reinterpret_cast<const uint4 *>(a.normw)[u] : make_uint4(0u, 0u, 0u, 0u);
            if constexpr (PRO == QPRO_SILUMUL) x2q[k] = ok ? reinterpret_cast<const uint4 *>(x2g)[u] : make_uint4(0u, 0u, 0u, 0u);
        }
    }
    auto prologue = [&]() {
        if (tid < 512u) tl[tid] = tl0;
        if (T == 256u) tl[tid + 256u] = tl1;
        if constexpr (PRO == QPRO_PRE) {  // x is the transformed fp16 input already (gq_qtip_transform): K need not be a power of two
            for (u32 i = tid; i < K / 8u; i += T) reinterpret_cast<uint4 *>(xs)[i] = reinterpret_cast<const uint4 *>(xg)[i];
            __syncthreads();
            return;
        }
        float nscale = 0.f;
        if constexpr (PRO == QPRO_RMSNORM) {
            float ss = 0.f;
            if (inreg) {
#pragma unroll
                for (u32 k = 0; k < NU; k++) {  // (units beyond K are zero)
                    const u32 w4[4] = {xq[k].x, xq[k].y, xq[k].z, xq[k].w};
#pragma unroll
                    for (int e = 0; e < 8; e++) {
                        const float f = (float)__builtin_bit_cast(h16, (uint16_t)(w4[e >> 1] >> (16 * (e & 1))));
                        ss += f * f;
                    }
                }
            } else {  // same elements per thread, same order (the two forms give the same sum bit for bit)
                for (u32 u = tid; u < K / 8u; u += T)
#pragma unroll
                    for (u32 e = 0; e < 8; e++) {
                        const float f = (float)__builtin_bit_cast(h16, xg[8u * u + e]);
                        ss += f * f;
                    }
            }
            for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
            if ((tid & 63u) == 0) redf[tid >> 6] = ss;
            __syncthreads();
            if (tid == 0) {
                float t = 0.f;
                for (u32 i = 0; i < (T >> 6); i++) t += redf[i];
                redf[16] = 1.0f / sqrtf(t / (float)K + a.eps);
            }
            __syncthreads();
            nscale = redf[16];
        }
        auto elem = [&](uint16_t xb, uint16_t x2b, uint16_t nwb, float su) {
            h16 xh = __builtin_bit_cast(h16, xb);
            if constexpr (PRO == QPRO_RMSNORM) xh = (h16)gq_pin_f32((float)xh * nscale) * __builtin_bit_cast(h16, nwb);
            if constexpr (PRO == QPRO_SILUMUL) {
                const float g = (float)xh;
                xh = (h16)(g / (1.0f + __expf(-g))) * __builtin_bit_cast(h16, x2b);
            }
            return (float)xh * su;
        };
        if (inreg) {
#pragma unroll
            for (u32 k = 0; k < NU; k++) {
                const u32 u = tid + k * T;
                if (u < K / 8u) {
                    const u32 xw[4] = {xq[k].x, xq[k].y, xq[k].z, xq[k].w};
                    const u32 x2w[4] = {PRO == QPRO_SILUMUL ? x2q[k].x : 0u, PRO == QPRO_SILUMUL ? x2q[k].y : 0u, PRO == QPRO_SILUMUL ? x2q[k].z : 0u,
                                        PRO == QPRO_SILUMUL ? x2q[k].w : 0u};
                    const u32 nw[4] = {PRO == QPRO_RMSNORM ? nwq[k].x : 0u, PRO == QPRO_RMSNORM ? nwq[k].y : 0u, PRO == QPRO_RMSNORM ? nwq[k].z : 0u,
                                       PRO == QPRO_RMSNORM ? nwq[k].w : 0u};
                    const float su[8] = {su0[k].x, su0[k].y, su0[k].z, su0[k].w, su1[k].x, su1[k].y, su1[k].z, su1[k].w};
                    float o[8];
#pragma unroll
                    for (int e = 0; e < 8; e++)
                        o[e] = elem((uint16_t)(xw[e >> 1] >> (16 * (e & 1))), (uint16_t)(x2w[e >> 1] >> (16 * (e & 1))),
                                    (uint16_t)(nw[e >> 1] >> (16 * (e & 1))), su[e]);
                    reinterpret_cast<float4 *>(v)[2u * u] = make_float4(o[0], o[1], o[2], o[3]);
                    reinterpret_cast<float4 *>(v)[2u * u + 1u] = make_float4(o[4], o[5], o[6], o[7]);
                }
            }
        } else {
            for (u32 i = tid; i < K; i += T)
                v[i] = elem(xg[i], PRO == QPRO_SILUMUL ? x2g[i] : (uint16_t)0, PRO == QPRO_RMSNORM ? a.normw[i] : (uint16_t)0, L.SU[i]);
        }
        __syncthreads();
        fwht_lds(v, K);
        const float sc = a.kscale;
        for (u32 u = tid; u < K / 8u; u += T) {  // 8 values per step: two 16-byte LDS reads, one 16-byte write
            const float4 f0 = reinterpret_cast<const float4 *>(v)[2u * u], f1 = reinterpret_cast<const float4 *>(v)[2u * u + 1u];
            const float f[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
            u32 o[4];
#pragma unroll
            for (int e = 0; e < 4; e++)
                o[e] = (u32)__builtin_bit_cast(uint16_t, (h16)((f[2 * e] * sc) / 32.0f)) |
                       ((u32)__builtin_bit_cast(uint16_t, (h16)((f[2 * e + 1] * sc) / 32.0f)) << 16);
            reinterpret_cast<uint4 *>(xs)[u] = make_uint4(o[0], o[1], o[2], o[3]);
        }
        __syncthreads();
    };
    const u32 nK2 = K / 32u, khalf = (nK2 + a.ksplit - 1u) / a.ksplit;
    qtip_band<R>(L.y32 + (size_t)ks * L.M, L.comp, xs, tl, part, bid - L.band0, K, prologue, ks * khalf, (ks + 1u) * khalf);
}

struct QtipOutArgs {
    QtipOut lin[3];
};
__global__ void __launch_bounds__(1024) qtip_linear_out_kernel(QtipOutArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *v = reinterpret_cast<float *>(smem);
    const QtipOut L = a.lin[blockIdx.x];
    const u32 T = blockDim.x, tid = threadIdx.x, M = L.M;
    // 4 consecutive outputs per thread and step: 16-byte loads of the sums, the scales and (8 bytes) the residual, all
    // requested before the transform (one block: nothing else hides their latency)
    constexpr u32 PRE = 2;  // M <= 8192 at 1024 threads
    const bool vec = !(((uintptr_t)L.y32 | (uintptr_t)L.SV32) & 15u) && !(((uintptr_t)L.resid | (uintptr_t)L.out) & 7u);
    const bool pre = vec && M <= 4u * PRE * T;
    float4 svr[PRE];
    uint2 rsr[PRE];
    if (pre) {
#pragma unroll
        for (u32 k = 0; k < PRE; k++) {
            const u32 u = tid + k * T;
            const bool ok = u < M / 4u;
            float4 y = ok ? reinterpret_cast<const float4 *>(L.y32)[u] : make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok && L.parts == 2u) {
                const float4 y2 = reinterpret_cast<const float4 *>(L.y32 + M)[u];
                y = make_float4(y.x + y2.x, y.y + y2.y, y.z + y2.z, y.w + y2.w);
            }
            if (ok) reinterpret_cast<float4 *>(v)[u] = y;
            svr[k] = ok ? reinterpret_cast<const float4 *>(L.SV32)[u] : make_float4(0.f, 0.f, 0.f, 0.f);
            rsr[k] = (ok && L.resid) ? reinterpret_cast<const uint2 *>(L.resid)[u] : make_uint2(0u, 0u);
        }
    } else {
        for (u32 i = tid; i < M; i += T) v[i] = L.parts == 2u ? L.y32[i] + L.y32[M + i] : L.y32[i];
    }
    __syncthreads();
    fwht_lds(v, M);
    const float sc = L.mscale;
    if (pre) {
#pragma unroll
        for (u32 k = 0; k < PRE; k++) {
            const u32 u = tid + k * T;
            if (u < M / 4u) {
                const float4 f = reinterpret_cast<const float4 *>(v)[u];
                const float fv[4] = {f.x, f.y, f.z, f.w}, sv[4] = {svr[k].x, svr[k].y, svr[k].z, svr[k].w};
                const u32 rw[2] = {rsr[k].x, rsr[k].y};
                uint16_t o[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    h16 y = (h16)gq_pin_f32((fv[e] * sc) * sv[e]);
                    if (L.resid) y = __builtin_bit_cast(h16, (uint16_t)(rw[e >> 1] >> (16 * (e & 1)))) + y;
                    o[e] = __builtin_bit_cast(uint16_t, y);
                }
                reinterpret_cast<uint2 *>(L.out)[u] = make_uint2((u32)o[0] | ((u32)o[1] << 16), (u32)o[2] | ((u32)o[3] << 16));
            }
        }
        return;
    }
    for (u32 i = tid; i < M; i += T) {
        h16 y = (h16)gq_pin_f32((v[i] * sc) * L.SV32[i]);
        if (L.resid) y = __builtin_bit_cast(h16, L.resid[i]) + y;
        L.out[i] = __builtin_bit_cast(uint16_t, y);
    }
}

// ------------------------------------------------------------------------------------------------ transform with a Hadamard factor
// Widths n = Kf * P with a non-power-of-two Hadamard factor Kf (matmul_had.py:13-67; Llama-2's 11008 = 172 * 64): the
// transform is the P-point Sylvester transform of every row of the [Kf][P] view followed by hadK @ (or hadK^T @) over
// the rows (matmul_hadU / matmul_hadUt, matmul_had.py:69-94).  The Kf x Kf product is too much work to repeat in every
// band block of the matvec, so these widths run transform -> gq_qtip_matvec -> transform: this kernel is either side.
// Block b owns RB rows of the result; every block loads the vector and does the (cheap) row transforms itself.
//   IN : v = pro(x) * SU            -> rows, hadK^T -> out16 = half(val * n^-1/2 / 32)          (input of the matvec)
//   OUT: v = y32                    -> rows, hadK   -> out16 = half(val * n^-1/2 * SV32) (+ resid)
struct QtipXfLin {
    const float *y32, *vec, *hadK;  // vec = SU (IN) or SV * 32 (OUT); hadK fp32 [Kf][Kf]
    const uint16_t *resid;
    uint16_t *out;
};
struct QtipXfArgs {
    const uint16_t *x, *x2, *normw;
    QtipXfLin lin[3];   // blockIdx.y (linears that share the source / the launch)
    float eps, nscale;  // nscale = (float)n^-1/2
    u32 n, Kf, P, RB, in, pro, transpose;
};
__global__ void __launch_bounds__(1024) qtip_transform_kernel(QtipXfArgs a) {
    __shared__ float redf[17];
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const u32 n = a.n, T = blockDim.x, tid = threadIdx.x;
    const u32 P = a.P, Kf = a.Kf, RB = a.RB, r0 = blockIdx.x * RB;
    float *v = reinterpret_cast<float *>(smem);  // [n]
    float *hs = v + n;                           // [Kf][4]: element k of the (up to 4) factor rows this block multiplies with
    float *ps = hs + 4u * Kf;                    // [KQ][RB][P] partial sums of the factor product
    const QtipXfLin L = a.lin[blockIdx.y];
    // this block's rows of hadK (columns for the transposed product), requested first; k-major so that one 16-byte LDS read
    // serves the 4 rows
    for (u32 e = tid; e < 4u * Kf; e += T) {
        const u32 k = e >> 2, rr = e & 3u, kr = r0 + rr;
        hs[e] = (rr < RB && kr < Kf) ? (a.transpose ? L.hadK[(size_t)k * Kf + kr] : L.hadK[(size_t)kr * Kf + k]) : 0.f;
    }
    if (a.in) {
        float rs = 0.f;
        if (a.pro == QPRO_RMSNORM) {
            float ss = 0.f;
            for (u32 i = tid; i < n; i += T) {
                const float f = (float)__builtin_bit_cast(h16, a.x[i]);
                ss += f * f;
            }
            for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
            if ((tid & 63u) == 0) redf[tid >> 6] = ss;
            __syncthreads();
            if (tid == 0) {
                float t = 0.f;
                for (u32 i = 0; i < (T >> 6); i++) t += redf[i];
                redf[16] = 1.0f / sqrtf(t / (float)n + a.eps);
            }
            __syncthreads();
            rs = redf[16];
        }
        for (u32 u = tid; u < n / 8u; u += T) {  // 8 activations per 16-byte load
            const uint4 xq = reinterpret_cast<const uint4 *>(a.x)[u];
            uint4 x2q = make_uint4(0u, 0u, 0u, 0u), nwq = x2q;
            if (a.pro == QPRO_SILUMUL) x2q = reinterpret_cast<const uint4 *>(a.x2)[u];
            if (a.pro == QPRO_RMSNORM) nwq = reinterpret_cast<const uint4 *>(a.normw)[u];
            const float4 s0 = reinterpret_cast<const float4 *>(L.vec)[2u * u], s1 = reinterpret_cast<const float4 *>(L.vec)[2u * u + 1u];
            const u32 xw[4] = {xq.x, xq.y, xq.z, xq.w}, x2w[4] = {x2q.x, x2q.y, x2q.z, x2q.w}, nw[4] = {nwq.x, nwq.y, nwq.z, nwq.w};
            const float su[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; e++) {
                h16 xh = __builtin_bit_cast(h16, (uint16_t)(xw[e >> 1] >> (16 * (e & 1))));
                if (a.pro == QPRO_RMSNORM)
                    xh = (h16)gq_pin_f32((float)xh * rs) * __builtin_bit_cast(h16, (uint16_t)(nw[e >> 1] >> (16 * (e & 1))));
                if (a.pro == QPRO_SILUMUL) {
                    const float g = (float)xh;
                    xh = (h16)(g / (1.0f + __expf(-g))) * __builtin_bit_cast(h16, (uint16_t)(x2w[e >> 1] >> (16 * (e & 1))));
                }
                o[e] = (float)xh * su[e];
            }
            reinterpret_cast<float4 *>(v)[2u * u] = make_float4(o[0], o[1], o[2], o[3]);
            reinterpret_cast<float4 *>(v)[2u * u + 1u] = make_float4(o[4], o[5], o[6], o[7]);
        }
    } else {
        for (u32 i = tid; i < n / 4u; i += T) reinterpret_cast<float4 *>(v)[i] = reinterpret_cast<const float4 *>(L.y32)[i];
    }
    __syncthreads();
    fwht_lds(v, n, P);  // every row of the [Kf][P] view
    // (hadamard() scales by n^-1/2 before the factor product, matmul_had.py:88-90; here the factor entries are +-1, the
    // scale is applied to the sum -- one rounding less, one pass over the vector less)
    // factor product: thread (p, kq) adds its share of the k range into the (up to 4) rows of column p -- one LDS read of
    // the activation and one 16-byte read of the 4 factor entries per 4 multiply-adds --; the KQ partial sums are added in
    // a fixed order
    const u32 KQ = T >= P ? T / P : 1u, kper = (Kf + KQ - 1u) / KQ, NO = RB * P;
    for (u32 pb = 0; pb < P; pb += T) {
        const u32 p = pb + (T >= P ? tid % P : tid), kq = T >= P ? tid / P : 0u;
        if (p < P && kq < KQ) {
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            const u32 k1 = min((kq + 1u) * kper, Kf);
            for (u32 k = kq * kper; k < k1; k++) {
                const float4 h4 = *reinterpret_cast<const float4 *>(hs + 4u * k);
                const float xv = v[k * P + p];
                acc[0] += h4.x * xv;
                acc[1] += h4.y * xv;
                acc[2] += h4.z * xv;
                acc[3] += h4.w * xv;
            }
            for (u32 rr = 0; rr < RB; rr++) ps[(kq * RB + rr) * P + p] = acc[rr];
        }
    }
    __syncthreads();
    for (u32 o = tid; o < NO; o += T) {
        const u32 rr = o / P, kr = r0 + rr, p = o % P;
        if (kr >= Kf) continue;
        float acc = 0.f;
        for (u32 q = 0; q < KQ; q++) acc += ps[(q * RB + rr) * P + p];
        acc *= a.nscale;
        const u32 i = kr * P + p;
        h16 y;
        if (a.in) y = (h16)(acc / 32.0f);
        else {
            y = (h16)gq_pin_f32(acc * L.vec[i]);
            if (L.resid) y = __builtin_bit_cast(h16, L.resid[i]) + y;
        }
        L.out[i] = __builtin_bit_cast(uint16_t, y);
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
    fwht_lds(v, n);
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
        static GqPerDeviceOnce once;                                                                                      \
        if (once.first_use()) {                                                                             \
            GQ_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(qtip_matvec_kernel<RR>),                  \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024)); /* + 2 KiB static */ \
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

namespace {
bool pow2(u32 n) { return n && !(n & (n - 1u)); }
}

extern "C" int gq_qtip_linear_in(const void *x, const void *x2, const void *norm_weight, float eps, int prologue, uint32_t K, int R,
                                 int n, const GqQtipIn *lin, int n_prev, const GqQtipOut *prev, int ksplit, void *stream) {
    if (ksplit < 1 || ksplit > 2) return gq_fail(GQ_EINVAL, "gq_qtip_linear_in: ksplit must be 1 or 2.");
    if (!lin || n < 1 || n > 3) return gq_fail(GQ_EINVAL, "gq_qtip_linear_in: 1..3 linears.");
    if (n_prev < 0 || n_prev > 2 || (n_prev && !prev)) return gq_fail(GQ_EINVAL, "gq_qtip_linear_in: 0..2 producing linears.");
    if (n_prev == 0 && !x) return gq_fail(GQ_EINVAL, "gq_qtip_linear_in: x is null.");
    if (n_prev && n_prev != (prologue == GQ_QPRO_SILU_MUL ? 2 : 1))
        return gq_fail(GQ_EINVAL, "gq_qtip_linear_in: one producing linear per input vector (two for SILU_MUL).");
    if (R < 2 || R > 4) return gq_fail(GQ_ENOTSUP, "R (bits per weight) must be 2, 3 or 4 (kernel_check.py:1-14).");
    if (prologue == GQ_QPRO_PRETRANSFORMED) {
        if (K == 0 || K % 32u || K > 32768u || n_prev) return gq_fail(GQ_EINVAL, "gq_qtip_linear_in: pre-transformed input: K a multiple of 32, no folding.");
    } else if (!pow2(K) || K < 32u || K > 16384u)
        return gq_fail(GQ_ENOTSUP, "fused QTIP linear: K must be a power of two in 32..16384.");
    if ((prologue == GQ_QPRO_RMSNORM && !norm_weight) || (prologue == GQ_QPRO_SILU_MUL && !x2 && !n_prev) || prologue < 0 || prologue > 3)
        return gq_fail(GQ_EINVAL, "gq_qtip_linear_in: prologue operand missing.");
    QtipInArgs a{};
    a.x = (const uint16_t *)x;
    a.x2 = (const uint16_t *)x2;
    a.normw = (const uint16_t *)norm_weight;
    a.eps = eps;
    a.kscale = (float)pow((double)K, -0.5);
    a.K = K;
    a.n = (u32)n;
    u32 bands = 0, minM = 0xFFFFFFFFu;
    for (int i = 0; i < n; i++) {
        if (!lin[i].trellis || (!lin[i].SU && prologue != GQ_QPRO_PRETRANSFORMED) || !lin[i].tlut || !lin[i].y32 || lin[i].M == 0 || lin[i].M % 32u)
            return gq_fail(GQ_EINVAL, "gq_qtip_linear_in: null pointer or M not a multiple of 32.");
        if (((uintptr_t)lin[i].trellis | (uintptr_t)lin[i].tlut) & 15u) return gq_fail(GQ_EINVAL, "buffers must be 16-byte aligned.");
        a.lin[i] = QtipIn{lin[i].trellis, lin[i].SU, (const uint16_t *)lin[i].tlut, lin[i].y32, bands, lin[i].M};
        bands += lin[i].M / 32u;
        if (lin[i].M < minM) minM = lin[i].M;
    }
    const u32 nK2 = K / 32u;
    u32 waves = nK2 >= 8u ? 8u : (nK2 >= 4u ? 4u : (nK2 >= 2u ? 2u : 1u));
    if (nK2 >= 32u && bands * (u32)ksplit <= 256u) waves = 16u;
    if (waves < 4u) waves = 4u;  // the transform wants threads
    a.ksplit = (u32)ksplit;
    a.nprev = (u32)n_prev;
    for (int i = 0; i < n_prev; i++) {
        if (!prev[i].y32 || !prev[i].SV32 || prev[i].M != K) return gq_fail(GQ_EINVAL, "gq_qtip_linear_in: producing linear must have M == K.");
        a.prev[i] = QtipOut{prev[i].y32, prev[i].SV32, (const uint16_t *)prev[i].resid, (uint16_t *)prev[i].out, K, (float)pow((double)K, -0.5),
                            prev[i].parts == 2u ? 2u : 1u};
    }
    const size_t smem = (size_t)K * (prologue == GQ_QPRO_PRETRANSFORMED ? 2u : 6u) + (size_t)waves * 32u * 4u + (size_t)n_prev * K * 2u;  // <= 130 KiB for K <= 16384
    if (smem > 150u * 1024u) return gq_fail(GQ_ENOTSUP, "gq_qtip_linear_in: K too large.");
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(bands * (u32)ksplit), block(waves * 64u);
#define GQ_LAUNCH_QIN(RR, PP)                                                                                         \
    do {                                                                                                              \
        static GqPerDeviceOnce once;                                                                                      \
        if (once.first_use()) {                                                                                                   \
            GQ_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(qtip_linear_in_kernel<RR, PP>),           \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024)); /* + static */ \
        }                                                                                                             \
        hipLaunchKernelGGL((qtip_linear_in_kernel<RR, PP>), grid, block, smem, s, a);                                 \
    } while (0)
#define GQ_LAUNCH_QIN_R(RR)                                                   \
    do {                                                                      \
        if (prologue == GQ_QPRO_RMSNORM) GQ_LAUNCH_QIN(RR, QPRO_RMSNORM);     \
        else if (prologue == GQ_QPRO_SILU_MUL) GQ_LAUNCH_QIN(RR, QPRO_SILUMUL); \
        else if (prologue == GQ_QPRO_PRETRANSFORMED) GQ_LAUNCH_QIN(RR, QPRO_PRE); \
        else GQ_LAUNCH_QIN(RR, QPRO_NONE);                                    \
    } while (0)
    if (R == 2) GQ_LAUNCH_QIN_R(2);
    else if (R == 3) GQ_LAUNCH_QIN_R(3);
    else GQ_LAUNCH_QIN_R(4);
#undef GQ_LAUNCH_QIN_R
#undef GQ_LAUNCH_QIN
    GQ_HIP_CHECK(hipGetLastError());
    return GQ_OK;
}

extern "C" int gq_qtip_linear_out(int n, const GqQtipOut *lin, void *stream) {
    if (!lin || n < 1 || n > 3) return gq_fail(GQ_EINVAL, "gq_qtip_linear_out: 1..3 linears.");
    QtipOutArgs a{};
    u32 maxM = 0;
    for (int i = 0; i < n; i++) {
        if (!lin[i].y32 || !lin[i].SV32 || !lin[i].out) return gq_fail(GQ_EINVAL, "gq_qtip_linear_out: null pointer argument.");
        if (!pow2(lin[i].M) || lin[i].M < 32u || lin[i].M > 32768u)
            return gq_fail(GQ_ENOTSUP, "fused QTIP linear: M must be a power of two in 32..32768.");
        if (lin[i].parts > 2u) return gq_fail(GQ_EINVAL, "gq_qtip_linear_out: parts must be 0 / 1 / 2.");
        a.lin[i] = QtipOut{lin[i].y32, lin[i].SV32, (const uint16_t *)lin[i].resid, (uint16_t *)lin[i].out, lin[i].M,
                           (float)pow((double)lin[i].M, -0.5), lin[i].parts == 2u ? 2u : 1u};
        if (lin[i].M > maxM) maxM = lin[i].M;
    }
    const size_t smem = (size_t)maxM * 4u;
    static GqPerDeviceOnce once;
    if (once.first_use()) {
        GQ_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(qtip_linear_out_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         160 * 1024));
    }
    hipLaunchKernelGGL(qtip_linear_out_kernel, dim3((u32)n), dim3(1024), smem, (hipStream_t)stream, a);
    GQ_HIP_CHECK(hipGetLastError());
    return GQ_OK;
}

extern "C" int gq_qtip_transform(int input_side, const void *x, const void *x2, const void *norm_weight, float eps, int prologue,
                                 int n_lin, const GqQtipXf *lin, uint32_t n, uint32_t Kf, int transpose, void *stream) {
    if (!lin || n_lin < 1 || n_lin > 3 || Kf < 2u || n == 0 || n % Kf) return gq_fail(GQ_EINVAL, "gq_qtip_transform: 1..3 linears; n a multiple of Kf.");
    const u32 P = n / Kf;
    if (!pow2(P) || P < 64u || n > 32768u) return gq_fail(GQ_ENOTSUP, "gq_qtip_transform: n / Kf must be a power of two >= 64, n <= 32768.");
    if (input_side && (!x || (prologue == GQ_QPRO_RMSNORM && !norm_weight) || (prologue == GQ_QPRO_SILU_MUL && !x2) || prologue < 0 || prologue > 2))
        return gq_fail(GQ_EINVAL, "gq_qtip_transform: source operand missing.");
    if (((uintptr_t)x | (uintptr_t)x2 | (uintptr_t)norm_weight) & 15u) return gq_fail(GQ_EINVAL, "gq_qtip_transform: buffers must be 16-byte aligned.");
    QtipXfArgs a{};
    a.x = (const uint16_t *)x;
    a.x2 = (const uint16_t *)x2;
    a.normw = (const uint16_t *)norm_weight;
    for (int i = 0; i < n_lin; i++) {
        if (((uintptr_t)lin[i].y32 | (uintptr_t)lin[i].vec) & 15u) return gq_fail(GQ_EINVAL, "gq_qtip_transform: buffers must be 16-byte aligned.");
        if (!lin[i].vec || !lin[i].out || !lin[i].hadK || (!input_side && !lin[i].y32)) return gq_fail(GQ_EINVAL, "gq_qtip_transform: null pointer argument.");
        a.lin[i] = QtipXfLin{lin[i].y32, lin[i].vec, lin[i].hadK, (const uint16_t *)lin[i].resid, (uint16_t *)lin[i].out};
    }
    a.eps = eps;
    a.nscale = (float)pow((double)n, -0.5);
    a.n = n;
    a.Kf = Kf;
    a.P = P;
    a.RB = (u32)gq_env_int("GQ_QTIP_XF_RB", P >= 256u ? 1 : (P >= 128u ? 2 : 4));  // rows of the result per block (<= 4)
    if (a.RB < 1u || a.RB > 4u) a.RB = 1u;
    a.in = input_side ? 1u : 0u;
    a.pro = input_side ? (u32)prologue : 0u;
    a.transpose = transpose ? 1u : 0u;
    const u32 KQ = P >= 1024u ? 1u : 1024u / P;
    const size_t smem = ((size_t)n + 4u * (size_t)Kf + (size_t)KQ * a.RB * P) * 4u;
    if (smem > 150u * 1024u) return gq_fail(GQ_ENOTSUP, "gq_qtip_transform: n too large.");
    static GqPerDeviceOnce once;
    if (once.first_use()) {
        GQ_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(qtip_transform_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         156 * 1024));
    }
    hipLaunchKernelGGL(qtip_transform_kernel, dim3((Kf + a.RB - 1u) / a.RB, (u32)n_lin), dim3(1024), smem, (hipStream_t)stream, a);
    GQ_HIP_CHECK(hipGetLastError());
    return GQ_OK;
}

extern "C" int gq_hadamard(const float *x, float *y, uint32_t rows, uint32_t n, float scale, void *stream) {
    if (!x || !y) return gq_fail(GQ_EINVAL, "null pointer argument.");
    if (n == 0 || (n & (n - 1u))) return gq_fail(GQ_EINVAL, "hadamard: the last dimension must be a power of two.");
    if (rows == 0) return GQ_OK;
    const size_t smem = (size_t)n * 4u;
    if (smem > 160u * 1024u) return gq_fail(GQ_ENOTSUP, "hadamard: n too large (<= 32768).");
    static GqPerDeviceOnce once;
    if (once.first_use()) {
        GQ_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(fwht_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         160 * 1024));
    }
    const u32 T = n / 2u >= 1024u ? 1024u : (n / 2u >= 64u ? n / 2u : 64u);
    hipLaunchKernelGGL(fwht_kernel, dim3(rows), dim3(T), smem, (hipStream_t)stream, x, y, n, scale);
    GQ_HIP_CHECK(hipGetLastError());
    return GQ_OK;
}
