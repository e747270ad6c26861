// fwht.h -- the in-LDS Sylvester (Walsh-Hadamard) butterflies shared by qtip.hip (fused QTIP linears, gq_hadamard) and decode.hip
// (transform-out of q / k / v inside the attention launch).  One arithmetic for every user: results agree bit for bit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gq_fwht {
typedef uint32_t u32;

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
                // lower thread: r + other, upper thread: other - r  ==  other + (+-r): one sign flip (v_xor) and one add that takes
                // the partner's value through its DPP operand (v_add_f32_dpp; this file is built without the SLP vectoriser, which
                // packs the adds into v_pk_add_f32 and blocks that fold), instead of move / add / subtract / select
                const u32 smask = (t & m) ? 0x80000000u : 0u;
#pragma unroll
                for (u32 k = 0; k < 8; k++) {
                    // lane ^ 1, ^ 2: quad permutes; lane ^ 4 = half-row mirror (^ 7) of the quad reversal (^ 3): DPP moves on
                    // the VALU instead of 24 ds_bpermute per thread through the LDS crossbar
                    int o = __builtin_bit_cast(int, r[k]);
                    if (m == 1u) o = __builtin_amdgcn_update_dpp(0, o, 0xB1, 0xF, 0xF, true);       // quad_perm [1,0,3,2]
                    else if (m == 2u) o = __builtin_amdgcn_update_dpp(0, o, 0x4E, 0xF, 0xF, true);  // quad_perm [2,3,0,1]
                    else {
                        o = __builtin_amdgcn_update_dpp(0, o, 0x1B, 0xF, 0xF, true);   // quad_perm [3,2,1,0]
                        o = __builtin_amdgcn_update_dpp(0, o, 0x141, 0xF, 0xF, true);  // row_half_mirror
                    }
                    r[k] = __builtin_bit_cast(float, o) + __builtin_bit_cast(float, __builtin_bit_cast(u32, r[k]) ^ smask);
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

}  // namespace gq_fwht
