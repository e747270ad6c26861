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
//   stages h = 8 .. 256  : partner elements sit in lanes t ^ 1, t ^ 2, .. t ^ 32: cross-lane moves (DPP, v_permlane*_swap), no memory;
//   stages h >= 512      : three at a time on 8 registers per work item, element stride h0 >= 64: consecutive lanes touch
//                          consecutive words (a stride of 8 words, as plain radix-8 passes have it at h0 = 1 and 8,
//                          serialises 8 ways on the 64 banks: measured 7 us for n = 8192).
// P: transform length; the n floats are n / P independent consecutive segments (P = n: one transform)
// fwht_lds = fwht_first_pass (the register / cross-lane stages; returns the element stride of the next stage) + fwht_rest.
// E elements per work item: 8 (n / 8 work items), or 4 where that would leave threads idle (n / 8 < blockDim: a 4096-point transform
// on 1024 threads -- these passes are bound by the latency of a wave's dependent instructions, not by their number: twice the waves
// with half the chain each is nearly twice as fast).  Work items: whole waves, or the first n / E lanes of wave 0 (closed under t ^ m).
// the stages of a work item's E consecutive elements that need no memory: the in-register ones (strides 1 .. E / 2) and the
// cross-lane ones (strides E .. E mmax; work item t = the lane, partners t ^ m in the same wave)
template <u32 E>
__device__ __forceinline__ void fwht_item_stages(float (&r)[E], u32 t, u32 mmax) {
#pragma unroll
    for (u32 st = 1; st < E; st <<= 1)
#pragma unroll
        for (u32 k = 0; k < E; k++)
            if (!(k & st)) {
                const float x0 = r[k], x1 = r[k | st];
                r[k] = x0 + x1;
                r[k | st] = x0 - x1;
            }
#pragma unroll
    for (u32 m = 1; m <= 32u; m <<= 1) {  // element stride E m: the partner thread is t ^ m (same wave)
        if (m > mmax) continue;  // (uniform over the block)
        // lower thread: r + other, upper thread: other - r  ==  other + (+-r): one sign flip (v_xor) and one add that takes
        // the partner's value through its DPP operand (v_add_f32_dpp; this file is built without the SLP vectoriser, which
        // packs the adds into v_pk_add_f32 and blocks that fold), instead of move / add / subtract / select
        const u32 smask = (t & m) ? 0x80000000u : 0u;
#pragma unroll
        for (u32 k = 0; k < E; k++) {
            // lane ^ 1, ^ 2: quad permutes; lane ^ 4 = half-row mirror (^ 7) of the quad reversal (^ 3); lane ^ 8: row_ror:8 -- DPP
            // moves on the VALU instead of ds_bpermute through the LDS crossbar.  Round 6: lane ^ 16 and lane ^ 32 by
            // v_permlane16_swap / v_permlane32_swap (gfx950): of the two registers they return, [0] holds the LOWER partner's
            // value and [1] the UPPER partner's in both lanes of a pair -- lower: [0] + [1], upper: [0] - [1], the same butterfly.
            // Six cross-lane stages per pass instead of three: a 4096-point transform is two or three short LDS passes.
            if (m <= 8u) {
                int o = __builtin_bit_cast(int, r[k]);
                if (m == 1u) o = __builtin_amdgcn_update_dpp(0, o, 0xB1, 0xF, 0xF, true);       // quad_perm [1,0,3,2]
                else if (m == 2u) o = __builtin_amdgcn_update_dpp(0, o, 0x4E, 0xF, 0xF, true);  // quad_perm [2,3,0,1]
                else if (m == 4u) {
                    o = __builtin_amdgcn_update_dpp(0, o, 0x1B, 0xF, 0xF, true);   // quad_perm [3,2,1,0]
                    o = __builtin_amdgcn_update_dpp(0, o, 0x141, 0xF, 0xF, true);  // row_half_mirror
                } else {
                    o = __builtin_amdgcn_update_dpp(0, o, 0x128, 0xF, 0xF, true);  // row_ror:8
                }
                r[k] = __builtin_bit_cast(float, o) + __builtin_bit_cast(float, __builtin_bit_cast(u32, r[k]) ^ smask);
            } else {
                const u32 b = __builtin_bit_cast(u32, r[k]);
                u32 lo_v, hi_v;
                if (m == 16u) {
                    auto sw = __builtin_amdgcn_permlane16_swap(b, b, false, false);
                    lo_v = sw[0], hi_v = sw[1];
                } else {
                    auto sw = __builtin_amdgcn_permlane32_swap(b, b, false, false);
                    lo_v = sw[0], hi_v = sw[1];
                }
                r[k] = __builtin_bit_cast(float, lo_v) + __builtin_bit_cast(float, hi_v ^ smask);
            }
        }
    }
}
template <u32 E>
__device__ __forceinline__ u32 fwht_mmax(u32 P) { return P >= 64u * E ? 32u : P / (2u * E); }  // cross-lane stages: element strides E .. E mmax (< P)
template <u32 E>
__device__ __forceinline__ u32 fwht_first_pass_e(float *v, u32 n, u32 P) {
    static_assert(E == 4u || E == 8u, "4 or 8 elements per work item");
    const u32 T = blockDim.x, tid = threadIdx.x;
    const u32 mmax = fwht_mmax<E>(P);
    for (u32 t = tid; t < n / E; t += T) {
        float r[E];
        if constexpr (E == 8u) {
            const float4 lo4 = *reinterpret_cast<const float4 *>(v + 8u * t), hi4 = *reinterpret_cast<const float4 *>(v + 8u * t + 4u);
            r[0] = lo4.x, r[1] = lo4.y, r[2] = lo4.z, r[3] = lo4.w, r[4] = hi4.x, r[5] = hi4.y, r[6] = hi4.z, r[7] = hi4.w;
        } else {
            const float4 q = *reinterpret_cast<const float4 *>(v + 4u * t);
            r[0] = q.x, r[1] = q.y, r[2] = q.z, r[3] = q.w;
        }
        fwht_item_stages<E>(r, t, mmax);
        if constexpr (E == 8u) {
            *reinterpret_cast<float4 *>(v + 8u * t) = make_float4(r[0], r[1], r[2], r[3]);
            *reinterpret_cast<float4 *>(v + 8u * t + 4u) = make_float4(r[4], r[5], r[6], r[7]);
        } else {
            *reinterpret_cast<float4 *>(v + 4u * t) = make_float4(r[0], r[1], r[2], r[3]);
        }
    }
    __syncthreads();
    return 2u * E * mmax;
}
__device__ __forceinline__ u32 fwht_first_pass(float *v, u32 n, u32 P) {
    if (P < 64u) return 1u;
    // (E = 4 with two-stage LDS passes behind it -- every thread of a 1024-thread block busy on a 4096-point transform -- was measured:
    // three passes of ~1,500 cycles instead of two of 2,200 + 1,400, Llama-2-7b QTIP decode 426 vs 442 tokens/s: a pass costs its
    // barrier and its LDS round trips, not its arithmetic.  Kept compilable, not used.)
    return fwht_first_pass_e<8u>(v, n, P);
}
__device__ __forceinline__ void fwht_rest(float *v, u32 n, u32 P, u32 h0) {
    const u32 T = blockDim.x, tid = threadIdx.x;
    while (h0 < P) {
        const u32 left = P >> (u32)__builtin_ctz(h0);  // 2^(stages left)
        const u32 lh0 = (u32)__builtin_ctz(h0);  // (h0 is a power of two: shifts, not the ~30 instructions of an integer division)
        if (left >= 8u) {
            for (u32 t = tid; t < n / 8u; t += T) {
                const u32 lo = t & (h0 - 1u), hi = t >> lh0;
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
            for (u32 t = tid; t < (n >> (rad == 4u ? 2u : 1u)); t += T) {
                const u32 lo = t & (h0 - 1u), hi = t >> lh0;
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

__device__ __forceinline__ void fwht_lds(float *v, u32 n, u32 P) { fwht_rest(v, n, P, fwht_first_pass(v, n, P)); }
__device__ __forceinline__ void fwht_lds(float *v, u32 n) { fwht_lds(v, n, n); }

}  // namespace gq_fwht
