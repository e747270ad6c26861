// decode.hip -- the non-quantized pieces of one bs=1 decode step, gfx950:
//   * token embedding lookup                      (inference/model.py:123 tok_embeddings)
//   * RoPE + KV-cache update + single-query attention  (inference/model.py:206-241, 336-341, 63-79)
//   * dense fp16 GEMV with optional RMSNorm prologue   (final norm + lm_head `output`, inference/model.py:93-94,128-129)
// All entry points read the token id / position from DEVICE memory so a captured hipGraph can be replayed for
// every token without re-recording.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "gq_internal.h"
#include "fwht.h"

namespace {

typedef uint32_t u32;
typedef _Float16 h16;
typedef h16 h16x2 __attribute__((ext_vector_type(2)));
typedef u32 u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float h2f(uint16_t h) { return (float)__builtin_bit_cast(h16, h); }
__device__ __forceinline__ h16 u2h(uint16_t h) { return __builtin_bit_cast(h16, h); }
__device__ __forceinline__ uint16_t h2u(h16 h) { return __builtin_bit_cast(uint16_t, h); }

template <bool MAX>
__device__ __forceinline__ float wave_reduce(float v) {
    auto comb = [](float a, float b) { return MAX ? fmaxf(a, b) : a + b; };
    v = comb(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false)));
    v = comb(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false)));
    v = comb(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false)));
    v = comb(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false)));
    // row_bcast15 -> rows 1,3 ; row_bcast31 -> rows 2,3 ; lanes not written keep `old` = identity for the combine
    const float ident = MAX ? -3.0e38f : 0.f;
    v = comb(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, ident), __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, false)));
    v = comb(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, ident), __builtin_bit_cast(int, v), 0x143, 0xC, 0xF, false)));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// KV group of query head h: h / (H / Hkv).  Both divisions sit in front of the first load of an attention launch (the cache base
// needs g) -- ~25 scalar / vector instructions each on this chip (no integer divide); the group size is a power of two for every
// Llama (1, 4, 8): shifts then, the general quotient otherwise.  (Scalar: H, Hkv are kernel arguments, h is wave-uniform.)
__device__ __forceinline__ u32 kv_group_of(u32 h, u32 H, u32 Hkv) {
    if (H == Hkv) return h;
    if (H == 4u * Hkv) return h >> 2;
    if (H == 8u * Hkv) return h >> 3;
    if (H == 2u * Hkv) return h >> 1;
    return h / (H / Hkv);
}

// ------------------------------------------------------------------------------------------------ embedding
__global__ void embed_kernel(const int *tok, const uint16_t *table, uint16_t *out, u32 D, u32 V) {
    u32 t = (u32)tok[0];
    if (t >= V) t = 0;
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < D / 8u; i += gridDim.x * blockDim.x)
        reinterpret_cast<uint4 *>(out)[i] = reinterpret_cast<const uint4 *>(table + (size_t)t * D)[i];
}

// the same with the statistics hand-over for the RMSNorm prologue of layer 0's first launch (include/gq_hip.h, GQ_SSQ_SLOTS): one
// block, thread t leaves the sum of squares of the 16-byte units it copied in slot t
__global__ void __launch_bounds__(GQ_SSQ_SLOTS) embed_ssq_kernel(const int *tok, const uint16_t *table, uint16_t *out, u32 D, u32 V, float *ssq) {
    u32 t = (u32)tok[0];
    if (t >= V) t = 0;
    float acc = 0.f;
    for (u32 i = threadIdx.x; i < D / 8u; i += (u32)GQ_SSQ_SLOTS) {
        const uint4 v = reinterpret_cast<const uint4 *>(table + (size_t)t * D)[i];
        gq_store_wt(reinterpret_cast<uint4 *>(out) + i, v);
        const u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float a = (float)__builtin_bit_cast(_Float16, (uint16_t)(w[k] & 0xFFFFu)), b = (float)__builtin_bit_cast(_Float16, (uint16_t)(w[k] >> 16));
            acc += a * a;
            acc += b * b;
        }
    }
    gq_store_wt(ssq + threadIdx.x, acc);
}

// ------------------------------------------------------------------------------------------------ attention
// One block per query head.  RoPE exactly as apply_rotary_pos_emb on fp16 tensors (inference/model.py:336-341):
//   q_embed = (q * cos) + (rotate_half(q) * sin)   -- three fp16-rounded operations, cos/sin are fp16 tables
// built by the host from fp32 (LlamaRotaryEmbedding.forward, model.py:379-405).  The rotated k and the v of this
// token are written to the cache at `pos` (KVCache.update, model.py:69-79) by the first head of each KV group.
// Scores / softmax / weighted sum run in fp32 from the fp16 operands; the output is rounded to fp16 once.
#ifndef GQ_ATTN_WAVES
#define GQ_ATTN_WAVES 8
#endif
constexpr int ATTN_WAVES = GQ_ATTN_WAVES;  // 8 waves x 4 positions x 4 in flight = 128 positions per pass (HD = 128)
// QT (QTIP models): q / k / v are not read as fp16 vectors but rebuilt from the trellis matvecs' fp32 sums -- the transform-out of
// BitshiftLinear.forward (inference/lib/codebook/bitshift.py:470: hadamard(y) * m^-1/2 * (SV * 32) -> fp16), i.e. what
// gq_qtip_linear_out computes in a launch of its own.  A head needs HD of the M outputs of each vector and the Sylvester matrix
// factors, H_M = H_(M / HD) (x) H_HD: combine the M / HD segments with the signs of the head's row first, then ONE HD-point
// transform.  (Keeping the order of the full transform -- segment-local stages of ALL segments first, then a tree over the
// segments: bit-identical -- was built and measured: every head block repeats 7/12 of the whole vector's butterflies with half
// the threads gq_qtip_linear_out has per vector, 366 vs 385 tokens/s.)
struct AttnQt {
    const float *y32[3], *sv32[3];  // q, k, v: sums [parts][M] and SV * 32 [M]
    u32 M[3], parts[3];
    float mscale[3];                // (float)M^-1/2
};
template <int HD, bool QT>
__global__ void __launch_bounds__(64 * ATTN_WAVES) attn_decode_kernel(const uint16_t *qkv, const int *pos_ptr, const uint16_t *cos_t,
                                                          const uint16_t *sin_t, uint16_t *kc, uint16_t *vc, uint16_t *out,
                                                          u32 H, u32 Hkv, u32 max_seq, float scale, u32 nsplit, float *ws, AttnQt qt) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr u32 NW = ATTN_WAVES;
    float *sc = reinterpret_cast<float *>(smem);  // [2 * NW * 64 / (HD / 8)] running max / sum of the position streams
    float *qs = sc + 2u * NW * (64u / (HD / 8u));  // [HD]
    float *kcur = qs + HD;                        // [HD]
    float *vcur = kcur + HD;                      // [HD]
    float *red = vcur + HD;                       // [4 * HD] (+ 2 * NW scratch at 4*HD)
    float *red2 = red + 4 * HD + 2 * NW;          // [NW waves * positions-per-wave-instruction][HD] partial outputs
    const u32 tid = threadIdx.x, w = tid >> 6, l = tid & 63u;
    const u32 h = blockIdx.x, g = kv_group_of(h, H, Hkv);
    const uint16_t *q = qkv + (size_t)h * HD;
    const uint16_t *k = qkv + (size_t)H * HD + (size_t)g * HD;
    const uint16_t *v = qkv + (size_t)(H + Hkv) * HD + (size_t)g * HD;
    if constexpr (QT) {
        // LDS behind the attention's own arrays: 3 x HD floats (the combined segments), then 3 x HD fp16 results
        float *tv = red2 + (size_t)NW * (64u / (HD / 8u)) * HD;
        uint16_t *res16 = reinterpret_cast<uint16_t *>(tv + 3 * HD);
        // H_M = H_(M / HD) (x) H_HD: the HD outputs of segment `seg` are the HD-point transform of  z[b] = sum_c s(c) y[c HD + b],
        // s(c) = (-1)^popcount(c & seg)  (row `seg` of the M / HD-point Sylvester matrix): 4096 signed additions and ONE short
        // transform instead of the whole vector's butterflies.  The additions are those of the full transform in another order
        // (segments first): the result equals gq_qtip_linear_out's up to fp32 rounding, not bit for bit.
        // every thread takes 16-byte units of the three vectors (all loads of a thread in flight together: one round trip to L2):
        // unit u of a vector = segment c = u / (HD / 4), elements 4 (u % (HD / 4)) ..; a thread's units of one vector share the
        // element group (T is a multiple of HD / 4), so it keeps one signed partial sum per vector; the T / (HD / 4) partial sums
        // of an element are added in a fixed order behind the barrier
        {
            constexpr u32 Q = HD / 4u;
            const u32 T = blockDim.x, G = T / Q;  // G thread groups per element group
            float *ps = tv + 3 * HD + 3 * HD / 2;  // [3][G][HD] partial sums (behind the fp16 results)
            float4 y[3][4];
            u32 cnt[3];
#pragma unroll
            for (u32 vi = 0; vi < 3; vi++) {
                const u32 M4 = qt.M[vi] / 4u;
                cnt[vi] = 0;
#pragma unroll
                for (u32 k = 0; k < 4; k++) {
                    const u32 u = tid + k * T;
                    y[vi][k] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (u < M4) {
                        float4 t = reinterpret_cast<const float4 *>(qt.y32[vi])[u];
                        for (u32 p = 1; p < qt.parts[vi]; p++) {  // split-K parts, ascending
                            const float4 t2 = reinterpret_cast<const float4 *>(qt.y32[vi])[(size_t)p * M4 + u];
                            t = make_float4(t.x + t2.x, t.y + t2.y, t.z + t2.z, t.w + t2.w);
                        }
                        y[vi][k] = t;
                    }
                }
            }
#pragma unroll
            for (u32 vi = 0; vi < 3; vi++) {
                const u32 seg = vi == 0u ? h : g;
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (u32 k = 0; k < 4; k++) {
                    const u32 c = (tid + k * T) / Q;
                    const float sg = (__builtin_popcount(c & seg) & 1) ? -1.f : 1.f;  // (units past the vector hold zeros)
                    acc = make_float4(acc.x + sg * y[vi][k].x, acc.y + sg * y[vi][k].y, acc.z + sg * y[vi][k].z, acc.w + sg * y[vi][k].w);
                }
                reinterpret_cast<float4 *>(ps + ((size_t)vi * G + tid / Q) * HD)[tid % Q] = acc;
            }
            __syncthreads();
            if (tid < 3u * HD) {
                const u32 vi = tid / HD, b = tid % HD;
                float z = 0.f;
                for (u32 gi = 0; gi < G; gi++) z += ps[((size_t)vi * G + gi) * HD + b];
                tv[tid] = z;
            }
        }
        __syncthreads();
        gq_fwht::fwht_lds(tv, 3u * HD, (u32)HD);  // the three HD-point transforms (barrier behind)
        if (tid < 3u * HD) {
            const u32 vi = tid / HD, b = tid % HD, seg = vi == 0u ? h : g;
            const h16 o = (h16)gq_pin_f32((tv[tid] * qt.mscale[vi]) * qt.sv32[vi][(size_t)seg * HD + b]);
            res16[tid] = h2u(o);
        }
        __syncthreads();
        q = res16, k = res16 + HD, v = res16 + 2 * HD;
    }
    // q / k / v do not depend on the position: their loads are issued before the position is read, so the two
    // dependent round trips (position -> cos / sin rows) overlap with them
    uint16_t qd_b = 0, kd_b = 0, qr_b = 0, kr_b = 0, vd_b = 0;
    if (tid < HD) {
        const u32 d = tid, dr = d < HD / 2 ? d + HD / 2 : d - HD / 2;
        qd_b = q[d];
        kd_b = k[d];
        qr_b = q[dr];
        kr_b = k[dr];
        vd_b = v[d];
    }
    u32 pos = (u32)pos_ptr[0];
    if (pos >= max_seq) {  // decoding past the cache: nothing is written to it, the head's output is poisoned (NaN logits)
        if (blockIdx.y == 0 && tid < HD) out[(size_t)h * HD + tid] = 0x7e00u;
        return;
    }
    uint16_t *kcg = kc + (size_t)g * max_seq * HD;
    uint16_t *vcg = vc + (size_t)g * max_seq * HD;
    // split-KV (long contexts): block (head, sp) takes the positions [p0, p1) of the pos + 1 cached ones, in whole passes
    // of the block (NW waves x 64 / (HD / 8) positions x 4 in flight); one head per block leaves all but H CUs idle and
    // is bound by what H CUs can stream (51 us at 4096 positions)
    constexpr u32 PASS = NW * (64u / (HD / 8u)) * 4u;
    const u32 sp = blockIdx.y;
    // a short context (up to two passes) is not worth splitting: split 0 does it all and writes the result itself, the
    // other blocks and the combine launch return at once
    const bool solo = nsplit > 1u && pos + 1u <= 2u * PASS;
    if (solo) {
        if (sp > 0u) return;
        nsplit = 1u;
    }
    const u32 per = nsplit > 1u ? (((pos + nsplit) / nsplit + PASS - 1u) / PASS) * PASS : pos + 1u;
    const u32 p0 = sp * per, p1 = min(pos + 1u, p0 + per);  // (p0 >= p1: nothing to do, a neutral partial result is written)
    const bool has_cur = p0 <= pos && pos < p1;             // the split that covers the current token
    if (tid < HD) {
        const u32 d = tid;
        const h16 c = u2h(cos_t[(size_t)pos * HD + d]), s = u2h(sin_t[(size_t)pos * HD + d]);
        const h16 qd = u2h(qd_b), kd = u2h(kd_b);
        const h16 qr = d < HD / 2 ? -u2h(qr_b) : u2h(qr_b);
        const h16 kr = d < HD / 2 ? -u2h(kr_b) : u2h(kr_b);
        const h16 qe = (h16)(qd * c) + (h16)(qr * s);
        const h16 ke = (h16)(kd * c) + (h16)(kr * s);
        qs[d] = (float)qe;
        kcur[d] = (float)ke;
        vcur[d] = h2f(vd_b);
        if (h == g * (H / Hkv) && has_cur) {  // (the first head of its KV group; off the critical path)
            kcg[(size_t)pos * HD + d] = h2u(ke);
            vcg[(size_t)pos * HD + d] = vd_b;
        }
    }
    __syncthreads();
    // One pass, online softmax per position stream.  16 lanes per position (8 dims = one 16-byte load per lane, a
    // position's K / V row is one coalesced 256-byte line pair), PPW positions per wave instruction, U independent
    // positions in flight per lane group; the K and the V rows of a batch are requested together (one memory round trip
    // instead of two), and each lane group keeps a running (max, sum, weighted V) that is rescaled when the max moves.
    // The NW * PPW streams of the block are combined once, through LDS.
    constexpr int LPP = HD / 8;          // lanes per position: 16 (HD=128) or 8 (HD=64)
    constexpr int PPW = 64 / LPP;        // positions per wave instruction
    constexpr int U = 4;
    const u32 sub = l / LPP, ld = l % LPP;
    float qreg[8];
#pragma unroll
    for (int e = 0; e < 8; e++) qreg[e] = qs[ld * 8 + e];
    float m_run = -3.0e38f, s_run = 0.f, acc[8];
#pragma unroll
    for (int e = 0; e < 8; e++) acc[e] = 0.f;
    for (u32 t0 = p0 + w * PPW * U; t0 < p1; t0 += NW * PPW * U) {
        uint4 kv[U], vv[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const u32 t = t0 + (u32)u * PPW + sub;
            const bool cached = t < pos && t < p1;
            kv[u] = cached ? *reinterpret_cast<const uint4 *>(kcg + (size_t)t * HD + ld * 8) : make_uint4(0, 0, 0, 0);
            vv[u] = cached ? *reinterpret_cast<const uint4 *>(vcg + (size_t)t * HD + ld * 8) : make_uint4(0, 0, 0, 0);
        }
        // the U positions of the batch share one rescale of the running (max, sum, weighted V): scores first, then one
        // exp for the old maximum and one per position
        float pu[U], vfu[U][8];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const u32 t = t0 + (u32)u * PPW + sub;
            const u32 kw[4] = {kv[u].x, kv[u].y, kv[u].z, kv[u].w}, vw[4] = {vv[u].x, vv[u].y, vv[u].z, vv[u].w};
            float p = 0.f;
            if (t == pos) {
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    p += qreg[e] * kcur[ld * 8 + e];
                    vfu[u][e] = vcur[ld * 8 + e];
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    p += qreg[2 * e] * h2f((uint16_t)(kw[e] & 0xFFFF));
                    p += qreg[2 * e + 1] * h2f((uint16_t)(kw[e] >> 16));
                    vfu[u][2 * e] = h2f((uint16_t)(vw[e] & 0xFFFF));
                    vfu[u][2 * e + 1] = h2f((uint16_t)(vw[e] >> 16));
                }
            }
            // sum over the LPP lanes of this position (xor butterflies inside a 16-lane DPP row): every lane gets the score
            p += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, p), 0xB1, 0xF, 0xF, false));
            p += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, p), 0x4E, 0xF, 0xF, false));
            p += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, p), 0x141, 0xF, 0xF, false));
            if (LPP == 16) p += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, p), 0x140, 0xF, 0xF, false));
            pu[u] = t < p1 ? p * scale : -3.0e38f;
        }
        float m_new = m_run;
#pragma unroll
        for (int u = 0; u < U; u++) m_new = fmaxf(m_new, pu[u]);
        const float resc = __expf(m_run - m_new);
        s_run *= resc;
#pragma unroll
        for (int e = 0; e < 8; e++) acc[e] *= resc;
#pragma unroll
        for (int u = 0; u < U; u++) {
            const float wgt = pu[u] > -2.0e38f ? __expf(pu[u] - m_new) : 0.f;
            s_run += wgt;
#pragma unroll
            for (int e = 0; e < 8; e++) acc[e] += wgt * vfu[u][e];
        }
        m_run = m_new;
    }
    // combine the streams: red2[stream][HD] weighted sums, sc[stream] = running max, sc[NS + stream] = running sum
    constexpr u32 NS = NW * PPW;
    const u32 stream = w * PPW + sub;
#pragma unroll
    for (int e = 0; e < 8; e++) red2[stream * HD + ld * 8 + e] = acc[e];
    if (ld == 0) {
        sc[stream] = m_run;
        sc[NS + stream] = s_run;
    }
    __syncthreads();
    if (tid < HD) {
        float M = -3.0e38f;
#pragma unroll
        for (u32 i = 0; i < NS; i++) M = fmaxf(M, sc[i]);
        float o = 0.f, sum = 0.f;
#pragma unroll
        for (u32 i = 0; i < NS; i++) {
            const float f = __expf(sc[i] - M);  // streams without a position: exp(-3e38 - M) = 0
            sum += sc[NS + i] * f;
            o += red2[i * HD + tid] * f;
        }
        if (nsplit > 1u) {  // partial result of this split: weighted sum and sum relative to its own maximum M
            float *wp = ws + ((size_t)h * nsplit + sp) * (HD + 2u);
            wp[tid] = o;
            if (tid == 0) {
                wp[HD] = M;
                wp[HD + 1] = sum;
            }
        } else {
            out[(size_t)h * HD + tid] = h2u((h16)(o / sum));
        }
    }
}

// ------------------------------------------------------------------------------------------------ attention, q / k rotated upstream
// Round 4: the wqkv GEMV of the decode step applies RoPE in its epilogue and writes k / v of the current token straight into the
// caches (ap_stream.hip, gq_anyprec_gemv_qkv_rope), so this launch starts at q k^T: no cos / sin round trip behind the position, no
// rotation, no LDS staging of the current token, no barrier in front of the position streams.  And the first batch of cached rows
// does not wait for the position either: the rows [0, PASS) are requested together with q and the position (one memory round trip
// instead of two); which of them are valid (t <= pos) is decided when they have landed.  Rows past the position hold whatever
// an earlier sequence left (finite or not): their scores are replaced and their V rows zeroed before use.
// Same arithmetic as attn_decode_kernel otherwise (fp32 scores / softmax / weighted sum from fp16 operands, one fp16 rounding).
#ifndef GQ_ATTN_SPEC
#define GQ_ATTN_SPEC 32  // cached rows requested before the position is known (a multiple of 16)
#endif
// QH = 1: one query head per block (grid n_head x n_split).  QH > 1 (grouped-query models, long caches): the QH query heads of a
// KV group share a block (grid n_head / QH x n_split), every cached row is loaded ONCE and multiplied with all of them -- at
// 4096 positions the per-head form reads each K / V row four times and waits a full memory round trip per 128-position pass
// (19 us per layer on Llama-3-8B); here the next pass is requested before the current one is multiplied.  Per head the
// arithmetic, the position -> stream assignment and the merge order are those of the QH = 1 form: bit-identical outputs.
// A short context (the `solo` rule) is finished by ONE block per head: block (hq, sp < QH) takes head hq * QH + sp alone.
template <int HD, int QH>
__global__ void __launch_bounds__(64 * ATTN_WAVES) attn_roped_kernel(const uint16_t *q, const int *pos_ptr, const uint16_t *kc, const uint16_t *vc,
                                                                     uint16_t *out, u32 H, u32 Hkv, u32 max_seq, float scale, u32 nsplit, float *ws) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr u32 NW = ATTN_WAVES;
    constexpr int LPP = HD / 8, PPW = 64 / LPP, U = 4;
    constexpr u32 NS = NW * PPW, PASS = NW * PPW * U;
    // (the merge and the NaN poisoning below index the block's QH * HD outputs by thread id: a build with fewer waves would silently
    // leave the upper heads of a group unwritten)
    static_assert(64 * ATTN_WAVES >= QH * HD, "attn_roped_kernel: the block must have a thread per output of its QH heads (GQ_ATTN_WAVES)");
    float *sc = reinterpret_cast<float *>(smem);  // [QH][2 * NS] running max / sum of the position streams
    float *red2 = sc + (size_t)QH * 2u * NS;      // [QH][NS][HD] partial outputs
    const u32 tid = threadIdx.x, w = tid >> 6, l = tid & 63u;
#ifdef GQ_ATTN_PROBE  // timing probe (tools/r5): blockIdx.y = identical copies of the whole-group form, every copy writes the same outputs
    const u32 h0 = blockIdx.x * QH, g = kv_group_of(h0, H, Hkv), sp = 0u;
#else
    const u32 h0 = blockIdx.x * QH, g = kv_group_of(h0, H, Hkv), sp = blockIdx.y;
#endif
    const u32 sub = l / LPP, ld = l % LPP;
    const uint16_t *kcg = kc + (size_t)g * max_seq * HD, *vcg = vc + (size_t)g * max_seq * HD;
    // requests that do not depend on the position: q, and (blocks that may start at row 0) the first batch of cached rows
    uint4 q4[QH];
#pragma unroll
    for (int qh = 0; qh < QH; qh++) q4[qh] = *reinterpret_cast<const uint4 *>(q + (size_t)(h0 + qh) * HD + ld * 8);
    uint4 kv[U], vv[U];
    u32 t0 = w * PPW * U;
#pragma unroll
    for (int u = 0; u < U; u++) {
        const u32 t = t0 + (u32)u * PPW + sub;
        // (only the first GQ_ATTN_SPEC rows: every row requested ahead of the position is HBM traffic whether it is needed or not --
        // 128 rows x 32 heads = 2 MiB per layer, which costs what the saved round trip gains)
        const bool in = sp < (u32)QH && t < max_seq && t < (u32)GQ_ATTN_SPEC;
        kv[u] = in ? *reinterpret_cast<const uint4 *>(kcg + (size_t)t * HD + ld * 8) : make_uint4(0, 0, 0, 0);
        vv[u] = in ? *reinterpret_cast<const uint4 *>(vcg + (size_t)t * HD + ld * 8) : make_uint4(0, 0, 0, 0);
    }
    const u32 pos = (u32)pos_ptr[0];
    if (pos >= max_seq) {  // decoding past the cache: the head's output is poisoned (NaN logits), like attn_decode_kernel
        if (sp == 0u && tid < (u32)QH * HD) out[(size_t)h0 * HD + tid] = 0x7e00u;
        return;
    }
    const bool solo = nsplit > 1u && pos + 1u <= 2u * PASS;  // a short context is not worth splitting (see attn_decode_kernel)
    u32 nh = QH, hb = h0;  // heads of this block: [hb, hb + nh)
    if (solo) {
        if (sp >= (u32)QH) return;
        if (QH > 1) nh = 1u, hb = h0 + sp;
        nsplit = 1u;
    }
    const u32 spx = solo ? 0u : sp;
    const u32 per = nsplit > 1u ? (((pos + nsplit) / nsplit + PASS - 1u) / PASS) * PASS : pos + 1u;
    const u32 p0 = spx * per, p1 = min(pos + 1u, p0 + per);
    float qreg[QH][8];
#pragma unroll
    for (int qh = 0; qh < QH; qh++) {
        uint4 qq = q4[qh];
        if (QH > 1 && solo) {  // (the one head of this block: q of head h0 + sp)
#pragma unroll
            for (int j = 1; j < QH; j++)
                if (sp == (u32)j) qq = q4[j];
        }
        const u32 qw[4] = {qq.x, qq.y, qq.z, qq.w};
#pragma unroll
        for (int e = 0; e < 4; e++) qreg[qh][2 * e] = h2f((uint16_t)(qw[e] & 0xFFFF)), qreg[qh][2 * e + 1] = h2f((uint16_t)(qw[e] >> 16));
    }
    float m_run[QH], s_run[QH], acc[QH][8];
#pragma unroll
    for (int qh = 0; qh < QH; qh++) {
        m_run[qh] = -3.0e38f, s_run[qh] = 0.f;
#pragma unroll
        for (int e = 0; e < 8; e++) acc[qh][e] = 0.f;
    }
    auto request = [&](uint4 (&kd)[U], uint4 (&vd)[U], u32 tb) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const u32 t = tb + (u32)u * PPW + sub;
            const bool in = t < p1;
            kd[u] = in ? *reinterpret_cast<const uint4 *>(kcg + (size_t)t * HD + ld * 8) : make_uint4(0, 0, 0, 0);
            vd[u] = in ? *reinterpret_cast<const uint4 *>(vcg + (size_t)t * HD + ld * 8) : make_uint4(0, 0, 0, 0);
        }
    };
    t0 = p0 + w * PPW * U;
    const bool have = p0 == 0u && t0 + PPW * U <= (u32)GQ_ATTN_SPEC;  // the batch at t0 is already in registers
    if (!have && t0 < p1) request(kv, vv, t0);
    for (; t0 < p1; t0 += NW * PPW * U) {
        uint4 kn[U], vn[U];
        const bool more = QH > 1 && t0 + NW * PPW * U < p1;  // (QH > 1: the next pass is on its way while this one is multiplied)
        if (more) request(kn, vn, t0 + NW * PPW * U);
        float pu[QH][U], vfu[U][8];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const u32 t = t0 + (u32)u * PPW + sub;
            const bool valid = t < p1;
            const u32 kw[4] = {kv[u].x, kv[u].y, kv[u].z, kv[u].w}, vw[4] = {vv[u].x, vv[u].y, vv[u].z, vv[u].w};
            float kf[8];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                kf[2 * e] = h2f((uint16_t)(kw[e] & 0xFFFF));
                kf[2 * e + 1] = h2f((uint16_t)(kw[e] >> 16));
                vfu[u][2 * e] = valid ? h2f((uint16_t)(vw[e] & 0xFFFF)) : 0.f;
                vfu[u][2 * e + 1] = valid ? h2f((uint16_t)(vw[e] >> 16)) : 0.f;
            }
#pragma unroll
            for (int qh = 0; qh < QH; qh++) {
                if (QH > 1 && (u32)qh >= nh) continue;
                float p = 0.f;
#pragma unroll
                for (int e = 0; e < 8; e++) p += qreg[qh][e] * kf[e];
                p += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, p), 0xB1, 0xF, 0xF, false));
                p += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, p), 0x4E, 0xF, 0xF, false));
                p += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, p), 0x141, 0xF, 0xF, false));
                if (LPP == 16) p += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, p), 0x140, 0xF, 0xF, false));
                pu[qh][u] = valid ? p * scale : -3.0e38f;  // (a stale row past the position may have produced anything, NaN included)
            }
        }
#pragma unroll
        for (int qh = 0; qh < QH; qh++) {
            if (QH > 1 && (u32)qh >= nh) continue;  // (wave-uniform: a solo block multiplies one head)
            float m_new = m_run[qh];
#pragma unroll
            for (int u = 0; u < U; u++) m_new = fmaxf(m_new, pu[qh][u]);
            const float resc = __expf(m_run[qh] - m_new);
            s_run[qh] *= resc;
#pragma unroll
            for (int e = 0; e < 8; e++) acc[qh][e] *= resc;
#pragma unroll
            for (int u = 0; u < U; u++) {
                const float wgt = pu[qh][u] > -2.0e38f ? __expf(pu[qh][u] - m_new) : 0.f;
                s_run[qh] += wgt;
#pragma unroll
                for (int e = 0; e < 8; e++) acc[qh][e] += wgt * vfu[u][e];
            }
            m_run[qh] = m_new;
        }
        if (QH > 1) {
#pragma unroll
            for (int u = 0; u < U; u++) kv[u] = kn[u], vv[u] = vn[u];
        } else if (t0 + NW * PPW * U < p1) {
            request(kv, vv, t0 + NW * PPW * U);
        }
    }
    const u32 stream = w * PPW + sub;
#pragma unroll
    for (int qh = 0; qh < QH; qh++) {
        if (QH > 1 && (u32)qh >= nh) continue;
#pragma unroll
        for (int e = 0; e < 8; e++) red2[((size_t)qh * NS + stream) * HD + ld * 8 + e] = acc[qh][e];
        if (ld == 0) {
            sc[qh * 2u * NS + stream] = m_run[qh];
            sc[qh * 2u * NS + NS + stream] = s_run[qh];
        }
    }
    __syncthreads();
    // the factors e^(m_i - M) of the streams once per head (one thread per stream) instead of once per output element: the same
    // values, 32 exponentials less on the tail of every thread
    float *fl = red2 + (size_t)QH * NS * HD;  // [QH][NS] factors, [QH] maxima
    // Streams that saw no position (a short context: wave w starts at position p0 + 16 w) hold m = -3e38, l = 0, o = 0: their factor
    // is exactly 0 and they add exactly 0 -- the merge runs over the groups of 8 streams that can hold something, same sums bit for bit
    // (at the 50 positions of an average bench step: 16 of the 32 streams).
    const u32 nw_act = min(NW, (p1 - p0 + (u32)(PPW * U) - 1u) / (u32)(PPW * U));
    const u32 ng = (nw_act * (u32)PPW + 7u) >> 3;
    if (tid < nh * NS) {
        const u32 qh = tid / NS, i = tid % NS;
        const float *scq = sc + qh * 2u * NS;
        float M = -3.0e38f;
        for (u32 g8 = 0; g8 < ng; g8++)
#pragma unroll
            for (u32 k = 0; k < 8u; k++) M = fmaxf(M, scq[8u * g8 + k]);
        fl[qh * NS + i] = __expf(scq[i] - M);
        if (i == 0u) fl[(u32)QH * NS + qh] = M;
    }
    __syncthreads();
    if (tid < nh * HD) {  // thread (head, dim): the streams of the head merged in ascending order
        const u32 qh = tid / HD, dd = tid % HD, h = hb + qh;
        const float *scq = sc + qh * 2u * NS, *rq = red2 + (size_t)qh * NS * HD, *fq = fl + qh * NS;
        const float M = fl[(u32)QH * NS + qh];
        float o = 0.f, sum = 0.f;
        for (u32 g8 = 0; g8 < ng; g8++)
#pragma unroll
            for (u32 k = 0; k < 8u; k++) {
                const u32 i = 8u * g8 + k;
                const float f = fq[i];
                sum += scq[NS + i] * f;
                o += rq[i * HD + dd] * f;
            }
        if (nsplit > 1u) {
            float *wp = ws + ((size_t)h * nsplit + sp) * (HD + 2u);
            wp[dd] = o;
            if (dd == 0) {
                wp[HD] = M;
                wp[HD + 1] = sum;
            }
        } else {
            out[(size_t)h * HD + dd] = h2u((h16)(o / sum));
        }
    }
}

// split-KV combine: out[h] = sum_s o_s e^(M_s - M) / sum_s l_s e^(M_s - M)
template <int HD>
__global__ void __launch_bounds__(HD) attn_combine_kernel(const float *ws, uint16_t *out, u32 nsplit, const int *pos_ptr, u32 max_seq) {
    const u32 h = blockIdx.x, tid = threadIdx.x;
    {   // (the same rule as in attn_decode_kernel: a short context was finished by split 0)
        constexpr u32 PASS = ATTN_WAVES * (64u / (HD / 8u)) * 4u;
        u32 pos = (u32)pos_ptr[0];
        if (pos >= max_seq || pos + 1u <= 2u * PASS) return;
    }
    const float *wp = ws + (size_t)h * nsplit * (HD + 2u);
    // Every load of this kernel reads what another CU has just written (an L2 miss each): a loop with one dependent load per split
    // costs n_split memory round trips (measured: 19 -> 45 us per layer from 8 to 64 splits).  Lane s of every wave takes (M_s, l_s)
    // in one round trip, the partial outputs come 8 splits per round trip; the sums keep the ascending split order.
    const u32 l = tid & 63u;
    const float Ms = l < nsplit ? wp[l * (HD + 2u) + HD] : -3.0e38f;  // (n_split <= 64; empty splits: M_s = -3e38, l_s = 0, o_s = 0)
    const float ls = l < nsplit ? wp[l * (HD + 2u) + HD + 1] : 0.f;
    // (the partial outputs of the first 32 splits are requested in the same round trip as the scalars)
    float ov[32];
#pragma unroll
    for (u32 k = 0; k < 32; k++) ov[k] = k < nsplit ? wp[k * (HD + 2u) + tid] : 0.f;
    float M = Ms;
#pragma unroll
    for (int sh = 32; sh >= 1; sh >>= 1) M = fmaxf(M, __shfl_xor(M, sh, 64));
    const float fs = __expf(Ms - M);
    float o = 0.f, sum = 0.f;
#pragma unroll
    for (u32 k = 0; k < 32; k++)
        if (k < nsplit) {  // (uniform)
            const float f = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, fs), (int)k));
            sum += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ls), (int)k)) * f;
            o += ov[k] * f;
        }
    for (u32 s0 = 32u; s0 < nsplit; s0 += 8u) {
        float ow[8];
#pragma unroll
        for (u32 k = 0; k < 8; k++) ow[k] = s0 + k < nsplit ? wp[(s0 + k) * (HD + 2u) + tid] : 0.f;
#pragma unroll
        for (u32 k = 0; k < 8; k++)
            if (s0 + k < nsplit) {
                const float f = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, fs), (int)(s0 + k)));
                sum += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ls), (int)(s0 + k))) * f;
                o += ow[k] * f;
            }
    }
    out[(size_t)h * HD + tid] = h2u((h16)(o / sum));
}

// ------------------------------------------------------------------------------------------------ sampling
// Top-k sampling exactly as generate.py:53-73 defines the distribution: logits / max(T, 1e-5), keep the top k,
// softmax, then draw with the exponential race  argmax_i p_i / q_i, q ~ Exp(1)   (multinomial_sample_one_no_sync).
// Equivalent form used here: argmax_i (logit_i / T - log q_i) over the k candidates.  The random numbers come from a
// counter-based hash of (seed, step counter in device memory) -- torch's Philox stream is not reproduced (the
// reference's sampling is not reproducible across runs either: no fixed generator on the sampled path).
// Two launches: per-block candidates, then a single-block merge.  Writes the next token AND feeds it back into
// `tok_io` / increments `pos_io` so a captured graph advances by itself.
constexpr int SAMP_BLOCKS = 128, SAMP_K = 64;  // (the kernels are built for 32 and for 64 candidates per block)
static_assert(GQ_SSQ_SLOTS == 1024, "sample_stage2 writes one hand-over slot per thread");

__device__ __forceinline__ u32 hash32(u32 x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// Selection of the K largest by binary search on a UNIQUE integer key (monotone image of the fp16 value in the high
// bits, inverted index in the low bits so that ties go to the lowest index): nbits rounds of "count keys >= candidate"
// (registers + one DPP wave reduction + one LDS combine per round) instead of K rounds of block-wide arg-max.
__device__ __forceinline__ u32 ordered_key16(uint16_t h) { return (h & 0x8000u) ? (u32)(uint16_t)~h : ((u32)h | 0x8000u); }
__device__ __forceinline__ uint16_t unordered_key16(u32 o) { return (o & 0x8000u) ? (uint16_t)(o & 0x7FFFu) : (uint16_t)~o; }

// The K-th largest of the 64 * EPT keys of ONE wave by binary search on the key bits: the count of a round is EPT ballots +
// scalar population counts -- no LDS, no barrier (the block-wide version above pays a 1024-thread barrier per bit: 13 us
// for 33 bits).  Two levels of it replace a block-wide search: every wave keeps its own top K (a global top-K key is in
// the top K of its wave), one wave then searches the survivors.
// The low `lowbits` bits of a key are the tie-breaking position: they are searched only when the value bits leave more than
// K keys at the threshold (rare), which halves the rounds.
template <int EPT, typename KeyT>
__device__ __forceinline__ KeyT wave_select_threshold(const KeyT (&key)[EPT], int nbits, int lowbits, int K) {
    KeyT t = 0;
    int ct = 0x7FFFFFFF;  // count of keys >= t
    for (int bit = nbits - 1; bit >= 0; bit--) {
        if (bit == lowbits - 1 && ct == K) break;  // exactly K keys carry a value >= the threshold value: no tie to break
        const KeyT cand = t | ((KeyT)1 << bit);
        int c = 0;
#pragma unroll
        for (int e = 0; e < EPT; e++) c += __popcll(__builtin_amdgcn_ballot_w64(key[e] >= cand));
        if (c >= K) {  // wave-uniform
            t = cand;
            ct = c;
        }
    }
    return t;
}
// writes the keys >= t (and != 0) of the wave to dst[0 .. count) in lane order; returns the count (<= K for a threshold
// from wave_select_threshold: keys are unique)
template <int EPT, typename KeyT>
__device__ __forceinline__ int wave_compact(const KeyT (&key)[EPT], KeyT t, KeyT *dst, int cap, u32 l) {
    int base = 0;
#pragma unroll
    for (int e = 0; e < EPT; e++) {
        const bool p = key[e] >= t && key[e] != 0;
        const unsigned long long b = __builtin_amdgcn_ballot_w64(p);
        const int pos = base + __popcll(b & ((1ull << l) - 1ull));
        if (p && pos < cap) dst[pos] = key[e];
        base += __popcll(b);
    }
    return base;
}

// Extras of the extended entry point (gq_sample_topk_ex, round 5); all optional:
//   ban: device words {n (<= 4), until_pos, id0..id3} -- while *pos_io < until_pos the listed tokens cannot be drawn (HF's
//        MinNewTokensLengthLogitsProcessor: EOS is suppressed until min_new_tokens are out);
//   seq_out[*pos_io + 1] = the drawn token (the host reads whole chunks of the sequence instead of cloning one word per step);
//   emb_table / x_out / ssq_out: the embedding row of the drawn token -> the hidden-state buffer of the NEXT step (+ its statistics
//        hand-over, gq_embed_lookup_ho) -- the step's graph then starts at layer 0's first GEMV, one launch less per token.
struct SampleEx {
    const int *ban;
    int *seq_out;
    u32 seq_cap;
    const uint16_t *emb_table;
    uint16_t *x_out;
    u32 dim, vocab;
    float *ssq_out;
    float top_p;  // nucleus threshold (round 6: gq_sample_topk_p); >= 1 or <= 0: off
};

// what both draw kernels do with the token: outputs, token / position feedback, the sequence store
__device__ __forceinline__ void sample_publish(int tokc, u32 ctr, int *counter, int *tok_io, int *pos_io, int *next_tok, const SampleEx &ex) {
    next_tok[0] = tokc;
    counter[0] = (int)(ctr + 1u);
    if (tok_io) tok_io[0] = tokc;
    const int p = pos_io ? pos_io[0] : 0;
    if (ex.seq_out && (u32)(p + 1) < ex.seq_cap) ex.seq_out[p + 1] = tokc;
    if (pos_io) pos_io[0] = p + 1;
}
// the next step's hidden state: tok_embeddings[token] (+ the statistics hand-over for layer 0's RMSNorm); 1024 threads
__device__ __forceinline__ void sample_embed(int chosen, u32 tid, const SampleEx &ex) {
    u32 tk = (u32)chosen;
    if (tk >= ex.vocab) tk = 0;
    float acc = 0.f;
    for (u32 i = tid; i < ex.dim / 8u; i += 1024u) {
        const uint4 v = reinterpret_cast<const uint4 *>(ex.emb_table + (size_t)tk * ex.dim)[i];
        gq_store_wt(reinterpret_cast<uint4 *>(ex.x_out) + i, v);
        const u32 wd[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float a = (float)__builtin_bit_cast(_Float16, (uint16_t)(wd[k] & 0xFFFFu)), b = (float)__builtin_bit_cast(_Float16, (uint16_t)(wd[k] >> 16));
            acc += a * a;
            acc += b * b;
        }
    }
    if (ex.ssq_out) gq_store_wt(ex.ssq_out + tid, acc);  // (1024 threads = GQ_SSQ_SLOTS)
}

// stage 1: each of the 128 blocks selects the top KM of its slice (<= 1024 logits, 4 per thread in registers); KM = 32 or 64
template <int KM>
__global__ void __launch_bounds__(256) sample_stage1(const uint16_t *logits, u32 V, float *cand_val, int *cand_idx, const int *ban, const int *pos_io) {
    __shared__ u32 surv[4 * KM];
    __shared__ u32 fin[KM];
    const u32 per = (V + SAMP_BLOCKS - 1) / SAMP_BLOCKS;  // <= 1024
    const u32 lo = blockIdx.x * per, hi = min(lo + per, V);
    const u32 tid = threadIdx.x, w = tid >> 6, l = tid & 63u;
    int nban = 0, bid[4] = {-1, -1, -1, -1};
    if (ban) {  // (wave-uniform scalar loads)
        nban = ban[0];
        if (pos_io && pos_io[0] >= ban[1]) nban = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) bid[i] = i < nban ? ban[2 + i] : -1;
    }
    u32 key[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const u32 li = tid * 4u + (u32)e, gi = lo + li;
        // key = (order-preserving image of the fp16 logit, position): unique, and both parts are recovered from it
        const bool banned = (int)gi == bid[0] || (int)gi == bid[1] || (int)gi == bid[2] || (int)gi == bid[3];
        key[e] = (gi < hi && !banned) ? ((ordered_key16(logits[gi]) << 10) | (1023u - li)) : 0u;
    }
    if (l < KM / 2) reinterpret_cast<unsigned long long *>(surv + w * KM)[l] = 0ull;  // (a wave's LDS ops are in order)
    const u32 t = wave_select_threshold<4, u32>(key, 26, 10, KM);
    wave_compact<4, u32>(key, t, surv + w * KM, KM, l);
    __syncthreads();
    if (w == 0) {  // top KM of the 4 * KM survivors
        u32 k2[KM / 16];
#pragma unroll
        for (int e = 0; e < KM / 16; e++) k2[e] = surv[(u32)e * 64u + l];
        const u32 t2 = wave_select_threshold<KM / 16, u32>(k2, 26, 10, KM);
        if (l < KM) fin[l] = 0u;
        const int n = wave_compact<KM / 16, u32>(k2, t2, fin, KM, l);
        if (l < KM) {
            const u32 k = fin[l];
            const bool ok = (int)l < n && k != 0u;
            const u32 li = 1023u - (k & 1023u);
            cand_val[blockIdx.x * KM + l] = ok ? h2f(unordered_key16(k >> 10)) : -3.0e38f;  // slice shorter than K: padded
            cand_idx[blockIdx.x * KM + l] = ok ? (int)(lo + li) : -1;
        }
    }
}

// stage 2: one block, 128 * KM candidates (KM / 8 per thread), select the global top-k, then the exponential-race draw
template <int KM>
__global__ void __launch_bounds__(1024) sample_stage2(const float *cand_val, const int *cand_idx, int top_k, float temperature,
                                                      u32 seed, int *counter, int *tok_io, int *pos_io, int *next_tok, SampleEx ex) {
    constexpr int EPT = KM / 8;  // candidates per thread
    __shared__ float selv[KM];
    __shared__ int seli[KM];
    __shared__ unsigned long long surv[16 * KM];
    __shared__ unsigned long long fin[KM];
    __shared__ int slot, chosen;
    const u32 tid = threadIdx.x, w = tid >> 6, l = tid & 63u;
    unsigned long long key[EPT];
#pragma unroll
    for (int e = 0; e < EPT; e++) {
        const u32 c = tid * (u32)EPT + (u32)e;
        const float v = cand_val[c];
        const int id = cand_idx[c];
        const uint16_t hb = __builtin_bit_cast(uint16_t, (h16)v);  // candidates are fp16 values: exact
        key[e] = id >= 0 ? (((unsigned long long)ordered_key16(hb) << 17) | (unsigned long long)(131071u - (u32)id)) : 0ull;
    }
    const int K = top_k < 1 ? 1 : (top_k > KM ? KM : top_k);
    if (l < KM) surv[w * KM + l] = 0ull;
    const unsigned long long t = wave_select_threshold<EPT, unsigned long long>(key, 33, 17, K);
    wave_compact<EPT, unsigned long long>(key, t, surv + w * KM, KM, l);
    __syncthreads();
    if (w == 0) {
        unsigned long long k8[KM / 4];
#pragma unroll
        for (int e = 0; e < KM / 4; e++) k8[e] = surv[(u32)e * 64u + l];
        const unsigned long long t2 = wave_select_threshold<KM / 4, unsigned long long>(k8, 33, 17, K);
        if (l < KM) fin[l] = 0ull;
        const int n2 = wave_compact<KM / 4, unsigned long long>(k8, t2, fin, KM, l);
        if (l == 0) slot = n2;
        if (l < KM) {
            const unsigned long long k = fin[l];
            selv[l] = h2f(unordered_key16((u32)(k >> 17)));
            seli[l] = (int)(131071u - (u32)(k & 131071ull));
        }
    }
    __syncthreads();
    if (w == 0) {
        const int n = min(slot, K);
        const float T = fmaxf(temperature, 1e-5f);
        const u32 ctr = (u32)counter[0];
        float score = -3.0e38f;
        int tokc = 0x7FFFFFFF;
        // nucleus filter on the top-k survivors -- transformers' TopPLogitsWarper behind TopKLogitsWarper behind the temperature
        // (generation/logits_process.py: probabilities of the scaled, top-k-filtered scores, cumulative sum in ASCENDING order, a token
        // goes when the sum up to and including it is <= 1 - top_p, the most probable one always stays).  n <= 64 candidates, one per
        // lane: the cumulative sum of a lane is a loop over the wave (ties ordered by token id, higher id first = the lower key).
        bool keep = (int)l < n;
        if (ex.top_p > 0.f && ex.top_p < 1.f) {
            const float v = (int)l < n ? selv[l] / T : -3.0e38f;
            float mx = v;
#pragma unroll
            for (int sh = 32; sh >= 1; sh >>= 1) mx = fmaxf(mx, __shfl_xor(mx, sh, 64));
            const float e = (int)l < n ? __expf(v - mx) : 0.f;
            float z = e;
#pragma unroll
            for (int sh = 32; sh >= 1; sh >>= 1) z += __shfl_xor(z, sh, 64);
            const float pr = e / z;
            const int myid = (int)l < n ? seli[l] : -1;
            float cum = 0.f;
            int greater = 0;
            for (int j = 0; j < n; j++) {
                const float vj = __shfl(v, j, 64), pj = __shfl(pr, j, 64);
                const int idj = __shfl(myid, j, 64);
                const bool below = vj < v || (vj == v && idj > myid);  // j sorts before this lane in ascending order
                cum += (below || j == (int)l) ? pj : 0.f;
                greater += (vj > v || (vj == v && idj < myid)) ? 1 : 0;
            }
            keep = keep && (greater == 0 || !(cum <= 1.0f - ex.top_p));
        }
        if (keep) {
            // q ~ Exp(1) = -log(u), u in (0,1]; the random number is tied to the TOKEN id, not to the slot
            const u32 r = hash32(seed ^ hash32(ctr * 0x9E3779B9u + (u32)seli[l] + 1u));
            const float u = ((float)(r >> 8) + 1.0f) * (1.0f / 16777216.0f);
            score = selv[l] / T - __logf(-__logf(u));
            tokc = seli[l];
        }
#pragma unroll
        for (int sh = 32; sh >= 1; sh >>= 1) {
            const float os = __shfl_xor(score, sh, 64);
            const int ot = __shfl_xor(tokc, sh, 64);
            if (os > score || (os == score && ot < tokc)) { score = os; tokc = ot; }
        }
        if (l == 0) {
            sample_publish(tokc, ctr, counter, tok_io, pos_io, next_tok, ex);
            chosen = tokc;
        }
    }
    if (ex.x_out) {
        __syncthreads();
        sample_embed(chosen, tid, ex);
    }
}

// (Round 5, measured and removed: a ONE-launch greedy draw -- one 1024-thread block keeps the largest key of its 16-byte units of the
// logits -- for temperature 0.  21 us with a load loop, no better than the two launches above with all 16 loads of a thread in flight
// (865.6 / 864.1 against 867.9 / 868.3 tokens/s): one CU takes the 250 KiB of logits slower than 128 blocks + one do.)

// ------------------------------------------------------------------------------------------------ dense fp16 GEMV
// out[n] = sum_k x[k] * W[n][k], fp32 accumulation (v_dot2_f32_f16), fp16 output -- what nn.Linear(fp16) gives at M = 1.
// Optional RMSNorm prologue on x (final norm before lm_head).  One wave = RW rows at a time, 16-byte loads,
// all K/512 * RW loads of a row group in flight.
template <int RW>
__global__ void __launch_bounds__(256) dense_gemv_kernel(const uint16_t *x, const uint16_t *W, uint16_t *out, u32 N, u32 K,
                                                         const uint16_t *normw, float eps, u32 rows_per_block) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t *xs = reinterpret_cast<uint16_t *>(smem);
    float *red = reinterpret_cast<float *>(xs + K);
    const u32 tid = threadIdx.x, w = tid >> 6, l = tid & 63u;
    float nscale = 1.f;
    // (round 5: x and the norm weights of a thread's first two units are requested together and kept -- the weights are a layer
    // tensor that comes from HBM; requested only behind the sum of squares they cost the launch a second memory round trip)
    constexpr u32 NPRE = 2;
    uint4 xpre[NPRE], npre[NPRE];
    auto sq8 = [&](const uint4 &v, float &ss) {
        const u32 ww[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const float a = h2f(ww[i] & 0xFFFF), b = h2f(ww[i] >> 16);
            ss += a * a;
            ss += b * b;
        }
    };
    auto norm8 = [&](uint4 v, const uint4 &nv) {
        u32 ww[4] = {v.x, v.y, v.z, v.w};
        const u32 nw[4] = {nv.x, nv.y, nv.z, nv.w};
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const h16 a = (h16)gq_pin_f32(h2f(ww[i] & 0xFFFF) * nscale) * u2h((uint16_t)(nw[i] & 0xFFFF));
            const h16 b = (h16)gq_pin_f32(h2f(ww[i] >> 16) * nscale) * u2h((uint16_t)(nw[i] >> 16));
            ww[i] = (u32)h2u(a) | ((u32)h2u(b) << 16);
        }
        return make_uint4(ww[0], ww[1], ww[2], ww[3]);
    };
    if (normw) {
        float ss = 0.f;
#pragma unroll
        for (u32 k = 0; k < NPRE; k++) {
            const u32 g = tid + 256u * k;
            xpre[k] = g < K / 8u ? reinterpret_cast<const uint4 *>(x)[g] : make_uint4(0u, 0u, 0u, 0u);
            npre[k] = g < K / 8u ? reinterpret_cast<const uint4 *>(normw)[g] : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (u32 k = 0; k < NPRE; k++) sq8(xpre[k], ss);  // (units beyond K are zero)
        for (u32 g = tid + 256u * NPRE; g < K / 8u; g += 256) sq8(reinterpret_cast<const uint4 *>(x)[g], ss);
        ss = wave_reduce<false>(ss);
        if (l == 0) red[w] = ss;
        __syncthreads();
        nscale = 1.0f / sqrtf((red[0] + red[1] + red[2] + red[3]) / (float)K + eps);
#pragma unroll
        for (u32 k = 0; k < NPRE; k++) {
            const u32 g = tid + 256u * k;
            if (g < K / 8u) reinterpret_cast<uint4 *>(xs)[g] = norm8(xpre[k], npre[k]);
        }
        for (u32 g = tid + 256u * NPRE; g < K / 8u; g += 256)
            reinterpret_cast<uint4 *>(xs)[g] = norm8(reinterpret_cast<const uint4 *>(x)[g], reinterpret_cast<const uint4 *>(normw)[g]);
    } else {
        for (u32 g = tid; g < K / 8u; g += 256) reinterpret_cast<uint4 *>(xs)[g] = reinterpret_cast<const uint4 *>(x)[g];
    }
    __syncthreads();
    const u32 row_begin = blockIdx.x * rows_per_block;
    const u32 row_end = min(row_begin + rows_per_block, N);
    for (u32 r0 = row_begin + w * RW; r0 < row_end; r0 += 4 * RW) {
        float acc[RW];
#pragma unroll
        for (int r = 0; r < RW; r++) acc[r] = 0.f;
        for (u32 k0 = l * 8u; k0 < K; k0 += 512u) {
            const uint4 xv = *reinterpret_cast<const uint4 *>(xs + k0);
            uint4 wv[RW];
#pragma unroll
            for (int r = 0; r < RW; r++) {
                const u32 row = min(r0 + (u32)r, N - 1u);
                const u32x4 t4 = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(W + (size_t)row * K + k0));
                wv[r] = make_uint4(t4.x, t4.y, t4.z, t4.w);
            }
#pragma unroll
            for (int r = 0; r < RW; r++) {
                acc[r] = __builtin_amdgcn_fdot2(__builtin_bit_cast(h16x2, wv[r].x), __builtin_bit_cast(h16x2, xv.x), acc[r], false);
                acc[r] = __builtin_amdgcn_fdot2(__builtin_bit_cast(h16x2, wv[r].y), __builtin_bit_cast(h16x2, xv.y), acc[r], false);
                acc[r] = __builtin_amdgcn_fdot2(__builtin_bit_cast(h16x2, wv[r].z), __builtin_bit_cast(h16x2, xv.z), acc[r], false);
                acc[r] = __builtin_amdgcn_fdot2(__builtin_bit_cast(h16x2, wv[r].w), __builtin_bit_cast(h16x2, xv.w), acc[r], false);
            }
        }
#pragma unroll
        for (int r = 0; r < RW; r++) {
            const float s = wave_reduce<false>(acc[r]);
            if (l == 0 && r0 + (u32)r < row_end) out[r0 + r] = h2u((h16)s);
        }
    }
}

int cu_count() { return gq_cu_count(); }

}  // namespace

extern "C" int gq_embed_lookup(const int *token, const void *table, void *out, uint32_t dim, uint32_t vocab, void *stream) {
    if (!token || !table || !out) return gq_fail(GQ_EINVAL, "null pointer argument.");
    if (dim % 8u) return gq_fail(GQ_EINVAL, "embedding dim must be a multiple of 8.");
    hipLaunchKernelGGL(embed_kernel, dim3((dim / 8u + 255u) / 256u), dim3(256), 0, (hipStream_t)stream, token,
                       (const uint16_t *)table, (uint16_t *)out, dim, vocab);
    GQ_HIP_CHECK(hipGetLastError());
    return GQ_OK;
}
extern "C" int gq_embed_lookup_ho(const int *token, const void *table, void *out, uint32_t dim, uint32_t vocab, float *ssq_out, void *stream) {
    if (!ssq_out) return gq_embed_lookup(token, table, out, dim, vocab, stream);
    if (!token || !table || !out) return gq_fail(GQ_EINVAL, "null pointer argument.");
    if (dim % 8u) return gq_fail(GQ_EINVAL, "embedding dim must be a multiple of 8.");
    hipLaunchKernelGGL(embed_ssq_kernel, dim3(1), dim3(GQ_SSQ_SLOTS), 0, (hipStream_t)stream, token, (const uint16_t *)table, (uint16_t *)out, dim,
                       vocab, ssq_out);
    GQ_HIP_CHECK(hipGetLastError());
    return GQ_OK;
}

namespace {
template <bool QT>
int attn_launch(const void *qkv, const AttnQt &qt, const int *pos, const void *cos_table, const void *sin_table, void *k_cache, void *v_cache,
                void *out, uint32_t n_head, uint32_t n_kv_head, uint32_t head_dim, uint32_t max_seq, float scale, uint32_t n_split,
                float *workspace, void *stream) {
    if ((!QT && !qkv) || !pos || !cos_table || !sin_table || !k_cache || !v_cache || !out) return gq_fail(GQ_EINVAL, "null pointer argument.");
    if (n_kv_head == 0 || n_head % n_kv_head) return gq_fail(GQ_EINVAL, "n_head must be a multiple of n_kv_head.");
    if (head_dim != 64 && head_dim != 128) return gq_fail(GQ_ENOTSUP, "head_dim must be 64 or 128.");
    if (n_split < 1u || n_split > 64u || (n_split > 1u && !workspace)) return gq_fail(GQ_EINVAL, "n_split in 1..64, with a workspace when > 1.");
    const u32 nstreams = (u32)ATTN_WAVES * 64u / (head_dim / 8u);
    size_t smem = ((size_t)2u * nstreams + 7u * head_dim + 2u * ATTN_WAVES + 16u + (size_t)nstreams * head_dim) * 4u;
    if (QT) smem += 3u * (size_t)head_dim * 4u + 3u * head_dim * 2u + 3u * (size_t)(64u * ATTN_WAVES / (head_dim / 4u)) * head_dim * 4u;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(n_head, n_split);
    if (head_dim == 128) {
        static GqPerDeviceOnce once;
        GQ_HIP_CHECK(once.max_dynamic_lds(reinterpret_cast<const void *>(attn_decode_kernel<128, QT>), 160 * 1024));
        hipLaunchKernelGGL((attn_decode_kernel<128, QT>), grid, dim3(64 * ATTN_WAVES), smem, s, (const uint16_t *)qkv, pos,
                           (const uint16_t *)cos_table, (const uint16_t *)sin_table, (uint16_t *)k_cache, (uint16_t *)v_cache,
                           (uint16_t *)out, n_head, n_kv_head, max_seq, scale, n_split, workspace, qt);
        if (n_split > 1u) hipLaunchKernelGGL(attn_combine_kernel<128>, dim3(n_head), dim3(128), 0, s, workspace, (uint16_t *)out, n_split, pos, max_seq);
    } else {
        static GqPerDeviceOnce once;
        GQ_HIP_CHECK(once.max_dynamic_lds(reinterpret_cast<const void *>(attn_decode_kernel<64, QT>), 160 * 1024));
        hipLaunchKernelGGL((attn_decode_kernel<64, QT>), grid, dim3(64 * ATTN_WAVES), smem, s, (const uint16_t *)qkv, pos,
                           (const uint16_t *)cos_table, (const uint16_t *)sin_table, (uint16_t *)k_cache, (uint16_t *)v_cache,
                           (uint16_t *)out, n_head, n_kv_head, max_seq, scale, n_split, workspace, qt);
        if (n_split > 1u) hipLaunchKernelGGL(attn_combine_kernel<64>, dim3(n_head), dim3(64), 0, s, workspace, (uint16_t *)out, n_split, pos, max_seq);
    }
    GQ_HIP_CHECK(hipGetLastError());
    return GQ_OK;
}
}  // namespace

extern "C" int gq_attn_decode_split(const void *qkv, const int *pos, const void *cos_table, const void *sin_table, void *k_cache,
                                    void *v_cache, void *out, uint32_t n_head, uint32_t n_kv_head, uint32_t head_dim, uint32_t max_seq,
                                    float scale, uint32_t n_split, float *workspace, void *stream) {
    return attn_launch<false>(qkv, AttnQt{}, pos, cos_table, sin_table, k_cache, v_cache, out, n_head, n_kv_head, head_dim, max_seq, scale, n_split,
                              workspace, stream);
}

// QTIP models: the same attention with the transform-out of the q, k and v linears folded in (qkv_lin[0..2]: the GqQtipOut
// descriptors gq_qtip_linear_out would take; resid / out unused).  Equal to gq_qtip_linear_out + gq_attn_decode_split up to fp32
// rounding (the segments are combined first: the additions of the full transform in another order), not bit for bit.
extern "C" int gq_attn_decode_qtip(const GqQtipOut *qkv_lin, const int *pos, const void *cos_table, const void *sin_table, void *k_cache,
                                   void *v_cache, void *out, uint32_t n_head, uint32_t n_kv_head, uint32_t head_dim, uint32_t max_seq,
                                   float scale, uint32_t n_split, float *workspace, void *stream) {
    if (!qkv_lin) return gq_fail(GQ_EINVAL, "null pointer argument.");
    AttnQt qt{};
    for (int i = 0; i < 3; i++) {
        const uint32_t want = (i == 0 ? n_head : n_kv_head) * head_dim, M = qkv_lin[i].M;
        if (M != want || (M & (M - 1u)) || !qkv_lin[i].y32 || !qkv_lin[i].SV32 || (((uintptr_t)qkv_lin[i].y32 | (uintptr_t)qkv_lin[i].SV32) & 15u) ||
            qkv_lin[i].parts > 4u || M > 16u * 64u * ATTN_WAVES)
            return gq_fail(GQ_ENOTSUP, "gq_attn_decode_qtip: q / k / v widths must be n_head (n_kv_head) * head_dim, powers of two, 16-byte aligned sums.");
        qt.y32[i] = qkv_lin[i].y32;
        qt.sv32[i] = qkv_lin[i].SV32;
        qt.M[i] = M;
        qt.parts[i] = qkv_lin[i].parts ? qkv_lin[i].parts : 1u;
        qt.mscale[i] = (float)pow((double)M, -0.5);
    }
    return attn_launch<true>(nullptr, qt, pos, cos_table, sin_table, k_cache, v_cache, out, n_head, n_kv_head, head_dim, max_seq, scale, n_split,
                             workspace, stream);
}


// Attention of the decode step when the wqkv launch has already rotated q / k and written k / v of the current token into the
// caches (gq_anyprec_gemv_qkv_rope): q fp16 [n_head * head_dim] rotated; the caches hold every position <= *pos.
extern "C" int gq_attn_decode_roped(const void *q, const int *pos, const void *k_cache, const void *v_cache, void *out, uint32_t n_head,
                                    uint32_t n_kv_head, uint32_t head_dim, uint32_t max_seq, float scale, uint32_t n_split, float *workspace,
                                    void *stream) {
    if (!q || !pos || !k_cache || !v_cache || !out) return gq_fail(GQ_EINVAL, "null pointer argument.");
    if (n_kv_head == 0 || n_head % n_kv_head) return gq_fail(GQ_EINVAL, "n_head must be a multiple of n_kv_head.");
    if (head_dim != 64 && head_dim != 128) return gq_fail(GQ_ENOTSUP, "head_dim must be 64 or 128.");
    if (n_split < 1u || n_split > 64u || (n_split > 1u && !workspace)) return gq_fail(GQ_EINVAL, "n_split in 1..64, with a workspace when > 1.");
    if (((uintptr_t)q | (uintptr_t)k_cache | (uintptr_t)v_cache) & 15u) return gq_fail(GQ_EINVAL, "buffers must be 16-byte aligned.");
    const u32 nstreams = (u32)ATTN_WAVES * 64u / (head_dim / 8u);
    hipStream_t s = (hipStream_t)stream;
    // grouped-query models with a split cache: the 4 query heads of a KV group in one block (every cached row loaded once)
    const bool gqa = n_split >= 4u && (n_head / n_kv_head) % 4u == 0u && gq_env_int("GQ_ATTN_GQA", 1);  // (8 heads per group: two blocks of 4)
#ifdef GQ_ATTN_PROBE
    const u32 copies = (u32)gq_env_int("GQ_ATTN_PROBE_COPIES", 0);
    const bool gqa_p = copies > 0u;
    const u32 qh = gqa_p || gqa ? 4u : 1u;
    const size_t smem = (size_t)qh * ((size_t)2u * nstreams + (size_t)nstreams * head_dim + nstreams + 1u) * 4u;
    const dim3 grid(n_head / qh, gqa_p ? copies : n_split);
    if (gqa_p) {
        static GqPerDeviceOnce once;
        GQ_HIP_CHECK(once.max_dynamic_lds(reinterpret_cast<const void *>(attn_roped_kernel<128, 4>), 160 * 1024));
        hipLaunchKernelGGL((attn_roped_kernel<128, 4>), grid, dim3(64 * ATTN_WAVES), smem, s, (const uint16_t *)q, pos, (const uint16_t *)k_cache,
                           (const uint16_t *)v_cache, (uint16_t *)out, n_head, n_kv_head, max_seq, scale, 1u, workspace);
        GQ_HIP_CHECK(hipGetLastError());
        return GQ_OK;
    }
#else
    const u32 qh = gqa ? 4u : 1u;
    const size_t smem = (size_t)qh * ((size_t)2u * nstreams + (size_t)nstreams * head_dim + nstreams + 1u) * 4u;
    const dim3 grid(n_head / qh, n_split);
#endif
#define GQ_LAUNCH_ROPED(HD_, QH_)                                                                                                    \
    do {                                                                                                                             \
        static GqPerDeviceOnce once;                                                                                                 \
        GQ_HIP_CHECK(once.max_dynamic_lds(reinterpret_cast<const void *>(attn_roped_kernel<HD_, QH_>), 160 * 1024));                  \
        hipLaunchKernelGGL((attn_roped_kernel<HD_, QH_>), grid, dim3(64 * ATTN_WAVES), smem, s, (const uint16_t *)q, pos,             \
                           (const uint16_t *)k_cache, (const uint16_t *)v_cache, (uint16_t *)out, n_head, n_kv_head, max_seq, scale,   \
                           n_split, workspace);                                                                                      \
        if (n_split > 1u)                                                                                                            \
            hipLaunchKernelGGL(attn_combine_kernel<HD_>, dim3(n_head), dim3(HD_), 0, s, workspace, (uint16_t *)out, n_split, pos, max_seq); \
    } while (0)
    if (head_dim == 128) {
        if (gqa) GQ_LAUNCH_ROPED(128, 4);
        else GQ_LAUNCH_ROPED(128, 1);
    } else {
        if (gqa) GQ_LAUNCH_ROPED(64, 4);
        else GQ_LAUNCH_ROPED(64, 1);
    }
#undef GQ_LAUNCH_ROPED
    GQ_HIP_CHECK(hipGetLastError());
    return GQ_OK;
}

extern "C" int gq_attn_decode(const void *qkv, const int *pos, const void *cos_table, const void *sin_table, void *k_cache,
                              void *v_cache, void *out, uint32_t n_head, uint32_t n_kv_head, uint32_t head_dim, uint32_t max_seq,
                              float scale, void *stream) {
    return gq_attn_decode_split(qkv, pos, cos_table, sin_table, k_cache, v_cache, out, n_head, n_kv_head, head_dim, max_seq, scale, 1u,
                                nullptr, stream);
}

extern "C" int gq_dense_gemv_f16(const void *x, const void *W, void *out, uint32_t N, uint32_t K, const void *norm_weight,
                                 float eps, void *stream) {
    if (!x || !W || !out) return gq_fail(GQ_EINVAL, "null pointer argument.");
    if (K == 0 || K % 512u || N == 0) return gq_fail(GQ_EINVAL, "dense GEMV needs N > 0 and K a positive multiple of 512.");
    if ((((uintptr_t)x | (uintptr_t)W | (uintptr_t)norm_weight) & 15u)) return gq_fail(GQ_EINVAL, "buffers must be 16-byte aligned.");
    const size_t smem = (size_t)K * 2u + 64u;
    if (smem > 160u * 1024u) return gq_fail(GQ_ENOTSUP, "K too large.");
    constexpr int RW = 4;
    const u32 ncu = (u32)cu_count();
    // ~3 blocks per CU (measured over 2 .. 16 on the 128256 x 4096 lm_head inside the decode step: 3 is 0.4 % of the token faster than 4,
    // 6 is 0.7 % slower), every block a multiple of 4 waves * RW rows
    const int bpc_env = gq_env_int("GQ_DENSE_BPC", 3);
    const u32 bpc = bpc_env >= 1 ? (u32)bpc_env : 1u;
    u32 rpb = (N + ncu * bpc - 1u) / (ncu * bpc);
    rpb = ((rpb + 4u * RW - 1u) / (4u * RW)) * (4u * RW);
    {   // (GQ_DENSE_RPB: rows per block directly, a multiple of 16 -- the sweep of profiles/r05_knob_sweep.txt)
        const int rpb_env = gq_env_int("GQ_DENSE_RPB", 0);
        if (rpb_env >= 16 && rpb_env % 16 == 0) rpb = (u32)rpb_env;
    }
    const u32 grid = (N + rpb - 1u) / rpb;
    static GqPerDeviceOnce once;
    GQ_HIP_CHECK(once.max_dynamic_lds(reinterpret_cast<const void *>(dense_gemv_kernel<RW>), 160 * 1024));
    hipLaunchKernelGGL(dense_gemv_kernel<RW>, dim3(grid), dim3(256), smem, (hipStream_t)stream, (const uint16_t *)x,
                       (const uint16_t *)W, (uint16_t *)out, N, K, (const uint16_t *)norm_weight, eps, rpb);
    GQ_HIP_CHECK(hipGetLastError());
    return GQ_OK;
}

namespace {
int sample_launch(const void *logits, uint32_t vocab, int top_k, float temperature, uint32_t seed, int *counter, float *work_val, int *work_idx,
                  int *tok_io, int *pos_io, int *next_tok, const SampleEx &ex, void *stream) {
    if (!logits || !counter || !work_val || !work_idx || !next_tok) return gq_fail(GQ_EINVAL, "null pointer argument.");
    if (top_k > SAMP_K) return gq_fail(GQ_ENOTSUP, "top_k > 64 is not supported by the fused sampler.");
    const u32 per = (vocab + SAMP_BLOCKS - 1) / SAMP_BLOCKS;
    if (per > 1024u || vocab > 131072u) return gq_fail(GQ_ENOTSUP, "vocab too large for the fused sampler (<= 131072).");
    hipStream_t s = (hipStream_t)stream;
    if (top_k <= 32) {
        hipLaunchKernelGGL(sample_stage1<32>, dim3(SAMP_BLOCKS), dim3(256), 0, s, (const uint16_t *)logits, vocab, work_val, work_idx, ex.ban, pos_io);
        hipLaunchKernelGGL(sample_stage2<32>, dim3(1), dim3(1024), 0, s, work_val, work_idx, top_k, temperature, seed, counter, tok_io, pos_io, next_tok, ex);
    } else {
        hipLaunchKernelGGL(sample_stage1<64>, dim3(SAMP_BLOCKS), dim3(256), 0, s, (const uint16_t *)logits, vocab, work_val, work_idx, ex.ban, pos_io);
        hipLaunchKernelGGL(sample_stage2<64>, dim3(1), dim3(1024), 0, s, work_val, work_idx, top_k, temperature, seed, counter, tok_io, pos_io, next_tok, ex);
    }
    GQ_HIP_CHECK(hipGetLastError());
    return GQ_OK;
}
}  // namespace

extern "C" int gq_sample_topk(const void *logits, uint32_t vocab, int top_k, float temperature, uint32_t seed, int *counter,
                              float *work_val, int *work_idx, int *tok_io, int *pos_io, int *next_tok, void *stream) {
    if (top_k > 32) return gq_fail(GQ_ENOTSUP, "top_k > 32 is not supported by gq_sample_topk (work buffers of 128 * 32: gq_sample_topk_ex takes 64).");
    return sample_launch(logits, vocab, top_k, temperature, seed, counter, work_val, work_idx, tok_io, pos_io, next_tok, SampleEx{}, stream);
}

extern "C" int gq_sample_topk_ex(const void *logits, uint32_t vocab, int top_k, float temperature, uint32_t seed, int *counter,
                                 float *work_val, int *work_idx, int *tok_io, int *pos_io, int *next_tok, const int *ban, int *seq_out,
                                 uint32_t seq_cap, const void *embed_table, void *x_out, uint32_t dim, float *ssq_out, void *stream) {
    return gq_sample_topk_p(logits, vocab, top_k, 1.0f, temperature, seed, counter, work_val, work_idx, tok_io, pos_io, next_tok, ban, seq_out, seq_cap,
                            embed_table, x_out, dim, ssq_out, stream);
}
extern "C" int gq_sample_topk_p(const void *logits, uint32_t vocab, int top_k, float top_p, float temperature, uint32_t seed, int *counter,
                                float *work_val, int *work_idx, int *tok_io, int *pos_io, int *next_tok, const int *ban, int *seq_out,
                                uint32_t seq_cap, const void *embed_table, void *x_out, uint32_t dim, float *ssq_out, void *stream) {
    if (!(top_p > 0.f)) return gq_fail(GQ_EINVAL, "top_p must be in (0, 1] (1 = no nucleus filter).");
    SampleEx ex{};
    ex.top_p = top_p;
    ex.ban = ban;
    ex.seq_out = seq_out;
    ex.seq_cap = seq_cap;
    if (x_out) {
        if (!embed_table || dim % 8u) return gq_fail(GQ_EINVAL, "x_out needs the embedding table and dim % 8 == 0.");
        ex.emb_table = (const uint16_t *)embed_table;
        ex.x_out = (uint16_t *)x_out;
        ex.dim = dim;
        ex.vocab = vocab;
        ex.ssq_out = ssq_out;
    } else if (ssq_out) {
        return gq_fail(GQ_EINVAL, "ssq_out without x_out.");
    }
    return sample_launch(logits, vocab, top_k, temperature, seed, counter, work_val, work_idx, tok_io, pos_io, next_tok, ex, stream);
}
