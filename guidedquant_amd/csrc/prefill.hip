// prefill.hip -- the element-wise steps of the PROMPT pass (seq_len > 1) between the fused prefill GEMMs (ap_gemm.hip), one
// launch each instead of the ~45 eager tensor ops per layer of the module forward (inference/model.py:121-266 semantics:
// `Transformer.forward` -> `TransformerBlock.forward` -> `Attention.forward` / `FeedForward.forward`).  At S = 128 the
// module forward of the 8B model spends 10 of its 14.5 ms in launch overhead of those ops; the GEMMs take 4.2 ms.
// Every kernel keeps the fp16 rounding points of the tensor expressions it replaces:
//   gq_rmsnorm_rows     RMSNorm.forward (model.py:84-96):  (x.float() * rsqrt(mean(x^2) + eps)).half() * weight, optionally
//                       behind the residual add in front of it (TransformerBlock.forward, model.py:311-313: h = x + attn(..))
//   gq_rope_cache_rows  apply_rotary_pos_emb (model.py:336-341) on q and k -- (q * cos) + (rotate_half(q) * sin), three fp16
//                       operations -- and KVCache.update (model.py:69-79): k, v written at their positions
//   gq_silu_mul_rows    FeedForward.forward (model.py:266):  F.silu(w1(x)) * w3(x), silu rounded to fp16 before the product
// All of them are HBM-bound streams over [S][D] fp16 rows; 16-byte accesses, one block per row (norm) / grid-stride.
#include <hip/hip_runtime.h>

#include "gq_internal.h"

namespace {
using u32 = uint32_t;
typedef _Float16 h16;
__device__ __forceinline__ float h2f(uint16_t h) { return (float)__builtin_bit_cast(h16, h); }
__device__ __forceinline__ uint16_t f2h(float f) { return __builtin_bit_cast(uint16_t, (h16)f); }
__device__ __forceinline__ h16 hbits(uint16_t h) { return __builtin_bit_cast(h16, h); }
__device__ __forceinline__ uint16_t bitsh(h16 h) { return __builtin_bit_cast(uint16_t, h); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int sh = 32; sh >= 1; sh >>= 1) v += __shfl_xor(v, sh, 64);
    return v;
}

// one block (256 threads) per row; the row stays in registers between the two passes (D <= 256 * 8 * 8 = 16384)
// delta != null: the residual add of the block in front (x = x + delta, one fp16 rounding, written back) rides along
__global__ void __launch_bounds__(256) rmsnorm_rows_kernel(uint16_t *__restrict__ x, const uint16_t *__restrict__ delta, const uint16_t *__restrict__ w,
                                                           uint16_t *__restrict__ out, u32 D, float eps) {
    __shared__ float red[4];
    const u32 tid = threadIdx.x, row = blockIdx.x, nu = D / 8u;  // 16-byte units of the row
    uint4 *xr = reinterpret_cast<uint4 *>(x + (size_t)row * D);
    const uint4 *dr = delta ? reinterpret_cast<const uint4 *>(delta + (size_t)row * D) : nullptr;
    uint4 v[8];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const u32 u = tid + 256u * (u32)i;
        v[i] = u < nu ? xr[u] : make_uint4(0u, 0u, 0u, 0u);
        if (delta && u < nu) {
            const uint4 dv = dr[u];
            const u32 xa[4] = {v[i].x, v[i].y, v[i].z, v[i].w}, da[4] = {dv.x, dv.y, dv.z, dv.w};
            u32 r[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const h16 lo = hbits((uint16_t)(xa[e] & 0xFFFF)) + hbits((uint16_t)(da[e] & 0xFFFF)), hi = hbits((uint16_t)(xa[e] >> 16)) + hbits((uint16_t)(da[e] >> 16));
                r[e] = (u32)bitsh(lo) | ((u32)bitsh(hi) << 16);
            }
            v[i] = make_uint4(r[0], r[1], r[2], r[3]);
            xr[u] = v[i];
        }
        const u32 ww[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const float a = h2f((uint16_t)(ww[e] & 0xFFFF)), b = h2f((uint16_t)(ww[e] >> 16));
            ss += a * a;
            ss += b * b;
        }
    }
    ss = wave_sum(ss);
    if ((tid & 63u) == 0u) red[tid >> 6] = ss;
    __syncthreads();
    const float rs = rsqrtf((red[0] + red[1] + red[2] + red[3]) / (float)D + eps);
    uint4 *orow = reinterpret_cast<uint4 *>(out + (size_t)row * D);
    const uint4 *wr = reinterpret_cast<const uint4 *>(w);
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const u32 u = tid + 256u * (u32)i;
        if (u >= nu) continue;
        const uint4 wv = wr[u];
        const u32 xx[4] = {v[i].x, v[i].y, v[i].z, v[i].w}, ww[4] = {wv.x, wv.y, wv.z, wv.w};
        u32 o[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            // (x.float() * rs).half(), then the fp16 product with the weight
            const h16 n0 = (h16)(h2f((uint16_t)(xx[e] & 0xFFFF)) * rs), n1 = (h16)(h2f((uint16_t)(xx[e] >> 16)) * rs);
            const h16 p0 = n0 * hbits((uint16_t)(ww[e] & 0xFFFF)), p1 = n1 * hbits((uint16_t)(ww[e] >> 16));
            o[e] = (u32)bitsh(p0) | ((u32)bitsh(p1) << 16);
        }
        orow[u] = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// qkv [S][(H + 2 Hkv) HD] -> q_out [H][S][HD] (rotated), k_cache / v_cache [Hkv][max_seq][HD] at pos[s] (k rotated).
// One thread per (token, head of q | k | v, 8 elements of the FIRST half): it owns elements d .. d + 7 and d + HD / 2 ..
__global__ void __launch_bounds__(256) rope_cache_rows_kernel(const uint16_t *__restrict__ qkv, const int *__restrict__ pos,
                                                              const uint16_t *__restrict__ cos_t, const uint16_t *__restrict__ sin_t,
                                                              uint16_t *__restrict__ q_out, uint16_t *__restrict__ kc, uint16_t *__restrict__ vc,
                                                              u32 S, u32 H, u32 Hkv, u32 HD, u32 max_seq) {
    const u32 upr = HD / 16u, heads = H + 2u * Hkv, total = S * heads * upr;  // units of 8 in the first half
    for (u32 i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const u32 u = i % upr, hh = (i / upr) % heads, s = i / (upr * heads), d = 8u * u;
        const u32 p = (u32)pos[s];
        const uint16_t *src = qkv + (size_t)s * heads * HD + (size_t)hh * HD;
        const uint4 a = *reinterpret_cast<const uint4 *>(src + d), b = *reinterpret_cast<const uint4 *>(src + d + HD / 2u);
        if (hh >= H + Hkv) {  // v: copied
            if (p < max_seq) {
                uint16_t *dst = vc + ((size_t)(hh - H - Hkv) * max_seq + p) * HD;
                *reinterpret_cast<uint4 *>(dst + d) = a;
                *reinterpret_cast<uint4 *>(dst + d + HD / 2u) = b;
            }
            continue;
        }
        const uint16_t *cr = cos_t + (size_t)min(p, max_seq - 1u) * HD, *sr = sin_t + (size_t)min(p, max_seq - 1u) * HD;
        const uint4 c1 = *reinterpret_cast<const uint4 *>(cr + d), c2 = *reinterpret_cast<const uint4 *>(cr + d + HD / 2u);
        const uint4 s1 = *reinterpret_cast<const uint4 *>(sr + d), s2 = *reinterpret_cast<const uint4 *>(sr + d + HD / 2u);
        const u32 x1[4] = {a.x, a.y, a.z, a.w}, x2[4] = {b.x, b.y, b.z, b.w};
        const u32 cc1[4] = {c1.x, c1.y, c1.z, c1.w}, cc2[4] = {c2.x, c2.y, c2.z, c2.w}, ss1[4] = {s1.x, s1.y, s1.z, s1.w}, ss2[4] = {s2.x, s2.y, s2.z, s2.w};
        u32 o1[4], o2[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            u32 r1 = 0, r2 = 0;
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const u32 sh = 16u * (u32)k;
                const h16 a1 = hbits((uint16_t)(x1[e] >> sh)), a2 = hbits((uint16_t)(x2[e] >> sh));
                // first half:  x1 * cos + (-x2) * sin ;  second half:  x2 * cos + x1 * sin   (rotate_half = cat(-x2, x1))
                const h16 f = (a1 * hbits((uint16_t)(cc1[e] >> sh))) + ((-a2) * hbits((uint16_t)(ss1[e] >> sh)));
                const h16 g = (a2 * hbits((uint16_t)(cc2[e] >> sh))) + (a1 * hbits((uint16_t)(ss2[e] >> sh)));
                r1 |= (u32)bitsh(f) << sh;
                r2 |= (u32)bitsh(g) << sh;
            }
            o1[e] = r1, o2[e] = r2;
        }
        uint16_t *dst;
        if (hh < H) {
            dst = q_out + ((size_t)hh * S + s) * HD;
        } else {
            if (p >= max_seq) continue;
            dst = kc + ((size_t)(hh - H) * max_seq + p) * HD;
        }
        *reinterpret_cast<uint4 *>(dst + d) = make_uint4(o1[0], o1[1], o1[2], o1[3]);
        *reinterpret_cast<uint4 *>(dst + d + HD / 2u) = make_uint4(o2[0], o2[1], o2[2], o2[3]);
    }
}

// y [S][2 I] -> out [S][I]; paired: (gate_i, up_i) adjacent (the decode step's row order), else gate = y[:, :I], up = y[:, I:]
__global__ void __launch_bounds__(256) silu_mul_rows_kernel(const uint16_t *__restrict__ y, uint16_t *__restrict__ out, u32 S, u32 I, u32 paired) {
    const u32 upr = I / 8u, total = S * upr;
    for (u32 i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const u32 u = i % upr, s = i / upr;
        const uint16_t *row = y + (size_t)s * 2u * I;
        u32 gw[4], uw[4];
        if (paired) {
            const uint4 a = *reinterpret_cast<const uint4 *>(row + 16u * u), b = *reinterpret_cast<const uint4 *>(row + 16u * u + 8u);
            const u32 pw[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};  // word = gate | up << 16
#pragma unroll
            for (int e = 0; e < 4; e++) {
                gw[e] = (pw[2 * e] & 0xFFFFu) | (pw[2 * e + 1] << 16);
                uw[e] = (pw[2 * e] >> 16) | (pw[2 * e + 1] & 0xFFFF0000u);
            }
        } else {
            const uint4 a = *reinterpret_cast<const uint4 *>(row + 8u * u), b = *reinterpret_cast<const uint4 *>(row + I + 8u * u);
            gw[0] = a.x, gw[1] = a.y, gw[2] = a.z, gw[3] = a.w;
            uw[0] = b.x, uw[1] = b.y, uw[2] = b.z, uw[3] = b.w;
        }
        u32 o[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            u32 r = 0;
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const u32 sh = 16u * (u32)k;
                const float gf = h2f((uint16_t)(gw[e] >> sh));
                const h16 sl = (h16)(gf / (1.0f + expf(-gf)));  // F.silu on fp16: computed in fp32, rounded
                r |= (u32)bitsh(sl * hbits((uint16_t)(uw[e] >> sh))) << sh;
            }
            o[e] = r;
        }
        *reinterpret_cast<uint4 *>(out + (size_t)s * I + 8u * u) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}
}  // namespace

extern "C" int gq_rmsnorm_rows(void *x, const void *delta, const void *weight, void *out, uint32_t S, uint32_t D, float eps, void *stream) {
    if (!x || !weight || !out) return gq_fail(GQ_EINVAL, "null pointer argument.");
    if ((uintptr_t)delta & 15u) return gq_fail(GQ_EINVAL, "gq_rmsnorm_rows: 16-byte aligned pointers.");
    if (S == 0) return GQ_OK;
    if (D == 0 || D % 8u || D > 16384u) return gq_fail(GQ_ENOTSUP, "gq_rmsnorm_rows: D must be a multiple of 8, <= 16384.");
    if (((uintptr_t)x | (uintptr_t)weight | (uintptr_t)out) & 15u) return gq_fail(GQ_EINVAL, "gq_rmsnorm_rows: 16-byte aligned pointers.");
    hipLaunchKernelGGL(rmsnorm_rows_kernel, dim3(S), dim3(256), 0, (hipStream_t)stream, (uint16_t *)x, (const uint16_t *)delta, (const uint16_t *)weight, (uint16_t *)out, D, eps);
    GQ_HIP_CHECK(hipGetLastError());
    return GQ_OK;
}

extern "C" int gq_rope_cache_rows(const void *qkv, const int *pos, const void *cos_table, const void *sin_table, void *q_out, void *k_cache,
                                  void *v_cache, uint32_t S, uint32_t n_head, uint32_t n_kv_head, uint32_t head_dim, uint32_t max_seq, void *stream) {
    if (!qkv || !pos || !cos_table || !sin_table || !q_out || !k_cache || !v_cache) return gq_fail(GQ_EINVAL, "null pointer argument.");
    if (S == 0) return GQ_OK;
    if (head_dim == 0 || head_dim % 16u || n_head == 0 || n_kv_head == 0 || max_seq == 0) return gq_fail(GQ_ENOTSUP, "gq_rope_cache_rows: head_dim must be a multiple of 16.");
    if (((uintptr_t)qkv | (uintptr_t)cos_table | (uintptr_t)sin_table | (uintptr_t)q_out | (uintptr_t)k_cache | (uintptr_t)v_cache) & 15u)
        return gq_fail(GQ_EINVAL, "gq_rope_cache_rows: 16-byte aligned pointers.");
    const uint64_t total = (uint64_t)S * (n_head + 2u * n_kv_head) * (head_dim / 16u);
    if (total >= 0x7FFFFFFFull) return gq_fail(GQ_ENOTSUP, "gq_rope_cache_rows: problem too large.");
    const u32 blocks = (u32)((total + 255u) / 256u);
    hipLaunchKernelGGL(rope_cache_rows_kernel, dim3(blocks > 65535u ? 65535u : blocks), dim3(256), 0, (hipStream_t)stream, (const uint16_t *)qkv, pos,
                       (const uint16_t *)cos_table, (const uint16_t *)sin_table, (uint16_t *)q_out, (uint16_t *)k_cache, (uint16_t *)v_cache, S, n_head,
                       n_kv_head, head_dim, max_seq);
    GQ_HIP_CHECK(hipGetLastError());
    return GQ_OK;
}

extern "C" int gq_silu_mul_rows(const void *y, void *out, uint32_t S, uint32_t inter, int paired, void *stream) {
    if (!y || !out) return gq_fail(GQ_EINVAL, "null pointer argument.");
    if (S == 0) return GQ_OK;
    if (inter == 0 || inter % 8u) return gq_fail(GQ_ENOTSUP, "gq_silu_mul_rows: the intermediate size must be a multiple of 8.");
    if (((uintptr_t)y | (uintptr_t)out) & 15u) return gq_fail(GQ_EINVAL, "gq_silu_mul_rows: 16-byte aligned pointers.");
    const uint64_t total = (uint64_t)S * (inter / 8u);
    if (total >= 0x7FFFFFFFull) return gq_fail(GQ_ENOTSUP, "gq_silu_mul_rows: problem too large.");
    const u32 blocks = (u32)((total + 255u) / 256u);
    hipLaunchKernelGGL(silu_mul_rows_kernel, dim3(blocks > 65535u ? 65535u : blocks), dim3(256), 0, (hipStream_t)stream, (const uint16_t *)y, (uint16_t *)out, S,
                       inter, (u32)(paired != 0));
    GQ_HIP_CHECK(hipGetLastError());
    return GQ_OK;
}
