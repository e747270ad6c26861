// ap_plane.hip -- "fast mode" Any-Precision GEMV for gfx950: binary bit-plane GEMVs on the matrix cores.
//
// See plane_core.h for the algorithm.  What runs where:
//   HBM    : the stored bit-planes, 128 contiguous bytes per row per plane per 1024-weight chunk
//            (lane (row r, k-block kb) loads 32 B; 4 lanes of a row cover one cache line), register ring of D chunks.
//   VALU   : one v_and_b32 per 4 weights per plane-subset (bits -> bf8 {0, 2^e}); ANDs of planes for the subsets.
//   MFMA   : v_mfma_scale_f32_16x16x128_f8f6f4 (A = bf8 bit patterns, scale cancels 2^e; B = 4 bf8 pieces of x in
//            columns 0..3, per-32-element block scale), fp32 accumulate: 8 MFMAs per (chunk, plane-subset).
//   LDS    : the B image (activation pieces, 4*K bytes) built once per block, partial sums of K-split items.
// Epilogue : y = coef[0] * sum(x) + sum_S coef[S] * T[S]  (Moebius coefficients of the row's LUT), fp32 -> fp16.
//
// This path is NOT bit-identical to the reference's fp16-accumulated kernel (anyprec.cu:372-542): it is closer
// to the exact product than the reference is (products exact, fp32 accumulation).  The bit-exact path is
// ap_gemv.hip; gq_set_ap_mode() / GQ_AP_MODE choose.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gq_internal.h"
#include "plane_core.h"

using namespace gqp;

namespace {

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef u32 u32x4 __attribute__((ext_vector_type(4)));

struct PlaneArgs {
    const u32 *qw;
    const uint16_t *lut;
    const uint16_t *x;
    uint16_t *out;
    const uint16_t *normw;
    const uint16_t *resid;
    u32 N, K;
    u32 RGB;     // row groups (16 rows) per block
    u32 log2CS;  // a row group's chunks are split over 2^log2CS wave items
    u32 cpi;     // chunks per item
    float eps;
};

enum { PRO_NONE = 0, PRO_RMSNORM = 1, PRO_SILUMUL = 2 };

__device__ __forceinline__ float h2f(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
__device__ __forceinline__ uint16_t f2h(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }

template <int PRO>
__device__ __forceinline__ float load_x(const uint16_t *x, const uint16_t *normw, u32 K, u32 e, float nscale) {
    if constexpr (PRO == PRO_RMSNORM) {
        // (x.float() * rsqrt(mean(x^2)+eps)).half() * w   -- inference/model.py:281-292, both fp16 roundings kept
        _Float16 a = (_Float16)(h2f(x[e]) * nscale);
        _Float16 r = a * __builtin_bit_cast(_Float16, normw[e]);
        return (float)r;
    } else if constexpr (PRO == PRO_SILUMUL) {
        // F.silu(gate) * up on fp16 tensors -- inference/model.py:266
        float g = h2f(x[e]);
        _Float16 s = (_Float16)(g / (1.0f + __expf(-g)));
        _Float16 r = s * __builtin_bit_cast(_Float16, x[K + e]);
        return (float)r;
    } else {
        return h2f(x[e]);
    }
}

template <int BITS, int D, int PRO>
__global__ void __launch_bounds__(512) ap_plane_kernel(PlaneArgs a) {
    constexpr int NP = 1 << BITS, NP1 = NP - 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    Geom G;
    G.init(a.K);
    const u32 T = blockDim.x, tid = threadIdx.x;
    const u32 W = T >> 6;
    const u32 w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const u32 l = tid & 63u;
    unsigned char *bimg = smem;                            // [chunk][s][kb][piece][32]
    unsigned char *bscale = bimg + G.nchunks * 4096u;      // [chunk][kb][s]  E8M0 in 16-bit slots (byte select 3 is unusable)
    unsigned char *zero32 = bscale + G.nchunks * 64u;      // 32 zero bytes
    float *red = reinterpret_cast<float *>(zero32 + 32);   // 32 floats
    float *part = red + 32;                                // [RGB << log2CS][16][NP1]

    const u32 CS = 1u << a.log2CS, cpi = a.cpi;
    const u32 nIt = a.RGB * CS;  // wave items of this block
    const u32 rg0 = blockIdx.x * a.RGB;
    const u32 m = blockIdx.y;
    const u32 r = l & 15u, kb = l >> 4;

    // ---------------------------------------------------------------- plane ring: put HBM requests in flight first
    constexpr u32 OOB = 0x80000000u;
    const u32 plane_bytes = a.N * G.wpr * 4u;
    __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void *)a.qw, 0, (int)(plane_bytes * (u32)BITS), 0x00020000);
    u32x4 P[D][BITS][2];
    u32 iq_item = w, iq_c = 0;
    auto issue = [&](int d) {
        const u32 chunk = (iq_item & (CS - 1u)) * cpi + iq_c;
        const u32 row = (rg0 + (iq_item >> a.log2CS)) * 16u + r;
        const bool ok = iq_item < nIt && chunk < G.nchunks && row < a.N && 8u * kb < G.tpw(chunk);
        const u32 off = (row * G.wpr + 32u * chunk + 8u * kb) * 4u;
#pragma unroll
        for (int p = 0; p < BITS; p++) {
            P[d][p][0] = __builtin_amdgcn_raw_buffer_load_b128(rq, ok ? off + (u32)p * plane_bytes : OOB, 0, 2);
            P[d][p][1] = __builtin_amdgcn_raw_buffer_load_b128(rq, ok ? off + (u32)p * plane_bytes + 16u : OOB, 0, 2);
        }
        if (++iq_c == cpi) {
            iq_c = 0;
            iq_item += W;
        }
    };
#pragma unroll
    for (int d = 0; d < D; d++) issue(d);

    // ---------------------------------------------------------------- prologue: activation pieces -> LDS image
    const uint16_t *x = a.x + (size_t)m * (PRO == PRO_SILUMUL ? 2u * G.K : G.K);
    float nscale = 0.f;
    if constexpr (PRO == PRO_RMSNORM) {
        float ss = 0.f;
        for (u32 g = tid; g < G.K / 8u; g += T) {
            uint4 v = *reinterpret_cast<const uint4 *>(x + 8u * g);
            const u32 ww[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int i = 0; i < 4; i++) {
                float p = h2f(ww[i] & 0xFFFF), q = h2f(ww[i] >> 16);
                ss += p * p;
                ss += q * q;
            }
        }
#pragma unroll
        for (int sh = 32; sh >= 1; sh >>= 1) ss += __shfl_xor(ss, sh, 64);
        if (l == 0) red[16 + w] = ss;
        __syncthreads();
        float tot = 0.f;
        for (u32 i = 0; i < W; i++) tot += red[16 + i];
        nscale = 1.0f / sqrtf(tot / (float)G.K + a.eps);
    }
    float xsum = 0.f;
    for (u32 it = tid; it < G.nchunks * 256u; it += T) {
        // item bits: [1:0] v&3, [2] kb&1, [5:3] j, [6] v>>2, [7] kb>>1, [8+] chunk  -> the 8 lanes that differ in bits
        // 0..2 hold one hardware scale block (blk_of) together with their 4 bytes c
        const u32 v = (it & 3u) | (((it >> 6) & 1u) << 2), kbi = ((it >> 2) & 1u) | (((it >> 7) & 1u) << 1);
        const u32 j = (it >> 3) & 7u, chunk = it >> 8;
        const u32 s = 7u - j, t = 8u * kbi + v, tp = G.tpw(chunk);
        float xe[4];
        float mx = 0.f;
#pragma unroll
        for (u32 c = 0; c < 4; c++) {
            const u32 e = 1024u * chunk + 8u * tp * c + 8u * t + j;
            xe[c] = t < tp ? load_x<PRO>(x, a.normw, G.K, e, nscale) : 0.f;
            xsum += xe[c];
            mx = fmaxf(mx, fabsf(xe[c]));
        }
        mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 4, 64));
        const int eb = mx > 0.f ? (int)((__builtin_bit_cast(u32, mx) >> 23) & 0xFFu) - 126 : 15;  // mx = f * 2^eb, f in [.5,1)
        const float sc = __builtin_bit_cast(float, (u32)(15 - eb + 127) << 23);                      // 2^(15-eb): |x*sc| < 2^15
        if ((it & 7u) == 0)  // scale of block b is read by lane group kb == b
            reinterpret_cast<uint16_t *>(bscale)[chunk * 32u + blk_of(kbi, v) * 8u + s] = (uint16_t)(127 + eb - 15);
        float xs[4] = {xe[0] * sc, xe[1] * sc, xe[2] * sc, xe[3] * sc};
#pragma unroll
        for (u32 p = 0; p < 4; p++) {
            // bytes B = 0..3 hold c = 3..0
            int wd = __builtin_amdgcn_cvt_pk_bf8_f32(xs[3], xs[2], 0, false);
            wd = __builtin_amdgcn_cvt_pk_bf8_f32(xs[1], xs[0], wd, true);
            *reinterpret_cast<int *>(bimg + bimg_off(chunk, s, kbi, p) + 4u * v) = wd;
            if (p < 3) {
                xs[3] -= __builtin_amdgcn_cvt_f32_bf8(wd, 0);
                xs[2] -= __builtin_amdgcn_cvt_f32_bf8(wd, 1);
                xs[1] -= __builtin_amdgcn_cvt_f32_bf8(wd, 2);
                xs[0] -= __builtin_amdgcn_cvt_f32_bf8(wd, 3);
            }
        }
    }
#pragma unroll
    for (int sh = 32; sh >= 1; sh >>= 1) xsum += __shfl_xor(xsum, sh, 64);
    if (l == 0) red[w] = xsum;
    if (tid < 8) reinterpret_cast<u32 *>(zero32)[tid] = 0u;
    __syncthreads();
    float X = 0.f;
    for (u32 i = 0; i < W; i++) X += red[i];

    // ---------------------------------------------------------------- main loop: (item, chunk) steps of this wave
    const u32 items_w = nIt > w ? (nIt - w + W - 1u) / W : 0u;
    const u32 my_steps = items_w * cpi;
    const u32 col = l & 15u;
    const bool bcol = col < 4u;
    v4f acc[NP1];
#pragma unroll
    for (int i = 0; i < NP1; i++) acc[i] = (v4f){0.f, 0.f, 0.f, 0.f};
    u32 cq_item = w, cq_c = 0;

    for (u32 q = 0; q < my_steps; q += D) {
#pragma unroll
        for (int d = 0; d < D; d++) {
            if (q + (u32)d < my_steps) {
                const u32 chunk = (cq_item & (CS - 1u)) * cpi + cq_c;
                u32 Wd[BITS][8];
#pragma unroll
                for (int p = 0; p < BITS; p++)
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        Wd[p][4 * h + 0] = P[d][p][h].x;
                        Wd[p][4 * h + 1] = P[d][p][h].y;
                        Wd[p][4 * h + 2] = P[d][p][h].z;
                        Wd[p][4 * h + 3] = P[d][p][h].w;
                    }
                issue(d);
                if (chunk < G.nchunks) {
                    // plane-subset words: code bit i lives in plane BITS-1-i
                    u32 PW[NP1][8];
#pragma unroll
                    for (int cm = 1; cm < NP; cm++) {
                        const int low = cm & -cm, i0 = __builtin_ctz(cm), rest = cm ^ low;
#pragma unroll
                        for (int v = 0; v < 8; v++)
                            PW[cm - 1][v] = rest ? (PW[rest - 1][v] & Wd[BITS - 1 - i0][v]) : Wd[BITS - 1 - i0][v];
                    }
                    const uint4 sbw = *reinterpret_cast<const uint4 *>(bscale + (chunk * 32u + kb * 8u) * 2u);
                    const u32 sbws[4] = {sbw.x, sbw.y, sbw.z, sbw.w};
#pragma unroll
                    for (int s = 0; s < 8; s++) {
                        const unsigned char *bsrc = bcol ? bimg + bimg_off(chunk, (u32)s, kb, col) : zero32;
                        const uint4 b0 = *reinterpret_cast<const uint4 *>(bsrc);
                        const uint4 b1 = *reinterpret_cast<const uint4 *>(bsrc + 16);
                        v8i Bv = {(int)b0.x, (int)b0.y, (int)b0.z, (int)b0.w, (int)b1.x, (int)b1.y, (int)b1.z, (int)b1.w};
                        const int sb = (int)sbws[s >> 1];
#pragma unroll
                        for (int cm = 1; cm < NP; cm++) {
                            v8i Av;
#pragma unroll
                            for (int v = 0; v < 8; v++) Av[v] = (int)extract(PW[cm - 1][v], s);
                            if (s & 1)
                                acc[cm - 1] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(Av, Bv, acc[cm - 1], 1, 1, 0, scale_byte(s), 2, sb);
                            else
                                acc[cm - 1] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(Av, Bv, acc[cm - 1], 1, 1, 0, scale_byte(s), 0, sb);
                        }
                    }
                }
                if (++cq_c == cpi) {
                    // item done: add the 4 piece columns, park the 16 x NP1 sums in LDS
#pragma unroll
                    for (int cm = 0; cm < NP1; cm++)
#pragma unroll
                        for (int rg = 0; rg < 4; rg++) {
                            float v = acc[cm][rg];
                            v += __shfl_xor(v, 1, 64);
                            v += __shfl_xor(v, 2, 64);
                            if (col == 0) part[((size_t)cq_item * 16u + 4u * kb + (u32)rg) * NP1 + cm] = v;
                            acc[cm][rg] = 0.f;
                        }
                    cq_c = 0;
                    cq_item += W;
                }
            }
        }
    }
    __syncthreads();

    // ---------------------------------------------------------------- epilogue: coefficients x plane sums
    for (u32 i = tid; i < a.RGB * 16u; i += T) {
        const u32 rgl = i >> 4, rr = i & 15u;
        const u32 row = (rg0 + rgl) * 16u + rr;
        if (row >= a.N) continue;
        float f[NP];
#pragma unroll
        for (int c = 0; c < NP; c++) f[c] = h2f(a.lut[(size_t)row * NP + c]);
        moebius<BITS>(f);
        float y = f[0] * X;
#pragma unroll
        for (int cm = 1; cm < NP; cm++) {
            float tsum = 0.f;
            for (u32 cs = 0; cs < CS; cs++) tsum += part[((size_t)((rgl << a.log2CS) + cs) * 16u + rr) * NP1 + (cm - 1)];
            y += f[cm] * tsum;
        }
        _Float16 yh = (_Float16)y;
        if (a.resid) yh = __builtin_bit_cast(_Float16, a.resid[(size_t)m * a.N + row]) + yh;
        a.out[(size_t)m * a.N + row] = __builtin_bit_cast(uint16_t, yh);
    }
}

struct PlaneCfg {
    u32 grid, T, RGB, log2CS, cpi;
    int D;
    size_t smem;
};

int g_cus = 0;
int cus() {
    if (!g_cus) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
            g_cus = n;
        else
            g_cus = 256;
    }
    return g_cus;
}

bool pick_plane_cfg(u32 N, u32 K, int bits, PlaneCfg &c) {
    if (K % 256u) return false;
    const u32 nchunks = K / 1024u + ((K % 1024u) ? 1u : 0u);
    const u32 RGt = (N + 15u) / 16u;
    const u32 ncu = (u32)cus();
    // split K of a row group over 2^log2CS wave items until there are ~1.5 items per SIMD
    u32 lcs = 0;
    const u32 want = (u32)gq_env_int("GQ_PL_ITEMS", (int)(ncu * 6u));
    while ((RGt << lcs) < want && (2u << lcs) <= nchunks) lcs++;
    const int envcs = gq_env_int("GQ_PL_LOG2CS", -1);
    if (envcs >= 0 && (1u << envcs) <= nchunks) lcs = (u32)envcs;
    c.log2CS = lcs;
    c.cpi = (nchunks + (1u << lcs) - 1u) >> lcs;
    u32 W = (u32)gq_env_int("GQ_PL_WAVES", 8);
    if (W < 1 || W > 8) W = 8;
    c.T = 64u * W;
    // row groups per block: one block per CU when the matrix is big enough, never more items than ~2 per wave
    u32 rgb = (RGt + ncu - 1u) / ncu;
    const u32 envbpc = (u32)gq_env_int("GQ_PL_BPC", 1);
    if (envbpc > 1) rgb = (RGt + ncu * envbpc - 1u) / (ncu * envbpc);
    if (rgb < 1) rgb = 1;
    while (rgb > 1 && (rgb << lcs) > 2u * W) rgb--;
    c.RGB = rgb;
    c.grid = (RGt + rgb - 1u) / rgb;
    int d = gq_env_int("GQ_PL_D", 0);
    if (d < 1 || d > 4) d = bits == 2 ? 4 : (bits == 3 ? 2 : 1);
    c.D = d;
    const u32 np1 = (1u << bits) - 1u;
    c.smem = (size_t)nchunks * 4096u + nchunks * 64u + 32u + 32u * 4u + (size_t)(rgb << lcs) * 16u * np1 * 4u;
    return c.smem <= 160u * 1024u;
}

template <int BITS, int D, int PRO>
int launch_plane_inst(const PlaneArgs &a, const PlaneCfg &c, u32 M, hipStream_t s) {
    static bool attr_set = false;
    auto kern = ap_plane_kernel<BITS, D, PRO>;
    if (c.smem > 48u * 1024u && !attr_set) {
        GQ_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)(160u * 1024u)));
        attr_set = true;
    }
    dim3 grid(c.grid, M), block(c.T);
    hipLaunchKernelGGL(kern, grid, block, c.smem, s, a);
    GQ_HIP_CHECK(hipGetLastError());
    return GQ_OK;
}

template <int BITS, int PRO>
int launch_plane_d(const PlaneArgs &a, const PlaneCfg &c, u32 M, hipStream_t s) {
    if constexpr (BITS == 4) {
        return launch_plane_inst<BITS, 1, PRO>(a, c, M, s);
    } else if constexpr (BITS == 3) {
        switch (c.D) {
            case 1: return launch_plane_inst<BITS, 1, PRO>(a, c, M, s);
            default: return launch_plane_inst<BITS, 2, PRO>(a, c, M, s);
        }
    } else {
        switch (c.D) {
            case 1: return launch_plane_inst<BITS, 1, PRO>(a, c, M, s);
            case 2: return launch_plane_inst<BITS, 2, PRO>(a, c, M, s);
            default: return launch_plane_inst<BITS, 4, PRO>(a, c, M, s);
        }
    }
}

template <int BITS>
int launch_plane(const PlaneArgs &a, const PlaneCfg &c, u32 M, int pro, hipStream_t s) {
    switch (pro) {
        case PRO_RMSNORM: return launch_plane_d<BITS, PRO_RMSNORM>(a, c, M, s);
        case PRO_SILUMUL: return launch_plane_d<BITS, PRO_SILUMUL>(a, c, M, s);
        default: return launch_plane_d<BITS, PRO_NONE>(a, c, M, s);
    }
}

}  // namespace

// returns GQ_ENOTSUP when the shape is not served by this path (caller falls back to the exact kernels)
int gq_plane_gemv_try(const void *x, void *out, const uint32_t *qweight, const void *lut, uint32_t M, uint32_t N, uint32_t K,
                      int bits, const void *normw, float eps, const void *resid, int pro, hipStream_t stream) {
    if (bits < 2 || bits > 4) return GQ_ENOTSUP;
    const uint64_t qbytes = (uint64_t)bits * N * (K / 8u);
    if (qbytes >= 0x7FFFFFFFull) return GQ_ENOTSUP;
    if (((uintptr_t)qweight | (uintptr_t)x | (uintptr_t)normw) & 15u) return GQ_ENOTSUP;
    PlaneCfg c;
    if (!pick_plane_cfg(N, K, bits, c)) return GQ_ENOTSUP;
    PlaneArgs a{};
    a.qw = qweight;
    a.lut = (const uint16_t *)lut;
    a.x = (const uint16_t *)x;
    a.out = (uint16_t *)out;
    a.normw = (const uint16_t *)normw;
    a.resid = (const uint16_t *)resid;
    a.N = N;
    a.K = K;
    a.RGB = c.RGB;
    a.log2CS = c.log2CS;
    a.cpi = c.cpi;
    a.eps = eps;
    switch (bits) {
        case 2: return launch_plane<2>(a, c, M, pro, stream);
        case 3: return launch_plane<3>(a, c, M, pro, stream);
        default: return launch_plane<4>(a, c, M, pro, stream);
    }
}
