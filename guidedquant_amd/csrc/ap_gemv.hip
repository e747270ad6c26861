// ap_gemv.hip -- Any-Precision LUT-GEMV / dequant kernels for gfx950 (MI355X) and their launchers.
//
// Replaces inference/ap_gemv/anyprec.cu (matmul_kbit_32, dequant_kbit_store) of the reference.  Not a
// translation: the reference is a warp-32 program with an LDS pair table; this is a wave-64 program that
//   * streams the bit-planes with one 16-byte load per lane per plane ("quad" = 4 of the reference's lanes),
//     RT rows in flight per lane, every row of a wave contiguous in memory (1 KiB per wave-load),
//   * decodes with v_perm_b32 byte lookups out of VGPR-resident LUT pools (no LDS on the weight path),
//   * keeps the activations of the lane's quad in 64 VGPRs for all its rows (staged once through LDS in a
//     bank-conflict-free, lane-linear image),
//   * accumulates in packed fp16 FMAs in EXACTLY the reference's per-lane order, so results are
//     bit-identical to the CUDA kernel (see ap_core.h and DESIGN.md),
//   * reduces through LDS in the reference's chunk order + 16/8/4/2/1 tree.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ap_core.h"
#include "gq_internal.h"

using namespace gq;

namespace {

struct ApArgs {
    const u32 *qw;         // [bits][N][K/32]
    const uint16_t *lut;   // [N][2^bits]
    const uint16_t *x;     // [M][K]
    uint16_t *out;         // [M][N] (or [M][N/2] with SILU_MUL)
    const uint16_t *normw; // [K] or null
    const uint16_t *resid; // [N] or null
    u32 N, K;
    u32 RS;                // row slots per block step = blockDim.x / Q
    u32 epilogue;
    float eps;
};

__device__ __forceinline__ uint4 ld16(const void *p) { return *reinterpret_cast<const uint4 *>(p); }
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ld16_nt(const void *p) {
    u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}

__device__ __forceinline__ float h2f(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
__device__ __forceinline__ uint16_t f2h(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }

// ----------------------------------------------------------------------------------------------
// Stage the activation vector into LDS in the lane-linear image of ap_core.h::xlds_pos.
// Optional fused RMSNorm prologue (inference/model.py:281-292): y = (x.float() * rsqrt(mean(x^2)+eps))
// .half() * w  -- the same two fp16 roundings as the reference's separate kernel.
// ----------------------------------------------------------------------------------------------
__device__ __forceinline__ void stage_x(const RowGeom &G, const uint16_t *x, const uint16_t *normw, float eps,
                                        uint16_t *xlds, float *red /* >= 16 floats of LDS */) {
    const u32 T = blockDim.x, tid = threadIdx.x;
    float scale = 0.f;
    if (normw) {
        float ss = 0.f;
        for (u32 g = tid; g < G.K / 8u; g += T) {
            uint4 v = ld16(x + 8u * g);
            const u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int i = 0; i < 4; i++) {
                float a = h2f(w[i] & 0xFFFF), b = h2f(w[i] >> 16);
                ss += a * a;
                ss += b * b;
            }
        }
#pragma unroll
        for (int sh = 32; sh >= 1; sh >>= 1) ss += __shfl_xor(ss, sh, 64);
        if ((tid & 63u) == 0) red[tid >> 6] = ss;
        __syncthreads();
        float tot = 0.f;
        for (u32 w = 0; w < (T + 63u) / 64u; w++) tot += red[w];
        scale = 1.0f / sqrtf(tot / (float)G.K + eps);
    }
    for (u32 g = tid; g < G.K / 8u; g += T) {
        uint4 v = ld16(x + 8u * g);
        u32 w[4] = {v.x, v.y, v.z, v.w};
        if (normw) {
            uint4 nv = ld16(normw + 8u * g);
            const u32 nw[4] = {nv.x, nv.y, nv.z, nv.w};
#pragma unroll
            for (int i = 0; i < 4; i++) {
                uint16_t a = f2h(h2f(w[i] & 0xFFFF) * scale), b = f2h(h2f(w[i] >> 16) * scale);
                _Float16 ra = __builtin_bit_cast(_Float16, a) * __builtin_bit_cast(_Float16, (uint16_t)(nw[i] & 0xFFFF));
                _Float16 rb = __builtin_bit_cast(_Float16, b) * __builtin_bit_cast(_Float16, (uint16_t)(nw[i] >> 16));
                w[i] = (u32)__builtin_bit_cast(uint16_t, ra) | ((u32)__builtin_bit_cast(uint16_t, rb) << 16);
            }
        }
        u32 q, vv, c;
        xgroup(G, g, q, vv, c);
#pragma unroll
        for (u32 j = 0; j < 8; j++)
            xlds[xlds_pos(G.Q, q, vv, c, j)] = (uint16_t)((w[j >> 1] >> (16 * (j & 1))) & 0xFFFF);
    }
}

// ----------------------------------------------------------------------------------------------
// Final reduction of one row from its per-(chunk, virtual lane) fp16 partials in LDS:
// chunks in ascending order per lane (anyprec.cu:505), then the 16,8,4,2,1 shuffle tree (anyprec.cu:363-370).
// Called by 32 consecutive lanes (t = lane & 31).  Returns the row value in lane t == 0.
// ----------------------------------------------------------------------------------------------
__device__ __forceinline__ uint16_t reduce_row(const RowGeom &G, const uint16_t *sv, u32 t, u32 c0, u32 c1) {
    uint16_t p = 0;
    for (u32 i = c0; i < c1; i++) {
        if (i == G.nfull && t >= G.eff) break;
        p = h_add(p, sv[i * 32u + t]);
    }
#pragma unroll
    for (int sh = 16; sh >= 1; sh >>= 1) {
        uint16_t o = (uint16_t)__shfl_down((int)p, sh, 32);
        p = h_add(p, o);
    }
    return p;
}

__device__ __forceinline__ uint16_t silu_mul_h(uint16_t g, uint16_t u) {
    // F.silu(w1_out) * w3_out on fp16 tensors (inference/model.py:266): silu evaluated in fp32 and rounded to
    // fp16 (ATen's half silu kernel computes x / (1 + exp(-x)) in float), then an fp16 multiply.
    float x = h2f(g);
    _Float16 s = (_Float16)(x / (1.0f + __expf(-x)));
    _Float16 r = s * __builtin_bit_cast(_Float16, u);
    return __builtin_bit_cast(uint16_t, r);
}

// ----------------------------------------------------------------------------------------------
// Fast path: BITS in {2,3,4}, K % 128 == 0, Q = K/128 <= blockDim.x.
// ----------------------------------------------------------------------------------------------
template <int BITS, int RT, bool NT>
__global__ void __launch_bounds__(512) ap_gemv_quad_kernel(ApArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    RowGeom G;
    G.init(a.K);
    uint16_t *xlds = reinterpret_cast<uint16_t *>(smem);
    uint16_t *sv = xlds + G.K;  // [RB][nchunks*32]
    const u32 svrow = G.nchunks * 32u;
    float *red = reinterpret_cast<float *>(sv);  // scratch for the RMSNorm reduction (before sv is used)

    const u32 T = blockDim.x, tid = threadIdx.x;
    const u32 RS = a.RS, RB = RS * RT;
    const u32 row0 = blockIdx.x * RB;
    const u32 m = blockIdx.y;
    const bool active = tid < RS * G.Q;
    const u32 rs = active ? tid / G.Q : 0u;
    const u32 q = active ? tid - rs * G.Q : 0u;

    // 1. put every plane byte this lane will need in flight before touching anything else
    constexpr int NRAW = (1 << BITS) / 2;
    uint4 P[RT][BITS];
    u32 lraw[RT][NRAW];
#pragma unroll
    for (int it = 0; it < RT; it++) {
        const u32 row = row0 + (u32)it * RS + rs;
        const bool ok = active && row < a.N;
#pragma unroll
        for (int p = 0; p < BITS; p++) {
            const u32 *src = a.qw + ((size_t)p * a.N + row) * G.wpr + 4u * q;
            P[it][p] = ok ? (NT ? ld16_nt(src) : ld16(src)) : make_uint4(0, 0, 0, 0);
        }
        const u32 *lsrc = reinterpret_cast<const u32 *>(a.lut) + (size_t)row * NRAW;
        if constexpr (NRAW == 2) {
            uint2 v = ok ? *reinterpret_cast<const uint2 *>(lsrc) : make_uint2(0, 0);
            lraw[it][0] = v.x;
            lraw[it][1] = v.y;
        } else {
#pragma unroll
            for (int i = 0; i < NRAW; i += 4) {
                uint4 v = ok ? ld16(lsrc + i) : make_uint4(0, 0, 0, 0);
                lraw[it][i] = v.x;
                lraw[it][i + 1] = v.y;
                lraw[it][i + 2] = v.z;
                lraw[it][i + 3] = v.w;
            }
        }
    }

    // 2. activations -> LDS (lane-linear image) -> 64 VGPRs
    stage_x(G, a.x + (size_t)m * G.K, a.normw, a.eps, xlds, red);
    __syncthreads();
    XRegs xr;
#pragma unroll
    for (u32 c = 0; c < 4; c++)
#pragma unroll
        for (u32 jj = 0; jj < 4; jj++) {
            uint4 v = *reinterpret_cast<const uint4 *>(xlds + (((c * 4u + jj) * G.Q + q) << 3));
            xr.r[c][jj][0] = v.x;
            xr.r[c][jj][1] = v.y;
            xr.r[c][jj][2] = v.z;
            xr.r[c][jj][3] = v.w;
        }

    // 3. decode + packed fp16 FMA chains, one item per row
    u32 chunk, t0, tpw;
    G.quad(q, chunk, t0, tpw);
#pragma unroll
    for (int it = 0; it < RT; it++) {
        u32 Pw[BITS][4];
#pragma unroll
        for (int p = 0; p < BITS; p++) {
            Pw[p][0] = P[it][p].x;
            Pw[p][1] = P[it][p].y;
            Pw[p][2] = P[it][p].z;
            Pw[p][3] = P[it][p].w;
        }
        LutPools<BITS> L;
        L.build(lraw[it]);
        u32 s01, s23;
        Item<BITS>::run(Pw, L, xr, s01, s23);
        if (active) {
            const u32 rl = (u32)it * RS + rs;
            *reinterpret_cast<uint2 *>(sv + (size_t)rl * svrow + chunk * 32u + t0) = make_uint2(s01, s23);
        }
    }
    __syncthreads();

    // 4. ordered reduction + epilogue, 32 lanes per row
    const u32 t = tid & 31u;
    const bool silu = (a.epilogue & GQ_EPI_SILU_MUL) != 0;
    for (u32 rl = tid >> 5; rl < RB; rl += T >> 5) {
        const u32 row = row0 + rl;
        uint16_t y = reduce_row(G, sv + (size_t)rl * svrow, t, 0u, G.nchunks);
        if (silu) {
            sv[(size_t)rl * svrow] = y;  // park the row value for the pairing pass below (lane 0 only matters)
        } else if (t == 0 && row < a.N) {
            if (a.resid) y = h_add(a.resid[row], y);
            a.out[(size_t)m * a.N + row] = y;
        }
    }
    if (silu) {
        // fused gate/up: the launcher interleaves blocks so that local rows [0,RB/2) are gate rows i and
        // [RB/2,RB) the matching up rows I+i  (see launch_quad)
        __syncthreads();
        const u32 half_rb = RB / 2u, I = a.N / 2u;
        for (u32 i = tid; i < half_rb; i += T) {
            const u32 gi = blockIdx.x * half_rb + i;
            if (gi < I) a.out[(size_t)m * I + gi] = silu_mul_h(sv[(size_t)i * svrow], sv[(size_t)(half_rb + i) * svrow]);
        }
    }
}

// ----------------------------------------------------------------------------------------------
// Generic path: any 2 <= BITS <= 8, any K % 32 == 0.  32 lanes per row exactly like the reference warp
// (lane = virtual lane), LUT in LDS, 4-byte plane loads.  Slow; exists for API completeness (bits 5..8,
// odd K) and as an on-device cross-check of the fast path.
// ----------------------------------------------------------------------------------------------
template <int BITS>
__global__ void __launch_bounds__(256) ap_gemv_generic_kernel(ApArgs a, int ksplit) {
    __shared__ uint16_t lutl[8][1 << BITS];
    RowGeom G;
    G.init(a.K);
    const u32 tid = threadIdx.x, t = tid & 31u, g = tid >> 5;
    const u32 row = blockIdx.x * 8u + g;
    const u32 m = blockIdx.y;
    const bool ok = row < a.N;
    if (ok)
        for (u32 i = t; i < (1u << BITS); i += 32u) lutl[g][i] = a.lut[(size_t)row * (1u << BITS) + i];
    __syncthreads();
    const uint16_t *x = a.x + (size_t)m * G.K;
    const u32 ngroups = ksplit ? (G.nchunks + 3u) / 4u : 1u;
    uint16_t total = 0, result = 0;
    for (u32 grp = 0; grp < ngroups; grp++) {
        const u32 c0 = ksplit ? grp * 4u : 0u;
        const u32 c1 = ksplit ? min(c0 + 4u, G.nchunks) : G.nchunks;
        uint16_t partial = 0;
        for (u32 i = c0; i < c1 && ok; i++) {
            u32 tpw = 32u;
            if (i == G.nfull) {
                tpw = G.eff;
                if (t >= G.eff) break;
            }
            u32 P[BITS];
#pragma unroll
            for (int p = 0; p < BITS; p++) P[p] = a.qw[((size_t)p * a.N + row) * G.wpr + i * 32u + t];
            u32 acc = 0;  // (even chain, odd chain)
#pragma unroll
            for (int c = 3; c >= 0; c--) {
                const uint16_t *xs = x + 1024u * i + 8u * tpw * (u32)c + 8u * t;
                uint4 xv = ld16(xs);
                const u32 xw[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    u32 ce = 0, co = 0;
#pragma unroll
                    for (int p = 0; p < BITS; p++) {
                        ce = (ce << 1) | ((P[p] >> (31 - (8 * c + 2 * k))) & 1u);
                        co = (co << 1) | ((P[p] >> (30 - (8 * c + 2 * k))) & 1u);
                    }
                    u32 w = (u32)lutl[g][ce] | ((u32)lutl[g][co] << 16);
                    acc = pk_fma(w, xw[k], acc);
                }
            }
            partial = h_add(partial, h_add((uint16_t)(acc & 0xFFFF), (uint16_t)(acc >> 16)));
        }
#pragma unroll
        for (int sh = 16; sh >= 1; sh >>= 1) partial = h_add(partial, (uint16_t)__shfl_down((int)partial, sh, 32));
        if (ksplit)
            total = h_add(total, partial);
        else
            result = partial;
    }
    if (t == 0 && ok) {
        uint16_t y = ksplit ? total : result;
        if (a.resid) y = h_add(a.resid[row], y);
        a.out[(size_t)m * a.N + row] = y;
    }
}

// ----------------------------------------------------------------------------------------------
// Dequantise: W[n][e] = lut[n][code(n,e)]  (anyprec.cu:294-359).  One lane per plane word (32 weights):
// reads BITS words, writes four 16-byte runs (one per byte c) -- the reference's store pattern
// (anyprec.cu:355-357), 64 lanes wide.
// ----------------------------------------------------------------------------------------------
template <int BITS>
__global__ void __launch_bounds__(256) ap_dequant_kernel(const u32 *qw, const uint16_t *lut, uint16_t *W, u32 N, u32 K) {
    __shared__ uint16_t lutl[4][1 << BITS];
    RowGeom G;
    G.init(K);
    const u32 tid = threadIdx.x, g = tid >> 6, l = tid & 63u;
    const u32 row = blockIdx.x * 4u + g;
    const bool ok = row < N;
    if (ok)
        for (u32 i = l; i < (1u << BITS); i += 64u) lutl[g][i] = lut[(size_t)row * (1u << BITS) + i];
    __syncthreads();
    if (!ok) return;
    for (u32 w = l; w < G.wpr; w += 64u) {
        u32 chunk, t, tpw;
        if (w < 32u * G.nfull) {
            chunk = w / 32u;
            t = w % 32u;
            tpw = 32u;
        } else {
            chunk = G.nfull;
            t = w - 32u * G.nfull;
            tpw = G.eff;
        }
        u32 P[BITS];
#pragma unroll
        for (int p = 0; p < BITS; p++) P[p] = qw[((size_t)p * N + row) * G.wpr + w];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            u32 o[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                u32 ce = 0, co = 0;
#pragma unroll
                for (int p = 0; p < BITS; p++) {
                    ce = (ce << 1) | ((P[p] >> (31 - (8 * c + 2 * k))) & 1u);
                    co = (co << 1) | ((P[p] >> (30 - (8 * c + 2 * k))) & 1u);
                }
                o[k] = (u32)lutl[g][ce] | ((u32)lutl[g][co] << 16);
            }
            uint16_t *dst = W + (size_t)row * K + 1024u * chunk + 8u * tpw * (u32)c + 8u * t;
            *reinterpret_cast<uint4 *>(dst) = make_uint4(o[0], o[1], o[2], o[3]);
        }
    }
}

// ----------------------------------------------------------------------------------------------
// Launchers
// ----------------------------------------------------------------------------------------------
struct QuadCfg {
    u32 T, RS, RT;
    size_t smem;
};

bool pick_quad_cfg(u32 N, u32 K, QuadCfg &c) {
    if (K % 128u) return false;
    const u32 Q = K / 128u;
    if (Q > 512u) return false;
    // block size: multiple of 64 in [192, 512] wasting the fewest lanes; ties -> 256, then smaller
    u32 bestT = 0;
    double bestw = 2.0;
    const u32 cand[] = {256, 192, 320, 384, 448, 512};
    for (u32 T : cand) {
        if (T < Q) continue;
        double w = double(T - (T / Q) * Q) / double(T);
        if (w < bestw - 1e-9) {
            bestw = w;
            bestT = T;
        }
    }
    if (!bestT) return false;
    c.T = bestT;
    c.RS = bestT / Q;
    // rows in flight per lane: keep >= ~3 blocks per CU where the matrix allows it
    u32 rt = 4;
    const int env = gq_env_int("GQ_AP_RT", 0);
    if (env == 1 || env == 2 || env == 4)
        rt = (u32)env;
    else {
        while (rt > 1 && (N + c.RS * rt - 1) / (c.RS * rt) < 768u) rt >>= 1;
    }
    c.RT = rt;
    const u32 nchunks = K / 1024u + ((K % 1024u) ? 1u : 0u);
    c.smem = (size_t)K * 2u + (size_t)c.RS * rt * nchunks * 32u * 2u;
    if (c.smem < (size_t)K * 2u + 64u) c.smem = (size_t)K * 2u + 64u;
    return c.smem <= 160u * 1024u;
}

template <int BITS, int RT, bool NT>
int launch_quad_inst(const ApArgs &a, const QuadCfg &c, u32 M, hipStream_t s) {
    static size_t attr_set = 0;
    auto kern = ap_gemv_quad_kernel<BITS, RT, NT>;
    if (c.smem > 48u * 1024u && c.smem > attr_set) {
        GQ_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)c.smem));
        attr_set = c.smem;
    }
    const u32 RB = c.RS * c.RT;
    dim3 grid((a.N + RB - 1) / RB, M), block(c.T);
    hipLaunchKernelGGL(kern, grid, block, c.smem, s, a);
    GQ_HIP_CHECK(hipGetLastError());
    return GQ_OK;
}

template <int BITS>
int launch_quad(const ApArgs &a, const QuadCfg &c, u32 M, hipStream_t s) {
    const bool nt = gq_env_int("GQ_AP_NT", 1) != 0;
    switch (c.RT) {
        case 1: return nt ? launch_quad_inst<BITS, 1, true>(a, c, M, s) : launch_quad_inst<BITS, 1, false>(a, c, M, s);
        case 2: return nt ? launch_quad_inst<BITS, 2, true>(a, c, M, s) : launch_quad_inst<BITS, 2, false>(a, c, M, s);
        default: return nt ? launch_quad_inst<BITS, 4, true>(a, c, M, s) : launch_quad_inst<BITS, 4, false>(a, c, M, s);
    }
}

template <int BITS>
int launch_generic(const ApArgs &a, u32 M, hipStream_t s) {
    const int ksplit = (M == 1 && a.K > 4096 && BITS >= 7) ? 1 : 0;  // anyprec.cu:611
    dim3 grid((a.N + 7) / 8, M), block(256);
    hipLaunchKernelGGL(ap_gemv_generic_kernel<BITS>, grid, block, 0, s, a, ksplit);
    GQ_HIP_CHECK(hipGetLastError());
    return GQ_OK;
}

int ap_gemv_dispatch(ApArgs a, u32 M, int bits, hipStream_t s) {
    if (bits < 2 || bits > 8) return gq_fail(GQ_EINVAL, "Bitwidth must be between 2 and 8.");
    if (M < 1 || M > 8) return gq_fail(GQ_EINVAL, "batch size M must be between 1 and 8 (anyprec.cu:602).");
    if (a.K == 0 || a.K % 32u) return gq_fail(GQ_EINVAL, "input_feat (K) must be a positive multiple of 32.");
    if (a.N == 0) return gq_fail(GQ_EINVAL, "output_feat (N) must be positive.");
    if (!a.x || !a.out || !a.qw || !a.lut) return gq_fail(GQ_EINVAL, "null pointer argument.");
    const bool force_generic = gq_env_int("GQ_AP_FORCE_GENERIC", 0) != 0;
    QuadCfg c;
    if (!force_generic && bits <= 4 && pick_quad_cfg(a.N, a.K, c) && (((uintptr_t)a.qw | (uintptr_t)a.x) & 15u) == 0 &&
        ((uintptr_t)a.lut & (bits == 2 ? 7u : 15u)) == 0) {
        if (a.epilogue & GQ_EPI_SILU_MUL) return gq_fail(GQ_ENOTSUP, "SILU_MUL epilogue is served by gq_anyprec_gemv_fused only.");
        a.RS = c.RS;
        switch (bits) {
            case 2: return launch_quad<2>(a, c, M, s);
            case 3: return launch_quad<3>(a, c, M, s);
            default: return launch_quad<4>(a, c, M, s);
        }
    }
    if (a.normw || (a.epilogue & GQ_EPI_SILU_MUL))
        return gq_fail(GQ_ENOTSUP, "fused prologue/epilogue needs bits in 2..4, K % 128 == 0 and 16-byte aligned buffers.");
    if ((uintptr_t)a.x & 15u) return gq_fail(GQ_EINVAL, "input must be 16-byte aligned.");
    switch (bits) {
        case 2: return launch_generic<2>(a, M, s);
        case 3: return launch_generic<3>(a, M, s);
        case 4: return launch_generic<4>(a, M, s);
        case 5: return launch_generic<5>(a, M, s);
        case 6: return launch_generic<6>(a, M, s);
        case 7: return launch_generic<7>(a, M, s);
        default: return launch_generic<8>(a, M, s);
    }
}

}  // namespace

extern "C" int gq_anyprec_gemv(const void *x, void *out, const uint32_t *qweight, const void *lut, uint32_t M, uint32_t N,
                               uint32_t K, int bits, int dtype, void *stream) {
    if (dtype != GQ_DTYPE_F16) return gq_fail(GQ_ENOTSUP, "only fp16 is implemented (as in the reference, gemv.cu:46-49).");
    ApArgs a{};
    a.qw = qweight;
    a.lut = (const uint16_t *)lut;
    a.x = (const uint16_t *)x;
    a.out = (uint16_t *)out;
    a.N = N;
    a.K = K;
    a.epilogue = GQ_EPI_NONE;
    return ap_gemv_dispatch(a, M, bits, (hipStream_t)stream);
}

extern "C" int gq_anyprec_gemv_fused(const void *x, void *out, const uint32_t *qweight, const void *lut, uint32_t N,
                                     uint32_t K, int bits, const void *norm_weight, float eps, const void *residual,
                                     uint32_t epilogue, void *stream) {
    ApArgs a{};
    a.qw = qweight;
    a.lut = (const uint16_t *)lut;
    a.x = (const uint16_t *)x;
    a.out = (uint16_t *)out;
    a.normw = (const uint16_t *)norm_weight;
    a.resid = (epilogue & GQ_EPI_RESIDUAL) ? (const uint16_t *)residual : nullptr;
    a.eps = eps;
    a.N = N;
    a.K = K;
    a.epilogue = epilogue;
    if ((epilogue & GQ_EPI_RESIDUAL) && !residual) return gq_fail(GQ_EINVAL, "RESIDUAL epilogue needs a residual pointer.");
    if (epilogue & GQ_EPI_SILU_MUL) return gq_fail(GQ_ENOTSUP, "SILU_MUL epilogue not built yet.");
    return ap_gemv_dispatch(a, 1, bits, (hipStream_t)stream);
}

extern "C" int gq_anyprec_dequant(const uint32_t *qweight, const void *lut, void *W, uint32_t N, uint32_t K, int bits,
                                  void *stream) {
    if (bits < 2 || bits > 8) return gq_fail(GQ_EINVAL, "Bitwidth must be between 2 and 8.");
    if (K == 0 || K % 32u || N == 0) return gq_fail(GQ_EINVAL, "bad shape: need N > 0 and K a positive multiple of 32.");
    if (!qweight || !lut || !W) return gq_fail(GQ_EINVAL, "null pointer argument.");
    if ((uintptr_t)W & 15u) return gq_fail(GQ_EINVAL, "output must be 16-byte aligned.");
    dim3 grid((N + 3) / 4), block(256);
    hipStream_t s = (hipStream_t)stream;
    const uint16_t *l = (const uint16_t *)lut;
    uint16_t *w = (uint16_t *)W;
    switch (bits) {
        case 2: hipLaunchKernelGGL(ap_dequant_kernel<2>, grid, block, 0, s, qweight, l, w, N, K); break;
        case 3: hipLaunchKernelGGL(ap_dequant_kernel<3>, grid, block, 0, s, qweight, l, w, N, K); break;
        case 4: hipLaunchKernelGGL(ap_dequant_kernel<4>, grid, block, 0, s, qweight, l, w, N, K); break;
        case 5: hipLaunchKernelGGL(ap_dequant_kernel<5>, grid, block, 0, s, qweight, l, w, N, K); break;
        case 6: hipLaunchKernelGGL(ap_dequant_kernel<6>, grid, block, 0, s, qweight, l, w, N, K); break;
        case 7: hipLaunchKernelGGL(ap_dequant_kernel<7>, grid, block, 0, s, qweight, l, w, N, K); break;
        default: hipLaunchKernelGGL(ap_dequant_kernel<8>, grid, block, 0, s, qweight, l, w, N, K); break;
    }
    GQ_HIP_CHECK(hipGetLastError());
    return GQ_OK;
}
