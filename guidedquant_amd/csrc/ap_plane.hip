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
    unsigned long long *dbg;  // optional: per-wave phase timestamps of block 0 (tools/phase_timing.py)
};

enum { PRO_NONE = 0, PRO_RMSNORM = 1, PRO_SILUMUL = 2 };

__device__ __forceinline__ float h2f(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
__device__ __forceinline__ uint16_t f2h(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }

// wave64 reductions on the DPP path (no LDS round trips): result valid in lane 63
template <bool MAX>
__device__ __forceinline__ float wave_reduce(float v) {
    auto step = [&](auto dpp) {
        float o = __builtin_bit_cast(float, dpp(__builtin_bit_cast(int, v)));
        v = MAX ? fmaxf(v, o) : v + o;
    };
    step([](int x) { return __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xF, 0xF, false); });   // quad_perm [1,0,3,2]
    step([](int x) { return __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xF, 0xF, false); });   // quad_perm [2,3,0,1]
    step([](int x) { return __builtin_amdgcn_update_dpp(x, x, 0x141, 0xF, 0xF, false); });  // row_half_mirror
    step([](int x) { return __builtin_amdgcn_update_dpp(x, x, 0x140, 0xF, 0xF, false); });  // row_mirror
    {   // row_bcast15 into rows 1 and 3, row_bcast31 into rows 2 and 3 (identity elsewhere)
        const int ident = MAX ? 0 : 0;  // |x| >= 0 and the sum identity are both 0.0f
        float o = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(ident, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, false));
        v = MAX ? fmaxf(v, o) : v + o;
        o = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(ident, __builtin_bit_cast(int, v), 0x143, 0xC, 0xF, false));
        v = MAX ? fmaxf(v, o) : v + o;
    }
    return v;
}

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
__global__ void __launch_bounds__(BITS == 2 ? 1024 : 512) ap_plane_kernel(PlaneArgs a) {
    constexpr int NP = 1 << BITS, NP1 = NP - 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    Geom G;
    G.init(a.K);
    const u32 T = blockDim.x, tid = threadIdx.x;
    const u32 W = T >> 6;
    const u32 w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const u32 l = tid & 63u;
    unsigned char *bimg = smem;                            // [chunk][s][kb][piece][32]
    unsigned char *bscale = bimg + G.nchunks * 4096u;      // u32 [chunk][blk]: E8M0 scale of the hardware scale block (64 B per chunk reserved)
    unsigned char *zero32 = bscale + G.nchunks * 64u;      // 32 zero bytes
    uint16_t *xs = reinterpret_cast<uint16_t *>(zero32 + 32 + 64 * 4);  // fp16 copy of the (transformed) activations
    uint16_t *aux = xs + G.K;  // RMSNorm weight / up vector (prologue modes only)
    float *red = reinterpret_cast<float *>(zero32 + 32);   // 64 floats
    float *part = reinterpret_cast<float *>(xs + (PRO == PRO_NONE ? G.K : 2u * G.K));  // [RGB << log2CS][NP1][4 piece columns][16 rows]

    const u32 CS = 1u << a.log2CS, cpi = a.cpi;
    const u32 nIt = a.RGB * CS;  // wave items of this block
    const u32 rg0 = blockIdx.x * a.RGB;
    const u32 m = blockIdx.y;
    const u32 r = l & 15u, kb = l >> 4;
    auto stamp = [&](int i) {
        if (a.dbg && blockIdx.x == gridDim.x / 2 && l == 0) a.dbg[w * 8u + (u32)i] = __builtin_readcyclecounter();
    };
    stamp(0);

    // ---------------------------------------------------------------- plane ring (register ring of D chunk steps)
    constexpr u32 OOB = 0x80000000u;
    const u32 plane_bytes = a.N * G.wpr * 4u;
    __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void *)a.qw, 0, (int)(plane_bytes * (u32)BITS), 0x00020000);
    u32x4 P[D][BITS][2];
    const u32 wi = (w + W - (W > 1 ? W / 2u : 0u)) % W;  // item slot of this wave: upper-half (loader-first) waves come first
    u32 iq_item = wi, iq_c = 0;
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

    // Vector-memory results return in order, so a wave that issues its plane loads first cannot see its
    // activation loads until all of those landed.  Split roles: the upper half of the waves starts streaming
    // planes at t = 0 and goes straight to the barrier; the lower half builds the activation image, then streams.
    const u32 WP = W > 1 ? W / 2u : 1u;  // prologue waves
    const u32 TP = WP * 64u;
    const bool pro_wave = w < WP;
    if (!pro_wave) {
#pragma unroll
        for (int d = 0; d < D; d++) issue(d);
    }

    // ---------------------------------------------------------------- prologue: activation pieces -> LDS image
    // Phase A (prologue waves): x -> (optional RMSNorm / SiLU*up, reference rounding points) -> fp16 copy in LDS,
    //          plus max|x| and sum(x).   Phase B (all waves): split into 4 exact bf8 pieces of x * 2^(15-e_max)
    //          and scatter them into the MFMA B image.
    const uint16_t *x = a.x + (size_t)m * (PRO == PRO_SILUMUL ? 2u * G.K : G.K);
    // step 0 (prologue waves): raw copy global -> LDS (the only vector-memory traffic of the prologue)
    if (pro_wave) {
        for (u32 g = tid; g < G.K / 8u; g += TP) {
            *reinterpret_cast<uint4 *>(xs + 8u * g) = *reinterpret_cast<const uint4 *>(x + 8u * g);
            if constexpr (PRO == PRO_RMSNORM) *reinterpret_cast<uint4 *>(aux + 8u * g) = *reinterpret_cast<const uint4 *>(a.normw + 8u * g);
            if constexpr (PRO == PRO_SILUMUL) *reinterpret_cast<uint4 *>(aux + 8u * g) = *reinterpret_cast<const uint4 *>(x + G.K + 8u * g);
        }
        if (tid < 8) reinterpret_cast<u32 *>(zero32)[tid] = 0u;
    }
    stamp(6);
    __syncthreads();
    // step 1 (all waves): optional RMSNorm / SiLU*up with the reference's fp16 rounding points, max|x|, sum(x)
    float nscale = 0.f;
    if constexpr (PRO == PRO_RMSNORM) {
        float ss = 0.f;
        for (u32 g = tid; g < G.K / 8u; g += T) {
            const uint4 v4 = *reinterpret_cast<const uint4 *>(xs + 8u * g);
            const u32 ww[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                float p = h2f(ww[k] & 0xFFFF), q = h2f(ww[k] >> 16);
                ss += p * p;
                ss += q * q;
            }
        }
        ss = wave_reduce<false>(ss);
        if (l == 63) red[32 + w] = ss;
        __syncthreads();
        float tot = 0.f;
        for (u32 i = 0; i < W; i++) tot += red[32 + i];
        nscale = 1.0f / sqrtf(tot / (float)G.K + a.eps);
    }
    {
        float xsum = 0.f, mx = 0.f;
        for (u32 g = tid; g < G.K / 8u; g += T) {
            const uint4 v4 = *reinterpret_cast<const uint4 *>(xs + 8u * g);
            u32 ww[4] = {v4.x, v4.y, v4.z, v4.w};
            if constexpr (PRO != PRO_NONE) {
                const uint4 a4 = *reinterpret_cast<const uint4 *>(aux + 8u * g);
                const u32 aw[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    _Float16 h[2];
#pragma unroll
                    for (int hh = 0; hh < 2; hh++) {
                        const uint16_t hx = (uint16_t)(ww[k] >> (16 * hh)), ha = (uint16_t)(aw[k] >> (16 * hh));
                        if constexpr (PRO == PRO_RMSNORM) {
                            // (x.float() * rsqrt(mean(x^2)+eps)).half() * w  -- inference/model.py:281-292
                            h[hh] = (_Float16)(h2f(hx) * nscale) * __builtin_bit_cast(_Float16, ha);
                        } else {
                            // F.silu(gate) * up on fp16 tensors -- inference/model.py:266
                            const float gv = h2f(hx);
                            h[hh] = (_Float16)(gv / (1.0f + __expf(-gv))) * __builtin_bit_cast(_Float16, ha);
                        }
                    }
                    ww[k] = (u32)__builtin_bit_cast(uint16_t, h[0]) | ((u32)__builtin_bit_cast(uint16_t, h[1]) << 16);
                }
                *reinterpret_cast<uint4 *>(xs + 8u * g) = make_uint4(ww[0], ww[1], ww[2], ww[3]);
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const float p = h2f(ww[k] & 0xFFFF), q = h2f(ww[k] >> 16);
                xsum += p;
                xsum += q;
                mx = fmaxf(mx, fmaxf(fabsf(p), fabsf(q)));
            }
        }
        xsum = wave_reduce<false>(xsum);
        mx = wave_reduce<true>(mx);
        if (l == 63) {
            red[w] = xsum;
            red[16 + w] = mx;
        }
    }
    stamp(7);
    __syncthreads();
    float X = 0.f, xmax = 0.f;
    for (u32 i = 0; i < W; i++) {
        X += red[i];
        xmax = fmaxf(xmax, red[16 + i]);
    }
    // one power-of-two scale for the whole vector: |x| * 2^(15-eb) < 2^15 (bf8 has 32 binades below that)
    const int eb = xmax > 0.f ? (int)((__builtin_bit_cast(u32, xmax) >> 23) & 0xFFu) - 126 : 15;
    const float sc = __builtin_bit_cast(float, (u32)(15 - eb + 127) << 23);
    const int sb = 127 + eb - 15;  // E8M0 scale of every B block
    // Phase B: item = (chunk, t, j pair): the 4 bytes c of two adjacent weights -> 2 x 4 image words
    for (u32 it = tid; it < G.nchunks * 128u; it += T) {
        const u32 t = it & 31u, jp = (it >> 5) & 3u, chunk = it >> 7;  // 32 consecutive lanes = the 32 virtual lanes
        const u32 tp = G.tpw(chunk), kbi = t >> 3, v = t & 7u;
        float xe[4][2];
#pragma unroll
        for (u32 c = 0; c < 4; c++) {
            const u32 e = 1024u * chunk + 8u * tp * c + 8u * t + 2u * jp;
            const u32 rw = t < tp ? *reinterpret_cast<const u32 *>(xs + e) : 0u;
            xe[c][0] = h2f((uint16_t)(rw & 0xFFFF)) * sc;
            xe[c][1] = h2f((uint16_t)(rw >> 16)) * sc;
        }
#pragma unroll
        for (u32 jj = 0; jj < 2; jj++) {
            const u32 s = 7u - (2u * jp + jj);
            float q0 = xe[0][jj], q1 = xe[1][jj], q2 = xe[2][jj], q3 = xe[3][jj];
#pragma unroll
            for (u32 p = 0; p < 4; p++) {
                // bytes B = 0..3 hold c = 3..0
                int wd = __builtin_amdgcn_cvt_pk_bf8_f32(q3, q2, 0, false);
                wd = __builtin_amdgcn_cvt_pk_bf8_f32(q1, q0, wd, true);
                *reinterpret_cast<int *>(bimg + bimg_off(chunk, s, kbi, p) + 4u * v) = wd;
                if (p < 3) {
                    q3 -= __builtin_amdgcn_cvt_f32_bf8(wd, 0);
                    q2 -= __builtin_amdgcn_cvt_f32_bf8(wd, 1);
                    q1 -= __builtin_amdgcn_cvt_f32_bf8(wd, 2);
                    q0 -= __builtin_amdgcn_cvt_f32_bf8(wd, 3);
                }
            }
        }
    }
    if (pro_wave) {
#pragma unroll
        for (int d = 0; d < D; d++) issue(d);
    }
    stamp(1);
    __syncthreads();
    stamp(2);

    // ---------------------------------------------------------------- main loop: (item, chunk) steps of this wave
    const u32 items_w = nIt > wi ? (nIt - wi + W - 1u) / W : 0u;
    const u32 my_steps = items_w * cpi;
    const u32 col = l & 15u;
    const bool bcol = col < 4u;
    v4f acc[NP1];
#pragma unroll
    for (int i = 0; i < NP1; i++) acc[i] = (v4f){0.f, 0.f, 0.f, 0.f};
    u32 cq_item = wi, cq_c = 0;

    for (u32 q = 0; q < my_steps; q += D) {
#pragma unroll
        for (int d = 0; d < D; d++) {
            if (q + (u32)d < my_steps) {
                const u32 chunk = (cq_item & (CS - 1u)) * cpi + cq_c;
                auto stamp2 = [&](int i) {
                    if (a.dbg && blockIdx.x == gridDim.x / 2 && l == 0 && q + (u32)d < 2u)
                        a.dbg[64u + w * 16u + (q + (u32)d) * 8u + (u32)i] = __builtin_readcyclecounter();
                };
                stamp2(0);
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
                if (a.dbg) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                stamp2(1);
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
                    stamp2(2);
                    // B operand (activation pieces) double-buffered over the bit position s
                    const unsigned char *bbase = bcol ? bimg + bimg_off(chunk, 0u, kb, col) : zero32;
                    const u32 bstep = bcol ? 512u : 0u;  // bimg_off(.., s+1, ..) - bimg_off(.., s, ..) = 4 pieces * 4 kb * 32 B
                    uint4 bn0 = *reinterpret_cast<const uint4 *>(bbase);
                    uint4 bn1 = *reinterpret_cast<const uint4 *>(bbase + 16);
#pragma unroll
                    for (int s = 0; s < 8; s++) {
                        const uint4 b0 = bn0, b1 = bn1;
                        if (s < 7) {
                            bn0 = *reinterpret_cast<const uint4 *>(bbase + (u32)(s + 1) * bstep);
                            bn1 = *reinterpret_cast<const uint4 *>(bbase + (u32)(s + 1) * bstep + 16);
                        }
                        v8i Bv = {(int)b0.x, (int)b0.y, (int)b0.z, (int)b0.w, (int)b1.x, (int)b1.y, (int)b1.z, (int)b1.w};
#pragma unroll
                        for (int cm = 1; cm < NP; cm++) {
                            v8i Av;
#pragma unroll
                            for (int v = 0; v < 8; v++) Av[v] = (int)extract(PW[cm - 1][v], s);
                            acc[cm - 1] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(Av, Bv, acc[cm - 1], 1, 1, 0, scale_byte(s), 0, sb);
                        }
                    }
                }
                stamp2(3);
                if (++cq_c == cpi) {
                    // item done: add the 4 piece columns, park the 16 x NP1 sums in LDS
#pragma unroll
                    for (int cm = 0; cm < NP1; cm++) {
                        // part[item][plane][piece column][row]: lane (col, kb) owns rows 4kb..4kb+3 of its column
                        if (bcol) *reinterpret_cast<v4f *>(part + (((size_t)cq_item * NP1 + cm) * 4u + col) * 16u + 4u * kb) = acc[cm];
                        acc[cm] = (v4f){0.f, 0.f, 0.f, 0.f};
                    }
                    cq_c = 0;
                    cq_item += W;
                    stamp2(4);
                }
            }
        }
    }
    stamp(3);
    __syncthreads();
    stamp(4);

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
            for (u32 cs = 0; cs < CS; cs++)
#pragma unroll
                for (u32 pc = 0; pc < 4; pc++)
                    tsum += part[((((size_t)((rgl << a.log2CS) + cs)) * NP1 + (cm - 1)) * 4u + pc) * 16u + rr];
            y += f[cm] * tsum;
        }
        _Float16 yh = (_Float16)y;
        if (a.resid) yh = __builtin_bit_cast(_Float16, a.resid[(size_t)m * a.N + row]) + yh;
        a.out[(size_t)m * a.N + row] = __builtin_bit_cast(uint16_t, yh);
    }
    stamp(5);
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
    if (K % 256u || K > 16384u) return false;
    const u32 nchunks = K / 1024u + ((K % 1024u) ? 1u : 0u);
    const u32 RGt = (N + 15u) / 16u;
    const u32 ncu = (u32)cus();
    // split K of a row group over 2^log2CS wave items until there are ~1.5 items per SIMD
    u32 lcs = 0;
    const u32 want = (u32)gq_env_int("GQ_PL_ITEMS", (int)(ncu * 24u));
    while ((RGt << lcs) < want && (2u << lcs) <= nchunks) lcs++;
    const int envcs = gq_env_int("GQ_PL_LOG2CS", -1);
    if (envcs >= 0 && (1u << envcs) <= nchunks) lcs = (u32)envcs;
    c.log2CS = lcs;
    c.cpi = (nchunks + (1u << lcs) - 1u) >> lcs;
    const u32 Wmax = bits == 2 ? 16u : 8u;
    u32 W = (u32)gq_env_int("GQ_PL_WAVES", (int)Wmax);
    if (W < 2 || W > Wmax) W = Wmax;
    c.T = 64u * W;
    // row groups per block: one block per CU when the matrix is big enough, never more items than ~2 per wave
    u32 rgb = (RGt + ncu - 1u) / ncu;
    const u32 envbpc = (u32)gq_env_int("GQ_PL_BPC", 1);
    if (envbpc > 1) rgb = (RGt + ncu * envbpc - 1u) / (ncu * envbpc);
    if (rgb < 1) rgb = 1;
    c.RGB = rgb;
    c.grid = (RGt + rgb - 1u) / rgb;
    int d = gq_env_int("GQ_PL_D", 0);
    if (d < 1 || d > 2) d = 1;
    c.D = d;
    const u32 np1 = (1u << bits) - 1u;
    c.smem = (size_t)nchunks * 4096u + nchunks * 64u + 32u + 64u * 4u + (size_t)K * 4u + (size_t)(rgb << lcs) * 16u * np1 * 4u * 4u;
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
        return launch_plane_inst<BITS, 1, PRO>(a, c, M, s);
    } else {
        switch (c.D) {
            case 1: return launch_plane_inst<BITS, 1, PRO>(a, c, M, s);
            default: return launch_plane_inst<BITS, 2, PRO>(a, c, M, s);
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

unsigned long long *g_dbg = nullptr;
}  // namespace

extern "C" void gq_debug_set_timing_buffer(void *p) { g_dbg = (unsigned long long *)p; }

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
    a.dbg = g_dbg;
    switch (bits) {
        case 2: return launch_plane<2>(a, c, M, pro, stream);
        case 3: return launch_plane<3>(a, c, M, pro, stream);
        default: return launch_plane<4>(a, c, M, pro, stream);
    }
}
