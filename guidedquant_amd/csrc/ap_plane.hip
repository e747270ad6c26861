// ap_plane.hip -- "fast mode" Any-Precision GEMV for gfx950: binary bit-plane GEMVs on the matrix cores.
//
// See plane_core.h for the algorithm.  What runs where:
//   HBM    : the stored bit-planes, whole 128-byte lines (one row x one plane x one 1024-weight chunk), fetched by
//            direct-to-LDS loads (buffer_load_dwordx4 ... lds) that every wave issues for ALL of its steps (up to
//            its LDS ring depth) in its first instructions: the block's share of the matrix is in flight before
//            the activation prologue starts, and the matrix cores chase the stream.
//   VALU   : one v_and_b32 per 4 weights per plane-subset (bits -> bf8 {0, 2^e}); ANDs of planes for the subsets.
//   MFMA   : v_mfma_scale_f32_16x16x128_f8f6f4 (A = bf8 bit patterns, scale cancels 2^e; B = 4 bf8 pieces of x in
//            columns 0..3, per-32-element block scale), fp32 accumulate: 8 MFMAs per (chunk, plane-subset).
//   LDS    : per-wave rings of A tiles (plane_core.h atile_unit swizzle), the B image (activation pieces, 4*K bytes)
//            built once per block, partial sums of K-split items.
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
    u32 S;       // LDS ring slots (steps) per wave
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


typedef _Float16 h2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ h2v u2h2(u32 u) { return __builtin_bit_cast(h2v, u); }
__device__ __forceinline__ u32 h22u(h2v h) { return __builtin_bit_cast(u32, h); }

// --- hand-scheduled vector memory.  The compiler's waitcnt pass treats an LDS-DMA load as "may alias every later
// LDS access" and drains vmcnt before the first ds_read / barrier, which would serialise the prologue behind the
// whole plane stream.  So the plane loads and the activation loads that must overtake them are issued from inline
// asm (invisible to that pass) and waited for with explicit s_waitcnt vmcnt(n) (vector memory returns in order).
__device__ __forceinline__ void dma16(u32x4 rsrc, u32 lds_base, u32 voff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen nt lds" ::"s"(lds_base), "v"(voff), "s"(rsrc)
                 : "memory");  // m0 is not otherwise used in this kernel (no compiler-generated LDS-DMA / movrel)
}
__device__ __forceinline__ u32 bload32(u32x4 rsrc, u32 voff) {
    u32 r;
    asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(r) : "v"(voff), "s"(rsrc) : "memory");
    return r;
}
template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// wait until at most n * LPS vector-memory instructions of this wave are outstanding (n uniform, 0..4)
template <int LPS>
__device__ __forceinline__ void wait_vm_steps(u32 n) {
    switch (n) {
        case 0: wait_vm<0>(); break;
        case 1: wait_vm<LPS>(); break;
        case 2: wait_vm<2 * LPS>(); break;
        case 3: wait_vm<3 * LPS>(); break;
        default: wait_vm<4 * LPS>(); break;
    }
}
// data dependence of asm-loaded registers on the preceding s_waitcnt (volatile asm statements keep their order)
__device__ __forceinline__ void tie4(u32 (&r)[4]) { asm volatile("" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3])); }

__device__ __forceinline__ u32x4 make_rsrc(const void *p, u32 bytes) {
    const uint64_t a = (uint64_t)(uintptr_t)p;
    return (u32x4){(u32)a, (u32)(a >> 32) & 0xFFFFu, bytes, 0x00020000u};  // raw buffer: stride 0, num_records = bytes
}

template <int BITS, int PRO>
__global__ void __launch_bounds__(BITS == 2 ? 1024 : 512) ap_plane_kernel(PlaneArgs a) {
    constexpr int NP = 1 << BITS, NP1 = NP - 1;
    constexpr u32 T = BITS == 2 ? 1024u : 512u, W = T / 64u;
    constexpr u32 NI = 2048u / T;   // prologue passes: K <= 16384 -> at most 2048 (chunk, virtual lane, weight pair) items
    constexpr u32 LPS = 2u * BITS;  // direct-to-LDS loads per step
    constexpr u32 SLOT = 2048u * BITS;
    constexpr u32 OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    Geom G;
    G.init(a.K);
    const u32 tid = threadIdx.x;
    const u32 w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const u32 l = tid & 63u;
    const u32 CS = 1u << a.log2CS, cpi = a.cpi, S = a.S;
    const u32 nIt = a.RGB * CS;  // wave items of this block
    // LDS: [A rings: W * S slots][B image: nchunks * 4096][zero32][red: 64 floats][part: nIt * NP1 * 16 floats]
    unsigned char *ring = smem + (size_t)w * S * SLOT;
    unsigned char *bimg = smem + (size_t)W * S * SLOT;
    unsigned char *zero32 = bimg + G.nchunks * 4096u;
    float *red = reinterpret_cast<float *>(zero32 + 64);
    float *part = red + 64;
    const u32 rg0 = blockIdx.x * a.RGB;
    const u32 m = blockIdx.y;
    const u32 r = l & 15u, kb = l >> 4;
    auto stamp = [&](int i) {
        if (a.dbg && blockIdx.x == gridDim.x / 2 && l == 0) a.dbg[w * 8u + (u32)i] = __builtin_readcyclecounter();
    };
    stamp(0);

    // ---------------------------------------------------------------- 1. activation loads (must land first)
    // item = (chunk, virtual lane t, weight pair jp): the two adjacent weights j = 2jp, 2jp+1 of the 4 bytes c of t
    const u32x4 rsx = make_rsrc(a.x + (size_t)m * (PRO == PRO_SILUMUL ? 2u * G.K : G.K), (PRO == PRO_SILUMUL ? 4u : 2u) * G.K);
    const u32x4 rsa = make_rsrc(PRO == PRO_RMSNORM ? a.normw : a.x, 2u * G.K);
    u32 xr[NI][4], ar[NI][4];
#pragma unroll
    for (u32 n = 0; n < NI; n++) {
        const u32 it = tid + n * T;
        const u32 t = it & 31u, jp = (it >> 5) & 3u, chunk = it >> 7;
        const u32 tp = G.tpw(chunk);
        const bool ok = chunk < G.nchunks && t < tp;
#pragma unroll
        for (u32 c = 0; c < 4; c++) {
            const u32 e = 1024u * chunk + 8u * tp * c + 8u * t + 2u * jp;
            xr[n][c] = bload32(rsx, ok ? 2u * e : OOB);
            if constexpr (PRO == PRO_RMSNORM) ar[n][c] = bload32(rsa, ok ? 2u * e : OOB);
            if constexpr (PRO == PRO_SILUMUL) ar[n][c] = bload32(rsx, ok ? 2u * (G.K + e) : OOB);
        }
    }

    // ---------------------------------------------------------------- 2. the whole plane stream of this wave
    // steps of wave w: item i = w + n*W (n = 0, 1, ..), chunks c = 0..cpi-1 of the item; step q uses ring slot q % S
    const u32 items_w = nIt > w ? (nIt - w + W - 1u) / W : 0u;
    const u32 my_steps = items_w * cpi;
    const u32 plane_bytes = a.N * G.wpr * 4u;
    const u32x4 rq = make_rsrc(a.qw, plane_bytes * (u32)BITS);
    const u32 ring_lds = (u32)(uintptr_t)ring;
    u32 iq_item = w, iq_c = 0, iq_slot = 0;
    auto issue = [&]() {
        const u32 chunk = (iq_item & (CS - 1u)) * cpi + iq_c;
        const u32 rgi = rg0 + (iq_item >> a.log2CS);
#pragma unroll
        for (u32 h = 0; h < 2; h++) {
            u32 rr, seg;
            atile_src(h, l, rr, seg);
            const u32 row = rgi * 16u + rr;
            const bool ok = chunk < G.nchunks && row < a.N && 4u * seg < G.tpw(chunk);
            const u32 off = (row * G.wpr + 32u * chunk + 4u * seg) * 4u;
#pragma unroll
            for (u32 p = 0; p < (u32)BITS; p++)
                dma16(rq, ring_lds + iq_slot * SLOT + (p * 2u + h) * 1024u, ok ? off + p * plane_bytes : OOB);
        }
        if (++iq_c == cpi) {
            iq_c = 0;
            iq_item += W;
        }
        if (++iq_slot == S) iq_slot = 0;
    };
    // The activation loads must be back before the plane stream starts: once every CU has its share of the matrix
    // in flight, the L2 / fabric queues are thousands of lines deep and a late L2 hit waits behind them (measured:
    // activations land after 4000-9000 cycles when the plane loads are issued first or right behind them, ~900 alone).
    wait_vm<0>();
#pragma unroll
    for (u32 n = 0; n < NI; n++) {
        tie4(xr[n]);
        if constexpr (PRO != PRO_NONE) tie4(ar[n]);
    }
    const u32 init_steps = my_steps < S ? my_steps : S;
    for (u32 q = 0; q < init_steps; q++) issue();
    if (tid < 8) reinterpret_cast<u32 *>(zero32)[tid] = 0u;
    stamp(6);

    // ---------------------------------------------------------------- 3. statistics -> one barrier
    // RMSNorm: sum x^2 and max |x * w| (bounds the normalised maximum); otherwise max |x'| of the transformed vector
    float nscale = 0.f, xmax = 0.f;
    {
        float ss = 0.f, mx = 0.f;
        u32 mxi = 0;
#pragma unroll
        for (u32 n = 0; n < NI; n++)
#pragma unroll
            for (u32 c = 0; c < 4; c++) {
                if constexpr (PRO == PRO_RMSNORM) {
                    const float p = h2f(xr[n][c] & 0xFFFF), q = h2f(xr[n][c] >> 16);
                    ss += p * p;
                    ss += q * q;
                    mx = fmaxf(mx, fmaxf(fabsf(p * h2f(ar[n][c] & 0xFFFF)), fabsf(q * h2f(ar[n][c] >> 16))));
                } else {
                    if constexpr (PRO == PRO_SILUMUL) {
                        // F.silu(gate) * up on fp16 tensors -- inference/model.py:266
                        _Float16 hh[2];
#pragma unroll
                        for (int k = 0; k < 2; k++) {
                            const float gv = h2f((uint16_t)(xr[n][c] >> (16 * k)));
                            hh[k] = (_Float16)(gv / (1.0f + __expf(-gv))) * __builtin_bit_cast(_Float16, (uint16_t)(ar[n][c] >> (16 * k)));
                        }
                        xr[n][c] = (u32)__builtin_bit_cast(uint16_t, hh[0]) | ((u32)__builtin_bit_cast(uint16_t, hh[1]) << 16);
                    }
                    // |fp16| bit patterns order like unsigned integers
                    const u32 ab = xr[n][c] & 0x7FFF7FFFu;
                    mxi = max(mxi, max(ab & 0xFFFFu, ab >> 16));
                }
            }
        if constexpr (PRO != PRO_RMSNORM) mx = h2f((uint16_t)mxi);
        mx = wave_reduce<true>(mx);
        if constexpr (PRO == PRO_RMSNORM) ss = wave_reduce<false>(ss);
        if (l == 63) {
            red[w] = mx;
            if constexpr (PRO == PRO_RMSNORM) red[16 + w] = ss;
        }
        __syncthreads();
#pragma unroll
        for (u32 i = 0; i < W; i++) xmax = fmaxf(xmax, red[i]);
        if constexpr (PRO == PRO_RMSNORM) {
            float tot = 0.f;
#pragma unroll
            for (u32 i = 0; i < W; i++) tot += red[16 + i];
            nscale = 1.0f / sqrtf(tot / (float)G.K + a.eps);
            xmax = xmax * nscale * 1.002f;  // covers the two fp16 roundings of the transform
        }
    }
    stamp(7);

    // ---------------------------------------------------------------- 4. transform, scale, split, scatter
    const int ksh = piece_shift(xmax);
    const int sb = 127 - ksh;  // E8M0 scale of every B block
    {
        const uint16_t k16 = pow2_f16(ksh);
        const h2v kk = u2h2((u32)k16 * 0x10001u), one2 = u2h2(0x3C003C00u);
        float xsum = 0.f;
#pragma unroll
        for (u32 n = 0; n < NI; n++) {
            const u32 it = tid + n * T;
            const u32 t = it & 31u, jp = (it >> 5) & 3u, chunk = it >> 7;
            if (chunk >= G.nchunks) continue;
            const u32 kbi = t >> 3, v = t & 7u;
            u32 P[4][4];  // [piece][c]
#pragma unroll
            for (u32 c = 0; c < 4; c++) {
                u32 xw = xr[n][c];
                if constexpr (PRO == PRO_RMSNORM) {
                    // (x.float() * rsqrt(mean(x^2)+eps)).half() * w  -- inference/model.py:281-292, both fp16 roundings kept
                    const _Float16 h0 = (_Float16)(h2f(xw & 0xFFFF) * nscale), h1 = (_Float16)(h2f(xw >> 16) * nscale);
                    xw = h22u((h2v){h0, h1} * u2h2(ar[n][c]));
                }
                xsum = __builtin_amdgcn_fdot2(u2h2(xw), one2, xsum, false);
                h2v rem = u2h2(xw) * kk;
#pragma unroll
                for (u32 p = 0; p < 4; p++) {
                    P[p][c] = h22u(rem) & 0xFF00FF00u;
                    if (p < 3) rem = rem - u2h2(P[p][c]);
                }
            }
#pragma unroll
            for (u32 p = 0; p < 4; p++) {
                // image bytes B = 0..3 hold c = 3..0; the bf8 of weight 2jp is byte 1 of P, of weight 2jp+1 byte 3
                const u32 hi = __builtin_amdgcn_perm(P[p][3], P[p][2], 0x03070105u);
                const u32 lo = __builtin_amdgcn_perm(P[p][1], P[p][0], 0x03070105u);
                const u32 w0 = __builtin_amdgcn_perm(lo, hi, 0x05040100u), w1 = __builtin_amdgcn_perm(lo, hi, 0x07060302u);
                *reinterpret_cast<u32 *>(bimg + bimg_off(chunk, 7u - 2u * jp, kbi, p) + 4u * v) = w0;
                *reinterpret_cast<u32 *>(bimg + bimg_off(chunk, 6u - 2u * jp, kbi, p) + 4u * v) = w1;
            }
        }
        xsum = wave_reduce<false>(xsum);
        if (l == 63) red[32 + w] = xsum;
    }
    stamp(1);
    __syncthreads();
    stamp(2);
    float X = 0.f;
#pragma unroll
    for (u32 i = 0; i < W; i++) X += red[32 + i];

    // ---------------------------------------------------------------- 5. main loop: the steps of this wave
    const u32 col = l & 15u;
    const bool bcol = col < 4u;
    const u32 offA0 = atile_unit(r, 2u * kb) * 16u, offA1 = atile_unit(r, 2u * kb + 1u) * 16u;
    v4f acc[NP1];
#pragma unroll
    for (int i = 0; i < NP1; i++) acc[i] = (v4f){0.f, 0.f, 0.f, 0.f};
    u32 cq_item = w, cq_c = 0, cq_slot = 0;
    for (u32 q = 0; q < my_steps; q++) {
        const u32 chunk = (cq_item & (CS - 1u)) * cpi + cq_c;
        const u32 after = my_steps - 1u - q;
        wait_vm_steps<LPS>(after < S - 1u ? after : S - 1u);
        u32 Wd[BITS][8];
        {
            const unsigned char *slot = ring + cq_slot * SLOT;
#pragma unroll
            for (int p = 0; p < BITS; p++) {
                const uint4 a0 = *reinterpret_cast<const uint4 *>(slot + p * 2048 + offA0);
                const uint4 a1 = *reinterpret_cast<const uint4 *>(slot + p * 2048 + offA1);
                Wd[p][0] = a0.x, Wd[p][1] = a0.y, Wd[p][2] = a0.z, Wd[p][3] = a0.w;
                Wd[p][4] = a1.x, Wd[p][5] = a1.y, Wd[p][6] = a1.z, Wd[p][7] = a1.w;
            }
        }
        if (q + S < my_steps) {
            // refill this slot with the step S ahead: the ds_reads above must have returned first
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            issue();
        }
        if (++cq_slot == S) cq_slot = 0;
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
        if (++cq_c == cpi) {
            // item done: add the 4 piece columns (lanes col = 0..3 of each 16-lane group), park 16 x NP1 sums in LDS
#pragma unroll
            for (int cm = 0; cm < NP1; cm++) {
                v4f v = acc[cm];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    float f = v[k];
                    f += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, f), 0xB1, 0xF, 0xF, false));
                    f += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, f), 0x4E, 0xF, 0xF, false));
                    v[k] = f;
                }
                // part[item][subset][row]: lane (col 0, kb) owns rows 4kb..4kb+3
                if (col == 0u) *reinterpret_cast<v4f *>(part + ((size_t)cq_item * NP1 + cm) * 16u + 4u * kb) = v;
                acc[cm] = (v4f){0.f, 0.f, 0.f, 0.f};
            }
            cq_c = 0;
            cq_item += W;
        }
    }
    stamp(3);
    __syncthreads();
    stamp(4);

    // ---------------------------------------------------------------- 6. epilogue: coefficients x plane sums
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
            for (u32 cs = 0; cs < CS; cs++) tsum += part[(((size_t)((rgl << a.log2CS) + cs)) * NP1 + (cm - 1)) * 16u + rr];
            y += f[cm] * tsum;
        }
        _Float16 yh = (_Float16)y;
        if (a.resid) yh = __builtin_bit_cast(_Float16, a.resid[(size_t)m * a.N + row]) + yh;
        a.out[(size_t)m * a.N + row] = __builtin_bit_cast(uint16_t, yh);
    }
    stamp(5);
}

struct PlaneCfg {
    u32 grid, T, RGB, log2CS, cpi, S;
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
    const u32 W = bits == 2 ? 16u : 8u;
    c.T = 64u * W;
    // one block per CU when the matrix is big enough
    u32 rgb = (RGt + ncu - 1u) / ncu;
    if (rgb < 1) rgb = 1;
    c.RGB = rgb;
    c.grid = (RGt + rgb - 1u) / rgb;
    // split K of a row group over 2^log2CS wave items until the block has at least one item per wave
    u32 lcs = 0;
    while ((rgb << lcs) < W && (2u << lcs) <= nchunks) lcs++;
    const int envcs = gq_env_int("GQ_PL_LOG2CS", -1);
    if (envcs >= 0 && (1u << envcs) <= nchunks) lcs = (u32)envcs;
    c.log2CS = lcs;
    c.cpi = (nchunks + (1u << lcs) - 1u) >> lcs;
    // ring depth: as many steps per wave as the wave has / as fit next to the B image (<= 4: vmcnt immediates)
    const u32 nIt = rgb << lcs;
    const u32 steps_w = ((nIt + W - 1u) / W) * c.cpi;
    const u32 np1 = (1u << bits) - 1u;
    const size_t fixed = (size_t)nchunks * 4096u + 64u + 64u * 4u + (size_t)nIt * np1 * 16u * 4u;
    const size_t slot = 2048u * (size_t)bits, lds = 160u * 1024u;
    if (fixed + W * slot > lds) return false;
    u32 S = (u32)((lds - fixed) / (W * slot));
    if (S > 4u) S = 4u;
    if (S > steps_w) S = steps_w;
    const int envs = gq_env_int("GQ_PL_S", 0);
    if (envs >= 1 && (u32)envs < S) S = (u32)envs;
    c.S = S;
    c.smem = fixed + (size_t)W * S * slot;
    return true;
}

template <int BITS, int PRO>
int launch_plane_inst(const PlaneArgs &a, const PlaneCfg &c, u32 M, hipStream_t s) {
    static bool attr_set = false;
    auto kern = ap_plane_kernel<BITS, PRO>;
    if (!attr_set) {
        GQ_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)(160u * 1024u)));
        attr_set = true;
    }
    dim3 grid(c.grid, M), block(c.T);
    hipLaunchKernelGGL(kern, grid, block, c.smem, s, a);
    GQ_HIP_CHECK(hipGetLastError());
    return GQ_OK;
}

template <int BITS>
int launch_plane(const PlaneArgs &a, const PlaneCfg &c, u32 M, int pro, hipStream_t s) {
    switch (pro) {
        case PRO_RMSNORM: return launch_plane_inst<BITS, PRO_RMSNORM>(a, c, M, s);
        case PRO_SILUMUL: return launch_plane_inst<BITS, PRO_SILUMUL>(a, c, M, s);
        default: return launch_plane_inst<BITS, PRO_NONE>(a, c, M, s);
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
    a.S = c.S;
    a.eps = eps;
    a.dbg = g_dbg;
    switch (bits) {
        case 2: return launch_plane<2>(a, c, M, pro, stream);
        case 3: return launch_plane<3>(a, c, M, pro, stream);
        default: return launch_plane<4>(a, c, M, pro, stream);
    }
}
