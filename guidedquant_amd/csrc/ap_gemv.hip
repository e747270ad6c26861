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
#include <type_traits>

#include "ap_core.h"
#include "gq_internal.h"

using namespace gq;

unsigned long long *gq_debug_timing_buffer();  // ap_plane.hip (gq_debug_set_timing_buffer)

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
    u32 SPB;               // row steps per block
    u32 epilogue;
    float eps;
    void *ws;              // optional caller workspace (gq_anyprec_gemv_fused_ws) and its size
    size_t ws_bytes;
    const float *ssq_in;   // statistics hand-over (gq_anyprec_gemv_fused_ho): partial sums of squares of x / of the outputs
    float *ssq_out;
    unsigned long long *dbg = nullptr;  // GQ_STAMPS builds: the debug buffer of gq_debug_set_timing_buffer (tools/r6/exact_stamps.py)
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
// Work item = (quad q, byte c): 4 virtual lanes x 8 activations = 64 contiguous bytes of x, regrouped with
// 16 v_perm_b32 into the 4 sixteen-byte slots (c, jj) of quad q.
// Optional fused prologues, with the same fp16 rounding points as the reference's separate kernels:
//   PRO_RMSNORM: y = (x.float() * rsqrt(mean(x^2) + eps)).half() * w        (inference/model.py:281-292)
//   PRO_SILUMUL: y = silu(g) * u with g = x[0:K], u = x[K:2K]                (inference/model.py:266)
// ----------------------------------------------------------------------------------------------
enum { PRO_NONE = 0, PRO_RMSNORM = 1, PRO_SILUMUL = 2 };

__device__ __forceinline__ u32 silu_mul_pk(u32 g, u32 u) {
    // F.silu on an fp16 tensor evaluates x / (1 + exp(-x)) in fp32 and rounds to fp16; then an fp16 multiply.
    float g0 = h2f(g & 0xFFFF), g1 = h2f(g >> 16);
    _Float16 s0 = (_Float16)(g0 / (1.0f + __expf(-g0)));
    _Float16 s1 = (_Float16)(g1 / (1.0f + __expf(-g1)));
    _Float16 r0 = s0 * __builtin_bit_cast(_Float16, (uint16_t)(u & 0xFFFF));
    _Float16 r1 = s1 * __builtin_bit_cast(_Float16, (uint16_t)(u >> 16));
    return (u32)__builtin_bit_cast(uint16_t, r0) | ((u32)__builtin_bit_cast(uint16_t, r1) << 16);
}

template <int PRO>
__device__ __forceinline__ void stage_x(const RowGeom &G, const uint16_t *x, const uint16_t *normw, float eps,
                                        uint16_t *xlds, float *red /* >= 16 floats of LDS */, bool stager, u32 T) {
    const u32 tid = threadIdx.x;
    float scale = 0.f;
    // RMSNorm (round 5): ONE memory round trip.  A stager thread requests the activations AND the norm weights of its first item (32
    // activations: 4 + 4 sixteen-byte loads) up front, takes the sum of squares from those registers, and normalises from them behind
    // the barrier.  The first version read x, reduced, and only then requested x again together with the norm weights -- a layer
    // weight that comes from HBM: two extra round trips, +2.1 us (3 bits) / +3.3 us (4 bits) on the 8B wqkv launch against the plain one
    // (bench.py roofline_by_shape, round 5).  Items beyond the first (K > 16 T) are re-read as before.
    u32 in0[4][4], nw0[4][4];
    const u32 idx0 = stager ? tid : 4u * G.Q;
    if constexpr (PRO == PRO_RMSNORM) {
        float ss = 0.f;
        for (u32 idx = idx0; idx < 4u * G.Q; idx += T) {
            const u32 e0 = G.xindex(idx >> 2, 0u, idx & 3u, 0u);
#pragma unroll
            for (int v = 0; v < 4; v++) {
                const uint4 t4 = ld16(x + e0 + 8 * v);
                const u32 w[4] = {t4.x, t4.y, t4.z, t4.w};
                if (idx == idx0) {
                    const uint4 n4 = ld16(normw + e0 + 8 * v);
                    in0[v][0] = t4.x, in0[v][1] = t4.y, in0[v][2] = t4.z, in0[v][3] = t4.w;
                    nw0[v][0] = n4.x, nw0[v][1] = n4.y, nw0[v][2] = n4.z, nw0[v][3] = n4.w;
                }
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    float a = h2f(w[i] & 0xFFFF), b = h2f(w[i] >> 16);
                    ss += a * a;
                    ss += b * b;
                }
            }
        }
        ss = gq_wave_allsum(ss);
        if ((tid & 63u) == 0 && stager) red[tid >> 6] = ss;
        __syncthreads();
        // (the stager waves' sums: one LDS round trip + the same register tree, instead of one dependent ds_read per wave)
        const float tot = gq_wave_allsum((tid & 63u) < (T + 63u) / 64u ? red[tid & 63u] : 0.f);
        scale = 1.0f / sqrtf(tot / (float)G.K + eps);
    }
    for (u32 idx = idx0; idx < 4u * G.Q; idx += T) {
        const u32 q = idx >> 2, c = idx & 3u;
        const u32 e0 = G.xindex(q, 0u, c, 0u);  // 32 consecutive activations: v = 0..3, j = 0..7
        u32 in[4][4];
#pragma unroll
        for (int v = 0; v < 4; v++) {
            if constexpr (PRO == PRO_RMSNORM) {
                u32 nw[4];
                if (idx == idx0) {
#pragma unroll
                    for (int i = 0; i < 4; i++) in[v][i] = in0[v][i], nw[i] = nw0[v][i];
                } else {
                    const uint4 t4 = ld16(x + e0 + 8 * v), n4 = ld16(normw + e0 + 8 * v);
                    in[v][0] = t4.x, in[v][1] = t4.y, in[v][2] = t4.z, in[v][3] = t4.w;
                    nw[0] = n4.x, nw[1] = n4.y, nw[2] = n4.z, nw[3] = n4.w;
                }
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    uint16_t a = f2h(gq_pin_f32(h2f(in[v][i] & 0xFFFF) * scale)), b = f2h(gq_pin_f32(h2f(in[v][i] >> 16) * scale));
                    _Float16 ra = __builtin_bit_cast(_Float16, a) * __builtin_bit_cast(_Float16, (uint16_t)(nw[i] & 0xFFFF));
                    _Float16 rb = __builtin_bit_cast(_Float16, b) * __builtin_bit_cast(_Float16, (uint16_t)(nw[i] >> 16));
                    in[v][i] = (u32)__builtin_bit_cast(uint16_t, ra) | ((u32)__builtin_bit_cast(uint16_t, rb) << 16);
                }
            } else {
                uint4 t4 = ld16(x + e0 + 8 * v);
                in[v][0] = t4.x;
                in[v][1] = t4.y;
                in[v][2] = t4.z;
                in[v][3] = t4.w;
                if constexpr (PRO == PRO_SILUMUL) {
                    uint4 u4 = ld16(x + G.K + e0 + 8 * v);
                    const u32 uw[4] = {u4.x, u4.y, u4.z, u4.w};
#pragma unroll
                    for (int i = 0; i < 4; i++) in[v][i] = silu_mul_pk(in[v][i], uw[i]);
                }
            }
        }
#pragma unroll
        for (u32 mth = 0; mth < 4; mth++) {
            uint4 o;
            o.x = perm(in[1][mth], in[0][mth], 0x05040100u);  // (v0,v1) @ j = 2m
            o.y = perm(in[3][mth], in[2][mth], 0x05040100u);  // (v2,v3) @ j = 2m
            o.z = perm(in[1][mth], in[0][mth], 0x07060302u);  // (v0,v1) @ j = 2m+1
            o.w = perm(in[3][mth], in[2][mth], 0x07060302u);  // (v2,v3) @ j = 2m+1
            *reinterpret_cast<uint4 *>(xlds + (((c * 4u + mth) * G.Q + q) << 3)) = o;
        }
    }
}

// ----------------------------------------------------------------------------------------------
// Final reduction of one row from its per-(chunk, virtual lane) fp16 partials in LDS:
// chunks in ascending order per lane (anyprec.cu:505), then the 16,8,4,2,1 shuffle tree (anyprec.cu:363-370).
// Called by 32 consecutive lanes (t = lane & 31).  Returns the row value in lane t == 0.
// ----------------------------------------------------------------------------------------------
// The chain is what the reference computes; how the values travel is this chip's: the partials of four chunks are read from LDS in
// one round trip, the 16-lane step is one v_permlane16_swap (rows 0 / 1 of the wave exchanged in registers), the 8 / 4 / 2 / 1 steps are
// DPP operands of the adds (row_ror:8, row_shl:4, quad_perm) -- lane t < sh receives lane t + sh exactly as __shfl_down(.., sh, 32) hands
// it over, the other lanes' values are never used.  (Round 6: with one ds_read / ds_bpermute round trip per add the epilogue of a block
// was 0.6 us per pass of T / 32 rows -- 2.8 of the 12.7 us of a w1w3 block; tools/r6/exact_stamps.py.)
__device__ __forceinline__ uint16_t h_add_dpp16(uint16_t p) {  // p[t] + p[t + 16], t < 16 of every 32
    const u32 v = p;
    auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);  // r[0] = rows (0, 0, 2, 2), r[1] = rows (1, 1, 3, 3)
    return h_add((uint16_t)r[0], (uint16_t)r[1]);
}
template <int CTRL>
__device__ __forceinline__ uint16_t h_add_dpp(uint16_t p) {
    return h_add(p, (uint16_t)__builtin_amdgcn_update_dpp(0, (int)(u32)p, CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ uint16_t reduce_tree(uint16_t p) {
    p = h_add_dpp16(p);
    p = h_add_dpp<0x128>(p);  // row_ror:8   lane t <- t + 8 (mod 16)
    p = h_add_dpp<0x104>(p);  // row_shl:4   lane t <- t + 4
    p = h_add_dpp<0x4E>(p);   // quad_perm [2,3,0,1]
    p = h_add_dpp<0xB1>(p);   // quad_perm [1,0,3,2]
    return p;
}
__device__ __forceinline__ uint16_t reduce_row(const RowGeom &G, const uint16_t *sv, u32 t, u32 c0, u32 c1) {
    uint16_t p = 0;
    for (u32 i0 = c0; i0 < c1; i0 += 4u) {
        uint16_t v[4];
        bool ok[4];
#pragma unroll
        for (u32 j = 0; j < 4u; j++) {
            const u32 i = i0 + j;
            ok[j] = i < c1 && !(i == G.nfull && t >= G.eff);  // (the lanes past the end of a partial last chunk hold no partial: no add)
            v[j] = sv[(ok[j] ? i : c0) * 32u + t];
        }
#pragma unroll
        for (u32 j = 0; j < 4u; j++) p = ok[j] ? h_add(p, v[j]) : p;
    }
    return reduce_tree(p);
}

// NCH > 0: rows of exactly NCH full chunks (no predicates, one LDS round trip for all partials of a row)
template <int NCH>
__device__ __forceinline__ uint16_t reduce_row_n(const RowGeom &G, const uint16_t *sv, u32 t) {
    if constexpr (NCH == 0) return reduce_row(G, sv, t, 0u, G.nchunks);
    else {
        uint16_t v[NCH];
#pragma unroll
        for (int i = 0; i < NCH; i++) v[i] = sv[(u32)i * 32u + t];
        uint16_t p = 0;
#pragma unroll
        for (int i = 0; i < NCH; i++) p = h_add(p, v[i]);
        return reduce_tree(p);
    }
}

// Ordered reduction + epilogue of the R rows of a block, 32 lanes per row, TWO rows per 32-lane group and trip (their LDS round trips
// and add chains are independent: the compiler interleaves them); the residual is requested before the row's partials are read.
template <int NCH>
__device__ __forceinline__ void rows_epilogue_n(const RowGeom &G, const ApArgs &a, const uint16_t *sv, u32 svrow, u32 R, u32 row0, u32 m,
                                                u32 tid, u32 T) {
    const u32 t = tid & 31u, stride = T >> 5;
    const bool pairs = a.epilogue & GQ_EPI_SILU_PAIRS;
    for (u32 r0 = tid >> 5; r0 < R; r0 += 2u * stride) {
        const u32 r1 = r0 + stride;
        const bool has1 = r1 < R;
        const u32 row[2] = {row0 + r0, row0 + r1};
        const bool live[2] = {true, has1};
        uint16_t res[2] = {0, 0};
        if (!pairs && a.resid) {
#pragma unroll
            for (int e = 0; e < 2; e++)
                if (live[e] && row[e] < a.N) res[e] = a.resid[(size_t)m * a.N + row[e]];
        }
        uint16_t y[2];
        y[0] = reduce_row_n<NCH>(G, sv + (size_t)r0 * svrow, t);
        y[1] = reduce_row_n<NCH>(G, sv + (size_t)(has1 ? r1 : r0) * svrow, t);
#pragma unroll
        for (int e = 0; e < 2; e++) {
            if (pairs) {
                // rows (2i, 2i+1) = (gate_i, up_i) sit in the two halves of the wave: F.silu(gate) * up on fp16 values
                // (inference/model.py:266), written to out[i]
                const u32 yy = y[e];
                auto sw = __builtin_amdgcn_permlane32_swap(yy, yy, false, false);  // sw[1]: lanes 32..63 of y in both halves
                const uint16_t yo = (uint16_t)sw[1];
                if (live[e] && t == 0 && !(tid & 32u) && row[e] + 1u < a.N) {
                    const float gv = (float)__builtin_bit_cast(_Float16, y[e]);
                    const _Float16 o = (_Float16)(gv / (1.0f + __expf(-gv))) * __builtin_bit_cast(_Float16, yo);
                    a.out[(size_t)m * (a.N >> 1) + (row[e] >> 1)] = __builtin_bit_cast(uint16_t, o);
                }
            } else if (live[e] && t == 0 && row[e] < a.N) {
                if (a.resid) y[e] = h_add(res[e], y[e]);
                a.out[(size_t)m * a.N + row[e]] = y[e];
            }
        }
    }
}
__device__ __forceinline__ void rows_epilogue(const RowGeom &G, const ApArgs &a, const uint16_t *sv, u32 svrow, u32 R, u32 row0, u32 m, u32 tid,
                                              u32 T) {
    // (compiled-in chunk counts of the Llama rows: K = 4096, 8192, 14336; everything else -- partial last chunks too -- on the general one)
    const u32 nch = G.eff ? 0u : G.nchunks;
    if (nch == 4u) rows_epilogue_n<4>(G, a, sv, svrow, R, row0, m, tid, T);
    else if (nch == 8u) rows_epilogue_n<8>(G, a, sv, svrow, R, row0, m, tid, T);
    else if (nch == 14u) rows_epilogue_n<14>(G, a, sv, svrow, R, row0, m, tid, T);
    else rows_epilogue_n<0>(G, a, sv, svrow, R, row0, m, tid, T);
}

// ----------------------------------------------------------------------------------------------
// Fast path: BITS in {2,3,4}, K % 128 == 0, Q = K/128 <= blockDim.x.
//
// A block owns SPB consecutive "row steps" (one step = RS = blockDim/Q consecutive rows, one item per lane,
// a wave-load is 1 KiB of contiguous plane bytes).  The plane words + LUT of D steps ahead sit in a register
// ring, loaded with raw buffer loads: rows past the end get an out-of-range offset, which the hardware
// answers with zeros and no memory traffic, so the loop has no divergent control flow and the compiler's
// vmcnt bookkeeping stays exact.  The lane's activations stay in 64 VGPRs for the whole loop.
// ----------------------------------------------------------------------------------------------
// ----------------------------------------------------------------------------------------------
// INREG (rows of 4096 weights: Q = 32, a row is one half of a wave): the ordered reduction of a row step stays in registers.
// Lane hl = q of the half-wave holds the packed chunk partials of CUDA lanes t = 4 (q % 8) + v, v = 0..3, of chunk q / 8.  The chunk
// sums of anyprec.cu:505 (chunks ascending, starting from +0) meet in lanes hl < 8: own value, lane + 8 (row_shl:8), lane + 16
// (v_permlane16_swap), lane + 24 (row_shl:8 of that); the shuffle tree of anyprec.cu:363-370 is t + 16 = lane + 4 (row_shl:4), t + 8
// = lane ^ 2, t + 4 = lane ^ 1 (quad_perm), t + 2 = the second register, t + 1 = the upper half -- the same additions in the same
// order, ~30 VALU per step instead of the LDS partials + block barrier + reduction pass (1.2 of the 11.6 us of a w1w3 block,
// 0.6 of the 3.4 us of a wo block: profiles/r06_exact_epilogue.txt).  The row's value lands in lane hl = 0.
// ----------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ u32 mov_dpp(u32 v) { return (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true); }
__device__ __forceinline__ u32 inreg_chunk_sum4(u32 s) {
    const u32 c1 = mov_dpp<0x108>(s);  // row_shl:8
    auto sw = __builtin_amdgcn_permlane16_swap(s, s, false, false);
    const u32 c2 = sw[1];
    const u32 c3 = mov_dpp<0x108>(c2);
    u32 p = pk_add(0u, s);
    p = pk_add(p, c1);
    p = pk_add(p, c2);
    return pk_add(p, c3);
}
__device__ __forceinline__ uint16_t inreg_reduce_row(u32 s01, u32 s23) {
    u32 p01 = inreg_chunk_sum4(s01), p23 = inreg_chunk_sum4(s23);
    p01 = pk_add(p01, mov_dpp<0x104>(p01)), p23 = pk_add(p23, mov_dpp<0x104>(p23));  // t + 16: row_shl:4
    p01 = pk_add(p01, mov_dpp<0x4E>(p01)), p23 = pk_add(p23, mov_dpp<0x4E>(p23));    // t + 8: quad_perm [2,3,0,1]
    p01 = pk_add(p01, mov_dpp<0xB1>(p01)), p23 = pk_add(p23, mov_dpp<0xB1>(p23));    // t + 4: quad_perm [1,0,3,2]
    const u32 pp = pk_add(p01, p23);                                                // t + 2
    return h_add((uint16_t)(pp & 0xFFFFu), (uint16_t)(pp >> 16));                  // t + 1
}

template <int BITS, int D, int PRO, bool INREG = false>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(INREG && BITS == 2 ? 4 : ((BITS <= 3 || INREG) && D == 1 ? 3 : 2))))
ap_gemv_quad_kernel(ApArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    RowGeom G;
    G.init(a.K);
    uint16_t *xlds = reinterpret_cast<uint16_t *>(smem);
    uint16_t *sv = xlds + G.K;  // [SPB*RS][nchunks*32]
    const u32 svrow = G.nchunks * 32u;
    float *red = reinterpret_cast<float *>(sv);  // scratch for the RMSNorm reduction (before sv is used)

    const u32 T = blockDim.x, tid = threadIdx.x;
    const u32 RS = a.RS, SPB = a.SPB;
    const u32 row0 = blockIdx.x * SPB * RS;
    const u32 m = blockIdx.y;
    const bool active = tid < RS * G.Q;
    const u32 rs = active ? tid / G.Q : 0u;
    const u32 q = active ? tid - rs * G.Q : 0u;
    // (GQ_STAMPS builds: s_memrealtime per wave of the middle block -- start, x in registers, after every step, loop barrier, end)
    // (GQ_STAMPS == 2: start / end of EVERY block instead -- wave 0 of block b writes dbg[2 b], dbg[2 b + 1]: the dispatch ramp of a launch)
    unsigned long long *qdbg = (GQ_STAMPS == 1 && a.dbg && blockIdx.x == gridDim.x / 2u && (tid & 63u) == 0u) ? a.dbg + (size_t)(tid >> 6) * 32u
                               : ((GQ_STAMPS == 2 && a.dbg && tid == 0u && blockIdx.y == 0u) ? a.dbg + 2u * (size_t)blockIdx.x : nullptr);
    u32 nst = 0;
    auto qstamp = [&]() {
        if (GQ_STAMPS == 1 && qdbg && nst < 32u) qdbg[nst++] = __builtin_amdgcn_s_memrealtime();
    };
    qstamp();
    if (GQ_STAMPS == 2 && qdbg) qdbg[0] = __builtin_amdgcn_s_memrealtime();

    constexpr int NRAW = (1 << BITS) / 2;
    constexpr u32 OOB = 0x80000000u;
    const u32 plane_bytes = a.N * G.wpr * 4u;
    __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void *)a.qw, 0, (int)(plane_bytes * (u32)BITS), 0x00020000);
    __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc((void *)a.lut, 0, (int)(a.N * (u32)NRAW * 4u), 0x00020000);

    u32x4 P[D][BITS];
    u32 lraw[D][NRAW];
    auto issue = [&](int d, u32 step) {
        const u32 row = row0 + step * RS + rs;
        const bool ok = active && step < SPB && row < a.N;
        const u32 off = ok ? (row * G.wpr + 4u * q) * 4u : OOB;
#pragma unroll
        for (int p = 0; p < BITS; p++)
            P[d][p] = __builtin_amdgcn_raw_buffer_load_b128(rq, ok ? off + (u32)p * plane_bytes : OOB, 0, 2 /* nt */);
        const u32 loff = ok ? row * (u32)NRAW * 4u : OOB;
        if constexpr (NRAW == 2) {
            auto v = __builtin_amdgcn_raw_buffer_load_b64(rl, loff, 0, 0);
            lraw[d][0] = v[0];
            lraw[d][1] = v[1];
        } else {
#pragma unroll
            for (int i = 0; i < NRAW; i += 4) {
                u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rl, ok ? loff + (u32)i * 4u : OOB, 0, 0);
                lraw[d][i] = v.x;
                lraw[d][i + 1] = v.y;
                lraw[d][i + 2] = v.z;
                lraw[d][i + 3] = v.w;
            }
        }
    };

    // 1. vector-memory results return in order: waves that stage x must not have plane loads queued in front of
    //    their x loads.  Upper half of the waves: fill the ring now; lower half: stage x first, then fill.
    const u32 nw = T >> 6, wv = tid >> 6;
    const bool stager = nw < 2u || wv < nw / 2u;
    if (!stager) {
#pragma unroll
        for (int d = 0; d < D; d++) issue(d, (u32)d);
    }
    // 2. activations -> LDS (lane-linear image) -> 64 VGPRs
    stage_x<PRO>(G, a.x + (size_t)m * (PRO == PRO_SILUMUL ? 2u * G.K : G.K), a.normw, a.eps, xlds, red, stager,
                 nw < 2u ? T : (nw / 2u) * 64u);
    if (stager) {
#pragma unroll
        for (int d = 0; d < D; d++) issue(d, (u32)d);
    }
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
    if (GQ_STAMPS == 1) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        qstamp();
    }

    // 3. decode + packed fp16 FMA chains, one item per step
    u32 chunk, t0, tpw;
    G.quad(q, chunk, t0, tpw);
    uint16_t *svp = sv + (size_t)rs * svrow + chunk * 32u + t0;
    for (u32 i = 0; i < SPB; i += D) {
#pragma unroll
        for (int d = 0; d < D; d++) {
            if (GQ_STAMPS == 1) {  // (words in hand)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                qstamp();
            }
            u32 Pw[BITS][4];
#pragma unroll
            for (int p = 0; p < BITS; p++) {
                Pw[p][0] = P[d][p].x;
                Pw[p][1] = P[d][p].y;
                Pw[p][2] = P[d][p].z;
                Pw[p][3] = P[d][p].w;
            }
            LutPools<BITS> L;
            L.build(lraw[d]);
            issue(d, i + (u32)d + (u32)D);  // refill this ring slot (its registers are now copied out)
            const u32 row = row0 + (i + (u32)d) * RS + rs;
            const bool rvalid = i + (u32)d < SPB && row < a.N;
            uint16_t res = 0;
            if constexpr (INREG) {
                if (a.resid && !(a.epilogue & GQ_EPI_SILU_PAIRS) && rvalid) res = a.resid[(size_t)m * a.N + row];
            }
            u32 s01, s23;
            Item<BITS>::run(Pw, L, xr, s01, s23);
            if constexpr (INREG) {
                uint16_t y = inreg_reduce_row(s01, s23);
                const bool lead = (tid & 31u) == 0u;
                if (a.epilogue & GQ_EPI_SILU_PAIRS) {
                    // rows (2i, 2i+1) = (gate_i, up_i) sit in the two halves of the wave: F.silu(gate) * up on fp16 values
                    // (inference/model.py:266), written to out[i]
                    const u32 yy = y;
                    auto sw = __builtin_amdgcn_permlane32_swap(yy, yy, false, false);  // sw[1]: lanes 32..63 of y in both halves
                    if (lead && !(tid & 32u) && rvalid && row + 1u < a.N) {
                        const float gv = (float)__builtin_bit_cast(_Float16, y);
                        const _Float16 o = (_Float16)(gv / (1.0f + __expf(-gv))) * __builtin_bit_cast(_Float16, (uint16_t)sw[1]);
                        a.out[(size_t)m * (a.N >> 1) + (row >> 1)] = __builtin_bit_cast(uint16_t, o);
                    }
                } else if (lead && rvalid) {
                    if (a.resid) y = h_add(res, y);
                    a.out[(size_t)m * a.N + row] = y;
                }
            } else {
                if (active && i + (u32)d < SPB)
                    *reinterpret_cast<uint2 *>(svp + (size_t)(i + (u32)d) * RS * svrow) = make_uint2(s01, s23);
            }
            if (GQ_STAMPS) qstamp();
        }
    }
    if constexpr (INREG) {
        if (GQ_STAMPS) qstamp();
        if (GQ_STAMPS == 2 && qdbg) qdbg[1] = __builtin_amdgcn_s_memrealtime();
        return;
    }
    __syncthreads();
    if (GQ_STAMPS) qstamp();

    // 4. ordered reduction + epilogue
    rows_epilogue(G, a, sv, svrow, SPB * RS, row0, m, tid, T);
    if (GQ_STAMPS) qstamp();
    if (GQ_STAMPS == 2 && qdbg) qdbg[1] = __builtin_amdgcn_s_memrealtime();
}

// ----------------------------------------------------------------------------------------------
// Round 6 -- exact mode, 2 bits: the reference's LDS PAIR TABLE (anyprec.cu:397-419,454-490) on wave64.
//
// ap_gemv_quad_kernel decodes with v_perm_b32 out of VGPR byte pools: 2.25 VALU operations per weight, the kernel is VALU-bound
// (DESIGN.md section 3.1).  The reference looks a PAIR of weights up in a per-row table of 2^b x 2^b half2 entries in shared memory:
// one load per two weights, and the entry IS the hfma2 operand (even weight, odd weight).  Here the same table -- 16 entries of 4
// bytes per row at 2 bits -- lives in LDS per (wave, row of the wave: the lanes of a wave span at most two rows), rebuilt per row
// step by 32 lanes with one v_perm and one ds_write each; a lane (row slot, quad) forms the sixteen 4-bit indices of a plane-word
// pair with two v_bfi (M1 / M2 below), and per pair issues one v_lshrrev + one v_and_or (byte address) + one ds_read_b32 + one
// v_pk_fma_f16: 3.25 VALU per PAIR instead of 4.5, the look-ups on the LDS pipe (16 words of one table sit in 16 different banks,
// equal addresses broadcast: conflict-free).  The fp16 chains are the reference's own: per word (CUDA lane) ONE half2 accumulator
// (even chain, odd chain), byte c = 3..0, pair k = 0..3 (anyprec.cu:495-504), then sum.x + sum.y -- bit-identical to
// ap_gemv_quad_kernel and to the order-faithful oracle (tests/test_ap_gemv_gpu.py runs both kernels over every exact-mode case).
// The activation image in LDS is x in its natural order (a lane's 8 activations of (word v, byte c) are 16 contiguous bytes).
// ----------------------------------------------------------------------------------------------
template <int PRO>
__device__ __forceinline__ void stage_x_natural(const RowGeom &G, const uint16_t *x, const uint16_t *normw, float eps, uint16_t *xlds, float *red,
                                                bool stager, u32 T) {
    // 16-byte units (8 activations) per stager thread; same arithmetic and rounding points as stage_x
    const u32 tid = threadIdx.x, nun = G.K / 8u;
    float scale = 0.f;
    if constexpr (PRO == PRO_RMSNORM) {
        // (the sum of squares in stage_x's order -- thread idx takes the 32 activations of (quad idx / 4, byte idx % 4) -- so that both
        // exact kernels normalise with the same fp32 statistic, bit for bit)
        float ss = 0.f;
        for (u32 idx = stager ? tid : 4u * G.Q; idx < 4u * G.Q; idx += T) {
            const u32 e0 = G.xindex(idx >> 2, 0u, idx & 3u, 0u);
#pragma unroll
            for (int v = 0; v < 4; v++) {
                const uint4 t4 = ld16(x + e0 + 8 * v);
                const u32 w[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const float a = h2f(w[i] & 0xFFFF), b = h2f(w[i] >> 16);
                    ss += a * a;
                    ss += b * b;
                }
            }
        }
        ss = gq_wave_allsum(ss);
        if ((tid & 63u) == 0 && stager) red[tid >> 6] = ss;
        __syncthreads();
        // (the stager waves' sums: one LDS round trip + the same register tree, instead of one dependent ds_read per wave)
        const float tot = gq_wave_allsum((tid & 63u) < (T + 63u) / 64u ? red[tid & 63u] : 0.f);
        scale = 1.0f / sqrtf(tot / (float)G.K + eps);
    }
    for (u32 u = stager ? tid : nun; u < nun; u += T) {
        uint4 t4 = ld16(x + 8u * u);
        u32 in[4] = {t4.x, t4.y, t4.z, t4.w};
        if constexpr (PRO == PRO_RMSNORM) {
            const uint4 n4 = ld16(normw + 8u * u);
            const u32 nw[4] = {n4.x, n4.y, n4.z, n4.w};
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint16_t a = f2h(gq_pin_f32(h2f(in[i] & 0xFFFF) * scale)), b = f2h(gq_pin_f32(h2f(in[i] >> 16) * scale));
                const _Float16 ra = __builtin_bit_cast(_Float16, a) * __builtin_bit_cast(_Float16, (uint16_t)(nw[i] & 0xFFFF));
                const _Float16 rb = __builtin_bit_cast(_Float16, b) * __builtin_bit_cast(_Float16, (uint16_t)(nw[i] >> 16));
                in[i] = (u32)__builtin_bit_cast(uint16_t, ra) | ((u32)__builtin_bit_cast(uint16_t, rb) << 16);
            }
        }
        if constexpr (PRO == PRO_SILUMUL) {
            const uint4 u4 = ld16(x + G.K + 8u * u);
            const u32 uw[4] = {u4.x, u4.y, u4.z, u4.w};
#pragma unroll
            for (int i = 0; i < 4; i++) in[i] = silu_mul_pk(in[i], uw[i]);
        }
        *reinterpret_cast<uint4 *>(xlds + 8u * u) = make_uint4(in[0], in[1], in[2], in[3]);
    }
}

template <int PRO>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4))) ap_gemv_pt2_kernel(ApArgs a) {  // (105 VGPRs)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int BITS = 2, NRAW = 2;
    RowGeom G;
    G.init(a.K);
    uint16_t *xlds = reinterpret_cast<uint16_t *>(smem);
    uint16_t *sv = xlds + G.K;  // [SPB*RS][nchunks*32]
    const u32 svrow = G.nchunks * 32u;
    float *red = reinterpret_cast<float *>(sv);
    const u32 T = blockDim.x, tid = threadIdx.x, l = tid & 63u, wv = tid >> 6;
    const u32 RS = a.RS, SPB = a.SPB;
    const u32 row0 = blockIdx.x * SPB * RS;
    const u32 m = blockIdx.y;
    const bool active = tid < RS * G.Q;
    const u32 rs = active ? tid / G.Q : 0u;
    const u32 q = active ? tid - rs * G.Q : 0u;
    // pair tables of this wave: [2 rows of the wave][16 entries] words, behind sv
    u32 *tabs = reinterpret_cast<u32 *>(smem + (((size_t)G.K * 2u + (size_t)SPB * RS * svrow * 2u + 63u) & ~(size_t)63u)) + wv * 32u;
    constexpr u32 OOB = 0x80000000u;
    const u32 plane_bytes = a.N * G.wpr * 4u;
    __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void *)a.qw, 0, (int)(plane_bytes * (u32)BITS), 0x00020000);
    __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc((void *)a.lut, 0, (int)(a.N * (u32)NRAW * 4u), 0x00020000);
    u32x4 P[BITS];
    u32 lraw[NRAW];
    auto issue = [&](u32 step) {
        const u32 row = row0 + step * RS + rs;
        const bool ok = active && step < SPB && row < a.N;
        const u32 off = ok ? (row * G.wpr + 4u * q) * 4u : OOB;
#pragma unroll
        for (int p = 0; p < BITS; p++) P[p] = __builtin_amdgcn_raw_buffer_load_b128(rq, ok ? off + (u32)p * plane_bytes : OOB, 0, 2 /* nt */);
        auto v = __builtin_amdgcn_raw_buffer_load_b64(rl, ok ? row * (u32)NRAW * 4u : OOB, 0, 0);
        lraw[0] = v[0];
        lraw[1] = v[1];
    };
    const u32 nw = T >> 6;
    const bool stager = nw < 2u || wv < nw / 2u;
    if (!stager) issue(0u);
    stage_x_natural<PRO>(G, a.x + (size_t)m * (PRO == PRO_SILUMUL ? 2u * G.K : G.K), a.normw, a.eps, xlds, red, stager, nw < 2u ? T : (nw / 2u) * 64u);
    if (stager) issue(0u);
    __syncthreads();
    // the lane's activations: xr[v][c] = the 8 activations of (word v, byte c) = 4 half2 operands (pair k = 0..3)
    u32 chunk, t0, tpw;
    G.quad(q, chunk, t0, tpw);
    uint4 xr[4][4];
#pragma unroll
    for (u32 v = 0; v < 4; v++)
#pragma unroll
        for (u32 c = 0; c < 4; c++) xr[v][c] = *reinterpret_cast<const uint4 *>(xlds + 1024u * chunk + 8u * tpw * c + 8u * (t0 + v));
    // table builders: lanes 0..15 build the table of the row of lane 0, lanes 16..31 that of the row of lane 63; entry i = (h_even,
    // h_odd, l_even, l_odd) -> half2(lut[2 h_even + l_even], lut[2 h_odd + l_odd]): the selector of the lane's entry is a constant
    const u32 ei = l & 15u;
    const u32 ce = ((ei >> 3) & 1u) * 2u + ((ei >> 1) & 1u), co = ((ei >> 2) & 1u) * 2u + (ei & 1u);
    const u32 esel = (2u * ce) | ((2u * ce + 1u) << 8) | ((2u * co) << 16) | ((2u * co + 1u) << 24);
    // which of the wave's two tables this lane reads: the row slot of lane 0 -> table 0, else table 1 (byte offset 64)
    // (the last ACTIVE lane of the wave carries its second row; a wave without active lanes builds tables nobody reads)
    const unsigned long long act = __builtin_amdgcn_ballot_w64(active);
    const u32 hi_lane = act ? 63u - (u32)__builtin_clzll(act) : 0u;
    const u32 rs_lo = (u32)__builtin_amdgcn_readfirstlane((int)rs);
    const u32 tbase = (u32)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)reinterpret_cast<unsigned char *>(tabs) + (rs == rs_lo ? 0u : 64u);
    uint16_t *svp = sv + (size_t)rs * svrow + chunk * 32u + t0;
    for (u32 i = 0; i < SPB; i++) {
        u32 Pw[BITS][4];
#pragma unroll
        for (int p = 0; p < BITS; p++) {
            Pw[p][0] = P[p].x;
            Pw[p][1] = P[p].y;
            Pw[p][2] = P[p].z;
            Pw[p][3] = P[p].w;
        }
        // this step's tables (the LDS serves a wave's operations in order: the previous step's look-ups are behind us)
        {
            const u32 lo0 = (u32)__builtin_amdgcn_readlane((int)lraw[0], 0), lo1 = (u32)__builtin_amdgcn_readlane((int)lraw[1], 0);
            const u32 hi0 = (u32)__builtin_amdgcn_readlane((int)lraw[0], (int)hi_lane), hi1 = (u32)__builtin_amdgcn_readlane((int)lraw[1], (int)hi_lane);
            const u32 ent = l < 16u ? perm(lo1, lo0, esel) : perm(hi1, hi0, esel);
            if (l < 32u) tabs[l] = ent;
        }
        issue(i + 1u);
        u32 s[4];
#pragma unroll
        for (int v = 0; v < 4; v++) {
            // nibble n of M1: bits (4 n + 3, 4 n + 2) of H and L; of M2: bits (4 n + 1, 4 n).  Byte c of a word = bits 8 (3 - c) ..;
            // pair k of byte c = bits (7 - 2 k, 6 - 2 k): k = 0, 2 from M1 (upper, lower nibble of the byte), k = 1, 3 from M2
            const u32 M1 = bfi(0xCCCCCCCCu, Pw[0][v], Pw[1][v] >> 2), M2 = bfi(0xCCCCCCCCu, Pw[0][v] << 2, Pw[1][v]);
            u32 acc = 0u;
#pragma unroll
            for (int c = 3; c >= 0; c--) {
                const u32 xw[4] = {xr[v][c].x, xr[v][c].y, xr[v][c].z, xr[v][c].w};
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const u32 Mk = (k & 1) ? M2 : M1;
                    const int nib = 2 * (3 - c) + ((k & 2) ? 0 : 1);  // nibble index inside the word
                    const int sh = 4 * nib - 2;                       // bring the nibble to bits 5..2 (= index * 4)
                    const u32 addr = ((sh >= 0 ? (Mk >> sh) : (Mk << 2)) & 0x3Cu) | tbase;
                    // (an LDS-address-space load of the byte offset: through the generic `smem + addr` every look-up paid a v_add for the base)
                    const u32 w2 = *(const __attribute__((address_space(3))) u32 *)(uintptr_t)addr;
                    acc = pk_fma(w2, xw[k], acc);
                }
            }
            s[v] = (u32)h_add((uint16_t)(acc & 0xFFFFu), (uint16_t)(acc >> 16));
        }
        if (active) *reinterpret_cast<uint2 *>(svp + (size_t)i * RS * svrow) = make_uint2(s[0] | (s[1] << 16), s[2] | (s[3] << 16));
    }
    __syncthreads();
    // ordered reduction + epilogue: as ap_gemv_quad_kernel
    rows_epilogue(G, a, sv, svrow, SPB * RS, row0, m, tid, T);
}

// ----------------------------------------------------------------------------------------------
// Round 6 -- "decode to fp16, accumulate on the matrix cores" (fast mode, 3 and 4 bits, one batch row).
//
// The plane-MFMA kernels (ap_plane.hip / ap_stream.hip) multiply 2^b - 1 BINARY matrices per b-bit LUT: 3 at 2 bits, 7 at 3, 15
// at 4 -- 24 / 56 / 120 FP4 x BF8 MFMAs of 32 cycles per 16 rows x 1024 weights, with ~8 VALU instructions per MFMA to build the
// operands: at 4 bits ~1,040 instructions per block, instruction-issue-bound at 2.3 x the matrix floor (w1w3 24 us, 0.30 of the
// HBM roofline).  The decode of the exact kernels (ap_core.h: v_perm_b32 look-ups in the row's LUT held as VGPR byte pools) grows far
// more slowly with b -- 1.6 / 2.2 / 3.9 VALU operations per weight -- and what it yields, packed fp16 pairs of four weights at one bit
// position, IS an A fragment of v_mfma_f32_16x16x32_f16: lane (row r = l % 16, quad g = l / 16) holds 8 weights of row r, and the K
// index of a matrix-core product is only a label -- any 8 weights do, as long as the B fragment of lane (column, g) holds the 8
// activations that belong to them: one 16-byte slot of an activation image staged in LDS in the decode's own order (the first version
// used the exact kernels' image, ap_core.h::xlds_pos, and their byte transpose of the quad; the second -- DqItem / stage_x_dqv below --
// drops the transpose: profiles/r06_dq_kernel.txt).  So:
//   * A = two look-ups (4 VGPRs), B = one ds_read_b128 of the staged image (column 0 reads it, columns 1..15 read zeros from
//     outside the block's LDS allocation -- idle columns fed real data cost 8 % in clocks), one MFMA
//     of 16 cycles per 16 rows x 32 weights: 32 per 16 x 1024 block instead of 56 / 120, no bf8 pieces, no image build, no
//     extraction of large activations, no Moebius coefficients;
//   * products of fp16 values are exact in the fp32 accumulator, sums in fp32, one fp16 rounding: the fast-mode envelope
//     (tests/ap_helpers.py::_check_fast) -- not the reference's fp16 accumulation order (that is the exact mode);
//   * prologues (RMSNorm, SiLU * up) as stage_x has them, with the reference's rounding points; plain / residual / gate-up pair epilogues.
// Work: item = (16-row group, unit of 4 quads = 512 weights per row); wave w of a 16-wave block takes items w, w + 16, ..; the
// K-split partial sums of a row group meet in LDS and are added in unit order (deterministic).
// ----------------------------------------------------------------------------------------------
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int BITS>
struct DqItem;  // run(P, L, emit): emit(v, k, w01 @ 2 k, w23 @ 2 k, w01 @ 2 k + 1, w23 @ 2 k + 1) for word v = 0..3, pair k = 0..3
// NO byte transpose of the quad (ap_core.h::transpose_quad, 8 v_perm per plane): the exact kernels need it to keep the reference's
// chain order, a matrix-core product does not care which 8 weights share an operand.  A selector taken from ONE word holds the codes
// of the same bit position j of its four bytes c = 3, 2, 1, 0 (byte 0 = c 3: pack.py's byte order); the look-up returns
// w01 = (w[c3], w[c2]), w23 = (w[c1], w[c0]) at that j -- the activation image (stage_x_dqv) keeps the matching order.
template <>
struct DqItem<2> {
    template <typename E>
    __device__ __forceinline__ static void run(const u32 P[2][4], const LutPools<2> &L, E emit) {
#pragma unroll
        for (int v = 0; v < 4; v++) {
            const u32 Cm = bfi(0xAAAAAAAAu, P[0][v], P[1][v] >> 1), Dm = bfi(0xAAAAAAAAu, P[0][v] << 1, P[1][v]);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                u32 a01, a23, b01, b23;
                lookup4<2>(L, (Cm >> (6 - 2 * k)) & 0x03030303u, 0u, a01, a23);
                lookup4<2>(L, (Dm >> (6 - 2 * k)) & 0x03030303u, 0u, b01, b23);
                emit(v, k, a01, a23, b01, b23);
            }
        }
    }
};
template <>
struct DqItem<3> {
    template <typename E>
    __device__ __forceinline__ static void run(const u32 P[3][4], const LutPools<3> &L, E emit) {
#pragma unroll
        for (int v = 0; v < 4; v++) {
            u32 nib[4];
#pragma unroll
            for (int r = 0; r < 4; r++) nib[r] = bfi(0x44444444u, shl(P[0][v], 2 - r), bfi(0x22222222u, shl(P[1][v], 1 - r), P[2][v] >> r));
#pragma unroll
            for (int k = 0; k < 4; k++) {
                u32 w[2][2];
#pragma unroll
                for (int par = 0; par < 2; par++) {
                    const int s = 7 - (2 * k + par);
                    const u32 S = ((s & 4) ? (nib[s & 3] >> 4) : nib[s & 3]) & 0x07070707u;
                    lookup4<3>(L, S, 0u, w[par][0], w[par][1]);
                }
                emit(v, k, w[0][0], w[0][1], w[1][0], w[1][1]);
            }
        }
    }
};
template <>
struct DqItem<4> {
    template <typename E>
    __device__ __forceinline__ static void run(const u32 P[4][4], const LutPools<4> &L, E emit) {
#pragma unroll
        for (int v = 0; v < 4; v++) {
            u32 nib[4];
#pragma unroll
            for (int r = 0; r < 4; r++) nib[r] = bfi(0x44444444u, shl(P[1][v], 2 - r), bfi(0x22222222u, shl(P[2][v], 1 - r), P[3][v] >> r));
#pragma unroll
            for (int k = 0; k < 4; k++) {
                u32 w[2][2];
#pragma unroll
                for (int par = 0; par < 2; par++) {
                    const int s = 7 - (2 * k + par);
                    const u32 S = ((s & 4) ? (nib[s & 3] >> 4) : nib[s & 3]) & 0x07070707u;
                    // the MSB plane picks the pool: 0x00 / 0xFF per byte as x * 255 = (x << 8) - x (mod 2^32: exact for bytes of 0 / 1) -- two
                    // VOP2 operations instead of ap_core.h's selector OR + v_perm.  (A 24-bit multiply by 255 is one operation and WRONG: it
                    // drops byte 3 -- 531 tokens/s and 11 failed tests.)
                    const u32 mb = (P[0][v] >> s) & 0x01010101u;
                    const u32 m = (mb << 8) - mb;
                    const u32 lo = bfi(m, perm(L.lo[3], L.lo[2], S), perm(L.lo[1], L.lo[0], S));
                    const u32 hi = bfi(m, perm(L.hi[3], L.hi[2], S), perm(L.hi[1], L.hi[0], S));
                    w[par][0] = perm(hi, lo, 0x05010400u);
                    w[par][1] = perm(hi, lo, 0x07030602u);
                }
                emit(v, k, w[0][0], w[0][1], w[1][0], w[1][1]);
            }
        }
    }
};

// The activation image of the dq kernel: 16-byte slot (v * 4 + k) * Q + q = the 8 activations that meet the A operand of (quad q,
// word v, pair k): [x(c3, 2k), x(c2, 2k), x(c1, 2k), x(c0, 2k), x(c3, 2k+1), x(c2, 2k+1), x(c1, 2k+1), x(c0, 2k+1)], x(c, j) =
// x[G.xindex(q, v, c, j)].  Work item = (quad q, word v): four 16-byte loads (one per byte c), four v_perm per slot.  Prologues as stage_x.
template <int PRO>
__device__ __forceinline__ void stage_x_dqv(const RowGeom &G, const uint16_t *x, const uint16_t *normw, float eps, uint16_t *xlds, float *red, bool stager,
                                            u32 T) {
    const u32 tid = threadIdx.x;
    float scale = 0.f;
    const u32 idx0 = stager ? tid : 4u * G.Q;
    // (K % 1024 == 0 for this kernel: whole chunks only -- item idx = (quad q = idx / 4, word v = idx % 4) starts at activation
    // 1024 (q / 8) + 8 (4 (q % 8) + v) = 1024 (idx / 32) + 8 (idx % 32); byte c adds 256)
    auto xbase = [](u32 idx) -> u32 { return 1024u * (idx >> 5) + 8u * (idx & 31u); };
    // RMSNorm in ONE memory round trip (as stage_x): the activations AND the norm weights of a thread's first item are requested up front,
    // the sum of squares is taken from those registers, and they are normalised behind the barrier without being read again
    u32 in0[4][4], nw0[4][4];
    if constexpr (PRO == PRO_RMSNORM) {
        float ss = 0.f;
        for (u32 idx = idx0; idx < 4u * G.Q; idx += T) {
#pragma unroll
            for (u32 c = 0; c < 4; c++) {
                const u32 e0 = xbase(idx) + 256u * c;
                const uint4 t4 = ld16(x + e0);
                const u32 w[4] = {t4.x, t4.y, t4.z, t4.w};
                if (idx == idx0) {
                    const uint4 n4 = ld16(normw + e0);
                    in0[c][0] = t4.x, in0[c][1] = t4.y, in0[c][2] = t4.z, in0[c][3] = t4.w;
                    nw0[c][0] = n4.x, nw0[c][1] = n4.y, nw0[c][2] = n4.z, nw0[c][3] = n4.w;
                }
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const float a = h2f(w[i] & 0xFFFF), b = h2f(w[i] >> 16);
                    ss += a * a;
                    ss += b * b;
                }
            }
        }
        ss = gq_wave_allsum(ss);
        if ((tid & 63u) == 0 && stager) red[tid >> 6] = ss;
        __syncthreads();
        // (the stager waves' sums: one LDS round trip + the same register tree, instead of one dependent ds_read per wave)
        const float tot = gq_wave_allsum((tid & 63u) < (T + 63u) / 64u ? red[tid & 63u] : 0.f);
        scale = 1.0f / sqrtf(tot / (float)G.K + eps);
    }
    for (u32 idx = idx0; idx < 4u * G.Q; idx += T) {
        const u32 q = idx >> 2, v = idx & 3u;
        u32 in[4][4];  // [c][k]: (x(c, 2k), x(c, 2k+1))
#pragma unroll
        for (u32 c = 0; c < 4; c++) {
            const u32 e0 = xbase(idx) + 256u * c;
            if constexpr (PRO == PRO_RMSNORM) {
                u32 nw[4];
                if (idx == idx0) {
#pragma unroll
                    for (int i = 0; i < 4; i++) in[c][i] = in0[c][i], nw[i] = nw0[c][i];
                } else {
                    const uint4 t4 = ld16(x + e0), n4 = ld16(normw + e0);
                    in[c][0] = t4.x, in[c][1] = t4.y, in[c][2] = t4.z, in[c][3] = t4.w;
                    nw[0] = n4.x, nw[1] = n4.y, nw[2] = n4.z, nw[3] = n4.w;
                }
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const uint16_t a = f2h(gq_pin_f32(h2f(in[c][i] & 0xFFFF) * scale)), b = f2h(gq_pin_f32(h2f(in[c][i] >> 16) * scale));
                    const _Float16 ra = __builtin_bit_cast(_Float16, a) * __builtin_bit_cast(_Float16, (uint16_t)(nw[i] & 0xFFFF));
                    const _Float16 rb = __builtin_bit_cast(_Float16, b) * __builtin_bit_cast(_Float16, (uint16_t)(nw[i] >> 16));
                    in[c][i] = (u32)__builtin_bit_cast(uint16_t, ra) | ((u32)__builtin_bit_cast(uint16_t, rb) << 16);
                }
            } else {
                const uint4 t4 = ld16(x + e0);
                in[c][0] = t4.x, in[c][1] = t4.y, in[c][2] = t4.z, in[c][3] = t4.w;
                if constexpr (PRO == PRO_SILUMUL) {
                    const uint4 u4 = ld16(x + G.K + e0);
                    const u32 uw[4] = {u4.x, u4.y, u4.z, u4.w};
#pragma unroll
                    for (int i = 0; i < 4; i++) in[c][i] = silu_mul_pk(in[c][i], uw[i]);
                }
            }
        }
#pragma unroll
        for (u32 k = 0; k < 4; k++) {
            uint4 o;
            o.x = perm(in[2][k], in[3][k], 0x05040100u);  // (x(c3, 2k), x(c2, 2k))
            o.y = perm(in[0][k], in[1][k], 0x05040100u);  // (x(c1, 2k), x(c0, 2k))
            o.z = perm(in[2][k], in[3][k], 0x07060302u);  // (x(c3, 2k+1), x(c2, 2k+1))
            o.w = perm(in[0][k], in[1][k], 0x07060302u);  // (x(c1, 2k+1), x(c0, 2k+1))
            *reinterpret_cast<uint4 *>(xlds + (((v * 4u + k) * G.Q + q) << 3)) = o;
        }
    }
}

struct DqArgs {
    ApArgs a;
    u32 RGB;      // 16-row groups per block
    u32 NU;       // units (512 weights) per row
    u32 part_off; // LDS offset of the partial sums
    unsigned long long *dbg;  // GQ_STAMPS builds: the debug buffer of gq_debug_set_timing_buffer
};
#ifndef DQ_WAVES_N
#define DQ_WAVES_N 16
#endif
constexpr u32 DQ_WAVES = DQ_WAVES_N;
// (also measured slower, round 6: two or three items' plane words in flight per wave with ONE copy of the decode -- slots moved into work
// registers behind explicit s_waitcnt vmcnt(n), re-requested 2 / 3 items ahead: w1w3 3-bit 18.0 / 19.6 vs 15.7 us, 4-bit 26.3 / 40.9 vs 23.0.
// More plane words in flight do not help this kernel: issuing a load blocks the wave once the CU's memory pipe is full.)
// ring depth 1: the next item's plane words are requested while this one is decoded.  Deeper rings measured SLOWER (w1w3 3-bit 15.5 /
// 17.4 / 19.1 us, 4-bit 22.9 / 25.2 / 27.7 us at depth 1 / 2 / 3, profiles/r06_dq_kernel.txt): the unrolled decode of one item is
// ~600 instructions (5 KB) already, and every slot of a ring is another copy of it in the instruction cache
#ifndef DQ_RING
#define DQ_RING 1
#endif
#ifndef DQ_KO
#define DQ_KO 0  // build-time knock-outs, WRONG numerics, timing only (bit mask): 1 no B reads, 2 no MFMAs, 4 no plane loads
#endif
// occupancy the instances are compiled for (waves per SIMD): the decode is a dependent chain (transpose -> selectors -> look-ups ->
// MFMA), so more resident waves = fewer idle issue slots
// (measured: ONE 16-wave block per CU is the fastest form -- 8-wave blocks at 3 / 4 per CU, <= 64 / 80 VGPRs: w1w3 3-bit 17.6 vs 16.3 us,
// 4-bit 26.9 vs 22.9, and their RMSNorm instances spill; profiles/r06_dq_kernel.txt)
#ifndef DQ_WPE3
#define DQ_WPE3 4
#endif
#ifndef DQ_WPE4
#define DQ_WPE4 4
#endif
template <int BITS>
constexpr int dq_wpe() { return BITS <= 3 ? DQ_WPE3 : DQ_WPE4; }

// QT: quads per row compiled in (32: K = 4096, 112: K = 14336 -- the 8B widths; 0: read from the launch): the sixteen B-fragment
// addresses of an item are then immediate offsets of ONE base register instead of sixteen v_add (7 % of the 3-bit item's VALU)
template <int BITS, int PRO, int QT>
__global__ void __launch_bounds__(64 * DQ_WAVES) __attribute__((amdgpu_waves_per_eu(dq_wpe<BITS>()))) ap_gemv_dq_kernel(DqArgs da) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const ApArgs &a = da.a;
    RowGeom G;
    G.init(a.K);
    uint16_t *xlds = reinterpret_cast<uint16_t *>(smem);
    float *part = reinterpret_cast<float *>(smem + da.part_off);
    float *red = part;  // (RMSNorm reduction scratch: before the partial sums exist)
    const u32 tid = threadIdx.x, l = tid & 63u, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const u32 r = l & 15u, g = l >> 4;
    const u32 rg0 = blockIdx.x * da.RGB, NU = da.NU;
    const u32 RGt = (a.N + 15u) >> 4;
    const u32 rgb = min(da.RGB, RGt - rg0);  // row groups of this block
    const u32 nitems = rgb * NU;
    constexpr int NRAW = (1 << BITS) / 2;
    constexpr u32 OOB = 0x80000000u;
    const u32 plane_bytes = a.N * G.wpr * 4u;
    __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void *)a.qw, 0, (int)(plane_bytes * (u32)BITS), 0x00020000);
    __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc((void *)a.lut, 0, (int)(a.N * (u32)NRAW * 4u), 0x00020000);
    constexpr int D = DQ_RING;
    u32x4 P[D][BITS];
    u32 lraw[D][NRAW];
    // (item it = (row group rgl, unit u) = (it / NU, it % NU): the loop below carries the pair along -- an integer division is ~30
    // instructions on this chip, and NU = 28 for the 14336-wide rows)
    auto issue = [&](int d, u32 it, u32 rgl, u32 u) {
        const u32 row = (rg0 + rgl) * 16u + r;
        const bool ok = it < nitems && row < a.N && !(DQ_KO & 4);  // (knock-out 4: no plane loads)
        const u32 off = ok ? (row * G.wpr + 4u * (4u * u + g)) * 4u : OOB;
#pragma unroll
        for (int p = 0; p < BITS; p++) P[d][p] = __builtin_amdgcn_raw_buffer_load_b128(rq, ok ? off + (u32)p * plane_bytes : OOB, 0, 2 /* nt */);
        const u32 loff = ok ? row * (u32)NRAW * 4u : OOB;
        if constexpr (NRAW == 2) {
            auto v = __builtin_amdgcn_raw_buffer_load_b64(rl, loff, 0, 0);
            lraw[d][0] = v[0];
            lraw[d][1] = v[1];
        } else {
#pragma unroll
            for (int i = 0; i < NRAW; i += 4) {
                u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rl, ok ? loff + (u32)i * 4u : OOB, 0, 0);
                lraw[d][i] = v.x;
                lraw[d][i + 1] = v.y;
                lraw[d][i + 2] = v.z;
                lraw[d][i + 3] = v.w;
            }
        }
    };
    auto first_requests = [&]() {
#pragma unroll
        for (int d = 0; d < D; d++) {
            const u32 it = w + (u32)d * DQ_WAVES;
            issue(d, it, it / NU, it % NU);
        }
    };
    // the waves that stage x (the lower half) request their first items behind the activations, the others at once (the CU's
    // memory pipe serves requests in order)
    const bool stager = DQ_WAVES < 2u || w < DQ_WAVES / 2u;
    if (!stager) first_requests();
    stage_x_dqv<PRO>(G, a.x, a.normw, a.eps, xlds, red, stager, (DQ_WAVES < 2u ? 1u : DQ_WAVES / 2u) * 64u);
    if (stager) first_requests();
    __syncthreads();
    // B fragments: column 0 reads the image, the other columns read zeros from beyond the block's LDS allocation (192 KiB up)
    const u32 Q = QT ? (u32)QT : G.Q;
    auto decode_item = [&](u32 it, u32 u, const u32 (&Pw)[BITS][4], const u32 (&lw)[NRAW]) {
        LutPools<BITS> L;
        L.build(lw);
        const u32 q = 4u * u + g;
        const unsigned char *bb = r == 0u ? reinterpret_cast<const unsigned char *>(xlds) + (size_t)q * 16u
                                          : reinterpret_cast<const unsigned char *>(smem) + 0x30000u;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        DqItem<BITS>::run(Pw, L, [&](int v, int jj, u32 a01, u32 a23, u32 b01, u32 b23) {
            const u32x4 av = {a01, a23, b01, b23};
#if DQ_KO & 1  // (timing knock-out, WRONG numerics: no B reads)
            const u32x4 bvv = {(u32)v, (u32)jj, q, Q};
#else
            const uint4 bv = *reinterpret_cast<const uint4 *>(bb + (size_t)((u32)(v * 4 + jj) * Q) * 16u);
            const u32x4 bvv = {bv.x, bv.y, bv.z, bv.w};
#endif
#if DQ_KO & 2  // (no MFMAs)
            acc0[0] += __builtin_bit_cast(float, av[0] ^ av[1] ^ av[2] ^ av[3] ^ bvv[0]);
#else
            if (jj & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, av), __builtin_bit_cast(f16x8, bvv), acc1, 0, 0, 0);
            else acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, av), __builtin_bit_cast(f16x8, bvv), acc0, 0, 0, 0);
#endif
        });
        // D[row 4 g + e][column r]: column 0 holds the sums
        if (r == 0u) *reinterpret_cast<f32x4 *>(part + (size_t)it * 16u + 4u * g) = acc0 + acc1;
    };
    // (GQ_STAMPS builds: s_memrealtime per wave of the middle block -- kernel start, then per item {top, plane words in hand, next item
    // requested, decoded}: tools/r6/dq_stamps.py)
    unsigned long long *dqdbg = (GQ_STAMPS && da.dbg && blockIdx.x == gridDim.x / 2u && l == 0u) ? da.dbg + (size_t)w * 32u : nullptr;
    u32 nst = 0;
    auto dstamp = [&]() {
        if (GQ_STAMPS && dqdbg && nst < 32u) dqdbg[nst++] = __builtin_amdgcn_s_memrealtime();
    };
    dstamp();
    static_assert(D == 1, "the (row group, unit) pair of an item is carried for ring depth 1");
    const u32 step_rg = DQ_WAVES / NU, step_u = DQ_WAVES - step_rg * NU;  // one division per launch
    u32 c_rg = w / NU, c_u = w - c_rg * NU;                                 // (row group, unit) of the item being decoded
    u32 n_rg = c_rg + step_rg, n_u = c_u + step_u;                           // .. and of the one being requested
    if (n_u >= NU) n_u -= NU, n_rg++;
    for (u32 it0 = w; it0 < nitems; it0 += (u32)D * DQ_WAVES) {
#pragma unroll
        for (int d = 0; d < D; d++) {
            const u32 it = it0 + (u32)d * DQ_WAVES;
            dstamp();
            u32 Pw[BITS][4];
#pragma unroll
            for (int p = 0; p < BITS; p++) {
                Pw[p][0] = P[d][p].x;
                Pw[p][1] = P[d][p].y;
                Pw[p][2] = P[d][p].z;
                Pw[p][3] = P[d][p].w;
            }
            u32 lw[NRAW];
#pragma unroll
            for (int i = 0; i < NRAW; i++) lw[i] = lraw[d][i];
            if (GQ_STAMPS) {
#pragma unroll
                for (int p = 0; p < BITS; p++) asm volatile("" : "+v"(Pw[p][0]), "+v"(Pw[p][3]));  // (the wait for the plane words sits here)
                dstamp();
            }
            issue(d, it + (u32)D * DQ_WAVES, n_rg, n_u);
            if (GQ_STAMPS) dstamp();
            const u32 u_now = c_u;
            c_rg = n_rg, c_u = n_u;
            n_rg += step_rg, n_u += step_u;
            if (n_u >= NU) n_u -= NU, n_rg++;
            if (it >= nitems) continue;  // (wave-uniform)
            decode_item(it, u_now, Pw, lw);
            if (GQ_STAMPS) dstamp();
        }
    }
    __syncthreads();
    // ordered K-split sum + epilogue: one thread per row of the block.  The residual is requested BEFORE the sums are read, and the
    // partial sums of eight units come in one LDS round trip (added in unit order as before): with one dependent ds_read per unit and the
    // residual behind the sum, the tail of a w2 launch (28 units) was 28 LDS round trips + one memory round trip long.
    const bool pairs = (a.epilogue & GQ_EPI_SILU_PAIRS) != 0u;
    for (u32 rl = tid; rl < rgb * 16u; rl += 64u * DQ_WAVES) {
        const u32 row = rg0 * 16u + rl;
        uint16_t res = 0;
        if (!pairs && a.resid && row < a.N) res = a.resid[row];
        const float *pp = part + (size_t)(rl >> 4) * NU * 16u + (rl & 15u);
        float y = 0.f;
        u32 u = 0;
        for (; u + 8u <= NU; u += 8u) {
            float v[8];
#pragma unroll
            for (u32 j = 0; j < 8u; j++) v[j] = pp[(size_t)(u + j) * 16u];
#pragma unroll
            for (u32 j = 0; j < 8u; j++) y += v[j];
        }
        for (; u < NU; u++) y += pp[(size_t)u * 16u];
        _Float16 yh = (_Float16)y;
        if (pairs) {
            // rows (2 i, 2 i + 1) = (gate_i, up_i): F.silu(gate) * up on fp16 values (inference/model.py:266), written to out[i]
            const uint16_t yo = (uint16_t)__builtin_amdgcn_update_dpp(0, (int)__builtin_bit_cast(uint16_t, yh), 0xB1, 0xF, 0xF, true);  // lane ^ 1
            if (!(rl & 1u) && row + 1u < a.N) {
                const float gv = (float)yh;
                const _Float16 o = (_Float16)(gv / (1.0f + __expf(-gv))) * __builtin_bit_cast(_Float16, yo);
                gq_store_wt(a.out + (row >> 1), __builtin_bit_cast(uint16_t, o));
            }
        } else if (row < a.N) {
            if (a.resid) yh = __builtin_bit_cast(_Float16, res) + yh;
            gq_store_wt(a.out + row, __builtin_bit_cast(uint16_t, yh));
        }
    }
}

struct DqCfg {
    u32 grid, RGB, NU, part_off;
    size_t smem;
};
bool pick_dq_cfg(u32 N, u32 K, int bits, DqCfg &c) {
    if (bits < 2 || bits > 4 || K % 1024u || K > 32768u || N < 16u) return false;
    const u32 RGt = (N + 15u) / 16u, cus = (u32)gq_cu_count();
    c.NU = K / 512u;
    // blocks per CU the instance's occupancy allows (waves per SIMD x 4 SIMDs / waves per block)
    u32 bpc = (u32)gq_env_int("GQ_DQ_BPC", 0);
    if (!bpc) bpc = (u32)(bits <= 3 ? DQ_WPE3 : DQ_WPE4) * 4u / DQ_WAVES;
    if (bpc < 1u) bpc = 1u;
    c.RGB = (RGt + cus * bpc - 1u) / (cus * bpc);
    c.grid = (RGt + c.RGB - 1u) / c.RGB;
    c.part_off = (K * 2u + 255u) / 256u * 256u;
    c.smem = (size_t)c.part_off + (size_t)c.RGB * c.NU * 64u + 64u;
    return c.smem <= 150u * 1024u;
}
template <int BITS, int PRO, int QT>
int launch_dq_q(const DqArgs &da, const DqCfg &c, hipStream_t s) {
    static GqPerDeviceOnce once;
    auto kern = ap_gemv_dq_kernel<BITS, PRO, QT>;
    GQ_HIP_CHECK(once.max_dynamic_lds(reinterpret_cast<const void *>(kern), (int)(160u * 1024u)));
    hipLaunchKernelGGL(kern, dim3(c.grid), dim3(64u * DQ_WAVES), c.smem, s, da);
    GQ_HIP_CHECK(hipGetLastError());
    return GQ_OK;
}
template <int BITS, int PRO>
int launch_dq_inst(const DqArgs &da, const DqCfg &c, hipStream_t s) {
    switch (da.a.K) {
        case 4096u: return launch_dq_q<BITS, PRO, 32>(da, c, s);
        case 14336u: return launch_dq_q<BITS, PRO, 112>(da, c, s);
        default: return launch_dq_q<BITS, PRO, 0>(da, c, s);
    }
}
template <int BITS>
int launch_dq(const DqArgs &da, const DqCfg &c, int pro, hipStream_t s) {
    switch (pro) {
        case PRO_RMSNORM: return launch_dq_inst<BITS, PRO_RMSNORM>(da, c, s);
        case PRO_SILUMUL: return launch_dq_inst<BITS, PRO_SILUMUL>(da, c, s);
        default: return launch_dq_inst<BITS, PRO_NONE>(da, c, s);
    }
}
// GQ_ENOTSUP when the shape is not served (the caller goes on to the plane / exact kernels)
int dq_gemv_try(const ApArgs &a, u32 M, int bits, int pro, hipStream_t s) {
    DqCfg c;
    if (M != 1u || !pick_dq_cfg(a.N, a.K, bits, c)) return GQ_ENOTSUP;
    if ((uint64_t)bits * a.N * (a.K / 8u) >= 0x7FFFFFFFull) return GQ_ENOTSUP;
    if ((((uintptr_t)a.qw | (uintptr_t)a.x | (uintptr_t)a.normw | (uintptr_t)a.lut) & 15u) != 0) return GQ_ENOTSUP;
    if ((a.epilogue & GQ_EPI_SILU_PAIRS) && (a.N & 1u)) return GQ_ENOTSUP;
    DqArgs da{a, c.RGB, c.NU, c.part_off, GQ_STAMPS ? gq_debug_timing_buffer() : nullptr};
    switch (bits) {
        case 2: return launch_dq<2>(da, c, pro, s);
        case 3: return launch_dq<3>(da, c, pro, s);
        default: return launch_dq<4>(da, c, pro, s);
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
#ifndef GQ_AP_PT_DEFAULT
#define GQ_AP_PT_DEFAULT 0  // exact mode, 2 bits: the LDS pair-table kernel (ap_gemv_pt2_kernel) instead of the v_perm kernel: 1 long rows, 2 all
#endif
#ifndef GQ_DQ_DEFAULT
#define GQ_DQ_DEFAULT 6  // (bit mask over the bit widths 2, 3, 4 the decode-to-fp16 kernel may take: 3 and 4)
#endif
struct QuadCfg {
    u32 T, RS, SPB, D, grid;
    size_t smem;
};

int num_cus() { return gq_cu_count(); }

// `pt2`: the configuration is for the 2-bit pair-table kernel (4 waves per SIMD)
bool pick_quad_cfg(u32 N, u32 K, int bits, QuadCfg &c, int pro = PRO_NONE, bool pt2 = false) {
    if (K % 128u) return false;
    const u32 Q = K / 128u;
    if (Q > 512u) return false;
    // block size: multiple of 64 in [192, 512] wasting the fewest lanes; ties -> 256, then the listed order
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
    // rows of more than 8192 weights: the widest block with the same row slots per step -- the spare waves shorten the activation
    // staging and the ordered reduction (8B w2, K = 14336: 448 -> 512 threads, 10.9 -> 10.1 us per launch in exact mode)
    // (and rows of exactly 8192: 512 threads = 8 row slots instead of 4 -- Llama-3.3-70B's wqkv 12.4 vs 15.2 us, wo 8.7 vs 10.5, w1w3 41.7 vs 45.9)
    if (bestT && (Q == 64u || (Q > 64u && 512u / Q == bestT / Q))) bestT = 512u;
    const int envT = gq_env_int("GQ_AP_T", 0);
    if (envT >= 64 && envT <= 512 && envT % 64 == 0 && (u32)envT >= Q) bestT = (u32)envT;
    if (!bestT) return false;
    c.T = bestT;
    c.RS = bestT / Q;
    const u32 steps = (N + c.RS - 1) / c.RS;
    // ring depth D (steps of plane words + LUT row in flight per lane).  Register budget (tests/test_kernel_resources_cpu.py checks that
    // no instance the dispatcher can pick spills): D = 1 fits 3 waves per SIMD at 2 / 3 bits (<= 168 VGPRs); deeper rings and 4 bits are
    // compiled for 2 waves per SIMD (256 VGPRs), where 4 bits fits D <= 3
    int d = gq_env_int("GQ_AP_D", 0);
    if (d < 1 || d > 4) d = bits <= 3 ? 1 : 2;
    if (bits == 4 && d > 3) d = 3;
    // persistent-style grid: about `bpc` blocks per CU (what the kernel's occupancy allows), every block the same number of steps
    const u32 cus = (u32)num_cus();
    u32 bpc_def = bits <= 3 && d == 1 ? 3u : 2u;
    // a block of an RMSNorm launch normalises the whole vector before its first step: with ONE step per block (few rows: wqkv) the
    // 2-3 resident blocks of a CU repeat that side by side -- one block per CU then (measured, 8B wqkv: 3-bit 8.8 -> 8.4 us, 4-bit
    // 11.9 -> 11.0; the large matrices keep the deeper grid: w1w3 exact 2-bit 17.2 vs 21.9 us with one block per CU)
    if (pro == PRO_RMSNORM && steps <= cus * bpc_def) bpc_def = 1u;
    // never more blocks per CU than the instance's occupancy holds at once (round 6, per-block start stamps: 8B w2 in exact mode, 512-thread
    // blocks of a 3-waves-per-SIMD kernel = ONE resident block per CU, was launched as 512 blocks -- two rounds of 5 us each;
    // profiles/r06_exact_epilogue.txt).  Waves per SIMD: the kernels' amdgpu_waves_per_eu attributes.
    const bool inreg = !pt2 && d == 1 && K == 4096u && c.T % 64u == 0u && c.RS * 32u == c.T;
    const u32 wpe = pt2 ? 4u : (inreg ? (bits == 2 ? 4u : 3u) : (bits <= 3 && d == 1 ? 3u : 2u));
    const u32 bpc_max = std::max(1u, wpe * 4u / ((c.T + 63u) / 64u));
    if (bpc_def > bpc_max) bpc_def = bpc_max;
    u32 bpc = (u32)gq_env_int("GQ_AP_BPC", (int)bpc_def);
    if (bpc < 1) bpc = 1;
    u32 target = cus * bpc;
    u32 spb = (steps + target - 1) / target;
    if (spb < 1) spb = 1;
    const u32 nchunks = K / 1024u + ((K % 1024u) ? 1u : 0u);
    // LDS: x image + one fp16 partial per (row, chunk, virtual lane)
    auto smem_for = [&](u32 s) { return (size_t)K * 2u + (size_t)s * c.RS * nchunks * 64u + 64u; };
    while (spb > 1 && smem_for(spb) > 150u * 1024u) spb--;
    if (smem_for(spb) > 160u * 1024u) return false;
    c.SPB = spb;
    c.grid = (steps + spb - 1) / spb;
    if ((u32)d > spb) d = (int)spb;
    c.D = (u32)d;
    c.smem = smem_for(spb);
    return true;
}

template <int BITS, int D, int PRO>
int launch_quad_inst(const ApArgs &a, const QuadCfg &c, u32 M, hipStream_t s) {
    dim3 grid(c.grid, M), block(c.T);
    ApArgs ad = a;
    if (GQ_STAMPS) ad.dbg = gq_debug_timing_buffer();
    // rows of 4096 weights at the shipped ring depth: the ordered reduction in registers (INREG above)
    if constexpr (D == 1) {
        if (a.K == 4096u && c.T % 64u == 0u && c.RS * 32u == c.T && gq_env_int("GQ_AP_INREG", 1) != 0) {
            static GqPerDeviceOnce once_r;
            auto kern_r = ap_gemv_quad_kernel<BITS, D, PRO, true>;
            GQ_HIP_CHECK(once_r.max_dynamic_lds(reinterpret_cast<const void *>(kern_r), (int)(160u * 1024u)));
            hipLaunchKernelGGL(kern_r, grid, block, c.smem, s, ad);
            GQ_HIP_CHECK(hipGetLastError());
            return GQ_OK;
        }
    }
    static GqPerDeviceOnce once;
    auto kern = ap_gemv_quad_kernel<BITS, D, PRO>;
    GQ_HIP_CHECK(once.max_dynamic_lds(reinterpret_cast<const void *>(kern), (int)(160u * 1024u)));
    hipLaunchKernelGGL(kern, grid, block, c.smem, s, ad);
    GQ_HIP_CHECK(hipGetLastError());
    return GQ_OK;
}

template <int BITS, int PRO>
int launch_quad_d(const ApArgs &a, const QuadCfg &c, u32 M, hipStream_t s) {
    switch (c.D) {
        case 1: return launch_quad_inst<BITS, 1, PRO>(a, c, M, s);
        case 2: return launch_quad_inst<BITS, 2, PRO>(a, c, M, s);
        case 3: return launch_quad_inst<BITS, 3, PRO>(a, c, M, s);
        default:
            if constexpr (BITS <= 3) return launch_quad_inst<BITS, 4, PRO>(a, c, M, s);
            else return launch_quad_inst<BITS, 3, PRO>(a, c, M, s);  // (pick_quad_cfg never asks for more at 4 bits: D = 4 would spill)
    }
}

template <int BITS>
int launch_quad(const ApArgs &a, const QuadCfg &c, u32 M, int pro, hipStream_t s) {
    switch (pro) {
        case PRO_RMSNORM: return launch_quad_d<BITS, PRO_RMSNORM>(a, c, M, s);
        case PRO_SILUMUL: return launch_quad_d<BITS, PRO_SILUMUL>(a, c, M, s);
        default: return launch_quad_d<BITS, PRO_NONE>(a, c, M, s);
    }
}

}  // namespace
extern "C" int gq_debug_exact_plan(uint32_t N, uint32_t K, int bits, int prologue, uint32_t *plan) {
    QuadCfg c;
    if (!plan || bits < 2 || bits > 4 || !pick_quad_cfg(N, K, bits, c, prologue)) return GQ_ENOTSUP;
    const bool inreg = c.D == 1u && K == 4096u && c.T % 64u == 0u && c.RS * 32u == c.T;
    const u32 wpe = inreg ? (bits == 2 ? 4u : 3u) : (bits <= 3 && c.D == 1u ? 3u : 2u);  // (the kernels' amdgpu_waves_per_eu attributes)
    plan[0] = c.T, plan[1] = c.RS, plan[2] = c.SPB, plan[3] = c.D, plan[4] = c.grid, plan[5] = std::max(1u, wpe * 4u / ((c.T + 63u) / 64u));
    return GQ_OK;
}
namespace {
// the 2-bit pair-table kernel on the quad kernel's configuration (+ the waves' tables behind the partial sums); GQ_ENOTSUP where a wave
// could span more than two rows (fewer than 64 quads per row unless exactly 32) or the tables do not fit
int launch_pt2(const ApArgs &a, const QuadCfg &c, u32 M, int pro, hipStream_t s) {
    const u32 Q = a.K / 128u;
    if (!(Q == 32u || Q >= 64u) || (c.T & 63u) || (Q == 32u && (c.T % 64u))) return GQ_ENOTSUP;
    const u32 nchunks = a.K / 1024u + ((a.K % 1024u) ? 1u : 0u);
    const size_t smem = (((size_t)a.K * 2u + (size_t)c.SPB * c.RS * nchunks * 64u + 63u) & ~(size_t)63u) + (size_t)(c.T / 64u) * 128u + 64u;
    if (smem > 160u * 1024u) return GQ_ENOTSUP;
    dim3 grid(c.grid, M), block(c.T);
#define GQ_PT2(PRO_)                                                                                                            \
    do {                                                                                                                        \
        static GqPerDeviceOnce once;                                                                                            \
        GQ_HIP_CHECK(once.max_dynamic_lds(reinterpret_cast<const void *>(ap_gemv_pt2_kernel<PRO_>), (int)(160u * 1024u)));        \
        hipLaunchKernelGGL(ap_gemv_pt2_kernel<PRO_>, grid, block, smem, s, a);                                                  \
    } while (0)
    switch (pro) {
        case PRO_RMSNORM: GQ_PT2(PRO_RMSNORM); break;
        case PRO_SILUMUL: GQ_PT2(PRO_SILUMUL); break;
        default: GQ_PT2(PRO_NONE); break;
    }
#undef GQ_PT2
    GQ_HIP_CHECK(hipGetLastError());
    return GQ_OK;
}

template <int BITS>
int launch_generic(const ApArgs &a, u32 M, hipStream_t s) {
    const int ksplit = (M == 1 && a.K > 4096 && BITS >= 7) ? 1 : 0;  // anyprec.cu:611
    dim3 grid((a.N + 7) / 8, M), block(256);
    hipLaunchKernelGGL(ap_gemv_generic_kernel<BITS>, grid, block, 0, s, a, ksplit);
    GQ_HIP_CHECK(hipGetLastError());
    return GQ_OK;
}

}  // namespace

int gq_plane_gemv_try(const void *x, void *out, const uint32_t *qweight, const void *lut, uint32_t M, uint32_t N, uint32_t K,
                      int bits, const void *normw, float eps, const void *resid, int pro, int pairs, hipStream_t stream, void *ws, size_t ws_bytes,
                      GqHandover *ho);
size_t gq_stream_ksplit_ws_bytes(uint32_t N, uint32_t K, int bits);  // ap_stream.hip
bool gq_plane_local_shape(uint32_t N, uint32_t K, int bits);

namespace {

int g_ap_mode = -1;  // -1: unset (env GQ_AP_MODE or default fast), 0: fast (plane-MFMA), 1: exact (fp16-order)
bool exact_mode() {
    if (g_ap_mode >= 0) return g_ap_mode == 1;
    return gq_env_int("GQ_AP_EXACT", 0) != 0;
}
}  // namespace
bool gq_ap_exact_mode() { return exact_mode(); }  // (ap_stream.hip: the fused q / k / v + RoPE launch is a fast-mode kernel)
namespace {

// Stand-alone producer of the hand-over statistics: GQ_SSQ_SLOTS partial sums of squares of an fp16 vector (slot t: elements t, t + 1024,
// ..).  What a GEMV launch without the in-epilogue form is followed by when the caller asked for ssq_out; also gq_ssq_rows.
__global__ void __launch_bounds__(GQ_SSQ_SLOTS) ssq_rows_kernel(const uint16_t *x, u32 n, float *ssq) {
    float acc = 0.f;
    for (u32 i = threadIdx.x; i < n; i += (u32)GQ_SSQ_SLOTS) {
        const float v = h2f(x[i]);
        acc += v * v;
    }
    gq_store_wt(ssq + threadIdx.x, acc);
}

int ap_gemv_dispatch_inner(ApArgs a, u32 M, int bits, hipStream_t s, GqHandover *ho);
int ap_gemv_dispatch(ApArgs a, u32 M, int bits, hipStream_t s) {
    GqHandover ho;
    ho.ssq_in = a.ssq_in;
    ho.ssq_out = a.ssq_out;
    if (ho.ssq_out && ((a.epilogue & GQ_EPI_SILU_PAIRS) || M != 1u)) return gq_fail(GQ_EINVAL, "ssq_out: plain / residual epilogue, M = 1 only.");
    const int rc = ap_gemv_dispatch_inner(a, M, bits, s, &ho);
    if (rc == GQ_OK && ho.ssq_out && !ho.ssq_written) {  // the kernel that served the shape has no in-epilogue form: one small launch more
        hipLaunchKernelGGL(ssq_rows_kernel, dim3(1), dim3(GQ_SSQ_SLOTS), 0, s, a.out, a.N, ho.ssq_out);
        GQ_HIP_CHECK(hipGetLastError());
    }
    return rc;
}

int ap_gemv_dispatch_inner(ApArgs a, u32 M, int bits, hipStream_t s, GqHandover *ho) {
    if (bits < 2 || bits > 8) return gq_fail(GQ_EINVAL, "Bitwidth must be between 2 and 8.");
    if (M < 1 || M > 8) return gq_fail(GQ_EINVAL, "batch size M must be between 1 and 8 (anyprec.cu:602).");
    if (a.K == 0 || a.K % 32u) return gq_fail(GQ_EINVAL, "input_feat (K) must be a positive multiple of 32.");
    if (a.N == 0) return gq_fail(GQ_EINVAL, "output_feat (N) must be positive.");
    if (!a.x || !a.out || !a.qw || !a.lut) return gq_fail(GQ_EINVAL, "null pointer argument.");
    const bool force_generic = gq_env_int("GQ_AP_FORCE_GENERIC", 0) != 0;
    QuadCfg c;
    const int pro = a.normw ? PRO_RMSNORM : ((a.epilogue & GQ_PRO_SILU_MUL) ? PRO_SILUMUL : PRO_NONE);
    // fast mode serves the shapes on which the plane-MFMA kernel beats the exact kernel (measured):
    // 2-bit matrices of >= 20 M weights (wqkv, w1w3, w2 of the 8B / 70B models), 3- and 4-bit matrices of >= 32 M weights
    // (w1w3, w2); everything else runs the exact kernels, whose results are bit-identical to the reference.
    // GQ_PL_MIN_MWEIGHTS overrides the threshold for every bit width, GQ_PL_MAX_BITS the widest plane-served width.
    const int env_min = gq_env_int("GQ_PL_MIN_MWEIGHTS", -1);
    // (matrices of <= 16 rows per CU without the RMSNorm prologue run the local-image variant, which needs no block-wide
    // activation pass: >= 16 M weights at 2 and 3 bits -- wo)
    const bool local = pro != PRO_RMSNORM && bits <= 3 && gq_plane_local_shape(a.N, a.K, bits);
    // (round 5: behind the RMSNorm prologue the plane kernel wins from 20 M weights at 3 and 4 bits too -- 8B wqkv, 25 M: 7.2 vs 8.4 us
    // at 3 bits, 10.2 vs 12.0 at 4 -- the exact kernel normalises the whole vector in every block in front of its first row step;
    // without the prologue the 32 M threshold stands: wo at 4 bits 6.6 exact vs 7.6 plane.  profiles/r05_dispatch_3_4_bits.txt)
    const int def_min = local ? 16 : ((bits == 2 || pro == PRO_RMSNORM) ? 20 : 32);
    const uint64_t min_w = (uint64_t)(env_min >= 0 ? env_min : def_min) * 1000000ull;
    const int max_bits = gq_env_int("GQ_PL_MAX_BITS", 4);
    // round 6: decode-to-fp16 on the matrix cores (ap_gemv_dq_kernel) where it measured faster than the other kernels
    // (profiles/r06_dq_kernel.txt, 8B shapes, decode-graph launch forms, same box): at 4 bits every matrix of >= 16 M weights -- wqkv 9.1
    // vs 10.3 us, wo 6.6 vs 6.8 (exact kernel), w1w3 22.9 vs 24.4, w2 13.9 vs 15.0 --, at 3 bits the matrices of 16 .. 32 M weights (wqkv
    // 6.5 vs 7.2, wo 4.9 vs 5.3; Llama-2-7B's wqkv 50 M 8.1 vs 9.5, w1w3 90 M 12.7 vs 14.2) up to 100 M, and the long-row launches
    // without the RMSNorm prologue at any size (8B w2 9.35 vs 9.65, 70B w2 235 M 26.5 vs 34.8; 70B wqkv 84 M 13.2 vs 14.8); the big
    // RMSNorm + pair launches stay on the plane kernel (70B w1w3 470 M 45.7 vs 43.0; 8B w1w3 117 M was 15.3 vs 14.8 and, with the item
    // loop's divisions gone, is 14.6 vs 14.8 alone and 13.5 vs 14.0 in the decode graph: 3-bit decode 720 -> 736 tokens/s -- the bound moved
    // from 100 M to 200 M); never at 2 bits
    // (8B w1w3 11.6 vs 8.5, 70B w1w3 37 vs 25).
    // GQ_DQ: bit mask of the widths it may take (bit b - 2; 0 = never), GQ_DQ_MIN_MWEIGHTS >= 0: every matrix of at least that many
    // million weights at those widths.
    {
        const int dq_mask = gq_env_int("GQ_DQ", GQ_DQ_DEFAULT), dq_min = gq_env_int("GQ_DQ_MIN_MWEIGHTS", -1);
        const uint64_t nk = (uint64_t)a.N * a.K;
        const bool dq_shape = dq_min >= 0 ? nk >= (uint64_t)dq_min * 1000000ull
                                          // (and at least one 16-row group per CU: Llama-3.2-1B's w2, 2048 x 8192 = 128 blocks, is faster on the
                                          // exact kernel -- 4-bit 1B decode 1555 vs 1508 tokens/s; Llama-3.3-70B at 4 bits: 72 -> 86 tokens/s here)
                                          : (a.N >= 4096u &&
                                             (bits == 4 ? nk >= 16000000ull
                                                        : (bits == 3 && nk >= 16000000ull &&
                                                           (nk < 200000000ull || (pro != PRO_RMSNORM && a.K >= 8192u)))));
        if (!force_generic && !exact_mode() && bits <= 4 && ((dq_mask >> (bits - 2)) & 1) && dq_shape && !(ho && ho->dry)) {
            const int rc = dq_gemv_try(a, M, bits, pro, s);
            if (rc != GQ_ENOTSUP) return rc;
        }
    }
    if (!force_generic && !exact_mode() && bits <= max_bits && (uint64_t)a.N * a.K >= min_w) {
        int rc = gq_plane_gemv_try(a.x, a.out, a.qw, a.lut, M, a.N, a.K, bits, a.normw, a.eps, a.resid, pro, (a.epilogue & GQ_EPI_SILU_PAIRS) != 0, s, a.ws, a.ws_bytes, ho);
        if (rc != GQ_ENOTSUP) return rc;
    }
    if (ho && ho->dry) return GQ_OK;  // (the exact-mode kernels have no hand-over form)
    const uint64_t qbytes = (uint64_t)bits * a.N * (a.K / 8u);
    if (!force_generic && bits <= 4 && qbytes < 0x7FFFFFFFull && pick_quad_cfg(a.N, a.K, bits, c, pro) &&
        (((uintptr_t)a.qw | (uintptr_t)a.x | (uintptr_t)a.normw) & 15u) == 0 && ((uintptr_t)a.lut & 15u) == 0 &&
        !((a.epilogue & GQ_EPI_SILU_PAIRS) && ((c.SPB * c.RS) & 1u))) {
        a.RS = c.RS;
        a.SPB = c.SPB;
        // round 6: the pair-table kernel (GQ_AP_PT: 0 never -- the default --, 1 rows of >= 8192 weights, 2 every shape it serves).  It had
        // measured faster on 8B w2 (11.3 vs 12.7 us) while BOTH kernels ran that launch in two rounds of blocks (pick_quad_cfg: blocks per
        // CU beyond the occupancy); in one round the v_perm kernel is the faster one on every Llama shape (8B w2 8.9 vs 9.9 us; 70B w2 26.0
        // vs 33.6, wqkv 12.4 vs 13.5, wo 8.7 vs 10.0) except 70B w1w3 (41.7 vs 38.1): profiles/r06_exact_pair_table.txt, r06_exact_epilogue.txt
        {
            const int pt = gq_env_int("GQ_AP_PT", GQ_AP_PT_DEFAULT);
            QuadCfg cp;
            if (bits == 2 && (pt >= 2 || (pt == 1 && a.K >= 8192u)) && pick_quad_cfg(a.N, a.K, bits, cp, pro, true) &&
                !((a.epilogue & GQ_EPI_SILU_PAIRS) && ((cp.SPB * cp.RS) & 1u))) {
                ApArgs ap = a;
                ap.RS = cp.RS;
                ap.SPB = cp.SPB;
                const int rc = launch_pt2(ap, cp, M, pro, s);
                if (rc != GQ_ENOTSUP) return rc;
            }
        }
        switch (bits) {
            case 2: return launch_quad<2>(a, c, M, pro, s);
            case 3: return launch_quad<3>(a, c, M, pro, s);
            default: return launch_quad<4>(a, c, M, pro, s);
        }
    }
    if (pro != PRO_NONE || (a.epilogue & GQ_EPI_SILU_PAIRS))
        return gq_fail(GQ_ENOTSUP, "fused prologue / pair epilogue needs bits in 2..4, K % 128 == 0 and 16-byte aligned buffers.");
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

extern "C" int gq_set_ap_mode(int mode) {
    if (mode < -1 || mode > 1) return gq_fail(GQ_EINVAL, "mode must be -1 (default), 0 (fast) or 1 (exact)");
    g_ap_mode = mode;
    return GQ_OK;
}

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

extern "C" size_t gq_anyprec_gemv_fused_ws_bytes(uint32_t N, uint32_t K, int bits, uint32_t epilogue) {
    if ((epilogue & (GQ_PRO_SILU_MUL | GQ_EPI_SILU_PAIRS)) || exact_mode()) return 0;
    return gq_stream_ksplit_ws_bytes(N, K, bits);
}
extern "C" int gq_anyprec_gemv_fused(const void *x, void *out, const uint32_t *qweight, const void *lut, uint32_t N,
                                     uint32_t K, int bits, const void *norm_weight, float eps, const void *residual,
                                     uint32_t epilogue, void *stream) {
    return gq_anyprec_gemv_fused_ws(x, out, qweight, lut, N, K, bits, norm_weight, eps, residual, epilogue, nullptr, 0, stream);
}
extern "C" int gq_anyprec_gemv_fused_ws(const void *x, void *out, const uint32_t *qweight, const void *lut, uint32_t N,
                                        uint32_t K, int bits, const void *norm_weight, float eps, const void *residual,
                                        uint32_t epilogue, void *workspace, size_t workspace_bytes, void *stream) {
    return gq_anyprec_gemv_fused_ho(x, out, qweight, lut, N, K, bits, norm_weight, eps, residual, epilogue, workspace, workspace_bytes, nullptr,
                                    nullptr, stream);
}
extern "C" int gq_anyprec_handover_plan(uint32_t N, uint32_t K, int bits, int has_norm, uint32_t epilogue) {
    // which kernel would serve the launch, without launching: 1 = its RMSNorm prologue reads ssq_in, 2 = its epilogue writes ssq_out
    ApArgs a{};
    alignas(16) static const unsigned char al[16] = {0};  // (a 16-byte aligned stand-in for every pointer: never dereferenced)
    a.qw = (const u32 *)al;
    a.lut = (const uint16_t *)al;
    a.x = (const uint16_t *)al;
    a.out = (uint16_t *)al;
    a.normw = has_norm ? (const uint16_t *)al : nullptr;
    a.resid = (epilogue & GQ_EPI_RESIDUAL) ? (const uint16_t *)al : nullptr;
    a.N = N;
    a.K = K;
    a.epilogue = epilogue;
    GqHandover ho;
    ho.dry = true;
    ho.ssq_in = (const float *)al;
    ho.ssq_out = (epilogue & GQ_EPI_SILU_PAIRS) ? nullptr : (float *)al;
    if (ap_gemv_dispatch_inner(a, 1, bits, nullptr, &ho) != GQ_OK) return 0;
    return (ho.ssq_consumed ? 1 : 0) | (ho.ssq_written ? 2 : 0);
}
extern "C" int gq_ssq_rows(const void *x, uint32_t n, float *ssq_out, void *stream) {
    if (!x || !ssq_out || n == 0) return gq_fail(GQ_EINVAL, "null pointer argument / empty vector.");
    hipLaunchKernelGGL(ssq_rows_kernel, dim3(1), dim3(GQ_SSQ_SLOTS), 0, (hipStream_t)stream, (const uint16_t *)x, n, ssq_out);
    GQ_HIP_CHECK(hipGetLastError());
    return GQ_OK;
}
extern "C" int gq_anyprec_gemv_fused_ho(const void *x, void *out, const uint32_t *qweight, const void *lut, uint32_t N,
                                        uint32_t K, int bits, const void *norm_weight, float eps, const void *residual,
                                        uint32_t epilogue, void *workspace, size_t workspace_bytes, const float *ssq_in, float *ssq_out,
                                        void *stream) {
    ApArgs a{};
    a.ssq_in = norm_weight ? ssq_in : nullptr;
    a.ssq_out = ssq_out;
    a.ws = workspace;
    a.ws_bytes = workspace ? workspace_bytes : 0;
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
    if ((epilogue & GQ_PRO_SILU_MUL) && norm_weight) return gq_fail(GQ_EINVAL, "RMSNorm and SiLU-mul prologues are exclusive.");
    if ((epilogue & GQ_EPI_SILU_PAIRS) && ((epilogue & GQ_EPI_RESIDUAL) || (N & 1u)))
        return gq_fail(GQ_EINVAL, "SILU_PAIRS epilogue needs an even N and excludes the residual epilogue.");
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
