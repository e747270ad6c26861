// ap_stream.hip -- "stream" form of the plane-MFMA Any-Precision GEMV (fast mode, one batch row) for gfx950.
//
// Same arithmetic as ap_plane.hip (plane_core.h: a b-bit LUT is multilinear in the code bits, so y is 2^b - 1 BINARY GEMVs over
// the stored bit-planes; FP4 single-bit A operands built with one v_and per 8 weights, the activations as 4 exact bf8 pieces in
// 4 MFMA columns, fp32 accumulation) -- replaces anyprec.cu:372-542 for M = 1.  What is different is where the bytes go
// (round 4; measured reasons in DESIGN.md section 3.5):
//   * the plane words travel HBM -> VGPR (buffer_load_dwordx4, 16 rows x 64 B per instruction) and are masked in place: no
//     LDS ring, no direct-to-LDS loads, no ds_read of A tiles, no hand-counted vmcnt;
//   * a wave multiplies ONE K range ("unit": half a 1024-weight chunk, 512 activations) of several row groups and keeps the
//     activation image of that range in 32 VGPRs: no B re-reads in the loop (the round-3 kernel issued 20 ds_read_b128 per
//     24 MFMAs and its LDS pipe was 50 % busy); the K-split partial sums of a row group are added in the epilogue;
//   * every unit has its own power-of-two scale (the E8M0 scale operand of its MFMAs undoes it, partial sums of different
//     units meet in fp32 only): no whole-vector maximum, the only whole-vector statistic left is the RMSNorm sum of squares;
//   * image hand-over per unit through LDS flags: no block-wide barrier between launch and the first MFMA except the one
//     that publishes the zeroed flags (placed behind the activation loads).
// Numerics: products exact, fp32 accumulation, one fp16 rounding -- the fast-mode envelope of tests/ap_helpers.py; not
// bit-identical to ap_plane.hip (other summation order).  Elements above 64 x the unit's mean magnitude leave the image and are
// multiplied on their own (see ap_plane.hip "hot" elements; here per unit: fewer than 512 / 64 = 8 of them by Markov).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "gq_internal.h"
#include "plane_core.h"

using namespace gqp;

namespace {

typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef u32 u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
typedef uint16_t us2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) u32 lds_u32;

struct StreamArgs {
    const u32 *qw;
    const uint16_t *lut;
    const uint16_t *x;
    uint16_t *out;
    const uint16_t *normw;
    const uint16_t *resid;
    u32 N, K;
    u32 wpr_ld;  // plane words per stored row
    u32 RGB;     // row groups (16 rows) per block
    u32 lq, lw;  // log2 of the consumer units per row (31: not fewer than waves) and of the waves per block
    u32 img_off; // LDS offset of the images (behind meta, counters and the coefficient table)
    u32 dbg_off; // LDS offset of the phase stamps (only with a timing buffer)
    u32 pairs;   // GQ_EPI_SILU_PAIRS
    float eps;
    unsigned long long *dbg;  // per-wave phase timestamps of the middle block (tools/phase_timing.py)
    // RoPE epilogue of the fused q / k / v projection (gq_anyprec_gemv_qkv_rope): see the epilogue
    const int *pos;
    const uint16_t *cos_t, *sin_t;
    uint16_t *kc, *vc;
    u32 rope, H, Hkv, lhd, max_seq;  // lhd = log2(head_dim)
    // K split over blocks (gq_stream_gemv_ksplit: rows wider than 16384, the 70B down projection): blockIdx.y = K slice of a.K
    // activations out of Kx; the block leaves the fp32 sums of its slice in part_out[slice][N], ap_ksplit_reduce_kernel adds them
    float *part_out;
    u32 Kx;  // activations per row of x / per stored row (a.K = the slice this block multiplies); 0: a.K
    // statistics hand-over (include/gq_hip.h, GQ_SSQ_SLOTS): the partial sums of squares of x, left by the launch that wrote x
    const float *ssq_in;
};

enum { PRO_NONE = 0, PRO_RMSNORM = 1, PRO_SILUMUL = 2 };
constexpr u32 HOTCAP = 8u;  // Markov: |x| > 64 mean|x| holds for fewer than 512 / 64 elements of a unit
constexpr u32 OOB = 0x80000000u;

__device__ __forceinline__ float h2f(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
__device__ __forceinline__ h2v u2h2(u32 u) { return __builtin_bit_cast(h2v, u); }
__device__ __forceinline__ u32 h22u(h2v h) { return __builtin_bit_cast(u32, h); }

// wave64 reductions on the DPP path: result valid in lane 63
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
    {   // row_bcast15 into rows 1 and 3, row_bcast31 into rows 2 and 3 (|x| >= 0 and the sum identity are both 0.0f)
        float o = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, false));
        v = MAX ? fmaxf(v, o) : v + o;
        o = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xC, 0xF, false));
        v = MAX ? fmaxf(v, o) : v + o;
    }
    return v;
}
__device__ __forceinline__ float lane63(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63)); }

// A = FP4 (4 registers; the upper half of the builtin's vector is ignored for cbsz = 4), B = BF8.  Through the builtin: the
// compiler must see the instruction to insert the MFMA hazard wait states.
__device__ __forceinline__ void mfma_f4_bf8(v4f &acc, v4i a, v8i b, int scale_a, int scale_b) {
    const v8i a8 = __builtin_shufflevector(a, a, 0, 1, 2, 3, -1, -1, -1, -1);
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b, acc, 4, 1, 0, scale_a, 0, scale_b);
}

// F.silu(gate) * up on two packed fp16 pairs -- inference/model.py:266
__device__ __forceinline__ u32 silu_mul2(u32 gw, u32 uw) {
    _Float16 hh[2];
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const float gv = h2f((uint16_t)(gw >> (16 * k)));
        hh[k] = (_Float16)(gv / (1.0f + __expf(-gv))) * __builtin_bit_cast(_Float16, (uint16_t)(uw >> (16 * k)));
    }
    return (u32)__builtin_bit_cast(uint16_t, hh[0]) | ((u32)__builtin_bit_cast(uint16_t, hh[1]) << 16);
}

struct HotEnt {
    u32 key, pieces;  // key = b << 7 | k (logical k of the unit's (b) block), pieces: byte p = bf8 piece p of x * 2^ksh
};
#define GQ_ST_HOT_T 64.0f

// LDS flags / counters through explicit LDS pointers (a generic pointer becomes a FLAT access whose vmcnt(0) drains the
// wave's plane loads) and without fences: the LDS serves one wave's operations in order, so a flag written behind the data
// is seen behind the data; the compiler is held by the asm memory clobbers.
__device__ __forceinline__ void lds_store(u32 *p, u32 v) {
    asm volatile("" ::: "memory");
    *(volatile lds_u32 *)(lds_u32 *)p = v;
}
__device__ __forceinline__ u32 lds_load(const u32 *p) {
    const u32 v = *(volatile lds_u32 *)(lds_u32 *)p;
    asm volatile("" ::: "memory");
    return v;
}
__device__ __forceinline__ void lds_wait_ge(const u32 *p, u32 target) {
    while (lds_load(p) < target) __builtin_amdgcn_s_sleep(1);
}

typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void *p, u32 bytes) {  // raw buffer: stride 0, num_records = bytes
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), (short)0, (int)bytes, 0x00020000);
}
#ifndef ST_AUX
#define ST_AUX 2  // cache policy of the plane loads: 2 = nt (streamed once)
#endif
template <int AUX>
__device__ __forceinline__ u32x4 bload128(rsrc_t rsrc, u32 voff, u32 soff) {
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, (int)soff, AUX));
}

// The plane words are requested from inline asm and waited for with explicit s_waitcnt vmcnt(n) (vector memory returns in
// order): the compiler's own counting gives up at the loop's control-flow joins and drains the queue (vmcnt(0)) before every
// unit, which serialises the memory latency with the MFMAs.  The destination is an in-out operand of the request (an output
// of its own may be copied by the compiler before the data lands) and is tied again behind the wait.
__device__ __forceinline__ u32x4 make_rsrc4(const void *p, u32 bytes) {
    const uint64_t a = (uint64_t)(uintptr_t)p;
    return (u32x4){(u32)a, (u32)(a >> 32) & 0xFFFFu, bytes, 0x00020000u};
}
__device__ __forceinline__ void aload128(u32x4 &dst, u32x4 rsrc, u32 voff, u32 soff) {
#if ST_AUX == 2
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen nt" : "+v"(dst) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
#else
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "+v"(dst) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
#endif
}
template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// wait until at most n * LPU plane loads of this wave are outstanding (n wave-uniform)
template <int LPU>
__device__ __forceinline__ void wait_vm_units(u32 n) {
    switch (n) {
        case 0: wait_vm<0>(); break;
        case 1: wait_vm<LPU>(); break;
        case 2: wait_vm<2 * LPU>(); break;
        default: wait_vm<3 * LPU>(); break;
    }
}

// The MFMAs of one (unit half hh, nibble bit NB) block: FP4 operands of the plane subsets built on the fly (the mask commutes
// with AND; depth-first over the subset lattice).  Plane p holds code bit BITS-1-p; subset index cm = OR of the code bits.
template <int BITS, int NB, bool FIRST = false>
__device__ __forceinline__ void mfma_b(v4f (&acc)[(1 << BITS) - 1], const u32x4 (&A)[BITS], v8i Bv, int sb) {
    // FIRST: every accumulator is written by this block (each subset has one MFMA per nibble bit) -- SrcC is the inline constant 0
    // instead of a register cleared with 4 v_mov per subset and unit
    if constexpr (FIRST) {
#pragma unroll
        for (int i = 0; i < (1 << BITS) - 1; i++) acc[i] = (v4f){0.f, 0.f, 0.f, 0.f};
    }
    v4i Mp[BITS];
#pragma unroll
    for (int p = 0; p < BITS; p++)
#pragma unroll
        for (int v = 0; v < 4; v++) Mp[p][v] = (int)extract4(A[p][v], NB);
    const int sa = scale_byte4(NB);
#pragma unroll
    for (int p0 = 0; p0 < BITS; p0++) {
        const v4i A1 = Mp[p0];
        const int c1 = 1 << (BITS - 1 - p0);
        mfma_f4_bf8(acc[c1 - 1], A1, Bv, sa, sb);
#pragma unroll
        for (int p1 = p0 + 1; p1 < BITS; p1++) {
            const v4i A2 = A1 & Mp[p1];
            const int c2 = c1 | (1 << (BITS - 1 - p1));
            mfma_f4_bf8(acc[c2 - 1], A2, Bv, sa, sb);
#pragma unroll
            for (int p2 = p1 + 1; p2 < BITS; p2++) {
                const v4i A3 = A2 & Mp[p2];
                const int c3 = c2 | (1 << (BITS - 1 - p2));
                mfma_f4_bf8(acc[c3 - 1], A3, Bv, sa, sb);
#pragma unroll
                for (int p3 = p2 + 1; p3 < BITS; p3++) {
                    const v4i A4 = A3 & Mp[p3];
                    const int c4 = c3 | (1 << (BITS - 1 - p3));
                    mfma_f4_bf8(acc[c4 - 1], A4, Bv, sa, sb);
                }
            }
        }
    }
}

// the extracted elements of one unit half: per nibble bit b with entries one more MFMA set whose B operand is zero except for
// the listed bytes (exact products, nothing to align against); deterministic whatever the order of the list
template <int BITS>
__device__ __forceinline__ void hot_unit(v4f (&acc)[(1 << BITS) - 1], const u32x4 (&A)[BITS], const HotEnt *list, u32 nhot, int sb, u32 col,
                                         u32 kb) {
    u32 mask = 0;
    for (u32 e = 0; e < nhot; e++) mask |= 1u << ((__builtin_amdgcn_readfirstlane(list[e].key) >> 7) & 3u);
    while (mask) {
        const u32 b = (u32)__builtin_ctz(mask);
        mask &= mask - 1u;
        u32 B[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
        for (u32 e = 0; e < nhot; e++) {
            const u32 key = __builtin_amdgcn_readfirstlane(list[e].key);
            if (((key >> 7) & 3u) != b) continue;
            const u32 pieces = __builtin_amdgcn_readfirstlane(list[e].pieces);
            // lane (col = piece, kb) holds k = 64 (j / 16) + 16 kb + j % 16 as byte j of its 8 registers
            const u32 k = key & 127u, j = ((k >> 6) << 4) | (k & 15u);
            const bool mine = col < 4u && ((k >> 4) & 3u) == kb;
            const u32 val = mine ? ((pieces >> (8u * col)) & 0xFFu) << (8u * (j & 3u)) : 0u;
#pragma unroll
            for (u32 r = 0; r < 8; r++) B[r] |= (j >> 2) == r ? val : 0u;
        }
        const v8i Bv = {(int)B[0], (int)B[1], (int)B[2], (int)B[3], (int)B[4], (int)B[5], (int)B[6], (int)B[7]};
        const u32 msk = b == 3u ? 0x44444444u : 0x11111111u << b, sh = b == 3u ? 1u : 0u;
        v4i Mp[BITS];
#pragma unroll
        for (int p = 0; p < BITS; p++)
#pragma unroll
            for (int v = 0; v < 4; v++) Mp[p][v] = (int)((A[p][v] >> sh) & msk);
        const int sa = b == 0u ? 128 : (b == 1u ? 127 : 126);
#pragma unroll
        for (int cm = 1; cm < (1 << BITS); cm++) {
            v4i Am = {-1, -1, -1, -1};
#pragma unroll
            for (int p = 0; p < BITS; p++)
                if (cm & (1 << (BITS - 1 - p))) Am &= Mp[p];
            mfma_f4_bf8(acc[cm - 1], Am, Bv, sa, sb);
        }
    }
}

// 16-lane (DPP row) butterflies: every lane of the row ends up with the row's result -- no row_bcast steps, no readlane
__device__ __forceinline__ float row_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false));
    return v;
}
__device__ __forceinline__ u32 row_max(u32 v) {
    v = max(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false));
    v = max(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false));
    v = max(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false));
    v = max(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, false));
    return v;
}

// Geometry.  A 1024-weight chunk of a row is 32 plane words t; bit 8 (3 - c) + (7 - j) of word t is the weight of activation
// 1024 chunk + 256 c + 8 t + j (plane_core.h).  Unit q = (chunk, hh) = q >> 1, q & 1: words t = 16 hh + tt, tt = 0..15 -- 64
// bytes of every row and plane, 512 activations (4 runs of 128).
//   A: lane (r = l % 16, g = l / 16) loads the 16 bytes tt = 4 g + v (v = 0..3) of row r; register v masked at nibble bit b is
//      the FP4 operand element k = 32 g + 8 v + i = 8 tt + i (nibble i): activation c = 3 - i / 2, j = 7 - 4 (i & 1) - b.
//   image of a unit: [b][piece][k] bytes (4 x 4 x 128 = 2 KiB); pieces 2, 3 keep byte k at k ^ 32 (bank swizzle, plane_core.h).
//   B: lane (col = piece, kb) holds k = 16 kb .. + 15 and 64 + 16 kb .. + 15 of (b) in 8 registers, for all 4 b: 32 registers
//      per unit, loaded ONCE per (wave, unit); columns 4..15 hold zeros.
// Work: consumer unit cq = NH adjacent units (NH = 2: a whole chunk, one accumulation); wave w multiplies cq = w % NCU (+ W ..)
// for the row groups rgs, rgs + rgstep, ..  Image builders: wave w builds units w, w + W, .. (NPU of them); its lane
// (g = l / 16, v = l / 4 % 4, c = l % 4) owns the 8 activations (c, tt = 4 g + v, j = 0..7): a DPP row is a scale block.
// LDS: [meta: 64 units x 16 words {ready, extracted, scale byte, ..}][red: W floats, ctr][coefficients: rows x 2^b
//      floats][images: NC2 x 2 KiB][hot: NC2 x 8 x 8 B][part]
//   part (raw, 2-bit): [rgl][cq][subset][col 0..3][16 rows] floats; PSUM: the 4 piece columns added, [rgl][cq][subset][16]
constexpr u32 META_W = 16u;
#ifndef ST_XFLAGS
#define ST_XFLAGS 0  // build-time experiments: 1 no MFMA work, 2 no plane loads
#endif
#ifndef ST_MAINPRIO
#define ST_MAINPRIO 0
#endif
#ifndef ST_W2
#define ST_W2 16  // waves per block at 2 bits
#endif
#ifndef ST_W34
#define ST_W34 8  // ... at 3 and 4 bits (more accumulators and plane words per wave)
#endif
#ifndef ST_PF
#define ST_PF 2
#endif
#ifndef ST_KO
#define ST_KO 0   // build-time knock-outs, WRONG numerics, timing only (bit mask): 1 no sum-of-squares exchange, 8 no image reads, 16 the activations arrive as 4 fp32 partial vectors (loads only)
#endif
#ifndef ST_DPPIMG
#define ST_DPPIMG 0  // 1: consumers read their unit image with two dense ds_read_b128 and spread it over the piece lanes by DPP instead of
                     // eight reads that only the 16 piece lanes take part in -- correct (GPU suite green) and SLOWER: wqkv 5.80 vs 5.65 us, decode 904
                     // vs 911 tokens/s (profiles/r05_stream_phase_variants.txt): the masked reads are not what the image phase waits for
#endif
#ifndef ST_TAU_SCALE
#define ST_TAU_SCALE 1  // a unit's power of two from its extraction threshold (what stays in the image is <= tau) instead of its maximum:
                        // one wave-wide reduction less per unit; the extracted elements take their own power of two (round 5)
#endif
#ifndef ST_NH
#define ST_NH 1   // unit halves per consumer unit: 1 = a wave multiplies half chunks (32 image registers), 2 = whole chunks (64)
#endif
template <int BITS>
constexpr int st_waves() { return BITS == 2 ? ST_W2 : ST_W34; }
// EPI: the epilogue compiled in -- EPI_ANY reads the launch's flags (a.rope / a.pairs / a.part_out / a.resid / a.ssq_in) at run time;
// the decode step's wqkv launch (EPI_ROPE) has an instance of its own: every wave-uniform flag test is a scalar compare and a branch
// that no wave overlaps with anything in the prologue (the phase stamps alone, ~20 such sites, were 2 % of the decode step:
// profiles/r05_stamps_compiled_out.txt); 51 -> 33 conditional branches in front of the first MFMA, wqkv 5.77 -> 5.61 us
enum { EPI_ANY = -1, EPI_ROPE = 1, EPI_PAIRS = 2 };
template <int BITS, int PRO, int NPU, bool PSUM, int EPI = EPI_ANY>
__device__ __forceinline__ void ap_stream_body(const StreamArgs &a) {
    constexpr int WV = st_waves<BITS>(), NH = ST_NH;
    static_assert(EPI != EPI_ROPE || !PSUM, "the RoPE epilogue rotates raw-parked pairs");
    const bool f_rope = EPI == EPI_ANY ? a.rope != 0u : EPI == EPI_ROPE;
    const bool f_pairs = EPI == EPI_ANY ? a.pairs != 0u : EPI == EPI_PAIRS;
    const bool f_part = EPI == EPI_ANY && a.part_out != nullptr;
    const bool f_resid = EPI == EPI_ANY && a.resid != nullptr;
    constexpr int NP = 1 << BITS, NP1 = NP - 1;
    constexpr u32 W = WV, T = 64u * W;
#ifndef ST_RING
#define ST_RING 4
#endif
    constexpr u32 RING = ST_RING;   // plane-word register slots (consumer units) per wave
    constexpr u32 PF = ST_PF;  // consumer units requested ahead (< RING)
    constexpr u32 LPU = (u32)BITS * NH;  // plane loads per consumer unit
    constexpr u32 NCOL = PSUM ? 1u : 4u;
    static_assert(PSUM || BITS == 2, "raw parking: the (column, Moebius index) lanes of a row quad must fit a DPP row");
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    // Every wave instruction costs its SIMD a 4-cycle issue slot and four waves share a SIMD: code that all 16 waves run is
    // paid 4 x per SIMD (the first version spent 2,000 cycles in ~200 instructions per wave before its first plane load).
    // So: flags at a fixed LDS offset (zeroed by wave 0 without reading a kernel argument), shifts instead of divisions (the
    // host passes logarithms), the image builders at raised priority, statistics per DPP row.
    const u32 tid = threadIdx.x;
    const u32 w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const u32 l = tid & 63u;
    u32 *meta = reinterpret_cast<u32 *>(smem);
    float *red = reinterpret_cast<float *>(smem + 4096u);
    u32 *ctr = reinterpret_cast<u32 *>(smem + 4096u + 64u);
    float *ctab = reinterpret_cast<float *>(smem + 4096u + 128u);
    if (w == 0u) {
#pragma unroll
        for (u32 i = 0; i < 4u; i++) reinterpret_cast<uint4 *>(meta)[l + 64u * i] = make_uint4(0u, 0u, 0u, 0u);
        if (l == 0u) ctr[0] = 0u;
    }
    // phase stamps (tools/phase_timing.py): kept in LDS and written out at the very end -- a global store per stamp sits in
    // vmcnt and turns the next wait for a load into a wait for the store's acknowledgement (~1,000 cycles each)
    unsigned long long *dbgl = reinterpret_cast<unsigned long long *>(smem + a.dbg_off) + w * 24u;
    const bool dbg_on = GQ_STAMPS == 1 && a.dbg && blockIdx.x == gridDim.x / 2;
    // (GQ_STAMPS == 2: start / end of EVERY block, s_memrealtime -- dbg[2 b], dbg[2 b + 1]; tools/r6/block_ramp.py)
    if (GQ_STAMPS == 2 && a.dbg && tid == 0u && blockIdx.y == 0u) a.dbg[2u * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
    auto stamp = [&](int i) {
        if (dbg_on && l == 0) dbgl[i] = __builtin_readcyclecounter();
    };
    auto stamp2 = [&](int i) { stamp(8 + i); };
    if (dbg_on && l < 24u) dbgl[l] = 0ull;
    stamp(0);
    const u32 nchunks = a.K >> 10, NC2 = 2u * nchunks, NCU = NC2 / NH;
    const u32 npw = NC2 < W ? NC2 : W;  // waves with an image to build
    const bool is_pro = w < npw;
    // (round 5, measured and not kept: the later-started builder of a SIMD at a higher priority than the earlier one -- alone the
    // launch gains 0.1 us on w1w3, in the decode graph the step loses 0.5 %; a longer poll interval of the consumers' ready-flag
    // loop and the main loop aligned to 64 / 256 bytes change nothing: profiles/r05_stream_phase_variants.txt)
    if (is_pro) __builtin_amdgcn_s_setprio(3);
    // RoPE epilogue: the position is requested first thing (a vector load: it returns in order ahead of everything else of this
    // wave and is looked at only behind the image build -- reading it where the cos / sin addresses are formed would park the wave
    // for a memory round trip in front of its prologue: measured, 1 us per launch)
    u32 posv = 0;
    if (f_rope) posv = (u32)__builtin_amdgcn_raw_buffer_load_b32(make_rsrc(a.pos, 4u), 0, 0, 0);
    unsigned char *img = smem + a.img_off;
    HotEnt *hotl = reinterpret_cast<HotEnt *>(img + (size_t)NC2 * 2048u);
    float *xpart = reinterpret_cast<float *>(hotl + (size_t)NC2 * HOTCAP);  // [NC2][4][16]
    float *part = xpart + (size_t)NC2 * 64u;
    const u32 rg0 = blockIdx.x * a.RGB;
    const u32 ksl = blockIdx.y;  // K slice (0 without a K split over blocks)
    // Which 16 rows a row group is: any 16 rows do (every lane addresses its own row).  Plain: 16 consecutive rows.  RoPE epilogue:
    // MFMA row i of group rg is row  head * hd + 8 (rg % (hd / 16)) + i / 2 + (hd / 2) (i % 2),  head = rg / (hd / 16): rows (2 m,
    // 2 m + 1) of a group are the rotation partners (d, d + hd / 2) of one head (rotate_half, inference/model.py:330-341), so the
    // epilogue lane that owns 4 consecutive MFMA rows owns two whole pairs -- the stored tensor keeps the reference's row order.
    auto grp_base = [&](u32 rg) -> u32 { return f_rope ? ((rg >> (a.lhd - 4u)) << a.lhd) + 8u * (rg & ((1u << (a.lhd - 4u)) - 1u)) : rg * 16u; };
    auto grp_row = [&](u32 i) -> u32 { return f_rope ? (i >> 1) + ((i & 1u) << (a.lhd - 1u)) : i; };

    // ---------------------------------------------------------------- 0. requests: activations first, then one unit of planes
    const u32 xg = l >> 4, xc = l & 3u, xtt = 4u * xg + ((l >> 2) & 3u);
    u32x4 xv[NPU], nv[NPU];
    // RMSNorm with the sum of squares handed over by the producing launch (a.ssq_in: GQ_SSQ_SLOTS partial sums): ONE wave -- the
    // first that builds no image, else the last -- adds them (4 x 16 bytes per lane, requested first thing) and leaves the scale in
    // LDS in front of the launch barrier; the builders read it back behind the barrier while their activations are still in
    // flight.  No exchange of per-wave sums (an LDS counter, a spin, a read-back) between the activations' arrival and the
    // normalisation: measured upper bound 0.4 us of the wqkv launch (profiles/r05_stream_knockouts.txt).
    const bool ho = PRO == PRO_RMSNORM && EPI == EPI_ANY && a.ssq_in != nullptr;
    if (PRO == PRO_RMSNORM && ho && w == (npw < W ? npw : W - 1u)) {
        // Requests, wait and sum inside ONE branch, from inline asm: left to the compiler, the wait for these loads lands at the
        // join behind the branch as vmcnt(0) for EVERY wave -- the builders then drain their activation loads, the last waves
        // their LUT rows, in front of the launch barrier (measured: barrier at 4,000 instead of 2,200 cycles).  The wave is an
        // idle consumer at this point; its own first plane request waits for the round trip.
        const u32x4 rss = make_rsrc4(a.ssq_in, (u32)GQ_SSQ_SLOTS * 4u);
        u32x4 sqv[GQ_SSQ_SLOTS / 256];
#pragma unroll
        for (u32 i = 0; i < (u32)GQ_SSQ_SLOTS / 256u; i++) {
            asm volatile("" : "=v"(sqv[i]));
            asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:%3" : "+v"(sqv[i]) : "v"(16u * l), "s"(rss), "n"(1024 * i) : "memory");
        }
        wait_vm<0>();
        float ss = 0.f;
#pragma unroll
        for (u32 i = 0; i < (u32)GQ_SSQ_SLOTS / 256u; i++) {
            asm volatile("" : "+v"(sqv[i]));
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const u32 t = sqv[i][k];  // (through a scalar: a bit_cast of a vector element folds to element 0 with this hipcc)
                ss += __builtin_bit_cast(float, t);
            }
        }
        ss = wave_reduce<false>(ss);
        if (l == 63u) lds_store(reinterpret_cast<u32 *>(red), __builtin_bit_cast(u32, 1.0f / sqrtf(ss / (float)a.Kx + a.eps)));
    }
    if (is_pro) {
        const rsrc_t rsx = make_rsrc(a.x, (PRO == PRO_SILUMUL ? 4u : 2u) * a.Kx);
        const rsrc_t rsn = make_rsrc(PRO == PRO_RMSNORM ? a.normw : a.x, 2u * a.Kx);
        const u32 xs0 = 2u * ksl * a.K;  // (the block's K slice)
#pragma unroll
        for (u32 n = 0; n < (u32)NPU; n++) {
            const u32 q = w + n * W;
            const u32 voff = q < NC2 ? 2u * (1024u * (q >> 1) + 256u * xc + 8u * (16u * (q & 1u) + xtt)) : OOB;
            xv[n] = bload128<0>(rsx, voff, xs0);
            if constexpr (PRO == PRO_RMSNORM) nv[n] = bload128<0>(rsn, voff, xs0);
            if constexpr (PRO == PRO_SILUMUL) nv[n] = bload128<0>(rsx, voff, xs0 + 2u * a.Kx);
        }
#if ST_KO & 16  // probe: the activations as 4 fp32 partial vectors (8 more 16-byte loads per lane, the same 64 KiB for every block)
        {
            const rsrc_t rsp = make_rsrc(a.qw, 65536u);
            u32x4 pv[8];
#pragma unroll
            for (u32 i = 0; i < 8u; i++) pv[i] = bload128<0>(rsp, (w * 64u + l) * 32u + 16u * (i & 1u), 16384u * (i >> 1));
#pragma unroll
            for (u32 i = 0; i < 8u; i++) asm volatile("" ::"v"(pv[i]));
        }
#endif
    }
    stamp2(0);
    // the LUT rows of the block: thread T - 1 - i takes row i (the waves that start last and have the least to do); the
    // Moebius coefficients go to LDS for the epilogue
    const u32 crow = T - 1u - tid;
    const bool has_crow = crow < a.RGB * 16u;
    u32 lutw[NP / 2];
#pragma unroll
    for (int k = 0; k < NP / 2; k++) asm volatile("" : "=v"(lutw[k]));
    if (w >= W - ((a.RGB * 16u + 63u) >> 6)) {  // (wave-uniform)
        const u32x4 rl = make_rsrc4(a.lut, a.N * (u32)NP * 2u);
        const u32 voff = has_crow ? (grp_base(rg0 + (crow >> 4)) + grp_row(crow & 15u)) * (u32)NP * 2u : OOB;
        if constexpr (NP == 4) {
            asm volatile("buffer_load_dwordx2 %0, %1, %2, 0 offen" : "+v"(*reinterpret_cast<u32x2 *>(lutw)) : "v"(voff), "s"(rl) : "memory");
        } else {
#pragma unroll
            for (int k4 = 0; k4 < NP / 8; k4++)
                asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:%3"
                             : "+v"(*reinterpret_cast<u32x4 *>(lutw + 4 * k4))
                             : "v"(voff), "s"(rl), "n"(16 * k4)
                             : "memory");
        }
    }
    // consumer units of this wave (a.lq = log2 NCU when NCU < W -- the waves of a K range split its row groups --, else 31)
    const bool split = a.lq < 31u;
    const u32 q0 = split ? (w & (NCU - 1u)) : w;
    const u32 rgs = split ? w >> a.lq : 0u, lrs = split ? a.lw - a.lq : 0u;  // first row group, log2 of the row-group stride
    const u32 nq = q0 < NCU ? ((NCU - 1u - q0) >> a.lw) + 1u : 0u;
    const u32 nrg = a.RGB > rgs ? ((a.RGB - rgs - 1u) >> lrs) + 1u : 0u;
    const u32 n_units = nq * nrg;
    const u32 plane_bytes = a.N * a.wpr_ld * 4u;
    const u32x4 rq = make_rsrc4(a.qw, plane_bytes * (u32)BITS);
    const u32 lane_off = (ST_XFLAGS & 2) ? OOB : (grp_row(l & 15u) * a.wpr_ld + 4u * (l >> 4)) * 4u;  // (experiment 2: no plane loads)
    u32x4 Ar[RING][NH][BITS];
#pragma unroll
    for (u32 s = 0; s < RING; s++)
#pragma unroll
        for (u32 hh = 0; hh < (u32)NH; hh++)
#pragma unroll
            for (u32 p = 0; p < (u32)BITS; p++) asm volatile("" : "=v"(Ar[s][hh][p]));
    // the units of a wave in order: for each of its K ranges (cq = q0, q0 + W, ..) its row groups (rgs, rgs + rgstep, ..)
    u32 i_q = q0, i_rg = rgs, i_ri = 0, iu = 0;  // next unit to request
    auto issue = [&](auto SLOT) {
        constexpr u32 s = decltype(SLOT)::value;
        const u32 soff = (grp_base(rg0 + i_rg) * a.wpr_ld + 16u * NH * i_q) * 4u + ksl * (a.K >> 3);  // (+ the K slice: K / 32 words)
#pragma unroll
        for (u32 hh = 0; hh < (u32)NH; hh++)
#pragma unroll
            for (u32 p = 0; p < (u32)BITS; p++) aload128(Ar[s][hh][p], rq, lane_off, soff + 64u * hh + p * plane_bytes);
        i_rg += 1u << lrs;
        if (++i_ri == nrg) i_ri = 0, i_rg = rgs, i_q += W;
        iu++;
    };
    if (n_units > 0u) issue(std::integral_constant<u32, 0>{});
    stamp(6);
    // flags and counters start at zero: one hardware barrier.  (The CU's memory pipe serves requests in order: a builder's
    // activation requests are in front of its own plane words; the other waves' first units are requested ~100 instructions
    // into the kernel, behind the builders' first two loads.)
    __syncthreads();
    stamp2(1);
    auto coefficients = [&]() {  // the block's Moebius coefficients -> LDS (needs the LUT request landed: wait_vm<..> by the caller)
        if (w >= W - ((a.RGB * 16u + 63u) >> 6)) {
#pragma unroll
            for (int k = 0; k < NP / 2; k++) asm volatile("" : "+v"(lutw[k]));
            float f[NP];
#pragma unroll
            for (int cc = 0; cc < NP / 2; cc++) {
                f[2 * cc] = h2f((uint16_t)(lutw[cc] & 0xFFFF));
                f[2 * cc + 1] = h2f((uint16_t)(lutw[cc] >> 16));
            }
            moebius<BITS>(f);
            if (has_crow) {  // [c][row]: an epilogue lane reads the coefficients of its 4 rows in one 16-byte read
#pragma unroll
                for (int cc = 0; cc < NP; cc++) ctab[(size_t)cc * (a.RGB * 16u) + crow] = f[cc];
            }
        }
    };
    // epilogue lane = (piece column, Moebius index c, row quad, row group): its residual elements now
    const u32 e_col = tid & (NCOL - 1u), e_c = (tid / NCOL) & (u32)NP1, e_rq = (tid / (NCOL * NP)) & 3u, e_rgl = tid / (NCOL * NP * 4u);
    u32x2 rres = {0u, 0u};
    if (f_resid && !f_pairs && e_rgl < a.RGB && e_c == 0u && e_col == 0u)
        rres = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(make_rsrc(a.resid, a.N * 2u), (int)(2u * ((rg0 + e_rgl) * 16u + 4u * e_rq)), 0, 0));

    // ---------------------------------------------------------------- 1. images of this wave's units
    if (is_pro) {
        float nscale = 1.f;
        if (PRO == PRO_RMSNORM && ho) {
            nscale = __builtin_bit_cast(float, lds_load(reinterpret_cast<const u32 *>(red)));  // (published by the launch barrier)
            stamp2(2);
            stamp2(3);
        } else if constexpr (PRO == PRO_RMSNORM) {
            float ss = 0.f;
            asm volatile("" : "+v"(xv[0]));
            stamp2(2);
#pragma unroll
            for (u32 n = 0; n < (u32)NPU; n++)
#pragma unroll
                for (int k = 0; k < 4; k++)  // (requests outside the vector returned zeros)
                    ss = __builtin_amdgcn_fdot2(u2h2(xv[n][k]), u2h2(xv[n][k]), ss, false);
            ss = wave_reduce<false>(ss);
#if ST_KO & 1
            nscale = 1.0f / sqrtf(lane63(ss) * (float)npw / (float)a.K + a.eps);
#else
            if (l == 63) {
                reinterpret_cast<u32 *>(red)[w] = __builtin_bit_cast(u32, ss);
                asm volatile("" ::: "memory");
                __hip_atomic_fetch_add((lds_u32 *)ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            stamp2(3);
            lds_wait_ge(ctr, npw);
            // (all partial sums in one LDS round trip: volatile reads would be served one by one, ~100 cycles each)
            v4f rp[W / 4];
#pragma unroll
            for (u32 i = 0; i < W / 4u; i++) rp[i] = *reinterpret_cast<const v4f *>(red + 4u * i);
            float tot = 0.f;
#pragma unroll
            for (u32 i = 0; i < W; i++) tot += i < npw ? rp[i / 4u][i % 4u] : 0.f;
            nscale = 1.0f / sqrtf(tot / (float)a.K + a.eps);
#endif
        }
        stamp(7);
#pragma unroll
        for (u32 n = 0; n < (u32)NPU; n++) {
            const u32 q = w + n * W;
            if (q >= NC2) continue;
            // transform: (x.float() * rsqrt(mean(x^2)+eps)).half() * w -- inference/model.py:281-292, both fp16 roundings kept
            u32 xw[4];
            const h2v one2 = u2h2(0x3C003C00u);
            us2 mxp = {0, 0};
            float s1 = 0.f, xsum = 0.f;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                u32 t = xv[n][k];
                if constexpr (PRO == PRO_RMSNORM) {
                    const _Float16 h0 = (_Float16)gq_pin_f32(h2f(t & 0xFFFF) * nscale), h1 = (_Float16)gq_pin_f32(h2f(t >> 16) * nscale);
                    t = h22u((h2v){h0, h1} * u2h2(nv[n][k]));
                }
                if constexpr (PRO == PRO_SILUMUL) t = silu_mul2(t, nv[n][k]);
                xw[k] = t;
                const u32 ab = t & 0x7FFF7FFFu;  // |fp16| bit patterns order like unsigned integers
                mxp = __builtin_elementwise_max(mxp, __builtin_bit_cast(us2, ab));
                s1 = __builtin_amdgcn_fdot2(u2h2(ab), one2, s1, false);
                xsum = __builtin_amdgcn_fdot2(u2h2(t), one2, xsum, false);
            }
            // statistics of the unit: maximum -> power of two (ONE scale per unit, the same in every lane: a version with one scale
            // per 128-activation block, supplied by the block's lane group, failed test_hot_channels_* by 5.7e-4 of sum|w||x| as
            // soon as the blocks of a unit differed -- the scale operand does not act per 32-element block the way that assumed),
            // mean magnitude -> extraction threshold; per DPP row the sum for the coef[0] term
            const u32 mxl = max((u32)mxp[0], (u32)mxp[1]);
            const float tot1 = lane63(wave_reduce<false>(s1));
            xsum = row_sum(xsum);
            // threshold of the extraction as an fp16 bit pattern (rounded up): 64 x the unit's mean magnitude, + 1 % for the roundings
            u32 tau = 0x7C00u;
            {
                const float tf = GQ_ST_HOT_T * 1.01f * tot1 * (1.0f / 512.0f);
                if (tf < 65000.f) tau = (u32)__builtin_bit_cast(uint16_t, (_Float16)tf) + 1u;
            }
#if ST_TAU_SCALE
            // what stays in the image is <= tau: the unit's power of two from the threshold, no maximum over the wave (a unit whose
            // threshold is not finite -- mean magnitude above 1000 -- takes its maximum)
            float xmax = h2f((uint16_t)tau);
            if (__builtin_expect(tau >= 0x7C00u, 0)) xmax = lane63(wave_reduce<true>(h2f((uint16_t)mxl)));
            if (tot1 == 0.f) xmax = 0.f;
#else
            const float xmax = lane63(wave_reduce<true>(h2f((uint16_t)mxl)));
#endif
            stamp2(4);
            // x * 2^k with max|x| * 2^k in [2^14, 2^15) (plane_core.h piece_shift): k + 15 = 44 - (biased fp16 exponent of the maximum)
            const u32 ke = (u32)(piece_shift(xmax) + 15);
            const u32 k16 = ke << 10;  // fp16 bits of 2^k
            HotEnt *hl = hotl + (size_t)q * HOTCAP;
            u32 nh = 0;
            u32 k16h = k16, keh = ke;  // the extracted elements' own power of two (ST_TAU_SCALE: the image's comes from the threshold)
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(mxl > tau) != 0ull, 0)) {
                // rare: the lane's elements above the threshold leave the image (ascending lane order: deterministic list)
#if ST_TAU_SCALE
                keh = (u32)(piece_shift(lane63(wave_reduce<true>(h2f((uint16_t)mxl)))) + 15);
                k16h = keh << 10;
#endif
#pragma unroll
                for (u32 k = 0; k < 4; k++)
#pragma unroll
                    for (u32 hf = 0; hf < 2; hf++) {
                        const u32 xh = (xw[k] >> (16u * hf)) & 0xFFFFu;
                        const bool hot = (xh & 0x7FFFu) > tau;
                        const unsigned long long bal = __builtin_amdgcn_ballot_w64(hot);
                        if (bal == 0ull) continue;
                        const u32 idx = nh + (u32)__builtin_popcountll(bal & ((1ull << l) - 1ull));
                        if (hot && idx < HOTCAP) {
                            const u32 j = 2u * k + hf, sbit = 7u - j;  // plane bit s = 7 - j: b = s & 3, nibble i = 2 (3 - c) + (s >> 2)
                            _Float16 rem = __builtin_bit_cast(_Float16, (uint16_t)xh) * __builtin_bit_cast(_Float16, (uint16_t)k16h);
                            u32 pieces = 0;
#pragma unroll
                            for (u32 p = 0; p < 4; p++) {
                                const uint16_t pb = __builtin_bit_cast(uint16_t, rem) & 0xFF00u;
                                pieces |= (u32)(pb >> 8) << (8u * p);
                                rem = rem - __builtin_bit_cast(_Float16, pb);
                            }
                            hl[idx].key = ((sbit & 3u) << 7) | (8u * xtt + 2u * (3u - xc) + (sbit >> 2));
                            hl[idx].pieces = pieces;
                        }
                        if (hot) xw[k] &= ~(0xFFFFu << (16u * hf));
                        nh += (u32)__builtin_popcountll(bal);
                    }
                nh = nh < HOTCAP ? nh : HOTCAP;
            }
            // split into 4 exact bf8 pieces and scatter: the lane's 8 activations are, per nibble bit b, the two bytes
            // k0 (j = 7 - b) and k0 + 1 (j = 3 - b), k0 = 8 tt + 2 (3 - c)
            const h2v kk = u2h2(k16 * 0x10001u);
            u32 P[4][4];  // [piece][word]: bf8 of element 2 word in byte 1, of 2 word + 1 in byte 3
#pragma unroll
            for (int k = 0; k < 4; k++) {
                h2v rem = u2h2(xw[k]) * kk;
#pragma unroll
                for (u32 p = 0; p < 4; p++) {
                    P[p][k] = h22u(rem) & 0xFF00FF00u;
                    if (p < 3) rem = rem - u2h2(P[p][k]);
                }
            }
            stamp2(5);
            unsigned char *dst = img + (size_t)q * 2048u;
            const u32 k0 = 8u * xtt + 2u * (3u - xc);
#pragma unroll
            for (u32 p = 0; p < 4; p++) {
                // b = 0: j = 7, 3 (byte 3 of words 3, 1); b = 1: j = 6, 2 (byte 1 of words 3, 1)
                const u32 v01 = __builtin_amdgcn_perm(P[p][1], P[p][3], 0x05010703u);
                // b = 2: j = 5, 1 (byte 3 of words 2, 0); b = 3: j = 4, 0 (byte 1 of words 2, 0)
                const u32 v23 = __builtin_amdgcn_perm(P[p][0], P[p][2], 0x05010703u);
#if ST_DPPIMG
                // (dense-read layout: row c = 4 b + p keeps byte k at k ^ 16 (c / 2) = k ^ (32 b + 16 (p / 2)) -- see the consumers' read)
                unsigned char *d = dst + p * 128u;
                const u32 kx = k0 ^ (16u * (p >> 1));
                *reinterpret_cast<uint16_t *>(d + kx) = (uint16_t)v01;
                *reinterpret_cast<uint16_t *>(d + 512u + (kx ^ 32u)) = (uint16_t)(v01 >> 16);
                *reinterpret_cast<uint16_t *>(d + 1024u + (kx ^ 64u)) = (uint16_t)v23;
                *reinterpret_cast<uint16_t *>(d + 1536u + (kx ^ 96u)) = (uint16_t)(v23 >> 16);
#else
                unsigned char *d = dst + p * 128u + (k0 ^ bimg_swz(p));
                *reinterpret_cast<uint16_t *>(d) = (uint16_t)v01;
                *reinterpret_cast<uint16_t *>(d + 512u) = (uint16_t)(v01 >> 16);
                *reinterpret_cast<uint16_t *>(d + 1024u) = (uint16_t)v23;
                *reinterpret_cast<uint16_t *>(d + 1536u) = (uint16_t)(v23 >> 16);
#endif
            }
            // sum(x) as a pseudo partial sum [unit][piece column = g][16 rows]: the epilogue lanes of the coef[0] term run the
            // same loads and additions as the others (no divergent branch on the tail of the kernel)
            xpart[q * 64u + l] = xsum;
            if (l == 0u) {
                meta[META_W * q + 1u] = nh;
                meta[META_W * q + 2u] = 142u - ke;  // E8M0 byte of 2^-k: 127 - k
                meta[META_W * q + 3u] = 142u - keh;
            }
            lds_store(meta + META_W * q, 1u);  // (every lane stores the same flag behind its own image bytes)
        }
        __builtin_amdgcn_s_setprio(0);
    }
    stamp(1);
    // RoPE epilogue: lane (c = 0, col = k < 2) of a row quad rotates the pair (rows 2 k, 2 k + 1 of the quad): its cos / sin now
    u32 rp_row = 0, rp_cs[4] = {0u, 0u, 0u, 0u};
    u32 rp_pos = 0;
    if (f_rope) {
        rp_pos = __builtin_amdgcn_readfirstlane(posv);
        rp_row = grp_base(rg0 + e_rgl) + 2u * e_rq + e_col;  // the pair's first row (d < hd / 2), partner at + hd / 2
        if (e_rgl < a.RGB) {  // (the epilogue's waves)
            const u32 hd = 1u << a.lhd, d = rp_row & (hd - 1u);
            const bool mine = e_c == 0u && e_col < 2u && (rp_row >> a.lhd) < a.H + a.Hkv && rp_pos < a.max_seq;
            const rsrc_t rc = make_rsrc(a.cos_t, a.max_seq * hd * 2u), rs_ = make_rsrc(a.sin_t, a.max_seq * hd * 2u);
            const u32 o = mine ? (rp_pos * hd + d) * 2u : OOB;
            rp_cs[0] = (u32)__builtin_amdgcn_raw_buffer_load_b16(rc, (int)o, 0, 0);
            rp_cs[1] = (u32)__builtin_amdgcn_raw_buffer_load_b16(rc, (int)o, (int)hd, 0);  // (+ hd / 2 elements)
            rp_cs[2] = (u32)__builtin_amdgcn_raw_buffer_load_b16(rs_, (int)o, 0, 0);
            rp_cs[3] = (u32)__builtin_amdgcn_raw_buffer_load_b16(rs_, (int)o, (int)hd, 0);
        }
    }
    // (ONE request site per register slot: a second site in another branch gets registers of its own and a copy at the
    // join -- made before the data lands)
    if (is_pro) {
        wait_vm<0>();  // (the LUT rows and the first unit, long landed; nothing else of this wave is in flight)
        coefficients();
    }
    stamp2(6);
    if (1u < n_units) issue(std::integral_constant<u32, 1>{});
    if constexpr (PF >= 3u) {
        if (2u < n_units) issue(std::integral_constant<u32, 2>{});
    }
    stamp2(7);
    if (!is_pro) {
        wait_vm_units<LPU>(iu > 1u ? iu - 1u : 0u);  // everything in front of the second unit
        coefficients();
    }

    // ---------------------------------------------------------------- 2. the units of this wave
    // Instruction issue is arbitrated by priority, then age: left alone the oldest wave of a SIMD runs ahead and the youngest is left
    // to multiply its units alone at the end, at the single-wave rate (tools/ubench/mfma_mix.hip: 19.6 ns per MFMA against 15.2 ns
    // with four waves interleaving) -- ST_MAINPRIO: the later a wave started, the higher its priority in the main loop
#if ST_MAINPRIO == 1
    __builtin_amdgcn_s_setprio(0);
    if (w >= 3u * W / 4u) __builtin_amdgcn_s_setprio(3);
    else if (w >= W / 2u) __builtin_amdgcn_s_setprio(2);
    else if (w >= W / 4u) __builtin_amdgcn_s_setprio(1);
#elif ST_MAINPRIO == 2
    if (w >= W / 2u) __builtin_amdgcn_s_setprio(1);
#endif
    const u32 col = l & 15u, kb = l >> 4;
    v8i Bv[NH][4];
    int sb[NH];
    u32 nhot[NH];
    constexpr u32 PUNIT = (u32)NP1 * NCOL * 16u;  // floats parked per unit
    const u32 part_rg = (NCU * PUNIT) << lrs, part_q = W * PUNIT - nrg * part_rg;
    float *c_pp = part + (size_t)(rgs * NCU + q0) * PUNIT + (PSUM ? 4u * kb : col * 16u + 4u * kb);
    u32 c_q = q0, c_ri = 0;
    bool stamped = false;
    auto consume = [&](auto SLOT, u32 u) {
        constexpr u32 s = decltype(SLOT)::value;
        if (c_ri == 0u) {
            // a new K range: wait for its image, then hold it in registers for all row groups
#pragma unroll
            for (u32 hh = 0; hh < (u32)NH; hh++) {
                const u32 q = c_q * NH + hh;
                u32x2 fl;
                do {  // {ready, extracted} in one read
                    fl = *(volatile __attribute__((address_space(3))) u32x2 *)(lds_u32 *)(meta + META_W * q);
                    asm volatile("" ::: "memory");
                    if (fl[0] == 0u) __builtin_amdgcn_s_sleep(1);
                } while (fl[0] == 0u);
                nhot[hh] = __builtin_amdgcn_readfirstlane(fl[1]);
                sb[hh] = (int)meta[META_W * q + 2u];
#if ST_DPPIMG
                // Round 5 experiment: the image comes in with TWO dense reads (every lane 16 bytes: lane (row kb, column c = 4 b + p)
                // takes the units kb and kb + 4 of image row c) instead of eight reads that only the 16 piece lanes take part in.  The
                // piece lanes (columns 0..3 of each DPP row) then pull the rows of the other nibble bits from the lanes 4 b to
                // their right (row_shl, bank mask 1: the other lanes keep the zero they start from -- the idle MFMA columns stay
                // zero without a read).  Image row c keeps its unit u at u ^ (c / 2): the 16 lanes of a DPP row hit 16 different
                // 16-byte bank groups.
                {
                    const unsigned char *src = img + (size_t)q * 2048u + col * 128u;
                    const u32 u0x = (kb ^ (col >> 1)) << 4, u1x = ((kb + 4u) ^ (col >> 1)) << 4;
                    const uint4 r0 = *reinterpret_cast<const uint4 *>(src + u0x), r1 = *reinterpret_cast<const uint4 *>(src + u1x);
                    const int raw[8] = {(int)r0.x, (int)r0.y, (int)r0.z, (int)r0.w, (int)r1.x, (int)r1.y, (int)r1.z, (int)r1.w};
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        Bv[hh][0][j] = __builtin_amdgcn_update_dpp(0, raw[j], 0xE4, 0xF, 0x1, false);   // quad_perm [0,1,2,3]: own lane
                        Bv[hh][1][j] = __builtin_amdgcn_update_dpp(0, raw[j], 0x104, 0xF, 0x1, false);  // row_shl:4
                        Bv[hh][2][j] = __builtin_amdgcn_update_dpp(0, raw[j], 0x108, 0xF, 0x1, false);  // row_shl:8
                        Bv[hh][3][j] = __builtin_amdgcn_update_dpp(0, raw[j], 0x10C, 0xF, 0x1, false);  // row_shl:12
                    }
                }
#else
                const unsigned char *src = img + (size_t)q * 2048u + (col & 3u) * 128u + ((16u * kb) ^ bimg_swz(col & 3u));
#pragma unroll
                for (u32 b = 0; b < 4; b++) {
                    uint4 b0 = make_uint4(0, 0, 0, 0), b1 = make_uint4(0, 0, 0, 0);
                    if (col < 4u && !(ST_KO & 8)) {
                        b0 = *reinterpret_cast<const uint4 *>(src + b * 512u);
                        b1 = *reinterpret_cast<const uint4 *>(src + b * 512u + 64u);
                    }
                    Bv[hh][b] = (v8i){(int)b0.x, (int)b0.y, (int)b0.z, (int)b0.w, (int)b1.x, (int)b1.y, (int)b1.z, (int)b1.w};
                }
#endif
            }
            if (!stamped) stamp(2), stamped = true;
        }
        // the requests behind this unit's: up to PF - 1 more units, or none at the tail
        wait_vm_units<LPU>(iu - 1u - u);
#pragma unroll
        for (u32 hh = 0; hh < (u32)NH; hh++)
#pragma unroll
            for (u32 p = 0; p < (u32)BITS; p++) asm volatile("" : "+v"(Ar[s][hh][p]));
        v4f acc[NP1];
#if !(ST_XFLAGS & 1)
        {
#pragma unroll
            for (u32 hh = 0; hh < (u32)NH; hh++) {
                if (hh == 0u) mfma_b<BITS, 0, true>(acc, Ar[s][hh], Bv[hh][0], sb[hh]);
                else mfma_b<BITS, 0>(acc, Ar[s][hh], Bv[hh][0], sb[hh]);
                mfma_b<BITS, 1>(acc, Ar[s][hh], Bv[hh][1], sb[hh]);
                mfma_b<BITS, 2>(acc, Ar[s][hh], Bv[hh][2], sb[hh]);
                mfma_b<BITS, 3>(acc, Ar[s][hh], Bv[hh][3], sb[hh]);
                if (__builtin_expect(nhot[hh] != 0u, 0))
                    hot_unit<BITS>(acc, Ar[s][hh], hotl + (size_t)(c_q * NH + hh) * HOTCAP, nhot[hh], (int)meta[META_W * (c_q * NH + hh) + 3u], col, kb);
            }
        }
#else  // experiment: no MFMA work
#pragma unroll
        for (int i = 0; i < NP1; i++) acc[i] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (u32 hh = 0; hh < (u32)NH; hh++) acc[0][0] += __builtin_bit_cast(float, Ar[s][hh][0][0] ^ Ar[s][hh][BITS - 1][3]);
#endif
        // park: lane (col, kb) owns rows 4 kb .. 4 kb + 3 of column col
        if constexpr (PSUM) {
#pragma unroll
            for (int cm = 0; cm < NP1; cm++) {
                v4f v = acc[cm];
#pragma unroll
                for (int q4 = 0; q4 < 4; q4++) {
                    float f = v[q4];
                    f += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, f), 0xB1, 0xF, 0xF, false));
                    f += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, f), 0x4E, 0xF, 0xF, false));
                    v[q4] = f;
                }
                if (col == 0u) *reinterpret_cast<v4f *>(c_pp + (size_t)cm * 16u) = v;
            }
        } else if (col < 4u) {
#pragma unroll
            for (int cm = 0; cm < NP1; cm++) *reinterpret_cast<v4f *>(c_pp + (size_t)cm * 64u) = acc[cm];
        }
        c_pp += part_rg;
        if (++c_ri == nrg) c_ri = 0, c_q += W, c_pp += part_q;
        if (iu < n_units) issue(std::integral_constant<u32, (s + PF) % RING>{});  // (iu == u + PF here)
    };
    for (u32 u0 = 0; u0 < n_units; u0 += RING) {
        if (u0 + 0u < n_units) consume(std::integral_constant<u32, 0>{}, u0);
        if (u0 + 1u < n_units) consume(std::integral_constant<u32, 1>{}, u0 + 1u);
        if constexpr (RING > 2u) {
            if (u0 + 2u < n_units) consume(std::integral_constant<u32, 2>{}, u0 + 2u);
        }
        if constexpr (RING > 3u) {
            if (u0 + 3u < n_units) consume(std::integral_constant<u32, 3>{}, u0 + 3u);
        }
    }
    stamp(3);
    __syncthreads();
    stamp(4);

    // ---------------------------------------------------------------- 3. epilogue: coefficients x plane sums
    // lane = (piece column, Moebius index c, row quad, row group): the K-split partial sums of its 4 rows are added in unit
    // order (16-byte LDS reads), then the piece columns and the NP terms c of a row -- term c = coef[c] * (c == 0 ? sum(x) :
    // T[c]) -- by fixed DPP trees (adjacent lanes): deterministic
    if (e_rgl < a.RGB) {  // (whole waves except the last one)
        // (c == 0: the pseudo partial sums of sum(x), stride 64 floats per unit, "column" = scale block)
        const float *pp = e_c != 0u ? part + (((size_t)(e_rgl * NCU) * NP1 + (e_c - 1u)) * NCOL + e_col) * 16u + 4u * e_rq
                                    : xpart + (size_t)e_col * 16u + 4u * e_rq;
        const u32 pstride = e_c != 0u ? (u32)NP1 * NCOL * 16u : 64u, pcount = e_c != 0u ? NCU : NC2;
        v4f term = {0.f, 0.f, 0.f, 0.f};
        if constexpr (PSUM) {
            // (one column of real partial sums; sum(x) keeps its 4 block columns: lane c == 0 adds them itself)
            if (e_c == 0u) {
                for (u32 q = 0; q < NC2; q++) {
                    v4f t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (u32 g = 0; g < 4; g++) t += *reinterpret_cast<const v4f *>(xpart + (size_t)q * 64u + g * 16u + 4u * e_rq);
                    term += t;
                }
            } else {
                for (u32 cq = 0; cq < NCU; cq++) term += *reinterpret_cast<const v4f *>(pp + (size_t)cq * pstride);
            }
        } else {
            for (u32 cq = 0; cq < pcount; cq += 4u) {  // (multiples of 4: K >= 2048)
                v4f v[4];
#pragma unroll
                for (u32 k = 0; k < 4; k++) v[k] = *reinterpret_cast<const v4f *>(pp + (size_t)(cq + k) * pstride);
#pragma unroll
                for (u32 k = 0; k < 4; k++) term += v[k];
            }
        }
        asm volatile("" : "+v"(term));
        stamp2(8);
        const v4f cf = *reinterpret_cast<const v4f *>(ctab + (size_t)e_c * (a.RGB * 16u) + e_rgl * 16u + 4u * e_rq);
        float y[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            float f = term[r];
            if constexpr (!PSUM) {
                f += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, f), 0xB1, 0xF, 0xF, false));
                f += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, f), 0x4E, 0xF, 0xF, false));
            }
            f *= cf[r];
            constexpr int D0 = PSUM ? 0xB1 : 0x141, D1 = PSUM ? 0x4E : 0x140;
            f += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, f), D0, 0xF, 0xF, false));
            f += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, f), D1, 0xF, 0xF, false));
            if constexpr (NP >= 8) f += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, f), 0x141, 0xF, 0xF, false));
            if constexpr (NP >= 16) f += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, f), 0x140, 0xF, 0xF, false));
            y[r] = f;
        }
        asm volatile("" : "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]));
        stamp2(9);
        // every lane of the row quad holds the 4 sums
        const u32 row = (rg0 + e_rgl) * 16u + 4u * e_rq;
        if (f_rope) {
            // apply_rotary_pos_emb on fp16 values (inference/model.py:336-341): q_embed = (q * cos) + (rotate_half(q) * sin), three
            // fp16-rounded operations on the GEMV's fp16 outputs; rotated q -> out (reference row order), rotated k and plain v ->
            // the caches at *pos (KVCache.update, model.py:69-79).  A position past the cache writes nothing to the caches.
            if constexpr (!PSUM) {
                const u32 hd = 1u << a.lhd, head = rp_row >> a.lhd, d = rp_row & (hd - 1u);
                const _Float16 lo = (_Float16)(e_col ? y[2] : y[0]), hi = (_Float16)(e_col ? y[3] : y[1]);
                if (e_c == 0u && e_col < 2u && rp_row + (hd >> 1) < a.N) {
                    if (head < a.H + a.Hkv) {
                        const _Float16 c0 = __builtin_bit_cast(_Float16, (uint16_t)rp_cs[0]), c1 = __builtin_bit_cast(_Float16, (uint16_t)rp_cs[1]);
                        const _Float16 s0 = __builtin_bit_cast(_Float16, (uint16_t)rp_cs[2]), s1 = __builtin_bit_cast(_Float16, (uint16_t)rp_cs[3]);
                        const _Float16 olo = (_Float16)(lo * c0) + (_Float16)((-hi) * s0), ohi = (_Float16)(hi * c1) + (_Float16)(lo * s1);
                        if (head < a.H) {
                            gq_store_wt(a.out + rp_row, __builtin_bit_cast(uint16_t, olo));
                            gq_store_wt(a.out + rp_row + (hd >> 1), __builtin_bit_cast(uint16_t, ohi));
                        } else if (rp_pos < a.max_seq) {
                            uint16_t *kp = a.kc + ((size_t)(head - a.H) * a.max_seq + rp_pos) * hd + d;
                            gq_store_wt(kp, __builtin_bit_cast(uint16_t, olo));
                            gq_store_wt(kp + (hd >> 1), __builtin_bit_cast(uint16_t, ohi));
                        }
                    } else if (rp_pos < a.max_seq) {
                        uint16_t *vp = a.vc + ((size_t)(head - a.H - a.Hkv) * a.max_seq + rp_pos) * hd + d;
                        gq_store_wt(vp, __builtin_bit_cast(uint16_t, lo));
                        gq_store_wt(vp + (hd >> 1), __builtin_bit_cast(uint16_t, hi));
                    }
                }
            }
        } else if (f_pairs) {
            // rows (2 i, 2 i + 1) = (gate, up): F.silu(gate) * up on fp16 values -- inference/model.py:266.  Lane (c = 0, col = k)
            // takes pair k of the quad (one silu per lane instead of two on the tail)
            const u32 k = PSUM ? e_c : e_col;  // (PSUM: lanes c = 0, 1 of the quad)
            const _Float16 yg = (_Float16)(k ? y[2] : y[0]), yu = (_Float16)(k ? y[3] : y[1]);
            const float gv = (float)yg;
            const _Float16 o = (_Float16)(gv / (1.0f + __expf(-gv))) * yu;
            if ((PSUM ? true : e_c == 0u) && k < 2u && row + 2u * k + 1u < a.N) gq_store_wt(a.out + (row >> 1) + k, __builtin_bit_cast(uint16_t, o));
        } else if (f_part) {
            // K split over blocks: the fp32 sums of this slice (coefficient terms included: they are linear in x) for the reduce launch
            if (e_c == 0u && e_col == 0u) {
                float *po = a.part_out + (size_t)ksl * a.N + row;
                if (row + 3u < a.N) {
                    gq_store_wt(reinterpret_cast<uint2 *>(po), make_uint2(__builtin_bit_cast(u32, y[0]), __builtin_bit_cast(u32, y[1])));
                    gq_store_wt(reinterpret_cast<uint2 *>(po + 2), make_uint2(__builtin_bit_cast(u32, y[2]), __builtin_bit_cast(u32, y[3])));
                } else {
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        if (row + (u32)r < a.N) gq_store_wt(po + r, y[r]);
                }
            }
        } else if (e_c == 0u && e_col == 0u) {
            uint16_t o[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                _Float16 yh = (_Float16)y[r];
                if (f_resid) yh = __builtin_bit_cast(_Float16, (uint16_t)(rres[r >> 1] >> (16 * (r & 1)))) + yh;
                o[r] = __builtin_bit_cast(uint16_t, yh);
            }
            if (row + 3u < a.N) {
                gq_store_wt(reinterpret_cast<uint2 *>(a.out + row), make_uint2((u32)o[0] | ((u32)o[1] << 16), (u32)o[2] | ((u32)o[3] << 16)));
            } else {
#pragma unroll
                for (int r = 0; r < 4; r++)
                    if (row + (u32)r < a.N) gq_store_wt(a.out + row + r, o[r]);
            }
        }
    }
    stamp(5);
    if (dbg_on && l < 24u) {
        const unsigned long long v = dbgl[l];
        if (l < 8u) a.dbg[w * 8u + l] = v;
        else a.dbg[128u + w * 16u + (l - 8u)] = v;
    }
    if (GQ_STAMPS == 2 && a.dbg && blockIdx.y == 0u) {
        __syncthreads();
        if (tid == 0u) a.dbg[2u * blockIdx.x + 1u] = __builtin_amdgcn_s_memrealtime();
    }
}
template <int BITS, int PRO, int NPU, bool PSUM, int EPI = EPI_ANY>
__global__ void __launch_bounds__(64 * st_waves<BITS>()) ap_stream_kernel(StreamArgs a) {
    ap_stream_body<BITS, PRO, NPU, PSUM, EPI>(a);
}

// ================================================================================================================================
// Round 6: the attention heads INSIDE the wqkv launch (gq_anyprec_gemv_qkv_rope_attn).
//
// The decode step's wqkv GEMV occupies 192 of the 256 CUs (8B: 384 row groups, 2 per block) and ends with the rotated q in memory
// and k / v of the current token in the caches; the attention launch behind it (32 blocks, one per query head) then pays a
// dependent-launch gap (~1.6 us), the memory round trip of q / position / the cached rows (~1.1 us), and only then multiplies
// (4.6 us in the graph for <= 100 positions).  Here the head blocks are blocks [G, G + n_head) of the SAME launch, on CUs the GEMV
// leaves idle: they start with the GEMV, fetch the position and every cached row BELOW it while the GEMV runs (those rows were
// written by earlier launches: ordinary loads), and wait on a device flag for what this launch produces --
//   producers  every GEMV block, behind its epilogue's write-through stores (gq_store_wt: global_store .. sc1 = agent scope) and an
//              s_waitcnt vmcnt(0) + block barrier, adds 1 per row group to the flag word of every query head that reads the group
//              (a q head: its own; a k / v head: the n_head / n_kv_head heads of its group): global_atomic_add .. sc1;
//   consumers  each wave of a head block polls its head's flag with agent-scope loads until it reads 3 * head_dim / 16 (the row
//              groups of its q, k and v heads), then reads q and row *pos of the caches with AGENT-SCOPE loads (sc1: the block's own
//              L2 may hold older copies of those lines -- profiles/r06_flag_handoff.txt: plain loads do see them, sc1 loads never
//              did) and runs the arithmetic of attn_roped_kernel<HD, 1> (decode.hip) operation for operation: same position ->
//              stream assignment, same online softmax per stream, same merge order -- bit-identical outputs.
// One flag LINE per head (GQ_ATTN_FLAG_STRIDE words apart: agent-scope atomics on one line serialise at the memory side, 5.4 us
// instead of 0.7 us until the flag is seen); the head block re-arms its flag (stores 0) before it ends -- the next launch's
// producers start behind this kernel.  The poll is BOUNDED: on expiry the head's output is poisoned (NaN logits), nothing hangs.
// Producers never wait for anything and are dispatched before the head blocks (lower block indices), so a head block never holds
// a CU that a producer of its own launch still needs.  Measured hand-over (tools/ubench/flag_handoff.hip): data in hand 1.1 us
// (median) behind the last producer's stores.
struct AttnFuse {
    const uint16_t *q;       // the GEMV's q_out (rotated queries)
    uint16_t *out;           // attention output fp16 [H * HD]
    u32 *flags;              // [H][GQ_ATTN_FLAG_STRIDE]
    float scale;
    u32 gemv_blocks;         // blocks [0, gemv_blocks) are the GEMV's
    u32 spin_limit;
    unsigned long long *dbg; // GQ_STAMPS builds: s_memrealtime stamps, 8 per block (tools/fuse_timing.py)
};
#define FUSE_STAMP(i)                                                                                                   \
    do {                                                                                                                \
        if (GQ_STAMPS && f.dbg && threadIdx.x == 0) f.dbg[(size_t)blockIdx.x * 8u + (i)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
constexpr int FUSE_ATTN_WAVES = GQ_ATTN_WAVES;

__device__ __forceinline__ u32 ld_flag_sc1(const u32 *p) {
    u32 v;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
// agent-scope 16-byte loads, request and wait in ONE asm statement (an output the compiler sees before the data has landed may be copied)
__device__ __forceinline__ void ld16_sc1(u32x4 &d, const void *p) {
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(d) : "v"(p) : "memory");
}
__device__ __forceinline__ void ld16x2_sc1(u32x4 &d0, u32x4 &d1, const void *p0, const void *p1) {
    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(d0), "=&v"(d1)
                 : "v"(p0), "v"(p1)
                 : "memory");
}

// the GEMV block's signal: every row group of the block to the heads that read it
__device__ __forceinline__ void fuse_signal(const StreamArgs &a, const AttnFuse &f) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's write-through stores are acknowledged
    __syncthreads();                                  // ... and every other wave's
    if (threadIdx.x == 0) {
        const u32 rg0 = blockIdx.x * a.RGB, gq = a.H / a.Hkv, RGt = a.N >> 4;
        for (u32 e = 0; e < a.RGB; e++) {
            const u32 rg = rg0 + e;
            if (rg >= RGt) break;
            const u32 head = rg >> (a.lhd - 4u);
            if (head < a.H) {
                asm volatile("global_atomic_add %0, %1, off sc1" ::"v"(f.flags + (size_t)head * GQ_ATTN_FLAG_STRIDE), "v"(1u) : "memory");
            } else {
                const u32 g = head < a.H + a.Hkv ? head - a.H : head - a.H - a.Hkv;
                for (u32 i = 0; i < gq; i++)
                    asm volatile("global_atomic_add %0, %1, off sc1" ::"v"(f.flags + (size_t)(g * gq + i) * GQ_ATTN_FLAG_STRIDE), "v"(1u) : "memory");
            }
        }
    }
}

// one query head: attn_roped_kernel<HD, 1> (decode.hip) with n_split = 1, its loads re-ordered around the flag
template <int HD>
__device__ __forceinline__ void fuse_attn_head(const StreamArgs &a, const AttnFuse &f, u32 h) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    constexpr u32 NW = FUSE_ATTN_WAVES;
    constexpr int LPP = HD / 8, PPW = 64 / LPP, U = 4;
    constexpr u32 NS = NW * PPW;
    const u32 tid = threadIdx.x, w = tid >> 6, l = tid & 63u;
    if (w >= NW) return;  // (the launch has the GEMV's 16 waves per block; a head uses 8 -- ended waves leave the block's barriers)
    float *sc = reinterpret_cast<float *>(smem);  // [2 * NS] running max / sum of the position streams
    float *red2 = sc + 2u * NS;                   // [NS][HD] partial outputs
    const u32 H = a.H, Hkv = a.Hkv, max_seq = a.max_seq;
    const u32 g = h / (H / Hkv);
    const u32 sub = l / LPP, ld = l % LPP;
    const uint16_t *kcg = a.kc + (size_t)g * max_seq * HD, *vcg = a.vc + (size_t)g * max_seq * HD;
    FUSE_STAMP(0);
    const u32 pos = (u32)a.pos[0];  // (written by an earlier launch)
    const bool past = pos >= max_seq;  // decoding past the cache: the head's output is poisoned (NaN logits), like attn_roped_kernel
    const u32 p1 = past ? 0u : pos + 1u;
    // cached rows below the position: on their way while the GEMV runs (ordinary loads -- earlier launches wrote them)
    uint4 kv[U], vv[U];
    auto request = [&](uint4 (&kd)[U], uint4 (&vd)[U], u32 tb, bool flag_seen) {
        bool own = false;
#pragma unroll
        for (int u = 0; u < U; u++) {
            const u32 t = tb + (u32)u * PPW + sub;
            const bool in = t < pos && !past;
            kd[u] = in ? *reinterpret_cast<const uint4 *>(kcg + (size_t)t * HD + ld * 8) : make_uint4(0, 0, 0, 0);
            vd[u] = in ? *reinterpret_cast<const uint4 *>(vcg + (size_t)t * HD + ld * 8) : make_uint4(0, 0, 0, 0);
            own = own || (t == pos && !past);
        }
        if (flag_seen && own) {  // row *pos was written by THIS launch: agent-scope loads
#pragma unroll
            for (int u = 0; u < U; u++) {
                const u32 t = tb + (u32)u * PPW + sub;
                if (t == pos) {
                    u32x4 k4, v4;
                    ld16x2_sc1(k4, v4, kcg + (size_t)t * HD + ld * 8, vcg + (size_t)t * HD + ld * 8);
                    kd[u] = make_uint4(k4.x, k4.y, k4.z, k4.w);
                    vd[u] = make_uint4(v4.x, v4.y, v4.z, v4.w);
                }
            }
        }
    };
    u32 t0 = w * PPW * U;
    if (t0 < p1) request(kv, vv, t0, false);
    FUSE_STAMP(1);
    // ---- the flag: 3 * HD / 16 row groups (q, k, v heads) of this launch
    const u32 *flag = f.flags + (size_t)h * GQ_ATTN_FLAG_STRIDE;
    const u32 want = 3u * (u32)(HD / 16);
    bool ok = false;
    for (u32 i = 0; i < f.spin_limit; i++) {
        if (__builtin_amdgcn_readfirstlane(ld_flag_sc1(flag)) >= want) {
            ok = true;
            break;
        }
        __builtin_amdgcn_s_sleep(4);
    }
    if (past || !ok) {
        __syncthreads();  // (every wave is through its poll: the flag can be re-armed)
        if (tid < (u32)HD) f.out[(size_t)h * HD + tid] = 0x7e00u;
        if (tid == 0 && ok) gq_store_wt(const_cast<u32 *>(flag), 0u);
        return;
    }
    FUSE_STAMP(2);
    // ---- q and row *pos (agent scope), then attn_roped_kernel's arithmetic
    u32x4 q4;
    ld16_sc1(q4, f.q + (size_t)h * HD + ld * 8);
    if (t0 < p1) {
        bool own = false;
#pragma unroll
        for (int u = 0; u < U; u++) own = own || (t0 + (u32)u * PPW + sub == pos);
        if (own) {
#pragma unroll
            for (int u = 0; u < U; u++) {
                const u32 t = t0 + (u32)u * PPW + sub;
                if (t == pos) {
                    u32x4 k4, v4;
                    ld16x2_sc1(k4, v4, kcg + (size_t)t * HD + ld * 8, vcg + (size_t)t * HD + ld * 8);
                    kv[u] = make_uint4(k4.x, k4.y, k4.z, k4.w);
                    vv[u] = make_uint4(v4.x, v4.y, v4.z, v4.w);
                }
            }
        }
    }
    FUSE_STAMP(3);
    float qreg[8];
    {
        const u32 qw[4] = {q4.x, q4.y, q4.z, q4.w};
#pragma unroll
        for (int e = 0; e < 4; e++) qreg[2 * e] = h2f((uint16_t)(qw[e] & 0xFFFF)), qreg[2 * e + 1] = h2f((uint16_t)(qw[e] >> 16));
    }
    const float scale = f.scale;
    float m_run = -3.0e38f, s_run = 0.f, acc[8];
#pragma unroll
    for (int e = 0; e < 8; e++) acc[e] = 0.f;
    for (; t0 < p1; t0 += NW * PPW * U) {
        float pu[U], vfu[U][8];
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
            float p = 0.f;
#pragma unroll
            for (int e = 0; e < 8; e++) p += qreg[e] * kf[e];
            p += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, p), 0xB1, 0xF, 0xF, false));
            p += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, p), 0x4E, 0xF, 0xF, false));
            p += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, p), 0x141, 0xF, 0xF, false));
            if (LPP == 16) p += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, p), 0x140, 0xF, 0xF, false));
            pu[u] = valid ? p * scale : -3.0e38f;
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
        if (t0 + NW * PPW * U < p1) request(kv, vv, t0 + NW * PPW * U, true);
    }
    FUSE_STAMP(4);
    const u32 stream = w * PPW + sub;
#pragma unroll
    for (int e = 0; e < 8; e++) red2[(size_t)stream * HD + ld * 8 + e] = acc[e];
    if (ld == 0) {
        sc[stream] = m_run;
        sc[NS + stream] = s_run;
    }
    __syncthreads();
    FUSE_STAMP(5);
    if (tid == 0) gq_store_wt(const_cast<u32 *>(flag), 0u);  // (every wave has seen the flag: re-armed for the next launch)
    float *fl = red2 + (size_t)NS * HD;  // [NS] factors, [1] maximum
    const u32 nw_act = min(NW, (p1 + (u32)(PPW * U) - 1u) / (u32)(PPW * U));
    const u32 ng = (nw_act * (u32)PPW + 7u) >> 3;
    if (tid < NS) {
        float M = -3.0e38f;
        for (u32 g8 = 0; g8 < ng; g8++)
#pragma unroll
            for (u32 k = 0; k < 8u; k++) M = fmaxf(M, sc[8u * g8 + k]);
        fl[tid] = __expf(sc[tid] - M);
        if (tid == 0u) fl[NS] = M;
    }
    __syncthreads();
    if (tid < (u32)HD) {
        float o = 0.f, sum = 0.f;
        for (u32 g8 = 0; g8 < ng; g8++)
#pragma unroll
            for (u32 k = 0; k < 8u; k++) {
                const u32 i = 8u * g8 + k;
                const float fi = fl[i];
                sum += sc[NS + i] * fi;
                o += red2[i * HD + tid] * fi;
            }
        f.out[(size_t)h * HD + tid] = __builtin_bit_cast(uint16_t, (_Float16)(o / sum));
    }
    FUSE_STAMP(6);
}

template <int HD>
__global__ void __launch_bounds__(64 * ST_W2) ap_qkv_attn_kernel(StreamArgs a, AttnFuse f) {
    if (blockIdx.x < f.gemv_blocks) {
        FUSE_STAMP(0);
        ap_stream_body<2, PRO_RMSNORM, 1, false, EPI_ROPE>(a);
        FUSE_STAMP(1);
        fuse_signal(a, f);
        FUSE_STAMP(2);
    } else {
        fuse_attn_head<HD>(a, f, blockIdx.x - f.gemv_blocks);
    }
}

struct StreamCfg {
    u32 grid, gy, RGB, NPU, W, img_off;
    bool psum;
    size_t smem;
};

bool pick_stream_cfg(u32 N, u32 K, int bits, StreamCfg &c, u32 rgb_force = 0) {
    c.gy = 1u;
    if (K % 2048u || K > 32768u) return false;  // (the epilogue adds the K-split partial sums four units at a time)
    const u32 nchunks = K / 1024u, NC2 = 2u * nchunks;
    if (NC2 > 64u) return false;  // (meta: 64 entries at a fixed offset)
    const u32 RGt = (N + 15u) / 16u, ncu = (u32)gq_cu_count();
    c.W = bits == 2 ? (u32)ST_W2 : (u32)ST_W34;
    const u32 NCU = NC2 / (u32)ST_NH;
    if (NCU < c.W && (NCU & (NCU - 1u))) return false;  // the waves of a K range split its row groups evenly (shifts in the kernel)
    c.NPU = (NC2 + c.W - 1u) / c.W;
    if (c.NPU == 3u) c.NPU = 4u;
    if (c.NPU > (bits == 2 ? 4u : 2u)) return false;  // (images per builder wave compiled in)
    u32 rgb = rgb_force ? rgb_force : (RGt + ncu - 1u) / ncu;
    if (rgb < 1u) rgb = 1u;
    c.RGB = rgb;
    c.grid = (RGt + rgb - 1u) / rgb;
    const u32 np = 1u << bits, np1 = np - 1u;
    c.img_off = (4096u + 128u + rgb * 16u * np * 4u + 1023u) / 1024u * 1024u;
    const size_t fixed = c.img_off + (size_t)NC2 * 2048u + (size_t)NC2 * HOTCAP * 8u + (size_t)NC2 * 256u;
    const size_t raw = (size_t)rgb * NCU * np1 * 4u * 16u * 4u, lds = 160u * 1024u;
    c.psum = bits > 2 || fixed + raw > lds || (size_t)rgb * 4u * np * 4u > 64u * c.W || gq_env_int("GQ_ST_PSUM", 0);
    if ((size_t)rgb * 4u * np * (c.psum ? 1u : 4u) > 64u * c.W) return false;  // epilogue lanes
    if (rgb * 16u > 64u * c.W) return false;                                    // coefficient lanes
    c.smem = fixed + (c.psum ? raw / 4u : raw);
    return c.smem <= lds;
}

template <int BITS, int PRO, int NPU, bool PSUM, int EPI = EPI_ANY>
int launch_inst(const StreamArgs &a, const StreamCfg &c, hipStream_t s) {
    static GqPerDeviceOnce once;
    auto kern = ap_stream_kernel<BITS, PRO, NPU, PSUM, EPI>;
    GQ_HIP_CHECK(once.max_dynamic_lds(reinterpret_cast<const void *>(kern), (int)(160u * 1024u)));
    hipLaunchKernelGGL(kern, dim3(c.grid, c.gy), dim3(64u * c.W), c.smem, s, a);
    GQ_HIP_CHECK(hipGetLastError());
    return GQ_OK;
}
template <int BITS, int PRO>
int launch_npu(const StreamArgs &a, const StreamCfg &c, hipStream_t s) {
    if constexpr (BITS == 2) {
        if (c.NPU == 4u) {  // K > 16384 (the 70B down projection in one launch): summed parking only, no RMSNorm in front of it
            if constexpr (PRO != PRO_RMSNORM) return c.psum ? launch_inst<BITS, PRO, 4, true>(a, c, s) : GQ_ENOTSUP;
            return GQ_ENOTSUP;
        }
        if constexpr (PRO == PRO_RMSNORM) {
            // the decode step's wqkv launch: the instance with the RoPE epilogue compiled in (GQ_ST_EPI=0: the run-time form; =2: also
            // the pair-epilogue instance for w1w3, which measured SLOWER than the run-time form, 8.85 vs 8.65 us, with a main loop
            // that is instruction for instruction the same -- profiles/r05_compiled_in_epilogues.txt)
            const int epi = gq_env_int("GQ_ST_EPI", 1);
            if (!c.psum && c.NPU == 1u && !a.ssq_in && !a.part_out && !a.resid && epi) {
                if (a.rope && !a.pairs) return launch_inst<BITS, PRO, 1, false, EPI_ROPE>(a, c, s);
                if (a.pairs && !a.rope && epi >= 2) return launch_inst<BITS, PRO, 1, false, EPI_PAIRS>(a, c, s);
            }
        }
        if (!c.psum) return c.NPU == 1u ? launch_inst<BITS, PRO, 1, false>(a, c, s) : launch_inst<BITS, PRO, 2, false>(a, c, s);
    }
    if (c.NPU > 2u) return GQ_ENOTSUP;
    return c.NPU == 1u ? launch_inst<BITS, PRO, 1, true>(a, c, s) : launch_inst<BITS, PRO, 2, true>(a, c, s);
}
template <int BITS>
int launch_pro(const StreamArgs &a, const StreamCfg &c, int pro, hipStream_t s) {
    switch (pro) {
        case PRO_RMSNORM: return launch_npu<BITS, PRO_RMSNORM>(a, c, s);
        case PRO_SILUMUL: return launch_npu<BITS, PRO_SILUMUL>(a, c, s);
        default: return launch_npu<BITS, PRO_NONE>(a, c, s);
    }
}
}  // namespace

unsigned long long *gq_debug_timing_buffer();  // ap_plane.hip (gq_debug_set_timing_buffer)

namespace {
struct KSplit {
    float *part;      // [KS][N] fp32
    u32 KS, kslice;   // K = KS * kslice
};
int stream_launch(const void *x, void *out, const uint32_t *qweight, const void *lut, uint32_t N, uint32_t K, int bits, const void *normw,
                  float eps, const void *resid, int pro, int pairs, const StreamArgs *rope, hipStream_t stream, const KSplit *ksp = nullptr,
                  GqHandover *ho = nullptr, AttnFuse *fuse = nullptr, bool fuse_dry = false) {
    if (bits < 2 || bits > gq_env_int("GQ_ST_MAXBITS", 4)) return GQ_ENOTSUP;
    const uint64_t qbytes = (uint64_t)bits * N * (K / 8u);
    if (qbytes >= 0x7FFFFFFFull) return GQ_ENOTSUP;
    if (((uintptr_t)qweight | (uintptr_t)x | (uintptr_t)normw | (uintptr_t)lut) & 15u) return GQ_ENOTSUP;
    StreamCfg c;
    const u32 Kk = ksp ? ksp->kslice : K;  // activations a block multiplies
    if (ksp) {
        // K slices x row slices ~ one block per CU: row groups per block so that KS * ceil(RGt / rgb) <= CUs
        const u32 RGt = (N + 15u) / 16u, ncu = (u32)gq_cu_count(), rs = ncu / ksp->KS ? ncu / ksp->KS : 1u;
        if (!pick_stream_cfg(N, Kk, bits, c, (RGt + rs - 1u) / rs)) return GQ_ENOTSUP;
        c.gy = ksp->KS;
    } else if (!pick_stream_cfg(N, K, bits, c)) return GQ_ENOTSUP;
    if (rope && c.psum) return GQ_ENOTSUP;
    StreamArgs a{};
    if (rope) a = *rope;
    a.qw = qweight;
    a.lut = (const uint16_t *)lut;
    a.x = (const uint16_t *)x;
    a.out = (uint16_t *)out;
    a.normw = (const uint16_t *)normw;
    a.resid = (const uint16_t *)resid;
    a.N = N;
    a.K = Kk;
    a.Kx = K;
    a.part_out = ksp ? ksp->part : nullptr;
    a.ssq_in = (ho && pro == PRO_RMSNORM && !((uintptr_t)ho->ssq_in & 15u) && gq_env_int("GQ_SSQ_HANDOVER", 1)) ? ho->ssq_in : nullptr;
    if (ho) ho->ssq_consumed = a.ssq_in != nullptr;
    if (ho && ho->dry) return GQ_OK;
    a.wpr_ld = K / 32u;
    a.RGB = c.RGB;
    a.img_off = c.img_off;
    {
        const u32 NCU = 2u * (Kk / 1024u) / (u32)ST_NH;
        a.lw = c.W == 16u ? 4u : 3u;
        a.lq = 31u;
        if (NCU < c.W)
            for (u32 i = 0; i < 5u; i++)
                if ((1u << i) == NCU) a.lq = i;
    }
    a.pairs = pairs ? 1u : 0u;
    a.eps = eps;
    a.dbg = gq_debug_timing_buffer();
    a.dbg_off = (u32)((c.smem + 15u) & ~(size_t)15u);
    if (a.dbg) {
        c.smem = a.dbg_off + 16u * 24u * 8u;
        if (c.smem > 160u * 1024u) a.dbg = nullptr;
    }
#ifndef ST_MAXBITS
#define ST_MAXBITS 4  // bit widths compiled in
#endif
    if (fuse) {
        // the attention heads as extra blocks of this launch (ap_qkv_attn_kernel): the EPI_ROPE instance's configuration, one CU per block
        if (bits != 2 || pro != PRO_RMSNORM || !rope || c.psum || c.NPU != 1u || c.W != (u32)ST_W2 || c.gy != 1u || a.ssq_in || a.part_out || a.resid ||
            a.pairs || c.grid + a.H > (u32)gq_cu_count() || (u32)FUSE_ATTN_WAVES > c.W)
            return GQ_ENOTSUP;
        if (fuse_dry) return GQ_OK;
        const u32 hd = 1u << a.lhd, ns = (u32)FUSE_ATTN_WAVES * 64u / (hd / 8u);
        const size_t asmem = ((size_t)2u * ns + (size_t)ns * hd + ns + 1u) * 4u;
        const size_t smem = c.smem > asmem ? c.smem : asmem;
        fuse->gemv_blocks = c.grid;
        fuse->dbg = GQ_STAMPS ? gq_debug_timing_buffer() : nullptr;
        a.dbg = nullptr;
        if (hd == 128u) {
            static GqPerDeviceOnce once;
            GQ_HIP_CHECK(once.max_dynamic_lds(reinterpret_cast<const void *>(ap_qkv_attn_kernel<128>), (int)(160u * 1024u)));
            hipLaunchKernelGGL(ap_qkv_attn_kernel<128>, dim3(c.grid + a.H), dim3(64u * c.W), smem, stream, a, *fuse);
        } else {
            static GqPerDeviceOnce once;
            GQ_HIP_CHECK(once.max_dynamic_lds(reinterpret_cast<const void *>(ap_qkv_attn_kernel<64>), (int)(160u * 1024u)));
            hipLaunchKernelGGL(ap_qkv_attn_kernel<64>, dim3(c.grid + a.H), dim3(64u * c.W), smem, stream, a, *fuse);
        }
        GQ_HIP_CHECK(hipGetLastError());
        return GQ_OK;
    }
    if (bits == 2) return launch_pro<2>(a, c, pro, stream);
#if ST_MAXBITS >= 3
    if (bits == 3) return launch_pro<3>(a, c, pro, stream);
#endif
#if ST_MAXBITS >= 4
    if (bits == 4) return launch_pro<4>(a, c, pro, stream);
#endif
    return GQ_ENOTSUP;
}
}  // namespace

// ---- rows wider than 16384 activations (the 70B down projection, K = 28672): K split over BLOCKS.  One block multiplying the
// whole row builds 56 unit images for 2 row groups of work (measured: 25.9 us in one launch, 21.8 us as two round-3 launches); a
// block of the split form builds the 4 or 8 images of its K slice and multiplies them with ~15 row groups, the fp32 sums of the
// slices meet in a second, tiny launch (ascending slice order, ONE fp16 rounding like the reference's kernel, anyprec.cu:505-512
// -- the two-launch form rounds twice).  The partial sums live in a caller-supplied workspace (gq_anyprec_gemv_fused_ws).
namespace {
__global__ void __launch_bounds__(256) ap_ksplit_reduce_kernel(const float *part, const uint16_t *resid, uint16_t *out, u32 N, u32 KS) {
    const u32 i4 = blockIdx.x * 256u + threadIdx.x;  // 4 outputs per thread
    if (i4 * 4u >= N) return;
    if (i4 * 4u + 3u < N) {
        uint2 rw = make_uint2(0u, 0u);
        if (resid) rw = *reinterpret_cast<const uint2 *>(resid + 4u * i4);
        float4 p[16];
#pragma unroll
        for (u32 k = 0; k < 16; k++)
            if (k < KS) p[k] = *reinterpret_cast<const float4 *>(part + (size_t)k * N + 4u * i4);
        float4 acc = p[0];
#pragma unroll
        for (u32 k = 1; k < 16; k++)
            if (k < KS) acc = make_float4(acc.x + p[k].x, acc.y + p[k].y, acc.z + p[k].z, acc.w + p[k].w);
        const float y[4] = {acc.x, acc.y, acc.z, acc.w};
        uint16_t o[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            _Float16 yh = (_Float16)y[r];
            if (resid) yh = __builtin_bit_cast(_Float16, (uint16_t)((r >> 1 ? rw.y : rw.x) >> (16 * (r & 1)))) + yh;
            o[r] = __builtin_bit_cast(uint16_t, yh);
        }
        gq_store_wt(reinterpret_cast<uint2 *>(out + 4u * i4), make_uint2((u32)o[0] | ((u32)o[1] << 16), (u32)o[2] | ((u32)o[3] << 16)));
        return;
    }
    for (u32 i = 4u * i4; i < N; i++) {
        float acc = part[i];
        for (u32 k = 1; k < KS; k++) acc += part[(size_t)k * N + i];
        _Float16 yh = (_Float16)acc;
        if (resid) yh = __builtin_bit_cast(_Float16, resid[i]) + yh;
        out[i] = __builtin_bit_cast(uint16_t, yh);
    }
}
// the K slice of the split form: 4096 activations where they divide K (8 images per block), else 2048; 0: shape not served
u32 ksplit_slice(u32 K) {
    const u32 want = (u32)gq_env_int("GQ_ST_KSLICE", 0);
    if (want && want % 2048u == 0u && K % want == 0u && K / want <= 16u && K / want >= 2u) return want;
    if (K % 4096u == 0u && K / 4096u <= 16u) return 4096u;
    if (K % 2048u == 0u && K / 2048u <= 16u) return 2048u;
    return 0u;
}
}  // namespace

size_t gq_stream_ksplit_ws_bytes(uint32_t N, uint32_t K, int bits) {
    if (bits != 2 || K <= 16384u || K > 65536u || !gq_env_int("GQ_ST_KSPLIT", 1)) return 0;
    const u32 ksl = ksplit_slice(K);
    return ksl ? (size_t)(K / ksl) * N * 4u : 0;
}
// plain / residual epilogue, no prologue; GQ_ENOTSUP when the shape is not served or the workspace is too small
int gq_stream_gemv_ksplit(const void *x, void *out, const uint32_t *qweight, const void *lut, uint32_t N, uint32_t K, int bits,
                          const void *resid, void *ws, size_t ws_bytes, hipStream_t stream) {
    const size_t need = gq_stream_ksplit_ws_bytes(N, K, bits);
    if (!need || !ws || ws_bytes < need || ((uintptr_t)ws & 15u) || ((uintptr_t)out & 7u) || ((uintptr_t)resid & 7u)) return GQ_ENOTSUP;
    KSplit ks{(float *)ws, 0u, ksplit_slice(K)};
    ks.KS = K / ks.kslice;
    const int rc = stream_launch(x, out, qweight, lut, N, K, bits, nullptr, 0.f, nullptr, PRO_NONE, 0, nullptr, stream, &ks);
    if (rc != GQ_OK) return rc;
    hipLaunchKernelGGL(ap_ksplit_reduce_kernel, dim3((N / 4u + 256u) / 256u), dim3(256), 0, stream, (const float *)ws, (const uint16_t *)resid,
                       (uint16_t *)out, N, ks.KS);
    GQ_HIP_CHECK(hipGetLastError());
    return GQ_OK;
}

// returns GQ_ENOTSUP when the shape is not served by this kernel (the caller goes on to ap_plane.hip / the exact kernels)
int gq_stream_gemv_try(const void *x, void *out, const uint32_t *qweight, const void *lut, uint32_t M, uint32_t N, uint32_t K, int bits,
                       const void *normw, float eps, const void *resid, int pro, int pairs, hipStream_t stream, GqHandover *ho) {
    if (M != 1u) return GQ_ENOTSUP;
    return stream_launch(x, out, qweight, lut, N, K, bits, normw, eps, resid, pro, pairs, nullptr, stream, nullptr, ho);
}

// The fused q / k / v projection of a decode step with RoPE and the KV-cache write in its epilogue (include/gq_hip.h).
bool gq_ap_exact_mode();  // ap_gemv.hip
extern "C" int gq_anyprec_qkv_rope_supported(uint32_t N, uint32_t K, int bits, uint32_t head_dim) {
    StreamCfg c;
    return bits == 2 && (head_dim == 64u || head_dim == 128u) && N % head_dim == 0u && gq_env_int("GQ_QKV_ROPE", 1) &&
                   !gq_ap_exact_mode() && pick_stream_cfg(N, K, bits, c) && !c.psum
               ? 1 : 0;
}
extern "C" int gq_anyprec_gemv_qkv_rope(const void *x, void *q_out, const uint32_t *qweight, const void *lut, uint32_t N, uint32_t K, int bits,
                                        const void *norm_weight, float eps, const int *pos, const void *cos_table, const void *sin_table,
                                        void *k_cache, void *v_cache, uint32_t n_head, uint32_t n_kv_head, uint32_t head_dim, uint32_t max_seq,
                                        void *stream) {
    return gq_anyprec_gemv_qkv_rope_ho(x, q_out, qweight, lut, N, K, bits, norm_weight, eps, pos, cos_table, sin_table, k_cache, v_cache, n_head,
                                       n_kv_head, head_dim, max_seq, nullptr, stream);
}
extern "C" int gq_anyprec_gemv_qkv_rope_ho(const void *x, void *q_out, const uint32_t *qweight, const void *lut, uint32_t N, uint32_t K, int bits,
                                           const void *norm_weight, float eps, const int *pos, const void *cos_table, const void *sin_table,
                                           void *k_cache, void *v_cache, uint32_t n_head, uint32_t n_kv_head, uint32_t head_dim,
                                           uint32_t max_seq, const float *ssq_in, void *stream) {
    if (!x || !q_out || !qweight || !lut || !norm_weight || !pos || !cos_table || !sin_table || !k_cache || !v_cache)
        return gq_fail(GQ_EINVAL, "null pointer argument.");
    if (N != (n_head + 2u * n_kv_head) * head_dim) return gq_fail(GQ_EINVAL, "N must be (n_head + 2 n_kv_head) * head_dim.");
    if (!gq_anyprec_qkv_rope_supported(N, K, bits, head_dim)) return gq_fail(GQ_ENOTSUP, "gq_anyprec_gemv_qkv_rope: shape / bit width not served.");
    StreamArgs r{};
    r.pos = pos;
    r.cos_t = (const uint16_t *)cos_table;
    r.sin_t = (const uint16_t *)sin_table;
    r.kc = (uint16_t *)k_cache;
    r.vc = (uint16_t *)v_cache;
    r.rope = 1u;
    r.H = n_head;
    r.Hkv = n_kv_head;
    r.lhd = head_dim == 128u ? 7u : 6u;
    r.max_seq = max_seq;
    GqHandover ho;
    ho.ssq_in = ssq_in;
    const int rc = stream_launch(x, q_out, qweight, lut, N, K, bits, norm_weight, eps, nullptr, PRO_RMSNORM, 0, &r, (hipStream_t)stream, nullptr, &ho);
    return rc == GQ_ENOTSUP ? gq_fail(GQ_ENOTSUP, "gq_anyprec_gemv_qkv_rope: shape / bit width not served.") : rc;
}

// ---- round 6: the same launch with the attention heads as extra blocks (see ap_qkv_attn_kernel)
namespace {
int qkv_rope_attn_launch(const void *x, void *q_out, const uint32_t *qweight, const void *lut, uint32_t N, uint32_t K, int bits, const void *norm_weight,
                         float eps, const int *pos, const void *cos_table, const void *sin_table, void *k_cache, void *v_cache, uint32_t n_head,
                         uint32_t n_kv_head, uint32_t head_dim, uint32_t max_seq, void *attn_out, float scale, uint32_t *flags, void *stream, bool dry) {
    StreamArgs r{};
    r.pos = pos;
    r.cos_t = (const uint16_t *)cos_table;
    r.sin_t = (const uint16_t *)sin_table;
    r.kc = (uint16_t *)k_cache;
    r.vc = (uint16_t *)v_cache;
    r.rope = 1u;
    r.H = n_head;
    r.Hkv = n_kv_head;
    r.lhd = head_dim == 128u ? 7u : 6u;
    r.max_seq = max_seq;
    AttnFuse f{};
    f.q = (const uint16_t *)q_out;
    f.out = (uint16_t *)attn_out;
    f.flags = flags;
    f.scale = scale;
    f.spin_limit = (u32)gq_env_int("GQ_QKV_ATTN_SPINS", 1 << 21);
    return stream_launch(x, q_out, qweight, lut, N, K, bits, norm_weight, eps, nullptr, PRO_RMSNORM, 0, &r, (hipStream_t)stream, nullptr, nullptr, &f, dry);
}
}  // namespace
extern "C" int gq_anyprec_qkv_rope_attn_supported(uint32_t N, uint32_t K, int bits, uint32_t head_dim, uint32_t n_head, uint32_t n_kv_head) {
    // OFF by default (GQ_QKV_ATTN=1 turns it on): correct and bit-identical, and no faster -- the hand-over inside the launch (signal 0.6 us:
    // the producers' write-through stores acknowledged + block barrier + atomics; flag seen 0.4-1.0 us later; q / row *pos in hand 0.3-0.6 us
    // after that) costs what the kernel boundary it removes costs: 8B decode 904-905 vs 909.5 tokens/s (profiles/r06_attention_in_wqkv_launch.txt)
    if (!gq_env_int("GQ_QKV_ATTN", 0) || n_kv_head == 0u || n_head % n_kv_head || N != (n_head + 2u * n_kv_head) * head_dim) return 0;
    if (!gq_anyprec_qkv_rope_supported(N, K, bits, head_dim)) return 0;
    return qkv_rope_attn_launch(nullptr, nullptr, nullptr, nullptr, N, K, bits, nullptr, 0.f, nullptr, nullptr, nullptr, nullptr, nullptr, n_head, n_kv_head,
                                head_dim, 1u, nullptr, 0.f, nullptr, nullptr, true) == GQ_OK
               ? 1 : 0;
}
extern "C" int gq_anyprec_gemv_qkv_rope_attn(const void *x, void *q_out, const uint32_t *qweight, const void *lut, uint32_t N, uint32_t K, int bits,
                                             const void *norm_weight, float eps, const int *pos, const void *cos_table, const void *sin_table,
                                             void *k_cache, void *v_cache, uint32_t n_head, uint32_t n_kv_head, uint32_t head_dim, uint32_t max_seq,
                                             void *attn_out, float scale, uint32_t *flags, void *stream) {
    if (!x || !q_out || !qweight || !lut || !norm_weight || !pos || !cos_table || !sin_table || !k_cache || !v_cache || !attn_out || !flags)
        return gq_fail(GQ_EINVAL, "null pointer argument.");
    if (((uintptr_t)q_out | (uintptr_t)k_cache | (uintptr_t)v_cache) & 15u) return gq_fail(GQ_EINVAL, "buffers must be 16-byte aligned.");
    if (((uintptr_t)flags & 127u)) return gq_fail(GQ_EINVAL, "gq_anyprec_gemv_qkv_rope_attn: the flag words must be 128-byte aligned.");
    if (!gq_anyprec_qkv_rope_attn_supported(N, K, bits, head_dim, n_head, n_kv_head))
        return gq_fail(GQ_ENOTSUP, "gq_anyprec_gemv_qkv_rope_attn: shape / bit width / head geometry not served.");
    const int rc = qkv_rope_attn_launch(x, q_out, qweight, lut, N, K, bits, norm_weight, eps, pos, cos_table, sin_table, k_cache, v_cache, n_head, n_kv_head,
                                        head_dim, max_seq, attn_out, scale, flags, stream, false);
    return rc == GQ_ENOTSUP ? gq_fail(GQ_ENOTSUP, "gq_anyprec_gemv_qkv_rope_attn: shape / bit width / head geometry not served.") : rc;
}
