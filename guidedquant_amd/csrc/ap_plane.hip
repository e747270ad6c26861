// ap_plane.hip -- "fast mode" Any-Precision GEMV for gfx950: binary bit-plane GEMVs on the matrix cores.
//
// See plane_core.h for the algorithm and the measured hardware facts.  What runs where:
//   HBM    : the stored bit-planes, whole 128-byte lines (one row x one plane x one 1024-weight chunk), fetched by
//            direct-to-LDS loads (buffer_load_dwordx4 ... lds) into per-wave rings of A tiles; the late half of the
//            waves requests its tiles in its first instructions, the early half after it has built the B image.
//   VALU   : one v_and_b32 per 8 weights per plane-subset (nibble bit -> FP4 {0, 0.5 / 1 / 2}); the operand of a subset
//            is the AND of its planes' masked words (depth-first over the subset lattice).
//   MFMA   : v_mfma_scale_f32_16x16x128_f8f6f4, A = FP4 single-bit patterns (4 registers, the E8M0 scale cancels the
//            pattern value), B = 4 exact bf8 pieces of x * 2^k in columns 0..3, fp32 accumulate:
//            8 MFMAs per (chunk, plane-subset).
//   LDS    : per-wave rings of A tiles (plane_core.h atile_unit swizzle), the B image (activation pieces, 4*K bytes)
//            built once per block, the LUT rows of the block, partial sums of K-split items.
// Epilogue : y = coef[0] * sum(x) + sum_S coef[S] * T[S]  (Moebius coefficients of the row's LUT), fp32 -> fp16;
//            optional residual add or SiLU(gate) * up over row pairs.
//
// This path is NOT bit-identical to the reference's fp16-accumulated kernel (anyprec.cu:372-542): it is closer
// to the exact product than the reference is (products exact, fp32 accumulation).  The bit-exact path is
// ap_gemv.hip; gq_set_ap_mode() / GQ_AP_MODE choose.
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

struct PlaneArgs {
    const u32 *qw;
    const uint16_t *lut;
    const uint16_t *x;
    uint16_t *out;
    const uint16_t *normw;
    const uint16_t *resid;
    u32 N, K;    // K = the columns served by this launch (a slice of the row when the matrix is split along K)
    u32 wpr_ld;  // plane words per stored row (row stride), word0 = first plane word of the slice (multiple of 32)
    u32 word0;
    u32 x_ld;    // activations per batch row in memory
    u32 RGB;     // row groups (16 rows) per block
    u32 log2CS;  // a row group's chunks are split over 2^log2CS wave items
    u32 cpi;     // chunks per item
    u32 S;       // LDS ring slots (steps) per wave
    u32 rawx;    // the activations are staged through LDS (coalesced 16-byte loads into the early waves' still unused ring slots)
    u32 himg;    // the late waves build the second half of the image passes (needs rawx; they have little to request)
    u32 pairs;   // GQ_EPI_SILU_PAIRS: rows are (gate, up) pairs, out[i] = silu(y[2i]) * y[2i+1]
    u32 MB, M;   // batch rows per block (one-pass multi-row kernel instances; else 1) and in total
    u32 xflags;  // ablation experiments (GQ_PL_XFLAGS): 1 no MFMA work, 2 no steps at all, 8 no activation loads, 16 no LUT, 32 empty kernel,
                 // 256 late waves request only their first item up front, 1024 no plane loads (the MFMA phase runs on stale LDS)
    float eps;
    unsigned long long *dbg;  // optional: per-wave phase timestamps of block 0 (tools/phase_timing.py)
    float *ssq_out;           // statistics hand-over (gq_hip.h GQ_SSQ_SLOTS): block b leaves the sum of squares of its fp16 outputs in slot b
};

// The ablation switches of GQ_PL_XFLAGS (a.xflags) are compiled in only with -DGQ_ABLATION=1: in the shipped library every test of
// them would be a scalar compare and a branch on the prologue's critical path (like the phase stamps, gq_internal.h GQ_STAMPS).
#ifndef GQ_ABLATION
#define GQ_ABLATION 0
#endif
#define PL_XF(bit) (GQ_ABLATION && (a.xflags & (bit)))
#ifndef PL_SSQ_MODE
#define PL_SSQ_MODE 2  // statistics hand-over slots: plain stores, slots grouped by XCD (see plane_epilogue)
#endif
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
typedef uint16_t us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ h2v u2h2(u32 u) { return __builtin_bit_cast(h2v, u); }
__device__ __forceinline__ u32 h22u(h2v h) { return __builtin_bit_cast(u32, h); }

// The scaled MFMA with A = FP4 (4 registers; the upper half of the builtin's 8-register vector is ignored for cbsz = 4),
// B = BF8.  Through the BUILTIN, not inline asm: the compiler must see the instruction to insert the MFMA hazard wait
// states.  An asm version with a tied accumulator was tried (the register allocator otherwise lets vDst != SrcC wander):
// no faster, and unsafe -- the compiler copied accumulators between registers right behind an MFMA it could not see.
__device__ __forceinline__ void mfma_f4_bf8(v4f &acc, v4i a, v8i b, int scale_a, int scale_b) {
    const v8i a8 = __builtin_shufflevector(a, a, 0, 1, 2, 3, -1, -1, -1, -1);
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b, acc, 4, 1, 0, scale_a, 0, scale_b);
}

// --- hand-scheduled vector memory.  The compiler's waitcnt pass treats an LDS-DMA load as "may alias every later
// LDS access" and drains vmcnt before the first ds_read / barrier, which would serialise the prologue behind the
// whole plane stream.  So the plane loads and the activation loads that must overtake them are issued from inline
// asm (invisible to that pass) and waited for with explicit s_waitcnt vmcnt(n) (vector memory returns in order).
__device__ __forceinline__ void dma16(u32x4 rsrc, u32 lds_base, u32 voff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen nt lds" ::"s"(lds_base), "v"(voff), "s"(rsrc)
                 : "memory");  // m0 is not otherwise used in this kernel (no compiler-generated LDS-DMA / movrel)
}
__device__ __forceinline__ void dma16s(u32x4 rsrc, u32 lds_base, u32 voff, u32 soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen nt lds" ::"s"(lds_base), "v"(voff), "s"(rsrc), "s"(soff)
                 : "memory");
}
__device__ __forceinline__ u32 bload16s(u32x4 rsrc, u32 voff, u32 soff) {  // zero-extended 16-bit load
    u32 r;
    asm volatile("buffer_load_ushort %0, %1, %2, %3 offen" : "=v"(r) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
    return r;
}
__device__ __forceinline__ u32x4 bload128(u32x4 rsrc, u32 voff, u32 soff) {
    u32x4 r;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(r) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
    return r;
}
__device__ __forceinline__ void tie128(u32x4 &r) { asm volatile("" : "+v"(r)); }
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

// software barrier among a subset of the waves (LDS counter): waves that may sit in the vector-memory issue queue
// (issuing blocks once the CU's memory pipe is full) must not hold up a hardware barrier
__device__ __forceinline__ void arrive(u32 *ctr, u32 lane) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void wait_count(const u32 *ctr, u32 target) {
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// ---- pieces shared by the two kernels below
// one tile (16 rows x one chunk x all planes) from its ring slot: lane (row r, lane group kb) gets words 8 kb .. 8 kb + 7
template <int BITS>
__device__ __forceinline__ void load_tile(u32 (&Wd)[BITS][8], const unsigned char *slot, u32 offA0, u32 offA1) {
#pragma unroll
    for (int p = 0; p < BITS; p++) {
        const uint4 a0 = *reinterpret_cast<const uint4 *>(slot + p * 2048 + offA0);
        const uint4 a1 = *reinterpret_cast<const uint4 *>(slot + p * 2048 + offA1);
        Wd[p][0] = a0.x, Wd[p][1] = a0.y, Wd[p][2] = a0.z, Wd[p][3] = a0.w;
        Wd[p][4] = a1.x, Wd[p][5] = a1.y, Wd[p][6] = a1.z, Wd[p][7] = a1.w;
    }
}

// the 2^BITS - 1 MFMAs of one (nibble bit NB, word half HH) block of a step: FP4 A operands of the plane subsets, built on the
// fly: the mask commutes with AND, so a subset's operand is the AND of its planes' masked words; a depth-first walk over the
// subset lattice keeps only one partial product per level alive (no register-resident AND words: 4-bit would need 120 of
// them).  Plane p holds code bit BITS-1-p; subset index cm = OR of the code bits.
template <int BITS, int NB, int HH>
__device__ __forceinline__ void mfma_bh(v4f (&acc)[(1 << BITS) - 1], const u32 (&Wd)[BITS][8], v8i Bv, int sb) {
    v4i Mp[BITS];
#pragma unroll
    for (int p = 0; p < BITS; p++)
#pragma unroll
        for (int v = 0; v < 4; v++) Mp[p][v] = (int)extract4(Wd[p][4 * HH + v], NB);
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

// the 8 x (2^BITS - 1) MFMAs of one step.  bbase = this lane's 16 bytes of the chunk's (b = 0, h = 0) image block,
// bstep = distance of the next (b, h) block, bhalf = distance of the lane's second 16 bytes (k + 64)
template <int BITS>
__device__ __forceinline__ void mfma_chunk(v4f (&acc)[(1 << BITS) - 1], const u32 (&Wd)[BITS][8], const unsigned char *bbase, u32 bstep,
                                           u32 bhalf, int sb) {
    // B operand (activation pieces) double-buffered over the 8 (nibble bit b, word half h) MFMAs per subset
    uint4 bn0 = *reinterpret_cast<const uint4 *>(bbase);
    uint4 bn1 = *reinterpret_cast<const uint4 *>(bbase + bhalf);
    auto one = [&](auto BH) {
        constexpr int bh = decltype(BH)::value;
        const uint4 b0 = bn0, b1 = bn1;
        if (bh < 7) {
            bn0 = *reinterpret_cast<const uint4 *>(bbase + (u32)(bh + 1) * bstep);
            bn1 = *reinterpret_cast<const uint4 *>(bbase + (u32)(bh + 1) * bstep + bhalf);
        }
        // keep the loads of the next MFMA group ahead of this one (the scheduler otherwise sinks them behind the
        // MFMAs into a single B buffer and exposes the LDS latency 8 times per step)
        __builtin_amdgcn_sched_barrier(0);
        v8i Bv = {(int)b0.x, (int)b0.y, (int)b0.z, (int)b0.w, (int)b1.x, (int)b1.y, (int)b1.z, (int)b1.w};
        mfma_bh<BITS, (bh >> 1), (bh & 1)>(acc, Wd, Bv, sb);
    };
    one(std::integral_constant<int, 0>{});
    one(std::integral_constant<int, 1>{});
    one(std::integral_constant<int, 2>{});
    one(std::integral_constant<int, 3>{});
    one(std::integral_constant<int, 4>{});
    one(std::integral_constant<int, 5>{});
    one(std::integral_constant<int, 6>{});
    one(std::integral_constant<int, 7>{});
}

// ---- extracted ("hot") activations.  The matrix cores align the 128 products of a dot-product group to the largest exponent
// and keep a window of ~17 bits (tools/plane_dynrange_probe.py): next to a channel 2^10 .. 2^14 times larger than the rest
// (the massive activations of Llama hidden states) the group mates would lose their low pieces.  So every element with
// |x| > tau = 64 x the vector's MEAN magnitude is taken out of the image and appended to a short list -- Markov: fewer than K / 64
// such elements exist, the list cannot overflow; the mean (unlike the rms) is hardly moved by the largest channels, so smaller
// massive channels are not shielded by larger ones: a channel is extracted as soon as it holds 1 / 64 of the vector's total
// magnitude K * mean --; the wave that multiplies the element's chunk adds the list
// entries of that chunk with one extra MFMA set per (b, h) block whose only non-zero products are the extracted ones: exact
// products, fp32 accumulation, no alignment loss.  Who detects: in the shared-image kernel the idle late waves (scanners) or,
// without a staged copy, the early waves from their own maxima; the image builders write every element, and the pieces of the
// listed ones are cleared once the image is complete (one more barrier, only then).  The common case -- nothing above the
// threshold; 64 x the mean (51 sigma of a Gaussian) is far outside what SiLU(gate) * up or normalised hidden states produce by
// chance, and what stays in the image (<= 2^6 x the typical element) is below the ratio where the window starts to matter -- costs the early
// waves one more sum in the statistics pass and every wave one scalar compare per step.
//   entry = {key = row << 16 | chunk << 10 | (b * 2 + h) << 7 | k,  the 4 bf8 pieces of x * 2^ksh (byte p = piece p)}
struct HotEnt {
    u32 key, pieces;
};
typedef __attribute__((address_space(3))) u32 lds_u32;
__device__ __forceinline__ void hot_append_inl(u32 *cnt_, HotEnt *list_, u32 cap, u32 key, uint16_t xh, uint16_t kh16) {
    // explicit LDS pointers: through generic ones the counter becomes a FLAT atomic (vmcnt: the wave's plane stream drains)
    lds_u32 *cnt = (lds_u32 *)cnt_, *list = (lds_u32 *)list_;
    const u32 idx = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (idx >= cap) return;  // unreachable (Chebyshev bound), kept as a guard against memory corruption
    _Float16 rem = __builtin_bit_cast(_Float16, xh) * __builtin_bit_cast(_Float16, kh16);
    u32 pieces = 0;
#pragma unroll
    for (u32 p = 0; p < 4; p++) {
        const uint16_t pb = __builtin_bit_cast(uint16_t, rem) & 0xFF00u;
        pieces |= (u32)(pb >> 8) << (8u * p);
        rem = rem - __builtin_bit_cast(_Float16, pb);
    }
    list[2u * idx] = key;
    list[2u * idx + 1u] = pieces;
}
// out of line where the caller's loop is not unrolled (a call inside code that keeps register arrays alive costs spills)
__device__ __attribute__((noinline)) void hot_append(u32 *cnt, HotEnt *list, u32 cap, u32 key, uint16_t xh, uint16_t kh16) {
    hot_append_inl(cnt, list, cap, key, xh, kh16);
}
// the extracted elements of chunk `ckey` (wave-uniform), block by block in ascending (b, h) order: deterministic whatever
// the order of the list
template <int BITS>
__device__ __forceinline__ void hot_step(v4f (&acc)[(1 << BITS) - 1], const u32 (&Wd)[BITS][8], const HotEnt *list, u32 nhot, u32 ckey,
                                         int sbh, u32 col, u32 kb) {
    u32 mask = 0;
    for (u32 e = 0; e < nhot; e++) {
        const u32 key = __builtin_amdgcn_readfirstlane(list[e].key);
        if (((key >> 10) & 63u) == ckey) mask |= 1u << ((key >> 7) & 7u);
    }
    while (mask) {
        const u32 bh = (u32)__builtin_ctz(mask);
        mask &= mask - 1u;
        u32 B[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
        for (u32 e = 0; e < nhot; e++) {
            const u32 key = __builtin_amdgcn_readfirstlane(list[e].key);
            if (((key >> 7) & 511u) != ((ckey << 3) | bh)) continue;
            const u32 pieces = __builtin_amdgcn_readfirstlane(list[e].pieces);
            // lane (col = piece, kb) holds k = 64 (j / 16) + 16 kb + j % 16 as byte j of its 8 registers
            const u32 k = key & 127u, j = ((k >> 6) << 4) | (k & 15u);
            const bool mine = (col >> 2) == (key >> 16) && ((k >> 4) & 3u) == kb;  // (key >> 16: batch row of the multi-row instances)
            const u32 val = mine ? ((pieces >> (8u * (col & 3u))) & 0xFFu) << (8u * (j & 3u)) : 0u;
#pragma unroll
            for (u32 r = 0; r < 8; r++) B[r] |= (j >> 2) == r ? val : 0u;
        }
        const v8i Bv = {(int)B[0], (int)B[1], (int)B[2], (int)B[3], (int)B[4], (int)B[5], (int)B[6], (int)B[7]};
        // one copy of the subset walk with run-time (b, h): eight specialised copies cost > 100 registers of live ranges
        const u32 nb = bh >> 1, hh = bh & 1u;
        const u32 msk = nb == 3u ? 0x44444444u : 0x11111111u << nb, sh = nb == 3u ? 1u : 0u;
        v4i Mp[BITS];
#pragma unroll
        for (int p = 0; p < BITS; p++)
#pragma unroll
            for (int v = 0; v < 4; v++) {
                // (through opaque copies: a select of two array elements becomes a dynamically indexed -- scratch -- array)
                u32 lo = Wd[p][v], hi = Wd[p][4 + v];
                asm("" : "+v"(lo), "+v"(hi));
                Mp[p][v] = (int)(((hh ? hi : lo) >> sh) & msk);
            }
        const int sa = nb == 0u ? 128 : (nb == 1u ? 127 : 126);
#pragma unroll
        for (int cm = 1; cm < (1 << BITS); cm++) {
            v4i A = {-1, -1, -1, -1};
#pragma unroll
            for (int p = 0; p < BITS; p++)
                if (cm & (1 << (BITS - 1 - p))) A &= Mp[p];
            mfma_f4_bf8(acc[cm - 1], A, Bv, sa, sbh);
        }
    }
}
// threshold of the extraction as an fp16 bit pattern (rounded up): 64 x the mean magnitude, 1 % above it for the roundings of the
// estimate (sum1 = sum |x| over K elements)
#define GQ_HOT_T 64.0f
#ifndef HOT_ABL
#define HOT_ABL 0  // ablation experiments (build-time): 1 no threshold, 2 no detection, 4 no extra MFMA sets
#endif
__device__ __forceinline__ u32 hot_tau_bits(float sum1, float K) {
#if HOT_ABL & 1
    return 0x7C00u;
#endif
    // the hardware reciprocal (1 ulp) is plenty next to the 1 % margin
    const float tau = GQ_HOT_T * 1.01f * sum1 * __builtin_amdgcn_rcpf(K);
    if (!(tau < 65000.f)) return 0x7C00u;  // nothing exceeds +inf: no extraction
    return (u32)__builtin_bit_cast(uint16_t, (_Float16)tau) + 1u;
}

// item done: add the 4 piece columns (lanes col = 0..3 of each 16-lane group), park 16 x NP1 sums in LDS
// (pitem[subset][row]: lane (col 0, kb) owns rows 4kb..4kb+3) and clear the accumulators
template <int NP1>
__device__ __forceinline__ void park_item(v4f (&acc)[NP1], float *pitem, u32 col, u32 kb, u32 MB = 1u) {
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
        // (batch row mm = col / 4 of the multi-row instances: pitem[row][subset][16])
        if ((col & 3u) == 0u && (col >> 2) < MB) *reinterpret_cast<v4f *>(pitem + ((size_t)(col >> 2) * NP1 + cm) * 16u + 4u * kb) = v;
        acc[cm] = (v4f){0.f, 0.f, 0.f, 0.f};
    }
}

__device__ __forceinline__ void store_out16(uint16_t *p, uint16_t v) { gq_store_wt(p, v); }  // hand-off store (gq_internal.h)

// coefficients x plane sums.  lane = (output row, Moebius index c): term c = coef[c] * (c == 0 ? sum(x) : T[c]); the NP
// terms of a row sit in NP adjacent lanes and are added by a fixed DPP tree (deterministic order)
// SPEC: the decode step's form compiled in (plain / residual epilogue, no statistics hand-over) -- see ap_plane_local_kernel
template <int BITS, bool SPEC = false, bool NOSSQ = false>
__device__ __forceinline__ void plane_epilogue(const PlaneArgs &a, const u32 *lutl, const float *part, float X, u32 rg0, u32 m, u32 tid,
                                               u32 T, u32 PS /* floats between the K-split items of a row group */) {
    constexpr int NP = 1 << BITS;
    const bool f_pairs = !SPEC && a.pairs;
    float *const f_ssq = (SPEC || NOSSQ) ? nullptr : a.ssq_out;  // (NOSSQ: pairs stay a run-time flag)
    const u32 CS = 1u << a.log2CS;
    for (u32 e = tid; e < a.RGB * 16u * (u32)NP; e += T) {
        const u32 i = e / (u32)NP, c = e % (u32)NP;
        const u32 rgl = i >> 4, rr = i & 15u;
        const u32 row = (rg0 + rgl) * 16u + rr;
        float f[NP];
#pragma unroll
        for (int cc = 0; cc < NP / 2; cc++) {
            const u32 lw = lutl[(size_t)i * (NP / 2) + cc];
            f[2 * cc] = h2f((uint16_t)(lw & 0xFFFF));
            f[2 * cc + 1] = h2f((uint16_t)(lw >> 16));
        }
        moebius<BITS>(f);
        float coef = f[0];
#pragma unroll
        for (int cc = 1; cc < NP; cc++) coef = c == (u32)cc ? f[cc] : coef;
        float term = X;
        if (c != 0u) {
            // the K-split partial sums, added in item order; all loads of a batch are in flight together (a plain loop
            // pays the LDS latency once per partial: 2000 cycles for 16 of them)
            term = 0.f;
            const float *pp = part + ((size_t)(rgl << a.log2CS)) * PS + (c - 1u) * 16u + rr;
            if (CS >= 8u) {
                for (u32 cs = 0; cs < CS; cs += 8u) {
                    float v[8];
#pragma unroll
                    for (u32 k = 0; k < 8; k++) v[k] = pp[(size_t)(cs + k) * PS];
#pragma unroll
                    for (u32 k = 0; k < 8; k++) term += v[k];
                }
            } else {
                float v[4];
#pragma unroll
                for (u32 k = 0; k < 4; k++) v[k] = k < CS ? pp[(size_t)k * PS] : 0.f;
#pragma unroll
                for (u32 k = 0; k < 4; k++) term += v[k];
            }
        }
        float y = coef * term;
        y += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, y), 0xB1, 0xF, 0xF, false));
        y += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, y), 0x4E, 0xF, 0xF, false));
        if constexpr (NP >= 8) y += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, y), 0x141, 0xF, 0xF, false));
        if constexpr (NP >= 16) y += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, y), 0x140, 0xF, 0xF, false));
        _Float16 yh = (_Float16)y;
        // hand-off stores: write-through, one 2-byte store per output (gq_internal.h; two neighbouring outputs per 4-byte store
        // measured slower: 831 vs 839 tokens/s, plain stores 825)
        if (f_pairs) {
            // the partner row of the pair sits NP lanes away: F.silu(gate) * up on fp16 values -- inference/model.py:266
            const _Float16 yo = __builtin_bit_cast(_Float16, (uint16_t)__shfl_xor((int)__builtin_bit_cast(uint16_t, yh), NP));
            if (c == 0u && !(rr & 1u) && row + 1u < a.N) {
                const float gv = (float)yh;
                const _Float16 o = (_Float16)(gv / (1.0f + __expf(-gv))) * yo;
                store_out16(a.out + (size_t)m * (a.N >> 1) + (row >> 1), __builtin_bit_cast(uint16_t, o));
            }
        } else if (c == 0u && row < a.N) {
            // (requesting the residual elements at kernel start -- LDS-DMA by wave 0 -- was measured: no gain, the load is an L2 hit)
            if (a.resid) yh = __builtin_bit_cast(_Float16, a.resid[(size_t)m * a.N + row]) + yh;
            store_out16(a.out + (size_t)m * a.N + row, __builtin_bit_cast(uint16_t, yh));
        }
        if (f_ssq) {
            // hand-over to the RMSNorm prologue of the NEXT launch (include/gq_hip.h, GQ_SSQ_SLOTS): the sum of squares of the fp16
            // values just stored, one slot per epilogue wave (whole waves: 16 RGB 2^b is a multiple of 64; host: one pass of this
            // loop, M = 1, plain / residual epilogue).  Fixed DPP tree: deterministic.
            const float v = (c == 0u && row < a.N) ? (float)yh * (float)yh : 0.f;
            const float tot = wave_reduce<false>(v);
            // slot: blocks go to the XCDs round-robin (block b on XCD b % 8) -- the 32 slots of a 128-byte line belong to ONE XCD's
            // blocks, so the line is assembled in that XCD's L2 and written once (PL_SSQ_MODE: 0 write-through + linear slots, which
            // measured +0.6 us on wo / w2: 8 XCDs read-modify-write each 32-byte sector)
            const u32 sl = blockIdx.x * ((a.RGB * 16u * (u32)NP) >> 6) + (tid >> 6);
#if PL_SSQ_MODE == 0
            if ((tid & 63u) == 63u) gq_store_wt(a.ssq_out + sl, tot);
#elif PL_SSQ_MODE == 1
            if ((tid & 63u) == 63u) gq_store_wt(a.ssq_out + ((sl & 7u) * ((u32)GQ_SSQ_SLOTS / 8u) + (sl >> 3)), tot);
#elif PL_SSQ_MODE == 2
            if ((tid & 63u) == 63u) a.ssq_out[(sl & 7u) * ((u32)GQ_SSQ_SLOTS / 8u) + (sl >> 3)] = tot;
#else
            if ((tid & 63u) == 63u) a.ssq_out[sl] = tot;
#endif
        }
    }
    // (the consumer adds all GQ_SSQ_SLOTS: block 0 clears the slots no wave of the grid writes)
    if (f_ssq && blockIdx.x == 0u)
        for (u32 i = gridDim.x * ((a.RGB * 16u * (u32)NP) >> 6) + tid; i < (u32)GQ_SSQ_SLOTS; i += T) {
#if PL_SSQ_MODE == 1 || PL_SSQ_MODE == 2
            a.ssq_out[(i & 7u) * ((u32)GQ_SSQ_SLOTS / 8u) + (i >> 3)] = 0.f;
#else
            a.ssq_out[i] = 0.f;
#endif
        }
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


// Work unit ("step") = 16 rows x one 1024-weight chunk x all planes = one A tile of BITS * 2 KiB; an item = cpi
// consecutive chunks of a row group, accumulated in registers by one wave.  Every wave streams its own tiles
// (direct-to-LDS loads into a private ring of S slots, own vmcnt) and multiplies them; what differs is the start:
//   early waves (first half, launched first): activation loads -> statistics -> B image.  They request their tiles
//                only when the image is built, so they never sit in a full memory issue queue before that.
//   late waves : request their first item at t = 0 (right behind the activation loads of the early waves) and may
//                block in the issue queue; they only wait for the image, through an LDS counter.
// Items: i < L (= W - E late waves) -> the first item of late wave E + i; i >= L -> wave (i - L) mod W.
#ifndef PL_WPE
#define PL_WPE 4
#endif
#ifndef PL_BOOB
#define PL_BOOB 1
#endif
#ifndef PL_LOOB
#define PL_LOOB 1
#endif
// local-image kernel: activations before tiles in the memory queue (0: round-2 order -- tiles requested at once, the wait counts them;
// 1: tiles only when the activations have landed; 2: the later half of the waves also builds its image first).  Per bit width: at 2
// bits 2 is the fastest by 1-2 % of the decode step, at 3 / 4 bits (8 waves, 6-8 KiB tiles) 0 is: 3-bit decode 680 -> 689 tokens/s,
// 4-bit 478 -> 483 (profiles/r05_knob_sweep.txt).  -DPL_XFIRST=n forces one value for every width.
#ifdef PL_XFIRST
template <int BITS>
constexpr int pl_xfirst() { return PL_XFIRST; }
#else
template <int BITS>
constexpr int pl_xfirst() { return BITS == 2 ? 2 : 0; }
#endif
#ifndef PL_PRIO
#define PL_PRIO 4  // local-image prologue: 4 priority graded by start order (default), 0 one raised level, 1 none
#endif
// MBT > 1: up to MBT batch rows in ONE pass over the planes (the reference's multi_row GEMV, anyprec.cu:381,425,494-506: M = 2..8
// rows from one read of the weight words): row mm of the block's a.MB rows has its own activation image and uses the MFMA columns
// 4 mm .. 4 mm + 3 (a row needs 4 of the 16: its bf8 pieces), its own E8M0 scale (supplied per lane) and statistics; the images
// are built one after the other through the same staged copy.  Plain prologue only; gridDim.y = ceil(M / a.MB).
// threads per block of the shared-image kernel: 16 waves at 2 bits, 8 at 4 bits (150+ VGPRs per lane); 3 bits: PL_T3 (512; 1024 --
// 113-125 VGPRs fit -- measured slower, w1w3 15.35 vs 14.85 us, decode 660 vs 666 tokens/s: profiles/r05_plane_spec_3_4_bits.txt)
#ifndef PL_T3
#define PL_T3 512
#endif
template <int BITS>
constexpr int pl_threads() { return BITS == 2 ? 1024 : (BITS == 3 ? PL_T3 : 512); }
// SPEC: the instances of the decode step's launches (like SPEC of the local-image kernel below): the activations come through the
// staged copy (a.rawx: true for every model shape -- the item-layout loads are the fallback of rings too small for the vector), K is
// a multiple of 1024, no statistics hand-over; the pair epilogue stays a run-time flag (w1w3 has it, w2 has not).
template <int BITS, int PRO, int NI, int MBT = 1, bool SPEC = false>
__global__ void __launch_bounds__(pl_threads<BITS>(), BITS == 2 ? PL_WPE : (pl_threads<BITS>() == 1024 ? 1 : 2)) ap_plane_kernel(PlaneArgs a) {
    const bool rawx = SPEC || a.rawx != 0u;
    static_assert(MBT == 1 || PRO == PRO_NONE, "several rows per pass: plain prologue only");
    constexpr int NP = 1 << BITS, NP1 = NP - 1;
    constexpr u32 T = (u32)pl_threads<BITS>(), W = T / 64u, E = W / 2u, L = W - E;  // early / late waves
    // NI prologue passes over the (chunk, virtual lane, weight pair) items: pass n gives early wave w the chunk (w >> 1) + n * E / 2
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
    const u32 nIt = a.RGB * CS;  // items of this block
    // LDS: [A rings: W x S slots][LUT rows of the block][B images: rows x (nchunks * 4096 (+ 64))][zero32 (64 B)][red: rows x 64 floats, ctr in row 0's][hot list: rows x K / 64 x 8 B][part]
    unsigned char *ring = smem + (size_t)w * S * SLOT;
    const u32 lut_bytes = (a.RGB * 16u * (u32)NP * 2u + LPS * 1024u - 1u) / (LPS * 1024u) * (LPS * 1024u);  // whole pseudo steps
    unsigned char *lutb = smem + (size_t)W * S * SLOT;
    const u32 *lutl = reinterpret_cast<const u32 *>(lutb);  // fp16 [RGB * 16][NP]
    const u32 m = MBT == 1 ? blockIdx.y : blockIdx.y * a.MB;                   // first batch row of this block
    const u32 MB = MBT == 1 ? 1u : min(a.MB, a.M - m);                         // ... and how many it serves
    const u32 MBA = MBT == 1 ? 1u : a.MB;                                      // (layout: rows allocated)
    unsigned char *bimg = lutb + lut_bytes;
    // image stride: + 64 B between the rows' images, so that the MFMA lanes of a ds_read_b128 group -- columns of different
    // rows at the same piece / unit -- fall into different banks (units {0,8,2,10} + 4 mm of the 16 per bank row)
    const u32 IMGS = G.nchunks * 4096u + (MBT > 1 ? 64u : 0u);
    unsigned char *zero32 = bimg + MBA * IMGS;
    float *red = reinterpret_cast<float *>(zero32 + 64);  // 64 floats per batch row (statistics, sums, scale); the counters in row 0's
    u32 *ctr = reinterpret_cast<u32 *>(red + 60);  // 0, 1, 3: software barriers; 2: extracted elements
    HotEnt *hotl = reinterpret_cast<HotEnt *>(red + 64u * MBA);
    const u32 hot_cap = G.K / 64u * MBA;              // Markov: |x| > 64 mean|x| holds for fewer than K / 64 elements of a row
    float *part = red + 64u * MBA + 2u * hot_cap;     // [item][batch row][subset][16 rows]
    const u32 rg0 = blockIdx.x * a.RGB;
    auto stamp = [&](int i) {
        if (GQ_STAMPS == 1 && a.dbg && blockIdx.x == gridDim.x / 2 && l == 0) a.dbg[w * 8u + (u32)i] = __builtin_readcyclecounter();
    };
    stamp(0);
    if PL_XF(32u) return;
    const bool early = w < E;
    auto stamp2 = [&](int i) {  // finer stamps of the prologue (second table of tools/phase_timing.py)
        if (GQ_STAMPS == 1 && a.dbg && blockIdx.x == gridDim.x / 2 && l == 0) a.dbg[128u + w * 16u + (u32)i] = __builtin_readcyclecounter();
    };

    // ---------------------------------------------------------------- 0. activation loads, then the first tiles
    // prologue item = (chunk, virtual lane t, nibble bit b): the weights j = 7 - b and j = 3 - b (plane bits b and b + 4 of
    // every byte) of the 4 bytes c of lane t -- the 8 activations that meet the two image words of (t, b)
    // Two ways to get them: (rawx) coalesced 16-byte loads, staged through LDS in the early waves' ring slots (free until
    // the image is built), one extra early-wave barrier -- every line of x is requested once per CU; or, when the ring is
    // too small for the copy, 16-bit loads straight into the item layout (each line requested up to 9 times: measured 7300
    // cycles instead of ~1500 until the K = 14336 vector has landed).
    u32 xr[NI][4], ar[NI][4], xh[NI][4], ah[NI][4];
    u32x4 rawv[MBT][NI], rawa[NI];
    const u32 pt = l & 31u, pb = ((w & 1u) << 1) | (l >> 5);  // the same for every pass (E * 64 is a multiple of 128)
    stamp2(5);
    if (early) {
        const u32x4 rsx = make_rsrc(a.x + (size_t)m * a.x_ld, (PRO == PRO_SILUMUL ? 4u : 2u) * G.K);
        const u32x4 rsa = make_rsrc(PRO == PRO_RMSNORM ? a.normw : a.x, 2u * G.K);
        if (rawx) {
#pragma unroll
            for (u32 mm = 0; mm < (u32)MBT; mm++) {  // (every row's loads are in flight before the first one is used)
                const u32x4 rsm = MBT == 1 ? rsx : make_rsrc(a.x + (size_t)(m + mm) * a.x_ld, mm < MB ? 2u * G.K : 0u);
#pragma unroll
                for (u32 n = 0; n < (u32)NI; n++) {
                    const u32 idx = tid + n * (E * 64u);  // 16-byte unit of the vector
                    const u32 voff = (idx < G.K / 8u && !PL_XF(8u)) ? 16u * idx : OOB;
                    rawv[mm][n] = bload128(rsm, voff, 0u);
                    if constexpr (PRO == PRO_RMSNORM) rawa[n] = bload128(rsa, voff, 0u);
                    if constexpr (PRO == PRO_SILUMUL) rawa[n] = bload128(rsx, voff, 2u * G.K);
                }
            }
        } else {
#pragma unroll
            for (u32 n = 0; n < (u32)NI; n++) {
                const u32 chunk = (w >> 1) + n * (E / 2u);  // wave-uniform
                const u32 tp = SPEC ? 32u : G.tpw(chunk);
                const bool ok = chunk < G.nchunks && pt < tp && !PL_XF(8u);
                const u32 vlo = ok ? 16u * pt + 2u * (7u - pb) : OOB, vhi = ok ? 16u * pt + 2u * (3u - pb) : OOB;
#pragma unroll
                for (u32 c = 0; c < 4; c++) {
                    const u32 soff = 2048u * chunk + 16u * tp * c;  // element 1024*chunk + 8*tp*c (+ 8t + j per lane)
                    xr[n][c] = bload16s(rsx, vlo, soff);
                    xh[n][c] = bload16s(rsx, vhi, soff);
                    if constexpr (PRO == PRO_RMSNORM) {
                        ar[n][c] = bload16s(rsa, vlo, soff);
                        ah[n][c] = bload16s(rsa, vhi, soff);
                    }
                    if constexpr (PRO == PRO_SILUMUL) {
                        ar[n][c] = bload16s(rsx, vlo, soff + 2u * G.K);
                        ah[n][c] = bload16s(rsx, vhi, soff + 2u * G.K);
                    }
                }
            }
        }
    }
    stamp2(6);
    // The CU's vector-memory pipe serves requests in order across waves: an activation load (an L2 hit) queued behind
    // another wave's plane loads waits for HBM (measured: 5000 cycles instead of 1200).  So every activation load of
    // the block is issued before the first plane load; this barrier (the LDS counters are zero behind it) costs the
    // launch skew of the last wave, ~800 cycles.
    if (tid < 4) ctr[tid] = 0u;
    if (tid < 16)
        for (u32 mm = 0; mm < MB; mm++) red[64u * mm + 32u + tid] = 0.f;
    if (tid < 16) reinterpret_cast<u32 *>(zero32)[tid] = 0u;
    stamp2(7);
    __syncthreads();
    // (everything a wave needs only behind the barrier is computed behind it: the barrier ends with the last-started wave -- launch
    // skew ~800 cycles -- and every instruction that wave runs before it keeps the early waves, whose activations have landed by
    // then, waiting: ~1,850 -> ~1,100 cycles from the start of the block)
    __builtin_amdgcn_sched_barrier(0);
    // plane stream of this wave: its items in order (late wave: item w - E first), cpi steps each, ring slot = step % S
    const u32 first_late = (!early && w - E < nIt) ? 1u : 0u;
    const u32 items_w = first_late + (nIt > L + w ? (nIt - L - w + W - 1u) / W : 0u);
    const u32 my_steps = PL_XF(2u) ? 0u : items_w * cpi;
    auto item_after = [&](u32 item) { return item < L ? L + w : item + W; };  // the wave's next item
    const u32 item0 = first_late ? w - E : L + w;
    const u32 plane_bytes = a.N * a.wpr_ld * 4u;
    const u32x4 rq = make_rsrc(a.qw, plane_bytes * (u32)BITS);
    const u32 ring_lds = (u32)(uintptr_t)ring;
    // Per-lane part of a tile address is constant (atile_src); the tile part is wave-uniform and goes into the
    // scalar offset.  No per-lane validity: rows >= N and the words of a short tail chunk beyond the row fetch
    // in-range garbage (or zeros past the end of the tensor) that only ever meets zero activation pieces /
    // rows that are never stored.  Steps whose chunk does not exist are skipped by issue and use alike.
    u32 lane_off[2];
    {
        u32 lr, lseg;
        atile_src(0u, l, lr, lseg);
        lane_off[0] = (lr * a.wpr_ld + 4u * lseg) * 4u;
        atile_src(1u, l, lr, lseg);
        lane_off[1] = (lr * a.wpr_ld + 4u * lseg) * 4u;
    }
    u32 iq_item = item0, iq_c = 0, iq_slot = 0, iq_n = 0;  // next step to request
    auto issue = [&]() {
        if PL_XF(1024u) {  // experiment: no plane loads at all (the MFMA phase runs on whatever is in LDS)
            iq_n++;
            return;
        }
        const u32 chunk = (iq_item & (CS - 1u)) * cpi + iq_c;
        const u32 rgi = rg0 + (iq_item >> a.log2CS);
        const u32 slot_lds = ring_lds + iq_slot * SLOT;
        const u32 tile_off = (rgi * 16u * a.wpr_ld + a.word0 + 32u * (chunk < G.nchunks ? chunk : 0u)) * 4u;
#pragma unroll
        for (u32 p = 0; p < (u32)BITS; p++)
#pragma unroll
            for (u32 h = 0; h < 2; h++) dma16s(rq, slot_lds + (p * 2u + h) * 1024u, lane_off[h], tile_off + p * plane_bytes);
        if (++iq_c == cpi) {
            iq_c = 0;
            iq_item = item_after(iq_item);
        }
        if (++iq_slot == S) iq_slot = 0;
        iq_n++;
    };
    // The LUT rows of the block ride in the queue of the last wave as nlut pseudo steps of exactly LPS loads each
    // (padded with out-of-range loads), behind its first item: loads return in order, so its vmcnt bookkeeping stays in
    // units of steps, and the LUT is in LDS once it has seen its last tile.
    const u32 nlut = (w == W - 1u && !PL_XF(16u)) ? lut_bytes / (LPS * 1024u) : 0u;
    u32 lut_after = 0;  // pseudo steps queued behind the steps issued so far
    if (!early) {
        // late wave: first item now (this may block in the issue queue: nothing else to do before the image is built)
        while (iq_n < my_steps && iq_n < S && (!PL_XF(256u) || iq_n < cpi * first_late)) issue();
        if (nlut) {
            const u32x4 rl = make_rsrc(a.lut, a.N * (u32)NP * 2u);
            const u32 want = a.RGB * 16u * (u32)NP * 2u;
            for (u32 o = 0; o < lut_bytes; o += 1024u)
                dma16(rl, (u32)(uintptr_t)lutb + o, o + 16u * l < want ? rg0 * 16u * (u32)NP * 2u + o + 16u * l : OOB);
            lut_after = nlut;
        }
    }
    const u32 lut_from = iq_n;  // the pseudo steps sit behind steps 0 .. lut_from-1
    stamp(6);

    float X = 0.f;
    int sb = 127;
    float nscale = 0.f, xmax = 0.f;
    // Who looks for elements above the extraction threshold (hot_step): with the staged copy in LDS and idle late waves, the
    // first half of the late waves -- their tiles are requested, they only wait for the image: nothing is added to the early
    // waves' serial chain, where every instruction costs 5-8 cycles; they also take the mean magnitude the threshold needs --;
    // otherwise the early waves, from their own maxima and one more sum in the statistics pass.
    constexpr u32 NSC = L >= 2u ? L / 2u : 1u;
    const bool scan_late = rawx && !a.himg && !(HOT_ABL & 2) && !(HOT_ABL & 32);
    // image passes: early wave w takes passes [0, NIe), with a.himg late wave w the passes [NIe, NI) of early wave w - E
    constexpr u32 NIe = (NI + 1) / 2;
    const bool helper = !early && a.himg;
    const u32 vw = early ? w : w - E;
    const u32 n0 = a.himg ? (early ? 0u : NIe) : 0u, n1 = a.himg ? (early ? NIe : (u32)NI) : (u32)NI;
    const bool scanner = scan_late && !early && w - E < NSC;
    const u32 img_arrivals = (a.himg ? W : E) + (scan_late ? NSC : 0u);
    // ---- elements above the extraction threshold (see hot_step): scanner waves, or the early waves when there are none.  Needs the
    // staged copy (without it -- K = 16384 behind RMSNorm -- nothing is extracted: round-2 behaviour there).
    u32 mmv = 0;  // the batch row being processed (detect, builders)
    auto detect = [&]() {
        const float *redm = red + 64u * mmv;
        const uint16_t *rx = reinterpret_cast<const uint16_t *>(smem), *ra = rx + G.K;  // the staged copy
        // in the RMSNorm prologue everything up to the rare path works on the products x w BEFORE the normalisation: the common
        // factor cancels in |x'| > 64 mean|x'|
        // each detecting wave takes a strided sample of the vector -- K / ND elements, its own mean: no exchange between the waves;
        // Markov holds per sample
        const u32 ND = early ? E : NSC;
        const u32 first = (early ? w : w - E) * 64u + l, stride = ND * 64u;
        float s1 = 0.f, mine = 0.f;
        for (u32 idx = first; idx < G.K / 8u; idx += stride) {
            const u32x4 xv = *reinterpret_cast<const u32x4 *>(rx + 8u * idx);
            u32x4 wv = xv;
            if constexpr (PRO == PRO_RMSNORM) wv = *reinterpret_cast<const u32x4 *>(ra + 8u * idx);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                float p = fabsf(h2f(xv[k] & 0xFFFF)), q = fabsf(h2f(xv[k] >> 16));
                if constexpr (PRO == PRO_RMSNORM) p *= fabsf(h2f(wv[k] & 0xFFFF)), q *= fabsf(h2f(wv[k] >> 16));
                s1 = (s1 + p) + q;
                mine = fmaxf(mine, fmaxf(p, q));
            }
        }
        s1 = wave_reduce<false>(s1);
        const float sum1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s1), 63));
        const float cnt = (float)((G.K + ND - 1u) / ND);  // (>= the sample size: a lower threshold, the bound on the count stands)
        const float tau_raw = GQ_HOT_T * 1.01f * sum1 * __builtin_amdgcn_rcpf(cnt);
        if (__builtin_expect(!(mine * 1.002f <= tau_raw), 0)) {
            // rare: some element of this lane may exceed the threshold (or the vector holds inf / NaN: no threshold then).  The exact
            // test on the transformed element; what passes is appended to the list and leaves the image once that is complete
            float ns = 1.f, xm = 0.f;
#pragma unroll
            for (u32 i = 0; i < E; i++) xm = fmaxf(xm, redm[i]);
            if constexpr (PRO == PRO_RMSNORM) {  // (the builders' own formulas: identical values)
                float tot = 0.f;
#pragma unroll
                for (u32 i = 0; i < E; i++) tot += redm[16 + i];
                ns = 1.0f / sqrtf(tot / (float)G.K + a.eps);
                xm = xm * ns * 1.002f;
            }
            const float tau_f = tau_raw * ns;
            const u32 tau = (HOT_ABL & 1) || !(tau_f < 65000.f) ? 0x7C00u : (u32)__builtin_bit_cast(uint16_t, (_Float16)tau_f) + 1u;
            const uint16_t k16h = pow2_f16(piece_shift(xm));
            auto consider = [&](u32 xh, u32 wh, u32 key) {  // one element (and its norm weight) as fp16 bits
                if constexpr (PRO == PRO_RMSNORM) {
                    const _Float16 hn = (_Float16)gq_pin_f32(h2f((uint16_t)xh) * ns);  // the transform of the image builders
                    xh = __builtin_bit_cast(uint16_t, (_Float16)(hn * __builtin_bit_cast(_Float16, (uint16_t)wh)));
                }
                if ((xh & 0x7FFFu) > tau) hot_append(ctr + 2, hotl, hot_cap, key, (uint16_t)xh, k16h);
            };
            if (tau < 0x7C00u) {
                {
                    for (u32 idx = first; idx < G.K / 8u; idx += stride) {
#pragma nounroll
                        for (u32 j = 0; j < 8; j++) {
                            const u32 e = 8u * idx + j;
                            u32 chunk, bb, hh, k;
                            locate_x4(G, e, chunk, bb, hh, k);
                            consider(rx[e], PRO == PRO_RMSNORM ? ra[e] : 0u, (mmv << 16) | (chunk << 10) | ((bb * 2u + hh) << 7) | k);
                        }
                    }
                }
            }
        }
    };
    stamp2(0);
    if (early) wait_vm<0>();  // the activation loads of every row (an early wave has nothing else in flight)
    stamp2(1);
#pragma unroll
    for (u32 mm = 0; mm < (u32)MBT; mm++) {  // batch rows of this block, one after the other through the same staged copy
    if (mm < MB) {
    mmv = mm;
    float *redm = red + 64u * mm;
    unsigned char *bimgm = bimg + mm * IMGS;
    if (early) {
    // ---------------------------------------------------------------- 1. statistics (+ staging) -> one early-wave barrier
    // RMSNorm: sum x^2 and max |x * w| (bounds the normalised maximum); otherwise max |x'| of the transformed vector.
    // With the LDS staging they are taken in the coalesced (raw) domain, before the copy is written, so the staging
    // barrier is also the statistics barrier; SiLU(gate) * up is applied there too and only the product is staged.
    {
        // RMSNorm: sum x^2 (ss) and max |x w| (fp32); otherwise max |x'| (packed integer maximum of the bit patterns)
        float ss = 0.f, mxf = 0.f;
        us2 mxp = {0, 0};  // |fp16| bit patterns order like unsigned integers
        auto stat = [&](u32 xw, u32 aw) {  // one packed pair of activations (and of norm weights)
            if constexpr (PRO == PRO_RMSNORM) {
                const float p = h2f(xw & 0xFFFF), q = h2f(xw >> 16);
                ss += p * p;
                ss += q * q;
                mxf = fmaxf(mxf, fmaxf(fabsf(p * h2f(aw & 0xFFFF)), fabsf(q * h2f(aw >> 16))));
            } else {
                mxp = __builtin_elementwise_max(mxp, __builtin_bit_cast(us2, xw & 0x7FFF7FFFu));
            }
        };
        if (rawx) {
            unsigned char *raw = smem;  // the early waves' ring slots
#pragma unroll
            for (u32 n = 0; n < (u32)NI; n++) {
                tie128(rawv[mm][n]);
                if constexpr (PRO != PRO_NONE) tie128(rawa[n]);
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if constexpr (PRO == PRO_SILUMUL) rawv[mm][n][k] = silu_mul2(rawv[mm][n][k], rawa[n][k]);
                    stat(rawv[mm][n][k], PRO == PRO_RMSNORM ? rawa[n][k] : 0u);
                }
                const u32 idx = tid + n * (E * 64u);
                if (idx < G.K / 8u) {
                    *reinterpret_cast<u32x4 *>(raw + 16u * idx) = rawv[mm][n];
                    if constexpr (PRO == PRO_RMSNORM) *reinterpret_cast<u32x4 *>(raw + 2u * G.K + 16u * idx) = rawa[n];
                }
            }
        } else {
#pragma unroll
            for (u32 n = 0; n < (u32)NI; n++) {
                tie4(xr[n]);
                tie4(xh[n]);
                if constexpr (PRO != PRO_NONE) {
                    tie4(ar[n]);
                    tie4(ah[n]);
                }
#pragma unroll
                for (u32 c = 0; c < 4; c++) {  // low half = weight 7 - b, high half = weight 3 - b
                    xr[n][c] |= xh[n][c] << 16;
                    if constexpr (PRO != PRO_NONE) ar[n][c] |= ah[n][c] << 16;
                    if constexpr (PRO == PRO_SILUMUL) xr[n][c] = silu_mul2(xr[n][c], ar[n][c]);
                    stat(xr[n][c], PRO == PRO_RMSNORM ? ar[n][c] : 0u);
                }
            }
        }
        float mx = PRO == PRO_RMSNORM ? mxf : h2f(max(mxp[0], mxp[1]));
        mx = wave_reduce<true>(mx);
        if constexpr (PRO == PRO_RMSNORM) ss = wave_reduce<false>(ss);
        if (l == 63) {
            redm[w] = mx;
            if constexpr (PRO == PRO_RMSNORM) redm[16 + w] = ss;
        }
        stamp2(2);
        arrive(ctr + 0, l);
    }
    }  // early: staging + statistics
    if (scanner) {
        wait_count(ctr + 0, E * (mm + 1u));
        if (!(HOT_ABL & 16)) detect();
        arrive(ctr + 1, l);
    } else if (early || helper) {
    {
        wait_count(ctr + 0, E * (mm + 1u));
        stamp2(3);
        xmax = 0.f;
#pragma unroll
        for (u32 i = 0; i < E; i++) xmax = fmaxf(xmax, redm[i]);
        if constexpr (PRO == PRO_RMSNORM) {
            float tot = 0.f;
#pragma unroll
            for (u32 i = 0; i < E; i++) tot += redm[16 + i];
            nscale = 1.0f / sqrtf(tot / (float)G.K + a.eps);
            xmax = xmax * nscale * 1.002f;  // covers the two fp16 roundings of the transform
        }
        stamp2(4);
        if (!(HOT_ABL & 2) && !scan_late && rawx && early) detect();
        if (rawx) {
            // the staged copy is complete: gather this thread's items (for SiLU the staged vector is already the product)
            const uint16_t *rx = reinterpret_cast<const uint16_t *>(smem), *ra = rx + G.K;
#pragma unroll
            for (u32 n = 0; n < (u32)NI; n++) {
                if (n < n0 || n >= n1) continue;
                const u32 chunk = (vw >> 1) + n * (E / 2u);
                const u32 tp = SPEC ? 32u : G.tpw(chunk);
                const bool ok = chunk < G.nchunks && pt < tp;
#pragma unroll
                for (u32 c = 0; c < 4; c++) {  // low half = weight 7 - b, high half = weight 3 - b
                    const u32 e = 1024u * chunk + 8u * tp * c + 8u * pt + 3u - pb;
                    xr[n][c] = ok ? ((u32)rx[e + 4u] | ((u32)rx[e] << 16)) : 0u;
                    if constexpr (PRO == PRO_RMSNORM) ar[n][c] = ok ? ((u32)ra[e + 4u] | ((u32)ra[e] << 16)) : 0u;
                }
            }
        }
    }
    stamp(7);

    // ---------------------------------------------------------------- 2. transform, scale, split, scatter
    // one power of two for the whole vector, from its maximum (extracted elements included: their entries carry the same scale)
    const int ksh = piece_shift(xmax);
    {
        const uint16_t k16 = pow2_f16(ksh);
        const h2v kk = u2h2((u32)k16 * 0x10001u), one2 = u2h2(0x3C003C00u);
        float xsum = 0.f;
#pragma unroll
        for (u32 n = 0; n < (u32)NI; n++) {
            const u32 chunk = (vw >> 1) + n * (E / 2u);
            if (n < n0 || n >= n1 || chunk >= G.nchunks) continue;
            const u32 t = pt, g = t >> 3, hh = (t >> 2) & 1u, vA = t & 3u;
            u32 P[4][4];  // [piece][c]
#pragma unroll
            for (u32 c = 0; c < 4; c++) {
                u32 xw = xr[n][c];
                if constexpr (PRO == PRO_RMSNORM) {
                    // (x.float() * rsqrt(mean(x^2)+eps)).half() * w  -- inference/model.py:281-292, both fp16 roundings kept
                    const _Float16 h0 = (_Float16)gq_pin_f32(h2f(xw & 0xFFFF) * nscale), h1 = (_Float16)gq_pin_f32(h2f(xw >> 16) * nscale);
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
            unsigned char *img = bimgm + bimg4_off(chunk, pb, hh, 0u) + 8u * vA;
#pragma unroll
            for (u32 p = 0; p < 4; p++) {
                // k = 32g + 8v + i, nibble i = 2 (3 - c) + (s >> 2): bytes (i = 0..3) = c3.j(7-b), c3.j(3-b), c2.j(7-b), c2.j(3-b);
                // the bf8 of weight 7 - b is byte 1 of P (low half), of weight 3 - b byte 3 (high half)
                // (pieces 2, 3 keep byte k at k ^ 32: see bimg_swz)
                *reinterpret_cast<u32 *>(img + p * 128u + ((32u * g) ^ bimg_swz(p))) = __builtin_amdgcn_perm(P[p][2], P[p][3], 0x07050301u);
                *reinterpret_cast<u32 *>(img + p * 128u + ((32u * g) ^ bimg_swz(p)) + 4u) = __builtin_amdgcn_perm(P[p][0], P[p][1], 0x07050301u);
            }
        }
        xsum = wave_reduce<false>(xsum);
        if (l == 63) {
            redm[32 + w] = xsum;
            if (w == 0) redm[48] = __builtin_bit_cast(float, (u32)(127 - ksh));  // E8M0 scale of every B block, for the late waves
        }
    }
    stamp(1);
    arrive(ctr + 1, l);
    }  // image builders
    // the staged copy is free for the next row once every builder and scanner is done with this one
    if (MBT > 1 && mm + 1u < MB && (early || scanner)) wait_count(ctr + 1, img_arrivals * (mm + 1u));
    }
    }
    wait_count(ctr + 1, img_arrivals * MB);  // the B images are complete
    stamp(2);
    // per batch row: sum(x) and the E8M0 scale of its image; a lane's scale operand is that of ITS column's row
    float Xr[MBT];
    const u32 mcol = (l & 15u) >> 2;
#pragma unroll
    for (u32 mm = 0; mm < (u32)MBT; mm++) {
        Xr[mm] = 0.f;
        if (mm < MB) {
#pragma unroll
            for (u32 i = 0; i < W; i++) Xr[mm] += red[64u * mm + 32u + i];
            if (mm == 0u || mm == mcol) sb = (int)__builtin_bit_cast(u32, red[64u * mm + 48u]);
        }
    }
    X = Xr[0];
    // (a plain LDS read, like red[]: a volatile / atomic access through the generic pointer compiles to a FLAT load whose
    // s_waitcnt vmcnt(0) drains the wave's whole plane stream -- measured: w1w3 9.4 -> 10.3 us)
    const u32 nhot_raw = (HOT_ABL & 256) ? 0u : __builtin_amdgcn_readfirstlane(__builtin_bit_cast(u32, red[62]));
    const u32 nhot = nhot_raw < hot_cap ? nhot_raw : hot_cap;
    if (!(HOT_ABL & 8) && __builtin_expect(nhot != 0u, 0)) {
        // rare (the same for every wave: appends precede the appender's arrival at the image barrier): the extracted elements
        // leave the image -- byte k of their (chunk, b, h) block in each of the 4 piece columns --, one more barrier
        for (u32 e = tid; e < 4u * nhot; e += T) {
            const u32 key = hotl[e >> 2].key;
            bimg[(key >> 16) * IMGS + ((key >> 10) & 63u) * 4096u + ((key >> 7) & 7u) * 512u + (e & 3u) * 128u + ((key & 127u) ^ bimg_swz(e & 3u))] = 0;
        }
        arrive(ctr + 3, l);
        wait_count(ctr + 3, W);
    }

    // ---------------------------------------------------------------- 3. main loop: the steps of this wave
    // the rest of the ring is requested now: the first tiles of the block have (mostly) landed, the queues have room
    while (iq_n < my_steps && iq_n < S) issue();
    const u32 r = l & 15u, kb = l >> 4;
    const u32 col = l & 15u;
    const bool bcol = col < 4u * MB;
    const u32 offA0 = atile_unit(r, 2u * kb) * 16u, offA1 = atile_unit(r, 2u * kb + 1u) * 16u;
    v4f acc[NP1];
#pragma unroll
    for (int i = 0; i < NP1; i++) acc[i] = (v4f){0.f, 0.f, 0.f, 0.f};
    u32 cq_item = item0, cq_c = 0, cq_slot = 0;
    for (u32 q = 0; q < my_steps; q++) {
        const u32 chunk = (cq_item & (CS - 1u)) * cpi + cq_c;
        const u32 after = iq_n - 1u - q + (q < lut_from ? lut_after : 0u);  // (pseudo) steps requested behind this one
        wait_vm_steps<LPS>(after);
        u32 Wd[BITS][8];
        load_tile<BITS>(Wd, ring + cq_slot * SLOT, offA0, offA1);
        if (iq_n < my_steps) {
            // refill this slot with the step S ahead: the ds_reads above must have returned first
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            issue();
        }
        if (++cq_slot == S) cq_slot = 0;
        if (chunk < G.nchunks && !PL_XF(1u)) {
            // MFMA columns without a piece (col >= 4 MB) must read zeros (measured: with real data in them -- the same addresses as
            // columns 0..3, an LDS broadcast -- w1w3 runs 10.7 instead of 9.8 us: the matrix cores draw more power on non-zero
            // operands and the chip clocks down).  PL_BOOB: their base address lies outside the workgroup's LDS allocation (192 KiB
            // and up; this chip has 160 KiB), where ds_read returns zeros (tools/ubench/lds_oob.hip) -- so the (b, h) block and k + 64
            // distances are immediates of the ds_reads for EVERY lane instead of per-lane address arithmetic (1.1 instructions per
            // MFMA less in a loop that issues more than the SIMD hides).  PL_BOOB = 0: a 64-byte zero area and per-lane strides 0.
#if PL_BOOB
            const unsigned char *bbase = bcol ? bimg + mcol * IMGS + bimg4_off(chunk, 0u, 0u, col & 3u) + ((16u * kb) ^ bimg_swz(col & 3u)) : smem + 0x30000u;
            mfma_chunk<BITS>(acc, Wd, bbase, 512u, 64u, sb);
#else
            const unsigned char *bbase = bcol ? bimg + mcol * IMGS + bimg4_off(chunk, 0u, 0u, col & 3u) + ((16u * kb) ^ bimg_swz(col & 3u)) : zero32;
            // next (b, h): 4 pieces * 128 B; second run of the lane at k + 64
            mfma_chunk<BITS>(acc, Wd, bbase, bcol ? 512u : 0u, bcol ? 64u : 0u, sb);
#endif
            if (!(HOT_ABL & 4) && __builtin_expect(nhot != 0u, 0)) hot_step<BITS>(acc, Wd, hotl, nhot, chunk, sb, col, kb);
        } else if PL_XF(1u) {
            acc[0][0] += __builtin_bit_cast(float, Wd[0][0] ^ Wd[BITS - 1][7]);
        }
        if (++cq_c == cpi) {
            park_item<NP1>(acc, part + (size_t)cq_item * MB * NP1 * 16u, col, kb, MB);  // part[item][batch row][subset][row]
            cq_c = 0;
            cq_item = item_after(cq_item);
        }
    }
    wait_vm<0>();  // (the LUT loads of the last wave when it had no tile)
    stamp(3);
    __syncthreads();
    stamp(4);

    // ---------------------------------------------------------------- 4. epilogue: coefficients x plane sums
#pragma unroll
    for (u32 mm = 0; mm < (u32)MBT; mm++)
        if (mm < MB) plane_epilogue<BITS, false, SPEC>(a, lutl, part + (size_t)mm * NP1 * 16u, Xr[mm], rg0, m + mm, tid, T, MB * NP1 * 16u);
    (void)X;
    stamp(5);
}

// ---------------------------------------------------------------------------------------------------------------
// "Local image" variant for the prologues without a whole-vector statistic (plain, SiLU(gate) * up): every wave loads the
// activations of ITS chunks (contiguous 2 KiB each), scales them by a power of two of its own (the E8M0 scale of its
// MFMAs undoes it, so waves with different scales add up in fp32 like any other partial sums), splits them and scatters
// the pieces into a private image.  No wave waits for another one before the K-split sums are added: the launch ->
// activation -> statistics -> image -> MFMA chain of the kernel above, with its two block-wide barriers, is what bounds
// the small matrices (N = 4096: 57 KiB of planes per CU), not the plane stream.
//   items: wave w takes items w, w + W, ...; CS divides W, so all of them cover the same chunks cs * cpi .. + cpi - 1,
//   cs = w mod CS, of different row groups.  Waves with the same cs build the same image redundantly (RGB > 1).
//   LDS: [A rings: W x S slots][LUT rows][images: W x NC x 4096][red: 64 floats][hot lists: W x 16 NC x 8 B][part]
// SPEC: the instance of the decode step's wo / w2 launches -- K a multiple of 1024 (no short tail chunk), plain or residual
// epilogue, no statistics hand-over: the tests of those launch properties (wave-uniform compares and branches, most of them on the
// prologue's critical path) are compiled out.  The host picks it when the launch qualifies (launch_local_inst); same arithmetic.
template <int BITS, int PRO, int NC, bool SPEC = false>
__global__ void __launch_bounds__(BITS == 2 ? 1024 : 512) ap_plane_local_kernel(PlaneArgs a) {
    static_assert(PRO != PRO_RMSNORM, "the RMSNorm scale needs the whole vector");
    constexpr int NP = 1 << BITS, NP1 = NP - 1;
    constexpr u32 T = BITS == 2 ? 1024u : 512u, W = T / 64u;
    constexpr u32 LPS = 2u * BITS, SLOT = 2048u * BITS, OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    Geom G;
    G.init(a.K);
    const u32 tid = threadIdx.x;
    const u32 w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const u32 l = tid & 63u;
    const u32 CS = 1u << a.log2CS, cpi = a.cpi, S = a.S;
    const u32 nIt = a.RGB * CS;
    unsigned char *ring = smem + (size_t)w * S * SLOT;
    const u32 lut_bytes = (a.RGB * 16u * (u32)NP * 2u + LPS * 1024u - 1u) / (LPS * 1024u) * (LPS * 1024u);
    unsigned char *lutb = smem + (size_t)W * S * SLOT;
    const u32 *lutl = reinterpret_cast<const u32 *>(lutb);
    unsigned char *img = lutb + lut_bytes + (size_t)w * NC * 4096u;
    float *red = reinterpret_cast<float *>(lutb + lut_bytes + (size_t)W * NC * 4096u);
    // extracted elements of this wave's chunks (see hot_step): |x| > 64 x the mean magnitude of the wave's NC * 1024 elements, < 16 NC of them
    constexpr u32 hot_cap = 16u * NC;
    u32 *hot_cnt = reinterpret_cast<u32 *>(red) + w;
    HotEnt *hotl = reinterpret_cast<HotEnt *>(red + 64) + (size_t)w * hot_cap;
    float *part = red + 64 + 2u * W * hot_cap;
    const u32 rg0 = blockIdx.x * a.RGB;
    const u32 m = blockIdx.y;
    const u32 items_w = nIt > w ? (nIt - w + W - 1u) / W : 0u;
    const u32 my_steps = items_w * cpi;
    const u32 chunk0 = (w & (CS - 1u)) * cpi;
    auto stamp = [&](int i) {
        if (GQ_STAMPS == 1 && a.dbg && blockIdx.x == gridDim.x / 2 && l == 0) a.dbg[w * 8u + (u32)i] = __builtin_readcyclecounter();
    };
    stamp(0);
    // (GQ_STAMPS == 2: start / end of EVERY block, s_memrealtime -- dbg[2 b], dbg[2 b + 1]; tools/r6/block_ramp.py)
    if (GQ_STAMPS == 2 && a.dbg && tid == 0u && blockIdx.y == 0u) a.dbg[2u * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
    // A wave that is still building its image shares its SIMD with older waves already in their main loops; instruction issue is
    // arbitrated by priority, then age, so the latest-started waves (whose activations also land last) would build their images in
    // the slots the others leave over -- and the block ends with them.  Prologue work goes first, and the later a wave started the
    // higher its priority while it builds its image (w2 4096 x 14336: 7.7 -> 7.3 us with one raised level, 6.9 us graded; with one
    // working wave per SIMD -- wo: 4 items -- there is nobody to overtake and the raised priority measured slower, 4.38 -> 4.55
    // us: only with more than W / 4 items).  In the shared-image kernel the same for the builders changed nothing.
    const bool prio = !(PL_PRIO & 1) && nIt > W / 4u;
#if PL_PRIO & 4
    if (prio) {  // the later a wave starts, the higher its priority while it builds its image
        if (w >= 3u * W / 4u) __builtin_amdgcn_s_setprio(3);
        else if (w >= W / 2u) __builtin_amdgcn_s_setprio(2);
        else if (w >= W / 4u) __builtin_amdgcn_s_setprio(1);
    }
#else
    if (prio) __builtin_amdgcn_s_setprio(3);
#endif

    // ---- 0. this wave's activations (16-byte units u = l, l + 64 of each chunk), then all of its tiles
    u32x4 xv[NC][2], gv[NC][2];
    {
        const u32x4 rsx = make_rsrc(a.x + (size_t)m * a.x_ld, (PRO == PRO_SILUMUL ? 4u : 2u) * G.K);
#pragma unroll
        for (u32 n = 0; n < (u32)NC; n++) {
            const u32 chunk = chunk0 + n;
            const u32 tp = SPEC ? 32u : G.tpw(chunk);
#pragma unroll
            for (u32 k = 0; k < 2; k++) {
                const u32 u = l + 64u * k;
                const bool ok = items_w && n < cpi && chunk < G.nchunks && u < 4u * tp;
                const u32 voff = ok ? 2048u * chunk + 16u * u : OOB;
                xv[n][k] = bload128(rsx, voff, 0u);
                if constexpr (PRO == PRO_SILUMUL) gv[n][k] = bload128(rsx, voff, 2u * G.K);
            }
        }
    }
    const u32 plane_bytes = a.N * a.wpr_ld * 4u;
    const u32x4 rq = make_rsrc(a.qw, plane_bytes * (u32)BITS);
    const u32 ring_lds = (u32)(uintptr_t)ring;
    u32 lane_off[2];
    {
        u32 lr, lseg;
        atile_src(0u, l, lr, lseg);
        lane_off[0] = (lr * a.wpr_ld + 4u * lseg) * 4u;
        atile_src(1u, l, lr, lseg);
        lane_off[1] = (lr * a.wpr_ld + 4u * lseg) * 4u;
    }
    u32 iq_item = w, iq_c = 0, iq_slot = 0, iq_n = 0;  // next step to request
    auto issue = [&]() {
        const u32 chunk = chunk0 + iq_c;
        const u32 rgi = rg0 + (iq_item >> a.log2CS);
        const u32 slot_lds = ring_lds + iq_slot * SLOT;
        const u32 tile_off = (rgi * 16u * a.wpr_ld + a.word0 + 32u * (chunk < G.nchunks ? chunk : 0u)) * 4u;
#pragma unroll
        for (u32 p = 0; p < (u32)BITS; p++)
#pragma unroll
            for (u32 h = 0; h < 2; h++) dma16s(rq, slot_lds + (p * 2u + h) * 1024u, lane_off[h], tile_off + p * plane_bytes);
        if (++iq_c == cpi) {
            iq_c = 0;
            iq_item += W;
        }
        if (++iq_slot == S) iq_slot = 0;
        iq_n++;
    };
    // The tiles are requested only when the wave's activations have landed: the CU's memory pipe serves the waves' requests in
    // order, and a wave that starts late (launch skew ~800 cycles) would otherwise find ~6 KiB of tile requests per earlier wave
    // ahead of its 2 KiB of activations.  The second half of the waves goes one step further (pl_xfirst 2): by the time their
    // activations are there the pipe is full of the first half's tiles and issuing BLOCKS the wave (measured: 2,000 .. 3,000 cycles
    // in the issue queue) -- they build their image first and request their tiles then.  The tile stream (57 KiB per CU for w2) is
    // short next to the image builds it overlaps with.
    u32 lut_from = 0, lut_after = 0, nlut = 0;
    auto request_tiles = [&]() {
        while (iq_n < my_steps && iq_n < S) issue();
        // LUT rows of the block: pseudo steps of exactly LPS loads in the queue of the last wave (see the kernel above)
        nlut = w == W - 1u ? lut_bytes / (LPS * 1024u) : 0u;
        if (nlut) {
            const u32x4 rl = make_rsrc(a.lut, a.N * (u32)NP * 2u);
            const u32 want = a.RGB * 16u * (u32)NP * 2u;
            for (u32 o = 0; o < lut_bytes; o += 1024u)
                dma16(rl, (u32)(uintptr_t)lutb + o, o + 16u * l < want ? rg0 * 16u * (u32)NP * 2u + o + 16u * l : OOB);
        }
        lut_from = iq_n, lut_after = nlut;
    };
    constexpr int XFIRST = pl_xfirst<BITS>();
    const bool tiles_late = XFIRST == 2 && w >= W / 2u;
    if constexpr (XFIRST != 0) wait_vm<0>();
    if (!tiles_late) request_tiles();
    stamp(6);

    // ---- 1. scale of this wave, pieces, image
    if constexpr (XFIRST == 0) wait_vm_steps<LPS>(iq_n + nlut);  // the activation loads were issued first (host: S + nlut <= 4)
    stamp(7);
    int sb = 127;
    u32 nhot = 0;
    if (items_w && !PL_XF(64u)) {  // (GQ_PL_XFLAGS 64: knock-out of the image build -- WRONG numerics, timing only)
        const h2v one2 = u2h2(0x3C003C00u);
        us2 mxp = {0, 0};
        float s2 = 0.f, xsum = 0.f;
#pragma unroll
        for (u32 n = 0; n < (u32)NC; n++)
#pragma unroll
            for (u32 k = 0; k < 2; k++) {
                tie128(xv[n][k]);
                if constexpr (PRO == PRO_SILUMUL) tie128(gv[n][k]);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    if constexpr (PRO == PRO_SILUMUL) xv[n][k][q] = silu_mul2(xv[n][k][q], gv[n][k][q]);
                    const u32 ab = xv[n][k][q] & 0x7FFF7FFFu;  // |fp16| bit patterns order like unsigned integers
                    mxp = __builtin_elementwise_max(mxp, __builtin_bit_cast(us2, ab));
                    s2 = __builtin_amdgcn_fdot2(u2h2(ab), one2, s2, false);
                    xsum = __builtin_amdgcn_fdot2(u2h2(xv[n][k][q]), one2, xsum, false);  // (loads outside the vector returned zeros)
                }
            }
        const u32 mxi = max(mxp[0], mxp[1]);
        const float mx = wave_reduce<true>(h2f((uint16_t)mxi));
        s2 = wave_reduce<false>(s2);
        const float xmax = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mx), 63));
        const float tot2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s2), 63));
        const u32 tau = hot_tau_bits(tot2, (float)(NC * 1024u));
        const int ksh = piece_shift(xmax);  // one scale for the image and the extracted elements
        sb = 127 - ksh;
        const bool any_hot = !(HOT_ABL & 2) && __builtin_amdgcn_ballot_w64(mxi > tau) != 0ull;  // wave-uniform, no LDS traffic
        if (__builtin_expect(any_hot, 0)) {
            if (l == 0) *hot_cnt = 0u;  // LDS operations of one wave are served in order
        }
        if (__builtin_expect(any_hot, 0) && mxi > tau) {
            // rare: unit u = (byte c, virtual lane t), word q = weights j = 2q, 2q + 1; plane bit s = 7 - j: b = s & 3,
            // nibble i = 2 (3 - c) + (s >> 2), k = 32 (t / 8) + 8 (t % 4) + i, h = (t / 4) % 2
#pragma unroll
            for (u32 n = 0; n < (u32)NC; n++) {
                const u32 chunk = chunk0 + n;
                if (n >= cpi || chunk >= G.nchunks) continue;
                const u32 tp = SPEC ? 32u : G.tpw(chunk);
#pragma unroll
                for (u32 k = 0; k < 2; k++) {
                    const u32 u = l + 64u * k;
                    if (u >= 4u * tp) continue;
                    const u32 c = u / tp, t = u % tp;
#pragma unroll
                    for (u32 q = 0; q < 4; q++)
#pragma unroll
                        for (u32 hf = 0; hf < 2; hf++) {
                            const u32 xh = (xv[n][k][q] >> (16u * hf)) & 0xFFFFu;
                            if ((xh & 0x7FFFu) > tau) {
                                const u32 sbit = 7u - (2u * q + hf);
                                const u32 kpos = 32u * (t >> 3) + 8u * (t & 3u) + 2u * (3u - c) + (sbit >> 2);
                                hot_append_inl(hot_cnt, hotl, hot_cap, (n << 10) | (((sbit & 3u) * 2u + ((t >> 2) & 1u)) << 7) | kpos, (uint16_t)xh,
                                           pow2_f16(ksh));
                                xv[n][k][q] &= ~(0xFFFFu << (16u * hf));
                            }
                        }
                }
            }
        }
        const h2v kk = u2h2((u32)pow2_f16(ksh) * 0x10001u);
#pragma unroll
        for (u32 n = 0; n < (u32)NC; n++) {
            const u32 chunk = chunk0 + n;
            if (n >= cpi || chunk >= G.nchunks) continue;
            const u32 tp = SPEC ? 32u : G.tpw(chunk);
#pragma unroll
            for (u32 k = 0; k < 2; k++) {
                // unit u = the 8 activations j = 0..7 of (byte c, virtual lane t): word q holds j = 2q (low half), 2q + 1
                const u32 u = l + 64u * k;
                const u32 c = tp == 32u ? u >> 5 : u / tp, t = tp == 32u ? u & 31u : u % tp;
                u32 P[4][4];  // [piece][q]
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const u32 xw = xv[n][k][q];
                    h2v rem = u2h2(xw) * kk;
#pragma unroll
                    for (u32 p = 0; p < 4; p++) {
                        P[p][q] = h22u(rem) & 0xFF00FF00u;
                        if (p < 3) rem = rem - u2h2(P[p][q]);
                    }
                }
                if (u >= 4u * tp) continue;
                // weight j has plane bit s = 7 - j: nibble bit b = s & 3, nibble i = 2 (3 - c) + (s >> 2), k = 32 g + 8 v + i.
                // For b: the bytes at i = 2(3-c), 2(3-c)+1 are j = 7 - b (byte 1 or 3 of word (7-b)/2) and j = 3 - b.
                unsigned char *dst = img + n * 4096u + (((t >> 2) & 1u) << 9) + 8u * (t & 3u) + 2u * (3u - c);
#pragma unroll
                for (u32 p = 0; p < 4; p++) {
                    // b = 0: j = 7, 3 (byte 3 of words 3, 1); b = 1: j = 6, 2 (byte 1 of words 3, 1)
                    const u32 v01 = __builtin_amdgcn_perm(P[p][1], P[p][3], 0x05010703u);  // lo half: b = 0, hi half: b = 1
                    // b = 2: j = 5, 1 (byte 3 of words 2, 0); b = 3: j = 4, 0 (byte 1 of words 2, 0)
                    const u32 v23 = __builtin_amdgcn_perm(P[p][0], P[p][2], 0x05010703u);
                    // (b, h) blocks are 4 pieces * 128 B = 512 B apart, b major: b * 1024 + h * 512; pieces 2, 3 keep byte k at k ^ 32
                    unsigned char *d = dst + p * 128u + ((32u * (t >> 3)) ^ bimg_swz(p));
                    *reinterpret_cast<uint16_t *>(d) = (uint16_t)v01;
                    *reinterpret_cast<uint16_t *>(d + 1024u) = (uint16_t)(v01 >> 16);
                    *reinterpret_cast<uint16_t *>(d + 2048u) = (uint16_t)v23;
                    *reinterpret_cast<uint16_t *>(d + 3072u) = (uint16_t)(v23 >> 16);
                }
            }
        }
        // a short tail chunk leaves the words of its missing virtual lanes untouched: clear them (k positions with t >= tp)
#pragma unroll
        for (u32 n = 0; n < (u32)NC; n++) {
            const u32 chunk = chunk0 + n;
            if (SPEC || n >= cpi || chunk >= G.nchunks || G.tpw(chunk) == 32u) continue;
            const u32 tp = SPEC ? 32u : G.tpw(chunk);
            for (u32 o = l; o < 1024u; o += 64u) {  // dword o of the chunk image: block o / 32, k = 4 (o % 32) -> t = 8 (k/32) + 4 h + (k%32)/8
                const u32 blk = o >> 5, k4 = ((o & 31u) * 4u) ^ bimg_swz(blk & 3u), hh = (blk >> 2) & 1u;  // (logical k of the stored position)
                const u32 t = 8u * (k4 >> 5) + 4u * hh + ((k4 & 31u) >> 3);
                if (t >= tp) reinterpret_cast<u32 *>(img + n * 4096u)[o] = 0u;
            }
        }
        xsum = wave_reduce<false>(xsum);
        if (l == 63) red[32 + w] = w < CS ? xsum : 0.f;  // sum(x): every chunk range counted once
        if (__builtin_expect(any_hot, 0)) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const u32 nh = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(u32, red[w]));  // plain LDS read (see the shared-image kernel)
            nhot = nh < hot_cap ? nh : hot_cap;
        }
    } else if (l == 63) {
        red[32 + w] = 0.f;
    }
    stamp(1);
    asm volatile("" ::: "memory");  // the image of this wave is read by this wave only: program order suffices
    if (tiles_late) request_tiles();
    if (prio) __builtin_amdgcn_s_setprio(0);

    // ---- 2. the steps of this wave
    const u32 r = l & 15u, kb = l >> 4, col = l & 15u;
    const u32 offA0 = atile_unit(r, 2u * kb) * 16u, offA1 = atile_unit(r, 2u * kb + 1u) * 16u;
    // columns 4..15 carry no piece: zeros from outside the LDS allocation (PL_BOOB, see the shared-image kernel; their sums are
    // never read, but non-zero operands cost power and with it clock) -- PL_LOOB = 0: they repeat columns 0..3
#if PL_LOOB
    const unsigned char *blane = col < 4u ? img + ((col & 3u) << 7) + ((16u * kb) ^ bimg_swz(col & 3u)) : smem + 0x30000u;
#else
    const unsigned char *blane = img + ((col & 3u) << 7) + ((16u * kb) ^ bimg_swz(col & 3u));
#endif
    v4f acc[NP1];
#pragma unroll
    for (int i = 0; i < NP1; i++) acc[i] = (v4f){0.f, 0.f, 0.f, 0.f};
    u32 cq_item = w, cq_c = 0, cq_slot = 0;
    for (u32 q = 0; q < my_steps; q++) {
        const u32 chunk = chunk0 + cq_c;
        const u32 after = iq_n - 1u - q + (q < lut_from ? lut_after : 0u);
        wait_vm_steps<LPS>(after);
        u32 Wd[BITS][8];
        load_tile<BITS>(Wd, ring + cq_slot * SLOT, offA0, offA1);
        if (iq_n < my_steps) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            issue();
        }
        if (++cq_slot == S) cq_slot = 0;
        if (chunk < G.nchunks) {
            mfma_chunk<BITS>(acc, Wd, blane + cq_c * 4096u, 512u, 64u, sb);
            if (!(HOT_ABL & 4) && __builtin_expect(nhot != 0u, 0)) hot_step<BITS>(acc, Wd, hotl, nhot, cq_c, sb, col, kb);
        }
        if (++cq_c == cpi) {
            park_item<NP1>(acc, part + (size_t)cq_item * NP1 * 16u, col, kb);
            cq_c = 0;
            cq_item += W;
        }
    }
    wait_vm<0>();
    stamp(3);
    __syncthreads();
    stamp(4);
    float X = 0.f;
#pragma unroll
    for (u32 i = 0; i < W; i++) X += red[32 + i];
    plane_epilogue<BITS, SPEC>(a, lutl, part, X, rg0, m, tid, T, NP1 * 16u);
    stamp(5);
    if (GQ_STAMPS == 2 && a.dbg && blockIdx.y == 0u) {
        __syncthreads();
        if (tid == 0u) a.dbg[2u * blockIdx.x + 1u] = __builtin_amdgcn_s_memrealtime();
    }
}

struct PlaneCfg {
    u32 grid, T, RGB, log2CS, cpi, S, NI, E;
    size_t smem;
};

int cus() { return gq_cu_count(); }

bool pick_plane_cfg(u32 N, u32 K, int bits, PlaneCfg &c, u32 MB = 1u) {
    if (K % 256u || K > 16384u) return false;
    const u32 nchunks = K / 1024u + ((K % 1024u) ? 1u : 0u);
    const u32 RGt = (N + 15u) / 16u;
    const u32 ncu = (u32)cus();
    const u32 W = bits == 2 ? 16u : (bits == 3 ? (u32)PL_T3 / 64u : 8u);
    c.T = 64u * W;
    // one block per CU when the matrix is big enough
    u32 rgb = (RGt + ncu - 1u) / ncu;
    if (rgb < 1) rgb = 1;
    c.RGB = rgb;
    c.grid = (RGt + rgb - 1u) / rgb;
    // split K of a row group over 2^log2CS items until the block has about 3/4 of an item per wave (measured over the 8B and
    // 70B shapes: 12..16 items per 16-wave block is the optimum -- fewer item ends / partial sums, still every SIMD busy)
    u32 lcs = 0;
    while (4u * (rgb << lcs) < 3u * W && (2u << lcs) <= nchunks) lcs++;
    const int envcs = gq_env_int("GQ_PL_LOG2CS", -1);
    if (envcs >= 0 && (1u << envcs) <= nchunks) lcs = (u32)envcs;
    c.log2CS = lcs;
    c.cpi = (nchunks + (1u << lcs) - 1u) >> lcs;
    const u32 nIt = rgb << lcs;
    const u32 steps_w = ((nIt + W - 1u) / W) * c.cpi;
    // per-wave ring: as many tiles as the wave has / as fit next to the B image (<= 4: vmcnt immediates)
    const u32 np1 = (1u << bits) - 1u;
    const size_t lps_bytes = 2048u * (size_t)bits;  // LPS loads of 1 KiB
    const size_t lutb = ((size_t)rgb * 16u * (np1 + 1u) * 2u + lps_bytes - 1u) / lps_bytes * lps_bytes;
    // (MB batch rows per pass: MB images 64 B apart, MB x the statistics / list / partial-sum areas)
    const size_t fixed = lutb + (size_t)MB * ((size_t)nchunks * 4096u + (MB > 1u ? 64u : 0u)) + 64u + (size_t)MB * 64u * 4u + (size_t)MB * (K / 64u) * 8u +
                         (size_t)MB * nIt * np1 * 16u * 4u;
    const size_t slot = 2048u * (size_t)bits, lds = 160u * 1024u;
    if (fixed + W * slot > lds) return false;
    u32 S = (u32)((lds - fixed) / (W * slot));
    const u32 nlut = (u32)(lutb / lps_bytes);
    if (nlut > 3u) return false;
    if (S > 5u - nlut - 1u) S = 5u - nlut - 1u;  // <= 4 steps behind the awaited one (vmcnt immediates)
    if (S > steps_w) S = steps_w;
    const int envs = gq_env_int("GQ_PL_S", 0);
    if (envs >= 1 && (u32)envs < S) S = (u32)envs;
    c.S = S;
    const u32 E = W / 2u;  // early waves (activation prologue); more of them was measured slower for every shape
    c.E = E;
    c.NI = (nchunks * 128u + E * 64u - 1u) / (E * 64u);  // prologue passes
    c.smem = fixed + (size_t)W * S * slot;
    return true;
}

// the local-image kernel: same grid / K split, a private image of cpi chunks per wave, every tile requested up front
bool pick_local_cfg(u32 N, u32 K, int bits, PlaneCfg &c) {
    // (4 bits: rounds 2-4 kept the shared image with the late-wave helpers; re-measured in round 5 -- stamps and ablation switches
    // compiled out, one ring slot per wave -- the local image wins on w2: 16.8 -> 15.3 us, decode 461 -> 472 tokens/s.  wo at 4 bits
    // stays on the exact kernel: the dispatcher's threshold, ap_gemv.hip.  profiles/r05_knob_sweep.txt)
    if (K % 256u || bits > gq_env_int("GQ_PL_LOCAL_MAXBITS", 4)) return false;
    const u32 nchunks = K / 1024u + ((K % 1024u) ? 1u : 0u);
    const u32 RGt = (N + 15u) / 16u, ncu = (u32)cus(), W = bits == 2 ? 16u : 8u;
    c.T = 64u * W;
    c.RGB = (RGt + ncu - 1u) / ncu;
    if (c.RGB < 1) c.RGB = 1;
    // waves of different row groups would build the same image: measured slower than the shared image from 2 row groups on
    if (c.RGB > (u32)gq_env_int("GQ_PL_LOCAL_MAXRGB", 1)) return false;
    c.grid = (RGt + c.RGB - 1u) / c.RGB;
    u32 lcs = 0;
    while ((c.RGB << lcs) < W && (1u << lcs) < nchunks) lcs++;  // up to one item per wave: nothing is shared between waves
    c.log2CS = lcs;
    c.cpi = (nchunks + (1u << lcs) - 1u) >> lcs;
    if (c.cpi > 4u) return false;
    c.NI = c.cpi <= 1u ? 1u : (c.cpi == 2u ? 2u : 4u);  // NC
    const u32 nIt = c.RGB << lcs, steps_w = ((nIt + W - 1u) / W) * c.cpi;
    const u32 np1 = (1u << bits) - 1u;
    const size_t lps_bytes = 2048u * (size_t)bits, slot = lps_bytes, lds = 160u * 1024u;
    const size_t lutb = ((size_t)c.RGB * 16u * (np1 + 1u) * 2u + lps_bytes - 1u) / lps_bytes * lps_bytes;
    const u32 nlut = (u32)(lutb / lps_bytes);
    const size_t fixed = lutb + (size_t)W * c.NI * 4096u + 64u * 4u + (size_t)W * 16u * c.NI * 8u + (size_t)nIt * np1 * 16u * 4u;
    if (nlut > 2u || fixed + W * slot > lds) return false;
    u32 S = (u32)((lds - fixed) / (W * slot));
    if (S > 4u - nlut) S = 4u - nlut;  // the activation loads are waited for with everything else in flight (vmcnt immediates)
    if (S > steps_w) S = steps_w;
    c.S = S;
    c.E = 0;
    c.smem = fixed + (size_t)W * S * slot;
    return true;
}

template <int BITS, int PRO, int NC, bool SPEC = false>
int launch_local_inst(const PlaneArgs &a, const PlaneCfg &c, u32 M, hipStream_t s) {
    static GqPerDeviceOnce once;
    auto kern = ap_plane_local_kernel<BITS, PRO, NC, SPEC>;
    GQ_HIP_CHECK(once.max_dynamic_lds(reinterpret_cast<const void *>(kern), (int)(160u * 1024u)));
    dim3 grid(c.grid, M), block(c.T);
    hipLaunchKernelGGL(kern, grid, block, c.smem, s, a);
    GQ_HIP_CHECK(hipGetLastError());
    return GQ_OK;
}

template <int BITS>
int launch_local(const PlaneArgs &a, const PlaneCfg &c, u32 M, int pro, hipStream_t s) {
    if (pro == PRO_SILUMUL) {
        if (c.NI == 1) return launch_local_inst<BITS, PRO_SILUMUL, 1>(a, c, M, s);
        if (c.NI == 2) return launch_local_inst<BITS, PRO_SILUMUL, 2>(a, c, M, s);
        return launch_local_inst<BITS, PRO_SILUMUL, 4>(a, c, M, s);
    }
    {  // the decode step's wo / w2 launches: see SPEC at the kernel (GQ_PL_SPEC=0: the general instances)
        if (a.K % 1024u == 0u && !a.pairs && !a.ssq_out && !PL_XF(~0u) && gq_env_int("GQ_PL_SPEC", 1)) {
            if (c.NI == 1) return launch_local_inst<BITS, PRO_NONE, 1, true>(a, c, M, s);
            if (c.NI == 2) return launch_local_inst<BITS, PRO_NONE, 2, true>(a, c, M, s);
            return launch_local_inst<BITS, PRO_NONE, 4, true>(a, c, M, s);
        }
    }
    if (c.NI == 1) return launch_local_inst<BITS, PRO_NONE, 1>(a, c, M, s);
    if (c.NI == 2) return launch_local_inst<BITS, PRO_NONE, 2>(a, c, M, s);
    return launch_local_inst<BITS, PRO_NONE, 4>(a, c, M, s);
}

template <int BITS, int PRO, int NI, int MBT = 1, bool SPEC = false>
int launch_plane_inst(const PlaneArgs &a, const PlaneCfg &c, u32 M, hipStream_t s) {
    static GqPerDeviceOnce once;
    auto kern = ap_plane_kernel<BITS, PRO, NI, MBT, SPEC>;
    GQ_HIP_CHECK(once.max_dynamic_lds(reinterpret_cast<const void *>(kern), (int)(160u * 1024u)));
    dim3 grid(c.grid, MBT == 1 ? M : (M + a.MB - 1u) / a.MB), block(c.T);
    hipLaunchKernelGGL(kern, grid, block, c.smem, s, a);
    GQ_HIP_CHECK(hipGetLastError());
    return GQ_OK;
}

// several batch rows per pass (plain prologue, staged copy, NI <= 2: K <= 8192 at 2 bits, K <= 4096 at 3 / 4 bits)
template <int BITS>
int launch_plane_rows(const PlaneArgs &a, const PlaneCfg &c, u32 M, hipStream_t s) {
    if (c.NI <= 1) return launch_plane_inst<BITS, PRO_NONE, 1, 4>(a, c, M, s);
    if (c.NI == 2) return launch_plane_inst<BITS, PRO_NONE, 2, 4>(a, c, M, s);
    return GQ_ENOTSUP;
}

template <int BITS, int PRO>
int launch_plane_ni(const PlaneArgs &a, const PlaneCfg &c, u32 M, hipStream_t s) {
    // K <= 16384 -> at most 2048 items over 512 / 256 lanes
    if constexpr (PRO != PRO_SILUMUL) {  // the decode launches this kernel serves (3 / 4 bits: wqkv, w1w3 behind RMSNorm, w2 plain; 2 bits: the 70B widths): see SPEC at the kernel
        if (a.rawx && a.K % 1024u == 0u && !a.ssq_out && !PL_XF(~0u) && gq_env_int("GQ_PL_SPEC", 1)) {
            if (c.NI <= 1) return launch_plane_inst<BITS, PRO, 1, 1, true>(a, c, M, s);
            if (c.NI == 2) return launch_plane_inst<BITS, PRO, 2, 1, true>(a, c, M, s);
            if (c.NI <= 4) return launch_plane_inst<BITS, PRO, 4, 1, true>(a, c, M, s);
            if constexpr (BITS != 2) {
                if (c.NI <= 8) return launch_plane_inst<BITS, PRO, 8, 1, true>(a, c, M, s);
            }
        }
    }
    if (c.NI <= 1) return launch_plane_inst<BITS, PRO, 1>(a, c, M, s);
    if (c.NI == 2) return launch_plane_inst<BITS, PRO, 2>(a, c, M, s);
    if (c.NI <= 4) return launch_plane_inst<BITS, PRO, 4>(a, c, M, s);
    if constexpr (BITS != 2) {
        if (c.NI <= 8) return launch_plane_inst<BITS, PRO, 8>(a, c, M, s);
    }
    return GQ_ENOTSUP;
}

template <int BITS>
int launch_plane(const PlaneArgs &a, const PlaneCfg &c, u32 M, int pro, hipStream_t s) {
    switch (pro) {
        case PRO_RMSNORM: return launch_plane_ni<BITS, PRO_RMSNORM>(a, c, M, s);
        case PRO_SILUMUL: return launch_plane_ni<BITS, PRO_SILUMUL>(a, c, M, s);
        default: return launch_plane_ni<BITS, PRO_NONE>(a, c, M, s);
    }
}

unsigned long long *g_dbg = nullptr;
}  // namespace

extern "C" void gq_debug_set_timing_buffer(void *p) { g_dbg = (unsigned long long *)p; }
unsigned long long *gq_debug_timing_buffer() { return g_dbg; }  // (ap_stream.hip stamps into the same buffer)
int gq_stream_gemv_try(const void *x, void *out, const uint32_t *qweight, const void *lut, uint32_t M, uint32_t N, uint32_t K, int bits,
                       const void *normw, float eps, const void *resid, int pro, int pairs, hipStream_t stream, GqHandover *ho);  // ap_stream.hip

namespace {
int plane_launch_slice(const void *x, void *out, const uint32_t *qweight, const void *lut, uint32_t M, uint32_t N, uint32_t Kfull,
                       uint32_t k0, uint32_t Ks, int bits, const void *normw, float eps, const void *resid, int pro, int pairs,
                       hipStream_t stream, GqHandover *ho = nullptr) {
    PlaneCfg c;
    // M = 2 .. 8 batch rows: up to 4 of them share ONE pass over the planes (shared-image kernel, one image per row) when the images
    // fit LDS next to the rings and the staged copy -- K <= 4096 for 4 rows, K <= 8192 for 2 --; else one block row per batch row
    u32 MB = 1u;
    if (M >= 2u && pro == PRO_NONE && !pairs && gq_env_int("GQ_PL_ONEPASS", 1)) {
        for (u32 mb = M < 4u ? M : 4u; mb >= 2u && MB == 1u; mb--) {
            PlaneCfg cm;
            if (!pick_plane_cfg(N, Ks, bits, cm, mb) || cm.NI > 2u) continue;
            const size_t need = (size_t)Ks * 2u, have = (size_t)(cm.T / 128u) * cm.S * 2048u * (size_t)bits;
            if (need > have) continue;
            MB = mb;
            c = cm;
        }
    }
    const bool local = MB == 1u && pro != PRO_RMSNORM && gq_env_int("GQ_PL_LOCAL", 1) && pick_local_cfg(N, Ks, bits, c);
    if (MB == 1u && !local && !pick_plane_cfg(N, Ks, bits, c)) return GQ_ENOTSUP;
    // Round 5: ONE ring slot per wave whenever the staged activation copy still fits the ring (it lives in the early waves' slots) -- a wave
    // then has one tile in flight instead of two or three: w1w3 14.7 -> 14.1 us at 3 bits, 25.5 -> 23.4 at 4 bits, the other shapes and the
    // 70B widths at 2 bits unchanged or slightly up (profiles/r05_plane_ring_slots.txt).  GQ_PL_S1=0 restores the old choice.
    if (MB == 1u && !local && c.S > 1u && gq_env_int("GQ_PL_S", 0) == 0 && gq_env_int("GQ_PL_S1", 1)) {
        const size_t need = (size_t)Ks * 2u * (pro == PRO_RMSNORM ? 2u : 1u), have1 = (size_t)(c.T / 128u) * 2048u * (size_t)bits;
        if (need <= have1) {
            c.smem -= (size_t)(c.S - 1u) * (c.T / 64u) * 2048u * (size_t)bits;
            c.S = 1u;
        }
    }
    PlaneArgs a{};
    a.qw = qweight;
    a.lut = (const uint16_t *)lut;
    a.x = (const uint16_t *)x + k0;
    a.out = (uint16_t *)out;
    a.normw = (const uint16_t *)normw;
    a.resid = (const uint16_t *)resid;
    a.N = N;
    a.K = Ks;
    a.wpr_ld = Kfull / 32u;
    a.word0 = k0 / 32u;
    a.x_ld = pro == PRO_SILUMUL ? 2u * Kfull : Kfull;
    a.RGB = c.RGB;
    a.log2CS = c.log2CS;
    a.cpi = c.cpi;
    a.S = c.S;
    a.pairs = pairs ? 1u : 0u;
    a.MB = MB;
    a.M = M;
    {
        const size_t need = (size_t)Ks * 2u * (pro == PRO_RMSNORM ? 2u : 1u), have = (size_t)(c.T / 128u) * c.S * 2048u * (size_t)bits;
        a.rawx = (need <= have && gq_env_int("GQ_PL_RAWX", 1)) ? 1u : 0u;
        a.himg = (a.rawx && c.NI >= 2u && c.S == 1u && gq_env_int("GQ_PL_HIMG", 1)) ? 1u : 0u;
    }
    if (MB > 1u) {
        a.rawx = 1u;
        a.himg = 0u;
        switch (bits) {
            case 2: return launch_plane_rows<2>(a, c, M, stream);
            case 3: return launch_plane_rows<3>(a, c, M, stream);
            default: return launch_plane_rows<4>(a, c, M, stream);
        }
    }
    a.xflags = (u32)gq_env_int("GQ_PL_XFLAGS", 0);
    a.eps = eps;
    a.dbg = g_dbg;
    // the block's outputs in whole epilogue waves of ONE pass, one slot per wave (plane_epilogue)
    if (ho && ho->ssq_out && M == 1u && !pairs && c.RGB * 16u * (1u << bits) <= c.T && c.grid * ((c.RGB * 16u * (1u << bits)) >> 6) <= (u32)GQ_SSQ_SLOTS) {
        a.ssq_out = ho->ssq_out;
        ho->ssq_written = true;
    }
    if (ho && ho->dry) return GQ_OK;
    if (local) {
        a.rawx = a.himg = 0u;
        switch (bits) {
            case 2: return launch_local<2>(a, c, M, pro, stream);
            case 3: return launch_local<3>(a, c, M, pro, stream);
            default: return launch_local<4>(a, c, M, pro, stream);
        }
    }
    switch (bits) {
        case 2: return launch_plane<2>(a, c, M, pro, stream);
        case 3: return launch_plane<3>(a, c, M, pro, stream);
        default: return launch_plane<4>(a, c, M, pro, stream);
    }
}
}  // namespace

// true when the local-image variant would serve this shape (<= 16 rows per CU, 2/3-bit): the dispatcher's threshold differs
bool gq_plane_local_shape(uint32_t N, uint32_t K, int bits) {
    PlaneCfg c;
    return bits >= 2 && K <= 16384u && gq_env_int("GQ_PL_LOCAL", 1) && pick_local_cfg(N, K, bits, c);
}

// returns GQ_ENOTSUP when the shape is not served by this path (caller falls back to the exact kernels)
int gq_stream_gemv_ksplit(const void *x, void *out, const uint32_t *qweight, const void *lut, uint32_t N, uint32_t K, int bits,
                          const void *resid, void *ws, size_t ws_bytes, hipStream_t stream);  // ap_stream.hip
int gq_plane_gemv_try(const void *x, void *out, const uint32_t *qweight, const void *lut, uint32_t M, uint32_t N, uint32_t K,
                      int bits, const void *normw, float eps, const void *resid, int pro, int pairs, hipStream_t stream, void *ws, size_t ws_bytes,
                      GqHandover *ho) {
    if (bits < 2 || bits > 4) return GQ_ENOTSUP;
    const uint64_t qbytes = (uint64_t)bits * N * (K / 8u);
    if (qbytes >= 0x7FFFFFFFull) return GQ_ENOTSUP;
    if (((uintptr_t)qweight | (uintptr_t)x | (uintptr_t)normw) & 15u) return GQ_ENOTSUP;
    // the stream kernel (ap_stream.hip) first where it measured faster (profiles/r04_stream_kernel.txt): the RMSNorm-prologue
    // launches of the widths up to 4096 at 2 bits (8B wqkv / w1w3); GQ_ST = 0 never, 2 every shape it serves, 3 every prologue too
    {
        const int st = gq_env_int("GQ_ST", 1);
        // (and the 70B attention output projection, 8192 x 8192 without a prologue: 7.6 vs 7.9 us)
        if (st >= 3 || (st == 2 && pro == PRO_RMSNORM) || (st == 1 && pro == PRO_RMSNORM && bits == 2 && K <= 4096u) ||
            (st == 1 && pro == PRO_NONE && !pairs && bits == 2 && K == 8192u && N >= 8192u && M == 1u) ||
            // (round 5: and the plain launch -- the reference's own operator -- of the 8B gate / up matrix: 8.39 vs 9.26 us; wqkv, wo
            // and w2 stay on the plane kernels: 4.95 / 4.03 / 6.44 vs 5.26 / 4.68 / 7.62.  profiles/r05_plain_launch_dispatch.txt)
            (st == 1 && pro == PRO_NONE && bits == 2 && K <= 4096u && (uint64_t)N * K >= 100000000ull && M == 1u)) {
            const int rc = gq_stream_gemv_try(x, out, qweight, lut, M, N, K, bits, normw, eps, resid, pro, pairs, stream, ho);
            if (rc != GQ_ENOTSUP) return rc;
        }
    }
    if (K <= 16384u) return plane_launch_slice(x, out, qweight, lut, M, N, K, 0u, K, bits, normw, eps, resid, pro, pairs, stream, ho);
    // 16384 < K <= 32768 (the 70B down projection): the B image of the whole row does not fit LDS, so the row is split at a
    // chunk boundary into two launches; the second adds its half to the first one's fp16 result through the residual
    // epilogue (out[n] = out[n] + y2[n], every element read and written by the same lane).  Two fp16 roundings instead
    // of one; plain and residual epilogues only.
    if (K > 32768u || K % 256u || pro != PRO_NONE || pairs) return GQ_ENOTSUP;
    if (ho && ho->dry) return GQ_OK;  // (plan only: the chained / K-split forms have no hand-over form, and nothing may be launched)
    if (ws && M == 1u) {  // with a workspace: K split over blocks, one fp16 rounding (ap_stream.hip)
        const int rc = gq_stream_gemv_ksplit(x, out, qweight, lut, N, K, bits, resid, ws, ws_bytes, stream);
        if (rc != GQ_ENOTSUP) return rc;
    }
    const uint32_t k1 = ((K / 2u + 1023u) / 1024u) * 1024u;
    PlaneCfg c;
    if (!pick_plane_cfg(N, k1, bits, c) || !pick_plane_cfg(N, K - k1, bits, c)) return GQ_ENOTSUP;
    int rc = plane_launch_slice(x, out, qweight, lut, M, N, K, 0u, k1, bits, nullptr, eps, resid, pro, 0, stream);
    if (rc != GQ_OK) return rc;
    return plane_launch_slice(x, out, qweight, lut, M, N, K, k1, K - k1, bits, nullptr, eps, out, pro, 0, stream);
}
