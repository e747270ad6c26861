// qtip.hip -- QTIP trellis-decoded matvec and the Sylvester Hadamard transform for gfx950.
//
// Replaces kernel_decompress_matvec<16,9,R,1,M,1,K> (qtip/qtip-kernels/src/inference.cu:168-425) -- an mma.sync
// program with a 32x-replicated shared-memory codebook, inline-PTX streaming loads and 73 compile-time shapes -- by
// one runtime-shaped wave-64 kernel:
//   * a wave walks 2x2 tile blocks (32 rows x 32 columns, 128*R contiguous bytes); lane (a4 = lane/32, s = lane%32)
//     owns the 4 trellis states 4s..4s+3 of the tiles of tile-row parity a4, i.e. rows a, a+8 (a = s/4) and
//     columns 2b, 2b+1, 2b+8, 2b+9 (b = s%4) -- the A-fragment slot order of the format (finetune.py:291-296);
//   * the 16-bit sliding window needs the next lane's stream unit (one ds_bpermute per tile);
//   * state -> (state*(state+1)) >> 6 & 511 -> fp16 pair from a 2 KiB LDS table, sign bit folded in with one XOR
//     (quantlut_sym, bitshift.py:72-80); v_dot2_f32_f16 accumulates in fp32;
//   * the K range of a 32-row band is split over the waves of the block and combined in LDS in a fixed order.
// Also: gq_hadamard, y = scale * x @ H_n for n a power of two (fast_hadamard_transform as used by
// inference/lib/utils/matmul_had.py:96-106).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <type_traits>

#include "gq_internal.h"
#include "fwht.h"

#ifndef QT_XOOB
#define QT_XOOB 1
#endif
namespace {
typedef uint32_t u32;
using gq_fwht::fwht_lds;
typedef unsigned long long u64;
typedef _Float16 h16;
typedef h16 h16x2 __attribute__((ext_vector_type(2)));

template <int R>
__device__ __forceinline__ u32 unit_of(const u32 (&d)[R], u32 j);  // R-byte little-endian unit j of the 4R-byte s-row
template <>
__device__ __forceinline__ u32 unit_of<2>(const u32 (&d)[2], u32 j) { return (d[j >> 1] >> (16u * (j & 1u))) & 0xFFFFu; }
template <>
__device__ __forceinline__ u32 unit_of<3>(const u32 (&d)[3], u32 j) {
    switch (j) {
        case 0: return d[0] & 0xFFFFFFu;
        case 1: return (d[0] >> 24) | ((d[1] & 0xFFFFu) << 8);
        case 2: return (d[1] >> 16) | ((d[2] & 0xFFu) << 16);
        default: return d[2] >> 8;
    }
}
template <>
__device__ __forceinline__ u32 unit_of<4>(const u32 (&d)[4], u32 j) { return d[j]; }

#ifndef QT_ABL
#define QT_ABL 0  // experiments (tools/qtip_ablation.sh): 1 no codebook lookups, 2 no MFMA, 4 no ds_bpermute, 8 no activation read, 16 no state arithmetic, 32 no FWHT, 64 no codebook fill
#endif
typedef h16 h16x8 __attribute__((ext_vector_type(8)));
unsigned long long *g_qdbg = nullptr;  // optional per-wave phase stamps of the middle block (tools/qtip_phase_timing.py)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef u32 u32x2 __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------------------------------------ the band engine
// Round 2 redesign (round 1: one v_dot2 per state, one 2 KiB codebook with ~3.5-way bank conflicts, 8 VALU per state).
//
// Lane <-> trellis mapping.  The format hands lane s = 4 a + b of a 16x16 tile the states of rows a, a + 8 and columns
// 2b, 2b+1, 2b+8, 2b+9 -- the A fragment of the reference's mma.sync.  v_mfma_f32_16x16x32_f16 wants lane (g = l / 16,
// r = l % 16) to hold 8 k-slots of ONE row r, all 16 rows of a lane group g agreeing on the columns.  Which global bytes a
// lane loads is free, so lane l = 16 b + 8 a4 + a takes stream unit s = 4 a + b of tile row a4 (the two 16-row halves of
// the band are the two halves of the 16 logical rows), for BOTH column tiles a3 = 0, 1 of the 32-column tile block:
//   A operand "rows a"     = states i = 0, 2 of a3 = 0, 1  -> columns 2g + {0,1}, 16 + 2g + {0,1}, 8 + 2g + {0,1}, 24 + 2g + {0,1}
//   A operand "rows a + 8" = states i = 1, 3                (same columns),
//   B operand (all 16 columns equal) = those 8 activations: ONE ds_read_b128 of a copy of x stored in that order.
// Two MFMAs per 32 x 32 tile block replace 8 v_dot2 per lane and the cross-lane sums; fp32 accumulation as before.
//
// State arithmetic, R = 2 (two states per register; halves = column tiles a3 = 0 / 1):
//   pack  = {unit(a3 = 1), unit(a3 = 0)} (one v_perm), npack = the neighbour lane's pack (one ds_bpermute);
//   P_i   = 16-bit windows at bit offset 4 i of {pack : npack} per half (i = 0: pack itself; 7 ops for i = 1..3);
//   M_i   = v_pk_mad_u16(P_i, P_i, P_i) = st (st + 1) mod 2^16 per half: bits 6..14 = codebook entry, bit 15 = sign;
//   entry address = ((M & 0xFFC0) << 1) | 4 (lane % 32): the codebook is held 32 times, entry-major (128 bytes per entry,
//   lane l reads copy l % 32: ds_read_b32 is banked (a / 4) % 32 per 32-lane group, so every lookup is conflict-free), and
//   sign-expanded (1024 entries; entry 512 + e = entry e with the low half negated) when the 128 KiB fit beside the
//   activations (SX = 1), else 512 entries + 3 ops per register for the sign (SX = 0).
// R = 3, 4 build the same P_i registers through 64-bit shifts (generic path).
//
// Work distribution: `nitems` items = (32-row band, K range); block b walks items b, b + gridDim, ... with ONE codebook
// fill and ONE activation prologue; the 16 waves of a block split an item's tile blocks (K2 = k2lo + w, + W, ...), their
// register prefetch queue (PF tile blocks per wave) runs across item boundaries.  The sum of a band is formed in a fixed
// order: tile blocks in ascending order within a wave (MFMA accumulator), waves in ascending order through LDS.
template <int SX>
struct QtipTab {
    static constexpr u32 ENT = SX ? 1024u : 512u;
    static constexpr u32 WORDS = ENT * 32u;
};

// Codebook fill.  Lane c % 8 of a group of 8 writes 16-byte chunk c % 8 of entry c / 8: a wave-instruction covers 1 KiB of
// contiguous LDS (entry-per-lane stores -- stride 128 bytes -- put all lanes on the same 4 banks: measured 8,500 cycles for
// the fill, 5x).  With 1024 threads chunk tid + 1024 k belongs to entry tid / 8 + 128 k: four words per thread, REQUESTED
// BEFORE the first tile blocks (vector memory returns in order: behind them the words would arrive after ~4,000 cycles), the
// sign-expanded upper half of the table from the same four registers.
struct QtipTabRegs {
    u32 w[4];
};
__device__ __forceinline__ void qtip_table_request(QtipTabRegs &r, const uint16_t *tlut) {
    if (blockDim.x == 1024u) {
#pragma unroll
        for (u32 k = 0; k < 4; k++) r.w[k] = reinterpret_cast<const u32 *>(tlut)[(threadIdx.x >> 3) + 128u * k];
    }
}
template <int SX>
__device__ __forceinline__ void qtip_fill_table(u32 *tab, const uint16_t *tlut, const QtipTabRegs &r) {
    const u32 T = blockDim.x, tid = threadIdx.x;
#if QT_ABL & 64
    if (tid < 64u) tab[tid] = r.w[0];
    return;
#endif
    if (T == 1024u) {
#pragma unroll
        for (u32 k = 0; k < 4; k++) {
            const u32 wv = r.w[k];
            reinterpret_cast<u32x4 *>(tab)[tid + 1024u * k] = u32x4{wv, wv, wv, wv};
            if (SX) {
                const u32 ws = wv ^ 0x8000u;
                reinterpret_cast<u32x4 *>(tab)[tid + 1024u * (k + 4u)] = u32x4{ws, ws, ws, ws};
            }
        }
        return;
    }
    for (u32 c = tid; c < QtipTab<SX>::ENT * 8u; c += T) {
        const u32 e = c >> 3;
        u32 wv = reinterpret_cast<const u32 *>(tlut)[e & 511u];
        if (SX && (e & 512u)) wv ^= 0x8000u;
        reinterpret_cast<u32x4 *>(tab)[c] = u32x4{wv, wv, wv, wv};
    }
}

// position (in halves) of activation k inside the permuted copy: tile block K2 = k / 32, pair q = (k % 32) / 2 -> lane group
// g = q % 4, slot {0, 2, 1, 3}[q / 4] (the order of the A registers above), element k % 2
__device__ __forceinline__ u32 qtip_xpos(u32 k) {
    const u32 q = (k >> 1) & 15u, qc = q >> 2;
    const u32 slot = ((qc & 1u) << 1) | (qc >> 1);
    return (k & ~31u) + ((q & 3u) << 3) + (slot << 1) + (k & 1u);
}
// 8 consecutive activations (k = 8 u .. 8 u + 7, packed as 4 words) -> the permuted copy (4 ds_write_b32)
__device__ __forceinline__ void qtip_store_x8(uint16_t *xsp, u32 u, const u32 (&o)[4]) {
    u32 *dst = reinterpret_cast<u32 *>(xsp) + (u >> 2) * 16u;
    const u32 qc = u & 3u, slot = ((qc & 1u) << 1) | (qc >> 1);
#pragma unroll
    for (u32 e = 0; e < 4; e++) dst[e * 4u + slot] = o[e];
}

struct QtItem {
    u32 boff;    // byte offset of the band inside the linear's trellis tensor
    float *out;  // 32 sums
    u32 k2lo, k2hi;
};
struct QtipNoop {
    __device__ __forceinline__ void operator()() const {}
};

// The trellis stream is requested in CHUNKS of CH consecutive tile blocks, one wave-instruction of LW bytes per lane each
// (R = 2: 4 tile blocks = 64 x 16 bytes; R = 3: 2 = 64 x 12; R = 4: 2 = 64 x 16).  Measured (tools/ubench/qtip_stream.hip):
// 8 bytes per lane -- one tile block per instruction, what this kernel did before -- streams the 12.6 MB of q/k/v in 6.1 us,
// 16 bytes per lane in 3.9 us, whatever the queue depth.  The loaded rows are then in "loader order" (lane = tile block x
// row of units); a 1 KiB LDS slot per wave turns them into the MFMA lane order: the loader lane stores its 16 bytes (R = 2:
// already as the four {column tile 1 : column tile 0} packs of its two unit rows and both 16-row halves), each consumer lane
// reads its own and its neighbour's (one dword each at R = 2).  LDS operations of ONE wave are served in order, so the slot
// needs no barrier and no second buffer: the reads of a chunk are issued before the store of the next one.
template <int R>
struct QtChunk {
    static constexpr u32 CH = R == 2 ? 4u : 2u;
    static constexpr u32 LD = R == 3 ? 3u : 4u;          // dwords per lane and load
    static constexpr u32 BYTES = CH * 128u * (u32)R;     // = 64 * 4 * LD
};

template <int R>
__device__ __forceinline__ void qtip_load_chunk(u32 (&d)[QtChunk<R>::LD], __amdgpu_buffer_rsrc_t rs, u32 voff, u32 soff) {
    if constexpr (R == 3) {
        const auto v = __builtin_amdgcn_raw_buffer_load_b96(rs, voff, soff, 2 /* nt */);
        d[0] = v[0], d[1] = v[1], d[2] = v[2];
    } else {
        const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 2);
        d[0] = v[0], d[1] = v[1], d[2] = v[2], d[3] = v[3];
    }
}

// comp / comp_bytes: the trellis tensor the items of this block index into (one linear per block).
// stg: [waves] 1 KiB slots.  Items' K ranges start at multiples of CH tile blocks.
// WC: waves per block if known at compile time (16: the activation reads of a loop use immediate offsets), 0: blockDim / 64.
template <int R, int SX, int WC, class ItemFn, class F>
__device__ __forceinline__ void qtip_engine(const u32 *tab, const uint16_t *xsp, float *part, unsigned char *stg, const u32 *comp, u32 comp_bytes,
                                            u32 nitems, ItemFn item_of, F between, unsigned long long *dbg = nullptr) {
    constexpr u32 CH = QtChunk<R>::CH, LD = QtChunk<R>::LD, CB = QtChunk<R>::BYTES;
    const u32 T = blockDim.x, tid = threadIdx.x, W = WC ? (u32)WC : T >> 6, l = tid & 63u;
    const u32 w = (u32)__builtin_amdgcn_readfirstlane((int)(tid >> 6));  // (wave-uniform: the loop tests stay on the scalar unit)
    const u32 g = l >> 4, a4 = (l >> 3) & 1u, a = l & 7u, s = 4u * a + g;
    const u32 sn = (s + 1u) & 31u;  // the window of unit s ends in unit s + 1 of the same tile (cyclic)
    const u32 lane4 = (l & 31u) * 4u, voff = l * (4u * LD);
    const u32 sel3a = a4 ? 0x0C050403u : 0x0C020100u, sel3b = a4 ? 0x0C070605u : 0x0C040302u;  // (R = 3 unit selectors)
    const unsigned char *tabb = reinterpret_cast<const unsigned char *>(tab);
    unsigned char *slot = stg + w * 1024u;
    // R = 2: the slot holds, per loader lane (tile block tb = lane / 16, unit rows 2 j, 2 j + 1), 16 bytes
    // {pack(2j, half 0), pack(2j + 1, half 0), pack(2j, half 1), pack(2j + 1, half 1)} at 256 tb + 16 j
    // R = 4: raw rows of four dword units; this lane wants units a4 and 2 + a4 (column tiles 0 / 1) of its row and the neighbour's
    const unsigned char *own_p = slot + (R == 2 ? 16u * (s >> 1) + 8u * a4 + 4u * (s & 1u) : s * (4u * R) + (R == 4 ? 4u * a4 : 0u));
    const unsigned char *nbr_p = slot + (R == 2 ? 16u * (sn >> 1) + 8u * a4 + 4u * (sn & 1u) : sn * (4u * R) + (R == 4 ? 4u * a4 : 0u));
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)comp, 0, (int)comp_bytes, 0x00020000);
#ifndef QT_PF
#define QT_PF 2
#endif
    constexpr u32 PF = QT_PF;  // chunks per wave in flight
    u32 dq[PF][LD];
    u32 nstamp = 0;
    // (QT_SB: bit i = a scheduling barrier at stamp site i of a build without the stamps -- round 6: the library with the stamp sites
    // compiled out ran the bare matvec 4-5 % SLOWER than the one with them, 9.2 vs 8.8 us at 11008 x 4096: a site is a conditional
    // branch, which the compiler does not move code across)
#ifndef QT_SB
#define QT_SB 0
#endif
// QT_ENGINE_STAMPS = 1 (shipped): the five stamp sites of the band engine stay compiled IN (switched off at run time by the null debug
// pointer) while every other kernel's sites stay out.  Bisected in round 6 (profiles/r06_qtip_matvec_regression.txt): round 5's
// GQ_STAMPS = 0 build of this file ran the bare matvec 4-5 % slower than round 4's (4096^2 6.1 vs 5.8 us, 11008 x 4096 9.3 vs 8.85,
// 4096 x 11008 11.1 vs 10.8; same box, alternating) -- the only change in the kernel; with these five sites back the times are round
// 4's again and the QTIP decode keeps round 5's gain (434 tokens/s; all sites of the file back in: 427).  Scheduling barriers at the
// same places (QT_SB) do not reproduce it: the branch around a site ends a basic block, which changes what the scheduler hoists.
#ifndef QT_ENGINE_STAMPS
#define QT_ENGINE_STAMPS 1
#endif
    auto stamp = [&](auto SITE) {
        if ((GQ_STAMPS || QT_ENGINE_STAMPS) && dbg && blockIdx.x == gridDim.x / 2 && l == 0 && nstamp < 8u) dbg[w * 8u + nstamp++] = __builtin_readcyclecounter();
        if constexpr (!(GQ_STAMPS || QT_ENGINE_STAMPS) && ((QT_SB >> decltype(SITE)::value) & 1)) __builtin_amdgcn_sched_barrier(0);
    };
    stamp(std::integral_constant<int, 0>{});
    u32 j = blockIdx.x;
    if (j >= nitems) return;  // (whole block)
    u32 ord = 0;  // ordinal of the block's current item (item_of takes it: no division to recover it from j)
    QtItem cur = item_of(j, 0u);
    // Queue refill without vector ALU work: the lane part of the address is a constant VGPR, the chunk a scalar offset.
    // A slot with nothing left to fetch re-reads the last chunk of the item (a cache hit, never used).
    auto fetch = [&](u32 p, const QtItem &it, u32 c) {
        const u32 chi = (it.k2hi + CH - 1u) / CH, cc = c < chi ? c : chi - 1u;
        qtip_load_chunk<R>(dq[p], rs, voff, it.boff + cc * CB);
    };
#pragma unroll
    for (u32 p = 0; p < PF; p++) fetch(p, cur, cur.k2lo / CH + w + p * W);
    stamp(std::integral_constant<int, 1>{});
    between();
    stamp(std::integral_constant<int, 2>{});
    f32x4 accA[2], accB[2];  // even / odd tile blocks: no MFMA waits for the one before it
    const u32 emask = SX ? 0xFFC0FFC0u : 0x7FC07FC0u;
    auto lookup2 = [&](u32 P, u32 &wlo, u32 &whi) {
        u32 m;
        u32 alo, ahi;
#if QT_ABL & 16
        m = P;
        alo = (P & 0x7F80u) | lane4;
        ahi = alo + 128u;
#else
        asm("v_pk_mad_u16 %0, %1, %1, %1" : "=v"(m) : "v"(P));
        const u32 y = m & emask;
        // byte address = 2 * (entry << 6) + 4 * (lane % 32), per 16-bit half (v_mad_u32_u16: half select, times 2, plus lane)
        asm("v_mad_u32_u16 %0, %1, 2, %2 op_sel:[0,0,0,0]" : "=v"(alo) : "v"(y), "v"(lane4));
        asm("v_mad_u32_u16 %0, %1, 2, %2 op_sel:[1,0,0,0]" : "=v"(ahi) : "v"(y), "v"(lane4));
#endif
#if QT_ABL & 1
        wlo = alo, whi = ahi;
#else
        wlo = *reinterpret_cast<const u32 *>(tabb + alo);
        whi = *reinterpret_cast<const u32 *>(tabb + ahi);
#endif
        if (!SX) {
            const u32 sg = m & 0x80008000u;
            wlo ^= sg & 0xFFFFu;
            whi ^= sg >> 16;
        }
    };
    // A chunk and its tile blocks in stages, so that the LDS round trips of several tile blocks are in flight:
    //   chunk_in: the loaded registers -> the wave's LDS slot (the queue slot is refilled right away);
    //   stage_a : this lane's units and its neighbour's (both column tiles) out of the slot;
    //   stage_b : the four state-pair registers -> 8 codebook lookups + the activations (ds_read);
    //   stage_c : two MFMAs.
    struct TbA {
        u32 own[R == 2 ? 1 : 2], nbr[R == 2 ? 1 : 2];
    };
    struct TbB {
        u32x4 wa, wb, xb;
    };
    auto chunk_store = [&](const u32 (&d)[LD]) {
        if constexpr (R == 2) {
            const u32x4 q = {__builtin_amdgcn_perm(d[1], d[0], 0x05040100u), __builtin_amdgcn_perm(d[3], d[2], 0x05040100u),
                             __builtin_amdgcn_perm(d[1], d[0], 0x07060302u), __builtin_amdgcn_perm(d[3], d[2], 0x07060302u)};
            *reinterpret_cast<u32x4 *>(slot + l * 16u) = q;
        } else {
#pragma unroll
            for (u32 i = 0; i < LD; i++) reinterpret_cast<u32 *>(slot + l * (4u * LD))[i] = d[i];
        }
    };
    auto stage_a = [&](u32 tbk, TbA &t) {
        if constexpr (R == 2) {
            t.own[0] = *reinterpret_cast<const u32 *>(own_p + tbk * 256u);
            t.nbr[0] = *reinterpret_cast<const u32 *>(nbr_p + tbk * 256u);
        } else if constexpr (R == 4) {
            const u32 *po = reinterpret_cast<const u32 *>(own_p + tbk * 512u), *pn = reinterpret_cast<const u32 *>(nbr_p + tbk * 512u);
            t.own[0] = po[0], t.own[1] = po[2], t.nbr[0] = pn[0], t.nbr[1] = pn[2];  // (one ds_read2_b32 each)
        } else {
            // R = 3: a row is 12 bytes, unit j its bytes 3j .. 3j + 2; this lane wants units a4 and 2 + a4: one v_perm each with a
            // per-lane selector (0x0C = zero byte)
            u32 d0[R], d1[R];
#pragma unroll
            for (int i = 0; i < R; i++) {
                d0[i] = reinterpret_cast<const u32 *>(own_p + tbk * (128u * R))[i];
                d1[i] = reinterpret_cast<const u32 *>(nbr_p + tbk * (128u * R))[i];
            }
            t.own[0] = __builtin_amdgcn_perm(d0[1], d0[0], sel3a), t.own[1] = __builtin_amdgcn_perm(d0[2], d0[1], sel3b);
            t.nbr[0] = __builtin_amdgcn_perm(d1[1], d1[0], sel3a), t.nbr[1] = __builtin_amdgcn_perm(d1[2], d1[1], sel3b);
        }
    };
    auto stage_b = [&](const TbA &t, const uint16_t *xrow, TbB &o) {
        u32 P[4];
        if constexpr (R == 2) {
            // windows at bit offsets 4 i of {pack : npack}, both halves at once: (pack << 4 i) | (npack >> (16 - 4 i)) under a
            // per-half mask (one 3-input bit operation; shifts are 4-byte encodings: half the issue cost of VOP3 forms)
            const u32 pack = t.own[0], npack = t.nbr[0];
            P[0] = pack;
            P[1] = ((pack << 4) & 0xFFF0FFF0u) | ((npack >> 12) & 0x000F000Fu);
            P[2] = __builtin_amdgcn_perm(pack, npack, 0x06030401u);
            P[3] = ((pack << 12) & 0xF000F000u) | ((npack >> 4) & 0x0FFF0FFFu);
        } else if constexpr (R == 4) {
            // byte-aligned windows of {unit : neighbour's unit}: states 0..2 inside the unit, state 3 = its last byte + the
            // neighbour's first: six v_perm for both column tiles
            const u32 u0 = t.own[0], u1 = t.own[1];
            P[0] = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
            P[1] = __builtin_amdgcn_perm(u1, u0, 0x06050201u);
            P[2] = __builtin_amdgcn_perm(u1, u0, 0x05040100u);
            const u32 lo = __builtin_amdgcn_perm(u1, u0, 0x04040000u), hi = __builtin_amdgcn_perm(t.nbr[1], t.nbr[0], 0x07070303u);
            P[3] = __builtin_amdgcn_perm(lo, hi, 0x07020500u);
        } else {
            // R = 3: the 48-bit window {unit : neighbour's unit}; V = its bits 47..16, V2 = bits 39..8 (one v_perm each); the states
            // are bits 47..32, 41..26, 35..20, 29..14 = V[31:16], V[25:10], V[19:4], V2[21:6]: a shift per column tile and a
            // v_perm that pairs the two 16-bit results
            const u32 V0 = __builtin_amdgcn_perm(t.own[0], t.nbr[0], 0x06050402u), V1 = __builtin_amdgcn_perm(t.own[1], t.nbr[1], 0x06050402u);
            const u32 W0 = __builtin_amdgcn_perm(t.own[0], t.nbr[0], 0x05040201u), W1 = __builtin_amdgcn_perm(t.own[1], t.nbr[1], 0x05040201u);
            P[0] = __builtin_amdgcn_perm(V1, V0, 0x07060302u);
            P[1] = __builtin_amdgcn_perm(V1 >> 10, V0 >> 10, 0x05040100u);
            P[2] = __builtin_amdgcn_perm(V1 >> 4, V0 >> 4, 0x05040100u);
            P[3] = __builtin_amdgcn_perm(W1 >> 6, W0 >> 6, 0x05040100u);
        }
        u32 t0, t1;
        lookup2(P[0], t0, t1);
        o.wa[0] = t0, o.wa[1] = t1;
        lookup2(P[2], t0, t1);
        o.wa[2] = t0, o.wa[3] = t1;
        lookup2(P[1], t0, t1);
        o.wb[0] = t0, o.wb[1] = t1;
        lookup2(P[3], t0, t1);
        o.wb[2] = t0, o.wb[3] = t1;
#if QT_ABL & 8
        o.xb = o.wa;
#else
        o.xb = *reinterpret_cast<const u32x4 *>(xrow);
#endif
    };
    auto stage_c = [&](const TbB &o, u32 par) {
#if QT_ABL & 2
        accA[par][0] += __builtin_bit_cast(float, o.wa[0] ^ o.wa[1] ^ o.wa[2] ^ o.wa[3] ^ o.xb[0]);
        accB[par][0] += __builtin_bit_cast(float, o.wb[0] ^ o.wb[1] ^ o.wb[2] ^ o.wb[3] ^ o.xb[1]);
        return;
#endif
        accA[par] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, o.wa), __builtin_bit_cast(h16x8, o.xb), accA[par], 0, 0, 0);
        accB[par] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, o.wb), __builtin_bit_cast(h16x8, o.xb), accB[par], 0, 0, 0);
    };
    // this lane group's 8 activations of tile block K2: xg + 32 K2.  QT_XOOB: only column 0 of the B operand carries them; the
    // lanes of columns 1..15 read from 192 KiB above -- outside the LDS allocation, where ds_read returns zeros (tools/ubench/
    // lds_oob.hip): 15 equal copies of the activations in the matrix cores cost power, and with it clock (ap_plane.hip, PL_BOOB)
#if QT_XOOB
    const uint16_t *xg = xsp + 8u * g + ((l & 15u) ? 0x18000u : 0u);
#else
    const uint16_t *xg = xsp + 8u * g;
#endif
    u32 flip = 0;
    for (;;) {
        const u32 jn = j + gridDim.x;
        const bool has_next = jn < nitems;
        const QtItem nxt = has_next ? item_of(jn, ord + 1u) : cur;
        const u32 chi = (cur.k2hi + CH - 1u) / CH, first = cur.k2lo / CH + w;
        const u32 cnt = first < chi ? (chi - first + W - 1u) / W : 0u;  // chunks of this wave in this item
        const u32 nl = (cnt + PF - 1u) / PF;
#pragma unroll
        for (u32 q = 0; q < 2; q++) accA[q] = f32x4{0.f, 0.f, 0.f, 0.f}, accB[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (u32 i = 0; i < nl; i++) {
            const u32 cb = first + i * PF * W;  // chunk of slot 0 in this loop; slot p: cb + p W
            const bool lastloop = i + 1u == nl;
            // what the slots are refilled with: the next loop of this item, or the first loop of the block's next item
            const QtItem &ft = lastloop ? nxt : cur;
            const u32 fc = lastloop ? nxt.k2lo / CH + w : cb + PF * W;
            const uint16_t *xrow = xg + 32u * CH * cb;
            // slots p < nv hold chunks of this item (nv = PF except in its last loop); tile block k of slot p exists if
            // CH (cb + p W) + k < k2hi (a K range may end inside its last chunk).  The stages of consecutive tile blocks overlap:
            // the lookups of tile block t + 1 are issued before the MFMAs of t, a chunk enters its LDS slot ahead of that.
            const u32 rem = cnt - i * PF, nv = rem < PF ? rem : PF;
            auto exists = [&](u32 p, u32 k) { return p < nv && CH * (cb + p * W) + k < cur.k2hi; };
            TbA ta[2][CH];
            TbB tb[2];
            auto chunk_in = [&](u32 p) {
                if (p < nv) chunk_store(dq[p]);
                fetch(p, ft, fc + p * W);
                if (p < nv) {
#pragma unroll
                    for (u32 k = 0; k < CH; k++) stage_a(k, ta[p & 1u][k]);
                }
            };
            chunk_in(0);
            stage_b(ta[0][0], xrow, tb[0]);
#pragma unroll
            for (u32 t = 0; t < PF * CH; t++) {
                const u32 p = t / CH, k = t % CH, tn = t + 1u, pn = tn / CH, kn = tn % CH;
                if (tn < PF * CH) {
                    if (kn == 0u) chunk_in(pn);
                    if (exists(pn, kn)) stage_b(ta[pn & 1u][kn], xrow + 32u * (CH * pn * W + kn), tb[tn & 1u]);
                }
                if (exists(p, k)) stage_c(tb[t & 1u], t & 1u);
            }
        }
        if (nl == 0u && has_next) {  // (a wave without chunks in this item still feeds its queue)
#pragma unroll
            for (u32 p = 0; p < PF; p++) fetch(p, nxt, nxt.k2lo / CH + w + p * W);
        }
        stamp(std::integral_constant<int, 3>{});
        // D layout: lane (gq = l / 16, n = l % 16) holds logical rows 4 gq + 0..3 of column n (all columns are equal):
        // lanes n < 4 store element n of "rows a", lanes 4 <= n < 8 element n - 4 of "rows a + 8"
        {
            const f32x4 sa = accA[0] + accA[1], sb = accB[0] + accB[1];
            float *pw = part + flip * (W * 32u) + w * 32u;
#if QT_XOOB
            if ((l & 15u) == 0u) {  // column 0 holds the sums: its lane of each group stores the 4 + 4 rows
#pragma unroll
                for (u32 e = 0; e < 4u; e++) {
                    const u32 r = 4u * g + e;
                    pw[16u * (r >> 3) + (r & 7u)] = sa[e];
                    pw[16u * (r >> 3) + (r & 7u) + 8u] = sb[e];
                }
            }
#else
            const u32 n = l & 15u, e = n & 3u, r = 4u * g + e;
            const f32x4 src = (n & 4u) ? sb : sa;
            const float v = e == 0u ? src[0] : (e == 1u ? src[1] : (e == 2u ? src[2] : src[3]));
            if (n < 8u) pw[16u * (r >> 3) + (r & 7u) + 8u * (n >> 2)] = v;
#endif
        }
        __syncthreads();
        if (tid < 32u) {
            const float *pr = part + flip * (W * 32u);
            float t = 0.f;
            for (u32 i = 0; i < W; i++) t += pr[i * 32u + tid];
            cur.out[tid] = t;
        }
        stamp(std::integral_constant<int, 4>{});
        flip ^= 1u;
        if (!has_next) break;
        cur = nxt;
        j = jn;
        ord++;
    }
}

template <int R, int SX, int WC>
__global__ void __launch_bounds__(1024) qtip_matvec_kernel(float *out, const u32 *comp, const uint16_t *x, const uint16_t *tlut,
                                                          u32 M, u32 K, unsigned long long *dbg) {
    __shared__ __attribute__((aligned(128))) u32 tab[QtipTab<SX>::WORDS];
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t *xsp = reinterpret_cast<uint16_t *>(smem);          // [K] permuted copy
    float *part = reinterpret_cast<float *>(xsp + K);            // [2][waves][32 rows]
    unsigned char *stg = reinterpret_cast<unsigned char *>(part + 2u * (blockDim.x >> 6) * 32u);  // [waves] 1 KiB chunk slots
    const u32 T = blockDim.x, tid = threadIdx.x, nK2 = K / 32u;
    auto item_of = [&](u32 j, u32) {
        return QtItem{j * nK2 * 128u * R, out + (size_t)j * 32u, 0u, nK2};
    };
    // codebook words and activations first, the tile blocks behind them (in-order return)
    QtipTabRegs tr;
    qtip_table_request(tr, tlut);
    constexpr u32 NX = 2;  // 8-element units per thread held in registers (K <= 16 T)
    const bool xin = K <= 8u * NX * T;
    uint4 xq[NX];
    if (xin) {
#pragma unroll
        for (u32 k = 0; k < NX; k++) {
            const u32 u = tid + k * T;
            xq[k] = u < K / 8u ? reinterpret_cast<const uint4 *>(x)[u] : make_uint4(0u, 0u, 0u, 0u);
        }
    }
    auto prologue = [&]() {
        qtip_fill_table<SX>(tab, tlut, tr);
        if (xin) {
#pragma unroll
            for (u32 k = 0; k < NX; k++) {
                const u32 u = tid + k * T;
                const u32 o[4] = {xq[k].x, xq[k].y, xq[k].z, xq[k].w};
                if (u < K / 8u) qtip_store_x8(xsp, u, o);
            }
        } else {
            for (u32 u = tid; u < K / 8u; u += T) {
                const uint4 q = reinterpret_cast<const uint4 *>(x)[u];
                const u32 o[4] = {q.x, q.y, q.z, q.w};
                qtip_store_x8(xsp, u, o);
            }
        }
        __syncthreads();
    };
    qtip_engine<R, SX, WC>(tab, xsp, part, stg, comp, (M / 32u) * nK2 * 128u * R, M / 32u, item_of, prologue, dbg);
}

// ------------------------------------------------------------------------------------------------ fused QTIP linear
// BitshiftLinear.forward (inference/lib/codebook/bitshift.py:415-472, eval, one row, no tensor-parallel Hadamard split):
//   x32 = x.float() * SU;  x = hadamard(x32) * n^-1/2 / 32;  y = decompress_matvec(trellis, x.half(), tlut)       [kernel A]
//   y = hadamard(y) * m^-1/2;  out = (y * (SV * 32)).half()  (+ the residual add of the block, model.py:311-313)       [kernel B]
// Kernel A fuses what produces x in the decode step -- RMSNorm (model.py:281-292) or silu(gate) * up (model.py:266) --
// and serves up to 3 linears that share the input (q/k/v, gate/up) in one launch; every 32-row band block repeats the
// K-point transform (K log K / 2 butterflies: small next to its 32 x K trellis decode).
struct QtipIn {
    const u32 *comp;
    const float *SU;
    const uint16_t *tlut;
    float *y32;
    u32 blk0, nblk;  // the blocks [blk0, blk0 + nblk) walk the items (band, K range) of this linear
    u32 M;
};
struct QtipOut {
    const float *y32, *SV32;  // SV32 = SV * 32 (fp32)
    const uint16_t *resid;
    uint16_t *out;
    u32 M;
    float mscale;
    u32 parts;  // y32 is [parts][M] split-K partial sums, added in this order
};
struct QtipInArgs {
    const uint16_t *x, *x2, *normw;
    float eps, kscale;  // kscale = (float)K^-1/2, rounded from double like the scale argument of hadamard()
    u32 K, n;
    u32 P;       // QPRO_ROWS: length of the row transforms (K = Kf * P)
    u32 ksplit;  // 1..4 K ranges per band; range ks of a band writes its sums to y32 + ks * M
    QtipIn lin[3];
    u32 nprev;          // 1 / 2: x (and x2) are the outputs of the linears prev[] whose transform-out is done here (M == K)
    QtipOut prev[2];
    u32 xs_off, stg_off, part_off, xp_off;  // byte offsets inside the dynamic LDS (host: qtip_in_layout)
    // gq_qtip_linear: the block that finishes a linear LAST (device-scope counter) also runs its transform-out
    QtipOut fin[3];
    u32 *fin_ctr;  // [n] zero before the first launch; the finishing block resets its counter
    unsigned long long *dbg;
};
enum { QPRO_NONE = 0, QPRO_RMSNORM = 1, QPRO_SILUMUL = 2, QPRO_PRE = 3, QPRO_ROWS = 4 };

// y32 of a producing linear: its split-K parts added in ascending order (what gq_qtip_linear_out reads)
__device__ __forceinline__ float qtip_sum_parts(const float *y32, u32 M, u32 parts, u32 i) {
    float t = y32[i];
    for (u32 p = 1; p < parts; p++) t += y32[(size_t)p * M + i];
    return t;
}

// Transform-out of one linear by one block: y32 (split-K parts added in ascending order) -> hadamard * m^-1/2 -> * (SV * 32) ->
// fp16 (+ residual).  v: M floats of LDS.  Shared by gq_qtip_linear_out and the finishing block of gq_qtip_linear (same code:
// the two forms agree bit for bit).
__device__ __forceinline__ void qtip_transform_out(const QtipOut &L, float *v) {
    const u32 T = blockDim.x, tid = threadIdx.x, M = L.M;
    // 4 consecutive outputs per thread and step: 16-byte loads of the sums, the scales and (8 bytes) the residual, all
    // requested before the transform (one block: nothing else hides their latency)
    constexpr u32 PRE = 2;  // M <= 8192 at 1024 threads
    const bool vec = !(((uintptr_t)L.y32 | (uintptr_t)L.SV32) & 15u) && !(((uintptr_t)L.resid | (uintptr_t)L.out) & 7u);
    const bool pre = vec && M <= 4u * PRE * T;
    float4 svr[PRE];
    uint2 rsr[PRE];
    if (pre) {
#pragma unroll
        for (u32 k = 0; k < PRE; k++) {
            const u32 u = tid + k * T;
            const bool ok = u < M / 4u;
            float4 y = ok ? reinterpret_cast<const float4 *>(L.y32)[u] : make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok)
                for (u32 p = 1; p < L.parts; p++) {
                    const float4 y2 = reinterpret_cast<const float4 *>(L.y32 + (size_t)p * M)[u];
                    y = make_float4(y.x + y2.x, y.y + y2.y, y.z + y2.z, y.w + y2.w);
                }
            if (ok) reinterpret_cast<float4 *>(v)[u] = y;
            svr[k] = ok ? reinterpret_cast<const float4 *>(L.SV32)[u] : make_float4(0.f, 0.f, 0.f, 0.f);
            rsr[k] = (ok && L.resid) ? reinterpret_cast<const uint2 *>(L.resid)[u] : make_uint2(0u, 0u);
        }
    } else {
        for (u32 i = tid; i < M; i += T) v[i] = qtip_sum_parts(L.y32, M, L.parts, i);
    }
    __syncthreads();
    fwht_lds(v, M);
    const float sc = L.mscale;
    if (pre) {
#pragma unroll
        for (u32 k = 0; k < PRE; k++) {
            const u32 u = tid + k * T;
            if (u < M / 4u) {
                const float4 f = reinterpret_cast<const float4 *>(v)[u];
                const float fv[4] = {f.x, f.y, f.z, f.w}, sv[4] = {svr[k].x, svr[k].y, svr[k].z, svr[k].w};
                const u32 rw[2] = {rsr[k].x, rsr[k].y};
                uint16_t o[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    h16 y = (h16)gq_pin_f32((fv[e] * sc) * sv[e]);
                    if (L.resid) y = __builtin_bit_cast(h16, (uint16_t)(rw[e >> 1] >> (16 * (e & 1)))) + y;
                    o[e] = __builtin_bit_cast(uint16_t, y);
                }
                reinterpret_cast<uint2 *>(L.out)[u] = make_uint2((u32)o[0] | ((u32)o[1] << 16), (u32)o[2] | ((u32)o[3] << 16));
            }
        }
        return;
    }
    for (u32 i = tid; i < M; i += T) {
        h16 y = (h16)gq_pin_f32((v[i] * sc) * L.SV32[i]);
        if (L.resid) y = __builtin_bit_cast(h16, L.resid[i]) + y;
        L.out[i] = __builtin_bit_cast(uint16_t, y);
    }
}

// dynamic LDS (offsets from the host, qtip_in_layout): v fp32 [K] at 0, dead once the transform is done -- the chunk slots
// (and, from K = 8192 on, the permuted fp16 copy xs too) re-use it --, xs, part [2][W][32], xp fp16 [nprev][K];
// pre-transformed input: xs | part | slots
template <int R, int PRO, int SX, int WC>
__global__ void __launch_bounds__(1024) qtip_linear_in_kernel(QtipInArgs a) {
    __shared__ __attribute__((aligned(128))) u32 tab[QtipTab<SX>::WORDS];
    __shared__ float redf[17];
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const u32 K = a.K, T = blockDim.x, tid = threadIdx.x;
    float *v = reinterpret_cast<float *>(smem);                 // [K] fp32 transform buffer (none for a pre-transformed input)
    uint16_t *xs = reinterpret_cast<uint16_t *>(smem + a.xs_off);  // [K] fp16 matvec input, permuted (qtip_store_x8)
    float *part = reinterpret_cast<float *>(smem + a.part_off);    // [2][waves][32]
    unsigned char *stg = smem + a.stg_off;                          // [waves] 1 KiB chunk slots
    u32 li = 0;
    if (a.n > 1 && blockIdx.x >= a.lin[1].blk0) li = 1;
    if (a.n > 2 && blockIdx.x >= a.lin[2].blk0) li = 2;
    const QtipIn L = a.lin[li];
    // Folded transform-out of the producing linear(s): what gq_qtip_linear_out would have written is rebuilt in LDS by
    // every block (same arithmetic, same order) and stored once, by block 0, for the kernels that need it later
    // (the residual stream).  One launch and one global round trip less per linear.
    const uint16_t *xg = a.x, *x2g = a.x2;
    if (PRO == QPRO_PRE && !a.x) xg = reinterpret_cast<const uint16_t *>(L.SU);  // (one pre-transformed vector per linear: gq_qtip_linear_out_in)
    if (a.nprev) {
        uint16_t *xp = reinterpret_cast<uint16_t *>(smem + a.xp_off);  // [nprev][K] fp16
        for (u32 r = 0; r < a.nprev; r++) {
            const QtipOut P = a.prev[r];
            for (u32 i = tid; i < K; i += T) v[i] = qtip_sum_parts(P.y32, K, P.parts, i);
            __syncthreads();
            fwht_lds(v, K);
            for (u32 i = tid; i < K; i += T) {
                h16 y = (h16)gq_pin_f32((v[i] * P.mscale) * P.SV32[i]);
                if (P.resid) y = __builtin_bit_cast(h16, P.resid[i]) + y;
                xp[r * K + i] = __builtin_bit_cast(uint16_t, y);
                if (blockIdx.x == 0 && P.out) P.out[i] = __builtin_bit_cast(uint16_t, y);
            }
            __syncthreads();
        }
        xg = xp;
        x2g = xp + K;
    }
    // The input vectors are requested first (registers), the first tile blocks of the band behind them, and the prologue
    // runs while those are on their way from HBM (vector memory returns in order: requested the other way round, the
    // first use of x would wait for the tiles).
    QtipTabRegs tr;
    qtip_table_request(tr, L.tlut);
    constexpr u32 NU = 2;  // 8-element units per thread held in registers (K <= 16 T)
    const bool inreg = !a.nprev && K <= 8u * NU * T && PRO != QPRO_PRE && PRO != QPRO_ROWS &&
                       !(((uintptr_t)xg | (uintptr_t)x2g | (uintptr_t)a.normw | (uintptr_t)L.SU) & 15u);
    // the first pass of the Sylvester transform fused into the element-wise stage: units of whole waves (K / 8 a multiple of 64, or
    // fewer than 64: the first lanes of wave 0), every unit of a wave in the same loop trip, K >= 64
    const bool fused_first = inreg && K >= 64u && !(QT_ABL & 32) && ((K / 8u) % 64u == 0u || K / 8u < 64u) ;
    uint4 xq[NU], x2q[NU], nwq[NU];
    float4 su0[NU], su1[NU];
    // QPRO_ROWS: x is the fp32 vector gq_qtip_mlp_mid left -- the Kf x Kf factor product of the transform-in is done (column by
    // column, so it commutes with the row transforms); what is left is cheap enough to repeat in every block: the P-point
    // Sylvester transform of every row, * K^-1/2 / 32, fp16.  Requested with the other inputs (<= 4 x 16 bytes per thread).
    constexpr u32 NZ = 4;
    float4 zq[NZ];
    if constexpr (PRO == QPRO_ROWS) {
#pragma unroll
        for (u32 k = 0; k < NZ; k++) {
            const u32 u = tid + k * T;
            zq[k] = u < K / 4u ? reinterpret_cast<const float4 *>(xg)[u] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    if (inreg) {
#pragma unroll
        for (u32 k = 0; k < NU; k++) {
            const u32 u = tid + k * T;
            const bool ok = u < K / 8u;
            xq[k] = ok ? reinterpret_cast<const uint4 *>(xg)[u] : make_uint4(0u, 0u, 0u, 0u);
            su0[k] = ok ? reinterpret_cast<const float4 *>(L.SU)[2u * u] : make_float4(0.f, 0.f, 0.f, 0.f);
            su1[k] = ok ? reinterpret_cast<const float4 *>(L.SU)[2u * u + 1u] : make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (PRO == QPRO_RMSNORM) nwq[k] = ok ? reinterpret_cast<const uint4 *>(a.normw)[u] : make_uint4(0u, 0u, 0u, 0u);
            if constexpr (PRO == QPRO_SILUMUL) x2q[k] = ok ? reinterpret_cast<const uint4 *>(x2g)[u] : make_uint4(0u, 0u, 0u, 0u);
        }
    }
    // (prologue stamps of GQ_STAMPS builds: kept in registers and written out behind the engine -- a global store per stamp sits in vmcnt
    // and turns the next wait for a load into a wait for the store's acknowledgement, ~1,000 cycles each)
    u32 pst = 0;
    unsigned long long pstv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto pstamp = [&]() {
        if (GQ_STAMPS && a.dbg && blockIdx.x == gridDim.x / 2) {
            const unsigned long long now = __builtin_readcyclecounter();
#pragma unroll
            for (u32 i = 0; i < 8u; i++)
                if (i == pst) pstv[i] = now;
            pst++;
        }
    };
    auto prologue = [&]() {
        pstamp();
        qtip_fill_table<SX>(tab, L.tlut, tr);
        pstamp();
        if constexpr (PRO == QPRO_PRE) {  // x is the transformed fp16 input already (gq_qtip_transform): K need not be a power of two
            for (u32 u = tid; u < K / 8u; u += T) {
                const uint4 q = reinterpret_cast<const uint4 *>(xg)[u];
                const u32 o[4] = {q.x, q.y, q.z, q.w};
                qtip_store_x8(xs, u, o);
            }
            __syncthreads();
            return;
        }
        float nscale = 0.f;
        if constexpr (PRO == QPRO_RMSNORM) {
            float ss = 0.f;
            if (inreg) {
#pragma unroll
                for (u32 k = 0; k < NU; k++) {  // (units beyond K are zero)
                    const u32 w4[4] = {xq[k].x, xq[k].y, xq[k].z, xq[k].w};
#pragma unroll
                    for (int e = 0; e < 8; e++) {
                        const float f = (float)__builtin_bit_cast(h16, (uint16_t)(w4[e >> 1] >> (16 * (e & 1))));
                        ss += f * f;
                    }
                }
            } else {  // same elements per thread, same order (the two forms give the same sum bit for bit)
                for (u32 u = tid; u < K / 8u; u += T)
#pragma unroll
                    for (u32 e = 0; e < 8; e++) {
                        const float f = (float)__builtin_bit_cast(h16, xg[8u * u + e]);
                        ss += f * f;
                    }
            }
            ss = gq_wave_allsum(ss);  // (round 6: register tree instead of six ds_bpermute round trips)
            if ((tid & 63u) == 0) redf[tid >> 6] = ss;
            __syncthreads();
            {  // every wave adds the wave sums through the same tree (one LDS round trip): no second barrier, no serial thread
                const float t = gq_wave_allsum((tid & 63u) < (T >> 6) ? redf[tid & 63u] : 0.f);
                nscale = 1.0f / sqrtf(t / (float)K + a.eps);
            }
        }
        pstamp();
        auto elem = [&](uint16_t xb, uint16_t x2b, uint16_t nwb, float su) {
            h16 xh = __builtin_bit_cast(h16, xb);
            if constexpr (PRO == QPRO_RMSNORM) xh = (h16)gq_pin_f32((float)xh * nscale) * __builtin_bit_cast(h16, nwb);
            if constexpr (PRO == QPRO_SILUMUL) {
                const float g = (float)xh;
                xh = (h16)(g / (1.0f + __expf(-g))) * __builtin_bit_cast(h16, x2b);
            }
            return (float)xh * su;
        };
        if constexpr (PRO == QPRO_ROWS) {
#pragma unroll
            for (u32 k = 0; k < NZ; k++) {
                const u32 u = tid + k * T;
                if (u < K / 4u) reinterpret_cast<float4 *>(v)[u] = zq[k];
            }
            for (u32 u = tid + NZ * T; u < K / 4u; u += T) reinterpret_cast<float4 *>(v)[u] = reinterpret_cast<const float4 *>(xg)[u];
        } else if (inreg) {
#pragma unroll
            for (u32 k = 0; k < NU; k++) {
                const u32 u = tid + k * T;
                if (u < K / 8u) {
                    const u32 xw[4] = {xq[k].x, xq[k].y, xq[k].z, xq[k].w};
                    const u32 x2w[4] = {PRO == QPRO_SILUMUL ? x2q[k].x : 0u, PRO == QPRO_SILUMUL ? x2q[k].y : 0u, PRO == QPRO_SILUMUL ? x2q[k].z : 0u,
                                        PRO == QPRO_SILUMUL ? x2q[k].w : 0u};
                    const u32 nw[4] = {PRO == QPRO_RMSNORM ? nwq[k].x : 0u, PRO == QPRO_RMSNORM ? nwq[k].y : 0u, PRO == QPRO_RMSNORM ? nwq[k].z : 0u,
                                       PRO == QPRO_RMSNORM ? nwq[k].w : 0u};
                    const float su[8] = {su0[k].x, su0[k].y, su0[k].z, su0[k].w, su1[k].x, su1[k].y, su1[k].z, su1[k].w};
                    float o[8];
#pragma unroll
                    for (int e = 0; e < 8; e++)
                        o[e] = elem((uint16_t)(xw[e >> 1] >> (16 * (e & 1))), (uint16_t)(x2w[e >> 1] >> (16 * (e & 1))),
                                    (uint16_t)(nw[e >> 1] >> (16 * (e & 1))), su[e]);
                    // (round 6) unit u IS work item u of the transform's first pass: its register / cross-lane stages run on the
                    // values where they are -- one LDS round trip and one barrier less than store, barrier, fwht_first_pass
                    if (fused_first) gq_fwht::fwht_item_stages<8u>(o, u, gq_fwht::fwht_mmax<8u>(K));
                    reinterpret_cast<float4 *>(v)[2u * u] = make_float4(o[0], o[1], o[2], o[3]);
                    reinterpret_cast<float4 *>(v)[2u * u + 1u] = make_float4(o[4], o[5], o[6], o[7]);
                }
            }
        } else {
            for (u32 i = tid; i < K; i += T)
                v[i] = elem(xg[i], PRO == QPRO_SILUMUL ? x2g[i] : (uint16_t)0, PRO == QPRO_RMSNORM ? a.normw[i] : (uint16_t)0, L.SU[i]);
        }
        __syncthreads();
        pstamp();
#if !(QT_ABL & 32)
        {
            const u32 Pt = PRO == QPRO_ROWS ? a.P : K;
            const u32 hnext = fused_first ? 16u * gq_fwht::fwht_mmax<8u>(K) : gq_fwht::fwht_first_pass(v, K, Pt);
            if (GQ_STAMPS) pstamp();
            gq_fwht::fwht_rest(v, K, Pt, hnext);
        }
#endif
        pstamp();
        // fp32 -> fp16, IN PLACE (xs is the front half of v): every thread takes its (<= NU) units of 8 values into
        // registers, one barrier, then the permuted stores (the host checks K <= 16 T)
        const float sc = a.kscale;
        u32 o[NU][4];
#pragma unroll
        for (u32 k = 0; k < NU; k++) {
            const u32 u = tid + k * T;
            if (u < K / 8u) {
                const float4 f0 = reinterpret_cast<const float4 *>(v)[2u * u], f1 = reinterpret_cast<const float4 *>(v)[2u * u + 1u];
                const float f[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
#pragma unroll
                for (int e = 0; e < 4; e++)
                    o[k][e] = (u32)__builtin_bit_cast(uint16_t, (h16)((f[2 * e] * sc) / 32.0f)) |
                              ((u32)__builtin_bit_cast(uint16_t, (h16)((f[2 * e + 1] * sc) / 32.0f)) << 16);
            }
        }
        __syncthreads();
#pragma unroll
        for (u32 k = 0; k < NU; k++) {
            const u32 u = tid + k * T;
            if (u < K / 8u) qtip_store_x8(xs, u, o[k]);
        }
        __syncthreads();
    };
    constexpr u32 CH = QtChunk<R>::CH;
    const u32 nK2 = K / 32u, nch = (nK2 + CH - 1u) / CH, kpart = ((nch + a.ksplit - 1u) / a.ksplit) * CH, bl = blockIdx.x - L.blk0;
    // item t of this linear = (band t / ksplit, K range t % ksplit); this block takes t = bl, bl + nblk, ...
    // (no integer divide on this chip, and the item feeds the addresses of the NEXT item's first chunks at every item boundary: the
    // ordinal r comes from the engine, the K range by shifts / a constant divisor)
    auto item_of = [&](u32, u32 r) {  // item r of this block: jj = blockIdx.x + r * gridDim.x  ->  t = bl + r * nblk
        const u32 t = bl + r * L.nblk;
        u32 band, ks;
        if (a.ksplit == 1u) band = t, ks = 0u;
        else if (a.ksplit == 2u) band = t >> 1, ks = t & 1u;
        else if (a.ksplit == 4u) band = t >> 2, ks = t & 3u;
        else band = t / 3u, ks = t - 3u * band;
        const u32 lo = ks * kpart, hi = lo + kpart < nK2 ? lo + kpart : nK2;
        return QtItem{band * nK2 * 128u * R, L.y32 + (size_t)ks * L.M + (size_t)band * 32u, lo, hi};
    };
    // number of items of this block, expressed in the engine's index space (j = blockIdx.x + r * gridDim.x < nitems)
    const u32 total = (L.M / 32u) * a.ksplit, mine = bl < total ? (total - bl + L.nblk - 1u) / L.nblk : 0u;
    qtip_engine<R, SX, WC>(tab, xs, part, stg, L.comp, (L.M / 32u) * nK2 * 128u * R, mine ? blockIdx.x + (mine - 1u) * gridDim.x + 1u : 0u, item_of, prologue, a.dbg);
    if (GQ_STAMPS && a.dbg && blockIdx.x == gridDim.x / 2 && (tid & 63u) == 0) {
#pragma unroll
        for (u32 i = 0; i < 8u; i++)
            if (pstv[i]) a.dbg[128u + (tid >> 6) * 8u + i] = pstv[i];
    }
    if (a.fin_ctr) {
        // Every sum of this block is written (by wave 0, in program order before this point).  Release them to the device,
        // count the block in; the block that completes the count acquires the others' sums and transforms them.  No block
        // waits for another one: this is not a grid barrier, the last arrival simply carries on.
        __shared__ u32 is_last;
        __syncthreads();
        if (tid == 0) {
            __threadfence();
            const u32 old = atomicAdd(a.fin_ctr + li, 1u);
            is_last = old == L.nblk - 1u ? 1u : 0u;
            if (is_last) a.fin_ctr[li] = 0u;  // (nobody touches it again before the next launch)
        }
        __syncthreads();
        if (!is_last) return;
        __threadfence();
        qtip_transform_out(a.fin[li], reinterpret_cast<float *>(tab));  // the codebook is dead: its LDS holds the M floats
    }
}

struct QtipOutArgs {
    QtipOut lin[3];
};
__global__ void __launch_bounds__(1024) qtip_linear_out_kernel(QtipOutArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    qtip_transform_out(a.lin[blockIdx.x], reinterpret_cast<float *>(smem));
}

// Round 5: transform-out of ONE linear (o or down, residual added) followed by the transform-IN of the linears that consume its
// output (gate / up, or the next layer's q / k / v) -- block j of the launch repeats the transform-out (only block 0 stores the hidden
// state) and then runs RMSNorm -> * SU_j -> Hadamard -> * K^-1/2 / 32 -> fp16 for consumer j, leaving that linear's PRE-TRANSFORMED
// input in memory.  The matvec launch that follows takes it as is (GQ_QPRO_PRETRANSFORMED with one vector per linear): its 256
// blocks no longer repeat RMSNorm . SU . H each (~5 us of a 15 us launch, profiles/r04_qtip_decode_kernel_trace.txt).  Same
// element-wise operations, the same butterfly network and the same summation order of the mean of squares (1024 threads, the
// per-thread unit order of qtip_linear_in_kernel's prologue): the pre-transformed vectors are what that prologue builds.
struct QtipOutInArgs {
    QtipOut prev;
    const uint16_t *normw;
    float eps, kscale;
    const float *SU[3];
    uint16_t *xt[3];
};
__global__ void __launch_bounds__(1024) qtip_out_in_kernel(QtipOutInArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ float redf[17];
    const u32 K = a.prev.M, T = blockDim.x, tid = threadIdx.x, j = blockIdx.x;
    float *v = reinterpret_cast<float *>(smem);                       // [K] fp32 transform buffer
    uint16_t *hs = reinterpret_cast<uint16_t *>(smem + (size_t)K * 4u);  // [K] fp16: the hidden state this launch produces
    // the consumer's vectors are requested with the sums (one block: nothing else hides their latency)
    constexpr u32 NU = 2;  // 8-element units per thread (K <= 16384)
    float4 su0[NU], su1[NU];
    uint4 nwq[NU];
#pragma unroll
    for (u32 k = 0; k < NU; k++) {
        const u32 u = tid + k * T;
        const bool ok = u < K / 8u;
        su0[k] = ok ? reinterpret_cast<const float4 *>(a.SU[j])[2u * u] : make_float4(0.f, 0.f, 0.f, 0.f);
        su1[k] = ok ? reinterpret_cast<const float4 *>(a.SU[j])[2u * u + 1u] : make_float4(0.f, 0.f, 0.f, 0.f);
        nwq[k] = ok ? reinterpret_cast<const uint4 *>(a.normw)[u] : make_uint4(0u, 0u, 0u, 0u);
    }
    {   // transform-out into LDS (qtip_transform_out's arithmetic; the global store only by block 0)
        QtipOut P = a.prev;
        P.out = hs;
        qtip_transform_out(P, v);
        __syncthreads();
        if (j == 0u && a.prev.out)
            for (u32 u = tid; u < K / 4u; u += T) reinterpret_cast<uint2 *>(a.prev.out)[u] = reinterpret_cast<const uint2 *>(hs)[u];
    }
    // transform-in of consumer j: the prologue of qtip_linear_in_kernel<QPRO_RMSNORM>, unit by unit
    float ss = 0.f;
    for (u32 u = tid; u < K / 8u; u += T)
#pragma unroll
        for (u32 e = 0; e < 8; e++) {
            const float f = (float)__builtin_bit_cast(h16, hs[8u * u + e]);
            ss += f * f;
        }
    ss = gq_wave_allsum(ss);
    if ((tid & 63u) == 0) redf[tid >> 6] = ss;
    __syncthreads();
    const float t = gq_wave_allsum((tid & 63u) < (T >> 6) ? redf[tid & 63u] : 0.f);  // (as qtip_linear_in_kernel's prologue: the same tree)
    const float nscale = 1.0f / sqrtf(t / (float)K + a.eps);
#pragma unroll
    for (u32 k = 0; k < NU; k++) {
        const u32 u = tid + k * T;
        if (u < K / 8u) {
            const uint4 xq = reinterpret_cast<const uint4 *>(hs)[u];
            const u32 xw[4] = {xq.x, xq.y, xq.z, xq.w}, nw[4] = {nwq[k].x, nwq[k].y, nwq[k].z, nwq[k].w};
            const float su[8] = {su0[k].x, su0[k].y, su0[k].z, su0[k].w, su1[k].x, su1[k].y, su1[k].z, su1[k].w};
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; e++) {
                h16 xh = __builtin_bit_cast(h16, (uint16_t)(xw[e >> 1] >> (16 * (e & 1))));
                xh = (h16)gq_pin_f32((float)xh * nscale) * __builtin_bit_cast(h16, (uint16_t)(nw[e >> 1] >> (16 * (e & 1))));
                o[e] = (float)xh * su[e];
            }
            reinterpret_cast<float4 *>(v)[2u * u] = make_float4(o[0], o[1], o[2], o[3]);
            reinterpret_cast<float4 *>(v)[2u * u + 1u] = make_float4(o[4], o[5], o[6], o[7]);
        }
    }
    __syncthreads();
    fwht_lds(v, K);
    const float sc = a.kscale;
#pragma unroll
    for (u32 k = 0; k < NU; k++) {
        const u32 u = tid + k * T;
        if (u < K / 8u) {
            const float4 f0 = reinterpret_cast<const float4 *>(v)[2u * u], f1 = reinterpret_cast<const float4 *>(v)[2u * u + 1u];
            const float f[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
            u32 o[4];
#pragma unroll
            for (int e = 0; e < 4; e++)
                o[e] = (u32)__builtin_bit_cast(uint16_t, (h16)((f[2 * e] * sc) / 32.0f)) |
                       ((u32)__builtin_bit_cast(uint16_t, (h16)((f[2 * e + 1] * sc) / 32.0f)) << 16);
            reinterpret_cast<uint4 *>(a.xt[j])[u] = make_uint4(o[0], o[1], o[2], o[3]);
        }
    }
}

// The same transform-out spread over M / 128 blocks per linear (gq_qtip_linear_out_seg).  H_M = H_(M / 128) (x) H_128: block `seg`
// combines the M / 128 segments of the sums with the signs of row `seg` of the Sylvester matrix -- every thread takes 16-byte units
// of the vector, one round trip -- and runs ONE 128-point transform: no block repeats work another one does (the segment
// combination is what the last log2(M / 128) stages of the full transform do for this segment), and the launch ends after one
// load round trip and seven butterfly stages instead of twelve stages of a single block.  The additions of the full transform in
// another order: equal to qtip_transform_out up to fp32 rounding, not bit for bit (decode.hip's attention prologue does the same
// for q / k / v).
constexpr u32 QSEG = 128u, QSEG_T = 512u;
__global__ void __launch_bounds__(QSEG_T) qtip_linear_out_seg_kernel(QtipOutArgs a) {
    __shared__ __attribute__((aligned(16))) float ps[QSEG_T / (QSEG / 4u) * QSEG];  // [thread group][128] partial sums
    __shared__ __attribute__((aligned(16))) float tv[QSEG];
    const QtipOut L = a.lin[blockIdx.y];
    const u32 tid = threadIdx.x, seg = blockIdx.x, M4 = L.M / 4u;
    if (seg * QSEG >= L.M) return;
    constexpr u32 Q = QSEG / 4u, G = QSEG_T / Q;
    // scale and residual of this thread's output element: requested before the sums (nothing else hides their latency)
    float sv = 0.f;
    uint16_t rs = 0;
    if (tid < QSEG) {
        sv = L.SV32[seg * QSEG + tid];
        if (L.resid) rs = L.resid[seg * QSEG + tid];
    }
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    {
        float4 y[4];
#pragma unroll
        for (u32 k = 0; k < 4; k++) {  // M <= 8192: at most 4 units per thread, all in flight together
            const u32 u = tid + k * QSEG_T;
            y[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (u < M4) {
                float4 t = reinterpret_cast<const float4 *>(L.y32)[u];
                for (u32 p = 1; p < L.parts; p++) {  // split-K parts, ascending
                    const float4 t2 = reinterpret_cast<const float4 *>(L.y32)[(size_t)p * M4 + u];
                    t = make_float4(t.x + t2.x, t.y + t2.y, t.z + t2.z, t.w + t2.w);
                }
                y[k] = t;
            }
        }
#pragma unroll
        for (u32 k = 0; k < 4; k++) {
            const u32 c = (tid + k * QSEG_T) / Q;
            const float sg = (__builtin_popcount(c & seg) & 1) ? -1.f : 1.f;  // (units past the vector hold zeros)
            acc = make_float4(acc.x + sg * y[k].x, acc.y + sg * y[k].y, acc.z + sg * y[k].z, acc.w + sg * y[k].w);
        }
    }
    reinterpret_cast<float4 *>(ps + (size_t)(tid / Q) * QSEG)[tid % Q] = acc;
    __syncthreads();
    if (tid < QSEG) {
        float z = 0.f;
#pragma unroll
        for (u32 gi = 0; gi < G; gi++) z += ps[(size_t)gi * QSEG + tid];
        tv[tid] = z;
    }
    __syncthreads();
    fwht_lds(tv, QSEG);
    if (tid < QSEG) {
        h16 y = (h16)gq_pin_f32((tv[tid] * L.mscale) * sv);
        if (L.resid) y = __builtin_bit_cast(h16, rs) + y;
        L.out[seg * QSEG + tid] = __builtin_bit_cast(uint16_t, y);
    }
}

// ------------------------------------------------------------------------------------------------ transform with a Hadamard factor
// Widths n = Kf * P with a non-power-of-two Hadamard factor Kf (matmul_had.py:13-67; Llama-2's 11008 = 172 * 64): the
// transform is the P-point Sylvester transform of every row of the [Kf][P] view followed by hadK @ (or hadK^T @) over
// the rows (matmul_hadU / matmul_hadUt, matmul_had.py:69-94).  The Kf x Kf product is too much work to repeat in every
// band block of the matvec, so these widths run transform -> gq_qtip_matvec -> transform: this kernel is either side.
// Block b owns RB rows of the result; every block loads the vector and does the (cheap) row transforms itself.
//   IN : v = pro(x) * SU            -> rows, hadK^T -> out16 = half(val * n^-1/2 / 32)          (input of the matvec)
//   OUT: v = y32                    -> rows, hadK   -> out16 = half(val * n^-1/2 * SV32) (+ resid)
struct QtipXfLin {
    const float *y32, *vec, *hadK;  // vec = SU (IN) or SV * 32 (OUT); hadK fp32 [Kf][Kf]
    const uint16_t *resid;
    uint16_t *out;
};
struct QtipXfArgs {
    const uint16_t *x, *x2, *normw;
    QtipXfLin lin[3];   // blockIdx.y (linears that share the source / the launch)
    float eps, nscale;  // nscale = (float)n^-1/2
    u32 n, Kf, P, RB, in, pro, transpose;
};
__global__ void __launch_bounds__(1024) qtip_transform_kernel(QtipXfArgs a) {
    __shared__ float redf[17];
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const u32 n = a.n, T = blockDim.x, tid = threadIdx.x;
    const u32 P = a.P, Kf = a.Kf, RB = a.RB, r0 = blockIdx.x * RB;
    float *v = reinterpret_cast<float *>(smem);  // [n]
    float *hs = v + n;                           // [Kf][4]: element k of the (up to 4) factor rows this block multiplies with
    float *ps = hs + 4u * Kf;                    // [KQ][RB][P] partial sums of the factor product
    const QtipXfLin L = a.lin[blockIdx.y];
    // this block's rows of hadK (columns for the transposed product), requested first; k-major so that one 16-byte LDS read
    // serves the 4 rows
    for (u32 e = tid; e < 4u * Kf; e += T) {
        const u32 k = e >> 2, rr = e & 3u, kr = r0 + rr;
        hs[e] = (rr < RB && kr < Kf) ? (a.transpose ? L.hadK[(size_t)k * Kf + kr] : L.hadK[(size_t)kr * Kf + k]) : 0.f;
    }
    if (a.in) {
        float rs = 0.f;
        if (a.pro == QPRO_RMSNORM) {
            float ss = 0.f;
            for (u32 i = tid; i < n; i += T) {
                const float f = (float)__builtin_bit_cast(h16, a.x[i]);
                ss += f * f;
            }
            ss = gq_wave_allsum(ss);
            if ((tid & 63u) == 0) redf[tid >> 6] = ss;
            __syncthreads();
            rs = 1.0f / sqrtf(gq_wave_allsum((tid & 63u) < (T >> 6) ? redf[tid & 63u] : 0.f) / (float)n + a.eps);
        }
        for (u32 u = tid; u < n / 8u; u += T) {  // 8 activations per 16-byte load
            const uint4 xq = reinterpret_cast<const uint4 *>(a.x)[u];
            uint4 x2q = make_uint4(0u, 0u, 0u, 0u), nwq = x2q;
            if (a.pro == QPRO_SILUMUL) x2q = reinterpret_cast<const uint4 *>(a.x2)[u];
            if (a.pro == QPRO_RMSNORM) nwq = reinterpret_cast<const uint4 *>(a.normw)[u];
            const float4 s0 = reinterpret_cast<const float4 *>(L.vec)[2u * u], s1 = reinterpret_cast<const float4 *>(L.vec)[2u * u + 1u];
            const u32 xw[4] = {xq.x, xq.y, xq.z, xq.w}, x2w[4] = {x2q.x, x2q.y, x2q.z, x2q.w}, nw[4] = {nwq.x, nwq.y, nwq.z, nwq.w};
            const float su[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; e++) {
                h16 xh = __builtin_bit_cast(h16, (uint16_t)(xw[e >> 1] >> (16 * (e & 1))));
                if (a.pro == QPRO_RMSNORM)
                    xh = (h16)gq_pin_f32((float)xh * rs) * __builtin_bit_cast(h16, (uint16_t)(nw[e >> 1] >> (16 * (e & 1))));
                if (a.pro == QPRO_SILUMUL) {
                    const float g = (float)xh;
                    xh = (h16)(g / (1.0f + __expf(-g))) * __builtin_bit_cast(h16, (uint16_t)(x2w[e >> 1] >> (16 * (e & 1))));
                }
                o[e] = (float)xh * su[e];
            }
            reinterpret_cast<float4 *>(v)[2u * u] = make_float4(o[0], o[1], o[2], o[3]);
            reinterpret_cast<float4 *>(v)[2u * u + 1u] = make_float4(o[4], o[5], o[6], o[7]);
        }
    } else {
        for (u32 i = tid; i < n / 4u; i += T) reinterpret_cast<float4 *>(v)[i] = reinterpret_cast<const float4 *>(L.y32)[i];
    }
    __syncthreads();
    fwht_lds(v, n, P);  // every row of the [Kf][P] view
    // (hadamard() scales by n^-1/2 before the factor product, matmul_had.py:88-90; here the factor entries are +-1, the
    // scale is applied to the sum -- one rounding less, one pass over the vector less)
    // factor product: thread (p, kq) adds its share of the k range into the (up to 4) rows of column p -- one LDS read of
    // the activation and one 16-byte read of the 4 factor entries per 4 multiply-adds --; the KQ partial sums are added in
    // a fixed order
    const u32 KQ = T >= P ? T / P : 1u, kper = (Kf + KQ - 1u) / KQ, NO = RB * P;
    for (u32 pb = 0; pb < P; pb += T) {
        const u32 p = pb + (T >= P ? tid % P : tid), kq = T >= P ? tid / P : 0u;
        if (p < P && kq < KQ) {
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            const u32 k1 = min((kq + 1u) * kper, Kf);
            for (u32 k = kq * kper; k < k1; k++) {
                const float4 h4 = *reinterpret_cast<const float4 *>(hs + 4u * k);
                const float xv = v[k * P + p];
                acc[0] += h4.x * xv;
                acc[1] += h4.y * xv;
                acc[2] += h4.z * xv;
                acc[3] += h4.w * xv;
            }
            for (u32 rr = 0; rr < RB; rr++) ps[(kq * RB + rr) * P + p] = acc[rr];
        }
    }
    __syncthreads();
    for (u32 o = tid; o < NO; o += T) {
        const u32 rr = o / P, kr = r0 + rr, p = o % P;
        if (kr >= Kf) continue;
        float acc = 0.f;
        for (u32 q = 0; q < KQ; q++) acc += ps[(q * RB + rr) * P + p];
        acc *= a.nscale;
        const u32 i = kr * P + p;
        h16 y;
        if (a.in) y = (h16)(acc / 32.0f);
        else {
            y = (h16)gq_pin_f32(acc * L.vec[i]);
            if (L.resid) y = __builtin_bit_cast(h16, L.resid[i]) + y;
        }
        L.out[i] = __builtin_bit_cast(uint16_t, y);
    }
}

// ------------------------------------------------------------------------------------------------ middle of a gated MLP with a factor width
// gate / up leave their sums y32 [n]; down wants half(H(silu(g) * u * SU) * n^-1/2 / 32) with g, u = half(H(y32) * n^-1/2 * SV32),
// n = Kf * P (Llama-2: 11008 = 172 * 64).  As two gq_qtip_transform launches (out, then in) that is 2 x 6.9 us of a 78 us layer.
// H = hadK (x) H_P acts on the [Kf][P] view as  hadK @ X @ H_P: the two sides commute.  So
//     out:  rows first, then the factor product      (the order of gq_qtip_transform: g and u come out bit-identical to it)
//     in :  the factor product first, then the rows  (the other order: equal up to fp32 rounding)
// and everything between the two row passes -- out product, scales, fp16 roundings, silu * up, SU, in product -- is local to a
// COLUMN of the view.  Block j owns column j: it computes that column of the row transforms itself (a 64-term sum in butterfly
// order per row), both factor products and the element-wise middle, and stores column j of the fp32 vector whose row transforms the
// matvec launch of down runs in its prologue (QPRO_ROWS: one in-register pass per block).  One launch of P blocks instead of two
// launches of Kf / 4 blocks that each stage the whole vector.
// Every block needs both Kf x Kf tables: they come as fp16 (+-1 is exact; 2 x 59 KB per block instead of 2 x 118 KB -- the first
// version with fp32 tables and one row per thread spent 16 us, most of it in 96 four-byte loads per thread) in [k][r] order, wave w
// owns the k group w (the 16 groups of qtip_transform_kernel) and lane L the rows 4 L .. 4 L + 3: 8-byte loads, coalesced, requested
// before anything else -- they do not depend on the previous launch and arrive while its sums are still on their way --, kept in
// registers and multiplied by v_fma_mix_f32 (fp16 x fp32 + fp32: with +-1 entries the product is exact, so the fused form
// rounds like the multiply-add of qtip_transform_kernel).
struct QtipMidArgs {
    const float *y32g, *y32u, *sv32g, *sv32u, *su_down;
    const uint16_t *hadT_out, *had_in;  // fp16 [Kf][Kf]
    float *z32;
    uint16_t *gout, *uout;  // optional fp16 copies of gate / up (null: not stored)
    u32 parts, n, Kf, kper;
    float nscale;
    unsigned long long *dbg;  // phase stamps of block 32 (tools/qtip_mid_timing.py)
};
constexpr u32 MID_R = 256u, MID_T = 1024u, MID_KP = 12u;  // Kf <= 192 (3 rounds of 64 rows in stage 1), kper = ceil(Kf / 16) <= 12
__global__ void __launch_bounds__(MID_T) qtip_mlp_mid_kernel(QtipMidArgs a) {
    // dynamic LDS: both tables (fp16 [Kf][Kf], flat copies), then xr [256][gate, up] (column j of the row transforms), ps [16][gate,
    // up][256] (partial sums of the out product; the in product's [16][256] re-use it), vin [256]
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const u32 tbytes = a.Kf * a.Kf * 2u;  // (a multiple of 16: Kf % 4 == 0)
    const unsigned char *tabO = smem, *tabI = smem + tbytes;
    float *xr = reinterpret_cast<float *>(smem + 2u * tbytes);
    float *ps = xr + MID_R * 2u, *ps2 = ps;
    float *vin = ps + 16u * 2u * MID_R;
    constexpr u32 P = 64u;
    const u32 tid = threadIdx.x, j = blockIdx.x, Kf = a.Kf, kper = a.kper;
    const u32 w = __builtin_amdgcn_readfirstlane(tid >> 6), r0 = 4u * (tid & 63u);
    u32 nst = 0;
    auto stamp = [&]() {
        if (GQ_STAMPS && a.dbg && blockIdx.x == 32u && (tid & 63u) == 0u && nst < 8u) a.dbg[w * 8u + nst++] = __builtin_readcyclecounter();
    };
    stamp();
    // ---- requests.  First what stage 1 needs (the sums of the previous launch: the long latency), then the constants.
    // A row of the view is 256 bytes: 16 lanes x 16 bytes, so a wave instruction reads 4 whole rows (one row per thread -- 128
    // bytes at a 256-byte stride -- touched 64 cache lines per instruction: 11 us for the launch).  Rounds 0..2: gate rows
    // (tid / 16) + 64 i, rounds 3..5: up (the source of a round is uniform).
    const u32 c16 = tid & 15u, rq = tid >> 4;
    u32x4 xq[6];
    {
        const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.y32g), (short)0, (int)(a.parts * a.n * 4u), 0x00020000);
        const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.y32u), (short)0, (int)(a.parts * a.n * 4u), 0x00020000);
#pragma unroll
        for (u32 i = 0; i < 3; i++) {
            const u32 row = rq + 64u * i;
            const u32 off = row < Kf ? (row * P + 4u * c16) * 4u : 0x80000000u;  // (rows beyond Kf: out of range -> zeros)
            xq[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rg, (int)off, 0, 0));
            xq[3 + i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ru, (int)off, 0, 0));
        }
    }
    // the tables: flat 16-byte units, 4 per thread and table (a wave instruction moves 1 KiB: 8 instead of 32 requests per wave)
    u32x4 tq[8];
    {
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(a.hadT_out), (short)0, (int)tbytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(a.had_in), (short)0, (int)tbytes, 0x00020000);
#pragma unroll
        for (u32 i = 0; i < 4; i++) {
            tq[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ro, (int)((tid + MID_T * i) * 16u), 0, 0));
            tq[4 + i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ri, (int)((tid + MID_T * i) * 16u), 0, 0));
        }
    }
    float svg = 0.f, svu = 0.f, sud = 0.f;
    if (tid < Kf) svg = a.sv32g[tid * P + j], svu = a.sv32u[tid * P + j], sud = a.su_down[tid * P + j];
    auto h4 = [](uint2 hw, u32 i) -> float { return (float)__builtin_bit_cast(h16, (uint16_t)((i >> 1 ? hw.y : hw.x) >> (16u * (i & 1u)))); };
    // ---- 1. column j of the row transforms: out_j = the butterfly tree of fwht_lds for output index j -- at the stage of bit m
    // the pair (lo, hi) becomes lo + hi (bit m of j clear) or lo - hi = lo + (-hi) (set): the same additions, bit for bit.
    // Bits 0, 1 inside the lane's 4 elements, bits 2..5 across the 16 lanes of the row (DPP butterflies: every lane ends with the sum)
    {
        auto flip = [](float v, u32 sm) { return __builtin_bit_cast(float, __builtin_bit_cast(u32, v) ^ sm); };
        const u32 s0 = (j & 1u) << 31, s1 = ((j >> 1) & 1u) << 31;
#pragma unroll
        for (u32 i = 0; i < 6; i++) {
            float e[4];
#pragma unroll
            for (u32 c = 0; c < 4; c++) {  // (through a scalar: this hipcc folds __builtin_bit_cast(float, vector[c]) to element 0)
                const u32 t = xq[i][c];
                e[c] = __builtin_bit_cast(float, t);
            }
            if (a.parts > 1u) {  // split-K parts, ascending (qtip_sum_parts)
                const float *src = (i < 3u ? a.y32g : a.y32u);
                const u32 row = rq + 64u * (i % 3u);
                for (u32 p = 1; p < a.parts; p++)
                    if (row < Kf) {
                        const float4 t = *reinterpret_cast<const float4 *>(src + (size_t)p * a.n + row * P + 4u * c16);
                        e[0] += t.x, e[1] += t.y, e[2] += t.z, e[3] += t.w;
                    }
            }
            float t = (e[0] + flip(e[1], s0)) + flip(e[2] + flip(e[3], s0), s1);
#pragma unroll
            for (u32 m = 2; m < 6; m++) {
                int o = __builtin_bit_cast(int, t);
                if (m == 2u) o = __builtin_amdgcn_update_dpp(0, o, 0xB1, 0xF, 0xF, false);       // lane ^ 1
                else if (m == 3u) o = __builtin_amdgcn_update_dpp(0, o, 0x4E, 0xF, 0xF, false);  // lane ^ 2
                else if (m == 4u) {                                                              // lane ^ 4 = (^ 3) then (^ 7)
                    o = __builtin_amdgcn_update_dpp(0, o, 0x1B, 0xF, 0xF, false);
                    o = __builtin_amdgcn_update_dpp(0, o, 0x141, 0xF, 0xF, false);
                } else {                                                                         // lane ^ 8 = (^ 15) then (^ 7)
                    o = __builtin_amdgcn_update_dpp(0, o, 0x140, 0xF, 0xF, false);
                    o = __builtin_amdgcn_update_dpp(0, o, 0x141, 0xF, 0xF, false);
                }
                const float of = __builtin_bit_cast(float, o);
                const bool upper = (c16 >> (m - 2u)) & 1u;
                const float lo = upper ? of : t, hi = upper ? t : of;
                t = lo + flip(hi, ((j >> m) & 1u) << 31);
            }
            const u32 row = rq + 64u * (i % 3u);
            if (c16 == 0u && row < MID_R) xr[row * 2u + (i < 3u ? 0u : 1u)] = t;  // (rows beyond Kf: zeros)
        }
    }
    stamp();
#pragma unroll
    for (u32 i = 0; i < 4; i++) {
        const u32 u = tid + MID_T * i;
        if (u * 16u < tbytes) {
            *reinterpret_cast<u32x4 *>(smem + u * 16u) = tq[i];
            *reinterpret_cast<u32x4 *>(smem + tbytes + u * 16u) = tq[4 + i];
        }
    }
    __syncthreads();
    stamp();
    // ---- 2. out product, in the order of qtip_transform_kernel at P = 64: 16 k groups of kper terms, each summed from 0.f.
    // Branch-free: every LDS read of the group is issued before the first multiply (with a branch per k the compiler waits for each
    // read where it is used: 4,200 cycles for 11 terms); a term outside the group gets the factor 0 at a clamped address -- acc + 0 * x
    {
        float2 xk[MID_KP];
        uint2 hk[MID_KP];
#pragma unroll
        for (u32 kk = 0; kk < MID_KP; kk++) {
            const u32 k = w * kper + kk;
            const bool ok = kk < kper && k < Kf;
            const u32 kc = ok ? k : 0u;
            xk[kk] = *reinterpret_cast<const float2 *>(xr + 2u * kc);
            const uint2 hw = *reinterpret_cast<const uint2 *>(tabO + (kc * Kf + r0) * 2u);  // (lanes beyond Kf: inside the allocation, unused)
            hk[kk] = (ok && r0 < Kf) ? hw : make_uint2(0u, 0u);
        }
        float ag[4] = {0.f, 0.f, 0.f, 0.f}, au[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (u32 kk = 0; kk < MID_KP; kk++)
#pragma unroll
            for (u32 i = 0; i < 4; i++) {
                const float h = h4(hk[kk], i);
                ag[i] = __builtin_fmaf(h, xk[kk].x, ag[i]);
                au[i] = __builtin_fmaf(h, xk[kk].y, au[i]);
            }
        *reinterpret_cast<float4 *>(ps + (w * 2u + 0u) * MID_R + r0) = make_float4(ag[0], ag[1], ag[2], ag[3]);
        *reinterpret_cast<float4 *>(ps + (w * 2u + 1u) * MID_R + r0) = make_float4(au[0], au[1], au[2], au[3]);
    }
    __syncthreads();
    stamp();
    // ---- 3. the element-wise middle for row r = tid of column j (bitshift.py:470, model.py:266, bitshift.py:441)
    if (tid < MID_R) {
        float pg[16], pu[16];
#pragma unroll
        for (u32 q = 0; q < 16; q++) pg[q] = ps[(q * 2u + 0u) * MID_R + tid], pu[q] = ps[(q * 2u + 1u) * MID_R + tid];
        float yg = 0.f, yu = 0.f;
#pragma unroll
        for (u32 q = 0; q < 16; q++) yg += pg[q], yu += pu[q];
        yg *= a.nscale;
        yu *= a.nscale;
        const h16 g16 = (h16)gq_pin_f32(yg * svg), u16 = (h16)gq_pin_f32(yu * svu);
        if (tid < Kf) {
            if (a.gout) a.gout[tid * P + j] = __builtin_bit_cast(uint16_t, g16);
            if (a.uout) a.uout[tid * P + j] = __builtin_bit_cast(uint16_t, u16);
        }
        const float gf = (float)g16;
        const h16 hh = (h16)(gf / (1.0f + __expf(-gf))) * u16;
        vin[tid] = tid < Kf ? (float)hh * sud : 0.f;
    }
    __syncthreads();
    stamp();
    // ---- 4. in product (hadK^T @, matmul_hadUt): z[r] = sum_k had[k][r] * vin[k], the 16 k groups added in ascending order
    {
        float xk[MID_KP];
        uint2 hk[MID_KP];
#pragma unroll
        for (u32 kk = 0; kk < MID_KP; kk++) {
            const u32 k = w * kper + kk;
            const bool ok = kk < kper && k < Kf;
            const u32 kc = ok ? k : 0u;
            xk[kk] = vin[kc];
            const uint2 hw = *reinterpret_cast<const uint2 *>(tabI + (kc * Kf + r0) * 2u);
            hk[kk] = (ok && r0 < Kf) ? hw : make_uint2(0u, 0u);
        }
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (u32 kk = 0; kk < MID_KP; kk++)
#pragma unroll
            for (u32 i = 0; i < 4; i++) acc[i] = __builtin_fmaf(h4(hk[kk], i), xk[kk], acc[i]);
        *reinterpret_cast<float4 *>(ps2 + w * MID_R + r0) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
    __syncthreads();
    stamp();
    if (tid < Kf) {
        float z = 0.f;
#pragma unroll
        for (u32 g = 0; g < 16; g++) z += ps2[g * MID_R + tid];
        a.z32[tid * P + j] = z;
    }
    stamp();
}

// ------------------------------------------------------------------------------------------------ Hadamard (FWHT)
__global__ void __launch_bounds__(1024) fwht_kernel(const float *x, float *y, u32 n, float scale) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *v = reinterpret_cast<float *>(smem);
    const u32 T = blockDim.x, tid = threadIdx.x;
    const float *xr = x + (size_t)blockIdx.x * n;
    float *yr = y + (size_t)blockIdx.x * n;
    for (u32 i = tid; i < n; i += T) v[i] = xr[i];
    __syncthreads();
    fwht_lds(v, n);
    for (u32 i = tid; i < n; i += T) yr[i] = v[i] * scale;
}
}  // namespace

namespace {
// waves per block: the chunks of every K range are dealt out to them (a range shorter than the block still works: the
// surplus waves only take part in the barriers), 16 = four per SIMD for the usual widths
u32 qtip_waves(u32 min_range) {  // min_range: chunks in the shortest K range
    u32 wv = 1;
    while (wv < 16u && wv < min_range) wv *= 2u;
    return wv;
}
constexpr size_t QT_LDS = 160u * 1024u, QT_STATIC_SLACK = 256u;  // (redf + alignment)
bool qtip_sx_fits(size_t dyn) { return dyn + (size_t)QtipTab<1>::WORDS * 4u + QT_STATIC_SLACK <= QT_LDS; }
bool qtip_fits(size_t dyn) { return dyn + (size_t)QtipTab<0>::WORDS * 4u + QT_STATIC_SLACK <= QT_LDS; }
}  // namespace

extern "C" int gq_qtip_matvec(float *out, const uint32_t *compressed, const void *x, const void *codebook, uint32_t M,
                              uint32_t K, int R, void *stream) {
    if (!out || !compressed || !x || !codebook) return gq_fail(GQ_EINVAL, "null pointer argument.");
    if (R < 2 || R > 4) return gq_fail(GQ_ENOTSUP, "R (bits per weight) must be 2, 3 or 4 (kernel_check.py:1-14).");
    if (M == 0 || K == 0 || M % 32u || K % 32u) return gq_fail(GQ_EINVAL, "M and K must be positive multiples of 32.");
    if (((uintptr_t)x | (uintptr_t)compressed | (uintptr_t)codebook) & 15u) return gq_fail(GQ_EINVAL, "buffers must be 16-byte aligned.");
    const u32 nK2 = K / 32u, ch = R == 2 ? 4u : 2u, waves = qtip_waves((nK2 + ch - 1u) / ch), bands = M / 32u;
    const size_t smem = (size_t)K * 2u + (size_t)waves * 2u * 32u * 4u + (size_t)waves * 1024u;
    if (!qtip_fits(smem)) return gq_fail(GQ_ENOTSUP, "K too large.");
    const int sx = qtip_sx_fits(smem) && gq_env_int("GQ_QTIP_SX", 1) ? 1 : 0;
    const u32 cus = (u32)gq_cu_count();
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(bands < cus ? bands : cus), block(waves * 64u);
#define GQ_LAUNCH_QTIP_W(RR, SS, WW)                                                                                  \
    do {                                                                                                              \
        static GqPerDeviceOnce once;                                                                                  \
        GQ_HIP_CHECK(once.max_dynamic_lds(reinterpret_cast<const void *>(qtip_matvec_kernel<RR, SS, WW>), (int)(QT_LDS - (size_t)QtipTab<SS>::WORDS * 4u - QT_STATIC_SLACK))); \
        hipLaunchKernelGGL((qtip_matvec_kernel<RR, SS, WW>), grid, block, smem, s, out, compressed, (const uint16_t *)x, \
                           (const uint16_t *)codebook, M, K, g_qdbg);                                                 \
    } while (0)
#define GQ_LAUNCH_QTIP(RR, SS)                      \
    do {                                            \
        if (waves == 16u) GQ_LAUNCH_QTIP_W(RR, SS, 16); \
        else GQ_LAUNCH_QTIP_W(RR, SS, 0);           \
    } while (0)
#define GQ_LAUNCH_QTIP_R(RR)               \
    do {                                   \
        if (sx) GQ_LAUNCH_QTIP(RR, 1);     \
        else GQ_LAUNCH_QTIP(RR, 0);        \
    } while (0)
    if (R == 2) GQ_LAUNCH_QTIP_R(2);
    else if (R == 3) GQ_LAUNCH_QTIP_R(3);
    else GQ_LAUNCH_QTIP_R(4);
#undef GQ_LAUNCH_QTIP_R
#undef GQ_LAUNCH_QTIP
#undef GQ_LAUNCH_QTIP_W
    GQ_HIP_CHECK(hipGetLastError());
    return GQ_OK;
}

namespace {
bool pow2(u32 n) { return n && !(n & (n - 1u)); }

// blocks per linear: the `cus` blocks of a launch are shared out in proportion to the bands (every block serves ONE linear:
// its prologue multiplies by that linear's SU); returns the largest number of rounds (items per block) of any linear
u32 qtip_share_blocks(int n, const u32 *bands, u32 ksplit, u32 cus, u32 *nblk) {
    u32 total = 0, items = 0;
    for (int i = 0; i < n; i++) total += bands[i], items += bands[i] * ksplit;
    if (items <= cus) {
        for (int i = 0; i < n; i++) nblk[i] = bands[i] * ksplit;
        return 1u;
    }
    u32 used = 0;
    for (int i = 0; i < n; i++) {
        nblk[i] = (u32)((unsigned long long)cus * bands[i] / total);
        if (nblk[i] == 0u) nblk[i] = 1u;
        used += nblk[i];
    }
    while (used < cus) {  // leftover blocks to the linear with the most items per block
        int best = 0;
        double worst = -1.0;
        for (int i = 0; i < n; i++) {
            const double load = (double)bands[i] * ksplit / nblk[i];
            if (load > worst) worst = load, best = i;
        }
        nblk[best]++, used++;
    }
    u32 rounds = 0;
    for (int i = 0; i < n; i++) {
        const u32 r = (bands[i] * ksplit + nblk[i] - 1u) / nblk[i];
        if (r > rounds) rounds = r;
    }
    return rounds;
}
}  // namespace

extern "C" void gq_debug_set_qtip_timing_buffer(void *p) { g_qdbg = (unsigned long long *)p; }

extern "C" int gq_qtip_plan_ksplit(int n, const uint32_t *M, uint32_t K, int max_ksplit) {
    if (n < 1 || n > 3 || !M || K < 32u) return 1;
    if (max_ksplit > 4) max_ksplit = 4;
    const u32 cus = (u32)gq_cu_count(), nK2 = K / 32u;
    u32 bands[3], nblk[3];
    for (int i = 0; i < n; i++) bands[i] = M[i] / 32u;
    int best = 1;
    double best_cost = 1e30;
    for (int ks = 1; ks <= max_ksplit; ks++) {
        if (ks > 1 && nK2 / 4u / (u32)ks < 16u) break;  // every K range keeps the 16 waves of a block busy (one 4-tile-block chunk each)
        const u32 rounds = qtip_share_blocks(n, bands, (u32)ks, cus, nblk);
        // time ~ rounds of 1 / ks band each, plus a small price per extra part (its flush and the consumer's extra read)
        const double cost = (double)rounds / ks * (1.0 + 0.02 * (ks - 1));
        if (cost < best_cost - 1e-9) best_cost = cost, best = ks;
    }
    return best;
}

namespace {
int qtip_linear_in_impl(const void *x, const void *x2, const void *norm_weight, float eps, int prologue, uint32_t K, int R, int n,
                        const GqQtipIn *lin, int n_prev, const GqQtipOut *prev, int ksplit, const GqQtipOut *fin, void *counters, void *stream,
                        uint32_t rows_P = 0) {
    if (ksplit < 1 || ksplit > 4) return gq_fail(GQ_EINVAL, "gq_qtip_linear_in: ksplit must be 1..4.");
    if (!lin || n < 1 || n > 3) return gq_fail(GQ_EINVAL, "gq_qtip_linear_in: 1..3 linears.");
    if (n_prev < 0 || n_prev > 2 || (n_prev && !prev)) return gq_fail(GQ_EINVAL, "gq_qtip_linear_in: 0..2 producing linears.");
    if (n_prev == 0 && !x && prologue != GQ_QPRO_PRETRANSFORMED) return gq_fail(GQ_EINVAL, "gq_qtip_linear_in: x is null.");
    if (n_prev && n_prev != (prologue == GQ_QPRO_SILU_MUL ? 2 : 1))
        return gq_fail(GQ_EINVAL, "gq_qtip_linear_in: one producing linear per input vector (two for SILU_MUL).");
    if (R < 2 || R > 4) return gq_fail(GQ_ENOTSUP, "R (bits per weight) must be 2, 3 or 4 (kernel_check.py:1-14).");
    if (prologue == GQ_QPRO_PRETRANSFORMED) {
        if (K == 0 || K % 32u || K > 32768u || n_prev) return gq_fail(GQ_EINVAL, "gq_qtip_linear_in: pre-transformed input: K a multiple of 32, no folding.");
    } else if (prologue == 4) {  // QPRO_ROWS (gq_qtip_linear_in_rows: checked there)
        if (!rows_P || n_prev || fin) return gq_fail(GQ_EINVAL, "gq_qtip_linear_in: prologue 4 is reached through gq_qtip_linear_in_rows.");
    } else if (!pow2(K) || K < 32u || K > 16384u)
        return gq_fail(GQ_ENOTSUP, "fused QTIP linear: K must be a power of two in 32..16384.");
    if ((prologue == GQ_QPRO_RMSNORM && !norm_weight) || (prologue == GQ_QPRO_SILU_MUL && !x2 && !n_prev) || prologue < 0 || prologue > 4)
        return gq_fail(GQ_EINVAL, "gq_qtip_linear_in: prologue operand missing.");
    const u32 nK2 = K / 32u;
    if ((u32)ksplit > nK2) return gq_fail(GQ_EINVAL, "gq_qtip_linear_in: more K ranges than tile blocks.");
    QtipInArgs a{};
    a.x = (const uint16_t *)x;
    a.x2 = (const uint16_t *)x2;
    a.normw = (const uint16_t *)norm_weight;
    a.eps = eps;
    a.kscale = (float)pow((double)K, -0.5);
    a.K = K;
    a.n = (u32)n;
    a.P = rows_P;
    u32 bands[3] = {0, 0, 0}, nblk[3] = {0, 0, 0};
    for (int i = 0; i < n; i++) {
        if (!lin[i].trellis || (!lin[i].SU && prologue != GQ_QPRO_PRETRANSFORMED && prologue != 4) || !lin[i].tlut || !lin[i].y32 || lin[i].M == 0 || lin[i].M % 32u)
            return gq_fail(GQ_EINVAL, "gq_qtip_linear_in: null pointer or M not a multiple of 32.");
        if (prologue == GQ_QPRO_PRETRANSFORMED && !x && (!lin[i].SU || ((uintptr_t)lin[i].SU & 15u)))
            return gq_fail(GQ_EINVAL, "gq_qtip_linear_in: pre-transformed input without x: every linear's SU field must point to ITS fp16 vector (16-byte aligned).");
        if (((uintptr_t)lin[i].trellis | (uintptr_t)lin[i].tlut) & 15u) return gq_fail(GQ_EINVAL, "buffers must be 16-byte aligned.");
        bands[i] = lin[i].M / 32u;
    }
    qtip_share_blocks(n, bands, (u32)ksplit, (u32)gq_cu_count(), nblk);
    u32 blocks = 0;
    for (int i = 0; i < n; i++) {
        a.lin[i] = QtipIn{lin[i].trellis, lin[i].SU, (const uint16_t *)lin[i].tlut, lin[i].y32, blocks, nblk[i], lin[i].M};
        blocks += nblk[i];
    }
    // K ranges in chunks (the engine's load unit: 4 tile blocks at R = 2, else 2)
    const u32 ch = R == 2 ? 4u : 2u, nch = (nK2 + ch - 1u) / ch, cpart = (nch + (u32)ksplit - 1u) / (u32)ksplit;
    if (cpart * ((u32)ksplit - 1u) >= nch) return gq_fail(GQ_EINVAL, "gq_qtip_linear_in: an empty K range (ksplit too large for K).");
    const u32 clast = nch - ((u32)ksplit - 1u) * cpart;
    u32 waves = qtip_waves(clast < cpart ? clast : cpart);
    if (waves < 4u) waves = 4u;  // the transform wants threads
    if (prologue != GQ_QPRO_PRETRANSFORMED && K > 16u * waves * 64u) waves = 16u;  // (the in-place fp16 conversion: <= 2 units per thread)
    a.ksplit = (u32)ksplit;
    a.dbg = g_qdbg;
    a.fin_ctr = nullptr;
    if (fin) {
        if (!counters || ((uintptr_t)counters & 3u)) return gq_fail(GQ_EINVAL, "gq_qtip_linear: counters (u32 [n], zeroed once) missing.");
        for (int i = 0; i < n; i++) {
            if (!fin[i].SV32 || !fin[i].out || fin[i].M != lin[i].M) return gq_fail(GQ_EINVAL, "gq_qtip_linear: finish descriptor (SV32, out, M) does not match the linear.");
            if (!pow2(fin[i].M) || fin[i].M < 32u || fin[i].M > 16384u)
                return gq_fail(GQ_ENOTSUP, "gq_qtip_linear: M must be a power of two in 32..16384 (other widths: gq_qtip_linear_in + gq_qtip_transform).");
            a.fin[i] = QtipOut{lin[i].y32, fin[i].SV32, (const uint16_t *)fin[i].resid, (uint16_t *)fin[i].out, fin[i].M,
                               (float)pow((double)fin[i].M, -0.5), (u32)ksplit};
        }
        a.fin_ctr = (u32 *)counters;
    }
    a.nprev = (u32)n_prev;
    for (int i = 0; i < n_prev; i++) {
        if (!prev[i].y32 || !prev[i].SV32 || prev[i].M != K) return gq_fail(GQ_EINVAL, "gq_qtip_linear_in: producing linear must have M == K.");
        if (prev[i].parts > 4u) return gq_fail(GQ_EINVAL, "gq_qtip_linear_in: parts must be 0..4.");
        a.prev[i] = QtipOut{prev[i].y32, prev[i].SV32, (const uint16_t *)prev[i].resid, (uint16_t *)prev[i].out, K, (float)pow((double)K, -0.5),
                            prev[i].parts ? prev[i].parts : 1u};
    }
    // LDS layout: the transform buffer v (K floats at 0) is dead when the matvec starts: the chunk slots re-use it, and so does
    // the fp16 copy xs when both fit (K >= 8192; the conversion is staged through registers)
    const size_t slots = (size_t)waves * 1024u, pbytes = (size_t)waves * 2u * 32u * 4u;
    size_t end;
    if (prologue == GQ_QPRO_PRETRANSFORMED) {
        a.xs_off = 0, a.part_off = K * 2u, a.stg_off = (u32)(K * 2u + pbytes), end = (size_t)K * 2u + pbytes + slots;
    } else if ((size_t)K * 2u + slots <= (size_t)K * 4u) {
        a.xs_off = 0, a.stg_off = K * 2u, a.part_off = K * 4u, end = (size_t)K * 4u + pbytes;
    } else {
        const size_t va = (size_t)K * 4u > slots ? (size_t)K * 4u : slots;
        a.stg_off = 0, a.xs_off = (u32)va, a.part_off = (u32)(va + (size_t)K * 2u), end = va + (size_t)K * 2u + pbytes;
    }
    // the rebuilt input vector(s) of a folded transform-out: one vector fits the (not yet written) xs region when that is
    // not inside v; otherwise behind everything
    const bool xp_in_xs = n_prev == 1 && a.xs_off != 0u;
    a.xp_off = xp_in_xs ? a.xs_off : (u32)end;
    const size_t smem = xp_in_xs ? end : end + (size_t)n_prev * K * 2u;
    if (!qtip_fits(smem)) return gq_fail(GQ_ENOTSUP, "gq_qtip_linear_in: K too large.");
    const int sx = qtip_sx_fits(smem) && gq_env_int("GQ_QTIP_SX", 1) ? 1 : 0;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(blocks), block(waves * 64u);
#define GQ_LAUNCH_QIN_W(RR, PP, SS, WW)                                                                               \
    do {                                                                                                              \
        static GqPerDeviceOnce once;                                                                                  \
        GQ_HIP_CHECK(once.max_dynamic_lds(reinterpret_cast<const void *>(qtip_linear_in_kernel<RR, PP, SS, WW>), (int)(QT_LDS - (size_t)QtipTab<SS>::WORDS * 4u - QT_STATIC_SLACK))); \
        hipLaunchKernelGGL((qtip_linear_in_kernel<RR, PP, SS, WW>), grid, block, smem, s, a);                         \
    } while (0)
#define GQ_LAUNCH_QIN(RR, PP, SS)                       \
    do {                                                \
        if (waves == 16u) GQ_LAUNCH_QIN_W(RR, PP, SS, 16); \
        else GQ_LAUNCH_QIN_W(RR, PP, SS, 0);            \
    } while (0)
#define GQ_LAUNCH_QIN_S(RR, PP)            \
    do {                                   \
        if (sx) GQ_LAUNCH_QIN(RR, PP, 1);  \
        else GQ_LAUNCH_QIN(RR, PP, 0);     \
    } while (0)
#define GQ_LAUNCH_QIN_R(RR)                                                   \
    do {                                                                      \
        if (prologue == GQ_QPRO_RMSNORM) GQ_LAUNCH_QIN_S(RR, QPRO_RMSNORM);     \
        else if (prologue == GQ_QPRO_SILU_MUL) GQ_LAUNCH_QIN_S(RR, QPRO_SILUMUL); \
        else if (prologue == GQ_QPRO_PRETRANSFORMED) GQ_LAUNCH_QIN_S(RR, QPRO_PRE); \
        else if (prologue == 4) GQ_LAUNCH_QIN_S(RR, QPRO_ROWS);                   \
        else GQ_LAUNCH_QIN_S(RR, QPRO_NONE);                                    \
    } while (0)
    if (R == 2) GQ_LAUNCH_QIN_R(2);
    else if (R == 3) GQ_LAUNCH_QIN_R(3);
    else GQ_LAUNCH_QIN_R(4);
#undef GQ_LAUNCH_QIN_R
#undef GQ_LAUNCH_QIN_S
#undef GQ_LAUNCH_QIN
#undef GQ_LAUNCH_QIN_W
    GQ_HIP_CHECK(hipGetLastError());
    return GQ_OK;
}
}  // namespace

extern "C" int gq_qtip_linear_in(const void *x, const void *x2, const void *norm_weight, float eps, int prologue, uint32_t K, int R,
                                 int n, const GqQtipIn *lin, int n_prev, const GqQtipOut *prev, int ksplit, void *stream) {
    return qtip_linear_in_impl(x, x2, norm_weight, eps, prologue, K, R, n, lin, n_prev, prev, ksplit, nullptr, nullptr, stream);
}

extern "C" int gq_qtip_linear(const void *x, const void *x2, const void *norm_weight, float eps, int prologue, uint32_t K, int R, int n,
                              const GqQtipIn *lin, const GqQtipOut *finish, int ksplit, void *counters, void *stream) {
    if (!finish) return gq_fail(GQ_EINVAL, "gq_qtip_linear: finish descriptors missing.");
    return qtip_linear_in_impl(x, x2, norm_weight, eps, prologue, K, R, n, lin, 0, nullptr, ksplit, finish, counters, stream);
}

extern "C" int gq_qtip_mlp_mid(const GqQtipMid *m, uint32_t parts, uint32_t n, uint32_t Kf, void *stream) {
    if (!m || !m->y32_gate || !m->y32_up || !m->SV32_gate || !m->SV32_up || !m->hadT_right16 || !m->SU_down || !m->had_left_down16 || !m->z32)
        return gq_fail(GQ_EINVAL, "gq_qtip_mlp_mid: null pointer argument.");
    if (Kf < 2u || n == 0 || n % Kf) return gq_fail(GQ_EINVAL, "gq_qtip_mlp_mid: n a multiple of Kf.");
    if (n / Kf != 64u || Kf > 176u || Kf % 4u)  // (both fp16 tables in LDS: 4 Kf^2 + 35 KiB <= 160 KiB)
        return gq_fail(GQ_ENOTSUP, "gq_qtip_mlp_mid: serves n = Kf * 64 with Kf <= 176, Kf % 4 == 0 (other widths: two gq_qtip_transform launches).");
    if (parts < 1u || parts > 4u) return gq_fail(GQ_EINVAL, "gq_qtip_mlp_mid: parts must be 1..4.");
    if (((uintptr_t)m->y32_gate | (uintptr_t)m->y32_up) & 15u) return gq_fail(GQ_EINVAL, "gq_qtip_mlp_mid: the sums must be 16-byte aligned.");
    if (((uintptr_t)m->hadT_right16 | (uintptr_t)m->had_left_down16) & 7u) return gq_fail(GQ_EINVAL, "gq_qtip_mlp_mid: the tables must be 8-byte aligned.");
    QtipMidArgs a{};
    a.y32g = m->y32_gate, a.y32u = m->y32_up, a.sv32g = m->SV32_gate, a.sv32u = m->SV32_up;
    a.hadT_out = (const uint16_t *)m->hadT_right16, a.had_in = (const uint16_t *)m->had_left_down16;
    a.su_down = m->SU_down, a.z32 = m->z32;
    a.gout = (uint16_t *)m->gate_out, a.uout = (uint16_t *)m->up_out;
    a.parts = parts, a.n = n, a.Kf = Kf;
    a.kper = (Kf + 15u) / 16u;  // (qtip_transform_kernel: KQ = 1024 / 64 = 16 k groups)
    a.nscale = (float)pow((double)n, -0.5);
    a.dbg = g_qdbg;
    const size_t smem = 2u * (size_t)Kf * Kf * 2u + (MID_R * 2u + 16u * 2u * MID_R + MID_R) * 4u;
    static GqPerDeviceOnce once;
    GQ_HIP_CHECK(once.max_dynamic_lds(reinterpret_cast<const void *>(qtip_mlp_mid_kernel), 160 * 1024));
    hipLaunchKernelGGL(qtip_mlp_mid_kernel, dim3(64u), dim3(MID_T), smem, (hipStream_t)stream, a);
    GQ_HIP_CHECK(hipGetLastError());
    return GQ_OK;
}

extern "C" int gq_qtip_linear_in_rows(const float *z32, uint32_t K, uint32_t P, int R, int n, const GqQtipIn *lin, int ksplit, void *stream) {
    if (!z32 || ((uintptr_t)z32 & 15u)) return gq_fail(GQ_EINVAL, "gq_qtip_linear_in_rows: z32 null or not 16-byte aligned.");
    if (P < 2u || !pow2(P) || K == 0 || K % P || K % 32u || K > 16384u) return gq_fail(GQ_EINVAL, "gq_qtip_linear_in_rows: K = Kf * P <= 16384, P a power of two, K % 32 == 0.");
    return qtip_linear_in_impl(z32, nullptr, nullptr, 0.f, 4, K, R, n, lin, 0, nullptr, ksplit, nullptr, nullptr, stream, P);
}

extern "C" int gq_qtip_linear_out(int n, const GqQtipOut *lin, void *stream) {
    if (!lin || n < 1 || n > 3) return gq_fail(GQ_EINVAL, "gq_qtip_linear_out: 1..3 linears.");
    QtipOutArgs a{};
    u32 maxM = 0;
    for (int i = 0; i < n; i++) {
        if (!lin[i].y32 || !lin[i].SV32 || !lin[i].out) return gq_fail(GQ_EINVAL, "gq_qtip_linear_out: null pointer argument.");
        if (!pow2(lin[i].M) || lin[i].M < 32u || lin[i].M > 32768u)
            return gq_fail(GQ_ENOTSUP, "fused QTIP linear: M must be a power of two in 32..32768.");
        if (lin[i].parts > 4u) return gq_fail(GQ_EINVAL, "gq_qtip_linear_out: parts must be 0..4.");
        a.lin[i] = QtipOut{lin[i].y32, lin[i].SV32, (const uint16_t *)lin[i].resid, (uint16_t *)lin[i].out, lin[i].M,
                           (float)pow((double)lin[i].M, -0.5), lin[i].parts ? lin[i].parts : 1u};
        if (lin[i].M > maxM) maxM = lin[i].M;
    }
    const size_t smem = (size_t)maxM * 4u;
    static GqPerDeviceOnce once;
    GQ_HIP_CHECK(once.max_dynamic_lds(reinterpret_cast<const void *>(qtip_linear_out_kernel), 160 * 1024));
    hipLaunchKernelGGL(qtip_linear_out_kernel, dim3((u32)n), dim3(1024), smem, (hipStream_t)stream, a);
    GQ_HIP_CHECK(hipGetLastError());
    return GQ_OK;
}

extern "C" int gq_qtip_linear_out_in(const GqQtipOut *prev, const void *norm_weight, float eps, int n_next, const float *const *SU_next,
                                     void *const *xt_next, void *stream) {
    if (!prev || !norm_weight || !SU_next || !xt_next || n_next < 1 || n_next > 3) return gq_fail(GQ_EINVAL, "gq_qtip_linear_out_in: 1..3 consumers, no null pointers.");
    const u32 K = prev->M;
    if (!pow2(K) || K < 256u || K > 8192u) return gq_fail(GQ_ENOTSUP, "gq_qtip_linear_out_in: width must be a power of two in 256..8192.");
    if (!prev->y32 || !prev->SV32 || prev->parts > 4u) return gq_fail(GQ_EINVAL, "gq_qtip_linear_out_in: producing linear (y32, SV32, parts <= 4).");
    if (((uintptr_t)prev->y32 | (uintptr_t)prev->SV32 | (uintptr_t)norm_weight) & 15u || ((uintptr_t)prev->resid | (uintptr_t)prev->out) & 7u)
        return gq_fail(GQ_EINVAL, "gq_qtip_linear_out_in: buffers must be 16-byte (fp16 vectors: 8-byte) aligned.");
    QtipOutInArgs a{};
    a.prev = QtipOut{prev->y32, prev->SV32, (const uint16_t *)prev->resid, (uint16_t *)prev->out, K, (float)pow((double)K, -0.5), prev->parts ? prev->parts : 1u};
    a.normw = (const uint16_t *)norm_weight;
    a.eps = eps;
    a.kscale = (float)pow((double)K, -0.5);
    for (int i = 0; i < n_next; i++) {
        if (!SU_next[i] || !xt_next[i] || (((uintptr_t)SU_next[i] | (uintptr_t)xt_next[i]) & 15u)) return gq_fail(GQ_EINVAL, "gq_qtip_linear_out_in: consumer vectors (16-byte aligned).");
        a.SU[i] = SU_next[i];
        a.xt[i] = (uint16_t *)xt_next[i];
    }
    hipLaunchKernelGGL(qtip_out_in_kernel, dim3((unsigned)n_next), dim3(1024), (size_t)K * 6u, (hipStream_t)stream, a);
    GQ_HIP_CHECK(hipGetLastError());
    return GQ_OK;
}

extern "C" int gq_qtip_linear_out_seg(int n, const GqQtipOut *lin, void *stream) {
    if (!lin || n < 1 || n > 3) return gq_fail(GQ_EINVAL, "gq_qtip_linear_out_seg: 1..3 linears.");
    QtipOutArgs a{};
    u32 maxM = 0;
    for (int i = 0; i < n; i++) {
        if (!lin[i].y32 || !lin[i].SV32 || !lin[i].out) return gq_fail(GQ_EINVAL, "gq_qtip_linear_out_seg: null pointer argument.");
        if (!pow2(lin[i].M) || lin[i].M < QSEG || lin[i].M > 16u * QSEG_T)
            return gq_fail(GQ_ENOTSUP, "gq_qtip_linear_out_seg: M must be a power of two in 128..8192 (else gq_qtip_linear_out).");
        if (lin[i].parts > 4u) return gq_fail(GQ_EINVAL, "gq_qtip_linear_out_seg: parts must be 0..4.");
        if (((uintptr_t)lin[i].y32) & 15u) return gq_fail(GQ_EINVAL, "gq_qtip_linear_out_seg: the sums must be 16-byte aligned.");
        a.lin[i] = QtipOut{lin[i].y32, lin[i].SV32, (const uint16_t *)lin[i].resid, (uint16_t *)lin[i].out, lin[i].M,
                           (float)pow((double)lin[i].M, -0.5), lin[i].parts ? lin[i].parts : 1u};
        if (lin[i].M > maxM) maxM = lin[i].M;
    }
    hipLaunchKernelGGL(qtip_linear_out_seg_kernel, dim3(maxM / QSEG, (u32)n), dim3(QSEG_T), 0, (hipStream_t)stream, a);
    GQ_HIP_CHECK(hipGetLastError());
    return GQ_OK;
}

extern "C" int gq_qtip_transform(int input_side, const void *x, const void *x2, const void *norm_weight, float eps, int prologue,
                                 int n_lin, const GqQtipXf *lin, uint32_t n, uint32_t Kf, int transpose, void *stream) {
    if (!lin || n_lin < 1 || n_lin > 3 || Kf < 2u || n == 0 || n % Kf) return gq_fail(GQ_EINVAL, "gq_qtip_transform: 1..3 linears; n a multiple of Kf.");
    const u32 P = n / Kf;
    if (!pow2(P) || P < 64u || n > 32768u) return gq_fail(GQ_ENOTSUP, "gq_qtip_transform: n / Kf must be a power of two >= 64, n <= 32768.");
    if (input_side && (!x || (prologue == GQ_QPRO_RMSNORM && !norm_weight) || (prologue == GQ_QPRO_SILU_MUL && !x2) || prologue < 0 || prologue > 2))
        return gq_fail(GQ_EINVAL, "gq_qtip_transform: source operand missing.");
    if (((uintptr_t)x | (uintptr_t)x2 | (uintptr_t)norm_weight) & 15u) return gq_fail(GQ_EINVAL, "gq_qtip_transform: buffers must be 16-byte aligned.");
    QtipXfArgs a{};
    a.x = (const uint16_t *)x;
    a.x2 = (const uint16_t *)x2;
    a.normw = (const uint16_t *)norm_weight;
    for (int i = 0; i < n_lin; i++) {
        if (((uintptr_t)lin[i].y32 | (uintptr_t)lin[i].vec) & 15u) return gq_fail(GQ_EINVAL, "gq_qtip_transform: buffers must be 16-byte aligned.");
        if (!lin[i].vec || !lin[i].out || !lin[i].hadK || (!input_side && !lin[i].y32)) return gq_fail(GQ_EINVAL, "gq_qtip_transform: null pointer argument.");
        a.lin[i] = QtipXfLin{lin[i].y32, lin[i].vec, lin[i].hadK, (const uint16_t *)lin[i].resid, (uint16_t *)lin[i].out};
    }
    a.eps = eps;
    a.nscale = (float)pow((double)n, -0.5);
    a.n = n;
    a.Kf = Kf;
    a.P = P;
    a.RB = (u32)gq_env_int("GQ_QTIP_XF_RB", P >= 256u ? 1 : (P >= 128u ? 2 : 4));  // rows of the result per block (<= 4)
    if (a.RB < 1u || a.RB > 4u) a.RB = 1u;
    a.in = input_side ? 1u : 0u;
    a.pro = input_side ? (u32)prologue : 0u;
    a.transpose = transpose ? 1u : 0u;
    const u32 KQ = P >= 1024u ? 1u : 1024u / P;
    const size_t smem = ((size_t)n + 4u * (size_t)Kf + (size_t)KQ * a.RB * P) * 4u;
    if (smem > 150u * 1024u) return gq_fail(GQ_ENOTSUP, "gq_qtip_transform: n too large.");
    static GqPerDeviceOnce once;
    GQ_HIP_CHECK(once.max_dynamic_lds(reinterpret_cast<const void *>(qtip_transform_kernel), 156 * 1024));
    hipLaunchKernelGGL(qtip_transform_kernel, dim3((Kf + a.RB - 1u) / a.RB, (u32)n_lin), dim3(1024), smem, (hipStream_t)stream, a);
    GQ_HIP_CHECK(hipGetLastError());
    return GQ_OK;
}

extern "C" int gq_hadamard(const float *x, float *y, uint32_t rows, uint32_t n, float scale, void *stream) {
    if (!x || !y) return gq_fail(GQ_EINVAL, "null pointer argument.");
    if (n == 0 || (n & (n - 1u))) return gq_fail(GQ_EINVAL, "hadamard: the last dimension must be a power of two.");
    if (rows == 0) return GQ_OK;
    const size_t smem = (size_t)n * 4u;
    if (smem > 160u * 1024u) return gq_fail(GQ_ENOTSUP, "hadamard: n too large (<= 32768).");
    static GqPerDeviceOnce once;
    GQ_HIP_CHECK(once.max_dynamic_lds(reinterpret_cast<const void *>(fwht_kernel), 160 * 1024));
    const u32 T = n / 2u >= 1024u ? 1024u : (n / 2u >= 64u ? n / 2u : 64u);
    hipLaunchKernelGGL(fwht_kernel, dim3(rows), dim3(T), smem, (hipStream_t)stream, x, y, n, scale);
    GQ_HIP_CHECK(hipGetLastError());
    return GQ_OK;
}
