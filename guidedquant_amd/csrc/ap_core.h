// ap_core.h -- per-lane arithmetic of the Any-Precision LUT-GEMV for gfx950.
//
// One *item* = one output row x one *quad* of 4 consecutive 32-bit words of
// that row's bit-planes (one 16-byte load per plane).  In the reference
// (inference/ap_gemv/anyprec.cu:372-542) every word belongs to one CUDA lane
// ("virtual lane" t) of a 32-lane warp; a quad therefore carries 4 virtual
// lanes x 32 weights.  The reference accumulates, per virtual lane and per
// 1024-weight chunk, two fp16 chains (even / odd weight) of 16 fused
// multiply-adds each, in the order byte c = 3,2,1,0, pair k = 0..3
// (anyprec.cu:495-504).  We keep exactly those chains, but pair them ACROSS
// virtual lanes into v_pk_fma_f16 lanes: (v0,v1) and (v2,v3), even and odd
// -> 4 packed accumulators per item, 8 independent chains.
//
// Decode strategy (CDNA4, no LDS): the four plane words of a quad are
// byte-transposed with v_perm_b32 so that each 32-bit word holds the SAME byte
// position of the 4 virtual lanes.  A "selector" word S then carries, in its 4
// bytes, the codes of 4 weights (one per virtual lane, same (c,j)).  The row's
// LUT lives in VGPRs split into low-byte and high-byte pools; two v_perm_b32
// look up the low / high bytes of the 4 fp16 centroids, two more interleave
// them into the two half2 operands of the packed FMAs.
//
// This header compiles for the device (hipcc) and for the host (g++), where
// v_perm_b32 / v_bfi_b32 / fp16 FMA are emulated bit-exactly; tests/ builds a
// host harness from it to check the lane program against the oracle without a
// GPU.
#pragma once
#include <stdint.h>

#include <string.h>
#include <math.h>
#if defined(__HIP_DEVICE_COMPILE__)
#define GQ_DEV 1  // device pass of hipcc: real instructions
#else
#define GQ_DEV 0  // host pass of hipcc, or plain g++: bit-exact emulation
#endif
#if defined(__HIPCC__)
#define GQ_HD __host__ __device__ __forceinline__
#else
#define GQ_HD inline
#endif

namespace gq {

typedef uint32_t u32;

// ----------------------------------------------------------------------------- bit ops
// perm(hi, lo, sel): byte i of the result = byte sel[i] of the 8-byte pool {hi:lo}
// (0..3 -> lo, 4..7 -> hi), 0x0C -> 0x00, >=0x0D -> 0xFF.   == v_perm_b32 hi, lo, sel
GQ_HD u32 perm(u32 hi, u32 lo, u32 sel) {
#if GQ_DEV
    return __builtin_amdgcn_perm(hi, lo, sel);
#else
    uint64_t pool = ((uint64_t)hi << 32) | lo;
    u32 r = 0;
    for (int i = 0; i < 4; i++) {
        u32 s = (sel >> (8 * i)) & 0xFF;
        u32 b;
        if (s <= 7)
            b = (u32)((pool >> (8 * s)) & 0xFF);
        else if (s == 0x0C)
            b = 0;
        else if (s >= 0x0D)
            b = 0xFF;
        else
            b = 0; /* 8..11 (sign replication) never used here */
        r |= b << (8 * i);
    }
    return r;
#endif
}

// bfi(mask, a, b) = (mask & a) | (~mask & b)      == v_bfi_b32
GQ_HD u32 bfi(u32 mask, u32 a, u32 b) { return (mask & a) | (~mask & b); }

// shift left by s if s >= 0, else logical right by -s (compile-time s after inlining)
GQ_HD u32 shl(u32 v, int s) { return s >= 0 ? (v << s) : (v >> (-s)); }

// ----------------------------------------------------------------------------- fp16 arithmetic
#if GQ_DEV
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
GQ_HD h2 u2h(u32 u) { return __builtin_bit_cast(h2, u); }
GQ_HD u32 h2u(h2 h) { return __builtin_bit_cast(u32, h); }
// packed fused multiply-add, one rounding per lane (v_pk_fma_f16) == CUDA __hfma2
GQ_HD u32 pk_fma(u32 a, u32 b, u32 c) { return h2u(__builtin_elementwise_fma(u2h(a), u2h(b), u2h(c))); }
GQ_HD u32 pk_add(u32 a, u32 b) { return h2u(u2h(a) + u2h(b)); }
GQ_HD uint16_t h_add(uint16_t a, uint16_t b) {
    _Float16 r = __builtin_bit_cast(_Float16, a) + __builtin_bit_cast(_Float16, b);
    return __builtin_bit_cast(uint16_t, r);
}
#else
GQ_HD double h2d_(uint16_t h) {
    u32 e = (h >> 10) & 0x1F, m = h & 0x3FF;
    double v = e == 0 ? ldexp((double)m, -24) : (e == 31 ? (m ? NAN : INFINITY) : ldexp((double)(m | 0x400), (int)e - 25));
    return (h >> 15) ? -v : v;
}
GQ_HD uint16_t d2h_(double d) {
    uint64_t u;
    memcpy(&u, &d, 8);
    uint16_t sign = (uint16_t)((u >> 48) & 0x8000);
    int e = (int)((u >> 52) & 0x7FF);
    uint64_t m = u & 0xFFFFFFFFFFFFFULL;
    if (e == 0x7FF) return (uint16_t)(sign | (m ? 0x7E00 : 0x7C00));
    if (e == 0) return sign;
    m |= 1ULL << 52;
    int he = e - 1008, shift = 42;
    if (he < 1) {
        shift = 1051 - e;
        if (shift > 54) return sign;
        he = 0;
    }
    uint64_t q = m >> shift, rem = m & ((1ULL << shift) - 1), half = 1ULL << (shift - 1);
    if (rem > half || (rem == half && (q & 1))) q++;
    u32 bits = he >= 1 ? (((u32)(he - 1) << 10) + (u32)q) : (u32)q;
    if (bits >= 0x7C00) bits = 0x7C00;
    return (uint16_t)(sign | bits);
}
GQ_HD uint16_t h_fma_(uint16_t a, uint16_t b, uint16_t c) { return d2h_(h2d_(a) * h2d_(b) + h2d_(c)); }
GQ_HD uint16_t h_add(uint16_t a, uint16_t b) { return d2h_(h2d_(a) + h2d_(b)); }
GQ_HD u32 pk_fma(u32 a, u32 b, u32 c) {
    return (u32)h_fma_(a & 0xFFFF, b & 0xFFFF, c & 0xFFFF) | ((u32)h_fma_(a >> 16, b >> 16, c >> 16) << 16);
}
GQ_HD u32 pk_add(u32 a, u32 b) {
    return (u32)h_add(a & 0xFFFF, b & 0xFFFF) | ((u32)h_add(a >> 16, b >> 16) << 16);
}
#endif

// ----------------------------------------------------------------------------- quad helpers
// 4x4 byte transpose of the quad's words: out[c] = bytes (3-c) of in[0..3], i.e.
// out[c] byte v = the byte of virtual lane v that holds weights 8c..8c+7 (MSB byte = c 0,
// pack.py:58-75 endianness flip).
GQ_HD void transpose_quad(const u32 in[4], u32 out[4]) {
    u32 lo01 = perm(in[1], in[0], 0x05010400u);  // a0.b0 a1.b0 a0.b1 a1.b1
    u32 hi01 = perm(in[1], in[0], 0x07030602u);  // a0.b2 a1.b2 a0.b3 a1.b3
    u32 lo23 = perm(in[3], in[2], 0x05010400u);
    u32 hi23 = perm(in[3], in[2], 0x07030602u);
    out[3] = perm(lo23, lo01, 0x05040100u);  // byte 0 of every lane  -> c = 3
    out[2] = perm(lo23, lo01, 0x07060302u);  // byte 1                -> c = 2
    out[1] = perm(hi23, hi01, 0x05040100u);  // byte 2                -> c = 1
    out[0] = perm(hi23, hi01, 0x07060302u);  // byte 3 (MSB)          -> c = 0
}

// The row's LUT (2^BITS fp16 centroids) as VGPR byte pools.
template <int BITS>
struct LutPools {
    // lo[i] / hi[i]: low / high bytes of centroids 4i..4i+3
    u32 lo[(1 << BITS) / 4];
    u32 hi[(1 << BITS) / 4];
    // raw: centroids as packed pairs, raw[i] = lut[2i] | lut[2i+1] << 16
    GQ_HD void build(const u32 *raw) {
#pragma unroll
        for (int i = 0; i < (1 << BITS) / 4; i++) {
            lo[i] = perm(raw[2 * i + 1], raw[2 * i], 0x06040200u);
            hi[i] = perm(raw[2 * i + 1], raw[2 * i], 0x07050301u);
        }
    }
};

// Look up 4 centroids (codes in the bytes of S, one per virtual lane) and return them as the two
// packed operands w01 = (w_v0, w_v1), w23 = (w_v2, w_v3).
template <int BITS>
GQ_HD void lookup4(const LutPools<BITS> &L, u32 S, u32 sel_hi_plane, u32 &w01, u32 &w23);

template <>
GQ_HD void lookup4<2>(const LutPools<2> &L, u32 S, u32, u32 &w01, u32 &w23) {
    u32 lo = perm(0u, L.lo[0], S);
    u32 hi = perm(0u, L.hi[0], S);
    w01 = perm(hi, lo, 0x05010400u);
    w23 = perm(hi, lo, 0x07030602u);
}
template <>
GQ_HD void lookup4<3>(const LutPools<3> &L, u32 S, u32, u32 &w01, u32 &w23) {
    u32 lo = perm(L.lo[1], L.lo[0], S);
    u32 hi = perm(L.hi[1], L.hi[0], S);
    w01 = perm(hi, lo, 0x05010400u);
    w23 = perm(hi, lo, 0x07030602u);
}
// 4 bit: S holds the low 3 code bits; msel holds, per byte, 0x0C (code < 8) or 0x0D (code >= 8)
template <>
GQ_HD void lookup4<4>(const LutPools<4> &L, u32 S, u32 msel, u32 &w01, u32 &w23) {
    u32 m = perm(0u, 0u, msel);  // 0x00 / 0xFF per byte
    u32 lo = bfi(m, perm(L.lo[3], L.lo[2], S), perm(L.lo[1], L.lo[0], S));
    u32 hi = bfi(m, perm(L.hi[3], L.hi[2], S), perm(L.hi[1], L.hi[0], S));
    w01 = perm(hi, lo, 0x05010400u);
    w23 = perm(hi, lo, 0x07030602u);
}

// x operands of one item: xr[c][jj][r]: r = 0: (v0,v1)@j=2jj  1: (v2,v3)@2jj  2: (v0,v1)@2jj+1  3: (v2,v3)@2jj+1
struct XRegs {
    u32 r[4][4][4];
};

// Accumulators of one item: [0] (v0,v1) even  [1] (v0,v1) odd  [2] (v2,v3) even  [3] (v2,v3) odd
struct Acc {
    u32 a[4];
};

GQ_HD void fma_pair(Acc &acc, const XRegs &x, int c, int j, u32 w01, u32 w23) {
    const int jj = j >> 1, par = j & 1;
    acc.a[par] = pk_fma(w01, x.r[c][jj][2 * par], acc.a[par]);
    acc.a[2 + par] = pk_fma(w23, x.r[c][jj][2 * par + 1], acc.a[2 + par]);
}


// ----------------------------------------------------------------------------- index math
// Geometry of one row: K weights, K/32 words per plane, Q = K/128 quads; chunk = 1024 weights = 8 quads;
// a tail chunk of (K%1024)/32 words when K is not a multiple of 1024 (anyprec.cu:433-436).
struct RowGeom {
    u32 K, wpr, Q, nfull, eff, nchunks;
    GQ_HD void init(u32 K_) {
        K = K_;
        wpr = K_ / 32u;
        Q = K_ / 128u;
        nfull = K_ / 1024u;
        eff = (K_ % 1024u) / 32u;
        nchunks = nfull + (eff ? 1u : 0u);
    }
    // quad q -> chunk index, first virtual lane t0, lanes in that chunk (tpw)
    GQ_HD void quad(u32 q, u32 &chunk, u32 &t0, u32 &tpw) const {
        u32 w = 4u * q;
        if (w < 32u * nfull) {
            chunk = w / 32u;
            t0 = w % 32u;
            tpw = 32u;
        } else {
            chunk = nfull;
            t0 = w - 32u * nfull;
            tpw = eff;
        }
    }
    // activation index multiplied with weight (v, c, j) of quad q  (anyprec.cu:498)
    GQ_HD u32 xindex(u32 q, u32 v, u32 c, u32 j) const {
        u32 chunk, t0, tpw;
        quad(q, chunk, t0, tpw);
        return 1024u * chunk + 8u * tpw * c + 8u * (t0 + v) + j;
    }
};

// Position (in halves) of activation (q, v, c, j) inside the LDS staging buffer: 16-byte slot
// (c*4 + j/2)*Q + q, inside the slot [(v0,v1)@even j, (v2,v3)@even j, (v0,v1)@odd j, (v2,v3)@odd j].
GQ_HD u32 xlds_pos(u32 Q, u32 q, u32 v, u32 c, u32 j) { return (((c * 4u + (j >> 1)) * Q + q) << 3) + ((j & 1u) << 2) + v; }

// Inverse used by the staging loop: group g = 8 consecutive activations 8g..8g+7 (one 16-byte global
// load) -> (q, v, c); j runs 0..7 inside the group.
GQ_HD void xgroup(const RowGeom &G, u32 g, u32 &q, u32 &v, u32 &c) {
    u32 t;
    if (g < 128u * G.nfull) {
        u32 r = g % 128u;
        c = r / 32u;
        t = r % 32u;
        q = 8u * (g / 128u) + t / 4u;
    } else {
        u32 gg = g - 128u * G.nfull;
        c = gg / G.eff;
        t = gg % G.eff;
        q = 8u * G.nfull + t / 4u;
    }
    v = t % 4u;
}

// ----------------------------------------------------------------------------- item programs
// P[p][v]: plane p (0 = MSB), word of virtual lane v.  Returns s01 = (s_v0, s_v1), s23 = (s_v2, s_v3),
// s_v = chain_even + chain_odd in fp16 (anyprec.cu:505 `sum.x + sum.y`).
template <int BITS>
struct Item;

template <>
struct Item<2> {
    GQ_HD static void run(const u32 P[2][4], const LutPools<2> &L, const XRegs &x, u32 &s01, u32 &s23) {
        u32 H[4], Lo[4];
        transpose_quad(P[0], H);
        transpose_quad(P[1], Lo);
        Acc acc = {{0u, 0u, 0u, 0u}};
#pragma unroll
        for (int c = 3; c >= 0; c--) {
            // 2-bit fields (H,L): Cm holds weights at odd bit positions s (field [s:s-1]),
            // Dm those at even positions (field [s+1:s])
            u32 Cm = bfi(0xAAAAAAAAu, H[c], Lo[c] >> 1);
            u32 Dm = bfi(0xAAAAAAAAu, H[c] << 1, Lo[c]);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                u32 w01, w23;
                u32 Se = (Cm >> (6 - 2 * k)) & 0x03030303u;  // weight j = 2k   (bit 7-2k of its byte)
                lookup4<2>(L, Se, 0u, w01, w23);
                fma_pair(acc, x, c, 2 * k, w01, w23);
                u32 So = (Dm >> (6 - 2 * k)) & 0x03030303u;  // weight j = 2k+1 (bit 6-2k)
                lookup4<2>(L, So, 0u, w01, w23);
                fma_pair(acc, x, c, 2 * k + 1, w01, w23);
            }
        }
        s01 = pk_add(acc.a[0], acc.a[1]);
        s23 = pk_add(acc.a[2], acc.a[3]);
    }
};

template <>
struct Item<3> {
    GQ_HD static void run(const u32 P[3][4], const LutPools<3> &L, const XRegs &x, u32 &s01, u32 &s23) {
        u32 T0[4], T1[4], T2[4];
        transpose_quad(P[0], T0);
        transpose_quad(P[1], T1);
        transpose_quad(P[2], T2);
        Acc acc = {{0u, 0u, 0u, 0u}};
#pragma unroll
        for (int c = 3; c >= 0; c--) {
            // nib[r]: nibble n = code (bits 2:0, bit 3 junk) of the weight at bit position 4n + r
            u32 nib[4];
#pragma unroll
            for (int r = 0; r < 4; r++)
                nib[r] = bfi(0x44444444u, shl(T0[c], 2 - r), bfi(0x22222222u, shl(T1[c], 1 - r), T2[c] >> r));
#pragma unroll
            for (int k = 0; k < 4; k++) {
#pragma unroll
                for (int par = 0; par < 2; par++) {
                    const int j = 2 * k + par, s = 7 - j;  // bit position inside the byte
                    u32 S = ((s & 4) ? (nib[s & 3] >> 4) : nib[s & 3]) & 0x07070707u;
                    u32 w01, w23;
                    lookup4<3>(L, S, 0u, w01, w23);
                    fma_pair(acc, x, c, j, w01, w23);
                }
            }
        }
        s01 = pk_add(acc.a[0], acc.a[1]);
        s23 = pk_add(acc.a[2], acc.a[3]);
    }
};

template <>
struct Item<4> {
    GQ_HD static void run(const u32 P[4][4], const LutPools<4> &L, const XRegs &x, u32 &s01, u32 &s23) {
        u32 T0[4], T1[4], T2[4], T3[4];
        transpose_quad(P[0], T0);
        transpose_quad(P[1], T1);
        transpose_quad(P[2], T2);
        transpose_quad(P[3], T3);
        Acc acc = {{0u, 0u, 0u, 0u}};
#pragma unroll
        for (int c = 3; c >= 0; c--) {
            u32 nib[4];  // low 3 code bits per nibble (from planes 1..3), bit 3 junk
#pragma unroll
            for (int r = 0; r < 4; r++)
                nib[r] = bfi(0x44444444u, shl(T1[c], 2 - r), bfi(0x22222222u, shl(T2[c], 1 - r), T3[c] >> r));
#pragma unroll
            for (int k = 0; k < 4; k++) {
#pragma unroll
                for (int par = 0; par < 2; par++) {
                    const int j = 2 * k + par, s = 7 - j;
                    u32 S = ((s & 4) ? (nib[s & 3] >> 4) : nib[s & 3]) & 0x07070707u;
                    u32 msel = ((T0[c] >> s) & 0x01010101u) | 0x0C0C0C0Cu;  // MSB plane picks the pool
                    u32 w01, w23;
                    lookup4<4>(L, S, msel, w01, w23);
                    fma_pair(acc, x, c, j, w01, w23);
                }
            }
        }
        s01 = pk_add(acc.a[0], acc.a[1]);
        s23 = pk_add(acc.a[2], acc.a[3]);
    }
};

// Per-weight code of virtual lane v at (c, j) straight from the (un-transposed) plane words:
// generic path for any bit-width (LUT then comes from LDS / memory).
template <int BITS>
GQ_HD u32 code_at(const u32 P[BITS][4], int v, int c, int j) {
    const int bp = 31 - (8 * c + j);
    u32 code = 0;
#pragma unroll
    for (int p = 0; p < BITS; p++) code = (code << 1) | ((P[p][v] >> bp) & 1u);
    return code;
}

}  // namespace gq
