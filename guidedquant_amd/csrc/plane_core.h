// plane_core.h -- index math and small helpers of the "plane-MFMA" Any-Precision GEMV (fast mode).
//
// Idea.  A b-bit LUT is a multilinear function of the code bits c_i in {0,1}:
//     lut[c] = sum over subsets S of the code bits   coef[S] * prod_{i in S} c_i        (Moebius transform of lut)
// hence   y_n = sum_k x_k * lut_n[code(n,k)] = coef_n[{}] * sum_k x_k + sum_{S != {}} coef_n[S] * T_n[S],
//         T_n[S] = sum_k x_k * AND_{i in S} bit_i(n,k)
// i.e. 2^b - 1 BINARY GEMVs whose matrices are the stored bit-planes (and ANDs of them) -- exactly the bytes the
// reference streams (any_precision/quantization/pack.py:101-109), never expanded to fp16 weights.
//
// On CDNA4 a binary plane becomes an MFMA operand with ONE v_and_b32 per 8 weights: (word & 0x11111111 << b) is eight
// FP4 (e2m1) values in {0, 0.5 / 1 / 2}, a power of two that the scale operand of v_mfma_scale_f32_16x16x128_f8f6f4
// cancels for free.  The activations are split EXACTLY into 4 bf8 pieces (x * 2^k = p0 + p1 + p2 + p3, 3 significand
// bits each) that occupy 4 of the 16 MFMA columns, so the products are exact and the sums are accumulated in fp32: the
// result is MORE accurate than the reference's fp16-accumulated kernel (anyprec.cu:372-542), not bit-identical to it
// (that is what the "exact" mode in ap_gemv.hip is for).
//
// Chunk geometry is the reference's (anyprec.cu:498): a 1024-weight chunk of a row is 32 plane words ("virtual lanes"
// t = 0..31; a tail chunk has tpw = (K % 1024) / 32 of them); bit 8 (3 - c) + (7 - j) of word t is the weight of
// activation 1024 * chunk + 8 * tpw * c + 8 * t + j   (byte c = 0..3, weight j = 0..7).
#pragma once
#include <stdint.h>
#include <string.h>
#include <math.h>

#if defined(__HIPCC__)
#define GQP_HD __host__ __device__ __forceinline__
#else
#define GQP_HD inline
#endif

namespace gqp {

typedef uint32_t u32;

struct Geom {
    u32 K, wpr, nfull, eff, nchunks;
    GQP_HD void init(u32 K_) {
        K = K_;
        wpr = K_ / 32u;
        nfull = K_ / 1024u;
        eff = (K_ % 1024u) / 32u;
        nchunks = nfull + (eff ? 1u : 0u);
    }
    GQP_HD u32 tpw(u32 chunk) const { return chunk < nfull ? 32u : eff; }
};


// ---- FP4 A operand (v_mfma_scale_f32_16x16x128_f8f6f4 with cbsz = 4, blgp = 1: A = e2m1 nibbles, B = bf8 bytes).
// Measured (tools/ubench/fp4_probe.hip): single-bit nibble patterns 0x1 / 0x2 / 0x4 are 0.5 / 1 / 2 (0x8 is -0);
// A element (lane group g = lane / 16, VGPR v = 0..3, nibble i) is k = 32 g + 8 v + i; the bf8 B operand keeps
// k = 64 (v / 4) + 16 kb + 4 (v % 4) + byte.  The mixed MFMA costs the same matrix-pipe time as bf8 x bf8 but its A
// operand is 4 registers for the same 32 weights per lane: half the v_and_b32 per weight.
//   nibble bit b = 0..2:  w & (0x11111111 << b);     b = 3 (the sign position): (w >> 1) & 0x44444444
GQP_HD u32 extract4(u32 w, int b) { return b == 3 ? ((w >> 1) & 0x44444444u) : (w & (0x11111111u << b)); }
// E8M0 scale byte cancelling the pattern value (0.5, 1, 2, 2)
GQP_HD int scale_byte4(int b) { return b == 0 ? 128 : (b == 1 ? 127 : 126); }
// Lane (row r, group g) holds the plane words wd = 0..7 of virtual lanes t = 8 g + wd.  MFMA (b, h) takes the words
// 4 h + v (v = 0..3) masked at nibble bit b: nibble i of word wd is plane bit 4 i + b = byte B = i / 2 (c = 3 - B),
// bit s = 4 (i & 1) + b of that byte, i.e. weight j = 7 - s of virtual lane t.
// B image: [chunk][b][h][piece][k = 0..127] bytes, k = 32 g + 8 v + i; lane (col = piece, kb) of the MFMA reads the 16
// bytes at 16 kb and the 16 bytes at 64 + 16 kb.
GQP_HD u32 bimg4_off(u32 chunk, u32 b, u32 h, u32 piece) { return chunk * 4096u + (((b * 2u + h) * 4u + piece) << 7); }
// Bank swizzle of the image: the 4 piece rows of a (b, h) block are 128 B apart, so the MFMA lanes of one ds_read_b128 lane
// group -- pieces 0..3 at the same 16-byte unit kb -- would hit pieces 0 / 2 and 1 / 3 in the same banks (64 banks x 4 B = 256 B:
// measured 40 % of the LDS cycles of the 2-bit w1w3 kernel as bank conflicts).  Pieces 2 and 3 therefore keep byte k at
// k ^ 32 (16-byte unit kb ^ 2): units {0, 8, 2, 10} of the 16 per bank row.  Writers, the clearing of extracted / tail
// elements and the readers apply the same XOR; the lane still receives the bytes of ITS k range.
GQP_HD u32 bimg_swz(u32 piece) { return (piece & 2u) << 4; }
// activation e -> chunk, b, h, k  (inverse of the above)
GQP_HD void locate_x4(const Geom &G, u32 e, u32 &chunk, u32 &b, u32 &h, u32 &k) {
    u32 r, tp;
    if (e < 1024u * G.nfull) {
        chunk = e / 1024u;
        r = e % 1024u;
        tp = 32u;
    } else {
        chunk = G.nfull;
        r = e - 1024u * G.nfull;
        tp = G.eff;
    }
    const u32 c = r / (8u * tp), t = (r % (8u * tp)) / 8u, j = r % 8u, s = 7u - j;
    b = s & 3u;
    h = (t >> 2) & 1u;
    const u32 g = t >> 3, v = t & 3u, i = 2u * (3u - c) + (s >> 2);
    k = 32u * g + 8u * v + i;
}

// ---- A tile in LDS.  One step (16 rows x one 1024-weight chunk) of one plane is 16 lines of 128 B = 128 units of
// 16 B, deposited by two direct-to-LDS loads (buffer_load_dwordx4 ... lds: LDS address = base + 16 * lane, so the
// only freedom is WHICH global 16 B each lane fetches).  Instruction h (0/1) carries rows 8h..8h+7, lane = 8*a + b:
//     row = 8*h + rr,  rr = a ^ h,  segment (16 B of the line) = b ^ rr
// -> every instruction reads 8 whole 128-byte lines from HBM, and the MFMA lanes (row r = 0..15, fixed segment) of a
// ds_read_b128 pass hit 16 distinct 16-byte bank groups (unit mod 16 = 8*((rr&1)^h) + (seg^rr)).
GQP_HD u32 atile_unit(u32 r, u32 seg) {
    const u32 h = r >> 3, rr = r & 7u;
    return 64u * h + 8u * (rr ^ h) + (seg ^ rr);
}
GQP_HD void atile_src(u32 h, u32 lane, u32 &r, u32 &seg) {
    const u32 a = lane >> 3, b = lane & 7u, rr = a ^ h;
    r = 8u * h + rr;
    seg = b ^ rr;
}

// ---- activation pieces.  bf8 (e5m2) is fp16 with the low significand byte cut off, so an fp16 value splits EXACTLY
// into 4 bf8 pieces by truncate-and-subtract in fp16 arithmetic: p = h & 0xFF00, h -= p (exact), 4 times (3 significand
// bits each, 12 >= 11).  The vector is first multiplied by 2^k (fp16, exact unless it underflows) so that
// max|x| * 2^k is in [2^14, 2^15): pieces of every element within 2^-17 of the maximum stay bf8 NORMALS (the matrix
// cores are inexact on fp8 denormals).  k is clamped to what fp16 can hold as a factor.
GQP_HD int piece_shift(float xmax) {  // xmax >= 0 (an upper bound of max|x| is fine)
    u32 u;
    memcpy(&u, &xmax, 4);
    const int eb = xmax > 0.f ? (int)((u >> 23) & 0xFFu) - 126 : 15;  // xmax < 2^eb
    int k = 15 - eb;
    return k > 15 ? 15 : (k < -14 ? -14 : k);
}
GQP_HD uint16_t pow2_f16(int k) { return (uint16_t)((k + 15) << 10); }  // -14 <= k <= 15

// e5m2 ("bf8") software conversions (host emulation and documentation of what the hardware cvt does)
GQP_HD float bf8_to_f32(uint8_t b) {
    const int sg = b >> 7, e = (b >> 2) & 31, m = b & 3;
    float r = e ? ldexpf(1.0f + (float)m * 0.25f, e - 15) : ldexpf((float)m * 0.25f, -14);
    return sg ? -r : r;
}
GQP_HD uint8_t f32_to_bf8_rne(float f) {  // |f| < 57344 assumed (caller scales), RNE, subnormals kept
    u32 u;
    memcpy(&u, &f, 4);
    const uint8_t sg = (uint8_t)((u >> 24) & 0x80);
    const int e = (int)((u >> 23) & 0xFF);
    u32 m = u & 0x7FFFFFu;
    if (e == 0) return sg;  // f32 zero / subnormal
    m |= 0x800000u;         // 24-bit significand
    int he = e - 127 + 15, shift = 21;  // keep 3 bits (1 + 2)
    if (he < 1) {
        shift = 21 + (1 - he);
        if (shift > 25) return sg;
        he = 0;
    }
    u32 q = m >> shift, rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1u))) q++;
    u32 bits = he >= 1 ? (((u32)(he - 1) << 2) + q) : q;
    if (bits > 0x7Bu) bits = 0x7Bu;
    return (uint8_t)(sg | bits);
}

// Moebius transform in place: f[code] (lut values) -> coefficient of prod_{i in code} c_i
template <int BITS>
GQP_HD void moebius(float *f) {
#pragma unroll
    for (int i = 0; i < BITS; i++)
#pragma unroll
        for (int c = 0; c < (1 << BITS); c++)
            if (c & (1 << i)) f[c] -= f[c ^ (1 << i)];
}

}  // namespace gqp
