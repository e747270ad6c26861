// ap_gemm.hip -- prefill GEMM of an Any-Precision linear with the dequantisation fused into the matrix-core loop.
//
//   out[s][n] = sum_k x[s][k] * lut[n][code(n, k)]          x fp16 [S][K], out fp16 [S][N], S > 1 rows (a prompt)
//
// Replaces the seq_len > 1 branch of APLinear.forward (inference/APLinear.py:35-50: `anyprec_dequant` -> torch.matmul,
// i.e. dequant_kbit_store, anyprec.cu:294-359, + a cuBLAS GEMM): there a dense fp16 copy of W (235 MB for the 8B gate/up
// matrix) is written to HBM and read back for every prefill call; here the bit-planes are the only weight bytes that move.
//
// Mapping (v_mfma_f32_32x32x16_f16: D[i][j] += sum_k A[i][k] B[k][j]; lane = 32 g + r holds A[i = r][8 g .. 8 g + 7] and
// B[8 g .. 8 g + 7][j = r], 8 consecutive k each):
//   i = weight row n, j = token s.  The 8 consecutive k of a lane's A fragment are ONE BYTE of each bit-plane word:
//   byte c of word t of a 1024-weight chunk holds the weights 256 c + 8 t + (0..7), MSB first (pack.py:304-321;
//   anyprec.cu:498).  Lane (r, g) keeps the plane words t = (tpw / 2) g + q of its row in registers; K-step q of byte lane c
//   therefore multiplies k = chunk base + 8 tpw c + 8 ((tpw / 2) g + q) + (0..7): the A fragment is decoded in registers
//   (bits -> codes by a multiply-spread, codes -> fp16 by v_perm_b32 lookups in the row's LUT held as VGPR byte pools:
//   the decode of ap_core.h), never staged as fp16 in LDS.  The B fragment of the same k is one ds_read_b128 of the token's
//   row in an LDS tile of x (two 64-wide k segments per stage, rows padded to 272 B: conflict-free 16-byte slots).
// Block = 256 threads = 4 waves, tile 128 weight rows x 128 tokens (wave: 32 rows x 4 token tiles = 64 accumulator VGPRs);
// x tiles are double-buffered through registers (next stage's global loads in flight during the MFMAs); weights are read
// once per (row tile, token tile).  fp32 accumulation over all of K, one rounding to fp16 (as a cuBLAS fp16 GEMM).
// Requires K % 64 == 0 (tail chunks of 64 .. 960 weights are served), 2 <= bits <= 4; other widths keep the dequant path.
// Two kernels: `ap_gemm_kernel` (round 2: the mapping above, x staged through registers, any K % 64 == 0) and, for K % 256 == 0,
// `ap_gemm_pipe_kernel` further down (round 3: x through a ring of direct-to-LDS loads, plane words requested a group ahead,
// hand software pipeline, RF x CF fragments per wave, optional K split) -- `launch_gemm` picks kernel and tile per problem.
#include <hip/hip_runtime.h>

#include "ap_core.h"
#include "gq_internal.h"

namespace {
using gq::u32;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr u32 BS = 128, BN = 128, ROWB = 272;  // token tile, weight-row tile, bytes per token row of an x stage (256 + 16 pad)
constexpr u32 STAGE_BYTES = BS * ROWB;

// bit k of x (k = 0..3) -> LSB of byte k:  x * (1 + 2^7 + 2^14 + 2^21) puts bit i at i + 7 m, distinct positions, no carries
__device__ __forceinline__ u32 spread4(u32 x) { return __umul24(x, 0x00204081u) & 0x01010101u; }

// 4 weights (one nibble of every plane byte; nibble bit 3 = first weight) -> two packed fp16 pairs (w0 | w1 << 16, w2 | w3 << 16)
template <int BITS>
__device__ __forceinline__ void decode4(const u32 *nib, const gq::LutPools<BITS> &L, u32 &v01, u32 &v23) {
    u32 lo, hi;
    if constexpr (BITS == 2) {
        const u32 S = (spread4(nib[0]) << 1) | spread4(nib[1]);
        lo = gq::perm(0u, L.lo[0], S);
        hi = gq::perm(0u, L.hi[0], S);
    } else if constexpr (BITS == 3) {
        const u32 S = (spread4(nib[0]) << 2) | (spread4(nib[1]) << 1) | spread4(nib[2]);
        lo = gq::perm(L.lo[1], L.lo[0], S);
        hi = gq::perm(L.hi[1], L.hi[0], S);
    } else {
        const u32 S = (spread4(nib[1]) << 2) | (spread4(nib[2]) << 1) | spread4(nib[3]);
        const u32 m = spread4(nib[0]) * 0xFFu;  // 0x00 / 0xFF per byte: code >= 8
        lo = gq::bfi(m, gq::perm(L.lo[3], L.lo[2], S), gq::perm(L.lo[1], L.lo[0], S));
        hi = gq::bfi(m, gq::perm(L.hi[3], L.hi[2], S), gq::perm(L.hi[1], L.hi[0], S));
    }
    // byte k of lo / hi belongs to nibble bit k = weight 3 - k
    v01 = gq::perm(hi, lo, 0x06020703u);
    v23 = gq::perm(hi, lo, 0x04000501u);
}

template <int BITS>
__device__ __forceinline__ f16x8 decode_byte(const u32 *w, u32 shift, const gq::LutPools<BITS> &L) {
    u32 hn[BITS], ln[BITS];
#pragma unroll
    for (int p = 0; p < BITS; p++) {
        const u32 b = w[p] >> shift;
        hn[p] = (b >> 4) & 15u;
        ln[p] = b & 15u;
    }
    u32 v[4];
    decode4<BITS>(hn, L, v[0], v[1]);
    decode4<BITS>(ln, L, v[2], v[3]);
    typedef u32 u32x4 __attribute__((ext_vector_type(4)));
    u32x4 t = {v[0], v[1], v[2], v[3]};
    return __builtin_bit_cast(f16x8, t);
}

template <int BITS, int RF>
__global__ void __launch_bounds__(256, 2) ap_gemm_kernel(const uint16_t *__restrict__ x, uint16_t *__restrict__ out,
                                                          const u32 *__restrict__ qw, const uint16_t *__restrict__ lut, u32 S, u32 N,
                                                          u32 K, u32 dbg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // 2 stages of [BS][ROWB]
    const u32 tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u, r = lane & 31u, g = lane >> 5;
    // RF row fragments per wave (rows n0 + 32 f + r): every B fragment read from LDS feeds RF MFMAs
    const u32 n0 = blockIdx.x * (BN * RF) + wave * (32u * RF), s0 = blockIdx.y * BS;
    const u32 wpr = K / 32u, nfull = K / 1024u, eff = (K % 1024u) / 32u, nchunks = nfull + (eff ? 1u : 0u);
    u32 n[RF];  // rows past N are computed on a clamped row and never stored
    const size_t plane_stride = (size_t)N * wpr;

    gq::LutPools<BITS> L[RF];
#pragma unroll
    for (int f = 0; f < RF; f++) {
        n[f] = min(n0 + 32u * (u32)f + r, N - 1u);
        u32 raw[(1 << BITS) / 2];
        const u32 *lp = reinterpret_cast<const u32 *>(lut + (size_t)n[f] * (1u << BITS));
#pragma unroll
        for (int i = 0; i < (1 << BITS) / 2; i++) raw[i] = lp[i];
        L[f].build(raw);
    }

    f32x16 acc[RF][4];
#pragma unroll
    for (int f = 0; f < RF; f++)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[f][j][e] = 0.f;

    // x stage copy: 128 tokens x 2 segments x 8 pieces of 16 B = 2048 pieces, 8 per thread; piece p = tid + 256 i
    uint4 pre[8];
    auto load_stage = [&](u32 kseg0, u32 kseg1, u32 nq) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const u32 p = tid + 256u * (u32)i, tok = p >> 4, within = p & 15u, seg = within >> 3, part = within & 7u;
            const u32 s = s0 + tok;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (s < S && part < nq) v = *reinterpret_cast<const uint4 *>(x + (size_t)s * K + (seg ? kseg1 : kseg0) + 8u * part);
            pre[i] = v;
        }
    };
    auto store_stage = [&](u32 buf) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const u32 p = tid + 256u * (u32)i, tok = p >> 4, within = p & 15u;
            *reinterpret_cast<uint4 *>(smem + buf * STAGE_BYTES + tok * ROWB + within * 16u) = pre[i];
        }
    };

    // stage sequence: chunk -> group of up to 8 K-steps (q0) -> byte lane c; the plane words of a (chunk, q0) group serve 4 stages
    struct Cursor {
        u32 chunk, q0, c;
    };
    auto stage_geom = [&](const Cursor &cu, u32 &kseg0, u32 &kseg1, u32 &nq) {
        const u32 tpw = cu.chunk < nfull ? 32u : eff, hw = tpw >> 1;
        nq = min(8u, hw - cu.q0);
        const u32 base = 1024u * cu.chunk + 8u * tpw * cu.c;
        kseg0 = base + 8u * cu.q0;
        kseg1 = base + 8u * (hw + cu.q0);
    };
    auto advance = [&](Cursor &cu) {
        if (++cu.c < 4u) return;
        cu.c = 0;
        const u32 hw = (cu.chunk < nfull ? 32u : eff) >> 1;
        cu.q0 += 8u;
        if (cu.q0 < hw) return;
        cu.q0 = 0;
        cu.chunk++;
    };

    Cursor cur{0u, 0u, 0u};
    u32 k0, k1, nq;
    stage_geom(cur, k0, k1, nq);
    load_stage(k0, k1, nq);
    store_stage(0);
    __syncthreads();
    u32 buf = 0;
    u32 w[RF][BITS][8];
    while (cur.chunk < nchunks) {
        const u32 tpw = cur.chunk < nfull ? 32u : eff, hw = tpw >> 1;
        stage_geom(cur, k0, k1, nq);
        if (cur.c == 0u) {  // the plane words t = hw g + q0 .. + nq - 1 of this lane's row (two 16-byte loads per plane)
#pragma unroll
            for (int f = 0; f < RF; f++) {
                const u32 *base = qw + (size_t)n[f] * wpr + 32u * cur.chunk + hw * g + cur.q0;
#pragma unroll
                for (int p = 0; p < BITS; p++) {
                    const u32 *pp = base + (size_t)p * plane_stride;
                    if (nq == 8u && (hw & 3u) == 0u && (wpr & 3u) == 0u) {
                        const uint4 a = *reinterpret_cast<const uint4 *>(pp), b = *reinterpret_cast<const uint4 *>(pp + 4);
                        w[f][p][0] = a.x, w[f][p][1] = a.y, w[f][p][2] = a.z, w[f][p][3] = a.w;
                        w[f][p][4] = b.x, w[f][p][5] = b.y, w[f][p][6] = b.z, w[f][p][7] = b.w;
                    } else {
#pragma unroll
                        for (int q = 0; q < 8; q++) w[f][p][q] = (u32)q < nq ? pp[q] : 0u;
                    }
                }
            }
        }
        Cursor nx = cur;
        advance(nx);
        const bool more = nx.chunk < nchunks;
        if (more && !(dbg & 2u)) {  // (GQ_GEMM_DBG=2: no x traffic after the first stage)
            u32 a0, a1, an;
            stage_geom(nx, a0, a1, an);
            load_stage(a0, a1, an);  // in flight during the MFMAs below
        }
        const unsigned char *xs = smem + buf * STAGE_BYTES + r * ROWB + g * 128u;
        const u32 shift = 24u - 8u * cur.c;
#pragma unroll
        for (int q = 0; q < 8; q++) {
            if ((u32)q < nq) {
                f16x8 a[RF];
#pragma unroll
                for (int f = 0; f < RF; f++) {
                    u32 wq[BITS];
#pragma unroll
                    for (int p = 0; p < BITS; p++) wq[p] = w[f][p][q];
                    if (dbg & 1u) {  // ablation (GQ_GEMM_DBG=1): no decode
                        typedef u32 u32x4 __attribute__((ext_vector_type(4)));
                        u32x4 t = {wq[0], wq[1], wq[0] ^ shift, wq[1] + (u32)q};
                        a[f] = __builtin_bit_cast(f16x8, t);
                    } else {
                        a[f] = decode_byte<BITS>(wq, shift, L[f]);
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const f16x8 b = *reinterpret_cast<const f16x8 *>(xs + (u32)j * 32u * ROWB + (u32)q * 16u);
#pragma unroll
                    for (int f = 0; f < RF; f++) acc[f][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[f], b, acc[f][j], 0, 0, 0);
                }
            }
        }
        if (more) store_stage(buf ^ 1u);
        __syncthreads();
        buf ^= 1u;
        cur = nx;
    }

    // D layout: col j = lane & 31 (token), row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) (weight row)
#pragma unroll
    for (int f = 0; f < RF; f++)
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const u32 s = s0 + 32u * (u32)j + r;
        if (s >= S) continue;
#pragma unroll
        for (int rg = 0; rg < 4; rg++) {
            const u32 nn = n0 + 32u * (u32)f + 8u * (u32)rg + 4u * g;
            if (nn >= N) continue;
            uint16_t h[4];
#pragma unroll
            for (int e = 0; e < 4; e++) h[e] = __builtin_bit_cast(uint16_t, (_Float16)acc[f][j][4 * rg + e]);
            uint16_t *dst = out + (size_t)s * N + nn;
            if (nn + 3u < N && (N & 3u) == 0u) {
                *reinterpret_cast<uint2 *>(dst) = make_uint2((u32)h[0] | ((u32)h[1] << 16), (u32)h[2] | ((u32)h[3] << 16));
            } else {
#pragma unroll
                for (int e = 0; e < 4; e++)
                    if (nn + (u32)e < N) dst[e] = h[e];
            }
        }
    }
}

template <int BITS>
int launch_gemm_first(const void *x, void *out, const uint32_t *qw, const void *lut, u32 S, u32 N, u32 K, hipStream_t s) {
    constexpr int RF = 1;
    static GqPerDeviceOnce once;
    auto kern = ap_gemm_kernel<BITS, RF>;
    const size_t smem = 2u * STAGE_BYTES;
    GQ_HIP_CHECK(once.max_dynamic_lds(reinterpret_cast<const void *>(kern), (int)smem));
    dim3 grid((N + BN * RF - 1u) / (BN * RF), (S + BS - 1u) / BS), block(256);
    hipLaunchKernelGGL(kern, grid, block, smem, s, (const uint16_t *)x, (uint16_t *)out, qw, (const uint16_t *)lut, S, N, K, (u32)gq_env_int("GQ_GEMM_DBG", 0));
    GQ_HIP_CHECK(hipGetLastError());
    return GQ_OK;
}

// ---- the pipelined form: x tiles by direct-to-LDS loads into a 4-deep ring, plane words requested one group ahead --------------
// Same fragment mapping as above, K-stage of 64 (two 32-wide segments: 4 K-steps q per byte lane c), RF row fragments per wave.
//   ring slot = 128 tokens x 128 B, unpadded; 16-byte slot sl = 4 g + q of a token row sits at physical slot sl ^ (token & 7):
//   the 8 lanes r = 8 m .. 8 m + 7 of a ds_read_b128 (same g, q) hit 8 different 16-byte bank groups.  The permutation is
//   applied on the global side of the direct-to-LDS load (lane l of a wave instruction fills LDS bytes 16 l .. 16 l + 15 of a
//   1 KiB span = token row l >> 3, physical slot l & 7, so it FETCHES logical slot (l & 7) ^ (l >> 3)).
//   Vector memory returns in order and is issued here from inline asm with hand-counted s_waitcnt (the compiler's own
//   waitcnt pass would drain the queue before every LDS read): per iteration i the wave issues its 4 loads of stage i + 3
//   and, at the first stage of a plane-word group, the RF * BITS 16-byte plane loads of the NEXT group; "at most 8 behind"
//   at the end of iteration i therefore means stage i + 1 (and every plane word older than that) has landed.
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void g_dma16(u32x4 rsrc, u32 lds_base, u32 voff, u32 soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_base), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
__device__ __forceinline__ u32x4 g_load16(u32x4 rsrc, u32 voff, u32 soff) {
    u32x4 r;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(r) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
    return r;
}
template <int N>
__device__ __forceinline__ void g_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ u32x4 g_rsrc(const void *p, u32 bytes) {
    const uint64_t a = (uint64_t)(uintptr_t)p;
    return (u32x4){(u32)a, (u32)(a >> 32) & 0xFFFFu, bytes, 0x00020000u};  // raw buffer: reads past `bytes` return 0
}

constexpr u32 P_RING = 4u;  // stages in the ring (16 KiB each for 128 tokens, 32 KiB for 256)

// RF row fragments x CF token fragments per wave, NW waves per block: block tile 32 RF NW rows x 32 CF tokens, the x ring
// shared by all waves.  Per MFMA a wave issues (decode of one A fragment) / CF VALU instructions and 1 / RF B reads; the SIMD
// hides about 5 other instructions per MFMA (profiles/r03_prefill_gemm_counters.txt), so the shape of choice is 1 x 8 with 8
// waves (two per SIMD, 128 accumulator registers each): 2.5 decode instructions + 1 ds_read_b128 per MFMA at 2 bits.
template <int BITS, int RF, int CF, int NW>
__global__ void __launch_bounds__(64 * NW, NW == 8 ? 1 : 2) ap_gemm_pipe_kernel(const uint16_t *__restrict__ x, uint16_t *__restrict__ out,
                                                               const u32 *__restrict__ qw, const uint16_t *__restrict__ lut, u32 S,
                                                               u32 N, u32 K, u32 dbg, u32 nbx, u32 ntiles, u32 gper, float *__restrict__ part) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];  // P_RING slots
    const u32 tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63u, r = lane & 31u, g = lane >> 5;
    constexpr u32 BT = 32u * CF, P_SLOT = BT * 128u, DSPAN = BT / 8u / (u32)NW;  // DSPAN: 1 KiB spans of a stage per wave
    static_assert(DSPAN >= 2u && DSPAN % 2u == 0u, "span parity is part of the slot permutation");
    // XCD-aware tile order.  Workgroup b lands on XCD b % 8 (observed placement; only speed depends on it): XCD i takes the
    // i-th eighth of the tiles in token-block-major order, so that the blocks resident on one XCD at any time share one or
    // two token blocks of x (1 MiB each at K = 4096) in that XCD's 4 MiB L2 while the planes stream through.  In launch
    // order (row block fastest over ALL XCDs) every L2 sees every live token block and x is re-read from memory: measured
    // 4.3 TB/s of x traffic at S = 2048 on the 8B gate/up matrix, the bound of that version.
    const u32 per_xcd = (ntiles + 7u) >> 3, tile = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if (tile >= ntiles) return;
    // (split K, short grids: tile = (K range ks, token block, row block); range ks covers the groups [ks gper, (ks + 1) gper) and
    // leaves its fp32 sums in part[ks][S][N] for gq's reduce kernel -- `part` null: the whole K here, fp16 out)
    const u32 ntxy = nbx * ((S + 32u * CF - 1u) / (32u * CF)), ks = tile / ntxy, txy = tile - ks * ntxy;
    const u32 bx = txy % nbx, by = txy / nbx;
    const u32 n0 = bx * (32u * RF * NW) + wave * (32u * RF), s0 = by * BT;
    const u32 wpr = K / 32u, nfull = K / 1024u, eff = (K % 1024u) / 32u;
    const u32 hwt = eff >> 1, tail_groups = (hwt + 3u) >> 2;
    const u32 ngroups_all = 4u * nfull + tail_groups;  // group = (chunk, q0): 4 stages (byte lanes c)
    const u32 g0 = ks * gper, ngroups = min(ngroups_all, g0 + gper), nst = 4u * ngroups;  // this block: groups [g0, ngroups), stages [4 g0, nst)
    const u32 plane_bytes = N * wpr * 4u;

    const u32x4 rx = g_rsrc(x + (size_t)s0 * K, min(BT, S - s0) * K * 2u);
    const u32x4 rq = g_rsrc(qw, plane_bytes * (u32)BITS);

    gq::LutPools<BITS> L[RF];
    u32 rowoff[RF];  // byte offset of the lane's row in a plane
#pragma unroll
    for (int f = 0; f < RF; f++) {
        const u32 n = min(n0 + 32u * (u32)f + r, N - 1u);  // rows past N are computed on a clamped row and never stored
        rowoff[f] = n * wpr * 4u;
        u32 raw[(1 << BITS) / 2];
        const u32 *lp = reinterpret_cast<const u32 *>(lut + (size_t)n * (1u << BITS));
#pragma unroll
        for (int i = 0; i < (1 << BITS) / 2; i++) raw[i] = lp[i];
        L[f].build(raw);
    }

    f32x16 acc[RF][CF];
#pragma unroll
    for (int f = 0; f < RF; f++)
#pragma unroll
        for (int j = 0; j < CF; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[f][j][e] = 0.f;

    // geometry of group gi: chunk, first K-step q0, half width hw of the chunk's words (16, or the tail's)
    auto group_geom = [&](u32 gi, u32 &chunk, u32 &q0, u32 &hw) {
        if (gi < 4u * nfull) {
            chunk = gi >> 2, q0 = 4u * (gi & 3u), hw = 16u;
        } else {
            chunk = nfull, q0 = 4u * (gi - 4u * nfull), hw = hwt;
        }
    };
    // x stage `st` -> ring slot st % 4: this wave's DSPAN spans of 8 token rows.  Token t keeps logical slot sl at physical
    // slot sl ^ ((t >> 1) & 7): the 16 lanes of a ds_read_b128 group (rows {0-3, 12-15, 20-27} or {4-11, 16-19, 28-31}, same
    // g and q) then cover all 16 slots of the 256-byte bank row once (banks = (a / 4) mod 64; rows are 128 B, so the row
    // parity picks the half).  For span i of a wave (t >> 1) & 7 = 4 (i & 1) + (lane >> 4).
    const u32 dsl0 = (lane & 7u) ^ (lane >> 4), dpart = dsl0 & 3u;
    const u32 drow = 8u * DSPAN * wave + (lane >> 3);
    auto issue_stage = [&](u32 st) {
        u32 chunk, q0, hw;
        group_geom(st >> 2, chunk, q0, hw);
        const u32 c = st & 3u, tpw = 2u * hw;
        const u32 kseg0 = 1024u * chunk + 8u * tpw * c + 8u * q0;  // segment 1 starts 8 hw further
        const u32 lds0 = (u32)(uintptr_t)smem + (st & (P_RING - 1u)) * P_SLOT + wave * (1024u * DSPAN);
#pragma unroll
        for (int i = 0; i < (int)DSPAN; i++) {
            const u32 dseg = ((dsl0 >> 2) ^ (u32)(i & 1)) & 1u;
            g_dma16(rx, lds0 + 1024u * (u32)i, (drow + 8u * (u32)i) * (2u * K) + dpart * 16u + dseg * 16u * hw, 2u * kseg0);
        }
    };
    // plane words of group gi: the request writes INTO the loop-carried registers ("+v": a fresh output register would be
    // copied into them by the compiler right behind the request, before the data has landed)
    u32x4 wn[RF][BITS];
#pragma unroll
    for (int f = 0; f < RF; f++)
#pragma unroll
        for (int p = 0; p < BITS; p++) wn[f][p] = (u32x4){0u, 0u, 0u, 0u};
    auto issue_planes = [&](u32 gi) {
        u32 chunk, q0, hw;
        group_geom(gi, chunk, q0, hw);
#pragma unroll
        for (int f = 0; f < RF; f++)
#pragma unroll
            for (int p = 0; p < BITS; p++)
                asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen"
                             : "+v"(wn[f][p])
                             : "v"(rowoff[f] + 4u * hw * g), "s"(rq), "s"((u32)p * plane_bytes + 4u * (32u * chunk + q0))
                             : "memory");
    };

    // prologue: planes of group 0, stages 0..2
    issue_planes(g0);
    for (u32 st = 4u * g0; st < 4u * g0 + 3u; st++)
        if (st < nst) issue_stage(st);
    g_wait_vm<0>();
    __syncthreads();

    // one group = 4 stages (byte lanes c) on the same plane words.  FULL: all 4 K-steps present (every group of a whole chunk):
    // straight-line code, so the decode of step q + 1 can be scheduled under the MFMAs of step q; tail groups take the branchy copy
    auto stage_top = [&](u32 gi, u32 c) {
        const u32 st = 4u * gi + c;
        if (st + 3u < nst && !(dbg & 2u)) issue_stage(st + 3u);
        if (c == 0u) issue_planes(min(gi + 1u, ngroups - 1u));  // (the last group asks for its own words again: no branch around the request)
    };
    // stage st + 1 landed (this wave's part): at most the 2 DSPAN loads of stages st + 2, st + 3 behind it (more only while a plane
    // request sits between them, which makes the wait stricter, never weaker); the last stages drain the queue
    auto stage_end = [&](u32 st) {
        if (st + 3u < nst)
            g_wait_vm<2 * (int)DSPAN>();
        else
            g_wait_vm<0>();
        __syncthreads();
    };
    auto take_words = [&](u32 (&w)[RF][BITS][4], u32 nq) {
#pragma unroll
        for (int f = 0; f < RF; f++)
#pragma unroll
            for (int p = 0; p < BITS; p++) {
                asm volatile("" : "+v"(wn[f][p]));  // landed: every wait since the request left at most 2 DSPAN younger loads
#pragma unroll
                for (int q = 0; q < 4; q++) w[f][p][q] = (u32)q < nq ? wn[f][p][q] : 0u;
            }
    };
    auto decode_step = [&](const u32 (&w)[RF][BITS][4], int q, u32 shift, f16x8 (&a)[RF]) {
#pragma unroll
        for (int f = 0; f < RF; f++) {
            u32 wq[BITS];
#pragma unroll
            for (int p = 0; p < BITS; p++) wq[p] = w[f][p][q];
            a[f] = decode_byte<BITS>(wq, shift, L[f]);  // (no run-time ablation switch here: a branch would end the scheduling region)
        }
    };
    // a whole group (4 K-steps x 4 byte lanes), software-pipelined by hand: the scheduling region of step (c, q) holds its B
    // reads, its RF * CF MFMAs and the decode of the NEXT step's A fragments, interleaved VPM VALU instructions per MFMA
    // (left to itself the scheduler either keeps the steps apart -- decode, then MFMAs -- or hoists everything and spills)
    constexpr int DEC = BITS == 2 ? 22 : (BITS == 3 ? 30 : 52);  // VALU instructions of one fragment's decode (and a little slack)
    constexpr int VPM = (DEC * RF + RF * CF - 1) / (RF * CF);
    auto group_full = [&](u32 gi) {
        u32 w[RF][BITS][4];
        take_words(w, 4u);
        f16x8 a[RF], an[RF];
        decode_step(w, 0, 24u, a);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < 4; c++) {
            stage_top(gi, (u32)c);
            const unsigned char *xs = smem + ((4u * gi + (u32)c) & (P_RING - 1u)) * P_SLOT + r * 128u;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                __builtin_amdgcn_sched_barrier(0);
                const u32 slot = ((4u * g + (u32)q) ^ ((r >> 1) & 7u)) * 16u;
                f16x8 b[CF];
#pragma unroll
                for (int j = 0; j < CF; j++) b[j] = *reinterpret_cast<const f16x8 *>(xs + (u32)j * 32u * 128u + slot);
                const bool more = !(c == 3 && q == 3);
                if (more) decode_step(w, (q + 1) & 3, 24u - 8u * (u32)(q == 3 ? c + 1 : c), an);
#pragma unroll
                for (int j = 0; j < CF; j++)
#pragma unroll
                    for (int f = 0; f < RF; f++) acc[f][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[f], b[j], acc[f][j], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, CF, 0);
#pragma unroll
                for (int i = 0; i < RF * CF; i++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (more) {
#pragma unroll
                    for (int f = 0; f < RF; f++) a[f] = an[f];
                }
            }
            stage_end(4u * gi + (u32)c);
        }
    };
    for (u32 gi = g0; gi < ngroups; gi++) group_full(gi);  // K % 256 == 0 (host): the tail chunk's groups are full as well

#pragma unroll
    for (int f = 0; f < RF; f++)
#pragma unroll
        for (int j = 0; j < CF; j++) {
            const u32 s = s0 + 32u * (u32)j + r;
            if (s >= S) continue;
#pragma unroll
            for (int rg = 0; rg < 4; rg++) {
                const u32 nn = n0 + 32u * (u32)f + 8u * (u32)rg + 4u * g;
                if (nn >= N) continue;
                if (part) {  // fp32 partial sums of this K range
                    float *dst = part + ((size_t)ks * S + s) * N + nn;
                    if (nn + 3u < N && (N & 3u) == 0u) {
                        *reinterpret_cast<float4 *>(dst) = make_float4(acc[f][j][4 * rg], acc[f][j][4 * rg + 1], acc[f][j][4 * rg + 2], acc[f][j][4 * rg + 3]);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; e++)
                            if (nn + (u32)e < N) dst[e] = acc[f][j][4 * rg + e];
                    }
                    continue;
                }
                uint16_t h[4];
#pragma unroll
                for (int e = 0; e < 4; e++) h[e] = __builtin_bit_cast(uint16_t, (_Float16)acc[f][j][4 * rg + e]);
                uint16_t *dst = out + (size_t)s * N + nn;
                if (nn + 3u < N && (N & 3u) == 0u) {
                    *reinterpret_cast<uint2 *>(dst) = make_uint2((u32)h[0] | ((u32)h[1] << 16), (u32)h[2] | ((u32)h[3] << 16));
                } else {
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        if (nn + (u32)e < N) dst[e] = h[e];
                }
            }
        }
}

// out[i] = fp16(sum over the K ranges of part[ks][i]) in range order: one rounding, as the single-pass kernel
__global__ void __launch_bounds__(256) gemm_reduce_kernel(const float *__restrict__ part, uint16_t *__restrict__ out, u32 total, u32 nks) {
    const u32 i = (blockIdx.x * 256u + threadIdx.x) * 4u;
    if (i >= total) return;
    if (i + 3u < total && (total & 3u) == 0u) {
        float4 a = *reinterpret_cast<const float4 *>(part + i);
        for (u32 k = 1; k < nks; k++) {
            const float4 b = *reinterpret_cast<const float4 *>(part + (size_t)k * total + i);
            a.x += b.x, a.y += b.y, a.z += b.z, a.w += b.w;
        }
        const u32 lo = (u32)__builtin_bit_cast(uint16_t, (_Float16)a.x) | ((u32)__builtin_bit_cast(uint16_t, (_Float16)a.y) << 16);
        const u32 hi = (u32)__builtin_bit_cast(uint16_t, (_Float16)a.z) | ((u32)__builtin_bit_cast(uint16_t, (_Float16)a.w) << 16);
        *reinterpret_cast<uint2 *>(out + i) = make_uint2(lo, hi);
    } else {
        for (u32 e = i; e < total && e < i + 4u; e++) {
            float a = part[e];
            for (u32 k = 1; k < nks; k++) a += part[(size_t)k * total + e];
            out[e] = __builtin_bit_cast(uint16_t, (_Float16)a);
        }
    }
}

template <int BITS, int RF, int CF, int NW>
int launch_gemm_pipe(const void *x, void *out, const uint32_t *qw, const void *lut, u32 S, u32 N, u32 K, hipStream_t s, u32 nks = 1u,
                     float *part = nullptr) {
    static GqPerDeviceOnce once;
    auto kern = ap_gemm_pipe_kernel<BITS, RF, CF, NW>;
    constexpr u32 BT = 32u * CF;
    const size_t smem = (size_t)P_RING * BT * 128u;
    GQ_HIP_CHECK(once.max_dynamic_lds(reinterpret_cast<const void *>(kern), (int)smem));
    constexpr u32 BR = 32u * RF * NW;
    const u32 ngroups = K / 256u, gper = (ngroups + nks - 1u) / nks, nks_eff = (ngroups + gper - 1u) / gper;
    const u32 nbx = (N + BR - 1u) / BR, ntiles = nbx * ((S + BT - 1u) / BT) * nks_eff;
    dim3 grid(8u * ((ntiles + 7u) / 8u)), block(64 * NW);
    hipLaunchKernelGGL(kern, grid, block, smem, s, (const uint16_t *)x, (uint16_t *)out, qw, (const uint16_t *)lut, S, N, K, (u32)gq_env_int("GQ_GEMM_DBG", 0),
                       nbx, ntiles, gper, nks_eff > 1u ? part : nullptr);
    GQ_HIP_CHECK(hipGetLastError());
    if (nks_eff > 1u) {
        const u32 total = S * N;
        hipLaunchKernelGGL(gemm_reduce_kernel, dim3((total / 4u + 256u) / 256u), dim3(256), 0, s, part, (uint16_t *)out, total, nks_eff);
        GQ_HIP_CHECK(hipGetLastError());
    }
    return GQ_OK;
}

// Shape of the wave / block tile (GQ_GEMM_SHAPE forces one: 14, 24, 18 = RF, CF of the table; 0 = the first kernel):
//   1 x 4, 4 waves  128 rows x 128 tokens  the most blocks: small grids
//   2 x 4, 4 waves  256 rows x 128 tokens  2- and 3-bit once there is a block per CU: half the B reads per MFMA
//   1 x 8, 8 waves  256 rows x 256 tokens  4-bit (its decode is 2.5 x the 2-bit one: twice the MFMAs per decoded fragment)
//   (2 x 8 with 4 waves -- one wave per SIMD, 256 accumulator registers -- was built and measured slower: DESIGN.md section 4)
// the pipelined kernel: K-steps in whole groups of 4 (K % 256 == 0: every model width here), planes and a token slab of x
// addressed through 32-bit buffer offsets; other shapes keep the first kernel
bool gemm_pipe_ok(u32 N, u32 K, int bits) {
    return K % 256u == 0u && (uint64_t)bits * N * (K / 8u) < 0x7FFFFFFFull && 256ull * K * 2u < 0x7FFFFFFFull;
}
// K ranges for a short grid (fewer 128 x 128 tiles than CUs), each at least 1024 weights: the 1 x 4 tile with about 1.5 blocks per
// CU; with more than 128 tokens, at 4 bits or (3 bits and K >= 8192), the 8-wave 1 x 8 tile (half the decode per MFMA) with about one
// block per CU when that gives every other CU a block at least (measured: profiles/r03_prefill_gemm.txt, split-K block).  GQ_GEMM_KSPLIT forces the range count, GQ_GEMM_KSHAPE (14 | 18) the tile.
struct GemmSplit {
    u32 nks;
    int shape;
};
GemmSplit gemm_plan_ksplit(u32 S, u32 N, u32 K, int bits) {
    GemmSplit none{1u, 14};
    if (!gemm_pipe_ok(N, K, bits) || gq_env_int("GQ_GEMM_SHAPE", -1) >= 0) return none;
    const int env = gq_env_int("GQ_GEMM_KSPLIT", -1), envshape = gq_env_int("GQ_GEMM_KSHAPE", -1);
    const u32 cus = (u32)gq_cu_count(), t14 = ((N + 127u) / 128u) * ((S + 127u) / 128u), t18 = ((N + 255u) / 256u) * ((S + 255u) / 256u);
    const u32 ngroups = K / 256u, cap = ngroups / 4u < 16u ? ngroups / 4u : 16u;
    if (t14 >= cus && env < 0) return none;
    auto ranges = [&](u32 want) {
        u32 nks = env >= 0 ? (u32)env : want;
        if (nks > cap) nks = cap;
        if (nks <= 1u) return 1u;
        const u32 gper = (ngroups + nks - 1u) / nks;
        return (ngroups + gper - 1u) / gper;
    };
    const u32 n14 = ranges((3u * cus + 2u * t14 - 1u) / (2u * t14)), n18 = ranges((cus + t18 - 1u) / t18);
    int shape = envshape == 14 || envshape == 18 ? envshape : ((bits >= 3 && S > 128u && 2u * t18 * n18 >= cus && (bits == 4 || ngroups >= 32u)) ? 18 : 14);
    const u32 nks = shape == 18 ? n18 : n14;
    if (nks <= 1u) return none;
    return GemmSplit{nks, shape};
}

template <int BITS>
int launch_gemm(const void *x, void *out, const uint32_t *qw, const void *lut, u32 S, u32 N, u32 K, hipStream_t s, float *ws, size_t ws_bytes) {
    const bool pipe = gemm_pipe_ok(N, K, BITS);
    const GemmSplit sp = ws ? gemm_plan_ksplit(S, N, K, BITS) : GemmSplit{1u, 14};
    if (sp.nks > 1u && ws_bytes >= (size_t)sp.nks * S * N * 4u)
        return sp.shape == 18 ? launch_gemm_pipe<BITS, 1, 8, 8>(x, out, qw, lut, S, N, K, s, sp.nks, ws) : launch_gemm_pipe<BITS, 1, 4, 4>(x, out, qw, lut, S, N, K, s, sp.nks, ws);
    int shape = gq_env_int("GQ_GEMM_SHAPE", -1);
    if (shape < 0) {
        const u32 cus = (u32)gq_cu_count();
        const u32 t24 = ((N + 255u) / 256u) * ((S + 127u) / 128u), t18 = ((N + 255u) / 256u) * ((S + 255u) / 256u);
        // measured on the 8B shapes at S = 128 .. 2048 (profiles/r03_prefill_gemm.txt): the 8-wave tile pays from about 0.7
        // blocks per CU on (3- and 4-bit), the two-row tile from one block per CU on
        const bool big = 10u * t18 >= 7u * cus;
        if (BITS == 2)
            shape = t24 >= cus ? 24 : 14;
        else if (BITS == 3)
            shape = big ? 18 : (t24 >= cus ? 24 : 14);
        else
            shape = big ? 18 : 14;
    }
    if (!pipe || shape == 0) return launch_gemm_first<BITS>(x, out, qw, lut, S, N, K, s);
    switch (shape) {
        case 18: return launch_gemm_pipe<BITS, 1, 8, 8>(x, out, qw, lut, S, N, K, s);
        case 24:
            if constexpr (BITS != 4) return launch_gemm_pipe<BITS, 2, 4, 4>(x, out, qw, lut, S, N, K, s);  // (4-bit: does not fit 256 registers)
            [[fallthrough]];
        default: return launch_gemm_pipe<BITS, 1, 4, 4>(x, out, qw, lut, S, N, K, s);
    }
}
}  // namespace

extern "C" size_t gq_anyprec_gemm_ws_bytes(uint32_t S, uint32_t N, uint32_t K, int bits) {
    if (bits < 2 || bits > 4 || K == 0 || S == 0 || N == 0) return 0;
    const u32 nks = gemm_plan_ksplit(S, N, K, bits).nks;
    return nks > 1u ? (size_t)nks * S * N * 4u : 0;
}

extern "C" int gq_anyprec_gemm_ws(const void *x, void *out, const uint32_t *qweight, const void *lut, uint32_t S, uint32_t N, uint32_t K,
                                  int bits, void *workspace, size_t ws_bytes, void *stream) {
    if (bits < 2 || bits > 4) return gq_fail(GQ_ENOTSUP, "gq_anyprec_gemm: bits must be 2, 3 or 4 (wider: dequantise + GEMM).");
    if (K == 0 || K % 64u) return gq_fail(GQ_ENOTSUP, "gq_anyprec_gemm: K must be a positive multiple of 64.");
    if (S == 0 || N == 0) return gq_fail(GQ_EINVAL, "gq_anyprec_gemm: empty problem.");
    if (!x || !out || !qweight || !lut) return gq_fail(GQ_EINVAL, "null pointer argument.");
    if (((uintptr_t)x | (uintptr_t)qweight) & 15u || ((uintptr_t)out & 7u) || ((uintptr_t)lut & 3u))
        return gq_fail(GQ_EINVAL, "gq_anyprec_gemm: x / qweight must be 16-byte, out 8-byte aligned.");
    if (workspace && ((uintptr_t)workspace & 15u)) return gq_fail(GQ_EINVAL, "gq_anyprec_gemm_ws: workspace must be 16-byte aligned.");
    hipStream_t s = (hipStream_t)stream;
    float *ws = (float *)workspace;
    switch (bits) {
        case 2: return launch_gemm<2>(x, out, qweight, lut, S, N, K, s, ws, ws_bytes);
        case 3: return launch_gemm<3>(x, out, qweight, lut, S, N, K, s, ws, ws_bytes);
        default: return launch_gemm<4>(x, out, qweight, lut, S, N, K, s, ws, ws_bytes);
    }
}

extern "C" int gq_anyprec_gemm(const void *x, void *out, const uint32_t *qweight, const void *lut, uint32_t S, uint32_t N,
                               uint32_t K, int bits, void *stream) {
    return gq_anyprec_gemm_ws(x, out, qweight, lut, S, N, K, bits, nullptr, 0, stream);
}
