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

template <int BITS>
__global__ void __launch_bounds__(256, 2) ap_gemm_kernel(const uint16_t *__restrict__ x, uint16_t *__restrict__ out,
                                                          const u32 *__restrict__ qw, const uint16_t *__restrict__ lut, u32 S, u32 N,
                                                          u32 K, u32 dbg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // 2 stages of [BS][ROWB]
    const u32 tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u, r = lane & 31u, g = lane >> 5;
    const u32 n0 = blockIdx.x * BN + wave * 32u, s0 = blockIdx.y * BS;
    const u32 wpr = K / 32u, nfull = K / 1024u, eff = (K % 1024u) / 32u, nchunks = nfull + (eff ? 1u : 0u);
    const u32 n = min(n0 + r, N - 1u);  // rows past N are computed on a clamped row and never stored
    const size_t plane_stride = (size_t)N * wpr;

    gq::LutPools<BITS> L;
    {
        u32 raw[(1 << BITS) / 2];
        const u32 *lp = reinterpret_cast<const u32 *>(lut + (size_t)n * (1u << BITS));
#pragma unroll
        for (int i = 0; i < (1 << BITS) / 2; i++) raw[i] = lp[i];
        L.build(raw);
    }

    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[j][e] = 0.f;

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
    u32 w[BITS][8];
    while (cur.chunk < nchunks) {
        const u32 tpw = cur.chunk < nfull ? 32u : eff, hw = tpw >> 1;
        stage_geom(cur, k0, k1, nq);
        if (cur.c == 0u) {  // the plane words t = hw g + q0 .. + nq - 1 of this lane's row (two 16-byte loads per plane)
            const u32 *base = qw + (size_t)n * wpr + 32u * cur.chunk + hw * g + cur.q0;
#pragma unroll
            for (int p = 0; p < BITS; p++) {
                const u32 *pp = base + (size_t)p * plane_stride;
                if (nq == 8u && (hw & 3u) == 0u && (wpr & 3u) == 0u) {
                    const uint4 a = *reinterpret_cast<const uint4 *>(pp), b = *reinterpret_cast<const uint4 *>(pp + 4);
                    w[p][0] = a.x, w[p][1] = a.y, w[p][2] = a.z, w[p][3] = a.w;
                    w[p][4] = b.x, w[p][5] = b.y, w[p][6] = b.z, w[p][7] = b.w;
                } else {
#pragma unroll
                    for (int q = 0; q < 8; q++) w[p][q] = (u32)q < nq ? pp[q] : 0u;
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
                u32 wq[BITS];
#pragma unroll
                for (int p = 0; p < BITS; p++) wq[p] = w[p][q];
                f16x8 a;
                if (dbg & 1u) {  // ablation (GQ_GEMM_DBG=1): no decode
                    typedef u32 u32x4 __attribute__((ext_vector_type(4)));
                    u32x4 t = {wq[0], wq[1], wq[0] ^ shift, wq[1] + (u32)q};
                    a = __builtin_bit_cast(f16x8, t);
                } else {
                    a = decode_byte<BITS>(wq, shift, L);
                }
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const f16x8 b = *reinterpret_cast<const f16x8 *>(xs + (u32)j * 32u * ROWB + (u32)q * 16u);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
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
    for (int j = 0; j < 4; j++) {
        const u32 s = s0 + 32u * (u32)j + r;
        if (s >= S) continue;
#pragma unroll
        for (int rg = 0; rg < 4; rg++) {
            const u32 nn = n0 + 8u * (u32)rg + 4u * g;
            if (nn >= N) continue;
            uint16_t h[4];
#pragma unroll
            for (int e = 0; e < 4; e++) h[e] = __builtin_bit_cast(uint16_t, (_Float16)acc[j][4 * rg + e]);
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
int launch_gemm(const void *x, void *out, const uint32_t *qw, const void *lut, u32 S, u32 N, u32 K, hipStream_t s) {
    static GqPerDeviceOnce once;
    auto kern = ap_gemm_kernel<BITS>;
    const size_t smem = 2u * STAGE_BYTES;
    GQ_HIP_CHECK(once.max_dynamic_lds(reinterpret_cast<const void *>(kern), (int)smem));
    dim3 grid((N + BN - 1u) / BN, (S + BS - 1u) / BS), block(256);
    hipLaunchKernelGGL(kern, grid, block, smem, s, (const uint16_t *)x, (uint16_t *)out, qw, (const uint16_t *)lut, S, N, K, (u32)gq_env_int("GQ_GEMM_DBG", 0));
    GQ_HIP_CHECK(hipGetLastError());
    return GQ_OK;
}
}  // namespace

extern "C" int gq_anyprec_gemm(const void *x, void *out, const uint32_t *qweight, const void *lut, uint32_t S, uint32_t N,
                               uint32_t K, int bits, void *stream) {
    if (bits < 2 || bits > 4) return gq_fail(GQ_ENOTSUP, "gq_anyprec_gemm: bits must be 2, 3 or 4 (wider: dequantise + GEMM).");
    if (K == 0 || K % 64u) return gq_fail(GQ_ENOTSUP, "gq_anyprec_gemm: K must be a positive multiple of 64.");
    if (S == 0 || N == 0) return gq_fail(GQ_EINVAL, "gq_anyprec_gemm: empty problem.");
    if (!x || !out || !qweight || !lut) return gq_fail(GQ_EINVAL, "null pointer argument.");
    if (((uintptr_t)x | (uintptr_t)qweight) & 15u || ((uintptr_t)out & 7u) || ((uintptr_t)lut & 3u))
        return gq_fail(GQ_EINVAL, "gq_anyprec_gemm: x / qweight must be 16-byte, out 8-byte aligned.");
    hipStream_t s = (hipStream_t)stream;
    switch (bits) {
        case 2: return launch_gemm<2>(x, out, qweight, lut, S, N, K, s);
        case 3: return launch_gemm<3>(x, out, qweight, lut, S, N, K, s);
        default: return launch_gemm<4>(x, out, qweight, lut, S, N, K, s);
    }
}
