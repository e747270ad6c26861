// lutgemm.hip -- LUT-GEMM (BCQ) GEMV for gfx950.  Replaces nqmv_bias (inference/ap_gemv/lutgemm.cu:24-149).
//
// The reference launches one block per 32-activation tile and atomically adds fp16 tile results into the output, in
// an undefined order.  Here one block owns a slice of the outputs and walks the tiles in ASCENDING order (one valid
// execution of the reference's atomics, and deterministic): per group of G tiles the 4 x 256 sign-sum tables of every
// tile are built in LDS with the reference's fp16 operation order (lutgemm.cu:40-78); the four waves of a block compute
// the per-tile results of its 64 outputs in parallel (lutgemm.cu:94-145, same fp16 multiply / add sequence) and wave 0
// adds them to the running sums in tile order.  Weight words are read
// coalesced (consecutive outputs of one (tile, plane) are consecutive in memory).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gq_internal.h"

namespace {
typedef uint32_t u32;
typedef _Float16 h16;
__device__ __forceinline__ h16 u2h(uint16_t h) { return __builtin_bit_cast(h16, h); }
__device__ __forceinline__ uint16_t h2u(h16 h) { return __builtin_bit_cast(uint16_t, h); }

constexpr int G = 16;   // tiles per LDS round: 16 * 4 * 256 halves = 32 KiB
constexpr int OB = 64;  // outputs per block; the 4 waves of a block take the tiles of a round in an interleaved way
constexpr int TPW = G / 4;  // tiles per wave per round

template <int BITS>
__global__ void __launch_bounds__(256) lutgemm_kernel(const uint16_t *x, uint16_t *out, const u32 *W, const uint16_t *alpha,
                                                      const uint16_t *q_bias, u32 N, u32 K, u32 group_size) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t (*lut)[4][256] = reinterpret_cast<uint16_t (*)[4][256]>(smem);                          // [G][4][256]
    uint16_t (*obuf)[G][OB] = reinterpret_cast<uint16_t (*)[G][OB]>(smem + G * 4 * 256 * 2);        // [2][G][OB]
    uint16_t *xs = reinterpret_cast<uint16_t *>(smem + G * 4 * 256 * 2 + 2 * G * OB * 2);           // [K]
    const u32 tid = threadIdx.x, q = tid >> 6, mo = tid & 63u;
    const u32 m = blockIdx.x * OB + mo;  // the output of this lane (all 4 waves of the block share the 64 outputs)
    const bool ok = m < N;
    for (u32 i = tid; i < K / 2u; i += 256u) reinterpret_cast<u32 *>(xs)[i] = reinterpret_cast<const u32 *>(x)[i];
    h16 acc = (ok && q == 0) ? u2h(out[m]) : (h16)0;
    const u32 ntiles = K / 32u;
    u32 par = 0;
    __syncthreads();
    for (u32 kt0 = 0; kt0 < ntiles; kt0 += G, par ^= 1u) {
        // the weight words / scales of this wave's tiles are requested first: their latency overlaps the table build
        u32 wq[TPW][BITS];
        uint16_t al[TPW], qb[TPW];
#pragma unroll
        for (int j = 0; j < TPW; j++) {
            const u32 kt = kt0 + q + 4u * (u32)j;
            const bool v = ok && kt < ntiles;
            const u32 g = v ? (kt * 32u) / group_size : 0u;
#pragma unroll
            for (int b = 0; b < BITS; b++) wq[j][b] = v ? W[((size_t)kt * BITS + b) * N + m] : 0u;
            al[j] = v ? alpha[(size_t)g * BITS * N + m] : (uint16_t)0;
            qb[j] = v ? q_bias[(size_t)g * N + m] : (uint16_t)0;
        }
        // build: thread -> (tile gi, table y, v < 64) for 4 passes, then the two doubling steps
        for (u32 e = tid; e < (u32)G * 4u * 64u; e += 256u) {
            const u32 v = e & 63u, y = (e >> 6) & 3u, gi = e >> 8;
            if (kt0 + gi < ntiles) {
                const uint16_t *xi = xs + 32u * (kt0 + gi) + 8u * y;
                h16 a = (h16)0;
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const h16 sg = ((v >> i) & 1u) ? (h16)1.0f : (h16)-1.0f;
                    const h16 term = sg * u2h(xi[i]);
                    a = i == 0 ? term : a + term;
                }
                const h16 i6 = (h16)2.0f * u2h(xi[6]), i7 = (h16)2.0f * u2h(xi[7]);
                const h16 b = a + i6;        // v + 64
                lut[gi][y][v] = h2u(a);
                lut[gi][y][v + 64] = h2u(b);
                lut[gi][y][v + 128] = h2u(a + i7);
                lut[gi][y][v + 192] = h2u(b + i7);
            }
        }
        __syncthreads();
        // per-tile results: wave q takes tiles gi = q, q + 4, .. of the round (independent of the running sum)
#pragma unroll
        for (int j = 0; j < TPW; j++) {
            const u32 gi = q + 4u * (u32)j;
            if (kt0 + gi >= ntiles) break;
            h16 all = (h16)0;
#pragma unroll
            for (int y = 0; y < 4; y++) all = all + u2h(lut[gi][y][255]);
            h16 o = (h16)0 + u2h(qb[j]) * all;
            h16 a = u2h(al[j]);
#pragma unroll
            for (int b = 0; b < BITS; b++) {
                const u32 w = wq[j][b];
                h16 t = (h16)0;
#pragma unroll
                for (int y = 0; y < 4; y++) t = t + u2h(lut[gi][y][(w >> (8 * y)) & 255u]);
                o = o + a * t;
                a = a * (h16)2.0f;
            }
            obuf[par][gi][mo] = h2u(o);
        }
        __syncthreads();
        // ascending-tile chain of the fp16 adds (the reference's atomicAdd sequence in tile order)
        if (q == 0)
            for (u32 gi = 0; gi < (u32)G && kt0 + gi < ntiles; gi++) acc = acc + u2h(obuf[par][gi][mo]);
    }
    if (ok && q == 0) out[m] = h2u(acc);
}

// ---- two-kernel form: the tables do not depend on the output row, so they are built ONCE per call into a workspace
// (K / 32 tiles x 4 x 256 halves = 64 bytes per activation) instead of by every block of the GEMV (64..448 times), and the
// GEMV blocks copy them into LDS.  Same arithmetic, same tile order: bit-identical to the single-kernel form.
__global__ void __launch_bounds__(256) lutgemm_tables_kernel(const uint16_t *x, uint16_t *tab, u32 ntiles) {
    const u32 kt = blockIdx.x, tid = threadIdx.x, v = tid & 63u, y = tid >> 6;
    if (kt >= ntiles) return;
    const uint16_t *xi = x + 32u * kt + 8u * y;
    h16 a = (h16)0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const h16 sg = ((v >> i) & 1u) ? (h16)1.0f : (h16)-1.0f;
        const h16 term = sg * u2h(xi[i]);
        a = i == 0 ? term : a + term;
    }
    const h16 i6 = (h16)2.0f * u2h(xi[6]), i7 = (h16)2.0f * u2h(xi[7]);
    const h16 b = a + i6;
    uint16_t *t = tab + ((size_t)kt * 4u + y) * 256u;
    t[v] = h2u(a);
    t[v + 64] = h2u(b);
    t[v + 128] = h2u(a + i7);
    t[v + 192] = h2u(b + i7);
}

// OBK outputs per block (64 or 16): with 16, a wave covers 4 tiles x 16 outputs per instruction, so a small N still
// fills the chip (N = 4096: 256 blocks instead of 64) -- the tables come from memory, a smaller block repeats no work.
template <int BITS, int OBK>
__global__ void __launch_bounds__(256) lutgemm_apply_kernel(const uint16_t *tab, uint16_t *out, const u32 *W, const uint16_t *alpha,
                                                            const uint16_t *q_bias, u32 N, u32 K, u32 group_size) {
    constexpr int TS = 64 / OBK;          // tile slots per wave instruction
    constexpr int TPWK = G / (4 * TS);    // tile iterations per wave per round
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t (*lut)[4][256] = reinterpret_cast<uint16_t (*)[4][256]>(smem);                      // [G][4][256]
    uint16_t (*obuf)[G][OBK] = reinterpret_cast<uint16_t (*)[G][OBK]>(smem + G * 4 * 256 * 2);  // [2][G][OBK]
    const u32 tid = threadIdx.x, q = tid >> 6, l = tid & 63u, mo = l % (u32)OBK, ts = l / (u32)OBK;
    const u32 m = blockIdx.x * OBK + mo;
    const bool ok = m < N;
    const bool chain = q == 0 && ts == 0;  // the lanes that carry the running sums
    h16 acc = (ok && chain) ? u2h(out[m]) : (h16)0;
    const u32 ntiles = K / 32u;
    u32 par = 0;
    constexpr u32 CH = G * 4 * 256 * 2 / 16 / 256;  // 16-byte units of a round's tables per thread (8)
    uint4 tnext[CH];
    u32 wq[TPWK][BITS], wqn[TPWK][BITS];
    uint16_t al[TPWK], qb[TPWK], aln[TPWK], qbn[TPWK];
    auto fetch = [&](u32 kt0) {  // tables, weight words and scales of the round starting at tile kt0
#pragma unroll
        for (u32 c = 0; c < CH; c++) {
            const u32 unit = tid + c * 256u;  // 16-byte unit of the round: tile = unit / 128
            const bool v = kt0 + unit / 128u < ntiles;
            tnext[c] = v ? reinterpret_cast<const uint4 *>(tab + (size_t)kt0 * 1024u)[unit] : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int j = 0; j < TPWK; j++) {
            const u32 kt = kt0 + (q + 4u * (u32)j) * (u32)TS + ts;
            const bool v = ok && kt < ntiles;
            const u32 g = v ? (kt * 32u) / group_size : 0u;
#pragma unroll
            for (int b = 0; b < BITS; b++) wqn[j][b] = v ? W[((size_t)kt * BITS + b) * N + m] : 0u;
            aln[j] = v ? alpha[(size_t)g * BITS * N + m] : (uint16_t)0;
            qbn[j] = v ? q_bias[(size_t)g * N + m] : (uint16_t)0;
        }
    };
    fetch(0u);
    for (u32 kt0 = 0; kt0 < ntiles; kt0 += G, par ^= 1u) {
#pragma unroll
        for (u32 c = 0; c < CH; c++) reinterpret_cast<uint4 *>(smem)[tid + c * 256u] = tnext[c];
#pragma unroll
        for (int j = 0; j < TPWK; j++) {
#pragma unroll
            for (int b = 0; b < BITS; b++) wq[j][b] = wqn[j][b];
            al[j] = aln[j];
            qb[j] = qbn[j];
        }
        __syncthreads();
        if (kt0 + G < ntiles) fetch(kt0 + G);  // the next round travels while this one is computed
#pragma unroll
        for (int j = 0; j < TPWK; j++) {
            const u32 gi = (q + 4u * (u32)j) * (u32)TS + ts;
            if (kt0 + gi < ntiles) {
                h16 all = (h16)0;
#pragma unroll
                for (int y = 0; y < 4; y++) all = all + u2h(lut[gi][y][255]);
                h16 o = (h16)0 + u2h(qb[j]) * all;
                h16 a = u2h(al[j]);
#pragma unroll
                for (int b = 0; b < BITS; b++) {
                    const u32 w = wq[j][b];
                    h16 t = (h16)0;
#pragma unroll
                    for (int y = 0; y < 4; y++) t = t + u2h(lut[gi][y][(w >> (8 * y)) & 255u]);
                    o = o + a * t;
                    a = a * (h16)2.0f;
                }
                obuf[par][gi][mo] = h2u(o);
            }
        }
        __syncthreads();
        if (chain)
            for (u32 gi = 0; gi < (u32)G && kt0 + gi < ntiles; gi++) acc = acc + u2h(obuf[par][gi][mo]);
    }
    if (ok && chain) out[m] = h2u(acc);
}
}  // namespace

extern "C" int gq_lutgemm_gemv(const void *x, void *out, const uint32_t *qweight, const void *alpha, const void *q_bias,
                               uint32_t N, uint32_t K, int bits, int group_size, void *stream) {
    if (bits < 1 || bits > 8) return gq_fail(GQ_EINVAL, "Bitwidth must be between 1 and 8.");
    if (!x || !out || !qweight || !alpha || !q_bias) return gq_fail(GQ_EINVAL, "null pointer argument.");
    if (K == 0 || K % 32u || N == 0) return gq_fail(GQ_EINVAL, "need N > 0 and K a positive multiple of 32.");
    if (group_size <= 0 || K % (uint32_t)group_size || (uint32_t)group_size % 32u)
        return gq_fail(GQ_EINVAL, "group_size must be a multiple of 32 that divides input_feat.");
    const size_t smem = (size_t)G * 4 * 256 * 2 + 2 * (size_t)G * OB * 2 + (size_t)K * 2u;
    if (smem > 160u * 1024u) return gq_fail(GQ_ENOTSUP, "input_feat too large for the LDS activation copy.");
    if (((uintptr_t)x) & 3u) return gq_fail(GQ_EINVAL, "input must be 4-byte aligned.");
    const dim3 grid((N + (u32)OB - 1u) / (u32)OB), block(256);
    hipStream_t st = (hipStream_t)stream;
#define GQ_LG_CASE(B)                                                                                                    \
    case B: {                                                                                                            \
        static GqPerDeviceOnce once;                                                                                         \
        GQ_HIP_CHECK(once.max_dynamic_lds(reinterpret_cast<const void *>(lutgemm_kernel<B>), 160 * 1024)); \
        hipLaunchKernelGGL(lutgemm_kernel<B>, grid, block, smem, st, (const uint16_t *)x, (uint16_t *)out, qweight,      \
                           (const uint16_t *)alpha, (const uint16_t *)q_bias, N, K, (u32)group_size);                    \
    } break;
    switch (bits) {
        GQ_LG_CASE(1) GQ_LG_CASE(2) GQ_LG_CASE(3) GQ_LG_CASE(4) GQ_LG_CASE(5) GQ_LG_CASE(6) GQ_LG_CASE(7) GQ_LG_CASE(8)
    }
#undef GQ_LG_CASE
    GQ_HIP_CHECK(hipGetLastError());
    return GQ_OK;
}

extern "C" int gq_lutgemm_gemv_ws(const void *x, void *out, const uint32_t *qweight, const void *alpha, const void *q_bias,
                                  uint32_t N, uint32_t K, int bits, int group_size, void *workspace, uint64_t workspace_bytes,
                                  void *stream) {
    if (bits < 1 || bits > 8) return gq_fail(GQ_EINVAL, "Bitwidth must be between 1 and 8.");
    if (!x || !out || !qweight || !alpha || !q_bias || !workspace) return gq_fail(GQ_EINVAL, "null pointer argument.");
    if (K == 0 || K % 32u || N == 0) return gq_fail(GQ_EINVAL, "need N > 0 and K a positive multiple of 32.");
    if (group_size <= 0 || K % (uint32_t)group_size || (uint32_t)group_size % 32u)
        return gq_fail(GQ_EINVAL, "group_size must be a multiple of 32 that divides input_feat.");
    if (workspace_bytes < (uint64_t)K * 64u || ((uintptr_t)workspace & 15u))
        return gq_fail(GQ_EINVAL, "workspace must hold 64 bytes per input feature, 16-byte aligned.");
    const u32 ntiles = K / 32u;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(lutgemm_tables_kernel, dim3(ntiles), dim3(256), 0, st, (const uint16_t *)x, (uint16_t *)workspace, ntiles);
    // few outputs: 16 per block, so that there are at least as many blocks as CUs
    const bool small = (N + (u32)OB - 1u) / (u32)OB < (u32)gq_env_int("GQ_LG_SMALL_BLOCKS", 256);
    const u32 obk = small ? 16u : (u32)OB;
    const size_t smem = (size_t)G * 4 * 256 * 2 + 2 * (size_t)G * obk * 2;
    const dim3 grid((N + obk - 1u) / obk), block(256);
#define GQ_LGW_CASE(B)                                                                                                   \
    case B:                                                                                                              \
        if (small)                                                                                                       \
            hipLaunchKernelGGL((lutgemm_apply_kernel<B, 16>), grid, block, smem, st, (const uint16_t *)workspace,        \
                               (uint16_t *)out, qweight, (const uint16_t *)alpha, (const uint16_t *)q_bias, N, K,        \
                               (u32)group_size);                                                                         \
        else                                                                                                             \
            hipLaunchKernelGGL((lutgemm_apply_kernel<B, OB>), grid, block, smem, st, (const uint16_t *)workspace,        \
                               (uint16_t *)out, qweight, (const uint16_t *)alpha, (const uint16_t *)q_bias, N, K,        \
                               (u32)group_size);                                                                         \
        break;
    switch (bits) {
        GQ_LGW_CASE(1) GQ_LGW_CASE(2) GQ_LGW_CASE(3) GQ_LGW_CASE(4) GQ_LGW_CASE(5) GQ_LGW_CASE(6) GQ_LGW_CASE(7) GQ_LGW_CASE(8)
    }
#undef GQ_LGW_CASE
    GQ_HIP_CHECK(hipGetLastError());
    return GQ_OK;
}
