/*
 * gq_oracle.c -- CPU restatement of GuidedQuant's quantized-linear decode path.
 *
 * THIS FILE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the smoke
 * check in __graft_entry__.py and bench.py's cpu_baseline leg may load it.  The
 * product path (guidedquant_amd/) never links or calls anything in here.
 *
 * Every function cites the reference file:line (relative to the upstream
 * snu-mllab/GuidedQuant tree) whose behaviour it restates.  Nothing here is
 * copied from the reference: the CUDA kernels are warp-32 SIMT programs, this is
 * a scalar re-derivation of the arithmetic they perform, including the exact
 * order of the fp16 roundings (the reference accumulates in fp16).
 *
 * Parity status (see DESIGN.md "Oracle"):
 *   - AP packed format (pack/unpack), AP dequant:  PINNED against golden vectors
 *     generated from the reference's own pack.py / finetune_utils.py
 *     (tests/golden/, tests/make_golden.py).
 *   - AP GEMV fp16 rounding order: restated from anyprec.cu:372-542; the
 *     reference has no test or CPU implementation of it ("parity unpinned" for
 *     the rounding order; values pinned through the dequant goldens to within
 *     fp16 accumulation error).
 *   - LUT-GEMM: restated from lutgemm.cu:24-149, parity unpinned (the reference
 *     holds no producer, test or CPU statement of it).
 *   - QTIP decode: PINNED against decode_compressed goldens.
 *   - Hadamard: PINNED against matmul_hadU goldens.
 *
 * Build: make -C oracle   (gcc -O2 -fopenmp -shared -fPIC)
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------- */
/* IEEE binary16 helpers (software, exact).                                   */
/* ------------------------------------------------------------------------- */

static inline double h2d(uint16_t h) {
    uint32_t sign = (uint32_t)(h >> 15);
    uint32_t e = (h >> 10) & 0x1F;
    uint32_t m = h & 0x3FF;
    double v;
    if (e == 0) {
        v = ldexp((double)m, -24);
    } else if (e == 31) {
        v = m ? NAN : INFINITY;
    } else {
        v = ldexp((double)(m | 0x400), (int)e - 25);
    }
    return sign ? -v : v;
}

/* double -> half, round-to-nearest-even, subnormals kept, directly from the
 * 53-bit significand (no intermediate float, so no double rounding). */
static inline uint16_t d2h(double d) {
    uint64_t u;
    memcpy(&u, &d, 8);
    uint16_t sign = (uint16_t)((u >> 48) & 0x8000);
    int e = (int)((u >> 52) & 0x7FF);
    uint64_t m = u & 0xFFFFFFFFFFFFFULL;
    if (e == 0x7FF) return (uint16_t)(sign | (m ? 0x7E00 : 0x7C00));
    if (e == 0) return sign; /* double zero / subnormal: far below half range */
    m |= 1ULL << 52;
    int he = e - 1023 + 15; /* biased half exponent if normal */
    int shift;
    if (he >= 1) {
        shift = 42;
    } else {
        shift = 1051 - e; /* align to the 2^-24 subnormal grid */
        if (shift > 54) return sign;
        he = 0;
    }
    uint64_t q = m >> shift;
    uint64_t rem = m & ((1ULL << shift) - 1);
    uint64_t half = 1ULL << (shift - 1);
    if (rem > half || (rem == half && (q & 1))) q++;
    uint32_t bits;
    if (he >= 1)
        bits = ((uint32_t)(he - 1) << 10) + (uint32_t)q; /* q carries the hidden bit */
    else
        bits = (uint32_t)q;
    if (bits >= 0x7C00) bits = 0x7C00;
    return (uint16_t)(sign | bits);
}

/* fused multiply-add in half: one rounding (== CUDA __hfma / AMD v_fma_f16).
 * a*b is exact in double (22 significant bits); the sum is exact in double for
 * every case in which it can influence the half rounding. */
static inline uint16_t h_fma(uint16_t a, uint16_t b, uint16_t c) {
    return d2h(h2d(a) * h2d(b) + h2d(c));
}
static inline uint16_t h_add(uint16_t a, uint16_t b) { return d2h(h2d(a) + h2d(b)); }
static inline uint16_t h_mul(uint16_t a, uint16_t b) { return d2h(h2d(a) * h2d(b)); }

/* exported for tests */
uint16_t gq_oracle_d2h(double d) { return d2h(d); }
double gq_oracle_h2d(uint16_t h) { return h2d(h); }
uint16_t gq_oracle_hfma(uint16_t a, uint16_t b, uint16_t c) { return h_fma(a, b, c); }

/* ------------------------------------------------------------------------- */
/* Any-Precision packed weight format.                                        */
/* Reference writer: any_precision/quantization/pack.py:304-321               */
/* (pack_single_weight), :12-83 (_permute_bitmaps, _calculate_new_indices).   */
/* Reference reader: inference/ap_gemv/anyprec.cu:446 (word index) and the    */
/* activation index at anyprec.cu:498.                                        */
/*                                                                            */
/* Closed form: weight e of row n, plane p (0 = MSB of the code):             */
/*   full = (K/1024)*1024                                                     */
/*   e <  full: chunk=e/1024, r=e%1024, tpw=32,            base=32*chunk      */
/*   e >= full:               r=e-full, tpw=(K-full)/32,   base=full/32       */
/*   c = r/(8*tpw), t = (r%(8*tpw))/8, j = r%8                                */
/*   bit lives in word base+t of qweight[p][n][:], bit position 31-(8c+j).    */
/* ------------------------------------------------------------------------- */

static inline void ap_locate(uint32_t K, uint32_t e, uint32_t *word, uint32_t *bitpos) {
    uint32_t full = (K / 1024u) * 1024u;
    uint32_t r, tpw, base;
    if (e < full) {
        r = e % 1024u;
        tpw = 32u;
        base = 32u * (e / 1024u);
    } else {
        r = e - full;
        tpw = (K - full) / 32u;
        base = full / 32u;
    }
    uint32_t c = r / (8u * tpw);
    uint32_t t = (r % (8u * tpw)) / 8u;
    uint32_t j = r % 8u;
    *word = base + t;
    *bitpos = 31u - (8u * c + j);
}

/* codes uint8[N][K] -> qweight uint32[bits][N][K/32]; returns 0 or -1. */
int gq_oracle_ap_pack(const uint8_t *codes, uint32_t N, uint32_t K, int bits, uint32_t *qweight) {
    if (K % 32u || bits < 1 || bits > 8) return -1;
    uint32_t wpr = K / 32u;
    memset(qweight, 0, (size_t)bits * N * wpr * 4u);
    for (uint32_t n = 0; n < N; n++)
        for (uint32_t e = 0; e < K; e++) {
            uint32_t word, bp;
            ap_locate(K, e, &word, &bp);
            uint8_t code = codes[(size_t)n * K + e];
            for (int p = 0; p < bits; p++)
                if ((code >> (bits - 1 - p)) & 1u)
                    qweight[((size_t)p * N + n) * wpr + word] |= 1u << bp;
        }
    return 0;
}

/* inverse of the above (pack.py:324-347 unpack_single_weight).  `nplanes` is the
 * number of planes actually used (the first `nplanes` of the stored tensor,
 * any-precision property); plane stride is N*K/32 words (anyprec.cu:446). */
int gq_oracle_ap_unpack(const uint32_t *qweight, uint32_t N, uint32_t K, int nplanes, uint8_t *codes) {
    if (K % 32u || nplanes < 1 || nplanes > 8) return -1;
    uint32_t wpr = K / 32u;
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < (int64_t)N; n++)
        for (uint32_t e = 0; e < K; e++) {
            uint32_t word, bp;
            ap_locate(K, e, &word, &bp);
            uint8_t code = 0;
            for (int p = 0; p < nplanes; p++)
                code = (uint8_t)((code << 1) | ((qweight[((size_t)p * N + n) * wpr + word] >> bp) & 1u));
            codes[(size_t)n * K + e] = code;
        }
    return 0;
}

/* anyprec_dequant: W[n][e] = lut[n][code(n,e)]   (anyprec.cu:294-359,
 * gemv.cu:109-134; python statement finetune_utils.py:19-38). */
int gq_oracle_ap_dequant(const uint32_t *qweight, const uint16_t *lut, uint32_t N, uint32_t K, int bits,
                         uint16_t *W) {
    if (K % 32u || bits < 1 || bits > 8) return -1;
    uint8_t *codes = (uint8_t *)malloc((size_t)N * K);
    if (!codes) return -2;
    gq_oracle_ap_unpack(qweight, N, K, bits, codes);
    uint32_t nc = 1u << bits;
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < (int64_t)N; n++)
        for (uint32_t e = 0; e < K; e++) W[(size_t)n * K + e] = lut[(size_t)n * nc + codes[(size_t)n * K + e]];
    free(codes);
    return 0;
}

/* Exact (fp64-accumulated) GEMV: y[m][n] = sum_e x[m][e] * lut[n][code(n,e)]. */
int gq_oracle_ap_gemv_f64(const uint16_t *x, const uint32_t *qweight, const uint16_t *lut, uint32_t M, uint32_t N,
                          uint32_t K, int bits, double *y) {
    if (K % 32u || bits < 1 || bits > 8) return -1;
    uint8_t *codes = (uint8_t *)malloc((size_t)N * K);
    if (!codes) return -2;
    gq_oracle_ap_unpack(qweight, N, K, bits, codes);
    uint32_t nc = 1u << bits;
    double *xd = (double *)malloc(sizeof(double) * (size_t)M * K);
    for (size_t i = 0; i < (size_t)M * K; i++) xd[i] = h2d(x[i]);
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < (int64_t)N; n++) {
        double lv[256];
        for (uint32_t c = 0; c < nc; c++) lv[c] = h2d(lut[(size_t)n * nc + c]);
        for (uint32_t m = 0; m < M; m++) {
            double acc = 0.0;
            for (uint32_t e = 0; e < K; e++) acc += xd[(size_t)m * K + e] * lv[codes[(size_t)n * K + e]];
            y[(size_t)m * N + n] = acc;
        }
    }
    free(xd);
    free(codes);
    return 0;
}

/* Order-faithful fp16 GEMV == matmul_kbit_32<maxm,bits,use_ksplit>
 * (anyprec.cu:372-542), launched by anyprec_matmul (anyprec.cu:591-620).
 *
 * One "virtual lane" t (a CUDA lane, 32 per output row) owns word t of every
 * 1024-weight chunk.  Per chunk it runs ONE half2 accumulator over its 32
 * weights: 16 __hfma2 in the order byte c = 3,2,1,0 ; pair k = 0..3
 * (anyprec.cu:495-504: `for j=3..0`, `for k=0..3`), .x lane = even weight,
 * .y lane = odd weight; then `partial += sum.x + sum.y` in half
 * (anyprec.cu:505); chunks in ascending order; the tail chunk only on lanes
 * t < (K%1024)/32 (anyprec.cu:433-436).  Finally a 5-step __shfl_down tree
 * 16,8,4,2,1 in half (anyprec.cu:363-370,510-512) and lane 0 stores
 * (anyprec.cu:532-541).
 *
 * use_ksplit (anyprec.cu:611: M==1 && K>4096 && bits>=7): chunks are split in
 * groups of 4 over threadIdx.z, each group tree-reduced separately and then
 * atomically added in half into a zeroed shared cell (anyprec.cu:514-530).  The
 * order of those atomics is not defined by the reference; this restatement adds
 * the groups in ascending z order.
 */
static uint16_t ap_row_f16(const uint16_t *x, const uint8_t *codes, const uint16_t *lutrow, uint32_t K, int ksplit) {
    uint32_t nfull = K / 1024u;
    uint32_t tail = K % 1024u;
    uint32_t eff = tail / 32u;
    uint32_t nchunks = nfull + (tail ? 1u : 0u);
    uint32_t ngroups = ksplit ? (nchunks + 3u) / 4u : 1u;
    uint16_t total = 0; /* shO cell, +0 */
    uint16_t result = 0;
    for (uint32_t g = 0; g < ngroups; g++) {
        uint32_t c0 = ksplit ? g * 4u : 0u;
        uint32_t c1 = ksplit ? (c0 + 4u < nchunks ? c0 + 4u : nchunks) : nchunks;
        uint16_t part[32];
        for (uint32_t t = 0; t < 32; t++) {
            uint16_t partial = 0;
            for (uint32_t i = c0; i < c1; i++) {
                uint32_t tpw = 32u;
                if (i == nfull) {
                    tpw = eff;
                    if (t >= eff) break;
                }
                uint16_t sx = 0, sy = 0;
                for (int c = 3; c >= 0; c--)
                    for (uint32_t k = 0; k < 4; k++) {
                        uint32_t e0 = 1024u * i + 8u * tpw * (uint32_t)c + 8u * t + 2u * k;
                        sx = h_fma(lutrow[codes[e0]], x[e0], sx);
                        sy = h_fma(lutrow[codes[e0 + 1]], x[e0 + 1], sy);
                    }
                partial = h_add(partial, h_add(sx, sy));
            }
            part[t] = partial;
        }
        for (uint32_t sh = 16; sh >= 1; sh >>= 1)
            for (uint32_t t = 0; t < sh; t++) part[t] = h_add(part[t], part[t + sh]);
        if (ksplit)
            total = h_add(total, part[0]);
        else
            result = part[0];
    }
    return ksplit ? total : result;
}

int gq_oracle_ap_gemv_f16(const uint16_t *x, const uint32_t *qweight, const uint16_t *lut, uint32_t M, uint32_t N,
                          uint32_t K, int bits, uint16_t *y) {
    if (K % 32u || bits < 2 || bits > 8 || M < 1 || M > 8) return -1;
    uint8_t *codes = (uint8_t *)malloc((size_t)N * K);
    if (!codes) return -2;
    gq_oracle_ap_unpack(qweight, N, K, bits, codes);
    uint32_t nc = 1u << bits;
    int ksplit = (M == 1 && K > 4096 && bits >= 7);
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < (int64_t)N; n++)
        for (uint32_t m = 0; m < M; m++)
            y[(size_t)m * N + n] = ap_row_f16(x + (size_t)m * K, codes + (size_t)n * K, lut + (size_t)n * nc, K, ksplit);
    free(codes);
    return 0;
}

/* Native-float GEMV: what a CPU runs when it is NOT asked to emulate binary16 -- the "packed C/OpenMP GEMV" CPU
 * baseline of BASELINE.md section 4.  Same packed format (closed form above), products of the two halves exact in float,
 * float accumulation in 8 interleaved partial sums per row (weight e goes to sum e % 8), one rounding to half.  Walks the
 * planes word by word: no code buffer, so it streams the matrix once like the GPU kernels do.  NOT order-faithful to
 * anyprec.cu; used as the honest CPU timing baseline and as a second, independent fp32-class value check. */
static float h2f_native(uint16_t h) { return (float)h2d(h); }

int gq_oracle_ap_gemv_f32(const uint16_t *x, const uint32_t *qweight, const uint16_t *lut, uint32_t M, uint32_t N,
                          uint32_t K, int bits, uint16_t *y) {
    if (K % 32u || bits < 2 || bits > 8 || M < 1) return -1;
    const uint32_t wpr = K / 32u, nc = 1u << bits;
    const uint32_t nfull = K / 1024u, tail = K % 1024u;
    float *xf = (float *)malloc(sizeof(float) * (size_t)M * K);
    if (!xf) return -2;
    for (size_t i = 0; i < (size_t)M * K; i++) xf[i] = h2f_native(x[i]);
#pragma omp parallel for schedule(static, 16)
    for (int64_t n = 0; n < (int64_t)N; n++) {
        float lv[256];
        for (uint32_t c = 0; c < nc; c++) lv[c] = h2f_native(lut[(size_t)n * nc + c]);
        for (uint32_t m = 0; m < M; m++) {
            const float *xm = xf + (size_t)m * K;
            float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (uint32_t i = 0; i < nfull + (tail ? 1u : 0u); i++) {
                const uint32_t tpw = i < nfull ? 32u : tail / 32u;
                for (uint32_t t = 0; t < tpw; t++) {
                    uint32_t w[8];
                    for (int p = 0; p < bits; p++) w[p] = qweight[((size_t)p * N + (size_t)n) * wpr + 32u * i + t];
                    for (uint32_t c = 0; c < 4; c++) {
                        const float *xp = xm + 1024u * i + 8u * tpw * c + 8u * t;
                        for (uint32_t j = 0; j < 8; j++) {
                            const uint32_t bp = 31u - (8u * c + j);
                            uint32_t code = 0;
                            for (int p = 0; p < bits; p++) code = (code << 1) | ((w[p] >> bp) & 1u);
                            acc[j] += lv[code] * xp[j];
                        }
                    }
                }
            }
            const float s = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
            y[(size_t)m * N + n] = d2h((double)s);
        }
    }
    free(xf);
    return 0;
}

void gq_oracle_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int gq_oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------- */
/* LUT-GEMM (BCQ) GEMV  == nqmv_bias  (inference/ap_gemv/lutgemm.cu:24-149),  */
/* launched by lutgemm_gemv_templated (gemv.cu:140-168).                      */
/* PARITY UNPINNED: the reference holds no test, producer or CPU statement of */
/* this op; this is a restatement of the CUDA kernel's arithmetic.            */
/*                                                                            */
/* One CUDA block per 32-activation tile kt builds 4 tables of 256 halves:    */
/* entry lut[y][v] = sum_{i<8} (bit_i(v) ? +x : -x)[32kt + 8y + i], summed    */
/* left to right in half for v < 64 (bits 6,7 clear), then extended with      */
/* "+ 2*x[6]" and "+ 2*x[7]" (lutgemm.cu:40-78).  Per output m:               */
/*   o  = q_bias[g][m] * (lut[0][255]+lut[1][255]+lut[2][255]+lut[3][255])    */
/*   o += alpha_b * (lut[0][byte0] + lut[1][byte1] + lut[2][byte2] +          */
/*                   lut[3][byte3]) for every plane b, alpha_b = alpha[g][0][m]*/
/*        * 2^b (only plane 0 of alpha is read, doubled per plane :143-144)   */
/* all in half; the tile result is atomically added (half2) to out[m]         */
/* (:147), in an order the reference does not define.  This restatement adds  */
/* the tiles in ascending kt (one valid execution of the atomics).            */
/* ------------------------------------------------------------------------- */
static void lutgemm_build_tile(const uint16_t *x32, uint16_t lut[4][256]) {
    for (int y = 0; y < 4; y++) {
        const uint16_t *xi = x32 + 8 * y;
        for (int v = 0; v < 64; v++) {
            uint16_t acc = 0;
            for (int i = 0; i < 8; i++) {
                uint16_t sgn = ((v >> i) & 1) ? 0x3C00 : 0xBC00; /* +1 / -1 */
                uint16_t term = h_mul(sgn, xi[i]);
                acc = (i == 0) ? term : h_add(acc, term); /* "+ a*x0 + b*x1 ..." left to right; the leading + is unary */
            }
            lut[y][v] = acc;
        }
        for (int s = 6; s < 8; s++) {
            uint16_t iv = h_mul(0x4000 /* 2.0 */, xi[s]);
            for (int v = 1 << s; v < (1 << (s + 1)); v++) lut[y][v] = h_add(lut[y][v - (1 << s)], iv);
        }
    }
}

int gq_oracle_lutgemm_f16(const uint16_t *x, const uint32_t *W, const uint16_t *alpha, const uint16_t *q_bias, uint32_t N,
                          uint32_t K, int bits, int group_size, uint16_t *out /* accumulated into */) {
    if (K % 32u || bits < 1 || bits > 8 || group_size <= 0 || K % (uint32_t)group_size) return -1;
    uint32_t ntiles = K / 32u;
    for (uint32_t kt = 0; kt < ntiles; kt++) {
        uint16_t lut[4][256];
        lutgemm_build_tile(x + 32u * kt, lut);
        uint32_t g = (kt * 32u) / (uint32_t)group_size;
        uint16_t all = 0;
        for (int y = 0; y < 4; y++) all = h_add(all, lut[y][255]);
#pragma omp parallel for schedule(static)
        for (int64_t m = 0; m < (int64_t)N; m++) {
            uint16_t o = h_add(0, h_mul(q_bias[(size_t)g * N + m], all));
            uint16_t a = alpha[(size_t)g * bits * N + m];
            for (int b = 0; b < bits; b++) {
                uint32_t w = W[((size_t)kt * bits + b) * N + m];
                uint16_t t = 0;
                for (int y = 0; y < 4; y++) t = h_add(t, lut[y][(w >> (8 * y)) & 255u]);
                o = h_add(o, h_mul(a, t));
                a = h_mul(a, 0x4000);
            }
            out[m] = h_add(out[m], o);
        }
    }
    return 0;
}

int gq_oracle_lutgemm_f64(const uint16_t *x, const uint32_t *W, const uint16_t *alpha, const uint16_t *q_bias, uint32_t N,
                          uint32_t K, int bits, int group_size, double *out) {
    if (K % 32u || bits < 1 || bits > 8 || group_size <= 0 || K % (uint32_t)group_size) return -1;
    for (uint32_t m = 0; m < N; m++) out[m] = 0.0;
    for (uint32_t kt = 0; kt < K / 32u; kt++) {
        uint32_t g = (kt * 32u) / (uint32_t)group_size;
        double xs = 0;
        for (int i = 0; i < 32; i++) xs += h2d(x[32u * kt + i]);
        for (uint32_t m = 0; m < N; m++) {
            double acc = h2d(q_bias[(size_t)g * N + m]) * xs;
            double a = h2d(alpha[(size_t)g * bits * N + m]);
            for (int b = 0; b < bits; b++) {
                uint32_t w = W[((size_t)kt * bits + b) * N + m];
                double t = 0;
                for (int i = 0; i < 32; i++) t += ((w >> i) & 1u) ? h2d(x[32u * kt + i]) : -h2d(x[32u * kt + i]);
                acc += a * t;
                a *= 2.0;
            }
            out[m] += acc;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------- */
/* QTIP packed trellis: decode + matvec.                                      */
/* Format writer: qtip/lib/codebook/bitshift.py:294-327 (pack_trellis) + the  */
/* kernel swizzle qtip/lib/algo/finetune.py:291-296; python inverse           */
/* inference/lib/utils/kernel_decompress.py:5-55 (decode_compressed) -- the   */
/* golden vectors in tests/golden/qtip_*.npz are its outputs.  Decode rule    */
/* (HYB code, quantlut_sym, bitshift.py:72-80): state = 16-bit window of the  */
/* tile's big-endian bit stream at bit offset 2R*p (circular in the tile),    */
/* idx = state*(state+1) mod 2^32, q = (idx >> 6) & 511, value = tlut[q][e],  */
/* negated iff e == 0 and bit 15 of idx is set.  Pair p of a 16x16 tile is    */
/* the mma m16n8k16 A-fragment slot: a = p/16, b = (p/4)%4, cc = (p/2)%2,     */
/* d = p%2 -> row a + 8d, cols 2b + 8cc + {0,1}.  Tile (tm,tk) lives in the   */
/* 2x2 tile block (tm/2, tk/2) of 128R bytes; stream byte n = s*R + rr of     */
/* tile (a4 = tm%2, a3 = tk%2) is source byte ((2s+a3)*2+a4)*R + (R-1-rr).    */
/* Matvec == kernel_decompress_matvec (qtip/qtip-kernels/src/inference.cu:    */
/* 168-425): fp16 weights x fp16 x, fp32 accumulation, fp32 output.           */
/* ------------------------------------------------------------------------- */
int gq_oracle_qtip_decode(const uint32_t *compressed, const uint16_t *tlut /* [512][2] */, uint32_t M, uint32_t K, int R,
                          uint16_t *W /* [M][K] */) {
    if (M % 32u || K % 32u || R < 2 || R > 4) return -1;
    const uint8_t *src = (const uint8_t *)compressed;
    uint32_t tile_bytes = 32u * (uint32_t)R;
#pragma omp parallel for schedule(static)
    for (int64_t tm = 0; tm < (int64_t)(M / 16u); tm++)
        for (uint32_t tk = 0; tk < K / 16u; tk++) {
            uint8_t stream[32 * 4 + 4];
            size_t base = ((size_t)(tm / 2) * (K / 32u) + tk / 2u) * 128u * (uint32_t)R;
            uint32_t a4 = (uint32_t)tm % 2u, a3 = tk % 2u;
            for (uint32_t s = 0; s < 32; s++)
                for (uint32_t rr = 0; rr < (uint32_t)R; rr++)
                    stream[s * R + rr] = src[base + ((2u * s + a3) * 2u + a4) * R + ((uint32_t)R - 1u - rr)];
            for (uint32_t p = 0; p < 128; p++) {
                uint32_t off = 2u * (uint32_t)R * p; /* bit offset */
                uint32_t st = 0;
                for (uint32_t bb = 0; bb < 16; bb++) {
                    uint32_t bit = (off + bb) % (8u * tile_bytes);
                    st = (st << 1) | ((stream[bit / 8u] >> (7u - bit % 8u)) & 1u);
                }
                uint32_t idx = st * (st + 1u);
                uint32_t q = (idx >> 6) & 0x1FFu;
                uint32_t a = p / 16u, b = (p / 4u) % 4u, cc = (p / 2u) % 2u, d = p % 2u;
                uint32_t r = 16u * (uint32_t)tm + a + 8u * d, c = 16u * tk + 2u * b + 8u * cc;
                uint16_t v0 = tlut[2u * q], v1 = tlut[2u * q + 1u];
                if (idx & 0x8000u) v0 ^= 0x8000u;
                W[(size_t)r * K + c] = v0;
                W[(size_t)r * K + c + 1u] = v1;
            }
        }
    return 0;
}

int gq_oracle_qtip_matvec(const uint32_t *compressed, const uint16_t *tlut, const uint16_t *x, uint32_t M, uint32_t K, int R,
                          double *out) {
    uint16_t *W = (uint16_t *)malloc((size_t)M * K * 2u);
    if (!W) return -2;
    int rc = gq_oracle_qtip_decode(compressed, tlut, M, K, R, W);
    if (rc) {
        free(W);
        return rc;
    }
#pragma omp parallel for schedule(static)
    for (int64_t m = 0; m < (int64_t)M; m++) {
        double acc = 0;
        for (uint32_t k = 0; k < K; k++) acc += h2d(W[(size_t)m * K + k]) * h2d(x[k]);
        out[m] = acc;
    }
    free(W);
    return 0;
}

/* quantlut_sym expansion of the 9-bit table to all 2^16 states (bitshift.py:72-80) */
void gq_oracle_quantlut_sym(const uint16_t *tlut /* [512][2] */, uint16_t *expanded /* [65536][2] */) {
    for (uint32_t st = 0; st < 65536u; st++) {
        uint32_t idx = st * (st + 1u);
        uint32_t q = (idx >> 6) & 0x1FFu;
        uint16_t v0 = tlut[2u * q], v1 = tlut[2u * q + 1u];
        if (idx & 0x8000u) v0 = d2h(-h2d(v0)); /* "* sflp" with sflp = -1 (sign flip, -0 * x) */
        expanded[2u * st] = v0;
        expanded[2u * st + 1u] = v1;
    }
}

/* ------------------------------------------------------------------------- */
/* Hadamard:  y = x @ H_n * scale, Sylvester order (n a power of two) ==      */
/* fast_hadamard_transform.hadamard_transform as used by matmul_hadU_cuda     */
/* (inference/lib/utils/matmul_had.py:96-119); with a K x K factor matrix     */
/* hadK for n = K * 2^j: reshape [K, n/K], FWHT over the last dim, then       */
/* hadK @ (matmul_had.py:109-119).  Pinned against matmul_hadU goldens.       */
/* ------------------------------------------------------------------------- */
int gq_oracle_hadamard(const float *x, float *y, uint32_t rows, uint32_t n, float scale) {
    if (n == 0 || (n & (n - 1u))) return -1;
    for (uint32_t r = 0; r < rows; r++) {
        float *yr = y + (size_t)r * n;
        if (yr != x + (size_t)r * n) memcpy(yr, x + (size_t)r * n, (size_t)n * 4u);
        for (uint32_t h = 1; h < n; h <<= 1)
            for (uint32_t i = 0; i < n; i += 2u * h)
                for (uint32_t j = i; j < i + h; j++) {
                    float a = yr[j], b = yr[j + h];
                    yr[j] = a + b;
                    yr[j + h] = a - b;
                }
        for (uint32_t i = 0; i < n; i++) yr[i] *= scale;
    }
    return 0;
}
