/*
 * gq_hip.h -- C ABI of libgq_hip.so, the MI355X (gfx950) implementation of
 * GuidedQuant's quantized-linear decode path.
 *
 * Plain pointers + sizes, no torch types.  Every device pointer must be
 * resident on the current HIP device; work is enqueued on `stream`
 * (a hipStream_t passed as void*; NULL = the default stream).  No entry point
 * allocates, frees, synchronises or retains a pointer, so all of them are
 * hipGraph-capturable.  Return value: 0 on success, a negative GQ_E* code on
 * error (never exit()/abort(), unlike the reference's HANDLE_ERROR,
 * inference/ap_gemv/gemv.cu:20-27); gq_last_error() returns a thread-local
 * message for the last failure.
 *
 * Each entry point names the reference interface it replaces (file:line in
 * the upstream snu-mllab/GuidedQuant tree).  The reference-side bindings a
 * maintainer would add are shown in INTEGRATION.md.
 */
#ifndef GQ_HIP_H
#define GQ_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GQ_OK 0
#define GQ_EINVAL (-22)   /* bad argument (shape / bitwidth / alignment) */
#define GQ_ENOTSUP (-95)  /* valid request this build has no kernel for */
#define GQ_EHIP (-5)      /* HIP runtime error, see gq_last_error() */

#define GQ_DTYPE_F16 0

/* epilogue / prologue flags of the fused GEMV entry point */
#define GQ_EPI_NONE 0u
#define GQ_EPI_RESIDUAL 1u  /* out[n] = residual[n] + y[n]          (fp16 add) */
#define GQ_PRO_SILU_MUL 2u  /* prologue: x holds 2K values (gate|up), the GEMV input is silu(x[0:K]) * x[K:2K] */
#define GQ_EPI_SILU_PAIRS 4u /* epilogue: the rows are (gate_i, up_i) pairs (row 2i, row 2i+1); out has N/2 elements,
                              * out[i] = silu(y[2i]) * y[2i+1] with the reference's fp16 rounding points */

int gq_version(void);
const char *gq_last_error(void);
/* number of HIP devices visible, <0 on error; lets a host check the library is usable */
int gq_device_count(void);

/*
 * Arithmetic mode of the Any-Precision GEMV (process-wide; also env GQ_AP_EXACT=1):
 *   0  fast  (default): bit-plane GEMVs on the matrix cores, exact products, fp32 accumulation -- closer to the
 *            exact result than the reference kernel, within the north-star tolerance of it (bits 2..4, K % 256 == 0;
 *            other shapes fall back to the exact kernels)
 *   1  exact: reproduces the reference kernel's fp16 accumulation order bit for bit (anyprec.cu:495-512)
 *  -1  reset to the default / environment
 */
int gq_set_ap_mode(int mode);

/*
 * Any-Precision LUT-GEMV.   out[m][n] = sum_k x[m][k] * lut[n][code(n,k)]
 * Replaces ap_gemv.anyprec_gemv -> anyprec_gemv_stream -> anyprec_matmul ->
 * matmul_kbit_32<M,bits,ksplit>  (inference/ap_gemv/bindings.cpp:14,
 * gemv.cu:56-107, anyprec.cu:372-542,591-620).
 *   x       fp16 [M][K]              (M = 1..8; decode uses M = 1)
 *   out     fp16 [M][N]              written directly, no pre-zeroing needed
 *   qweight u32  [>=bits][N][K/32]   bit-plane packed, MSB plane first
 *                                    (any_precision/quantization/pack.py:304-321);
 *                                    plane stride is N*K/32 words
 *   lut     fp16 [N][2^bits]
 * In exact mode results are bit-identical to the reference kernel's fp16 accumulation order; in fast mode they
 * are within 1e-3 (relative, fp16) of it and closer to the exact product.
 * Requires K % 32 == 0, 2 <= bits <= 8.
 */
int gq_anyprec_gemv(const void *x, void *out, const uint32_t *qweight, const void *lut, uint32_t M, uint32_t N,
                    uint32_t K, int bits, int dtype, void *stream);

/*
 * Any-Precision dequantise.   W[n][k] = lut[n][code(n,k)]   (fp16 [N][K])
 * Replaces ap_gemv.anyprec_dequant -> anyprec_dequant_kbit -> dequant_kbit_store<bits>
 * (bindings.cpp:15, gemv.cu:109-134, anyprec.cu:294-359,627-645).
 */
int gq_anyprec_dequant(const uint32_t *qweight, const void *lut, void *W, uint32_t N, uint32_t K, int bits,
                       void *stream);

/*
 * Any-Precision packer on the device: codes u8 [N][K] (values < 2^bits) -> qweight u32 [bits][N][K/32], bit-identical to the
 * host packer of a quantization run (any_precision/quantization/pack.py:12-83,101-110,304-321: np.packbits per plane + the
 * warp byte permutation).  codes 8-byte aligned, K % 32 == 0, bits 1..8 (the parent precision).
 */
int gq_anyprec_pack(const uint8_t *codes, uint32_t *qweight, uint32_t N, uint32_t K, int bits, void *stream);

/*
 * LNQ coordinate descent, one 128-column block (any_precision/quantization/layerwise_quantize.py:93-118, the inner loop of
 * update_P): for j = col_start .. col_end - 1 in order, per output row:  sol = W[j] - B[j];  a = argmin_c |sol - C[c]| (lowest
 * index on ties);  assign[j] = a, What[j] = C[a];  B[k] += (C[a] - W[j]) * Hn[j][k] for j < k < col_end.
 *   W, B, What f32 [N][d] (B is read, its running copy lives in LDS);  Hn f32 [groups][d][d] = H with column k divided by
 *   H[k][k];  C f32 [N][n_cluster], 2 <= n_cluster <= 16;  assign u8 [N][d];  group_rows = N / groups, a multiple of 32.
 * fp32, the reference's operation order, no contraction.
 */
int gq_lnq_cd_block(const float *W, const float *B, const float *Hn, const float *C, uint8_t *assign, float *What, uint32_t N,
                    uint32_t d, uint32_t n_cluster, uint32_t group_rows, uint32_t col_start, uint32_t col_end, void *stream);

/*
 * Any-Precision prefill GEMM with the dequantisation fused into the matrix-core loop.
 *   out[s][n] = sum_k x[s][k] * lut[n][code(n,k)]        x fp16 [S][K], out fp16 [S][N] (written)
 * Replaces the seq_len > 1 branch of APLinear.forward / AnyPrecisionLinear.forward (inference/APLinear.py:35-50,
 * any_precision/modules/AnyPrecisionLinear.py:69-71): ap_gemv.anyprec_dequant (dequant_kbit_store, anyprec.cu:294-359) followed
 * by torch.matmul -- without writing and re-reading the dense fp16 copy of W.  fp32 accumulation, one rounding to fp16.
 * bits 2..4, K % 64 == 0 (GQ_ENOTSUP otherwise: the caller keeps the dequant path); qweight may hold more planes than bits.
 */
int gq_anyprec_gemm(const void *x, void *out, const uint32_t *qweight, const void *lut, uint32_t S, uint32_t N, uint32_t K,
                    int bits, void *stream);
/*
 * The same GEMM with a caller-supplied fp32 workspace for SHORT GRIDS (fewer 128 x 128 output tiles than compute units: a
 * prompt of <= 512 tokens on the 4096-row matrices): K is then split over up to 16 ranges, each block leaves fp32 partial sums
 * in the workspace and a second launch adds the ranges in order and rounds once to fp16 (results equal to gq_anyprec_gemm's up
 * to the fp32 summation order).  gq_anyprec_gemm_ws_bytes = the workspace this problem would use (0: no split planned);
 * a null / too small workspace runs the single pass.  workspace 16-byte aligned.
 */
size_t gq_anyprec_gemm_ws_bytes(uint32_t S, uint32_t N, uint32_t K, int bits);
int gq_anyprec_gemm_ws(const void *x, void *out, const uint32_t *qweight, const void *lut, uint32_t S, uint32_t N, uint32_t K,
                       int bits, void *workspace, size_t ws_bytes, void *stream);

/*
 * The element-wise steps of the prompt pass (seq_len > 1) between the prefill GEMMs, one launch each; fp16 rows, the fp16 rounding
 * points of the tensor expressions they replace (`Transformer.forward`, inference/model.py:206-266):
 *   gq_rmsnorm_rows     RMSNorm.forward (model.py:84-96) on S rows of D: (x.float() * rsqrt(mean(x^2) + eps)).half() * weight;
 *                       delta != NULL: the residual add in front of the norm rides along, x = x + delta (fp16, written back to x)
 *   gq_rope_cache_rows  apply_rotary_pos_emb (model.py:336-341) on the q and k parts of qkv [S][(n_head + 2 n_kv_head) head_dim],
 *                       rotated q -> q_out [n_head][S][head_dim], rotated k and v -> the caches [n_kv_head][max_seq][head_dim] at
 *                       pos[s] (KVCache.update, model.py:69-79; positions >= max_seq are not written)
 *   gq_silu_mul_rows    F.silu(gate) * up (model.py:266) of y [S][2 inter] -> out [S][inter]; paired != 0: (gate_i, up_i) adjacent
 *                       (the decode step's row order of the fused gate/up matrix), else gate = y[:, :inter], up = y[:, inter:]
 * All pointers 16-byte aligned device pointers; D, inter multiples of 8 (D <= 16384), head_dim a multiple of 16.
 */
int gq_rmsnorm_rows(void *x, const void *delta, const void *weight, void *out, uint32_t S, uint32_t D, float eps, void *stream);
int gq_rope_cache_rows(const void *qkv, const int *pos, const void *cos_table, const void *sin_table, void *q_out, void *k_cache, void *v_cache,
                       uint32_t S, uint32_t n_head, uint32_t n_kv_head, uint32_t head_dim, uint32_t max_seq, void *stream);
int gq_silu_mul_rows(const void *y, void *out, uint32_t S, uint32_t inter, int paired, void *stream);

/*
 * Host (CPU) twins of the two Any-Precision entry points: same arguments with HOST pointers, no stream; `nthreads` <= 0
 * uses the OpenMP default.  They serve BASELINE.json configs[0] ("CPU reference APLinear path via generate.py"): the module
 * semantics of inference/APLinear.py:35-60 with the tensors in host memory (the reference hard-codes 'cuda',
 * APLinear.py:17,22,33, and has no CPU kernel).  Read the packed planes directly (pack.py:304-321).
 * gq_anyprec_gemv_cpu: exact fp16 x fp16 products, fp32 accumulation, one rounding to fp16:
 *     |out - exact| <= 2^-11 |exact| + 1e-5 * sum_k |w_k x_k|      (any M >= 1, bits 2..8, K % 32 == 0)
 * gq_anyprec_dequant_cpu: bit-exact table lookup, W fp16 [N][K].
 */
int gq_anyprec_gemv_cpu(const void *x, void *out, const uint32_t *qweight, const void *lut, uint32_t M, uint32_t N,
                        uint32_t K, int bits, int dtype, int nthreads);
int gq_anyprec_dequant_cpu(const uint32_t *qweight, const void *lut, void *W, uint32_t N, uint32_t K, int bits,
                           int nthreads);

/*
 * LUT-GEMM (BCQ) GEMV.  Replaces ap_gemv.lutgemm_gemv -> nqmv_bias
 * (bindings.cpp:16, gemv.cu:140-228, lutgemm.cu:24-149).
 *   x fp16 [K], out fp16 [N] (ACCUMULATED INTO, caller zeroes it as LUTGEMMLinear.py:74 does),
 *   qweight u32 [K/32][bits][N], alpha fp16 [K/group][bits][N], q_bias fp16 [K/group][N].
 */
int gq_lutgemm_gemv(const void *x, void *out, const uint32_t *qweight, const void *alpha, const void *q_bias,
                    uint32_t N, uint32_t K, int bits, int group_size, void *stream);
/* Same result (bit for bit), two launches: the 4 x 256 sign-sum tables of every 32-activation tile depend only on x, so
 * they are built once per call into `workspace` (64 bytes per input feature, 16-byte aligned, caller-owned scratch) and
 * the GEMV blocks copy them to LDS instead of rebuilding them 64..448 times. */
int gq_lutgemm_gemv_ws(const void *x, void *out, const uint32_t *qweight, const void *alpha, const void *q_bias,
                       uint32_t N, uint32_t K, int bits, int group_size, void *workspace, uint64_t workspace_bytes,
                       void *stream);

/*
 * QTIP trellis-decoded matvec.  out[m] = sum_k decode(compressed)[m][k] * x[k]
 * Replaces qtip_kernels.decompress_matvec_16_9_{R}_1_{M}_1_{K} -> kernel_decompress_matvec
 * (qtip/qtip-kernels/src/wrapper.cpp:557-645, qtip_torch.cu:14-57, inference.cu:168-471).
 *   out f32 [M], compressed u32 [R*M*K/32], x fp16 [K], codebook fp16 [1024] (tlut [512][2]).
 * One runtime-shaped entry point serves all 73 compile-time shapes of the reference.
 */
int gq_qtip_matvec(float *out, const uint32_t *compressed, const void *x, const void *codebook, uint32_t M,
                   uint32_t K, int R, void *stream);

/*
 * Hadamard transform on the last dim, y = scale * x @ H_n (Sylvester order), n a power of two.
 * Replaces hadamard::hadamard -> fast_hadamard_transform.hadamard_transform
 * (inference/lib/utils/matmul_had.py:96-106).  x,y f32 [rows][n]; in-place allowed.
 */
int gq_hadamard(const float *x, float *y, uint32_t rows, uint32_t n, float scale, void *stream);

/*
 * Fused QTIP linear for one decode row (bs = 1), in two launches.  Replaces the op sequence of
 * BitshiftLinear.forward (inference/lib/codebook/bitshift.py:415-472, eval mode, rcp == 0) around
 * decompress_matvec: x.float() * SU -> hadamard * n^-1/2 / 32 -> .half() -> matvec [gq_qtip_linear_in], then
 * hadamard * m^-1/2 -> * (SV * 32) -> .half() [gq_qtip_linear_out], with the decode step's own element-wise ops
 * folded in: the RMSNorm in front of q/k/v and gate/up (inference/model.py:281-292), silu(gate) * up in front of the
 * down projection (model.py:266), the residual add behind wo / w2 (model.py:311-313).  Up to 3 linears that share the
 * input (q/k/v; gate/up) go in one launch.  K and every M must be powers of two (the reference's non-power-of-two
 * Hadamard factors are data tables this library does not carry: those models run the unfused ops).
 * Same arithmetic and the same butterfly order as gq_hadamard + gq_qtip_matvec: bit-identical to the unfused chain.
 *   x fp16 [K] (prologue SILU_MUL: x = gate, x2 = up), norm_weight fp16 [K];
 *   GqQtipIn : trellis u32 [R*M*K/32], SU f32 [K], tlut fp16 [1024], y32 f32 [M] (written);
 *   GqQtipOut: y32 f32 [M], SV32 f32 [M] (= SV * 32), resid fp16 [M] or NULL, out fp16 [M] (may alias resid).
 * ksplit = 1..4: K ranges per 32-row band; range s of a band writes its partial sums to y32[s M .. (s + 1) M) (y32 must
 * hold ksplit * M floats; gq_qtip_linear_out with parts = ksplit adds them in ascending order).  A launch runs at most one
 * block per compute unit; every block serves ONE linear and walks that linear's (band, K range) items, so a split evens
 * out launches whose band count is not a multiple of the block count (q/k/v: 3 x 128 bands on 256 units -> ksplit 3).
 * gq_qtip_plan_ksplit(n, M[], K, max) returns the split with the fewest band-equivalents per block for this device.
 * Folding: with n_prev = 1 (2 for SILU_MUL) the input vector(s) of gq_qtip_linear_in are NOT read from x / x2 but rebuilt
 * from prev[] -- the transform-out of the linear(s) that produce them (M == K), residual included -- by every block, and
 * stored to prev[i].out (if not NULL; must not alias prev[i].resid) once: gq_qtip_linear_out and its launch are saved for
 * wo -> gate/up, gate/up -> down and down -> next layer's q/k/v.  Bit-identical to the two-launch form.
 */
#define GQ_QPRO_NONE 0
#define GQ_QPRO_RMSNORM 1
#define GQ_QPRO_SILU_MUL 2
#define GQ_QPRO_PRETRANSFORMED 3 /* x is the fp16 output of gq_qtip_transform(input_side): matvec only, any K % 32 == 0 */
typedef struct GqQtipIn {
    const uint32_t *trellis;
    const float *SU;
    const void *tlut;
    float *y32;
    uint32_t M;
} GqQtipIn;
typedef struct GqQtipOut {
    const float *y32;
    const float *SV32;
    const void *resid;
    void *out;
    uint32_t M;
    uint32_t parts; /* 0 / 1: y32 is [M]; 2..4: y32 is [parts][M] split-K partial sums (ksplit of gq_qtip_linear_in) */
} GqQtipOut;
int gq_qtip_linear_in(const void *x, const void *x2, const void *norm_weight, float eps, int prologue, uint32_t K, int R,
                      int n, const GqQtipIn *lin, int n_prev, const GqQtipOut *prev, int ksplit, void *stream);
int gq_qtip_linear_out(int n, const GqQtipOut *lin, void *stream);
/* The same transform-out spread over M / 128 blocks per linear: block s combines the M / 128 segments of the sums with the signs
 * of row s of the Sylvester matrix and runs one 128-point transform (H_M = H_(M/128) (x) H_128).  Equal to gq_qtip_linear_out up to
 * fp32 rounding (the additions of the full transform in another order), 2.9 instead of 4.6 us per launch in the decode step.
 * M a power of two in 128..8192, sums 16-byte aligned (GQ_ENOTSUP / GQ_EINVAL else). */
int gq_qtip_linear_out_seg(int n, const GqQtipOut *lin, void *stream);
/* Round 5: gq_qtip_linear_out of ONE linear (o or down; residual added, prev->out written once) FOLLOWED IN THE SAME LAUNCH by the
 * transform-in of the n_next <= 3 linears that read its output through an RMSNorm (gate / up; the next layer's q / k / v): block j
 * leaves xt_next[j] = half(H(RMSNorm(out) * SU_next[j]) * K^-1/2 / 32), fp16 [K] -- the vector gq_qtip_linear_in's RMSNorm prologue
 * would build in every one of its blocks (BitshiftLinear.forward, inference/lib/codebook/bitshift.py:441-447; RMSNorm of
 * inference/model.py:281-292).  The matvec launch then runs with GQ_QPRO_PRETRANSFORMED and x == NULL: every GqQtipIn.SU field
 * points to ITS pre-transformed vector.  K = prev->M a power of two in 256..8192. */
int gq_qtip_linear_out_in(const GqQtipOut *prev, const void *norm_weight, float eps, int n_next, const float *const *SU_next,
                          void *const *xt_next, void *stream);
int gq_qtip_plan_ksplit(int n, const uint32_t *M, uint32_t K, int max_ksplit); /* host-side helper, launches nothing */
/*
 * gq_qtip_linear_in + gq_qtip_linear_out in ONE launch: the block that finishes a linear LAST (a device-scope counter per
 * linear; no block waits for another one) also runs that linear's transform-out -- the same code as gq_qtip_linear_out, so the
 * results are bit-identical to the two-launch form.  finish[i] describes the output side of lin[i] (its y32 / parts fields are
 * ignored: the launch's own sums and ksplit are used); every M a power of two <= 16384.  counters: u32 [n] in device memory,
 * zero before the first call (the finishing block resets its counter; one array per stream of launches).
 */
int gq_qtip_linear(const void *x, const void *x2, const void *norm_weight, float eps, int prologue, uint32_t K, int R, int n,
                   const GqQtipIn *lin, const GqQtipOut *finish, int ksplit, void *counters, void *stream);

/*
 * Either side of a QTIP linear whose width n = Kf * P carries a non-power-of-two Hadamard factor (inference/lib/utils/
 * matmul_had.py:13-94; Llama-2's MLP width 11008 = 172 * 64): P-point Sylvester transform of every row of the [Kf][P]
 * view, then hadK @ rows (transpose != 0: hadK^T @, matmul_hadUt).  Such a linear runs transform -> gq_qtip_matvec ->
 * transform (the Kf x Kf product is too much to repeat in every block of the fused kernel).
 *   input_side != 0:  out fp16 [n] = half(H(pro(x) * vec) * n^-1/2 / 32),  vec = SU f32 [n], pro as GQ_QPRO_* (x2 / norm_weight)
 *   input_side == 0:  out fp16 [n] = half(H(y32) * n^-1/2 * vec) (+ resid),  vec = SV * 32 f32 [n]
 * Up to 3 linears per launch (same n, Kf; input side: same x, their own SU / table / out).
 * hadK f32 [Kf][Kf] is the caller's table (the reference's data, not shipped here); P a power of two >= 64.
 */
typedef struct GqQtipXf {
    const float *y32;   /* input_side == 0 */
    const float *vec;
    const float *hadK;
    const void *resid;  /* fp16 [n] or NULL */
    void *out;          /* fp16 [n] */
} GqQtipXf;
int gq_qtip_transform(int input_side, const void *x, const void *x2, const void *norm_weight, float eps, int prologue,
                      int n_lin, const GqQtipXf *lin, uint32_t n, uint32_t Kf, int transpose, void *stream);

/*
 * The middle of a gated MLP whose width n = Kf * 64 carries a Hadamard factor (Llama-2-7b: 11008 = 172 * 64), in ONE launch:
 * transform-out of gate and up (bitshift.py:466-470), silu(gate) * up (model.py:266), * SU of down and the FACTOR side of
 * down's transform-in (bitshift.py:441, matmul_had.py:69-94) -- what two gq_qtip_transform launches (output side for gate / up,
 * then input side with GQ_QPRO_SILU_MUL) do, except that the 64-point row transforms of the input side are left to the
 * consumer: z32 [n] (fp32) goes to gq_qtip_linear_in_rows, which runs them in its prologue.  hadK (x) H_64 acts on the two sides
 * of the [Kf][64] view, so the order of the sides is free: gate / up are computed in gq_qtip_transform's order (bit-identical
 * fp16 values; stored to gate_out / up_out when non-NULL), the input side in the other order (equal to the two-launch chain
 * up to fp32 rounding).
 *   hadT_right16    fp16 [Kf][Kf]: the right-side table of gate / up TRANSPOSED (hadT[k][r] = had_right[r][k]; both projections
 *                   have width n, hence the same table -- matmul_had.py get_hadK(n))
 *   had_left_down16 fp16 [Kf][Kf]: the left-side table of down as stored (the input side multiplies with its transpose)
 * The tables are passed in fp16 (every block reads both of them): their entries must be fp16 values -- a Hadamard factor's +-1 are,
 * and with +-1 entries every product is exact and the results are those of the fp32 tables; the caller checks its table.
 * parts: split-K parts of y32_gate / y32_up ([parts][n], added in ascending order).  Kf <= 176, Kf % 4 == 0.
 */
typedef struct GqQtipMid {
    const float *y32_gate, *y32_up;
    const float *SV32_gate, *SV32_up; /* SV * 32, f32 [n] */
    const void *hadT_right16;
    const float *SU_down;             /* f32 [n] */
    const void *had_left_down16;
    float *z32;                       /* out: f32 [n] */
    void *gate_out, *up_out;          /* fp16 [n] or NULL */
} GqQtipMid;
int gq_qtip_mlp_mid(const GqQtipMid *m, uint32_t parts, uint32_t n, uint32_t Kf, void *stream);
/*
 * gq_qtip_linear_in for an input whose transform-in lacks only the row transforms: x = z32 (f32 [K], K = Kf * P, from
 * gq_qtip_mlp_mid) -> P-point Sylvester transform of every row -> * K^-1/2 / 32 -> fp16 -> trellis matvec into lin[i].y32
 * (lin[i].SU is ignored).  Other arguments as gq_qtip_linear_in.
 */
int gq_qtip_linear_in_rows(const float *z32, uint32_t K, uint32_t P, int R, int n, const GqQtipIn *lin, int ksplit, void *stream);

/*
 * Single-query attention of a QTIP model with the transform-out of its q, k and v linears folded in: qkv_lin[0..2] are the
 * descriptors gq_qtip_linear_out would take (y32 sums, SV32, M, parts; resid / out are ignored), every other argument as in
 * gq_attn_decode_split (declared below).  Each head rebuilds its head_dim outputs of q (its KV group's of k and v): the
 * M / head_dim segments of the sums combined with the signs of the head's row of the Sylvester matrix, then one
 * head_dim-point transform, * M^-1/2 * SV32 -> fp16 (inference/lib/codebook/bitshift.py:470, inference/model.py:206-241).
 * The additions of the full transform in another order: equal to gq_qtip_linear_out + gq_attn_decode_split up to fp32
 * rounding (not bit for bit), one launch per layer less.  M[0] = n_head * head_dim, M[1] = M[2] = n_kv_head * head_dim,
 * powers of two (GQ_ENOTSUP else).
 */
int gq_attn_decode_qtip(const GqQtipOut *qkv_lin, const int *pos, const void *cos_table, const void *sin_table, void *k_cache,
                        void *v_cache, void *out, uint32_t n_head, uint32_t n_kv_head, uint32_t head_dim, uint32_t max_seq,
                        float scale, uint32_t n_split, float *workspace, void *stream);

/*
 * Fused decode-step variant of the AP GEMV, M = 1: optional prologue on x
 * (RMSNorm, or SiLU(gate)*up of a fused gate/up vector) and optional residual-add epilogue, with the same
 * fp16 rounding points as the reference's separate kernels (inference/model.py:151-166,259-266,281-292).
 *   norm_weight  fp16 [K] or NULL (no RMSNorm);  eps used when norm_weight != NULL
 *   flags        GQ_EPI_RESIDUAL: out[n] = residual[n] + y[n]
 *                GQ_PRO_SILU_MUL: x is fp16 [2K]; the GEMV input is silu(x[0:K]) * x[K:2K]
 *                GQ_EPI_SILU_PAIRS: qweight / lut rows are interleaved gate/up pairs (the caller permuted the fused
 *                                 [w1; w3] tensor, model.py:259-266): out fp16 [N/2], out[i] = silu(y[2i]) * y[2i+1];
 *                                 exclusive with GQ_EPI_RESIDUAL; N even
 */
int gq_anyprec_gemv_fused(const void *x, void *out, const uint32_t *qweight, const void *lut, uint32_t N,
                          uint32_t K, int bits, const void *norm_weight, float eps, const void *residual,
                          uint32_t epilogue, void *stream);
/*
 * The same with a caller-supplied workspace (device memory, 16-byte aligned, private to the stream of launches that use it):
 * rows wider than 16384 activations (Llama-3 70B's down projection, K = 28672) are then split along K over BLOCKS -- fp32 sums
 * per K slice in the workspace, a second launch adds the slices in ascending order and rounds ONCE (anyprec.cu:505-512) --
 * instead of over two launches through the fp16 output (two roundings).  gq_anyprec_gemv_fused_ws_bytes: the bytes this shape
 * wants (0: no workspace form; the call then equals gq_anyprec_gemv_fused).  workspace NULL or too small: gq_anyprec_gemv_fused.
 */
size_t gq_anyprec_gemv_fused_ws_bytes(uint32_t N, uint32_t K, int bits, uint32_t epilogue);
int gq_anyprec_gemv_fused_ws(const void *x, void *out, const uint32_t *qweight, const void *lut, uint32_t N,
                             uint32_t K, int bits, const void *norm_weight, float eps, const void *residual,
                             uint32_t epilogue, void *workspace, size_t workspace_bytes, void *stream);

/*
 * Round 5: statistics hand-over between the launches of a decode step.  The hidden state an RMSNorm prologue normalises was written
 * by the previous launch (the residual epilogue of wo / w2, or the embedding lookup): that launch leaves the partial sums of squares
 * of the fp16 values it stored next to them, and the consumer adds GQ_SSQ_SLOTS floats instead of exchanging per-wave sums through
 * LDS behind a counter between the activations' arrival and the normalisation (reference rounding points unchanged:
 * inference/model.py:281-292 -- fp32 mean of squares, rsqrt, fp16 rounding, fp16 multiply; only the fp32 summation ORDER differs,
 * as it already did from torch's).
 *   ssq_out  float [GQ_SSQ_SLOTS], 16-byte aligned, or NULL: the launch writes ALL slots (partial sums of squares of its fp16 outputs;
 *            their total is the vector's sum of squares).  Plain / residual epilogue only.  Kernels without the in-epilogue form are
 *            followed by one small launch (gq_ssq_rows) -- the result is there either way.
 *   ssq_in   the slots written by the launch that produced x (the caller's contract), or NULL.  Used by the RMSNorm prologues that
 *            have the hand-over form (the stream kernel, csrc/ap_stream.hip); ignored elsewhere.  GQ_SSQ_HANDOVER=0 ignores it.
 * gq_anyprec_gemv_fused_ho = gq_anyprec_gemv_fused_ws + the two pointers; gq_anyprec_gemv_qkv_rope_ho, gq_embed_lookup_ho likewise
 * (declared below); gq_ssq_rows: the statistics of any fp16 vector.
 */
#define GQ_SSQ_SLOTS 1024
int gq_anyprec_gemv_fused_ho(const void *x, void *out, const uint32_t *qweight, const void *lut, uint32_t N,
                             uint32_t K, int bits, const void *norm_weight, float eps, const void *residual,
                             uint32_t epilogue, void *workspace, size_t workspace_bytes, const float *ssq_in, float *ssq_out,
                             void *stream);
int gq_ssq_rows(const void *x, uint32_t n, float *ssq_out, void *stream);
/* Plan: which form the launch gq_anyprec_gemv_fused_ho(N, K, bits, norm_weight != NULL, epilogue) would take under the current mode /
 * environment -- bit 0: its RMSNorm prologue reads ssq_in; bit 1: its epilogue writes ssq_out itself (no extra launch).  A decode step
 * passes the pointers only where both ends of an edge are free (guidedquant_amd/model.py::native_layers). */
int gq_anyprec_handover_plan(uint32_t N, uint32_t K, int bits, int has_norm, uint32_t epilogue);

/*
 * The non-quantized pieces of one bs=1 decode step (inference/model.py:121-130,151-166,206-241).  Token id and
 * position are read from DEVICE memory so a captured hipGraph can be replayed per token.
 */
/* x = tok_embeddings[token]                    table fp16 [vocab][dim], out fp16 [dim] */
int gq_embed_lookup(const int *token, const void *table, void *out, uint32_t dim, uint32_t vocab, void *stream);
int gq_embed_lookup_ho(const int *token, const void *table, void *out, uint32_t dim, uint32_t vocab, float *ssq_out, void *stream);

/* RoPE(q,k) at *pos, KV-cache update at *pos, softmax(q k^T / sqrt(d)) v over positions 0..*pos.
 * *pos >= max_seq (decoding past the cache): nothing is written to the caches and `out` is filled with NaN, so the step's
 * logits are NaN -- a loud failure instead of silently overwriting the last slot.
 *   qkv fp16 [(n_head + 2 n_kv_head) * head_dim] (fused wqkv output: q | k | v, model.py:211)
 *   cos/sin fp16 [max_seq][head_dim] (LlamaRotaryEmbedding tables, model.py:379-405)
 *   k_cache, v_cache fp16 [n_kv_head][max_seq][head_dim] (KVCache, model.py:63-79);  out fp16 [n_head * head_dim] */
int gq_attn_decode(const void *qkv, const int *pos, const void *cos_table, const void *sin_table, void *k_cache,
                   void *v_cache, void *out, uint32_t n_head, uint32_t n_kv_head, uint32_t head_dim, uint32_t max_seq,
                   float scale, void *stream);
/* Split-KV form for long contexts: n_split blocks per head, each over a contiguous range of the cached positions (whole
 * passes of 128 positions at head_dim 128), partial (max, sum, weighted V) results in `workspace` (f32
 * [n_head][n_split][head_dim + 2]) combined by a second small launch.  One block per head keeps only n_head CUs
 * streaming the cache: 51 us per layer at 4096 positions.  n_split = 1 is gq_attn_decode (no workspace, one launch). */
int gq_attn_decode_split(const void *qkv, const int *pos, const void *cos_table, const void *sin_table, void *k_cache,
                         void *v_cache, void *out, uint32_t n_head, uint32_t n_kv_head, uint32_t head_dim, uint32_t max_seq,
                         float scale, uint32_t n_split, float *workspace, void *stream);

/*
 * Device-to-device hand-over of the layer pipeline (round 4; reference precedent: the host-side `.to(device)` hops of
 * qtip/lib/utils/shard_model.py:44-68).  gq_hop_send: copy `nbytes` (multiple of 16) from src into dst_remote -- memory of the
 * next stage, mapped into this process (hipIpc / torch's CUDA IPC; same GPU or a peer over xGMI) -- with system-scope stores, then
 * store *seq_remote = *tick + add behind them.  gq_hop_wait: spin (one wave, bounded: max_spins polls of ~1 us, 0 = 2^21) until
 * *seq_local >= *tick + add; on expiry *err = 1 and the launch returns (the GPU never hangs on a missing peer).  `tick` is a word
 * in device memory the caller's graph increments per tick.
 */
int gq_hop_send(const void *src, void *dst_remote, uint32_t nbytes, uint32_t *seq_remote, const uint32_t *tick, uint32_t add, void *stream);
int gq_hop_wait(const uint32_t *seq_local, const uint32_t *tick, uint32_t add, uint32_t *err, uint32_t max_spins, void *stream);
/* Round 5: what a PEER writes -- landing slots and sequence words -- lives in FINE-GRAINED device memory (ordinary
 * allocations are coarse-grained: a peer GPU's writes are not guaranteed visible to kernels running here).  gq_hop_alloc: zeroed
 * fine-grained memory of the current device; gq_hop_export / gq_hop_import: a 64-byte hipIpc handle of it / the mapping in another
 * process; gq_hop_close, gq_hop_free.  gq_hop_wait_copy = gq_hop_wait, then the payload is copied out of the landing slot with
 * system-scope loads into `dst`, the ordinary buffer the stage's kernels read: no kernel but these two touches peer-written memory. */
int gq_hop_alloc(size_t bytes, void **ptr);
int gq_hop_free(void *ptr);
int gq_hop_export(void *ptr, void *handle64);
int gq_hop_import(const void *handle64, void **ptr);
int gq_hop_close(void *ptr);
/* 1 when `ptr` lies in memory THIS process allocated fine-grained (hipPointerGetAttributes: allocationFlags has
 * hipDeviceMallocFinegrained), 0 when not, < 0 on error -- what pipeline.py / tp.py assert of their landing buffers before a
 * rank on another device is given the handle. */
int gq_hop_is_finegrained(const void *ptr);
int gq_hop_wait_copy(const uint32_t *seq_local, const uint32_t *tick, uint32_t add, uint32_t *err, uint32_t max_spins, const void *landed,
                     void *dst, uint32_t nbytes, void *stream);

/* One-time device self-check of a hardware behaviour the plane / QTIP kernels rest on (an LDS read beyond the workgroup's
 * allocation returns zeros: the idle MFMA columns take their zeros from there).  GQ_OK, or GQ_ENOTSUP with the rebuild flags in
 * gq_last_error().  The Python binding calls it once per process on a GPU box. */
int gq_selfcheck(void);
/* Measurement aid: one launch that only reads `bytes` of `buf` once (the one-shot streaming floor bench.py prices the GEMV launches
 * against: `frac_of_stream_floor`).  sink: one device word (never written in practice). */
int gq_debug_stream_read(const void *buf, size_t bytes, uint32_t *sink, void *stream);

/*
 * Round 4: RoPE and the KV-cache write in the EPILOGUE of the fused q / k / v projection, attention without them.
 *
 * gq_anyprec_gemv_qkv_rope = RMSNorm -> Any-Precision GEMV of the fused wqkv tensor (reference row order q | k | v, model.py:211;
 * nothing is permuted in memory: the kernel picks its 16-row groups so that a lane owns both rotation partners d, d + head_dim / 2)
 * -> apply_rotary_pos_emb on the fp16 outputs (model.py:330-341: three fp16-rounded operations, fp16 cos / sin tables) ->
 *   q_out fp16 [n_head * head_dim] rotated queries;  k_cache[g][*pos][:] rotated keys;  v_cache[g][*pos][:] values
 * (KVCache.update, model.py:69-79).  *pos >= max_seq writes nothing to the caches.  Replaces gq_anyprec_gemv_fused(wqkv) + the
 * rotation / cache write inside gq_attn_decode_split; results are bit-identical to that chain (same GEMV arithmetic, same fp16
 * rounding points).  gq_anyprec_qkv_rope_supported: 1 when this build serves (N, K, bits, head_dim) -- the caller keeps the
 * two-launch form otherwise (GQ_ENOTSUP).
 * gq_attn_decode_roped: single-query attention over positions 0..*pos of the caches for queries that are already rotated; the first
 * batch of cached rows is requested together with q and the position (one memory round trip); arguments as gq_attn_decode_split. */
int gq_anyprec_qkv_rope_supported(uint32_t N, uint32_t K, int bits, uint32_t head_dim);
int gq_anyprec_gemv_qkv_rope(const void *x, void *q_out, const uint32_t *qweight, const void *lut, uint32_t N, uint32_t K, int bits,
                             const void *norm_weight, float eps, const int *pos, const void *cos_table, const void *sin_table,
                             void *k_cache, void *v_cache, uint32_t n_head, uint32_t n_kv_head, uint32_t head_dim, uint32_t max_seq,
                             void *stream);
int gq_anyprec_gemv_qkv_rope_ho(const void *x, void *q_out, const uint32_t *qweight, const void *lut, uint32_t N, uint32_t K, int bits,
                                const void *norm_weight, float eps, const int *pos, const void *cos_table, const void *sin_table,
                                void *k_cache, void *v_cache, uint32_t n_head, uint32_t n_kv_head, uint32_t head_dim, uint32_t max_seq,
                                const float *ssq_in, void *stream);
int gq_attn_decode_roped(const void *q, const int *pos, const void *k_cache, const void *v_cache, void *out, uint32_t n_head,
                         uint32_t n_kv_head, uint32_t head_dim, uint32_t max_seq, float scale, uint32_t n_split, float *workspace,
                         void *stream);

/*
 * Round 6: gq_anyprec_gemv_qkv_rope AND gq_attn_decode_roped (n_split = 1) as ONE launch -- the attention heads are extra blocks of the
 * wqkv launch, on compute units the GEMV leaves idle; they fetch the position and the cached rows below it while the GEMV runs and
 * wait on device flags for the rotated q and row *pos of the caches (csrc/ap_stream.hip::ap_qkv_attn_kernel: agent-scope stores /
 * atomics / loads, a bounded poll -- on expiry the head's output is NaN, nothing hangs).  Replaces the two launches
 * (inference/model.py:206-241 between `self.wqkv(x)` and `self.wo(y)`); q_out, the caches and attn_out are bit-identical to theirs.
 *   attn_out  fp16 [n_head * head_dim]: softmax(q k^T * scale) v over positions 0..*pos (*pos >= max_seq: NaN, like the two launches)
 *   flags     uint32 [n_head * GQ_ATTN_FLAG_STRIDE], 128-byte aligned, ZERO before the first launch; every launch leaves them zero
 *             (one 128-byte line per head: agent-scope atomics on one line serialise at the memory side).  One buffer serves every
 *             layer of a model (the launches of a stream are ordered).
 * gq_anyprec_qkv_rope_attn_supported: 1 when this build serves the geometry that way (the fused wqkv launch is served, its grid +
 * n_head blocks fit the device's compute units one each, AND the environment asks for it: GQ_QKV_ATTN=1 -- the form is OFF by default,
 * measured no faster than the two launches, profiles/r06_attention_in_wqkv_launch.txt); the caller keeps the two launches otherwise.
 */
#define GQ_ATTN_FLAG_STRIDE 32
int gq_anyprec_qkv_rope_attn_supported(uint32_t N, uint32_t K, int bits, uint32_t head_dim, uint32_t n_head, uint32_t n_kv_head);
int gq_anyprec_gemv_qkv_rope_attn(const void *x, void *q_out, const uint32_t *qweight, const void *lut, uint32_t N, uint32_t K, int bits,
                                  const void *norm_weight, float eps, const int *pos, const void *cos_table, const void *sin_table,
                                  void *k_cache, void *v_cache, uint32_t n_head, uint32_t n_kv_head, uint32_t head_dim, uint32_t max_seq,
                                  void *attn_out, float scale, uint32_t *flags, void *stream);

/* out[n] = sum_k rmsnorm?(x)[k] * W[n][k]      dense fp16 GEMV (lm_head `output`, model.py:94,128-129); fp32
 * accumulation, fp16 output; norm_weight == NULL skips the RMSNorm prologue.  K % 512 == 0. */
int gq_dense_gemv_f16(const void *x, const void *W, void *out, uint32_t N, uint32_t K, const void *norm_weight, float eps,
                      void *stream);

/* Top-k sampling with the reference's distribution (inference/generate.py:53-73: logits / max(T,1e-5), top-k, softmax,
 * exponential-race draw).  work_val / work_idx: 128*32 floats / ints of scratch; counter: one int of RNG state in
 * device memory (incremented per call).  If tok_io / pos_io are non-NULL the sampled token is written to *tok_io and
 * *pos_io is incremented, so a captured decode graph advances by itself.  top_k <= 32. */
int gq_sample_topk(const void *logits, uint32_t vocab, int top_k, float temperature, uint32_t seed, int *counter,
                   float *work_val, int *work_idx, int *tok_io, int *pos_io, int *next_tok, void *stream);
/* Round 5: the same draw with up to 64 candidates (work_val / work_idx: 128 * 64 elements) and three optional extras for a decode
 * step that runs without the host (guidedquant_amd/generate.py::DecodeGraph, the HF-surface route of AnyPrecisionForCausalLM.generate):
 *   ban      device words {n <= 4, until_pos, id[0..3]} or NULL: while *pos_io < until_pos the listed tokens cannot be drawn -- what HF's
 *            MinNewTokensLengthLogitsProcessor does to EOS (transformers generation, used by inference_example.py:57-67 min_new_tokens);
 *   seq_out  int [seq_cap] or NULL: seq_out[*pos_io + 1] = token (positions as stored BEFORE the increment);
 *   embed_table / x_out / dim / ssq_out: x_out = embed_table[token] (fp16 [dim]) and, with ssq_out, its hand-over statistics
 *            (gq_embed_lookup_ho) -- the next step needs no embedding launch (inference/model.py:121-130). */
int gq_sample_topk_ex(const void *logits, uint32_t vocab, int top_k, float temperature, uint32_t seed, int *counter,
                      float *work_val, int *work_idx, int *tok_io, int *pos_io, int *next_tok, const int *ban, int *seq_out,
                      uint32_t seq_cap, const void *embed_table, void *x_out, uint32_t dim, float *ssq_out, void *stream);
/* Round 6: + nucleus (top-p) filtering of the top-k survivors, as transformers chains its warpers (temperature -> TopKLogitsWarper ->
 * TopPLogitsWarper, generation/logits_process.py): probabilities of the scaled top-k scores, cumulative sum in ascending order, a token
 * is removed when the sum up to and including it is <= 1 - top_p; the most probable token always stays.  top_p in (0, 1]; 1 = exactly
 * gq_sample_topk_ex.  What the HF-surface route needs for `generate(do_sample=True, top_p=0.9)` with the default top_k = 50. */
int gq_sample_topk_p(const void *logits, uint32_t vocab, int top_k, float top_p, float temperature, uint32_t seed, int *counter,
                     float *work_val, int *work_idx, int *tok_io, int *pos_io, int *next_tok, const int *ban, int *seq_out,
                     uint32_t seq_cap, const void *embed_table, void *x_out, uint32_t dim, float *ssq_out, void *stream);

/* Test / tuning hooks (not part of the reference's surface).  gq_reset_env_cache: drop the cached GQ_* environment
 * knobs so that a test can flip them between calls.  gq_debug_set_timing_buffer: device buffer the plane kernels write
 * s_memtime phase stamps into (tools/phase_timing.py); NULL (default) disables it. */
void gq_reset_env_cache(void);
void gq_debug_set_timing_buffer(void *device_buffer);
void gq_debug_set_qtip_timing_buffer(void *device_buffer); /* u64 [16 waves][8] phase stamps of gq_qtip_matvec's middle block */
/* The launch plan of the exact-order AP-GEMV kernel (ap_gemv.hip::pick_quad_cfg) for a shape, without launching: plan[0..5] = threads per
 * block, row slots per step, row steps per block, ring depth, blocks of the launch, and the blocks per CU the instance's occupancy holds at
 * once.  prologue: 0 none, 1 RMSNorm, 2 SiLU * up.  Returns GQ_ENOTSUP when the fast exact path does not serve the shape.  For the test that
 * no plan asks for more resident blocks than fit (a round-6 find: such a launch runs in two rounds). */
int gq_debug_exact_plan(uint32_t N, uint32_t K, int bits, int prologue, uint32_t *plan);

#ifdef __cplusplus
}
#endif
#endif /* GQ_HIP_H */
