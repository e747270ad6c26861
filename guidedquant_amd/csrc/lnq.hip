// lnq.hip -- the coordinate-descent inner loop of LNQ's assignment update on the device.
//
// Reference: update_P, any_precision/quantization/layerwise_quantize.py:93-118 -- for every input column j of a 128-column
// block, IN ORDER:   sol = W[:, j] - B[:, j];  a = argmin_c |sol - C[:, c]|;  W_hat[:, j] = C[:, a];
//                    B[:, j+1 : end] += (W_hat[:, j] - W[:, j]) (x) Hn[j, j+1 : end]
// where Hn = H with column k divided by H[k][k] and B carries the off-diagonal terms of the quadratic form.  The reference
// runs this as ~6 tiny torch kernels per column (d x cd_cycles x 6 launches per matrix: launch-bound for hours of a
// calibration run).  The loop is sequential in j but independent across output rows, so here ONE launch does a whole
// 128-column block: a thread owns a row, its 128 B values live in LDS ([column][thread]: consecutive threads hit
// consecutive banks), the 128 x 128 block of Hn is staged in LDS once per workgroup and read as broadcasts.
// Arithmetic is fp32 with the reference's operation order and no contraction (product rounded, then added): bit-identical to
// the oracle's restatement (oracle.lnq_cd_block_np).  Ties of the argmin go to the lowest centroid index.
// The two GEMM-shaped updates around the block (B = (W_hat - W) @ tril(Hn, -1) and B[:, end:] += ...) stay library GEMMs in
// the host mirror (guidedquant_amd/lnq.py).
#include <hip/hip_runtime.h>

#include "gq_internal.h"

namespace {
typedef uint32_t u32;
constexpr u32 CB = 128;  // columns per block (the reference's cd_block_size)

__global__ void __launch_bounds__(128) lnq_cd_block_kernel(const float *__restrict__ W, const float *__restrict__ B, const float *__restrict__ Hn,
                                                           const float *__restrict__ C, uint8_t *__restrict__ assign, float *__restrict__ What,
                                                           u32 N, u32 d, u32 ncl, u32 group_rows, u32 st, u32 end) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *Hs = reinterpret_cast<float *>(smem);          // [CB][CB] block of Hn (rows st.., columns st..)
    float *Bs = Hs + CB * CB;                              // [CB][TB] running B of this workgroup's rows
    const u32 TB = blockDim.x, tid = threadIdx.x;
    const u32 row0 = blockIdx.x * TB, row = row0 + tid;
    const u32 g = row0 / group_rows;                       // a workgroup never straddles two Hessian groups (host checks)
    const u32 nb = end - st;
    const float *Hg = Hn + (size_t)g * d * d;
    for (u32 i = tid; i < nb * nb; i += TB) {
        const u32 a = i / nb, b = i % nb;
        Hs[a * CB + b] = Hg[(size_t)(st + a) * d + st + b];
    }
    const bool live = row < N;
    const u32 r = live ? row : N - 1u;
    for (u32 k = 0; k < nb; k++) Bs[k * TB + tid] = B[(size_t)r * d + st + k];
    float cen[16];
#pragma unroll
    for (u32 c = 0; c < 16; c++) cen[c] = c < ncl ? C[(size_t)r * ncl + c] : 0.f;
    __syncthreads();
    for (u32 j = 0; j < nb; j++) {
        const float w = W[(size_t)r * d + st + j];
        const float sol = w - Bs[j * TB + tid];
        float best = fabsf(sol - cen[0]);
        u32 arg = 0;
        float val = cen[0];
#pragma unroll
        for (u32 c = 1; c < 16; c++) {
            if (c < ncl) {
                const float dist = fabsf(sol - cen[c]);
                if (dist < best) {
                    best = dist;
                    arg = c;
                    val = cen[c];
                }
            }
        }
        if (live) {
            assign[(size_t)row * d + st + j] = (uint8_t)arg;
            What[(size_t)row * d + st + j] = val;
        }
        const float delta = val - w;
        const float *hj = Hs + j * CB;
        for (u32 k = j + 1u; k < nb; k++) {
            const float p = delta * hj[k];  // rounded product (no contraction: -ffp-contract=off), then the add
            Bs[k * TB + tid] = Bs[k * TB + tid] + p;
        }
    }
}
}  // namespace

extern "C" int gq_lnq_cd_block(const float *W, const float *B, const float *Hn, const float *C, uint8_t *assign, float *What, uint32_t N,
                               uint32_t d, uint32_t n_cluster, uint32_t group_rows, uint32_t col_start, uint32_t col_end, void *stream) {
    if (!W || !B || !Hn || !C || !assign || !What) return gq_fail(GQ_EINVAL, "null pointer argument.");
    if (N == 0 || d == 0 || n_cluster < 2) return gq_fail(GQ_EINVAL, "gq_lnq_cd_block: n_cluster must be >= 2.");
    if (n_cluster > 16) return gq_fail(GQ_ENOTSUP, "gq_lnq_cd_block: more than 16 centroids (> 4 bits) are not served by the kernel.");
    if (col_start >= col_end || col_end > d || col_end - col_start > CB) return gq_fail(GQ_EINVAL, "gq_lnq_cd_block: a block is 1..128 columns.");
    if (group_rows == 0 || N % group_rows) return gq_fail(GQ_EINVAL, "gq_lnq_cd_block: N must be a multiple of the rows per Hessian group.");
    // rows per workgroup: the largest power of two <= 128 dividing the group size (a workgroup reads ONE group's Hessian)
    u32 TB = 128;
    while (TB > 1u && group_rows % TB) TB >>= 1;
    if (TB < 32u) return gq_fail(GQ_ENOTSUP, "gq_lnq_cd_block: rows per Hessian group must be a multiple of 32.");
    const size_t smem = (size_t)(CB * CB + CB * TB) * 4u;
    static GqPerDeviceOnce once;
    GQ_HIP_CHECK(once.max_dynamic_lds(reinterpret_cast<const void *>(lnq_cd_block_kernel), 160 * 1024));
    hipLaunchKernelGGL(lnq_cd_block_kernel, dim3((N + TB - 1u) / TB), dim3(TB), smem, (hipStream_t)stream, W, B, Hn, C, assign, What, N, d, n_cluster,
                       group_rows, col_start, col_end);
    GQ_HIP_CHECK(hipGetLastError());
    return GQ_OK;
}
