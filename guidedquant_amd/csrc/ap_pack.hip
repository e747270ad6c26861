// ap_pack.hip -- the Any-Precision packer on the device: codes uint8 [N][K] -> bit-planes
// u32 [bits][N][K/32], bit-identical to pack_single_weight / _process_layer_data + _permute_bitmaps_int32
// (any_precision/quantization/pack.py:12-83,101-110,304-321), which runs np.packbits and a byte permutation on the host for
// every layer of a quantization run.  Closed form of the layout (guidedquant_amd/pack.py): weight e of a row lives in word
// base + t, bit 31 - (8 c + j), with (chunk of 1024 weights, or a tail chunk of tpw = (K % 1024) / 32 words)
//     r = e - chunk base,  c = r / (8 tpw),  t = (r % (8 tpw)) / 8,  j = r % 8;   plane p holds code bit (bits - 1 - p).
// One thread per (row, word): 4 x 8 consecutive code bytes in, `bits` words out (every plane of the word at once, so the
// codes are read once); the word index is the fast thread index: stores of a plane are coalesced.  HBM-bound byte work.
#include <hip/hip_runtime.h>

#include "gq_internal.h"

namespace {
typedef uint32_t u32;

__global__ void __launch_bounds__(256) ap_pack_kernel(const uint8_t *__restrict__ codes, u32 *__restrict__ qw, u32 N, u32 K, int bits) {
    const u32 wpr = K / 32u, nfull = K / 1024u, eff = (K % 1024u) / 32u;
    const size_t idx = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (idx >= (size_t)N * wpr) return;
    const u32 n = (u32)(idx / wpr), w = (u32)(idx % wpr);
    const u32 chunk = w / 32u, tpw = chunk < nfull ? 32u : eff, t = w - 32u * chunk;
    const uint8_t *row = codes + (size_t)n * K + 1024u * chunk + 8u * t;
    u32 out[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (u32 c = 0; c < 4; c++) {
        const uint2 v = *reinterpret_cast<const uint2 *>(row + 8u * tpw * c);  // 8 consecutive codes, 8-byte aligned (K % 32 == 0)
#pragma unroll
        for (u32 j = 0; j < 8; j++) {
            const u32 code = ((j < 4 ? v.x : v.y) >> (8u * (j & 3u))) & 0xFFu;
            for (int p = 0; p < bits; p++) out[p] |= ((code >> (bits - 1 - p)) & 1u) << (31u - (8u * c + j));
        }
    }
    for (int p = 0; p < bits; p++) qw[((size_t)p * N + n) * wpr + w] = out[p];
}
}  // namespace

extern "C" int gq_anyprec_pack(const uint8_t *codes, uint32_t *qweight, uint32_t N, uint32_t K, int bits, void *stream) {
    if (bits < 1 || bits > 8) return gq_fail(GQ_EINVAL, "gq_anyprec_pack: bits must be 1..8.");
    if (K == 0 || K % 32u || N == 0) return gq_fail(GQ_EINVAL, "gq_anyprec_pack: need N > 0 and K a positive multiple of 32.");
    if (!codes || !qweight) return gq_fail(GQ_EINVAL, "null pointer argument.");
    if ((uintptr_t)codes & 7u) return gq_fail(GQ_EINVAL, "gq_anyprec_pack: codes must be 8-byte aligned.");
    const size_t words = (size_t)N * (K / 32u);
    hipLaunchKernelGGL(ap_pack_kernel, dim3((unsigned)((words + 255u) / 256u)), dim3(256), 0, (hipStream_t)stream, codes, qweight, N, K, bits);
    GQ_HIP_CHECK(hipGetLastError());
    return GQ_OK;
}
