// hop.hip -- device-to-device hand-over of the layer pipeline (the reference hops with a host-side
// `tensor.to(device)`, qtip/lib/utils/shard_model.py:44-68).  A stage tick of guidedquant_amd/pipeline.py is one captured graph;
// with these two launches at its ends the graphs of neighbouring stages synchronise on the device and the host only replays:
//   gq_hop_send  copies a small buffer (the fp16 hidden state, 8-16 KiB, or the sampled token) into the NEXT stage's memory --
//                an IPC-mapped allocation of the peer process: the same GPU (tests), or a peer GPU over xGMI -- with
//                system-scope write-through stores, then publishes a monotonically increasing sequence word behind them;
//   gq_hop_wait  first launch of the consumer's tick: one wave polls the sequence word (system-scope loads) until it reaches the
//                tick number kept in device memory.  The spin is BOUNDED (~2 s): on expiry it raises an error word and returns, the
//                tick computes garbage and the host reports the failure -- a missing peer must never hang the GPU.
// No RCCL call, no host round trip per hop (the isend / irecv pair of the default transport costs two host synchronisations).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gq_internal.h"

namespace {
typedef uint32_t u32;

__global__ void __launch_bounds__(256) hop_send_kernel(const uint4 *src, uint4 *dst, u32 n16, u32 *seq_remote, const u32 *tick, u32 add) {
    for (u32 i = threadIdx.x; i < n16; i += 256u) {
        const uint4 v = src[i];
        // system scope: the peer may be another process on this GPU or another GPU
        __builtin_nontemporal_store(v.x, &dst[i].x);
        __builtin_nontemporal_store(v.y, &dst[i].y);
        __builtin_nontemporal_store(v.z, &dst[i].z);
        __builtin_nontemporal_store(v.w, &dst[i].w);
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_store(seq_remote, tick[0] + add, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

__global__ void __launch_bounds__(64) hop_wait_kernel(const u32 *seq_local, const u32 *tick, u32 add, u32 *err, u32 max_spins) {
    if (threadIdx.x != 0) return;
    const u32 want = tick[0] + add;
    for (u32 i = 0; i < max_spins; i++) {
        if (__hip_atomic_load(seq_local, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) >= want) return;
        __builtin_amdgcn_s_sleep(32);
    }
    __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
}  // namespace

extern "C" int gq_hop_send(const void *src, void *dst_remote, uint32_t nbytes, uint32_t *seq_remote, const uint32_t *tick, uint32_t add,
                           void *stream) {
    if (!src || !dst_remote || !seq_remote || !tick) return gq_fail(GQ_EINVAL, "null pointer argument.");
    if (nbytes == 0 || nbytes % 16u || (((uintptr_t)src | (uintptr_t)dst_remote) & 15u)) return gq_fail(GQ_EINVAL, "gq_hop_send: 16-byte units.");
    hipLaunchKernelGGL(hop_send_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const uint4 *)src, (uint4 *)dst_remote, nbytes / 16u, seq_remote,
                       tick, add);
    GQ_HIP_CHECK(hipGetLastError());
    return GQ_OK;
}

extern "C" int gq_hop_wait(const uint32_t *seq_local, const uint32_t *tick, uint32_t add, uint32_t *err, uint32_t max_spins, void *stream) {
    if (!seq_local || !tick || !err) return gq_fail(GQ_EINVAL, "null pointer argument.");
    hipLaunchKernelGGL(hop_wait_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, seq_local, tick, add, err, max_spins ? max_spins : (1u << 21));
    GQ_HIP_CHECK(hipGetLastError());
    return GQ_OK;
}
