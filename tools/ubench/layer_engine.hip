// layer_engine.hip -- would a persistent run-ahead-loader "engine" beat the five launches of a decode layer AT OUR SIZES?
// (MI355X_MICROARCH.md price list rows engine-vs-launches / prefetch-credit / allgather; the guide's
// measured win, 0.87-0.89x, is for a bf16 1B layer of 121.6 MB -- the 2-bit 8B layer streams 54 MB.)
//
// Both sides are SKELETONS with identical memory traffic and no arithmetic: per CU and layer the five phases
//   wqkv   in 4096 halves   weights 24 KiB   out 24 halves per CU
//   attn   in  384 halves (32 CUs only: one head each)     out 128 halves per head-CU
//   wo     in 4096          weights 16 KiB   out 16
//   w1w3   in 4096          weights 112 KiB  out 56 (after the gate/up pairing)
//   w2     in 14336         weights 56 KiB   out 16
// (weights = the 2-bit Llama-3-8B planes / 256 CUs, from a 1.6 GB buffer: never cache-resident), every consumer reads each
// landed LDS byte once (ds_read_b128 + xor) instead of the MFMA work.
//   launches: 5 kernels per layer in one hipGraph, 256 x 256 threads, every wave streams a quarter of the block's share by
//             LDS-DMA, plain loads of the input vector, plain stores of the outputs  (= the shape of today's decode step)
//   engine  : ONE launch for all layers; per CU 1 loader wave (LDS-DMA, nt, ring of 8 x 16 KiB slots, two fills in flight,
//             runs ahead across phase boundaries) + 3 consumer waves; hand-offs as 8-byte {epoch, 2 halves} granules written
//             with relaxed agent-scope (sc1) stores and swept with sc1 loads by one consumer wave per CU until every tag
//             matches (guide recipe R2: the data is the flag), no grid barrier.  Every spin is bounded.
// Output: us per layer for both, plus the engine with the gathers disabled (edges free: pure stream + LDS protocol).
// Build: hipcc --offload-arch=gfx950 -O3 layer_engine.hip -o layer_engine
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef uint32_t u32;
typedef unsigned long long u64;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) u64 gu64;

constexpr int NPH = 5;
// -DCFG=0: Llama-3-8B 2-bit (54 MB per layer)   1: Llama-3-8B 4-bit (109 MB)   2: Llama-3.3-70B 2-bit (214 MB)
#ifndef CFG
#define CFG 0
#endif
#if CFG == 2
__host__ __device__ constexpr u32 kb_of(int ph) { return ph == 0 ? 80u : ph == 2 ? 64u : ph == 3 ? 448u : ph == 4 ? 224u : 0u; }
__host__ __device__ constexpr u32 nout_of(int ph) { return ph == 0 ? 40u : ph == 1 ? 128u : ph == 3 ? 112u : 32u; }
__host__ __device__ constexpr u32 nin_of(int ph) { return ph == 1 ? 384u : ph == 4 ? 28672u : 8192u; }
constexpr u32 NSLOT = 6, NATT = 64;  // (the gathered w2 input is 56 KiB of LDS: 6 ring slots fit beside it)
#else
__host__ __device__ constexpr u32 kb_of(int ph) { return (CFG + 1u) * (ph == 0 ? 24u : ph == 2 ? 16u : ph == 3 ? 112u : ph == 4 ? 56u : 0u); }    // weight KiB per CU
__host__ __device__ constexpr u32 nout_of(int ph) { return ph == 0 ? 24u : ph == 1 ? 128u : ph == 3 ? 56u : 16u; }                // output halves per producing CU
__host__ __device__ constexpr u32 nin_of(int ph) { return ph == 1 ? 384u : ph == 4 ? 14336u : 4096u; }
constexpr u32 NSLOT = 8, NATT = 32;
#endif
constexpr u32 SLOT = 16384, LAYER_KB = kb_of(0) + kb_of(2) + kb_of(3) + kb_of(4);
constexpr u32 XL_WORDS = nin_of(4) / 2u, NGRAN = 16384;
__host__ __device__ constexpr u32 fills_of(int ph) { return (kb_of(ph) + 15u) / 16u; }
__host__ __device__ constexpr u32 kb_before(int ph) { return ph == 0 ? 0u : ph <= 2 ? kb_of(0) : ph == 3 ? kb_of(0) + kb_of(2) : kb_of(0) + kb_of(2) + kb_of(3); }
constexpr u32 SPIN_MAX = 1u << 21;

__device__ __forceinline__ u32 sgpr(u32 v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ void dma16(u32x4 rsrc, u32 lds_base, u32 voff, u32 soff) {
    rsrc = u32x4{sgpr(rsrc.x), sgpr(rsrc.y), sgpr(rsrc.z), sgpr(rsrc.w)};
    lds_base = sgpr(lds_base);
    soff = sgpr(soff);
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen nt lds" ::"s"(lds_base), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
__device__ __forceinline__ u32x4 make_rsrc(const void *p, u32 bytes) {
    const u64 a = (u64)(uintptr_t)p;
    return u32x4{(u32)a, (u32)(a >> 32) & 0xFFFFu, bytes, 0x00020000u};
}
__device__ __forceinline__ u32 lds_ld(volatile u32 *p) { return *p; }

// ---------------------------------------------------------------------------------------------------- launches skeleton
template <int PH>
__global__ void __launch_bounds__(256) phase_kernel(const unsigned char *weights, u64 wbytes, u32 layer, const uint16_t *vin, uint16_t *vout, u32 *sink) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const u32 tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63u;
    if (PH == 1 && blockIdx.x >= NATT) return;
    u32 acc = 0;
    constexpr bool RINGED = kb_of(PH) > 128u;  // share larger than the LDS: every wave streams its quarter through two private 16 KiB slots
    // weight share of this CU: KB KiB at a rotating offset; wave w issues every 4th KiB
    const u64 base = (((u64)layer * LAYER_KB + kb_before(PH)) * 256u + (u64)blockIdx.x * kb_of(PH)) * 1024u % (wbytes - (1u << 20));
    const u32x4 rs = make_rsrc(weights + (base & ~1023ull), 1u << 20);
    if (kb_of(PH) && !RINGED) {
        for (u32 i = w; i < kb_of(PH); i += 4u) dma16(rs, (u32)(uintptr_t)smem + i * 1024u, l * 16u, i * 1024u);
    }
    if (RINGED) {
#pragma unroll
        for (u32 i = 0; i < 16u; i++) dma16(rs, (u32)(uintptr_t)smem + (w * 32u + i) * 1024u, l * 16u, (w * (kb_of(PH) / 4u) + i) * 1024u);
    }
    // input vector: plain 16-byte loads spread over the block (every CU reads all of it, as the real kernels do)
    const u32 npieces = PH == 1 ? 48u : nin_of(PH) / 8u;
    const uint16_t *src = PH == 1 ? vin + (blockIdx.x % NATT) * 128u : vin;
    for (u32 p = tid; p < npieces; p += 256u) {
        const uint4 v = *reinterpret_cast<const uint4 *>(src + 8u * p);
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (RINGED) {
        constexpr u32 NF = kb_of(PH) / 64u;  // 16 KiB fills per wave
        for (u32 f = 0; f < NF; f++) {
            if (f + 1u < NF) {
#pragma unroll
                for (u32 i = 0; i < 16u; i++)
                    dma16(rs, (u32)(uintptr_t)smem + (w * 32u + ((f + 1u) & 1u) * 16u + i) * 1024u, l * 16u, (w * (kb_of(PH) / 4u) + (f + 1u) * 16u + i) * 1024u);
                asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            const unsigned char *slot = smem + (w * 32u + (f & 1u) * 16u) * 1024u;
            for (u32 p = l; p < 1024u; p += 64u) {
                const uint4 v = *reinterpret_cast<const uint4 *>(slot + 16u * p);
                acc ^= v.x ^ v.y ^ v.z ^ v.w;
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);  // the slot is read before it is refilled
        }
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (u32 p = tid; p < kb_of(PH) * 64u; p += 256u) {
            const uint4 v = *reinterpret_cast<const uint4 *>(smem + 16u * p);
            acc ^= v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    if (tid < nout_of(PH)) vout[blockIdx.x * nout_of(PH) + tid] = (uint16_t)(acc | 1u);
    if (acc == 0x12345u) sink[0] = acc;
}

// ---------------------------------------------------------------------------------------------------- engine skeleton
struct Ctl {
    u32 landed, consumed, xready, fail;
};

template <bool GATHER, u32 GP, u32 INFLIGHT>
__global__ void __launch_bounds__(256) engine_kernel(const unsigned char *weights, u64 wbytes, u32 layers, gu64 *gran /* [4][NGRAN] */, u32 *err,
                                                     u32 *sink) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    unsigned char *ring = smem;                                      // NSLOT x SLOT
    u32 *xl = reinterpret_cast<u32 *>(smem + NSLOT * SLOT);          // gathered payloads (<= XL_WORDS words)
    __shared__ Ctl ctl_s;
    volatile Ctl *ctl = &ctl_s;
    const u32 tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63u;
    if (tid == 0) {
        ctl->landed = 0;
        ctl->consumed = 0;
        ctl->xready = 0;
        ctl->fail = 0;
    }
    __syncthreads();
    const u32 fills_per_phase[NPH] = {fills_of(0), 0, fills_of(2), fills_of(3), fills_of(4)};  // shares rounded up to whole 16 KiB slots
    if (w == 0) {
        // ------------------------------------------------ loader: runs ahead of the consumers by up to NSLOT fills
        u32 fill = 0;
        for (u32 layer = 0; layer < layers; layer++)
            for (int ph = 0; ph < NPH; ph++)
                for (u32 f = 0; f < fills_per_phase[ph]; f++, fill++) {
                    if (fill >= NSLOT) {  // slot free when its previous content was consumed by the 3 consumer waves
                        u32 spins = 0;
                        while (lds_ld(&ctl->consumed) < 3u * (fill - NSLOT + 1u)) {
                            __builtin_amdgcn_s_sleep(1);
                            if (++spins > SPIN_MAX || lds_ld(&ctl->fail)) {
                                ctl->fail = 1;
                                if (l == 0) atomicOr(err, 1u);
                                return;
                            }
                        }
                    }
                    const u64 base = ((u64)fill * 256u + blockIdx.x) * SLOT % (wbytes - (1u << 20));
                    const u32x4 rs = make_rsrc(weights + (base & ~1023ull), 1u << 20);
                    const u32 lb = (u32)(uintptr_t)ring + (fill % NSLOT) * SLOT;
#pragma unroll
                    for (u32 i = 0; i < 16; i++) dma16(rs, lb + i * 1024u, l * 16u, i * 1024u);
                    // INFLIGHT fills stay in flight; the one issued INFLIGHT - 1 fills ago has landed
                    if (INFLIGHT == 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
                    if (l == 0 && fill + 2u >= INFLIGHT) ctl->landed = fill + 2u - INFLIGHT;  // fills below that are readable
                }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (l == 0) ctl->landed = fill;
        return;
    }
    // ---------------------------------------------------- consumers (waves 1..3)
    u32 acc = 0, fillc = 0, epoch = 0;
    for (u32 layer = 0; layer < layers; layer++)
        for (int ph = 0; ph < NPH; ph++) {
            ++epoch;
            const bool active = !(ph == 1 && blockIdx.x >= NATT);
            // -------- gather the input vector = the previous phase's output (tag = epoch - 1)
            if (GATHER && epoch > 1u && active) {
                if (w == 1) {
                    const u32 ng = ph == 1 ? 192u : nin_of(ph) / 2u;
                    gu64 *g = gran + (size_t)((epoch - 1u) & 3u) * NGRAN + (ph == 1 ? (blockIdx.x % NATT) * 64u : 0u);
                    u32 spins = 0;
                    for (u32 k0 = 0; k0 < ng; k0 += 64u * GP) {  // GP granules per lane in flight per pass
                        for (;;) {
                            bool ok = true;
                            u32 pay[GP];
#pragma unroll
                            for (u32 k = 0; k < GP; k++) {
                                const u32 gi = k0 + k * 64u + l;
                                u64 v = ((u64)(epoch - 1u) << 32);
                                if (gi < ng) v = __hip_atomic_load(g + gi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                pay[k] = (u32)v;
                                ok &= (u32)(v >> 32) == epoch - 1u;
                            }
                            if (__all(ok)) {
#pragma unroll
                                for (u32 k = 0; k < GP; k++)
                                    if (k0 + k * 64u + l < ng) xl[k0 + k * 64u + l] = pay[k];
                                break;
                            }
                            __builtin_amdgcn_s_sleep(1);
                            if (++spins > SPIN_MAX || lds_ld(&ctl->fail)) {
                                ctl->fail = 1;
                                if (l == 0) atomicOr(err, 2u);
                                return;
                            }
                        }
                    }
                    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the LDS stores above
                    if (l == 0) ctl->xready = epoch;
                } else {
                    u32 spins = 0;
                    while (lds_ld(&ctl->xready) < epoch) {
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > SPIN_MAX || lds_ld(&ctl->fail)) {
                            ctl->fail = 1;
                            if (l == 0) atomicOr(err, 4u);
                            return;
                        }
                    }
                }
                // every consumer reads the whole gathered vector once (the image build of the real kernels)
                const u32 nw = ph == 1 ? 192u : nin_of(ph) / 2u;
                for (u32 p = l + 64u * (w - 1u); p < nw; p += 192u) acc ^= xl[p];
            }
            // -------- consume this phase's fills
            for (u32 f = 0; f < fills_per_phase[ph]; f++, fillc++) {
                u32 spins = 0;
                while (lds_ld(&ctl->landed) <= fillc) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > SPIN_MAX || lds_ld(&ctl->fail)) {
                        ctl->fail = 1;
                        if (l == 0) atomicOr(err, 8u);
                        return;
                    }
                }
                const unsigned char *slot = ring + (fillc % NSLOT) * SLOT;
                for (u32 p = l + 64u * (w - 1u); p < SLOT / 16u; p += 192u) {
                    const uint4 v = *reinterpret_cast<const uint4 *>(slot + 16u * p);
                    acc ^= v.x ^ v.y ^ v.z ^ v.w;
                }
                __builtin_amdgcn_s_waitcnt(0xc07f);
                if (l == 0) atomicAdd((u32 *)&ctl->consumed, 1u);
            }
            // -------- publish this CU's outputs as granules {epoch, 2 halves}: one wave, one 8-byte sc1 store per lane
            if (w == 1 && active) {
                const u32 ngo = nout_of(ph) / 2u;
                if (l < ngo)
                    __hip_atomic_store(gran + (size_t)(epoch & 3u) * NGRAN + blockIdx.x * ngo + l, ((u64)epoch << 32) | (acc | 1u), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    if (acc == 0x12345u) sink[0] = acc;
}

int main(int argc, char **argv) {
    const u32 layers = 32, iters = 5;
    const u64 wbytes = 1600ull << 20;
    unsigned char *weights;
    CHECK(hipMalloc(&weights, wbytes));
    CHECK(hipMemset(weights, 1, wbytes));
    uint16_t *va, *vb;
    CHECK(hipMalloc(&va, 65536));
    CHECK(hipMalloc(&vb, 65536));
    CHECK(hipMemset(va, 0, 65536));
    CHECK(hipMemset(vb, 0, 65536));
    u32 *sink, *err;
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMalloc(&err, 64));
    gu64 *gran;
    { void *p; CHECK(hipMalloc(&p, 4 * NGRAN * 8)); gran = (gu64 *)p; }
    hipStream_t s;
    CHECK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));

    // ---- launches: one graph of layers x 5 kernels
    CHECK(hipFuncSetAttribute((const void *)phase_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CHECK(hipFuncSetAttribute((const void *)phase_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CHECK(hipFuncSetAttribute((const void *)phase_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CHECK(hipFuncSetAttribute((const void *)phase_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    auto lds_of = [](int ph) { return (size_t)(kb_of(ph) > 128u ? 128u : kb_of(ph)) * 1024u; };
    printf("CFG %d: %u KiB of weights per CU and layer = %.1f MB per layer\n", CFG, LAYER_KB, LAYER_KB * 1024.0 * 256 / 1e6);
    hipGraph_t g;
    hipGraphExec_t ge;
    CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (u32 layer = 0; layer < layers; layer++) {
        hipLaunchKernelGGL(phase_kernel<0>, dim3(256), dim3(256), lds_of(0), s, weights, wbytes, layer, va, vb, sink);
        hipLaunchKernelGGL(phase_kernel<1>, dim3(256), dim3(256), 1024, s, weights, wbytes, layer, vb, va, sink);
        hipLaunchKernelGGL(phase_kernel<2>, dim3(256), dim3(256), lds_of(2), s, weights, wbytes, layer, va, vb, sink);
        hipLaunchKernelGGL(phase_kernel<3>, dim3(256), dim3(256), lds_of(3), s, weights, wbytes, layer, vb, va, sink);
        hipLaunchKernelGGL(phase_kernel<4>, dim3(256), dim3(256), lds_of(4), s, weights, wbytes, layer, va, vb, sink);
    }
    CHECK(hipStreamEndCapture(s, &g));
    CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CHECK(hipGraphLaunch(ge, s));
    CHECK(hipStreamSynchronize(s));
    float best = 1e9f;
    for (u32 r = 0; r < iters; r++) {
        CHECK(hipEventRecord(e0, s));
        CHECK(hipGraphLaunch(ge, s));
        CHECK(hipEventRecord(e1, s));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("launches skeleton (5 kernels per layer, hipGraph)      : %7.2f us per layer  (%.0f GB/s of weights)\n", best * 1e3f / layers,
           LAYER_KB * 1024.0 * 256 / (best * 1e3 / layers) / 1e3);

    // ---- engine
    const size_t smem = NSLOT * SLOT + XL_WORDS * 4;
    for (int variant = 0; variant < 5; variant++) {
        const int gather = variant != 4;
        // (24 or 32 granules per lane and pass trip a code-generation error of this hipcc: 16 is the widest sweep built)
        auto kern = variant == 0 ? engine_kernel<true, 8, 2> : variant == 1 ? engine_kernel<true, 16, 2> : variant == 2 ? engine_kernel<true, 8, 3>
                    : variant == 3 ? engine_kernel<true, 16, 3> : engine_kernel<false, 8, 3>;
        const char *vn[5] = {"8 granules/lane/pass, 2 fills in flight", "16 granules/lane/pass, 2 fills in flight", "8 granules/lane/pass, 3 fills in flight",
                             "16 granules/lane/pass, 3 fills in flight", "edges free, 3 fills in flight"};
        CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        best = 1e9f;
        u32 herr = 0;
        for (u32 r = 0; r < iters + 1; r++) {
            CHECK(hipMemsetAsync((void *)gran, 0, 4 * NGRAN * 8, s));  // tags re-initialised every call
            CHECK(hipMemsetAsync(err, 0, 4, s));
            CHECK(hipEventRecord(e0, s));
            hipLaunchKernelGGL(kern, dim3(256), dim3(256), smem, s, weights, wbytes, layers, gran, err, sink);
            CHECK(hipEventRecord(e1, s));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (r && ms < best) best = ms;
            CHECK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
            if (herr) break;
        }
        (void)gather;
        printf("engine skeleton (1 launch, loader + 3 consumers per CU; %s): %7.2f us per layer  (%.0f GB/s of weights)%s\n",
               vn[variant], best * 1e3f / layers, (fills_of(0) + fills_of(2) + fills_of(3) + fills_of(4)) * 16.0 * 1024 * 256 / (best * 1e3 / layers) / 1e3,
               herr ? "   ** spin limit hit: result invalid **" : "");
        if (herr) printf("   err mask %u\n", herr);
    }
    return 0;
}
