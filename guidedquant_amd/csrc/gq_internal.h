// gq_internal.h -- error plumbing shared by the translation units of libgq_hip.so
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <type_traits>

#include "../../include/gq_hip.h"

int gq_fail(int code, const char *msg);            // records msg (thread-local) and returns code
int gq_fail_hip(hipError_t e, const char *where);  // records the HIP error string, returns GQ_EHIP
int gq_env_int(const char *name, int dflt);        // cached getenv -> int (tuning knobs)
int gq_cu_count();                                 // compute units of the CURRENT device (cached per device)

// Statistics hand-over between the launches of a decode step (include/gq_hip.h: gq_anyprec_gemv_fused_ho): what the caller asked for,
// and whether the GEMV kernel that ran wrote the partial sums itself (else the dispatcher adds gq_ssq_rows)
struct GqHandover {
    const float *ssq_in = nullptr;
    float *ssq_out = nullptr;
    bool ssq_written = false;   // the GEMV kernel writes ssq_out in its epilogue
    bool ssq_consumed = false;  // the GEMV kernel's RMSNorm prologue reads ssq_in
    bool dry = false;           // plan only (gq_anyprec_handover_plan): the launchers record the two flags and return without launching
};

// One-time per-DEVICE actions (function attributes such as the > 64 KiB dynamic-LDS opt-in are per device: a process that
// drives several GPUs -- reference-style sequential sharding, device_map -- must repeat them on each one).
struct GqPerDeviceOnce {
    std::atomic<unsigned long long> done[4] = {{0}, {0}, {0}, {0}};
    // raise the kernel's dynamic-LDS limit once per device.  A device is marked only AFTER the call succeeded: a failure (a sticky
    // error, a call during capture) is reported and retried by the next launch instead of leaving the opt-in undone for good.
    // Racing threads may both make the call: it is idempotent.
    hipError_t max_dynamic_lds(const void *func, int bytes) {
        int dev = 0;
        const bool known = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 256;
        const unsigned long long bit = 1ull << (dev & 63);
        if (known && (done[dev >> 6].load(std::memory_order_acquire) & bit)) return hipSuccess;
        const hipError_t e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e == hipSuccess && known) done[dev >> 6].fetch_or(bit, std::memory_order_release);
        return e;
    }
};

// Phase stamps (tools/phase_timing.py, qtip_phase_timing.py, qtip_mid_timing.py: per-wave shader-clock timestamps into a debug buffer)
// are compiled OUT of the shipped library: even switched off at run time every stamp site costs a wave two or three scalar
// instructions and a branch on its critical path (the stream kernel's prologue alone had ~20 sites).  A library with the sites:
//   tools/build_variant.sh stamps ap_stream.hip -DGQ_STAMPS=1   (one translation unit per variant; select with GQ_LIB_PATH)
#ifndef GQ_STAMPS
#define GQ_STAMPS 0
#endif

// waves of an attention block (decode.hip; the head blocks inside the wqkv launch, ap_stream.hip::ap_qkv_attn_kernel, run the same code)
#ifndef GQ_ATTN_WAVES
#define GQ_ATTN_WAVES 8
#endif

#if defined(__HIPCC__)
// `(half)(a * b)` on floats: the compiler folds the conversion into v_fma_mixlo_f16, which rounds the EXACT product once.
// The reference's torch ops round the fp32 product first and convert then (two roundings; they differ on fp16 ties of the
// fp32 value).  Pinning the product in a register keeps the two-step rounding.
__device__ __forceinline__ float gq_pin_f32(float v) {
    asm volatile("" : "+v"(v));
    return v;
}

// the sum of a value over the 64 lanes of a wave, in every lane, without the LDS crossbar: quad_perm / row mirrors (DPP operands of the
// adds) inside a row of 16, v_permlane16_swap / v_permlane32_swap across rows.  (Round 6: six ds_bpermute + wait + add per statistic --
// what __shfl_xor compiles to -- sat between the activations' arrival and the first normalised value of every RMSNorm launch.)  A fixed
// tree: deterministic; the exact / dq GEMV kernels and the QTIP prologues take their RMSNorm statistic through it (kernels that must agree bit for bit do).
__device__ __forceinline__ float gq_wave_allsum(float v) {
    auto dpp = [](float x, auto ctrl) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xF, 0xF, false));
    };
    v += dpp(v, std::integral_constant<int, 0xB1>{});   // quad_perm [1,0,3,2]
    v += dpp(v, std::integral_constant<int, 0x4E>{});   // quad_perm [2,3,0,1]
    v += dpp(v, std::integral_constant<int, 0x141>{});  // row_half_mirror
    v += dpp(v, std::integral_constant<int, 0x140>{});  // row_mirror
    const uint32_t b16 = __builtin_bit_cast(uint32_t, v);
    auto r16 = __builtin_amdgcn_permlane16_swap(b16, b16, false, false);
    v = __builtin_bit_cast(float, (uint32_t)r16[0]) + __builtin_bit_cast(float, (uint32_t)r16[1]);
    const uint32_t b32 = __builtin_bit_cast(uint32_t, v);
    auto r32 = __builtin_amdgcn_permlane32_swap(b32, b32, false, false);
    return __builtin_bit_cast(float, (uint32_t)r32[0]) + __builtin_bit_cast(float, (uint32_t)r32[1]);
}

// Hand-off stores: what a launch of the decode step writes is read by the NEXT launch, mostly from other XCDs.  A plain store
// leaves the line dirty in the writer's XCD L2 until the end-of-kernel write-back; a write-through store (sc1) puts it in
// memory at once.  Used where it measured faster -- the outputs of the plane GEMV kernels (four of the five launches of a layer,
// 8 .. 28 KiB each): 8B decode 828 -> 843 tokens/s, same box, alternating, three rounds.  Measured neutral or slower and left
// plain: the attention output, the 128 K two-byte logits of the lm_head (one fabric write each), the sampler / embedding
// outputs, the fp32 sums of the QTIP engine, the outputs of the exact-mode GEMV kernels (564-566 vs 568-570 tokens/s).  GQ_WT_STORES=0 at build time brings the plain stores back.
#ifndef GQ_WT_STORES
#define GQ_WT_STORES 1
#endif
__device__ __forceinline__ void gq_store_wt(uint16_t *p, uint16_t v) {
#if GQ_WT_STORES
    asm volatile("global_store_short %0, %1, off sc1" ::"v"(p), "v"((uint32_t)v) : "memory");
#else
    *p = v;
#endif
}
__device__ __forceinline__ void gq_store_wt(uint32_t *p, uint32_t v) {
#if GQ_WT_STORES
    asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
#else
    *p = v;
#endif
}
__device__ __forceinline__ void gq_store_wt(int *p, int v) { gq_store_wt(reinterpret_cast<uint32_t *>(p), (uint32_t)v); }
__device__ __forceinline__ void gq_store_wt(float *p, float v) { gq_store_wt(reinterpret_cast<uint32_t *>(p), __builtin_bit_cast(uint32_t, v)); }
__device__ __forceinline__ void gq_store_wt(uint2 *p, uint2 v) {
#if GQ_WT_STORES
    typedef uint32_t gq_u32x2 __attribute__((ext_vector_type(2)));
    const gq_u32x2 t = {v.x, v.y};
    asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(t) : "memory");
#else
    *p = v;
#endif
}
__device__ __forceinline__ void gq_store_wt(uint4 *p, uint4 v) {
#if GQ_WT_STORES
    typedef uint32_t gq_u32x4 __attribute__((ext_vector_type(4)));
    const gq_u32x4 t = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(t) : "memory");
#else
    *p = v;
#endif
}
#endif

#define GQ_STR2(x) #x
#define GQ_STR(x) GQ_STR2(x)
#define GQ_HIP_CHECK(expr)                                                        \
    do {                                                                          \
        hipError_t e__ = (expr);                                                  \
        if (e__ != hipSuccess) return gq_fail_hip(e__, __FILE__ ":" GQ_STR(__LINE__)); \
    } while (0)
