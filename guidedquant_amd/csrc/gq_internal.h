// gq_internal.h -- error plumbing shared by the translation units of libgq_hip.so
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>

#include "../../include/gq_hip.h"

int gq_fail(int code, const char *msg);            // records msg (thread-local) and returns code
int gq_fail_hip(hipError_t e, const char *where);  // records the HIP error string, returns GQ_EHIP
int gq_env_int(const char *name, int dflt);        // cached getenv -> int (tuning knobs)
int gq_cu_count();                                 // compute units of the CURRENT device (cached per device)

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

#if defined(__HIPCC__)
// `(half)(a * b)` on floats: the compiler folds the conversion into v_fma_mixlo_f16, which rounds the EXACT product once.
// The reference's torch ops round the fp32 product first and convert then (two roundings; they differ on fp16 ties of the
// fp32 value).  Pinning the product in a register keeps the two-step rounding.
__device__ __forceinline__ float gq_pin_f32(float v) {
    asm volatile("" : "+v"(v));
    return v;
}
#endif

#define GQ_STR2(x) #x
#define GQ_STR(x) GQ_STR2(x)
#define GQ_HIP_CHECK(expr)                                                        \
    do {                                                                          \
        hipError_t e__ = (expr);                                                  \
        if (e__ != hipSuccess) return gq_fail_hip(e__, __FILE__ ":" GQ_STR(__LINE__)); \
    } while (0)
