// gq_internal.h -- error plumbing shared by the translation units of libgq_hip.so
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/gq_hip.h"

int gq_fail(int code, const char *msg);            // records msg (thread-local) and returns code
int gq_fail_hip(hipError_t e, const char *where);  // records the HIP error string, returns GQ_EHIP
int gq_env_int(const char *name, int dflt);        // cached getenv -> int (tuning knobs)
int gq_cu_count();                                 // compute units of the CURRENT device (cached per device)

// One-time per-DEVICE actions (function attributes such as the > 64 KiB dynamic-LDS opt-in are per device: a process that
// drives several GPUs -- reference-style sequential sharding, device_map -- must repeat them on each one).
struct GqPerDeviceOnce {
    unsigned long long done[4] = {0, 0, 0, 0};
    // true exactly once per device (racing threads may both see true: the guarded action is idempotent)
    bool first_use() {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 256) return true;
        const unsigned long long bit = 1ull << (dev & 63);
        if (done[dev >> 6] & bit) return false;
        done[dev >> 6] |= bit;
        return true;
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
