// gq_internal.h -- error plumbing shared by the translation units of libgq_hip.so
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/gq_hip.h"

int gq_fail(int code, const char *msg);            // records msg (thread-local) and returns code
int gq_fail_hip(hipError_t e, const char *where);  // records the HIP error string, returns GQ_EHIP
int gq_env_int(const char *name, int dflt);        // cached getenv -> int (tuning knobs)

#define GQ_STR2(x) #x
#define GQ_STR(x) GQ_STR2(x)
#define GQ_HIP_CHECK(expr)                                                        \
    do {                                                                          \
        hipError_t e__ = (expr);                                                  \
        if (e__ != hipSuccess) return gq_fail_hip(e__, __FILE__ ":" GQ_STR(__LINE__)); \
    } while (0)
