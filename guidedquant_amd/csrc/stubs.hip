// stubs.hip -- entry points declared in include/gq_hip.h whose kernels are not built yet.
#include "gq_internal.h"

extern "C" int gq_qtip_matvec(float *, const uint32_t *, const void *, const void *, uint32_t, uint32_t, int, void *) {
    return gq_fail(GQ_ENOTSUP, "gq_qtip_matvec: kernel not built yet");
}
extern "C" int gq_hadamard(const float *, float *, uint32_t, uint32_t, float, void *) {
    return gq_fail(GQ_ENOTSUP, "gq_hadamard: kernel not built yet");
}
