// capi.hip -- error reporting and misc entry points of the C ABI (include/gq_hip.h)
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>

#include "gq_internal.h"

namespace {
thread_local std::string g_last_error;
std::mutex g_env_mu;
std::map<std::string, int> g_env_cache;
}  // namespace

int gq_fail(int code, const char *msg) {
    g_last_error = msg ? msg : "";
    return code;
}

int gq_fail_hip(hipError_t e, const char *where) {
    g_last_error = std::string("HIP error: ") + hipGetErrorString(e) + " (" + where + ")";
    return GQ_EHIP;
}

int gq_env_int(const char *name, int dflt) {
    std::lock_guard<std::mutex> lk(g_env_mu);
    auto it = g_env_cache.find(name);
    if (it != g_env_cache.end()) return it->second;
    const char *v = getenv(name);
    int r = v && *v ? atoi(v) : dflt;
    g_env_cache[name] = r;
    return r;
}

int gq_cu_count() {
    static int cached[256];
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 256) return 256;
    if (cached[dev]) return cached[dev];
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cached[dev] = n;
    return n;
}

extern "C" int gq_version(void) { return 100; }

extern "C" const char *gq_last_error(void) { return g_last_error.c_str(); }

extern "C" int gq_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return gq_fail_hip(e, "hipGetDeviceCount");
    return n;
}

// test hook: drop the cached tuning knobs so a test can flip GQ_* env vars between calls
extern "C" void gq_reset_env_cache(void) {
    std::lock_guard<std::mutex> lk(g_env_mu);
    g_env_cache.clear();
}
