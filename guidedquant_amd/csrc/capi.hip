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
std::map<std::string, std::pair<bool, int>> g_env_cache;  // name -> (set in the environment, its value)
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
    // What is cached is the ENVIRONMENT's answer, never a caller's default: call sites pass different defaults for one name (the exact
    // kernel's blocks per CU depend on the launch, GQ_AP_BPC) -- round 5 found the first caller's default served to every later one
    // (exact-mode decode 537 instead of 575 tokens/s whenever the step's first exact launch was the small RMSNorm one).
    auto it = g_env_cache.find(name);
    if (it == g_env_cache.end()) {
        const char *v = getenv(name);
        it = g_env_cache.emplace(name, std::make_pair(v && *v, v && *v ? atoi(v) : 0)).first;
    }
    return it->second.first ? it->second.second : dflt;
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

// Self-check of a hardware behaviour three kernels rest on (ap_plane.hip PL_BOOB / PL_LOOB, qtip.hip QT_XOOB): a ds_read from an
// LDS address beyond the workgroup's allocation (192 KiB and up; the chip has 160 KiB) returns zeros -- the idle MFMA columns of
// those kernels take their zeros from there.  Verified on gfx950 (tools/ubench/lds_oob.hip); another architecture, a sanitizer or
// a compiler that lowers the access to flat addressing would corrupt sums silently, so the binding runs this once per process
// and refuses to go on if it fails (rebuild with -DPL_BOOB=0 -DPL_LOOB=0 -DQT_XOOB=0 then).
namespace {
__global__ void lds_oob_probe(unsigned *out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    sm[threadIdx.x] = 0xAB;
    __syncthreads();
    const uint4 v = *reinterpret_cast<const uint4 *>(sm + 0x30000u + 16u * threadIdx.x);
    const uint4 w = *reinterpret_cast<const uint4 *>(sm + 0x38000u + 16u * threadIdx.x);
    out[threadIdx.x] = v.x | v.y | v.z | v.w | w.x | w.y | w.z | w.w | (sm[threadIdx.x] == 0xAB ? 0u : 1u);
}
}  // namespace
extern "C" int gq_selfcheck(void) {
    unsigned *d = nullptr, h[64];
    GQ_HIP_CHECK(hipMalloc(&d, sizeof(h)));
    hipLaunchKernelGGL(lds_oob_probe, dim3(1), dim3(64), 1024, 0, d);
    hipError_t e = hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return gq_fail_hip(e, "gq_selfcheck");
    for (unsigned v : h)
        if (v) return gq_fail(GQ_ENOTSUP, "LDS reads beyond the workgroup's allocation do not return zero on this device: rebuild with "
                                           "-DPL_BOOB=0 -DPL_LOOB=0 -DQT_XOOB=0 (csrc/ap_plane.hip, csrc/qtip.hip).");
    return GQ_OK;
}

// ---- measurement aid (bench.py `frac_of_stream_floor`): a kernel that only READS `bytes` once (16-byte non-temporal loads, 8 per
// thread in flight) -- what a one-shot launch of that size can reach on this chip including its launch boundary; the GEMV launches are
// priced against it next to the 8 TB/s roofline (tools/ubench/stream_read.hip is the stand-alone probe).
namespace {
typedef uint32_t gq_u32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) stream_read_kernel(const gq_u32x4 *p, size_t n16, uint32_t *sink) {
    const size_t i = (size_t)blockIdx.x * 256u * 8u + threadIdx.x;
    gq_u32x4 v[8], acc = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int u = 0; u < 8; u++) {
        const size_t j = i + (size_t)u * 256u;
        v[u] = j < n16 ? __builtin_nontemporal_load(p + j) : (gq_u32x4){0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int u = 0; u < 8; u++) acc ^= v[u];
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1u;
}
}  // namespace
extern "C" int gq_debug_stream_read(const void *buf, size_t bytes, uint32_t *sink, void *stream) {
    if (!buf || !sink || bytes < 16 || ((uintptr_t)buf & 15u)) return gq_fail(GQ_EINVAL, "gq_debug_stream_read: 16-byte aligned buffer of >= 16 bytes.");
    const size_t n16 = bytes / 16u;
    hipLaunchKernelGGL(stream_read_kernel, dim3((unsigned)((n16 + 2047u) / 2048u)), dim3(256), 0, (hipStream_t)stream, (const gq_u32x4 *)buf, n16, sink);
    GQ_HIP_CHECK(hipGetLastError());
    return GQ_OK;
}
