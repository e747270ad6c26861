// flag_handoff.hip -- round 6 skeleton: a consumer block that waits INSIDE the producers' launch (the attention heads behind the wqkv
// GEMV of a decode step) instead of behind a kernel boundary.  What does the hand-over cost, and is it coherent across XCDs?
//
// One launch = 192 producer blocks (the wqkv GEMV's grid: block b produces 1/4 of head b / 4; heads 0..31 = q, 32..39 = k of group
// g, 40..47 = v of group g) + 32 consumer blocks (one per query head h, group g = h / 4).  A producer idles ~4 us (its GEMV), writes
// its 64 bytes with agent-scope write-through stores (global_store .. sc1: what gq_store_wt emits), waits for the stores' acknowledgement
// (s_waitcnt vmcnt(0); VARIANT 1: + buffer_wbl2 sc1 as the compiler's release sequence has it), stamps the time and adds 1 to the
// flag of every head that needs its piece (global_atomic_add .. sc1).  A consumer first PLANTS stale copies of what it will read in
// its own L1 / L2 (plain loads of the previous iteration's values), polls its flag with agent-scope loads (sc1) until it reads 12,
// then reads its 3 x 256 bytes -- with sc1 loads (the design) or plain loads (VARIANT 2: expected to see the planted stale lines) --
// checks every word against the iteration number and stamps the time.  Times: s_memrealtime (100 MHz, one clock for the chip).
// Kill criterion of the design: flag seen + data in hand later than ~2.2 us behind the last producer's stores = no gain over a kernel
// boundary (1.6 us launch gap + ~1.1 us for the first loads of the next kernel).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef unsigned int u32;
typedef unsigned long long u64;

__device__ __forceinline__ void st_sc1(u32 *p, u32 v) { asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ u32 ld_sc1(const u32 *p) {
    u32 v;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ u32 ld_plain(const u32 *p) {
    u32 v;
    asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void atomic_add_sc1(u32 *p, u32 v) { asm volatile("global_atomic_add %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }

struct Args {
    u32 *q, *k, *v;      // [32][64], [8][64], [8][64] dwords
    u32 *flags;          // [32][16]: word 0 = the atomic counter (variants 0..2); words 0..11 = one word per producer piece (variant 4)
    u64 *t_store;        // [192] per producer: time its stores were acknowledged
    u64 *t_seen, *t_data;  // [32] per consumer
    u32 *bad;            // [32] mismatching words seen by consumer h (this launch)
    u32 *spun;           // [32] polls of consumer h
    u32 iter, variant, work_ticks;
};

__global__ void __launch_bounds__(1024) handoff_kernel(Args a) {
    const u32 b = blockIdx.x, tid = threadIdx.x;
    if (b < 192u) {  // ---- producer
        const u64 t0 = __builtin_amdgcn_s_memrealtime();
        if (tid == 0)
            while (__builtin_amdgcn_s_memrealtime() - t0 < a.work_ticks) __builtin_amdgcn_s_sleep(8);
        __syncthreads();
        const u32 hh = b >> 2, part = b & 3u;
        u32 *dst = hh < 32u ? a.q + hh * 64u : (hh < 40u ? a.k + (hh - 32u) * 64u : a.v + (hh - 40u) * 64u);
        if (tid < 16u) st_sc1(dst + 16u * part + tid, a.iter);
        if (a.variant & 1u) asm volatile("buffer_wbl2 sc1" ::: "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            a.t_store[b] = __builtin_amdgcn_s_memrealtime();
            if (a.variant & 4u) {  // one word per (head, producer piece): plain agent-scope stores, nothing read-modify-write
                if (hh < 32u) st_sc1(a.flags + 16u * hh + part, 1u);
                else {
                    const u32 g = (hh - 32u) & 7u, slot = (hh < 40u ? 4u : 8u) + part;
                    for (u32 i = 0; i < 4u; i++) st_sc1(a.flags + 16u * (4u * g + i) + slot, 1u);
                }
            } else if (hh < 32u) atomic_add_sc1(a.flags + 16u * hh, 1u);
            else {
                const u32 g = (hh - 32u) & 7u;
                for (u32 i = 0; i < 4u; i++) atomic_add_sc1(a.flags + 16u * (4u * g + i), 1u);
            }
        }
        return;
    }
    // ---- consumer of head h
    const u32 h = b - 192u, g = h >> 2;
    if (tid >= 512u) return;
    const u32 *src = tid < 64u ? a.q + h * 64u + tid : (tid < 128u ? a.k + g * 64u + (tid - 64u) : a.v + g * 64u + (tid - 128u));
    u32 stale = 0;
    if (tid < 192u) stale = ld_plain(src);  // plant the previous iteration's lines in this CU's L1 and this XCD's L2
    __shared__ u32 ok_s;
    u32 polls = 0;
    if (tid < 64u) {  // (wave 0 polls: lane i < 12 looks at word i)
        bool ok = false;
        for (; polls < (1u << 20); polls++) {
            if (a.variant & 4u) {
                const u32 f = tid < 12u ? ld_sc1(a.flags + 16u * h + tid) : 1u;
                ok = __builtin_amdgcn_ballot_w64(f != 1u) == 0ull;
            } else {
                const u32 f = ld_sc1(a.flags + 16u * h);
                ok = __builtin_amdgcn_readfirstlane(f) >= 12u;
            }
            if (ok) break;
            __builtin_amdgcn_s_sleep(2);
        }
        if (tid == 0) {
            ok_s = ok;
            a.t_seen[h] = __builtin_amdgcn_s_memrealtime();
            a.spun[h] = polls;
        }
    }
    __syncthreads();
    u32 bad = 0;
    if (tid < 192u) {
        const u32 val = (a.variant & 2u) ? ld_plain(src) : ld_sc1(src);
        bad = val != a.iter || !ok_s;
        if (stale != a.iter - 1u && a.iter > 1u) bad |= 2u;  // (the planted value must be the previous launch's: kernel boundaries work)
    }
    const unsigned long long any = __builtin_amdgcn_ballot_w64(bad & 1u), any2 = __builtin_amdgcn_ballot_w64(bad & 2u);
    __shared__ u32 nb[8];
    if ((tid & 63u) == 0) nb[tid >> 6] = (u32)__builtin_popcountll(any) + 1000u * (u32)__builtin_popcountll(any2);
    __syncthreads();
    if (tid == 0) {
        a.t_data[h] = __builtin_amdgcn_s_memrealtime();
        a.bad[h] = nb[0] + nb[1] + nb[2];
    }
    if (tid < 16u) st_sc1(a.flags + 16u * h + tid, 0u);  // the consumer re-arms its flags (the next launch's producers start after this kernel ends)
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 300;
    u32 *q, *k, *v, *flags, *bad, *spun;
    u64 *ts, *tseen, *tdata;
    CHECK(hipMalloc(&q, 32 * 64 * 4)); CHECK(hipMalloc(&k, 8 * 64 * 4)); CHECK(hipMalloc(&v, 8 * 64 * 4));
    CHECK(hipMalloc(&flags, 32 * 16 * 4)); CHECK(hipMalloc(&bad, 32 * 4)); CHECK(hipMalloc(&spun, 32 * 4));
    CHECK(hipMalloc(&ts, 192 * 8)); CHECK(hipMalloc(&tseen, 32 * 8)); CHECK(hipMalloc(&tdata, 32 * 8));
    for (u32 variant : {0u, 4u, 6u, 4u}) {
        CHECK(hipMemset(q, 0, 32 * 64 * 4)); CHECK(hipMemset(k, 0, 8 * 64 * 4)); CHECK(hipMemset(v, 0, 8 * 64 * 4)); CHECK(hipMemset(flags, 0, 32 * 16 * 4));
        std::vector<double> seen_us, data_us, kern_us;
        u32 nbad = 0, nstale_plant = 0, maxpolls = 0, timeouts = 0;
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        for (int it = 1; it <= iters; it++) {
            Args a{q, k, v, flags, ts, tseen, tdata, bad, spun, (u32)it, variant, 400u /* 4 us at 100 MHz */};
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(handoff_kernel, dim3(224), dim3(1024), 0, 0, a);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipDeviceSynchronize());
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            u64 hts[192], hseen[32], hdata[32]; u32 hbad[32], hspun[32];
            CHECK(hipMemcpy(hts, ts, sizeof hts, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(hseen, tseen, sizeof hseen, hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(hdata, tdata, sizeof hdata, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(hbad, bad, sizeof hbad, hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(hspun, spun, sizeof hspun, hipMemcpyDeviceToHost));
            if (it <= 5) continue;  // warm-up
            kern_us.push_back(ms * 1e3);
            for (int h = 0; h < 32; h++) {
                u64 last = 0;
                for (int p = 0; p < 4; p++) last = std::max({last, hts[4 * h + p], hts[128 + 4 * (h / 4) + p], hts[160 + 4 * (h / 4) + p]});
                seen_us.push_back(((double)hseen[h] - (double)last) / 100.0);
                data_us.push_back(((double)hdata[h] - (double)last) / 100.0);
                nbad += hbad[h] % 1000u; nstale_plant += hbad[h] / 1000u;
                maxpolls = std::max(maxpolls, hspun[h]);
                if (hspun[h] >= (1u << 20)) timeouts++;
            }
        }
        auto pct = [](std::vector<double> &x, double p) { std::sort(x.begin(), x.end()); return x[(size_t)(p * (x.size() - 1))]; };
        printf("variant %u (%s): flag seen %.2f / %.2f / %.2f us, data in hand %.2f / %.2f / %.2f us behind the last producer's stores (median / p90 / max); "
               "launch %.2f us median; wrong words %u, planted-not-previous %u, time-outs %u, max polls %u\n",
               variant, variant == 0 ? "atomic counter; sc1 stores + vmcnt(0), sc1 loads" : (variant == 1 ? "+ buffer_wbl2 sc1 in the producers" : (variant == 2 ? "atomic counter, PLAIN loads of the data behind the flag" : (variant == 4 ? "one flag word per producer (sc1 stores), sc1 loads" : "flag words, PLAIN loads of the data behind the flags"))),
               pct(seen_us, 0.5), pct(seen_us, 0.9), pct(seen_us, 1.0), pct(data_us, 0.5), pct(data_us, 0.9), pct(data_us, 1.0), pct(kern_us, 0.5), nbad,
               nstale_plant, timeouts, maxpolls);
    }
    return 0;
}
