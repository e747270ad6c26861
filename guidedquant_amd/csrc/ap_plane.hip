// ap_plane.hip -- "fast mode" Any-Precision GEMV for gfx950: binary bit-plane GEMVs on the matrix cores.
//
// See plane_core.h for the algorithm.  What runs where:
//   HBM    : the stored bit-planes, whole 128-byte lines (one row x one plane x one 1024-weight chunk), fetched by
//            direct-to-LDS loads (buffer_load_dwordx4 ... lds) that every wave issues for ALL of its steps (up to
//            its LDS ring depth) in its first instructions: the block's share of the matrix is in flight before
//            the activation prologue starts, and the matrix cores chase the stream.
//   VALU   : one v_and_b32 per 4 weights per plane-subset (bits -> bf8 {0, 2^e}); ANDs of planes for the subsets.
//   MFMA   : v_mfma_scale_f32_16x16x128_f8f6f4 (A = bf8 bit patterns, scale cancels 2^e; B = 4 bf8 pieces of x in
//            columns 0..3, per-32-element block scale), fp32 accumulate: 8 MFMAs per (chunk, plane-subset).
//   LDS    : per-wave rings of A tiles (plane_core.h atile_unit swizzle), the B image (activation pieces, 4*K bytes)
//            built once per block, partial sums of K-split items.
// Epilogue : y = coef[0] * sum(x) + sum_S coef[S] * T[S]  (Moebius coefficients of the row's LUT), fp32 -> fp16.
//
// This path is NOT bit-identical to the reference's fp16-accumulated kernel (anyprec.cu:372-542): it is closer
// to the exact product than the reference is (products exact, fp32 accumulation).  The bit-exact path is
// ap_gemv.hip; gq_set_ap_mode() / GQ_AP_MODE choose.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gq_internal.h"
#include "plane_core.h"

using namespace gqp;

namespace {

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef u32 u32x4 __attribute__((ext_vector_type(4)));

struct PlaneArgs {
    const u32 *qw;
    const uint16_t *lut;
    const uint16_t *x;
    uint16_t *out;
    const uint16_t *normw;
    const uint16_t *resid;
    u32 N, K;
    u32 RGB;     // row groups (16 rows) per block
    u32 log2CS;  // a row group's chunks are split over 2^log2CS wave items
    u32 cpi;     // chunks per item
    u32 S;       // LDS ring slots (steps) per wave
    u32 xflags;  // experiments (GQ_PL_XFLAGS): 1 = no MFMA work, 2 = no plane stream, 4 = no prologue image
    float eps;
    unsigned long long *dbg;  // optional: per-wave phase timestamps of block 0 (tools/phase_timing.py)
};

enum { PRO_NONE = 0, PRO_RMSNORM = 1, PRO_SILUMUL = 2 };

__device__ __forceinline__ float h2f(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
__device__ __forceinline__ uint16_t f2h(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }

// wave64 reductions on the DPP path (no LDS round trips): result valid in lane 63
template <bool MAX>
__device__ __forceinline__ float wave_reduce(float v) {
    auto step = [&](auto dpp) {
        float o = __builtin_bit_cast(float, dpp(__builtin_bit_cast(int, v)));
        v = MAX ? fmaxf(v, o) : v + o;
    };
    step([](int x) { return __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xF, 0xF, false); });   // quad_perm [1,0,3,2]
    step([](int x) { return __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xF, 0xF, false); });   // quad_perm [2,3,0,1]
    step([](int x) { return __builtin_amdgcn_update_dpp(x, x, 0x141, 0xF, 0xF, false); });  // row_half_mirror
    step([](int x) { return __builtin_amdgcn_update_dpp(x, x, 0x140, 0xF, 0xF, false); });  // row_mirror
    {   // row_bcast15 into rows 1 and 3, row_bcast31 into rows 2 and 3 (identity elsewhere)
        const int ident = MAX ? 0 : 0;  // |x| >= 0 and the sum identity are both 0.0f
        float o = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(ident, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, false));
        v = MAX ? fmaxf(v, o) : v + o;
        o = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(ident, __builtin_bit_cast(int, v), 0x143, 0xC, 0xF, false));
        v = MAX ? fmaxf(v, o) : v + o;
    }
    return v;
}


typedef _Float16 h2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ h2v u2h2(u32 u) { return __builtin_bit_cast(h2v, u); }
__device__ __forceinline__ u32 h22u(h2v h) { return __builtin_bit_cast(u32, h); }

// --- hand-scheduled vector memory.  The compiler's waitcnt pass treats an LDS-DMA load as "may alias every later
// LDS access" and drains vmcnt before the first ds_read / barrier, which would serialise the prologue behind the
// whole plane stream.  So the plane loads and the activation loads that must overtake them are issued from inline
// asm (invisible to that pass) and waited for with explicit s_waitcnt vmcnt(n) (vector memory returns in order).
__device__ __forceinline__ void dma16(u32x4 rsrc, u32 lds_base, u32 voff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen nt lds" ::"s"(lds_base), "v"(voff), "s"(rsrc)
                 : "memory");  // m0 is not otherwise used in this kernel (no compiler-generated LDS-DMA / movrel)
}
__device__ __forceinline__ void dma16s(u32x4 rsrc, u32 lds_base, u32 voff, u32 soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen nt lds" ::"s"(lds_base), "v"(voff), "s"(rsrc), "s"(soff)
                 : "memory");
}
__device__ __forceinline__ u32 bload32(u32x4 rsrc, u32 voff) {
    u32 r;
    asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(r) : "v"(voff), "s"(rsrc) : "memory");
    return r;
}
__device__ __forceinline__ u32 bload32s(u32x4 rsrc, u32 voff, u32 soff) {
    u32 r;
    asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=v"(r) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
    return r;
}
template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// wait until at most n * LPS vector-memory instructions of this wave are outstanding (n uniform, 0..4)
template <int LPS>
__device__ __forceinline__ void wait_vm_steps(u32 n) {
    switch (n) {
        case 0: wait_vm<0>(); break;
        case 1: wait_vm<LPS>(); break;
        case 2: wait_vm<2 * LPS>(); break;
        case 3: wait_vm<3 * LPS>(); break;
        default: wait_vm<4 * LPS>(); break;
    }
}
// data dependence of asm-loaded registers on the preceding s_waitcnt (volatile asm statements keep their order)
__device__ __forceinline__ void tie4(u32 (&r)[4]) { asm volatile("" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3])); }

__device__ __forceinline__ u32x4 make_rsrc(const void *p, u32 bytes) {
    const uint64_t a = (uint64_t)(uintptr_t)p;
    return (u32x4){(u32)a, (u32)(a >> 32) & 0xFFFFu, bytes, 0x00020000u};  // raw buffer: stride 0, num_records = bytes
}

// software barrier among the consumer waves (the producer waves may be stalled in the vector-memory issue queue
// for microseconds, so s_barrier is only used once, at the very start of the kernel)
__device__ __forceinline__ void arrive_and_wait(u32 *ctr, u32 target, u32 lane) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ void spin_nonzero(const u32 *flag) {
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0u) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// Roles.  A block is W waves: NC consumers (activation prologue, MFMA, epilogue) and NPR = W / 4 producers (one per
// SIMD) that do nothing but issue the direct-to-LDS plane loads in consumption order and publish "tile ready"
// flags.  Work unit ("step") = 16 rows x one 1024-weight chunk x all planes; an item = cpi consecutive chunks of a
// row group, accumulated in registers by one consumer.  Sequence number of a step:
//     seq = (round * cpi + c) * NC + k      consumer k, its round-th item (item = round * NC + k), chunk c of the item
// Producer p owns seq = p (mod NPR); tile seq lives in ring slot seq % R.
template <int BITS, int PRO, int NI>
__global__ void __launch_bounds__(BITS == 2 ? 1024 : 512) ap_plane_kernel(PlaneArgs a) {
    constexpr int NP = 1 << BITS, NP1 = NP - 1;
    constexpr u32 T = BITS == 2 ? 1024u : 512u, W = T / 64u, NPR = W / 4u, NC = W - NPR, TC = NC * 64u;
    // NI prologue passes over the (chunk, virtual lane, weight pair) items: pass n gives wave w the chunk (w >> 1) + n * TC / 128
    static_assert(TC % 128u == 0, "a wave's prologue items must share one chunk");
    constexpr u32 LPS = 2u * BITS;              // direct-to-LDS loads per step
    constexpr u32 SLOT = 2048u * BITS;
    constexpr u32 OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    Geom G;
    G.init(a.K);
    const u32 tid = threadIdx.x;
    const u32 w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const u32 l = tid & 63u;
    const u32 CS = 1u << a.log2CS, cpi = a.cpi, R = a.S;
    const u32 nIt = a.RGB * CS;  // items of this block
    const u32 rounds = (nIt + NC - 1u) / NC;
    const u32 nseq = rounds * cpi * NC;
    // LDS: [A ring: R slots][LUT rows of the block][B image: nchunks * 4096][zero32 (64 B)][red: 64 floats][ctr: 4][ready: nseq][done: nseq][part]
    unsigned char *ring = smem;
    const u32 lut_bytes = (a.RGB * 16u * (u32)NP * 2u + 1023u) & ~1023u;
    const u32 *lutl = reinterpret_cast<const u32 *>(smem + (size_t)R * SLOT);  // fp16 [RGB * 16][NP], fetched by producer 0
    unsigned char *bimg = smem + (size_t)R * SLOT + lut_bytes;
    unsigned char *zero32 = bimg + G.nchunks * 4096u;
    float *red = reinterpret_cast<float *>(zero32 + 64);
    u32 *ctr = reinterpret_cast<u32 *>(red + 64);
    u32 *ready = ctr + 4;
    u32 *done = ready + nseq;
    float *part = reinterpret_cast<float *>(done + nseq);  // [item][subset][16 rows]
    const u32 rg0 = blockIdx.x * a.RGB;
    const u32 m = blockIdx.y;
    auto stamp = [&](int i) {
        if (a.dbg && blockIdx.x == gridDim.x / 2 && l == 0) a.dbg[w * 8u + (u32)i] = __builtin_readcyclecounter();
    };
    stamp(0);
    if (a.xflags & 32u) return;
    // waves 0..NC-1 are the consumers; the producers are the last waves of the block: they are launched ~500 cycles
    // after wave 0, by when the consumers' activation loads are in the memory pipe ahead of the plane stream
    const bool producer = w >= NC;
    const u32 k = w, ctid = tid;

    // ---------------------------------------------------------------- producer state (plane stream)
    // this producer's steps in consumption order: seq = (rd * cpi + c) * NC + kc with kc = w (mod NPR); NC % NPR == 0
    static_assert(NC % NPR == 0, "every producer serves a fixed set of consumers");
    struct It {
        u32 rd, c, kk;
    };
    constexpr u32 KPP = NC / NPR;
    const u32 pi = w - NC;  // producer index
    auto it_seq = [&](const It &i) { return (i.rd * cpi + i.c) * NC + pi + NPR * i.kk; };
    auto it_valid = [&](const It &i, u32 &item, u32 &chunk) {
        item = i.rd * NC + pi + NPR * i.kk;
        chunk = (item & (CS - 1u)) * cpi + i.c;
        return item < nIt && chunk < G.nchunks;
    };
    auto it_next = [&](It &i) {
        if (++i.kk == KPP) {
            i.kk = 0;
            if (++i.c == cpi) {
                i.c = 0;
                i.rd++;
            }
        }
    };
    auto it_skip = [&](It &i) {  // advance to the next valid step (or the end: rd == rounds)
        u32 item, chunk;
        while (i.rd < rounds && !it_valid(i, item, chunk)) it_next(i);
    };
    It iss{0, 0, 0}, pub{0, 0, 0};
    u32 issued = 0, published = 0;
    const u32 plane_bytes = a.N * G.wpr * 4u;
    const u32x4 rq = make_rsrc(a.qw, plane_bytes * (u32)BITS);
    const u32 ring_lds = (u32)(uintptr_t)ring;
    // Per-lane part of a tile address is constant (atile_src); the tile part is wave-uniform and goes into the
    // scalar offset.  No per-lane validity: rows >= N and the words of a short tail chunk beyond the row fetch
    // in-range garbage (or zeros past the end of the tensor) that only ever meets zero activation pieces /
    // rows that are never stored.
    u32 lane_off[2];
    {
        u32 lr, lseg;
        atile_src(0u, l, lr, lseg);
        lane_off[0] = (lr * G.wpr + 4u * lseg) * 4u;
        atile_src(1u, l, lr, lseg);
        lane_off[1] = (lr * G.wpr + 4u * lseg) * 4u;
    }
    auto issue_step = [&]() {  // the step at `iss`
        u32 item, chunk;
        it_valid(iss, item, chunk);
        const u32 seq = it_seq(iss);
        const u32 rgi = rg0 + (item >> a.log2CS);
        const u32 slot_lds = ring_lds + (R == nseq ? seq : seq % R) * SLOT;
        const u32 tile_off = (rgi * 16u * G.wpr + 32u * chunk) * 4u;
#pragma unroll
        for (u32 p = 0; p < (u32)BITS; p++)
#pragma unroll
            for (u32 h = 0; h < 2; h++) dma16s(rq, slot_lds + (p * 2u + h) * 1024u, lane_off[h], tile_off + p * plane_bytes);
        issued++;
        it_next(iss);
        it_skip(iss);
    };

    // ---------------------------------------------------------------- 0. first loads, flag init, the only barrier
    // prologue item = (chunk, virtual lane t, weight pair jp): the weights j = 2jp, 2jp+1 of the 4 bytes c of lane t
    u32 xr[NI][4], ar[NI][4];
    const u32 pt = l & 31u, pjp = ((k & 1u) << 1) | (l >> 5);  // the same for every pass (TC is a multiple of 128)
    if (producer) {
        if (!(a.xflags & 2u)) {
            __builtin_amdgcn_s_setprio(3);  // the stream must not wait for issue slots behind the MFMA waves of this SIMD
            if (pi == 0 && !(a.xflags & 16u)) {
                // the LUT rows of the block, ahead of this producer's first tile: loads return in order, so they are in
                // LDS before tile seq 0 is published, long before the epilogue reads them
                const u32x4 rl = make_rsrc(a.lut, a.N * (u32)NP * 2u);
                const u32 lut_lds = ring_lds + R * SLOT;
                for (u32 o = 0; o < lut_bytes; o += 1024u) dma16(rl, lut_lds + o, rg0 * 16u * (u32)NP * 2u + o + 16u * l);
            }
            it_skip(iss);
            it_skip(pub);
        }
    } else {
        const u32x4 rsx = make_rsrc(a.x + (size_t)m * (PRO == PRO_SILUMUL ? 2u * G.K : G.K), (PRO == PRO_SILUMUL ? 4u : 2u) * G.K);
        const u32x4 rsa = make_rsrc(PRO == PRO_RMSNORM ? a.normw : a.x, 2u * G.K);
#pragma unroll
        for (u32 n = 0; n < (u32)NI; n++) {
            const u32 chunk = (k >> 1) + n * (TC / 128u);  // wave-uniform
            const u32 tp = G.tpw(chunk);
            const u32 voff = (chunk < G.nchunks && pt < tp && !(a.xflags & 8u)) ? 16u * pt + 4u * pjp : OOB;
#pragma unroll
            for (u32 c = 0; c < 4; c++) {
                const u32 soff = 2048u * chunk + 16u * tp * c;  // element 1024*chunk + 8*tp*c (+ 8t + 2jp per lane)
                xr[n][c] = bload32s(rsx, voff, soff);
                if constexpr (PRO == PRO_RMSNORM) ar[n][c] = bload32s(rsa, voff, soff);
                if constexpr (PRO == PRO_SILUMUL) ar[n][c] = bload32s(rsx, voff, soff + 2u * G.K);
            }
        }
    }
    for (u32 i = tid; i < 4u + 2u * nseq; i += T) ctr[i] = 0u;
    if (tid < 16) reinterpret_cast<u32 *>(zero32)[tid] = 0u;
    // The only hardware barrier of the kernel, as early as possible: behind it the flags are zero and every activation
    // load is in the memory pipe ahead of the plane stream.  (Later a producer may sit in the memory issue queue
    // for microseconds -- issuing blocks once the queues are full -- so it can not take part in barriers.)
    __syncthreads();

    if (producer) {
        if (a.xflags & 2u) return;
        // ------------------------------------------------------------ producer: publish what lands, stream the rest
        auto publish = [&]() {  // tiles are published in issue order
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (l == 0) __hip_atomic_store(ready + it_seq(pub), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            it_next(pub);
            it_skip(pub);
            published++;
        };
        // vector-memory operations of this wave still in flight (IB_STS.VM_CNT, bits 3:0 and 23:22): loads return in
        // order, so everything but the youngest ceil(vmcnt / LPS) steps has landed in LDS
        auto publish_landed = [&]() {
            const u32 sts = __builtin_amdgcn_s_getreg(7 | (0 << 6) | (31 << 11));
            const u32 vm = (sts & 0xFu) | ((sts >> 18) & 0x30u);
            const u32 inflight = (vm + LPS - 1u) / LPS;
            while (issued - published > inflight) publish();
        };
        while (iss.rd < rounds) {
            const u32 seq = it_seq(iss);
            if (seq >= R) {
                // ring slot still held by its consumer: keep publishing what lands while waiting for the release
                while (__hip_atomic_load(done + (seq - R), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0u) {
                    publish_landed();
                    __builtin_amdgcn_s_sleep(1);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            }
            issue_step();
            publish_landed();
        }
        while (published < issued) {
            publish_landed();
            if (published < issued) __builtin_amdgcn_s_sleep(1);
        }
        stamp(3);
        return;
    }

    // ================================================================ consumers
    if (a.xflags & 64u) return;
    const u32 r = l & 15u, kb = l >> 4;
    wait_vm<0>();
#pragma unroll
    for (u32 n = 0; n < (u32)NI; n++) {
        tie4(xr[n]);
        if constexpr (PRO != PRO_NONE) tie4(ar[n]);
    }
    stamp(6);

    // ---------------------------------------------------------------- 1. statistics -> one consumer barrier
    // RMSNorm: sum x^2 and max |x * w| (bounds the normalised maximum); otherwise max |x'| of the transformed vector
    float nscale = 0.f, xmax = 0.f;
    {
        float ss = 0.f, mx = 0.f;
        u32 mxi = 0;
#pragma unroll
        for (u32 n = 0; n < (u32)NI; n++)
#pragma unroll
            for (u32 c = 0; c < 4; c++) {
                if constexpr (PRO == PRO_RMSNORM) {
                    const float p = h2f(xr[n][c] & 0xFFFF), q = h2f(xr[n][c] >> 16);
                    ss += p * p;
                    ss += q * q;
                    mx = fmaxf(mx, fmaxf(fabsf(p * h2f(ar[n][c] & 0xFFFF)), fabsf(q * h2f(ar[n][c] >> 16))));
                } else {
                    if constexpr (PRO == PRO_SILUMUL) {
                        // F.silu(gate) * up on fp16 tensors -- inference/model.py:266
                        _Float16 hh[2];
#pragma unroll
                        for (int k = 0; k < 2; k++) {
                            const float gv = h2f((uint16_t)(xr[n][c] >> (16 * k)));
                            hh[k] = (_Float16)(gv / (1.0f + __expf(-gv))) * __builtin_bit_cast(_Float16, (uint16_t)(ar[n][c] >> (16 * k)));
                        }
                        xr[n][c] = (u32)__builtin_bit_cast(uint16_t, hh[0]) | ((u32)__builtin_bit_cast(uint16_t, hh[1]) << 16);
                    }
                    // |fp16| bit patterns order like unsigned integers
                    const u32 ab = xr[n][c] & 0x7FFF7FFFu;
                    mxi = max(mxi, max(ab & 0xFFFFu, ab >> 16));
                }
            }
        if constexpr (PRO != PRO_RMSNORM) mx = h2f((uint16_t)mxi);
        mx = wave_reduce<true>(mx);
        if constexpr (PRO == PRO_RMSNORM) ss = wave_reduce<false>(ss);
        if (l == 63) {
            red[k] = mx;
            if constexpr (PRO == PRO_RMSNORM) red[16 + k] = ss;
        }
        arrive_and_wait(ctr + 0, NC, l);
#pragma unroll
        for (u32 i = 0; i < NC; i++) xmax = fmaxf(xmax, red[i]);
        if constexpr (PRO == PRO_RMSNORM) {
            float tot = 0.f;
#pragma unroll
            for (u32 i = 0; i < NC; i++) tot += red[16 + i];
            nscale = 1.0f / sqrtf(tot / (float)G.K + a.eps);
            xmax = xmax * nscale * 1.002f;  // covers the two fp16 roundings of the transform
        }
    }
    stamp(7);

    // ---------------------------------------------------------------- 2. transform, scale, split, scatter
    const int ksh = piece_shift(xmax);
    const int sb = 127 - ksh;  // E8M0 scale of every B block
    {
        const uint16_t k16 = pow2_f16(ksh);
        const h2v kk = u2h2((u32)k16 * 0x10001u), one2 = u2h2(0x3C003C00u);
        float xsum = 0.f;
#pragma unroll
        for (u32 n = 0; n < (u32)NI; n++) {
            const u32 chunk = (k >> 1) + n * (TC / 128u);
            if (chunk >= G.nchunks) continue;
            const u32 t = pt, jp = pjp, kbi = t >> 3, v = t & 7u;
            u32 P[4][4];  // [piece][c]
#pragma unroll
            for (u32 c = 0; c < 4; c++) {
                u32 xw = xr[n][c];
                if constexpr (PRO == PRO_RMSNORM) {
                    // (x.float() * rsqrt(mean(x^2)+eps)).half() * w  -- inference/model.py:281-292, both fp16 roundings kept
                    const _Float16 h0 = (_Float16)(h2f(xw & 0xFFFF) * nscale), h1 = (_Float16)(h2f(xw >> 16) * nscale);
                    xw = h22u((h2v){h0, h1} * u2h2(ar[n][c]));
                }
                xsum = __builtin_amdgcn_fdot2(u2h2(xw), one2, xsum, false);
                h2v rem = u2h2(xw) * kk;
#pragma unroll
                for (u32 p = 0; p < 4; p++) {
                    P[p][c] = h22u(rem) & 0xFF00FF00u;
                    if (p < 3) rem = rem - u2h2(P[p][c]);
                }
            }
#pragma unroll
            for (u32 p = 0; p < 4; p++) {
                // image bytes B = 0..3 hold c = 3..0; the bf8 of weight 2jp is byte 1 of P, of weight 2jp+1 byte 3
                const u32 hi = __builtin_amdgcn_perm(P[p][3], P[p][2], 0x03070105u);
                const u32 lo = __builtin_amdgcn_perm(P[p][1], P[p][0], 0x03070105u);
                const u32 w0 = __builtin_amdgcn_perm(lo, hi, 0x05040100u), w1 = __builtin_amdgcn_perm(lo, hi, 0x07060302u);
                *reinterpret_cast<u32 *>(bimg + bimg_off(chunk, 7u - 2u * jp, kbi, p) + 4u * v) = w0;
                *reinterpret_cast<u32 *>(bimg + bimg_off(chunk, 6u - 2u * jp, kbi, p) + 4u * v) = w1;
            }
        }
        xsum = wave_reduce<false>(xsum);
        if (l == 63) red[32 + k] = xsum;
    }
    stamp(1);
    arrive_and_wait(ctr + 1, NC, l);
    stamp(2);
    float X = 0.f;
#pragma unroll
    for (u32 i = 0; i < NC; i++) X += red[32 + i];

    // ---------------------------------------------------------------- 3. main loop: the steps of this consumer
    const u32 col = l & 15u;
    const bool bcol = col < 4u;
    const u32 offA0 = atile_unit(r, 2u * kb) * 16u, offA1 = atile_unit(r, 2u * kb + 1u) * 16u;
    v4f acc[NP1];
#pragma unroll
    for (int i = 0; i < NP1; i++) acc[i] = (v4f){0.f, 0.f, 0.f, 0.f};
    u32 seq = k, slot_i = k;  // slot_i = seq % R, tracked without a division (R >= NC)
    for (u32 item = k; item < nIt; item += NC) {
        for (u32 c = 0; c < cpi; c++, seq += NC, slot_i = slot_i + NC >= R ? slot_i + NC - R : slot_i + NC) {
            const u32 chunk = (item & (CS - 1u)) * cpi + c;
            if (chunk >= G.nchunks) continue;
            if (!(a.xflags & 2u)) spin_nonzero(ready + seq);
            u32 Wd[BITS][8];
            {
                const unsigned char *slot = ring + slot_i * SLOT;
#pragma unroll
                for (int p = 0; p < BITS; p++) {
                    const uint4 a0 = *reinterpret_cast<const uint4 *>(slot + p * 2048 + offA0);
                    const uint4 a1 = *reinterpret_cast<const uint4 *>(slot + p * 2048 + offA1);
                    Wd[p][0] = a0.x, Wd[p][1] = a0.y, Wd[p][2] = a0.z, Wd[p][3] = a0.w;
                    Wd[p][4] = a1.x, Wd[p][5] = a1.y, Wd[p][6] = a1.z, Wd[p][7] = a1.w;
                }
            }
            if (R < nseq) {
                // the tile is in registers: hand the ring slot back to the producer
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (l == 0) __hip_atomic_store(done + seq, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            if (a.xflags & 1u) {
                acc[0][0] += __builtin_bit_cast(float, Wd[0][0] ^ Wd[BITS - 1][7]);
                continue;
            }
            // plane-subset words: code bit i lives in plane BITS-1-i
            u32 PW[NP1][8];
#pragma unroll
            for (int cm = 1; cm < NP; cm++) {
                const int low = cm & -cm, i0 = __builtin_ctz(cm), rest = cm ^ low;
#pragma unroll
                for (int v = 0; v < 8; v++)
                    PW[cm - 1][v] = rest ? (PW[rest - 1][v] & Wd[BITS - 1 - i0][v]) : Wd[BITS - 1 - i0][v];
            }
            // B operand (activation pieces) double-buffered over the bit position s
            const unsigned char *bbase = bcol ? bimg + bimg_off(chunk, 0u, kb, col) : zero32;
            const u32 bstep = bcol ? 512u : 0u;  // bimg_off(.., s+1, ..) - bimg_off(.., s, ..) = 4 pieces * 4 kb * 32 B
            uint4 bn0 = *reinterpret_cast<const uint4 *>(bbase);
            uint4 bn1 = *reinterpret_cast<const uint4 *>(bbase + 16);
#pragma unroll
            for (int s = 0; s < 8; s++) {
                const uint4 b0 = bn0, b1 = bn1;
                if (s < 7) {
                    bn0 = *reinterpret_cast<const uint4 *>(bbase + (u32)(s + 1) * bstep);
                    bn1 = *reinterpret_cast<const uint4 *>(bbase + (u32)(s + 1) * bstep + 16);
                }
                // keep the loads of s + 1 ahead of the MFMAs of s (the scheduler otherwise sinks them behind the
                // MFMAs into a single B buffer and exposes the LDS latency 8 times per step)
                __builtin_amdgcn_sched_barrier(0);
                v8i Bv = {(int)b0.x, (int)b0.y, (int)b0.z, (int)b0.w, (int)b1.x, (int)b1.y, (int)b1.z, (int)b1.w};
#pragma unroll
                for (int cm = 1; cm < NP; cm++) {
                    v8i Av;
#pragma unroll
                    for (int v = 0; v < 8; v++) Av[v] = (int)extract(PW[cm - 1][v], s);
                    acc[cm - 1] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(Av, Bv, acc[cm - 1], 1, 1, 0, scale_byte(s), 0, sb);
                }
            }
        }
        // item done: add the 4 piece columns (lanes col = 0..3 of each 16-lane group), park 16 x NP1 sums in LDS
#pragma unroll
        for (int cm = 0; cm < NP1; cm++) {
            v4f v = acc[cm];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                float f = v[q];
                f += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, f), 0xB1, 0xF, 0xF, false));
                f += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, f), 0x4E, 0xF, 0xF, false));
                v[q] = f;
            }
            // part[item][subset][row]: lane (col 0, kb) owns rows 4kb..4kb+3
            if (col == 0u) *reinterpret_cast<v4f *>(part + ((size_t)item * NP1 + cm) * 16u + 4u * kb) = v;
            acc[cm] = (v4f){0.f, 0.f, 0.f, 0.f};
        }
    }
    stamp(3);
    arrive_and_wait(ctr + 2, NC, l);
    stamp(4);

    // ---------------------------------------------------------------- 4. epilogue: coefficients x plane sums
    // lane = (output row, Moebius index c): term c = coef[c] * (c == 0 ? sum(x) : T[c]); the NP terms of a row sit in
    // NP adjacent lanes and are added by a fixed DPP tree (deterministic order)
    for (u32 e = ctid; e < a.RGB * 16u * (u32)NP; e += TC) {
        const u32 i = e / (u32)NP, c = e % (u32)NP;
        const u32 rgl = i >> 4, rr = i & 15u;
        const u32 row = (rg0 + rgl) * 16u + rr;
        float f[NP];
#pragma unroll
        for (int cc = 0; cc < NP / 2; cc++) {
            const u32 lw = lutl[(size_t)i * (NP / 2) + cc];
            f[2 * cc] = h2f((uint16_t)(lw & 0xFFFF));
            f[2 * cc + 1] = h2f((uint16_t)(lw >> 16));
        }
        moebius<BITS>(f);
        float coef = f[0];
#pragma unroll
        for (int cc = 1; cc < NP; cc++) coef = c == (u32)cc ? f[cc] : coef;
        float term = X;
        if (c != 0u) {
            term = 0.f;
            const float *pp = part + (((size_t)(rgl << a.log2CS)) * NP1 + (c - 1u)) * 16u + rr;
            for (u32 cs = 0; cs < CS; cs++) term += pp[(size_t)cs * NP1 * 16u];
        }
        float y = coef * term;
        y += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, y), 0xB1, 0xF, 0xF, false));
        y += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, y), 0x4E, 0xF, 0xF, false));
        if constexpr (NP >= 8) y += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, y), 0x141, 0xF, 0xF, false));
        if constexpr (NP >= 16) y += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, y), 0x140, 0xF, 0xF, false));
        if (c == 0u && row < a.N) {
            _Float16 yh = (_Float16)y;
            if (a.resid) yh = __builtin_bit_cast(_Float16, a.resid[(size_t)m * a.N + row]) + yh;
            a.out[(size_t)m * a.N + row] = __builtin_bit_cast(uint16_t, yh);
        }
    }
    stamp(5);
}

struct PlaneCfg {
    u32 grid, T, RGB, log2CS, cpi, S, NI;
    size_t smem;
};

int g_cus = 0;
int cus() {
    if (!g_cus) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
            g_cus = n;
        else
            g_cus = 256;
    }
    return g_cus;
}

bool pick_plane_cfg(u32 N, u32 K, int bits, PlaneCfg &c) {
    if (K % 256u || K > 16384u) return false;
    const u32 nchunks = K / 1024u + ((K % 1024u) ? 1u : 0u);
    const u32 RGt = (N + 15u) / 16u;
    const u32 ncu = (u32)cus();
    const u32 W = bits == 2 ? 16u : 8u, NC = W - W / 4u;
    c.T = 64u * W;
    // one block per CU when the matrix is big enough
    u32 rgb = (RGt + ncu - 1u) / ncu;
    if (rgb < 1) rgb = 1;
    c.RGB = rgb;
    c.grid = (RGt + rgb - 1u) / rgb;
    // split K of a row group over 2^log2CS items until the block has at least two items per consumer wave
    u32 lcs = 0;
    while ((rgb << lcs) < 2u * NC && (2u << lcs) <= nchunks) lcs++;
    const int envcs = gq_env_int("GQ_PL_LOG2CS", -1);
    if (envcs >= 0 && (1u << envcs) <= nchunks) lcs = (u32)envcs;
    c.log2CS = lcs;
    c.cpi = (nchunks + (1u << lcs) - 1u) >> lcs;
    const u32 nIt = rgb << lcs;
    const u32 nseq = ((nIt + NC - 1u) / NC) * c.cpi * NC;
    if (nseq > 2048u) return false;
    // ring: the whole share of the block when it fits next to the B image, else as many slots as fit
    const u32 np1 = (1u << bits) - 1u;
    const size_t lutb = ((size_t)rgb * 16u * (np1 + 1u) * 2u + 1023u) & ~(size_t)1023u;
    const size_t fixed = lutb + (size_t)nchunks * 4096u + 64u + 64u * 4u + 16u + 8u * (size_t)nseq + (size_t)nIt * np1 * 16u * 4u;
    const size_t slot = 2048u * (size_t)bits, lds = 160u * 1024u;
    if (fixed + slot > lds) return false;
    u32 R = (u32)((lds - fixed) / slot);
    if (R > nseq) R = nseq;
    if (R < (nseq < NC ? nseq : NC)) return false;  // at least one tile per consumer wave
    const int envs = gq_env_int("GQ_PL_S", 0);
    if (envs >= 1 && (u32)envs < R) R = (u32)envs;
    c.S = R;
    c.NI = (nchunks * 128u + NC * 64u - 1u) / (NC * 64u);  // prologue passes
    c.smem = fixed + (size_t)R * slot;
    return true;
}

template <int BITS, int PRO, int NI>
int launch_plane_inst(const PlaneArgs &a, const PlaneCfg &c, u32 M, hipStream_t s) {
    static bool attr_set = false;
    auto kern = ap_plane_kernel<BITS, PRO, NI>;
    if (!attr_set) {
        GQ_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)(160u * 1024u)));
        attr_set = true;
    }
    dim3 grid(c.grid, M), block(c.T);
    hipLaunchKernelGGL(kern, grid, block, c.smem, s, a);
    GQ_HIP_CHECK(hipGetLastError());
    return GQ_OK;
}

template <int BITS, int PRO>
int launch_plane_ni(const PlaneArgs &a, const PlaneCfg &c, u32 M, hipStream_t s) {
    constexpr u32 NIMAX = BITS == 2 ? 3u : 6u;  // 2048 items over 768 / 384 consumer lanes
    if (c.NI > NIMAX) return GQ_ENOTSUP;
    if (c.NI <= 1) return launch_plane_inst<BITS, PRO, 1>(a, c, M, s);
    if (c.NI == 2) return launch_plane_inst<BITS, PRO, 2>(a, c, M, s);
    if (c.NI == 3) return launch_plane_inst<BITS, PRO, 3>(a, c, M, s);
    if constexpr (BITS != 2) return launch_plane_inst<BITS, PRO, 6>(a, c, M, s);
    return GQ_ENOTSUP;
}

template <int BITS>
int launch_plane(const PlaneArgs &a, const PlaneCfg &c, u32 M, int pro, hipStream_t s) {
    switch (pro) {
        case PRO_RMSNORM: return launch_plane_ni<BITS, PRO_RMSNORM>(a, c, M, s);
        case PRO_SILUMUL: return launch_plane_ni<BITS, PRO_SILUMUL>(a, c, M, s);
        default: return launch_plane_ni<BITS, PRO_NONE>(a, c, M, s);
    }
}

unsigned long long *g_dbg = nullptr;
}  // namespace

extern "C" void gq_debug_set_timing_buffer(void *p) { g_dbg = (unsigned long long *)p; }

// returns GQ_ENOTSUP when the shape is not served by this path (caller falls back to the exact kernels)
int gq_plane_gemv_try(const void *x, void *out, const uint32_t *qweight, const void *lut, uint32_t M, uint32_t N, uint32_t K,
                      int bits, const void *normw, float eps, const void *resid, int pro, hipStream_t stream) {
    if (bits < 2 || bits > 4) return GQ_ENOTSUP;
    const uint64_t qbytes = (uint64_t)bits * N * (K / 8u);
    if (qbytes >= 0x7FFFFFFFull) return GQ_ENOTSUP;
    if (((uintptr_t)qweight | (uintptr_t)x | (uintptr_t)normw) & 15u) return GQ_ENOTSUP;
    PlaneCfg c;
    if (!pick_plane_cfg(N, K, bits, c)) return GQ_ENOTSUP;
    PlaneArgs a{};
    a.qw = qweight;
    a.lut = (const uint16_t *)lut;
    a.x = (const uint16_t *)x;
    a.out = (uint16_t *)out;
    a.normw = (const uint16_t *)normw;
    a.resid = (const uint16_t *)resid;
    a.N = N;
    a.K = K;
    a.RGB = c.RGB;
    a.log2CS = c.log2CS;
    a.cpi = c.cpi;
    a.S = c.S;
    a.xflags = (u32)gq_env_int("GQ_PL_XFLAGS", 0);
    a.eps = eps;
    a.dbg = g_dbg;
    switch (bits) {
        case 2: return launch_plane<2>(a, c, M, pro, stream);
        case 3: return launch_plane<3>(a, c, M, pro, stream);
        default: return launch_plane<4>(a, c, M, pro, stream);
    }
}
