// valu_rates.hip -- measures per-instruction issue rates on gfx950 (lane-ops per cycle per SIMD) to size the
// AP-GEMV decode budget.  Each kernel runs ITER iterations of 16 independent copies of one instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int ITER = 2048;

#define REP16(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)

#define KERNEL3(NAME, ASM)                                                                 \
    __global__ void NAME(uint32_t *out, uint32_t seed) {                                    \
        uint32_t r[16], a = seed + threadIdx.x, b = seed * 3 + threadIdx.x, c = 0x03020100; \
        _Pragma("unroll") for (int i = 0; i < 16; i++) r[i] = a + i;                        \
        for (int it = 0; it < ITER; it++) {                                                 \
            REP16(ASM)                                                                      \
        }                                                                                   \
        uint32_t s = 0;                                                                     \
        _Pragma("unroll") for (int i = 0; i < 16; i++) s ^= r[i];                           \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                     \
    }

#define A_PERM(i) asm volatile("v_perm_b32 %0, %1, %0, %2" : "+v"(r[i]) : "v"(a), "v"(c));
#define A_PKFMA(i) asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(r[i]) : "v"(a), "v"(b));
#define A_AND(i) asm volatile("v_and_b32 %0, %1, %0" : "+v"(r[i]) : "v"(a));
#define A_LSHR(i) asm volatile("v_lshrrev_b32 %0, 2, %0" : "+v"(r[i]));
#define A_BFI(i) asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(r[i]) : "v"(a), "v"(b));
#define A_ANDOR(i) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
#define A_LSHLOR(i) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(r[i]) : "v"(a));
#define A_FMAF32(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r[i]) : "v"(a), "v"(b));
#define A_FMAF16(i) asm volatile("v_fma_f16 %0, %1, %2, %0" : "+v"(r[i]) : "v"(a), "v"(b));
#define A_MADU24(i) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(r[i]) : "v"(a), "v"(b));
#define A_MULU24(i) asm volatile("v_mul_u32_u24 %0, %1, %0" : "+v"(r[i]) : "v"(a));
#define A_DOT2(i) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(r[i]) : "v"(a), "v"(b));
#define A_DOT2C(i) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
#define A_PKADD(i) asm volatile("v_pk_add_f16 %0, %1, %0" : "+v"(r[i]) : "v"(a));
#define A_PKMUL(i) asm volatile("v_pk_mul_f16 %0, %1, %0" : "+v"(r[i]) : "v"(a));
#define A_BFE(i) asm volatile("v_bfe_u32 %0, %0, 3, 4" : "+v"(r[i]));
#define A_CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(a));
#define A_XOR(i) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(r[i]) : "v"(a));
#define A_ADD(i) asm volatile("v_add_u32 %0, %1, %0" : "+v"(r[i]) : "v"(a));
#define A_PKFMAF32(i) /* 64-bit regs: use pairs */
#define A_ALIGNBIT(i) asm volatile("v_alignbit_b32 %0, %1, %0, 5" : "+v"(r[i]) : "v"(a));
#define A_LSHLADD(i) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(r[i]) : "v"(a));
#define A_ADD3(i) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
#define A_OR3(i) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
#define A_PKLSHR16(i) asm volatile("v_pk_lshrrev_b16 %0, 2, %0" : "+v"(r[i]));
#define A_PKMADU16(i) asm volatile("v_pk_mad_u16 %0, %1, %2, %0" : "+v"(r[i]) : "v"(a), "v"(b));
#define A_BPERM(i) asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(r[i]) : "v"(a));
#define A_MOVDPP(i) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r[i]));
#define A_PERMSGPR(i) asm volatile("v_perm_b32 %0, %1, %0, %2" : "+v"(r[i]) : "v"(a), "s"(seed));
#define A_ANDSDWA(i) asm volatile("v_and_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "+v"(r[i]) : "v"(a));
#define A_CVTPKU8(i) asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(r[i]) : "v"(a));
#define A_SAD(i) asm volatile("v_sad_u8 %0, %1, %2, %0" : "+v"(r[i]) : "v"(a), "v"(b));
#define A_SWAP16(i) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(r[i]), "+v"(r[(i + 8) & 15]));
#define A_SWAP32(i) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(r[i]), "+v"(r[(i + 8) & 15]));
#define A_ADDDPP(i) asm volatile("v_add_f32_dpp %0, %0, %1 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(r[i]) : "v"(a));
// a DEPENDENT chain: every instruction reads what the previous one wrote (the latency, not the rate)
#define A_ADDDPP_CHAIN(i) asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(r[0]));
#define A_SWAP16_CHAIN(i) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(r[0]), "+v"(r[1]));
#define A_ADD_CHAIN(i) asm volatile("v_add_f32 %0, %0, %0" : "+v"(r[0]));
#define A_CNDMASK_S(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "s"(m64));
#define A_CNDMASK_V(i) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(a) : );
#define A_CMP(i) asm volatile("v_cmp_lt_u32_e64 %0, %1, %2" : "=s"(cm[i & 3]) : "v"(r[i]), "v"(a));
#define A_MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
#define A_MULHI(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
#define A_MFMA444(i)

KERNEL3(k_perm, A_PERM)
KERNEL3(k_pkfma, A_PKFMA)
KERNEL3(k_and, A_AND)
KERNEL3(k_lshr, A_LSHR)
KERNEL3(k_bfi, A_BFI)
KERNEL3(k_andor, A_ANDOR)
KERNEL3(k_lshlor, A_LSHLOR)
KERNEL3(k_fmaf32, A_FMAF32)
KERNEL3(k_fmaf16, A_FMAF16)
KERNEL3(k_madu24, A_MADU24)
KERNEL3(k_mulu24, A_MULU24)
KERNEL3(k_dot2, A_DOT2)
KERNEL3(k_dot2c, A_DOT2C)
KERNEL3(k_pkadd, A_PKADD)
KERNEL3(k_pkmul, A_PKMUL)
KERNEL3(k_bfe, A_BFE)
KERNEL3(k_cndmask, A_CNDMASK)
KERNEL3(k_xor, A_XOR)
KERNEL3(k_add, A_ADD)
KERNEL3(k_alignbit, A_ALIGNBIT)
KERNEL3(k_lshladd, A_LSHLADD)
KERNEL3(k_add3, A_ADD3)
KERNEL3(k_or3, A_OR3)
KERNEL3(k_pklshr16, A_PKLSHR16)
KERNEL3(k_pkmadu16, A_PKMADU16)
KERNEL3(k_bperm, A_BPERM)
KERNEL3(k_movdpp, A_MOVDPP)
KERNEL3(k_permsgpr, A_PERMSGPR)
KERNEL3(k_andsdwa, A_ANDSDWA)
KERNEL3(k_sad, A_SAD)
KERNEL3(k_swap16, A_SWAP16)
KERNEL3(k_swap32, A_SWAP32)
KERNEL3(k_adddpp, A_ADDDPP)
KERNEL3(k_adddpp_chain, A_ADDDPP_CHAIN)
KERNEL3(k_swap16_chain, A_SWAP16_CHAIN)
KERNEL3(k_add_chain, A_ADD_CHAIN)

__global__ void k_cndmask_s(uint32_t *out, uint32_t seed) {
    uint32_t r[16], a = seed + threadIdx.x;
    unsigned long long m64 = 0x5555555555555555ull * (seed | 1u);
    _Pragma("unroll") for (int i = 0; i < 16; i++) r[i] = a + i;
    for (int it = 0; it < ITER; it++) { REP16(A_CNDMASK_S) }
    uint32_t s = 0;
    _Pragma("unroll") for (int i = 0; i < 16; i++) s ^= r[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_cndmask_v(uint32_t *out, uint32_t seed) {
    uint32_t r[16], a = seed + threadIdx.x;
    _Pragma("unroll") for (int i = 0; i < 16; i++) r[i] = a + i;
    asm volatile("s_mov_b64 vcc, 0x5555" ::: "vcc");
    for (int it = 0; it < ITER; it++) { REP16(A_CNDMASK_V) }
    uint32_t s = 0;
    _Pragma("unroll") for (int i = 0; i < 16; i++) s ^= r[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_cmp(uint32_t *out, uint32_t seed) {
    uint32_t r[16], a = seed + threadIdx.x;
    unsigned long long cm[4] = {0, 0, 0, 0};
    _Pragma("unroll") for (int i = 0; i < 16; i++) r[i] = a + i;
    for (int it = 0; it < ITER; it++) { REP16(A_CMP) }
    uint32_t s = (uint32_t)(cm[0] ^ cm[1] ^ cm[2] ^ cm[3]);
    _Pragma("unroll") for (int i = 0; i < 16; i++) s ^= r[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
KERNEL3(k_mullo, A_MULLO)
KERNEL3(k_mulhi, A_MULHI)

// LDS random 4-byte reads (conflict pattern set by address), 16 outstanding per iteration
__global__ void k_ldsread(uint32_t *out, uint32_t seed) {
    __shared__ uint32_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i * 2654435761u;
    __syncthreads();
    uint32_t addr = ((threadIdx.x * 17 + seed) & 1023) * 4, s = 0;
    for (int it = 0; it < ITER; it++) {
        uint32_t v[16];
#pragma unroll
        for (int i = 0; i < 16; i++) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v[i]) : "v"(addr), "i"(i * 64));
        asm volatile("s_waitcnt lgkmcnt(0)");
#pragma unroll
        for (int i = 0; i < 16; i++) s ^= v[i];
        addr = (addr + (s & 4)) & 4095;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

typedef void (*kern_t)(uint32_t *, uint32_t);

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    printf("device %s CUs=%d clock=%d MHz\n", prop.name, cus, prop.clockRate / 1000);
    uint32_t *out;
    CHECK(hipMalloc(&out, 256 * 8 * 1024 * 4 * 4));
    struct K { const char *name; kern_t k; };
    std::vector<K> ks = {{"v_perm_b32", k_perm}, {"v_perm_b32(sgpr sel)", k_permsgpr}, {"v_pk_fma_f16", k_pkfma}, {"v_and_b32", k_and},
        {"v_lshrrev_b32", k_lshr}, {"v_bfi_b32", k_bfi}, {"v_and_or_b32", k_andor}, {"v_lshl_or_b32", k_lshlor},
        {"v_fma_f32", k_fmaf32}, {"v_fma_f16", k_fmaf16}, {"v_mad_u32_u24", k_madu24}, {"v_mul_u32_u24", k_mulu24},
        {"v_dot2_f32_f16", k_dot2}, {"v_dot2c_f32_f16", k_dot2c}, {"v_pk_add_f16", k_pkadd}, {"v_pk_mul_f16", k_pkmul},
        {"v_bfe_u32", k_bfe}, {"v_cndmask_b32", k_cndmask}, {"v_xor_b32", k_xor}, {"v_add_u32", k_add},
        {"v_alignbit_b32", k_alignbit}, {"v_lshl_add_u32", k_lshladd}, {"v_add3_u32", k_add3}, {"v_or3_b32", k_or3},
        {"v_pk_lshrrev_b16", k_pklshr16}, {"v_pk_mad_u16", k_pkmadu16}, {"v_mov_b32_dpp", k_movdpp},
        {"v_and_b32_sdwa", k_andsdwa}, {"v_sad_u8", k_sad},
        {"v_cndmask_b32_e64 (sgpr pair)", k_cndmask_s}, {"v_cndmask_b32_e32 (vcc set once)", k_cndmask_v}, {"v_cmp_lt_u32_e64 -> sgpr", k_cmp},
        {"v_mul_lo_u32", k_mullo}, {"v_mul_hi_u32", k_mulhi},
        {"v_permlane16_swap_b32", k_swap16}, {"v_permlane32_swap_b32", k_swap32}, {"v_add_f32_dpp row_ror:8", k_adddpp},
        {"v_add_f32 (dependent chain)", k_add_chain}, {"v_add_f32_dpp (dependent chain)", k_adddpp_chain}, {"v_permlane16_swap (dep. chain)", k_swap16_chain},
        {"ds_bpermute_b32(+wait)", k_bperm}, {"ds_read_b32 x16", k_ldsread}};
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int wps : {1, 2, 4}) {  // waves per SIMD
        printf("--- %d wave(s) per SIMD\n", wps);
        for (auto &k : ks) {
            dim3 grid(cus), block(256 * wps);
            if (256 * wps > 1024) { grid = dim3(cus * wps / 4 * 1); block = dim3(1024); }
            hipLaunchKernelGGL(k.k, grid, block, 0, 0, out, 1u);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(k.k, grid, block, 0, 0, out, 2u);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            double instr_per_wave = (double)ITER * 16;
            double waves_per_simd = (double)grid.x * block.x / 64 / (cus * 4);
            double ns_per_instr = ms * 1e6 / (instr_per_wave * waves_per_simd);
            printf("%-26s %8.3f ms  %6.2f ns/wave-instr/SIMD  (%.2f cycles @2.4GHz)  %.1f T lane-ops/s\n", k.name, ms,
                   ns_per_instr, ns_per_instr * 2.4, instr_per_wave * grid.x * block.x / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}
