// Host emulation of the gfx950 AP-GEMV lane program (guidedquant_amd/csrc/ap_core.h).
// Built with g++ by tests/test_lane_program_cpu.py; v_perm_b32 / v_bfi_b32 / v_pk_fma_f16 are emulated
// bit-exactly, so the decode logic and the index math can be checked against the oracle without a GPU.
#include <vector>
#include <cstdint>
#include <cstring>
#include "ap_core.h"

using namespace gq;

template <int BITS>
static void run_row(const RowGeom &G, const uint32_t *qw, uint32_t N, uint32_t n, const uint16_t *lutrow,
                    const std::vector<uint16_t> &xlds, std::vector<uint16_t> &sv) {
    u32 raw[(1 << BITS) / 2];
    for (int i = 0; i < (1 << BITS) / 2; i++) raw[i] = (u32)lutrow[2 * i] | ((u32)lutrow[2 * i + 1] << 16);
    LutPools<BITS> L;
    L.build(raw);
    for (u32 q = 0; q < G.Q; q++) {
        u32 P[BITS][4];
        for (int p = 0; p < BITS; p++)
            for (int v = 0; v < 4; v++) P[p][v] = qw[((size_t)p * N + n) * G.wpr + 4 * q + v];
        XRegs x;
        for (u32 c = 0; c < 4; c++)
            for (u32 jj = 0; jj < 4; jj++)
                for (u32 r = 0; r < 4; r++) {
                    // register r of slot (c,jj): halves at slot*8 + 2r, 2r+1
                    u32 base = (((c * 4 + jj) * G.Q + q) << 3) + 2 * r;
                    x.r[c][jj][r] = (u32)xlds[base] | ((u32)xlds[base + 1] << 16);
                }
        u32 s01, s23;
        Item<BITS>::run(P, L, x, s01, s23);
        u32 chunk, t0, tpw;
        G.quad(q, chunk, t0, tpw);
        sv[chunk * 32 + t0 + 0] = s01 & 0xFFFF;
        sv[chunk * 32 + t0 + 1] = s01 >> 16;
        sv[chunk * 32 + t0 + 2] = s23 & 0xFFFF;
        sv[chunk * 32 + t0 + 3] = s23 >> 16;
    }
}

extern "C" int gq_emul_ap_gemv(const uint16_t *x, const uint32_t *qw, const uint16_t *lut, uint32_t N, uint32_t K,
                               int bits, uint16_t *out) {
    if (K % 128) return -1;
    RowGeom G;
    G.init(K);
    // stage x exactly as the kernel does: 16-byte groups scattered to xlds_pos
    std::vector<uint16_t> xlds(K);
    for (u32 g = 0; g < K / 8; g++) {
        u32 q, v, c;
        xgroup(G, g, q, v, c);
        for (u32 j = 0; j < 8; j++) {
            if (G.xindex(q, v, c, j) != 8 * g + j) return -2;  // index math self-check
            xlds[xlds_pos(G.Q, q, v, c, j)] = x[8 * g + j];
        }
    }
    std::vector<uint16_t> sv(G.nchunks * 32);
    for (u32 n = 0; n < N; n++) {
        std::fill(sv.begin(), sv.end(), 0);
        const uint16_t *lr = lut + (size_t)n * (1u << bits);
        switch (bits) {
            case 2: run_row<2>(G, qw, N, n, lr, xlds, sv); break;
            case 3: run_row<3>(G, qw, N, n, lr, xlds, sv); break;
            case 4: run_row<4>(G, qw, N, n, lr, xlds, sv); break;
            default: return -3;
        }
        uint16_t part[32];
        for (u32 t = 0; t < 32; t++) {
            uint16_t p = 0;
            for (u32 i = 0; i < G.nchunks; i++) {
                if (i == G.nfull && t >= G.eff) break;
                p = h_add(p, sv[i * 32 + t]);
            }
            part[t] = p;
        }
        for (u32 sh = 16; sh >= 1; sh >>= 1)
            for (u32 t = 0; t < sh; t++) part[t] = h_add(part[t], part[t + sh]);
        out[n] = part[0];
    }
    return 0;
}
