// Host emulation of the plane-MFMA GEMV (guidedquant_amd/csrc/plane_core.h): same extraction masks, scale
// bytes, B-image addressing, piece splitting and Moebius coefficients as the HIP kernel; the MFMA itself is
// emulated as an exact dot product in double.  Validates the algorithm and index math on the CPU.
#include <vector>
#include <cstdint>
#include <cmath>
#include <cstring>
#include "plane_core.h"
using namespace gqp;

static inline float h2f(uint16_t h) {
    u32 e = (h >> 10) & 0x1F, m = h & 0x3FF;
    float v = e == 0 ? ldexpf((float)m, -24) : (e == 31 ? (m ? NAN : INFINITY) : ldexpf((float)(m | 0x400), (int)e - 25));
    return (h >> 15) ? -v : v;
}

template <int BITS>
static void run(const Geom &G, const uint16_t *x, const uint32_t *qw, const uint16_t *lut, uint32_t N, double *y) {
    // pieces + B image
    float mx = 0.f, X = 0.f;
    for (u32 e = 0; e < G.K; e++) mx = fmaxf(mx, fabsf(h2f(x[e])));
    int ex = 0;
    if (mx > 0.f) frexpf(mx, &ex);
    const float sc = ldexpf(1.0f, 15 - ex);
    std::vector<uint8_t> bimg((size_t)G.nchunks * 8 * 4 * 4 * 32, 0);
    // per hardware scale block (chunk, s, blk): exponent of the block maximum
    std::vector<int> beb((size_t)G.nchunks * 8 * 4, 15);
    std::vector<float> bmax((size_t)G.nchunks * 8 * 4, 0.f);
    for (u32 e = 0; e < G.K; e++) {
        u32 chunk, s, kb, v, B;
        locate_x(G, e, chunk, s, kb, v, B);
        float &mxb = bmax[(chunk * 8 + 0) * 4 + blk_of(kb, v)];
        mxb = fmaxf(mxb, fabsf(h2f(x[e])));
    }
    for (size_t i = 0; i < bmax.size(); i++)
        if (bmax[i] > 0.f) frexpf(bmax[i], &beb[i]);
    (void)sc;
    for (u32 e = 0; e < G.K; e++) {
        u32 chunk, s, kb, v, B;
        locate_x(G, e, chunk, s, kb, v, B);
        const int eb = beb[(chunk * 8 + 0) * 4 + blk_of(kb, v)];
        float r = h2f(x[e]) * ldexpf(1.0f, 15 - eb);
        X += h2f(x[e]);
        for (u32 p = 0; p < 4; p++) {
            uint8_t b = f32_to_bf8_rne(r);
            bimg[bimg_off(chunk, s, kb, p) + 4 * v + B] = b;
            r -= bf8_to_f32(b);
        }
    }
    ex = 15;  // results are in true units (the block scales undo the normalisation)
    constexpr int NP = (1 << BITS);
    for (u32 rg = 0; rg * 16 < N; rg++) {
        std::vector<double> T(16 * NP, 0.0);
        for (u32 chunk = 0; chunk < G.nchunks; chunk++)
            for (u32 l = 0; l < 64; l++) {
                const u32 r = l & 15, kb = l >> 4, row = rg * 16 + r;
                if (row >= N) continue;
                u32 W[BITS][8];
                for (int p = 0; p < BITS; p++)
                    for (u32 v = 0; v < 8; v++) {
                        u32 widx = 32 * chunk + 8 * kb + v;
                        W[p][v] = (8 * kb + v < G.tpw(chunk)) ? qw[((size_t)p * N + row) * G.wpr + widx] : 0u;
                    }
                for (int cm = 1; cm < NP; cm++) {
                    for (u32 s = 0; s < 8; s++) {
                        double scale = ldexp(1.0, scale_byte((int)s) - 127);
                        for (u32 v = 0; v < 8; v++) {
                            u32 pw = 0xFFFFFFFFu;
                            for (int i = 0; i < BITS; i++)
                                if (cm & (1 << i)) pw &= W[BITS - 1 - i][v];  // code bit i lives in plane BITS-1-i
                            u32 a = extract(pw, (int)s);
                            for (u32 B = 0; B < 4; B++) {
                                double av = bf8_to_f32((uint8_t)(a >> (8 * B)));
                                if (av == 0.0) continue;
                                // the hardware applies to element (kb, v) the scale supplied by lane group blk_of(kb, v)
                                const double sb = ldexp(1.0, beb[(chunk * 8 + 0) * 4 + blk_of(kb, v)] - 15);
                                for (u32 p = 0; p < 4; p++)
                                    T[r * NP + cm] += av * scale * sb * bf8_to_f32(bimg[bimg_off(chunk, s, kb, p) + 4 * v + B]);
                            }
                        }
                    }
                }
            }
        for (u32 r = 0; r < 16 && rg * 16 + r < N; r++) {
            float f[NP];
            for (int c = 0; c < NP; c++) f[c] = h2f(lut[(size_t)(rg * 16 + r) * NP + c]);
            moebius<BITS>(f);
            double acc = (double)f[0] * X;
            for (int cm = 1; cm < NP; cm++) acc += (double)f[cm] * T[r * NP + cm];
            y[rg * 16 + r] = acc * ldexp(1.0, ex - 15);
        }
    }
}

extern "C" int gq_emul_plane_gemv(const uint16_t *x, const uint32_t *qw, const uint16_t *lut, uint32_t N, uint32_t K,
                                  int bits, double *y) {
    if (K % 256) return -1;
    Geom G;
    G.init(K);
    switch (bits) {
        case 2: run<2>(G, x, qw, lut, N, y); break;
        case 3: run<3>(G, x, qw, lut, N, y); break;
        case 4: run<4>(G, x, qw, lut, N, y); break;
        default: return -2;
    }
    return 0;
}
