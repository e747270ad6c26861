// Host emulation of the plane-MFMA GEMV (guidedquant_amd/csrc/plane_core.h): same nibble masks, scale bytes, B-image
// addressing, truncate-and-subtract piece splitting, A-tile swizzle and Moebius coefficients as the HIP kernel; the
// MFMA itself is emulated as an exact dot product in double.  Validates the algorithm and index math on the CPU.
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
// float -> fp16 bits for values that are exactly representable (results of exact fp16 subtractions) or need RNE
static inline uint16_t f2h(float f) {
    if (f == 0.f) return std::signbit(f) ? 0x8000 : 0;
    const uint16_t sg = f < 0 ? 0x8000 : 0;
    float a = fabsf(f);
    int ex;
    frexpf(a, &ex);            // a = m * 2^ex, m in [0.5, 1)
    int e = ex - 1 + 15;       // biased exponent of 1.x form
    if (e >= 31) return sg | 0x7C00;
    if (e <= 0) {              // subnormal: multiples of 2^-24
        const float q = nearbyintf(ldexpf(a, 24));
        return sg | (uint16_t)q;
    }
    float q = nearbyintf(ldexpf(a, 10 - (ex - 1)));  // 11-bit significand
    if (q >= 2048.f) {
        q = 1024.f;
        e++;
        if (e >= 31) return sg | 0x7C00;
    }
    return sg | (uint16_t)((e << 10) | ((int)q & 0x3FF));
}
static inline float fp4_val(u32 nib) {  // e2m1
    const u32 e = (nib >> 1) & 3, m = nib & 1;
    const float v = e == 0 ? 0.5f * (float)m : ldexpf(1.0f + 0.5f * (float)m, (int)e - 1);
    return (nib & 8) ? -v : v;
}

template <int BITS>
static void run(const Geom &G, const uint16_t *x, const uint32_t *qw, const uint16_t *lut, uint32_t N, double *y) {
    // global power-of-two scale, exact truncate-and-subtract pieces, B image
    float mx = 0.f, X = 0.f;
    for (u32 e = 0; e < G.K; e++) mx = fmaxf(mx, fabsf(h2f(x[e])));
    const int ksh = piece_shift(mx);
    std::vector<uint8_t> bimg((size_t)G.nchunks * 4096, 0);
    for (u32 e = 0; e < G.K; e++) {
        u32 chunk, b, h, k;
        locate_x4(G, e, chunk, b, h, k);
        X += h2f(x[e]);
        uint16_t rem = f2h(h2f(x[e]) * h2f(pow2_f16(ksh)));  // fp16 product (exact unless it underflows)
        for (u32 p = 0; p < 4; p++) {
            const uint16_t piece = rem & 0xFF00;
            bimg[bimg4_off(chunk, b, h, p) + k] = (uint8_t)(piece >> 8);
            rem = f2h(h2f(rem) - h2f(piece));
        }
    }
    constexpr int NP = (1 << BITS);
    for (u32 rg = 0; rg * 16 < N; rg++) {
        // the A tile as the direct-to-LDS loads deposit it (atile_src) and the MFMA lanes read it back (atile_unit)
        std::vector<double> T(16 * NP, 0.0);
        for (u32 chunk = 0; chunk < G.nchunks; chunk++) {
            std::vector<u32> tile((size_t)BITS * 512, 0);
            for (int p = 0; p < BITS; p++)
                for (u32 h = 0; h < 2; h++)
                    for (u32 lane = 0; lane < 64; lane++) {
                        u32 r, seg;
                        atile_src(h, lane, r, seg);
                        const u32 row = rg * 16 + r;
                        for (u32 q = 0; q < 4; q++) {
                            const u32 wd = 4 * seg + q;  // word of the chunk line
                            const bool ok = row < N && wd < G.tpw(chunk);
                            tile[(size_t)p * 512 + (64 * h + lane) * 4 + q] = ok ? qw[((size_t)p * N + row) * G.wpr + 32 * chunk + wd] : 0u;
                        }
                    }
            for (u32 l = 0; l < 64; l++) {
                const u32 r = l & 15, g = l >> 4, row = rg * 16 + r;
                if (row >= N) continue;
                u32 W[BITS][8];
                for (int p = 0; p < BITS; p++)
                    for (u32 v = 0; v < 8; v++) W[p][v] = tile[(size_t)p * 512 + atile_unit(r, 2 * g + v / 4) * 4 + v % 4];
                for (int cm = 1; cm < NP; cm++)
                    for (u32 b = 0; b < 4; b++)
                        for (u32 h = 0; h < 2; h++) {
                            const double scale = ldexp(1.0, scale_byte4((int)b) - 127);
                            for (u32 v = 0; v < 4; v++) {
                                u32 pw = 0xFFFFFFFFu;
                                for (int i = 0; i < BITS; i++)
                                    if (cm & (1 << i)) pw &= W[BITS - 1 - i][4 * h + v];  // code bit i lives in plane BITS-1-i
                                const u32 a = extract4(pw, (int)b);
                                for (u32 i = 0; i < 8; i++) {
                                    const double av = fp4_val((a >> (4 * i)) & 15u);
                                    if (av == 0.0) continue;
                                    const u32 k = 32 * g + 8 * v + i;
                                    for (u32 p = 0; p < 4; p++)
                                        T[r * NP + cm] += av * scale * bf8_to_f32(bimg[bimg4_off(chunk, b, h, p) + k]);
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
            for (int cm = 1; cm < NP; cm++) acc += (double)f[cm] * T[r * NP + cm] * ldexp(1.0, -ksh);
            y[rg * 16 + r] = acc;
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

// A-tile swizzle properties: (1) atile_src and atile_unit are inverse bijections over the 128 units of a tile;
// (2) every instruction fetches 8 whole 128-byte lines; (3) the 16 lanes of a ds_read_b128 pass (rows 0..15, one
// segment) hit 16 distinct 16-byte bank groups.  Returns 0 when all hold, else a code naming the violated property.
extern "C" int gq_emul_atile_check() {
    bool seen[128] = {};
    for (u32 h = 0; h < 2; h++) {
        int rows_seen[16] = {};
        for (u32 lane = 0; lane < 64; lane++) {
            u32 r, seg;
            atile_src(h, lane, r, seg);
            if (r >= 16 || seg >= 8 || (r >> 3) != h) return 1;
            const u32 u = atile_unit(r, seg);
            if (u != 64 * h + lane || seen[u]) return 2;
            seen[u] = true;
            rows_seen[r] |= 1 << seg;
        }
        for (u32 r = 8 * h; r < 8 * h + 8; r++)
            if (rows_seen[r] != 0xFF) return 3;  // the 8 segments of each of the 8 rows: whole lines
    }
    for (u32 seg = 0; seg < 8; seg++) {
        int groups = 0;
        for (u32 r = 0; r < 16; r++) groups |= 1 << (atile_unit(r, seg) & 15u);
        if (groups != 0xFFFF) return 4;
    }
    // FP4 masks: the four nibble bits cover every plane bit exactly once, pattern values times the scale are 1
    for (u32 bit = 0; bit < 32; bit++) {
        int hits = 0;
        for (int b = 0; b < 4; b++) {
            const u32 a = extract4(1u << bit, b);
            if (!a) continue;
            hits++;
            u32 nib = 0;
            for (u32 i = 0; i < 8; i++)
                if ((a >> (4 * i)) & 15u) nib = (a >> (4 * i)) & 15u;
            if (fp4_val(nib) * ldexpf(1.0f, scale_byte4(b) - 127) != 1.0f) return 5;
        }
        if (hits != 1) return 6;
    }
    return 0;
}
