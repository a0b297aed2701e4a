// Zero-one-principle proof of the selection networks in tee_optical_flow_amd/csrc/median_net.h:
// a network of monotone operations (min, max, med3) outputs the median for every input iff it does so for every 0/1
// input.  Prints "OK" and exits 0 on success.
#include <cstdio>
#include <cstdint>
#include "median_net.h"

// 64 binary inputs at a time: min = AND, max = OR
struct B64 { uint64_t v; };
inline B64 tf_min(B64 a, B64 b) { return B64{a.v & b.v}; }
inline B64 tf_max(B64 a, B64 b) { return B64{a.v | b.v}; }

int main()
{
    for (uint32_t m = 0; m < (1u << 9); ++m) {
        int p[9]; int ones = 0;
        for (int i = 0; i < 9; ++i) { p[i] = (m >> i) & 1; ones += p[i]; }
        if (tf_median9(p) != (ones >= 5)) { std::printf("median9 FAIL %u\n", m); return 1; }
    }
    for (uint32_t m = 0; m < (1u << 25); ++m) {
        int p[25];
        for (int i = 0; i < 25; ++i) p[i] = (m >> i) & 1;
        if (tf_median25(p) != (__builtin_popcount(m) >= 13)) { std::printf("median25 FAIL %u\n", m); return 1; }
    }
    // building blocks of the four-output form, on every binary input that respects their preconditions
    for (uint32_t m = 0; m < 32; ++m) {
        int v[5], s[5], ones = __builtin_popcount(m);
        for (int i = 0; i < 5; ++i) v[i] = (m >> i) & 1;
        tf_sort5(v, s);
        for (int i = 0; i < 5; ++i) if (s[i] != (i >= 5 - ones)) { std::printf("sort5 FAIL %u\n", m); return 1; }
    }
    for (int ka = 0; ka <= 5; ++ka) for (int kb = 0; kb <= 5; ++kb) {
        int a[5], b[5], z[10];
        for (int i = 0; i < 5; ++i) { a[i] = i >= 5 - ka; b[i] = i >= 5 - kb; }
        tf_merge5(a, b, z);
        for (int i = 0; i < 10; ++i) if (z[i] != (i >= 10 - ka - kb)) { std::printf("merge5 FAIL %d %d\n", ka, kb); return 1; }
    }
    for (int ka = 0; ka <= 10; ++ka) for (int kb = 0; kb <= 10; ++kb) {
        int a[10], b[10], mm[6];
        for (int i = 0; i < 10; ++i) { a[i] = i >= 10 - ka; b[i] = i >= 10 - kb; }
        tf_middle6(a, b, mm);
        for (int i = 0; i < 6; ++i) if (mm[i] != (7 + i >= 20 - ka - kb)) { std::printf("middle6 FAIL %d %d\n", ka, kb); return 1; }
    }
    // tf_median25_row4: every output over all 2^25 binary values of ITS window, the other 15 inputs all 0 and all 1
    static const uint64_t PAT[6] = {0xAAAAAAAAAAAAAAAAull, 0xCCCCCCCCCCCCCCCCull, 0xF0F0F0F0F0F0F0F0ull,
                                    0xFF00FF00FF00FF00ull, 0xFFFF0000FFFF0000ull, 0xFFFFFFFF00000000ull};
    for (int o = 0; o < 4; ++o)
        for (int fill = 0; fill < 2; ++fill)
            for (uint32_t w = 0; w < (1u << 19); ++w) {
                B64 col[8][5], p[25], out[4];
                for (int j = 0; j < 8; ++j) for (int r = 0; r < 5; ++r) col[j][r] = B64{fill ? ~0ull : 0ull};
                for (int i = 0; i < 25; ++i) {
                    const uint64_t bits = i < 6 ? PAT[i] : (((w >> (i - 6)) & 1u) ? ~0ull : 0ull);
                    p[i] = B64{bits};
                    col[o + i / 5][i % 5] = B64{bits};
                }
                tf_median25_row4(col, out);
                const B64 ref = tf_median25(p);          // proven above
                if (out[o].v != ref.v) { std::printf("median25_row4 FAIL output %d fill %d word %u\n", o, fill, w); return 1; }
            }
    std::printf("OK\n");
    return 0;
}
