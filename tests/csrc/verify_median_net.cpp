// Zero-one-principle proof of the selection networks in tee_optical_flow_amd/csrc/median_net.h:
// a compare-exchange network outputs the median for every input iff it does so for every 0/1 input.
// Prints "OK" and exits 0 on success.
#include <cstdio>
#include <cstdint>
#include "median_net.h"
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
    std::printf("OK\n");
    return 0;
}
