// Checks ptl::fastdiv (csrc/pt_layout.h) against the hardware division: every divisor up to 5000 at the edge values
// and random (n, d) pairs below 2^30.  Exit code 0 = exact everywhere.
#include <cstdint>
#include <cstdio>
#include <random>

#include "pt_layout.h"

int main() {
    std::mt19937_64 rng(1);
    uint64_t bad = 0, tests = 0;
    const uint32_t top = (1u << 30) - 1;
    for (uint32_t d = 1; d < 5000; d++) {
        const ptl::FastDiv f = ptl::make_fastdiv(d);
        const uint32_t ns[] = {0u, 1u, d - 1, d, d + 1, 2 * d - 1, 2 * d, top, top - d, top - d + 1, (top / d) * d, (top / d) * d - 1};
        for (uint32_t n : ns) {
            if (n > top) continue;
            tests++;
            bad += ptl::fastdiv(n, f) != n / d;
        }
    }
    for (int k = 0; k < 4000000; k++) {
        uint32_t d = (uint32_t)(rng() % (1u << 30)) + 1;
        if (k & 1) d = d % 100000 + 1;
        const uint32_t n = (uint32_t)(rng() % (1u << 30));
        tests++;
        bad += ptl::fastdiv(n, ptl::make_fastdiv(d)) != n / d;
    }
    std::printf("tests %llu bad %llu\n", (unsigned long long)tests, (unsigned long long)bad);
    return bad != 0;
}
