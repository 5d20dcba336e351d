// TEST PROGRAM (CPU): cgrt_dd::exp_dd (cgraytracing_amd/csrc/cgrt_ddexp.hpp, the exp of the device-side bump-map build)
// against this machine's libm over the arguments the height field can take: -3.3 * luma for RGB bytes (texture.h:28-35).
// Prints: triples tested, results that differ from libm's exp, largest difference in ulps, and how many of the differing
// results are the one NEARER to expl's (long double) value.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../cgraytracing_amd/csrc/cgrt_ddexp.hpp"

int main(int argc, char **argv) {
    const int stride = argc > 1 ? std::atoi(argv[1]) : 1;  // 1: all 2^24 triples
    long n = 0, differ = 0, dd_nearer = 0, max_ulp = 0;
    for (int r = 0; r < 256; r++)
        for (int g = 0; g < 256; g += stride)
            for (int b = 0; b < 256; b += stride) {
                const double luma = (0.299 * ((double)r / 256.0) + 0.587 * ((double)g / 256.0) + 0.114 * ((double)b / 256.0));
                const double x = -3.3 * luma;
                const double a = std::exp(x), c = cgrt_dd::exp_dd(x);
                n++;
                if (a != c) {
                    differ++;
                    long long ia, ic;
                    std::memcpy(&ia, &a, 8);
                    std::memcpy(&ic, &c, 8);
                    const long d = (long)std::llabs(ia - ic);
                    if (d > max_ulp) max_ulp = d;
                    const long double e = expl((long double)x);
                    if (fabsl(e - (long double)c) <= fabsl(e - (long double)a)) dd_nearer++;
                }
            }
    std::printf("%ld %ld %ld %ld\n", n, differ, max_ulp, dd_nearer);
    return 0;
}
