/* TEST INFRASTRUCTURE.  Exhaustive check of the division the product uses for u01 = r / RAND_MAX
 * (cgraytracing_amd/csrc/cgrt_rng.hpp, div_rand_max): for EVERY integer 0 <= r < 2^31,
 *     q0 = r * rc;  q = fma(fma(-q0, D, r), rc, q0)      with D = 2147483647.0, rc = fl(1/D)
 * must equal the correctly rounded IEEE quotient (double)r / D that the reference computes
 * (sampling.h:32, `(double)rand() / RAND_MAX`).  Prints the mismatch count; exit status 0 iff it is 0. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>

int main(void) {
    const double D = 2147483647.0, rc = 1.0 / 2147483647.0;
    uint64_t bad = 0, plain_bad = 0;
#pragma omp parallel for reduction(+ : bad, plain_bad)
    for (int64_t r = 0; r < 2147483648LL; r++) {
        const double x = (double)r, q = x / D, q0 = x * rc;
        const double q1 = fma(fma(-q0, D, x), rc, q0);
        bad += (q1 != q);
        plain_bad += (q0 != q);
    }
    printf("inputs 2147483648 mismatches %llu (plain reciprocal multiply would mismatch on %llu)\n",
           (unsigned long long)bad, (unsigned long long)plain_bad);
    return bad != 0;
}
