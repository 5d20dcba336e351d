// Counter-based random streams for lens sampling and Bezier Newton starts.
//
// The reference draws `(double)rand() / RAND_MAX` from libc (sampling.h:31-43, bezier.h:183,236,239), which is
// neither reproducible across implementations nor usable from 10^5 concurrent lanes.  Every draw here is a pure
// function of (seed, pixel, sample, purpose, draw index):
//
//   k_pix = fin(fin(seed+G) + pixel + G)          once per pixel        G = 0x9E3779B97F4A7C15 (splitmix64)
//   k_smp = fin(k_pix + sample + G)               once per sample; this IS the lens stream's key (purpose 0)
//   key   = fin(k_smp + purpose + G)              other purposes (>= 1)
//   z_j   = fin(key + (j+1)*G)                    splitmix64 seeded with key; each 64-bit z_j gives TWO draws:
//   r31[2j] = z_j >> 33,  r31[2j+1] = (z_j >> 2) & 0x7fffffff       31 random bits each, like glibc rand()
//   u01   = (double)r31 / 2147483647.0            same value as sampling.h:32's rand()/RAND_MAX
//
// The division by RAND_MAX is evaluated as q0 = r*rc, q = fma(fma(-q0, D, r), rc, q0) with rc = fl(1/D): verified
// exhaustively (all 2^31 inputs: tests/test_rng_division.py) to
// equal the correctly rounded quotient, at 3 instructions instead of the ~15 of an IEEE divide.
//
// pixel = h*W + w on the GLOBAL image, so a render is independent of how rows are sharded over GPUs.
// purpose = 0 for the lens; (path_code << 16) | (object index + 1) for Bezier draws, where path_code is the ray's
// heap index in the reflection/refraction tree (primary 1, reflected 2k, refracted 2k+1).
#ifndef CGRT_RNG_HPP
#define CGRT_RNG_HPP
#include <stdint.h>
#include <cmath>

#if defined(__HIPCC__)
#define CGRT_HD __host__ __device__ __forceinline__
#else
#define CGRT_HD inline
#endif

namespace cgrt {

static constexpr uint64_t kGolden = 0x9E3779B97F4A7C15ULL;

CGRT_HD uint64_t fin64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
CGRT_HD uint64_t pixel_key(uint64_t seed, uint64_t pixel) { return fin64(fin64(seed + kGolden) + pixel + kGolden); }
CGRT_HD uint64_t sample_key(uint64_t k_pix, uint64_t sample) { return fin64(k_pix + sample + kGolden); }
CGRT_HD uint64_t purpose_key(uint64_t k_smp, uint64_t purpose) {
    return purpose ? fin64(k_smp + purpose + kGolden) : k_smp;
}
CGRT_HD uint64_t stream_key(uint64_t seed, uint64_t pixel, uint64_t sample, uint64_t purpose) {
    return purpose_key(sample_key(pixel_key(seed, pixel), sample), purpose);
}
// (double)r / 2147483647.0, correctly rounded, for integer 0 <= r < 2^31
CGRT_HD double div_rand_max(uint32_t r) {
    const double D = 2147483647.0, rc = 1.0 / 2147483647.0;
    const double x = (double)r;
    const double q0 = x * rc;
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_fma(__builtin_fma(-q0, D, x), rc, q0);
#else
    return std::fma(std::fma(-q0, D, x), rc, q0);
#endif
}

struct Stream {
    // Stateless form: draw n is a pure function of (key, n), so there is no carried "second half" flag (a carried
    // bool was lost across iterations of a divergent device loop); odd draws recompute the finaliser.
    uint64_t key;
    uint32_t n;  // draws consumed
    CGRT_HD explicit Stream(uint64_t k) : key(k), n(0) {}
    CGRT_HD uint32_t next31() {
        const uint64_t z = fin64(key + (uint64_t)(n / 2u + 1u) * kGolden);
        const uint32_t r = (n & 1u) ? (uint32_t)((z >> 2) & 0x7fffffffu) : (uint32_t)(z >> 33);
        n++;
        return r;
    }
    // two consecutive draws at an even position of the stream (the lens sampler's x, y): one finaliser
    CGRT_HD void pair(double &a, double &b) {
        const uint64_t z = fin64(key + (uint64_t)(n / 2u + 1u) * kGolden);
        n += 2u;
        a = div_rand_max((uint32_t)(z >> 33));
        b = div_rand_max((uint32_t)((z >> 2) & 0x7fffffffu));
    }
    CGRT_HD double u01() { return div_rand_max(next31()); }
};

}  // namespace cgrt
#endif
