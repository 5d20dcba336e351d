// Counter-based random streams for lens sampling and Bezier Newton starts.
//
// The reference draws `(double)rand() / RAND_MAX` from libc (sampling.h:31-43, bezier.h:183,236,239), which is
// neither reproducible across implementations nor usable from 10^5 concurrent lanes.  Every draw here is a pure
// function of (seed, pixel, sample, purpose, draw index):
//
//   key  = fin(fin(fin(fin(seed+G)+pixel+G)+sample+G)+purpose+G)        G = 0x9E3779B97F4A7C15 (splitmix64)
//   r31  = fin(key + (i+1)*G) >> 33                                       31 random bits, like glibc rand()
//   u01  = (double)r31 / 2147483647.0                                     same expression as sampling.h:32
//
// pixel = h*W + w on the GLOBAL image, so a render is independent of how rows are sharded over GPUs.
// purpose = 0 for the lens; (path_code << 16) | (object index + 1) for Bezier draws, where path_code is the ray's
// heap index in the reflection/refraction tree (primary 1, reflected 2k, refracted 2k+1).
#ifndef CGRT_RNG_HPP
#define CGRT_RNG_HPP
#include <stdint.h>

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
CGRT_HD uint64_t stream_key(uint64_t seed, uint64_t pixel, uint64_t sample, uint64_t purpose) {
    uint64_t k = fin64(seed + kGolden);
    k = fin64(k + pixel + kGolden);
    k = fin64(k + sample + kGolden);
    k = fin64(k + purpose + kGolden);
    return k;
}
CGRT_HD uint32_t rand31(uint64_t key, uint32_t i) { return (uint32_t)(fin64(key + (uint64_t)(i + 1u) * kGolden) >> 33); }

struct Stream {
    uint64_t key;
    uint32_t i;
    CGRT_HD double u01() { return (double)rand31(key, i++) / 2147483647.0; }
};

}  // namespace cgrt
#endif
