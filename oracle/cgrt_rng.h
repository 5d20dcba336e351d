/* TEST INFRASTRUCTURE (oracle side). Not part of the shipped product path.
 *
 * Counter-based random stream shared by the CPU oracle and the reference
 * harness.  It replaces libc rand() (reference: headers/sampling.h:31-43 draws
 * `(double)rand()/RAND_MAX`) with a keyed, stateless stream so that CPU and
 * GPU renders draw identical lens samples.  The product carries its own copy
 * of this integer recipe in cgraytracing_amd/csrc/cgrt_rng.hpp; tests check
 * the two agree bit for bit.
 *
 * splitmix64 seeded with K: z_j = fin(K + (j+1)*GOLDEN), j = 0,1,...  Each 64-bit output yields TWO 31-bit
 * draws: stream(K)[2j] = z_j >> 33, stream(K)[2j+1] = (z_j >> 2) & 0x7fffffff.
 * u01 = (double)r31 / 2147483647.0      (RAND_MAX of glibc)
 */
#ifndef CGRT_ORACLE_RNG_H
#define CGRT_ORACLE_RNG_H
#include <stdint.h>

#define CGRT_GOLDEN 0x9E3779B97F4A7C15ULL

static inline uint64_t cgrt_fin64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
/* key derivation: a = pixel index (h*W+w on the GLOBAL image), b = sample index,
 * c = purpose tag (0 = lens; Bezier: (path_code << 16) | (object_order + 1), always >= 1).
 *   k_pix = fin(fin(seed+G) + pixel + G)      once per pixel
 *   k_smp = fin(k_pix + sample + G)           once per sample  = the lens stream's key (purpose 0)
 *   key   = fin(k_smp + purpose + G)          purpose >= 1 */
static inline uint64_t cgrt_key(uint64_t seed, uint64_t a, uint64_t b, uint64_t c) {
    uint64_t k = cgrt_fin64(seed + CGRT_GOLDEN);
    k = cgrt_fin64(k + a + CGRT_GOLDEN);
    k = cgrt_fin64(k + b + CGRT_GOLDEN);
    if (c != 0) k = cgrt_fin64(k + c + CGRT_GOLDEN);
    return k;
}
static inline uint32_t cgrt_rand31(uint64_t key, uint32_t i) {
    uint64_t z = cgrt_fin64(key + (uint64_t)(i / 2u + 1u) * CGRT_GOLDEN);
    return (i & 1u) ? (uint32_t)((z >> 2) & 0x7fffffffu) : (uint32_t)(z >> 33);
}
#endif
