// Device-side Vec3 arithmetic in the reference's operation order (vec3.h:11-119).  Part of libcgrt.so (cgrt_hip.hip).
#ifndef CGRT_DEVICE_MATH_HPP
#define CGRT_DEVICE_MATH_HPP
#include <hip/hip_runtime.h>

#include "cgrt_rng.hpp"
#include "cgrt_types.h"

using namespace cgrt;

#ifdef CGRT_UTIL
__device__ unsigned long long g_util[64];
#define UTILP(k, pred) do { const unsigned long long m_ = __ballot(pred); const unsigned long long a_ = __ballot(true); if ((int)(threadIdx.x & 63) == __ffsll((long long)a_) - 1) { atomicAdd(&g_util[2*(k)], 1ull); atomicAdd(&g_util[2*(k)+1], (unsigned long long)__popcll(m_)); } } while (0)
#else
#define UTILP(k, pred) do {} while (0)
#endif
#define UTIL(k) UTILP(k, true)

// =====================================================================================================
// device math: the reference's Vec3 (vec3.h:11-119), same operation order, no contraction
// =====================================================================================================
struct V3 {
    double x, y, z;
};
__device__ __forceinline__ V3 mk(double x, double y, double z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
__device__ __forceinline__ V3 operator*(V3 a, double f) { return mk(a.x * f, a.y * f, a.z * f); }
__device__ __forceinline__ V3 mulv(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// The correctly rounded fp64 square root without the range scaling.  hipcc expands sqrt(x) into v_rsq_f64 and three fma
// corrections -- which is what stands below, instruction for instruction -- wrapped in a scaling by 2^256 for x < 2^-767 (so that
// the corrections do not underflow) and a final select for x = 0 / +inf: 20 VALU instructions, 7 of them range handling.  When no
// active lane of the wave has such an x (nor a negative one or a NaN) the wrapping does nothing and is left out; otherwise the
// whole wave takes the library form.  Same bits either way; a sphere-scene ray takes 4-5 square roots.
__device__ __forceinline__ double sqrt_cr(double x) {
    if (__ballot(!(x >= 0x1p-767 && x <= 0x1p1000)) != 0ull) return sqrt(x);
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = y * 0.5;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    double d = __builtin_fma(-g, g, x);
    h = __builtin_fma(h, r, h);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    return g;
}

// vec3.h:36-44: len = sqrt(x*x + y*y + z*z); if (len > 0) every component *= 1 / len.
// Both correctly rounded operations in their unwrapped form when no active lane needs the wrapping: sqrt_cr's condition on the sum
// of squares puts len in [2^-384, 2^500], where hipcc's expansion of 1.0 / len (v_div_scale, v_rcp_f64, two refinements, one
// residual correction, v_div_fmas, v_div_fixup: 11 instructions) scales nothing and fixes nothing up -- what is left is the 7
// instructions below, the same bits.
__device__ __forceinline__ V3 normalized(V3 a) {
    const double s2 = a.x * a.x + a.y * a.y + a.z * a.z;
    if (__ballot(!(s2 >= 0x1p-767 && s2 <= 0x1p1000)) != 0ull) {
        const double len = sqrt(s2);
        if (len > 0) {
            const double r = 1 / len;
            a.x *= r;
            a.y *= r;
            a.z *= r;
        }
        return a;
    }
    const double len = sqrt_cr(s2);
    double r = __builtin_amdgcn_rcp(len);
    r = __builtin_fma(__builtin_fma(-len, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-len, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-len, r, 1.0), r, r);  // q = fma(fma(-len, q0, 1), r, q0) with q0 = 1.0 * r
    a.x *= r;
    a.y *= r;
    a.z *= r;
    return a;
}
// vec3.h:95-97, Sarrus with the reference's association
__device__ __forceinline__ double det3(V3 a, V3 b, V3 c) {
    return (a.x * b.y * c.z + b.x * c.y * a.z + c.x * a.y * b.z - a.x * c.y * b.z - b.x * a.y * c.z -
            c.x * b.y * a.z);
}
__device__ __forceinline__ V3 ld3(const double *p) { return mk(p[0], p[1], p[2]); }

// A record of the uploaded scene read at a WAVE-UNIFORM address (a tree / height-field / Bezier header picked by a
// readfirstlane'd index).  The scene is immutable while a kernel runs, so the load goes through the constant address space:
// the compiler then issues scalar loads (s_load_dwordx4/x8 into SGPRs) whatever stores surround it.  Without this it falls
// back to per-lane vector loads of the same address as soon as the kernel stores to global memory before the load on some
// path -- measured on the opaque-mesh kernel: +16 % vector-memory instructions, the 40-byte TreeRec and 64-byte HFieldRec
// held in VGPRs, frame 49.8 -> 58.5 ms.
template <class T>
__device__ __forceinline__ T load_uniform(const T *p) {
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(sizeof(T) % 4 == 0 && alignof(T) >= 4, "dword-copyable record");
    typedef const uint32_t __attribute__((address_space(4))) *ConstWords;
    ConstWords q = (ConstWords)p;
    T out;
    uint32_t *o = reinterpret_cast<uint32_t *>(&out);
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 4; i++) o[i] = q[i];
    return out;
#else
    return *p;
#endif
}

#endif
