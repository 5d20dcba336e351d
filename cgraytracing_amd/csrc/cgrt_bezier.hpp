// Bezier surface of revolution on the device (Newton solves dealt over the wave).  Part of libcgrt.so (cgrt_hip.hip).
#ifndef CGRT_BEZIER_HPP
#define CGRT_BEZIER_HPP
#include "cgrt_device_math.hpp"
#include "cgrt_grid.hpp"

// =====================================================================================================
// Bezier surface of revolution: Bezier::intersect and helpers (bezier.h:30-40,72-290)
// =====================================================================================================
// The structure is the reference's: bounding-box reject; 10 Newton solves on (t,u,theta) from random starts
// (u ~ U(0,1), t = 20 + 10 U(0,1), theta = atan(px/pz)); stale inverse reuse and a random jitter when the
// Jacobian is singular; nearest accepted root wins; the cap-disc override that ignores the Newton flag.
// Random draws come from the ray's keyed stream (cgrt_rng.hpp) in the reference's draw order.
// pow(x, k) for the integer k <= 5 that occur is evaluated as a double-double product rounded once, which is the
// correctly rounded power in all but near-tie cases -- the closest device analogue of libm's pow.
__device__ const double kCni[7][7] = {{1, 0, 0, 0, 0, 0, 0}, {1, 1, 0, 0, 0, 0, 0}, {1, 2, 1, 0, 0, 0, 0},
                                      {1, 3, 3, 1, 0, 0, 0}, {1, 4, 6, 4, 1, 0, 0}, {1, 5, 10, 10, 5, 1, 0},
                                      {1, 6, 15, 20, 15, 6, 1}};  // bezier.h:17-23

__device__ __forceinline__ double ipow_dd(double x, int k) {
    if (k <= 0) return 1.0;
    if (k == 1) return x;
    double hi = x * x;
    if (k == 2) return hi;
    double lo = fma(x, x, -hi);
    for (int j = 2; j < k; j++) {
        const double p = hi * x;
        const double e = fma(hi, x, -p);
        const double l = lo * x + e;
        const double s = p + l;
        lo = l - (s - p);
        hi = s;
    }
    return hi;
}
// bezier.h:30-40
__device__ __forceinline__ double bern(int n, int i, double t) {
    if (i > n || i < 0) return 0;
    return kCni[n][i] * ipow_dd(1 - t, n - i) * ipow_dd(t, i);
}
__device__ __forceinline__ double dbern(int n, int i, double t) {
    return bern(n - 1, i - 1, t) * (double)i - bern(n - 1, i, t) * (double)(n - i);
}
// valueP / gradP (bezier.h:127-142).  With the number of control points a compile-time constant the loops unroll, the
// integer powers of u and 1-u -- which the reference recomputes inside every Bernstein term -- are computed once, and
// the binomials fold; the arithmetic per term is unchanged (same products in the same order), so are the values.
template <int N>
__device__ __forceinline__ V3 bez_value_n(const BezierRec &b, double u) {
    V3 r = mk(0, 0, 0);
#pragma unroll
    for (int i = 0; i < N; i++) r = r + ld3(b.cp[i]) * bern(N - 1, i, u);
    return r;
}
template <int N>
__device__ __forceinline__ V3 bez_grad_n(const BezierRec &b, double u) {
    V3 r = mk(0, 0, 0);
#pragma unroll
    for (int i = 0; i < N; i++) r = r + ld3(b.cp[i]) * dbern(N - 1, i, u);
    return r;
}
__device__ __forceinline__ V3 bez_value(const BezierRec &b, double u) {
    const int n = b.ncp;
    if (n == 4) return bez_value_n<4>(b, u);  // the reference's vase (main.cpp:371-376)
    if (n == 3) return bez_value_n<3>(b, u);
    V3 r = mk(0, 0, 0);
    for (int i = 0; i < n; i++) r = r + ld3(b.cp[i]) * bern(n - 1, i, u);
    return r;
}
__device__ __forceinline__ V3 bez_grad(const BezierRec &b, double u) {
    const int n = b.ncp;
    if (n == 4) return bez_grad_n<4>(b, u);
    if (n == 3) return bez_grad_n<3>(b, u);
    V3 r = mk(0, 0, 0);
    for (int i = 0; i < n; i++) r = r + ld3(b.cp[i]) * dbern(n - 1, i, u);
    return r;
}
// bezier.h:72-126: any face crossing inside the grown rectangle with 0 < t < 1e10
__device__ __forceinline__ bool bez_box(const BezierRec &b, V3 o, V3 d) {
    const double xmin = b.box[0], xmax = b.box[1], ymin = b.box[2], ymax = b.box[3], zmin = b.box[4], zmax = b.box[5];
    const double e = 1e-4;
    bool flag = false;
    double t;
    V3 p;
    t = (xmax - o.x) / d.x; p = o + d * t;
    flag |= (t > 0 && p.y >= ymin - e && p.y <= ymax + e && p.z >= zmin - e && p.z <= zmax + e && t < kInf);
    t = (xmin - o.x) / d.x; p = o + d * t;
    flag |= (t > 0 && p.y >= ymin - e && p.y <= ymax + e && p.z >= zmin - e && p.z <= zmax + e && t < kInf);
    t = (ymax - o.y) / d.y; p = o + d * t;
    flag |= (t > 0 && p.x >= xmin - e && p.x <= xmax + e && p.z >= zmin - e && p.z <= zmax + e && t < kInf);
    t = (ymin - o.y) / d.y; p = o + d * t;
    flag |= (t > 0 && p.x >= xmin - e && p.x <= xmax + e && p.z >= zmin - e && p.z <= zmax + e && t < kInf);
    t = (zmax - o.z) / d.z; p = o + d * t;
    flag |= (t > 0 && p.x >= xmin - e && p.x <= xmax + e && p.y >= ymin - e && p.y <= ymax + e && t < kInf);
    t = (zmin - o.z) / d.z; p = o + d * t;
    flag |= (t > 0 && p.x >= xmin - e && p.x <= xmax + e && p.y >= ymin - e && p.y <= ymax + e && t < kInf);
    return flag;
}

// Can the ray come within the acceptance radius of the surface at all?  Every solve of Bezier::intersect ends in the test
// |F(t, u, theta)| < 1e-4 with t > 0 and 0 <= u <= 1 (bezier.h:257): the ray point X(t) within 1e-4 of the surface point of
// parameter u -- hence within the heights and the (grown) radius of the piece of [0, 1] that holds u.  If for no piece the ray
// is, at some t > 0, inside that piece's cylinder and between its heights, no solve can be accepted, whatever its random
// start and however its iteration wanders: the ten solves (typically 100 iterations each: there is no root to converge to)
// are skipped and the result is the reference's "false".  Measured on the reference's vase, rays inside the bounding box that
// miss the surface are a third of the Newton work of a C5 frame.  Every doubtful comparison (NaN, a ray parallel to the axis
// or to the slabs) answers "maybe".  o, d relative to nothing: `pos` is the object's position.
__device__ __forceinline__ bool bez_shell_maybe(const BezSlabRec *__restrict__ slabs, V3 pos, V3 o, V3 d) {
    const double ox = o.x - pos.x, oz = o.z - pos.z, oy = o.y - pos.y;
    const double a = d.x * d.x + d.z * d.z, b = ox * d.x + oz * d.z, c0 = ox * ox + oz * oz;
    if (!(a > 1e-12) || !(fabs(d.y) > 1e-9)) return true;  // (nearly) parallel to the axis or to the slabs: not worth the cases
    const double ia = 1.0 / a, iy = 1.0 / d.y;
    bool maybe = false;
    for (int k = 0; k < kBezSlabs; k++) {
        const BezSlabRec sl = load_uniform(slabs + k);
        const double disc = b * b - a * (c0 - sl.r2);
        // inside the cylinder for t in [ta, tb]
        const double sq = sqrt(fmax(disc, 0.0));
        const double ta = (-b - sq) * ia, tb = (-b + sq) * ia;
        // between the heights for t in [tc, td]
        const double t1 = (sl.ylo - oy) * iy, t2 = (sl.yhi - oy) * iy;
        const double tc = fmin(t1, t2), td = fmax(t1, t2);
        const double lo = fmax(fmax(ta, tc), 0.0), hi = fmin(tb, td);
        // certainly outside: no real crossing of the cylinder, or empty overlap; anything else (NaN included) keeps the ray
        const bool out = (disc < 0.0) || (lo > hi);
        maybe = maybe || !out;
    }
    return maybe;
}

// One Newton iteration of newtonMethod (bezier.h:170-199) on the state (res, inverse columns, P, sin, cos, F).
// Returns false when the Jacobian is singular (the caller decides what the reference's jitter branch means for it).
struct NewtonState {
    V3 res;         // (t, u, theta)
    V3 iD, iE, iF;  // inverse columns, stale across singular steps
    V3 P;           // valueP(u)
    double sn, cs;  // sin(theta), cos(theta)
    V3 fv;          // F(res)
    int counter;
};
// The reference compares |F| = sqrt(x*x + y*y + z*z) with 1e-6 (carry on, bezier.h:171) and 1e-4 (accept, bezier.h:257).  With a
// correctly rounded square root (the reference's sqrtsd) those comparisons are comparisons of the sum of squares with the largest
// double whose root is <= 1e-6, resp. the smallest whose root is >= 1e-4 -- the same decisions without the root:
//   sqrt(s) > 1e-6  <=>  s > 0x1.19799812dea11p-40        sqrt(s) < 1e-4  <=>  s < 0x1.5798ee2308c3ap-27
// (tools/sqrt_thresholds.py derives and checks the two constants; NaN fails both forms, +inf passes the first in both).
__device__ __forceinline__ double sumsq3(V3 v) { return v.x * v.x + v.y * v.y + v.z * v.z; }
__device__ __forceinline__ bool newton_unconverged(V3 fv) { return sumsq3(fv) > 0x1.19799812dea11p-40; }

__device__ __forceinline__ void newton_init(const BezierRec &b, V3 pos, V3 o, V3 d, double u0, double t0, NewtonState &st) {
    V3 pt = o + d * t0;
    pt = pt - pos;
    const double th0 = (pt.z < 0) ? 3.14159265 + atan(pt.x / pt.z) : atan(pt.x / pt.z);  // bezier.h:243-247
    st.res = mk(t0, u0, th0);
    st.iD = mk(0, 0, 0);
    st.iE = mk(0, 0, 0);
    st.iF = mk(0, 0, 0);
    st.P = bez_value(b, st.res.y);
    sincos(st.res.z, &st.sn, &st.cs);
    st.fv = ((o + d * st.res.x) - pos) - mk(st.P.z * st.sn, st.P.y, st.P.z * st.cs);  // funcValue, bezier.h:144-149
    st.counter = 0;
}
// the Jacobian part: returns det and fills the inverse when regular
__device__ __forceinline__ bool newton_jacobian(const BezierRec &b, V3 d, NewtonState &st) {
    const V3 dP = bez_grad(b, st.res.y);
    const V3 A = d;  // gradValue, bezier.h:150-162
    const V3 B = mk(-st.sn * dP.z, -dP.y, -st.cs * dP.z);
    const V3 C = mk(-st.cs * st.P.z, 0, st.sn * st.P.z);
    const double dt = det3(A, B, C);  // inv(), vec3.h:103-119
    if (dt < 1e-4 && dt > -1e-4) return false;
    // Nine quotients by the same determinant.  hipcc expands every fp64 `x / dt` into: r0 = v_rcp_f64(dt), two
    // Newton refinements of r, q0 = x*r, e = fma(-dt, q0, x), q = fma(e, r, q0) (plus operand scaling that is the
    // identity unless an exponent is extreme).  Sharing the refined reciprocal and keeping the per-quotient part
    // gives the same correctly rounded quotients for 1e-4 <= |dt| and ordinary numerators at 3 instead of ~13
    // instructions each; a numerator that has already overflowed (a diverged solve, rejected either way) is handed
    // to a true division.
    double rc = __builtin_amdgcn_rcp(dt);
    rc = fma(fma(-dt, rc, 1.0), rc, rc);
    rc = fma(fma(-dt, rc, 1.0), rc, rc);
    const double n0 = B.y * C.z - B.z * C.y, n1 = C.y * A.z - C.z * A.y, n2 = A.y * B.z - A.z * B.y;
    const double n3 = C.x * B.z - C.z * B.x, n4 = A.x * C.z - A.z * C.x, n5 = B.x * A.z - B.z * A.x;
    const double n6 = B.x * C.y - C.x * B.y, n7 = C.x * A.y - C.y * A.x, n8 = A.x * B.y - A.y * B.x;
    auto fast = [&](double x) {
        const double q0 = x * rc;
        return fma(fma(-dt, q0, x), rc, q0);
    };
    // largest magnitude among the numerators (fmax drops a NaN operand: a NaN numerator gives NaN through either form)
    const bool ordinary = fmax(fmax(fmax(fabs(n0), fabs(n1)), fmax(fabs(n2), fabs(n3))),
                               fmax(fmax(fabs(n4), fabs(n5)), fmax(fmax(fabs(n6), fabs(n7)), fabs(n8)))) < 1e300;
    // The true divisions sit behind a WAVE-UNIFORM branch: written as a per-quotient select, the compiler evaluated
    // both forms for every quotient (nine ~13-instruction divisions per Newton iteration, a quarter of the loop).
    if (__ballot(!ordinary) == 0ull) {
        st.iD = mk(fast(n0), fast(n1), fast(n2));
        st.iE = mk(fast(n3), fast(n4), fast(n5));
        st.iF = mk(fast(n6), fast(n7), fast(n8));
    } else {
        auto quot = [&](double x) { return (fabs(x) < 1e300) ? fast(x) : x / dt; };
        st.iD = mk(quot(n0), quot(n1), quot(n2));
        st.iE = mk(quot(n3), quot(n4), quot(n5));
        st.iF = mk(quot(n6), quot(n7), quot(n8));
    }
    return true;
}
__device__ __forceinline__ void newton_step(const BezierRec &b, V3 pos, V3 o, V3 d, NewtonState &st) {
    const V3 step = (st.iD * st.fv.x + st.iE * st.fv.y) + st.iF * st.fv.z;  // matrixVectorProduct, vec3.h:99-101
    st.res = st.res - step;
    st.P = bez_value(b, st.res.y);
    sincos(st.res.z, &st.sn, &st.cs);  // one shared argument reduction; same values as sin() and cos()
    st.fv = ((o + d * st.res.x) - pos) - mk(st.P.z * st.sn, st.P.y, st.P.z * st.cs);
}
__device__ __forceinline__ bool newton_accept(const NewtonState &st) {  // bezier.h:257
    return sumsq3(st.fv) < 0x1.5798ee2308c3ap-27 && st.res.x > 0 && st.res.y <= 1 && st.res.y >= 0;
}
__device__ __forceinline__ V3 bez_normal(const BezierRec &b, double u, double sn, double cs) {  // bezier.h:215-224
    const V3 rp = normalized(bez_grad(b, u));
    return mk(rp.y * sn, -rp.z, rp.y * cs);
}

// The ten solves of Bezier::intersect (bezier.h:233-271), one lane, strictly sequential draws: the reference's
// exact semantics including the jitter branch.  Used by the function-level fallback below.
__device__ bool bezier_solve_serial(const BezierRec &b, V3 pos, V3 o, V3 d, Stream &rs, double &len, V3 &n) {
    bool flag = false;
    len = kInf;
    for (int k = 0; k < 10; k++) {  // num_of_samples_newton, bezier.h:27
        const double u0 = rs.u01();
        const double t0 = 20 + 10 * rs.u01();
        NewtonState st;
        newton_init(b, pos, o, d, u0, t0, st);
        while (newton_unconverged(st.fv) && st.counter < 100) {
            st.counter++;
            if (!newton_jacobian(b, d, st)) {
                // bezier.h:183: Vec3(u(),u(),u()) evaluates right to left under g++
                const double uz = rs.u01(), uy = rs.u01(), ux = rs.u01();
                st.res = mk(st.res.x + ux * 0.2 - 0.1, st.res.y + uy * 0.2 - 0.1, st.res.z + uz * 0.2 - 0.1);
            }
            newton_step(b, pos, o, d, st);
        }
        if (newton_accept(st) && st.res.x < len) {
            len = st.res.x;
            n = bez_normal(b, st.res.y, st.sn, st.cs);
            flag = true;
        }
    }
    return flag;
}

// Wave-level form: the (ray, start) pairs of all lanes whose ray enters the Bezier box -- 10 Newton solves each,
// 5 to 100 iterations apiece -- are dealt dynamically over all 64 lanes, so lanes whose own ray misses the box
// (or has no ray at all) work on their neighbours' solves and a lane that converges early takes the next pair.
// Per-lane serial solving costs the wave sum_k max_lanes(iterations); this costs about sum(iterations) / 64.
//   * start k of a ray draws (u0, t0) from the ray's stream at draws 2k, 2k+1 = one splitmix output, which is
//     what the sequential reference order gives as long as no earlier solve of that ray took the jitter branch;
//   * a solve that meets a singular Jacobian flags its ray, and flagged rays are redone by bezier_solve_serial
//     with the reference's exact sequential semantics (rare: |det J| < 1e-4);
//   * results come back through per-wave LDS; the ray's lane then takes the nearest accepted root, first start
//     winning ties (strict < in start order, bezier.h:260).
// Must be called by all lanes of the wave in uniform control flow.
static constexpr int kBezChunk = 32;  // rays whose solves are in flight together (results: 32 x 10 x 24 B per wave)
struct BezLds {
    double rt[kBezChunk * 10], ru[kBezChunk * 10], rth[kBezChunk * 10];
    uint32_t singular[kBezChunk];
    uint8_t lane_of_rank[64];
};

// `n0`: position of the ray's first draw in the stream `key` (0 for the eye pass, whose Bezier streams are keyed per ray;
// the photon pass continues the photon's own sequential stream, as the reference's rand() does).  On return `n0` has
// advanced by the draws the reference would have consumed (0 when the ray misses the box).
__device__ bool bezier_wave(const BezierRec &b_in, V3 pos, double cap_r, bool on, V3 o, V3 d, uint64_t key, uint32_t &n0,
                            double &len, V3 &n, volatile BezLds *L, const BezSlabRec *slabs = nullptr) {
    // the record sits at a wave-uniform address: through the constant address space its control points live in SGPRs for
    // the whole solve instead of being re-fetched per lane in every Bernstein sum (the LDS traffic of the loop keeps the
    // compiler from hoisting ordinary loads)
    const BezierRec b = load_uniform(&b_in);
    const int lane = threadIdx.x & 63;
    // `slabs` (eye pass: the ray's draws come from its own keyed stream, so skipped solves consume nothing anyone else sees):
    // rays that cannot come within the acceptance radius of the surface are treated like rays that miss the box
    bool want = on && bez_box(b, o, d);
    if (slabs && __ballot(want) != 0ull) want = want && bez_shell_maybe(slabs, pos, o, d);
    const unsigned long long wm = __ballot(want);
    if (wm == 0ull) return false;
    const int nwant = __popcll(wm);
    const int rank = __popcll(wm & ((1ull << lane) - 1ull));
    if (want) L->lane_of_rank[rank] = (uint8_t)lane;
    bool flag = false, redo = false;
    len = kInf;
    for (int base = 0; base < nwant; base += kBezChunk) {
        const int nsrc = (nwant - base < kBezChunk) ? nwant - base : kBezChunk;
        const int ntasks = nsrc * 10;
        if (lane < kBezChunk) L->singular[lane] = 0u;
        int next = 0;  // wave-uniform: first unassigned task
        bool busy = false;
        int task = 0;
        V3 so = o, sd = d;
        NewtonState st;
        st.counter = 0;
        st.fv = mk(0, 0, 0);
        while (true) {
            const unsigned long long fm = __ballot(!busy);
            const int nfree = __popcll(fm), avail = ntasks - next;
            // hand out tasks in batches (>= 8 lanes, or everything that is left) so that the solve set-up below
            // runs with many lanes active rather than once per finishing lane
            const int thresh = avail < 16 ? avail : 16;
            if (avail > 0 && nfree >= thresh) {
                const int r = __popcll(fm & ((1ull << lane) - 1ull));
                const bool take = !busy && r < avail;
                const int t = next + r;
                next += (nfree < avail) ? nfree : avail;
                const int src = take ? (int)L->lane_of_rank[base + t / 10] : lane;
                // every lane executes the shuffles (a disabled source lane would read as zero)
                const V3 fo = mk(__shfl(o.x, src), __shfl(o.y, src), __shfl(o.z, src));
                const V3 fd = mk(__shfl(d.x, src), __shfl(d.y, src), __shfl(d.z, src));
                const unsigned long long fkey = __shfl((unsigned long long)key, src);
                const uint32_t fn0 = (uint32_t)__shfl((int)n0, src);
                if (take) {
                    so = fo;
                    sd = fd;
                    task = t;
                    Stream ts(fkey);
                    ts.n = fn0 + 2u * (uint32_t)(t % 10);
                    double u0, t0;
                    if ((ts.n & 1u) == 0u) {
                        ts.pair(u0, t0);  // both draws from one finaliser
                    } else {
                        u0 = ts.u01();
                        t0 = ts.u01();
                    }
                    t0 = 20 + 10 * t0;
                    newton_init(b, pos, so, sd, u0, t0, st);
                    busy = true;
                }
            }
            if (__ballot(busy) == 0ull) break;
            if (busy) {
                if (newton_unconverged(st.fv) && st.counter < 100) {
                    st.counter++;
                    if (newton_jacobian(b, sd, st)) {
                        newton_step(b, pos, so, sd, st);
                    } else {
                        L->singular[task / 10] = 1u;  // this ray needs the sequential semantics
                        L->rt[task] = kInf;
                        busy = false;
                    }
                } else {
                    const bool acc = newton_accept(st);
                    L->rt[task] = acc ? st.res.x : kInf;
                    L->ru[task] = st.res.y;
                    L->rth[task] = st.res.z;
                    busy = false;
                }
            }
        }
        if (want && rank >= base && rank < base + nsrc) {
            const int sl = rank - base;
            if (L->singular[sl] != 0u) {
                redo = true;
            } else {
                int bk = -1;
                for (int k = 0; k < 10; k++) {
                    const double t = L->rt[sl * 10 + k];
                    if (t < len) {
                        len = t;
                        bk = k;
                    }
                }
                if (bk >= 0) {
                    const double th = L->rth[sl * 10 + bk];
                    n = bez_normal(b, L->ru[sl * 10 + bk], sin(th), cos(th));
                    flag = true;
                }
            }
        }
    }
    if (redo) {
        Stream rs(key);
        rs.n = n0;
        flag = bezier_solve_serial(b, pos, o, d, rs, len, n);
        n0 = rs.n;
    } else if (want) {
        n0 += 20u;
    }
    if (want) {
        n = (dot(n, d) < 0) ? n : -n;  // bezier.h:272
        double newt = b.box[3] - o.y;  // ymax - rayorig.y, bezier.h:273-281
        if (newt > 0.1) {
            newt = newt / d.y;
            const V3 np = o + d * newt;
            if ((np.x - pos.x) * (np.x - pos.x) + (np.z - pos.z) * (np.z - pos.z) <= cap_r * cap_r) {
                len = newt;
                n = mk(0, 1, 0);
            }
        }
    }
    return want && flag;
}

#endif
