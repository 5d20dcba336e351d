// TEST INFRASTRUCTURE — the CPU oracle (liborc.so).
//
// A from-scratch fp64 restatement of the reference's eye pass, written from SURVEY.md §8a and
// the reference sources read as text.  It exists only to CHECK the HIP path: only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The product
// (cgraytracing_amd/, include/cgrt.h) never links, imports or falls back to anything here.
//
// Parity status: PINNED.  tests/test_oracle_vs_golden.py compares this file against golden
// vectors generated from the compiled, unmodified reference (oracle/_ref, tests/golden/make_golden.py)
// and, where oracle/_ref/libcgrt_ref.so is present, against the reference run live.
//
// Written in C++ rather than C because quirk Q5 needs libstdc++ std::sort's exact tie behaviour
// (objects.h:254-260); everything else is plain scalar code.
//
// Citations are to /root/reference/<file>:<line>.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <omp.h>

#include "cgrt_rng.h"
#include "cgrt_testapi.h"

// `real`: the scalar of the restated algorithm.  The oracle proper (liborc.so) uses double.  The INSTRUMENTED build
// (liborc_flops.so, -DORC_COUNT_FLOPS; SURVEY.md section 8d "count these exactly") wraps the same double in a struct that
// counts every arithmetic operation (cgrt_flopcount.h); values, and therefore images, are identical.
#ifdef ORC_COUNT_FLOPS
#include "cgrt_flopcount.h"
namespace orc { thread_local FlopCounters g_fc; }
#else
namespace orc {
typedef double real;
using std::atan; using std::ceil; using std::cos; using std::exp; using std::fabs; using std::floor; using std::pow; using std::sin; using std::sqrt;
static inline double plain(double a) { return a; }
}
#endif
namespace orc {
// the C API's double arrays seen as arrays of `real` (the counting struct holds exactly one double)
static_assert(sizeof(real) == sizeof(double), "real wraps one double");
static inline real *as_real(double *p) { return reinterpret_cast<real *>(p); }
}

namespace orc {

// ---------------------------------------------------------------- vec3.h:11-119
struct V3 {
    real x, y, z;
    V3(real a = 0, real b = 0, real c = 0) : x(a), y(b), z(c) {}
};
static inline V3 operator+(const V3 &a, const V3 &b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline V3 operator-(const V3 &a, const V3 &b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline V3 operator-(const V3 &a) { return V3(-a.x, -a.y, -a.z); }
static inline V3 operator*(const V3 &a, real f) { return V3(a.x * f, a.y * f, a.z * f); }
static inline V3 mul(const V3 &a, const V3 &b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline real dot(const V3 &a, const V3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline V3 cross(const V3 &a, const V3 &b) {
    return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline real norm(const V3 &a) { return sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
// vec3.h:36-44: scales by 1/len only when len > 0
static inline V3 normalized(V3 a) {
    real len = norm(a);
    if (len > 0) {
        a.x *= 1 / len;
        a.y *= 1 / len;
        a.z *= 1 / len;
    }
    return a;
}
// vec3.h:95-97 (Sarrus, this exact association)
static inline real det3(const V3 &a, const V3 &b, const V3 &c) {
    return (a.x * b.y * c.z + b.x * c.y * a.z + c.x * a.y * b.z - a.x * c.y * b.z - b.x * a.y * c.z -
            c.x * b.y * a.z);
}
// vec3.h:103-119
static inline bool inv3(const V3 &a, const V3 &b, const V3 &c, V3 &ra, V3 &rb, V3 &rc) {
    real d = det3(a, b, c);
    if (d < 1e-4 && d > -1e-4) return false;
    ra.x = (b.y * c.z - b.z * c.y) / d;
    ra.y = (c.y * a.z - c.z * a.y) / d;
    ra.z = (a.y * b.z - a.z * b.y) / d;
    rb.x = (c.x * b.z - c.z * b.x) / d;
    rb.y = (a.x * c.z - a.z * c.x) / d;
    rb.z = (b.x * a.z - b.z * a.x) / d;
    rc.x = (b.x * c.y - c.x * b.y) / d;
    rc.y = (c.x * a.y - c.y * a.x) / d;
    rc.z = (a.x * b.y - a.y * b.x) / d;
    return true;
}
// util.h:16-42
static inline real max3(real a, real b, real c) { return (a > b && a > c) ? a : (b > c ? b : c); }
static inline real min3(real a, real b, real c) { return (a < b && a < c) ? a : (b < c ? b : c); }

static const real EPS = 1e-4;      // main.cpp:24
static const real INF = 1e10;      // main.cpp:25, objects.h:15
static const real BOXEPS = 1e-4;   // objects.h:144
static const int MINKD = 10;         // objects.h:143

// ---------------------------------------------------------------- keyed stream
struct Rng {
    uint64_t key;
    uint32_t ctr;
    real u01() { return (real)cgrt_rand31(key, ctr++) / 2147483647.0; }  // sampling.h:31-33
};
// sampling.h:35-43
static V3 lens_sample(Rng &r, real radius) {
    while (true) {
        real x = r.u01() * 2.0 - 1;
        real y = r.u01() * 2.0 - 1;
        if (x * x + y * y < 1) return V3(x, y, 0) * radius;
    }
}

// ---------------------------------------------------------------- objects.h:91-141
struct Tri {
    V3 pa, pb, pc;
};
// objects.h:96-111
static inline bool tri_intersect(const Tri &t, const V3 &o, const V3 &d, real &len, V3 &n) {
    V3 e1 = t.pa - t.pb, e2 = t.pa - t.pc, s = t.pa - o;
    real det1 = det3(d, e1, e2), det2 = det3(s, e1, e2), det3_ = det3(d, s, e2), det4 = det3(d, e1, s);
    if (det2 / det1 > 0.0 && det3_ / det1 >= 0.0 && det4 / det1 >= 0.0 && (det3_ + det4) / det1 <= 1.0) {
        len = det2 / det1;
        n = normalized(cross(t.pa - t.pb, t.pa - t.pc));
        return true;
    }
    return false;
}

// ---------------------------------------------------------------- objects.h:147-332
struct Node {
    real xmax, xmin, ymax, ymin, zmax, zmin;
    int left, right;
    std::vector<int> ids;  // triangleList (ids into Tree::tris)
};
struct Tree {
    std::vector<Tri> tris;  // construction order (pair.first)
    std::vector<Node> nodes;

    // objects.h:217-267.  The reference sorts vector<pair<int,Triangle>>; sorting the id list with a
    // comparator that returns the same answers drives std::sort through the same moves (Q5).
    void build(std::vector<int> sub, int par, bool isLeft, int dim, bool isRoot) {
        int cur = (int)nodes.size();
        if (!isRoot) (isLeft ? nodes[par].left : nodes[par].right) = cur;
        nodes.push_back(Node());
        {
            Node &nd = nodes[cur];
            nd.ids = sub;  // copied BEFORE sorting (Q7)
            nd.xmax = nd.ymax = nd.zmax = -INF;
            nd.xmin = nd.ymin = nd.zmin = INF;
            for (size_t i = 0; i < sub.size(); i++) {
                const Tri &a = tris[sub[i]];
                real mxx = max3(a.pa.x, a.pb.x, a.pc.x), mxy = max3(a.pa.y, a.pb.y, a.pc.y),
                       mxz = max3(a.pa.z, a.pb.z, a.pc.z);
                real mnx = min3(a.pa.x, a.pb.x, a.pc.x), mny = min3(a.pa.y, a.pb.y, a.pc.y),
                       mnz = min3(a.pa.z, a.pb.z, a.pc.z);
                if (nd.xmax < mxx) nd.xmax = mxx;
                if (nd.ymax < mxy) nd.ymax = mxy;
                if (nd.zmax < mxz) nd.zmax = mxz;
                if (nd.xmin > mnx) nd.xmin = mnx;
                if (nd.ymin > mny) nd.ymin = mny;
                if (nd.zmin > mnz) nd.zmin = mnz;
            }
            nd.left = nd.right = -1;
        }
        if ((int)sub.size() < MINKD) return;
        const std::vector<Tri> &T = tris;
        if (dim == 0)
            std::sort(sub.begin(), sub.end(), [&T](int p, int q) {
                return max3(T[p].pa.x, T[p].pb.x, T[p].pc.x) < max3(T[q].pa.x, T[q].pb.x, T[q].pc.x);
            });
        else if (dim == 1)
            std::sort(sub.begin(), sub.end(), [&T](int p, int q) {
                return max3(T[p].pa.y, T[p].pb.y, T[p].pc.y) < max3(T[q].pa.y, T[q].pb.y, T[q].pc.y);
            });
        else
            std::sort(sub.begin(), sub.end(), [&T](int p, int q) {
                return max3(T[p].pa.z, T[p].pb.z, T[p].pc.z) < max3(T[q].pa.z, T[q].pb.z, T[q].pc.z);
            });
        std::vector<int> l(sub.begin(), sub.begin() + sub.size() / 2), r(sub.begin() + sub.size() / 2, sub.end());
        int nd = (dim + 1) % 3;
        build(l, cur, true, nd, false);
        build(r, cur, false, nd, false);
    }
    void build_all() {
        std::vector<int> all(tris.size());
        for (size_t i = 0; i < all.size(); i++) all[i] = (int)i;
        nodes.clear();
        build(all, 0, false, 0, true);
    }

    // objects.h:166-200: six face tests, in this order
    static bool box_hit(const Node &b, const V3 &o, const V3 &d) {
        real t;
        V3 p;
        t = (b.xmax - o.x) / d.x; p = o + d * t;
        if (t > 0 && p.y >= b.ymin - BOXEPS && p.y <= b.ymax + BOXEPS && p.z >= b.zmin - BOXEPS && p.z <= b.zmax + BOXEPS) return true;
        t = (b.xmin - o.x) / d.x; p = o + d * t;
        if (t > 0 && p.y >= b.ymin - BOXEPS && p.y <= b.ymax + BOXEPS && p.z >= b.zmin - BOXEPS && p.z <= b.zmax + BOXEPS) return true;
        t = (b.ymax - o.y) / d.y; p = o + d * t;
        if (t > 0 && p.x >= b.xmin - BOXEPS && p.x <= b.xmax + BOXEPS && p.z >= b.zmin - BOXEPS && p.z <= b.zmax + BOXEPS) return true;
        t = (b.ymin - o.y) / d.y; p = o + d * t;
        if (t > 0 && p.x >= b.xmin - BOXEPS && p.x <= b.xmax + BOXEPS && p.z >= b.zmin - BOXEPS && p.z <= b.zmax + BOXEPS) return true;
        t = (b.zmax - o.z) / d.z; p = o + d * t;
        if (t > 0 && p.x >= b.xmin - BOXEPS && p.x <= b.xmax + BOXEPS && p.y >= b.ymin - BOXEPS && p.y <= b.ymax + BOXEPS) return true;
        t = (b.zmin - o.z) / d.z; p = o + d * t;
        if (t > 0 && p.x >= b.xmin - BOXEPS && p.x <= b.xmax + BOXEPS && p.y >= b.ymin - BOXEPS && p.y <= b.ymax + BOXEPS) return true;
        return false;
    }
    // objects.h:269-316.  Returns the number of times the running minimum improved (Q5), visits both
    // children unconditionally (Q6).
    int subtree(const V3 &o, const V3 &d, real &len, V3 &n, int cur, uint64_t *stats) const {
        const Node &nd = nodes[cur];
        if (stats) stats[0]++;
        if (!box_hit(nd, o, d)) return 0;
        if ((int)nd.ids.size() < MINKD) {
            real lt;
            V3 nt;
            int counter = 0;
            len = INF;
            for (size_t i = 0; i < nd.ids.size(); i++) {
                if (stats) stats[1]++;
                if (tri_intersect(tris[nd.ids[i]], o, d, lt, nt)) {
                    if (lt < len) {
                        len = lt;
                        n = nt;
                        counter++;
                    }
                }
            }
            return counter;
        }
        real ll, lr;
        V3 nl, nr;
        int cl = subtree(o, d, ll, nl, nd.left, stats);
        int cr = subtree(o, d, lr, nr, nd.right, stats);
        if (cl > 0) {
            if (cr > 0) {
                if (ll < lr) { len = ll; n = nl; } else { len = lr; n = nr; }
            } else { len = ll; n = nl; }
        } else if (cr > 0) { len = lr; n = nr; }
        return cl + cr;
    }
    // objects.h:318-332
    bool intersect(const V3 &o, const V3 &d, real &len, V3 &n, uint64_t *stats) const {
        int counter = subtree(o, d, len, n, 0, stats);
        if (counter > 0) {
            if (counter % 2 == 0) n = n * ((dot(n, d) < 0) ? 1 : -1);
            else n = n * ((dot(n, d) < 0) ? -1 : 1);
            return true;
        }
        return false;
    }
};

// ---------------------------------------------------------------- texture.h:14-83
struct Texture {
    int rows = 0, cols = 0;
    std::vector<uint8_t> rgb;  // texel = byte/256 (main.cpp:303-316)
    V3 normal, position;
    real lenx = 0, leny = 0;
    bool isbump = false;
    std::vector<real> height;  // texture.h:26-37
    V3 texel(int r, int c) const {
        // the reference would index out of bounds here; clamp
        r = r < 0 ? 0 : (r >= rows ? rows - 1 : r);
        c = c < 0 ? 0 : (c >= cols ? cols - 1 : c);
        const uint8_t *p = &rgb[3 * ((size_t)r * cols + c)];
        return V3((real)p[0] / 256.0, (real)p[1] / 256.0, (real)p[2] / 256.0);
    }
    void make_height() {
        height.assign((size_t)rows * cols, 0.0);
        if (!isbump) return;
        for (int i = 0; i < rows; i++)
            for (int j = 0; j < cols; j++) {
                V3 d = texel(i, j);
                real h = (0.299 * d.x + 0.587 * d.y + 0.114 * d.z);
                h = 1 - exp(-3.3 * h);
                h *= 0.5;
                height[(size_t)i * cols + j] = h;
            }
    }
    // texture.h:39-72
    bool color(const V3 &point, V3 &out) const {
        V3 d = point - position;
        d = d - normal * dot(d, normal);
        const real te = 1e-2;
        if (d.x < te && d.x > -te) {
            if (0 < d.y && d.y < lenx && 0 < d.z && d.z < leny) {
                int id1 = (int)floor(d.y / lenx * rows);
                int id2 = (int)floor(d.z / leny * cols);
                out = texel(id1, id2);
                return true;
            }
            return false;
        } else if (d.y < te && d.y > -te) {
            if (0 < d.x && d.x < lenx && 0 < d.z && d.z < leny) {
                int id1 = (int)floor(d.x / lenx * cols);
                int id2 = (int)floor(d.z / leny * rows);
                out = texel(id2, id1);
                return true;
            }
            return false;
        } else if (d.z < te && d.z > -te) {
            if (0 < d.x && d.x < lenx && 0 < d.y && d.y < leny) {
                int id1 = (int)floor(d.x / lenx * cols);
                int id2 = (int)floor(d.y / leny * rows);
                out = texel(rows - 1 - id2, id1);
                return true;
            }
            return false;
        }
        return false;
    }
};

// ---------------------------------------------------------------- objects
enum Kind { SPHERE, PLANE, MESH, BEZIER };
struct Obj {
    Kind kind;
    V3 color;
    real refl = 0, transp = 0;
    // sphere (objects.h:83-88)
    V3 center;
    real radius = 0, radius2 = 0;
    // plane (objects.h:541-547)
    V3 position, normal;
    int tex = -1;
    bool has_bump_tree = false;
    Tree bump;
    // mesh (objects.h:470-475)
    Tree tree;
    int objtype = 0;
    // bezier (bezier.h:303-313)
    std::vector<V3> cp;
    real xmax = 0, xmin = 0, ymax = 0, ymin = 0, zmax = 0, zmin = 0;
};

struct Scene {
    std::vector<Obj *> objs;
    std::vector<Texture *> textures;
    std::vector<int> mesh_ids, plane_ids;
    ~Scene() {
        for (auto o : objs) delete o;
        for (auto t : textures) delete t;
    }
};

// objects.h:45-68
static bool sphere_intersect(const Obj &s, const V3 &o, const V3 &d, real &len, V3 &n) {
    V3 l = s.center - o;
    real tca = dot(l, d);
    real l2 = dot(l, l);
    if (tca < 0 && l2 > s.radius2) return false;
    real d2 = dot(l, l) - tca * tca;
    if (d2 > s.radius2) return false;
    real thc = sqrt(s.radius2 - d2);
    real t0 = tca - thc, t1 = tca + thc;
    len = (t0 < 0) ? t1 : t0;
    V3 p = o + d * len;
    n = normalized(p - s.center);
    return true;
}
// objects.h:505-524
static bool plane_intersect(const Scene &sc, const Obj &pl, const V3 &o, const V3 &d, real &len, V3 &n,
                            uint64_t *stats) {
    V3 dd = pl.position - o;
    len = dot(dd, pl.normal) / dot(d, pl.normal);
    if (len > 0) {
        n = pl.normal;
        real lenp;
        V3 np;
        bool isbump = pl.tex >= 0 && sc.textures[pl.tex]->isbump;  // Q4: default Texture => false
        if (isbump && fabs(pl.normal.y - 1) < 1e-5 && pl.has_bump_tree && pl.bump.intersect(o, d, lenp, np, stats)) {
            if (lenp < len && lenp > 0) {
                len = lenp;
                n = np;
            }
        }
        return true;
    }
    return false;
}
// objects.h:405-455
static bool mesh_intersect(const Obj &m, const V3 &o, const V3 &d, real &len, V3 &n, uint64_t *stats) {
    bool res = m.tree.intersect(o, d, len, n, stats);
    if (m.objtype == 2) n = n * ((dot(n, V3(0, 1, 0)) > 0) ? 1 : -1);
    return res;
}

// ---------------------------------------------------------------- bezier.h
static const real Cni[7][7] = {{1, 0, 0, 0, 0, 0, 0}, {1, 1, 0, 0, 0, 0, 0}, {1, 2, 1, 0, 0, 0, 0},
                                 {1, 3, 3, 1, 0, 0, 0}, {1, 4, 6, 4, 1, 0, 0}, {1, 5, 10, 10, 5, 1, 0},
                                 {1, 6, 15, 20, 15, 6, 1}};
// bezier.h:30-40
static real Bern(int n, int i, real t) {
    if (i > n || i < 0) return 0;
    return Cni[n][i] * pow(1 - t, (real)(n - i)) * pow(t, (real)i);
}
static real dBern(int n, int i, real t) { return Bern(n - 1, i - 1, t) * (real)i - Bern(n - 1, i, t) * (real)(n - i); }
static V3 bez_valueP(const Obj &b, real u) {  // bezier.h:127-134
    V3 res;
    int n = (int)b.cp.size();
    for (int i = 0; i < n; i++) res = res + b.cp[i] * Bern(n - 1, i, u);
    return res;
}
static V3 bez_gradP(const Obj &b, real u) {  // bezier.h:135-142
    V3 res;
    int n = (int)b.cp.size();
    for (int i = 0; i < n; i++) res = res + b.cp[i] * dBern(n - 1, i, u);
    return res;
}
static V3 bez_func(const Obj &b, const V3 &p, const V3 &o, const V3 &d) {  // bezier.h:144-149
    V3 t = bez_valueP(b, p.y);
    t.x = t.z * sin(p.z);
    t.z *= cos(p.z);
    return o + d * p.x - b.position - t;
}
static void bez_grad(const Obj &b, const V3 &p, const V3 &d, V3 &ra, V3 &rb, V3 &rc) {  // bezier.h:150-162
    ra = d;
    V3 t1 = bez_gradP(b, p.y), t2 = bez_valueP(b, p.y);
    rb.x = -sin(p.z) * t1.z;
    rb.y = -t1.y;
    rb.z = -cos(p.z) * t1.z;
    rc.x = -cos(p.z) * t2.z;
    rc.y = 0;
    rc.z = sin(p.z) * t2.z;
}
static uint64_t g_jitter_events = 0;  // test observability only
// bezier.h:163-214
static V3 bez_newton(const Obj &b, const V3 &initial, const V3 &o, const V3 &dir, Rng &rng) {
    V3 res = initial;
    int counter = 0;
    V3 a, bb, c, d, e, f;  // d,e,f start at zero and keep stale values when the Jacobian is singular (Q11)
    V3 fv = bez_func(b, res, o, dir);
    while (norm(fv) > 1e-6 && counter < 100) {
        counter++;
        bez_grad(b, res, dir, a, bb, c);
        bool ok = inv3(a, bb, c, d, e, f);
        if (!ok) {
            g_jitter_events++;
            // bezier.h:183: Vec3(u(),u(),u()) -- g++ evaluates the arguments right to left (pinned by
            // tests/test_oracle_vs_ref.py::test_bezier_singular_jitter)
            real uz = rng.u01(), uy = rng.u01(), ux = rng.u01();
            V3 j = V3(ux, uy, uz) * 0.2;
            res = V3(res.x + j.x - 0.1, res.y + j.y - 0.1, res.z + j.z - 0.1);
        }
        V3 step = (d * fv.x + e * fv.y) + f * fv.z;  // vec3.h:99-101
        res = res - step;
        fv = bez_func(b, res, o, dir);
    }
    return res;
}
// bezier.h:72-126 (nearest-face t unused; a face only counts when t < 1e10)
static bool bez_box(const Obj &b, const V3 &o, const V3 &d) {
    real len = INF, t;
    bool flag = false;
    V3 p;
    t = (b.xmax - o.x) / d.x; p = o + d * t;
    if (t > 0 && p.y >= b.ymin - BOXEPS && p.y <= b.ymax + BOXEPS && p.z >= b.zmin - BOXEPS && p.z <= b.zmax + BOXEPS) if (t < len) { len = t; flag = true; }
    t = (b.xmin - o.x) / d.x; p = o + d * t;
    if (t > 0 && p.y >= b.ymin - BOXEPS && p.y <= b.ymax + BOXEPS && p.z >= b.zmin - BOXEPS && p.z <= b.zmax + BOXEPS) if (t < len) { len = t; flag = true; }
    t = (b.ymax - o.y) / d.y; p = o + d * t;
    if (t > 0 && p.x >= b.xmin - BOXEPS && p.x <= b.xmax + BOXEPS && p.z >= b.zmin - BOXEPS && p.z <= b.zmax + BOXEPS) if (t < len) { len = t; flag = true; }
    t = (b.ymin - o.y) / d.y; p = o + d * t;
    if (t > 0 && p.x >= b.xmin - BOXEPS && p.x <= b.xmax + BOXEPS && p.z >= b.zmin - BOXEPS && p.z <= b.zmax + BOXEPS) if (t < len) { len = t; flag = true; }
    t = (b.zmax - o.z) / d.z; p = o + d * t;
    if (t > 0 && p.x >= b.xmin - BOXEPS && p.x <= b.xmax + BOXEPS && p.y >= b.ymin - BOXEPS && p.y <= b.ymax + BOXEPS) if (t < len) { len = t; flag = true; }
    t = (b.zmin - o.z) / d.z; p = o + d * t;
    if (t > 0 && p.x >= b.xmin - BOXEPS && p.x <= b.xmax + BOXEPS && p.y >= b.ymin - BOXEPS && p.y <= b.ymax + BOXEPS) if (t < len) { len = t; flag = true; }
    return flag;
}
// bezier.h:225-290.  `n` is the caller's running normal (stale value is re-oriented when nothing hits,
// harmless because the return value is then false).
static bool bezier_intersect(const Obj &b, const V3 &o, const V3 &d, real &len, V3 &n, Rng &rng) {
    if (!bez_box(b, o, d)) return false;
    bool flag = false;
    len = INF;
    for (int i = 0; i < 10; i++) {
        real u0 = rng.u01();
        real t0 = 20 + 10 * rng.u01();
        V3 p = o + d * t0;
        p = p - b.position;
        real theta = (p.z < 0) ? 3.14159265 + atan(p.x / p.z) : atan(p.x / p.z);
        V3 res = bez_newton(b, V3(t0, u0, theta), o, d, rng);
        if (norm(bez_func(b, res, o, d)) < 1e-4 && res.x > 0 && res.y <= 1 && res.y >= 0) {
            if (res.x < len) {
                len = res.x;
                V3 rp = normalized(bez_gradP(b, res.y));  // bezier.h:215-224
                n = V3(rp.y * sin(res.z), -rp.z, rp.y * cos(res.z));
                flag = true;
            }
        }
    }
    n = n * ((dot(n, d) < 0) ? 1 : -1);
    real newt = b.ymax - o.y;
    if (newt > 0.1) {
        newt = newt / d.y;
        V3 np = o + d * newt;
        real cz = b.cp[b.cp.size() - 1].z;
        if ((np.x - b.position.x) * (np.x - b.position.x) + (np.z - b.position.z) * (np.z - b.position.z) <= cz * cz) {
            len = newt;
            n = V3(0, 1, 0);
        }
    }
    return flag;
}

// ---------------------------------------------------------------- trace(), main.cpp:42-100,129-157
struct Sink {
    real *acc = nullptr;      // 3 doubles of the current pixel
    uint32_t *nhit = nullptr;   // of the current pixel
    uint64_t nrays = 0;
    real *hp = nullptr;
    int64_t *hp_pix = nullptr;
    uint64_t hp_cap = 0, hp_n = 0;
    int64_t label = 0;
    uint64_t stats[2] = {0, 0};  // node tests, triangle tests
};
struct RayCtx {
    uint64_t seed, pixel, sample;
};

// `seq`: when non-null (photon pass) Bezier draws continue on that sequential stream, as the reference's rand()
// does; otherwise (eye pass) they come from the ray's own path-keyed stream.
static bool obj_intersect(const Scene &sc, int i, const V3 &o, const V3 &d, real &len, V3 &n, const RayCtx &rc,
                          uint32_t path, Sink &sink, Rng *seq = nullptr) {
    const Obj &ob = *sc.objs[i];
    switch (ob.kind) {
        case SPHERE: return sphere_intersect(ob, o, d, len, n);
        case PLANE: return plane_intersect(sc, ob, o, d, len, n, sink.stats);
        case MESH: return mesh_intersect(ob, o, d, len, n, sink.stats);
        case BEZIER: {
            if (seq) return bezier_intersect(ob, o, d, len, n, *seq);
            Rng r{cgrt_key(rc.seed, rc.pixel, rc.sample, ((uint64_t)path << 16) | (uint64_t)(i + 1)), 0};
            return bezier_intersect(ob, o, d, len, n, r);
        }
    }
    return false;
}

static void trace(const Scene &sc, const V3 &org, const V3 &dir, V3 adj, int depth_left, uint32_t path,
                  const RayCtx &rc, Sink &sink) {
    if (depth_left <= 0) return;  // main.cpp:46
    sink.nrays++;
    real len = 0;
    int id = -1;
    V3 normalvec, temp;
    real nearest = INF;
    for (int i = 0; i < (int)sc.objs.size(); i++) {  // main.cpp:55-63 (strict <: first object wins ties, Q1)
        if (obj_intersect(sc, i, org, dir, len, temp, rc, path, sink)) {
            if (len < nearest) {
                id = i;
                nearest = len;
                normalvec = temp;
            }
        }
    }
    if (id == -1) return;
    const Obj &obj = *sc.objs[id];
    V3 P = org + dir * nearest;
    bool into = true;
    V3 n_old = normalvec;
    if (dot(normalvec, dir) > 0) {  // main.cpp:73-76
        normalvec = -normalvec;
        into = false;
    }
    V3 f = obj.color;  // getSurfaceColor: objects.h:78,466,533 ; bezier.h:299
    if (obj.kind == PLANE && obj.tex >= 0) {
        V3 c;
        if (sc.textures[obj.tex]->color(P, c)) f = c;
    }
    if (obj.refl < EPS && obj.transp < EPS) {  // main.cpp:82-100
        V3 hf = mul(f, adj);
        sink.acc[0] += hf.x; sink.acc[1] += hf.y; sink.acc[2] += hf.z;
        (*sink.nhit)++;
        if (sink.hp && sink.hp_n < sink.hp_cap) {
            real *o = sink.hp + 9 * sink.hp_n;
            o[0] = hf.x; o[1] = hf.y; o[2] = hf.z;
            o[3] = P.x; o[4] = P.y; o[5] = P.z;
            o[6] = normalvec.x; o[7] = normalvec.y; o[8] = normalvec.z;
            sink.hp_pix[sink.hp_n] = sink.label;
        }
        sink.hp_n++;
    } else if (obj.transp < EPS) {  // mirror, main.cpp:129-134
        V3 newdir = dir - normalvec * 2.0 * dot(normalvec, dir);
        real refl = obj.refl;
        V3 P2 = P + normalvec * EPS;
        trace(sc, P2, newdir, mul(f, adj) * refl, depth_left - 1, path * 2, rc, sink);
    } else {  // glass, main.cpp:135-157
        real nc = 1.0, nt = 1.33, nnt = into ? nc / nt : nt / nc, ddn = dot(dir, normalvec), cos2t;
        V3 refl_dir = dir - n_old * 2.0 * dot(n_old, dir);
        if ((cos2t = 1 - nnt * nnt * (1 - ddn * ddn)) < 0) {  // TIR keeps adj (Q3)
            trace(sc, P + normalvec * EPS, refl_dir, adj, depth_left - 1, path * 2, rc, sink);
            return;
        }
        V3 refr_dir = normalized(dir * nnt - n_old * ((into ? 1 : -1) * (ddn * nnt + sqrt(cos2t))));
        real a = nt - nc, b = nt + nc, R0 = a * a / (b * b), c = 1 - (into ? -ddn : dot(refr_dir, n_old));
        real Re = R0 + (1 - R0) * c * c * c * c * c;
        V3 fa = mul(f, adj);
        trace(sc, P + normalvec * EPS, refl_dir, fa * Re, depth_left - 1, path * 2, rc, sink);
        trace(sc, P - normalvec * EPS, refr_dir, fa * (1 - Re), depth_left - 1, path * 2 + 1, rc, sink);
    }
}

// ---------------------------------------------------------------- photon pass, main.cpp:101-128,158-165,223-258
// hitpoints.h:6-20 / hash.h:20-70.  Buckets keep insertion order; a photon event visits the 27 cells around it and,
// for each, EVERY hitpoint of the bucket that cell hashes to (hash.h:32-34), so a hitpoint is updated once per
// neighbouring cell whose hash equals its bucket -- normally once, twice on a hash collision inside the 3x3x3 block.
struct Hitpt {
    V3 f, pos, normal, flux;
    real r2;
    int n, h, w;
};
struct HashGrid {
    int hashsize, ncell;
    real celllength;
    std::vector<std::vector<Hitpt> > buckets;
    HashGrid(int hs, real cl) : hashsize(hs) {  // hash.h:22-30
        ncell = (int)ceil(70.0 / cl);
        celllength = 70.0 / ncell;
        buckets.assign((size_t)hs, std::vector<Hitpt>());
    }
    unsigned hash(int ix, int iy, int iz) const {  // hash.h:35-37 (wrapping int products)
        return (((unsigned)ix * 73856093u) ^ ((unsigned)iy * 19349663u) ^ ((unsigned)iz * 83492791u)) % (unsigned)hashsize;
    }
    void coord(real x, real y, real z, int &ix, int &iy, int &iz) const {  // hash.h:38-42
        ix = (int)floor((x - (-35.0)) / celllength);
        iy = (int)floor((y - (-35.0)) / celllength);
        iz = (int)floor((z - (-15.0)) / celllength);
    }
    void insert(const Hitpt &hp) {  // hash.h:43-54
        int ix, iy, iz;
        coord(hp.pos.x, hp.pos.y, hp.pos.z, ix, iy, iz);
        buckets[hash(ix, iy, iz)].push_back(hp);
    }
};
static const real PI_REF = 3.14159265358979;  // main.cpp:26

// sampling.h:11-29 on the photon's sequential stream
static V3 sample_sphere(Rng &r) {
    while (true) {
        real x = r.u01() * 2.0 - 1, y = r.u01() * 2.0 - 1, z = r.u01() * 2.0 - 1;
        if (x * x + y * y + z * z <= 1) return normalized(V3(x, y, z));
    }
}
static V3 sample_halfsphere(Rng &r, const V3 &dir) {
    while (true) {
        V3 s = sample_sphere(r);
        if (dot(s, dir) > 0) return s;
    }
}

// eye-pass trace that stores Hitpoints (main.cpp:85-100) instead of accumulating them
static void trace_store(const Scene &sc, const V3 &org, const V3 &dir, V3 adj, int depth_left, uint32_t path,
                        const RayCtx &rc, HashGrid &ht, int lw, int lh, real r0, Sink &sink) {
    if (depth_left <= 0) return;
    real len = 0;
    int id = -1;
    V3 normalvec, temp;
    real nearest = INF;
    for (int i = 0; i < (int)sc.objs.size(); i++)
        if (obj_intersect(sc, i, org, dir, len, temp, rc, path, sink) && len < nearest) { id = i; nearest = len; normalvec = temp; }
    if (id == -1) return;
    const Obj &obj = *sc.objs[id];
    V3 P = org + dir * nearest;
    bool into = true;
    V3 n_old = normalvec;
    if (dot(normalvec, dir) > 0) { normalvec = -normalvec; into = false; }
    V3 f = obj.color;
    if (obj.kind == PLANE && obj.tex >= 0) { V3 c; if (sc.textures[obj.tex]->color(P, c)) f = c; }
    if (obj.refl < EPS && obj.transp < EPS) {
        Hitpt hp;
        hp.f = mul(f, adj); hp.pos = P; hp.normal = normalvec; hp.w = lw; hp.h = lh; hp.flux = V3(); hp.r2 = r0 * r0; hp.n = 0;
        ht.insert(hp);
    } else if (obj.transp < EPS) {
        V3 newdir = dir - normalvec * 2.0 * dot(normalvec, dir);
        trace_store(sc, P + normalvec * EPS, newdir, mul(f, adj) * obj.refl, depth_left - 1, path * 2, rc, ht, lw, lh, r0, sink);
    } else {
        real nc = 1.0, nt = 1.33, nnt = into ? nc / nt : nt / nc, ddn = dot(dir, normalvec), cos2t;
        V3 refl_dir = dir - n_old * 2.0 * dot(n_old, dir);
        if ((cos2t = 1 - nnt * nnt * (1 - ddn * ddn)) < 0) {
            trace_store(sc, P + normalvec * EPS, refl_dir, adj, depth_left - 1, path * 2, rc, ht, lw, lh, r0, sink);
            return;
        }
        V3 refr_dir = normalized(dir * nnt - n_old * ((into ? 1 : -1) * (ddn * nnt + sqrt(cos2t))));
        real a = nt - nc, b = nt + nc, R0 = a * a / (b * b), c = 1 - (into ? -ddn : dot(refr_dir, n_old));
        real Re = R0 + (1 - R0) * c * c * c * c * c;
        V3 fa = mul(f, adj);
        trace_store(sc, P + normalvec * EPS, refl_dir, fa * Re, depth_left - 1, path * 2, rc, ht, lw, lh, r0, sink);
        trace_store(sc, P - normalvec * EPS, refr_dir, fa * (1 - Re), depth_left - 1, path * 2 + 1, rc, ht, lw, lh, r0, sink);
    }
}

// trace(flag=false), main.cpp:42-81,101-128,129-165
struct EventLog {  // optional record of the diffuse photon hits, in serial order (function-level parity probe)
    real *out = nullptr;  // 10 doubles each: photon index, P(3), n(3), flux(3)
    uint64_t cap = 0, n = 0;
    int64_t photon = 0;
};
static EventLog *g_evlog = nullptr;

static void trace_photon(const Scene &sc, const V3 &org, const V3 &dir, V3 flux, V3 adj, int depth_left, Rng &rng,
                         HashGrid &ht, real alpha, Sink &sink) {
    if (depth_left <= 0) return;
    static const RayCtx none{0, 0, 0};
    real len = 0;
    int id = -1;
    V3 normalvec, temp;
    real nearest = INF;
    for (int i = 0; i < (int)sc.objs.size(); i++)
        if (obj_intersect(sc, i, org, dir, len, temp, none, 0, sink, &rng) && len < nearest) { id = i; nearest = len; normalvec = temp; }
    if (id == -1) return;
    const Obj &obj = *sc.objs[id];
    V3 P = org + dir * nearest;
    bool into = true;
    V3 n_old = normalvec;
    if (dot(normalvec, dir) > 0) { normalvec = -normalvec; into = false; }
    V3 f = obj.color;
    if (obj.kind == PLANE && obj.tex >= 0) { V3 c; if (sc.textures[obj.tex]->color(P, c)) f = c; }
    real p = max3(f.x, f.y, f.z);  // main.cpp:79
    if (obj.refl < EPS && obj.transp < EPS) {
        if (g_evlog) {
            if (g_evlog->n < g_evlog->cap) {
                real *e = g_evlog->out + 10 * g_evlog->n;
                e[0] = (real)g_evlog->photon;
                e[1] = P.x; e[2] = P.y; e[3] = P.z; e[4] = normalvec.x; e[5] = normalvec.y; e[6] = normalvec.z;
                e[7] = flux.x; e[8] = flux.y; e[9] = flux.z;
            }
            g_evlog->n++;
        }
        int ix, iy, iz;
        ht.coord(P.x, P.y, P.z, ix, iy, iz);
        ix -= 1; iy -= 1; iz -= 1;
        for (int dx = 0; dx < 3; dx++)
            for (int dy = 0; dy < 3; dy++)
                for (int dz = 0; dz < 3; dz++) {
                    std::vector<Hitpt> &bk = ht.buckets[ht.hash(ix + dx, iy + dy, iz + dz)];
                    for (size_t i = 0; i < bk.size(); i++) {
                        Hitpt &hp = bk[i];
                        V3 dd = hp.pos - P;
                        if ((dot(hp.normal, normalvec) > EPS) && (dot(dd, dd) <= hp.r2)) {  // main.cpp:116
                            real g = (hp.n * alpha + alpha) / (hp.n * alpha + 1.0);    // main.cpp:119
                            hp.r2 *= g;
                            hp.n++;
                            hp.flux = (hp.flux + mul(hp.f, flux) * (1.0 / PI_REF)) * g;  // main.cpp:122
                        }
                    }
                }
        V3 newdir = sample_halfsphere(rng, normalvec);                                   // main.cpp:126
        trace_photon(sc, P, newdir, mul(f, flux) * (1.0 / p), adj, depth_left - 1, rng, ht, alpha, sink);
    } else if (obj.transp < EPS) {
        V3 newdir = dir - normalvec * 2.0 * dot(normalvec, dir);
        trace_photon(sc, P + normalvec * EPS, newdir, mul(f, flux) * obj.refl, mul(f, adj) * obj.refl, depth_left - 1, rng, ht,
                     alpha, sink);
    } else {
        real nc = 1.0, nt = 1.33, nnt = into ? nc / nt : nt / nc, ddn = dot(dir, normalvec), cos2t;
        V3 refl_dir = dir - n_old * 2.0 * dot(n_old, dir);
        if ((cos2t = 1 - nnt * nnt * (1 - ddn * ddn)) < 0) {
            trace_photon(sc, P + normalvec * EPS, refl_dir, flux, adj, depth_left - 1, rng, ht, alpha, sink);
            return;
        }
        V3 refr_dir = normalized(dir * nnt - n_old * ((into ? 1 : -1) * (ddn * nnt + sqrt(cos2t))));
        real a = nt - nc, b = nt + nc, R0 = a * a / (b * b), c = 1 - (into ? -ddn : dot(refr_dir, n_old));
        real Re = R0 + (1 - R0) * c * c * c * c * c;
        V3 fa = mul(f, adj);
        if (rng.u01() < 0.5)  // main.cpp:160-164: Russian roulette; the photon's flux is not attenuated by glass
            trace_photon(sc, P + normalvec * EPS, refl_dir, flux, fa * Re * 0.3, depth_left - 1, rng, ht, alpha, sink);
        else
            trace_photon(sc, P - normalvec * EPS, refr_dir, flux, fa * (1 - Re * 0.3), depth_left - 1, rng, ht, alpha, sink);
    }
}

// ---------------------------------------------------------------- loaders, objects.h:338-403
// Token-level restatement of the three scanf formats (whitespace-insensitive like scanf, Q8).
// Malformed input stops the load (the reference would reuse stale values / loop); returns false.
struct Tok {
    std::vector<std::string> t;
    size_t i = 0;
    bool load(const char *file) {
        FILE *f = std::fopen(file, "r");
        if (!f) return false;
        char buf[256];
        while (std::fscanf(f, "%255s", buf) == 1) t.push_back(buf);
        std::fclose(f);
        return true;
    }
    bool more() const { return i < t.size(); }
    bool lit(const char *s) {
        if (i < t.size() && t[i] == s) { i++; return true; }
        return false;
    }
    bool num(real &v) {
        if (i >= t.size()) return false;
        char *e;
        v = std::strtod(t[i].c_str(), &e);
        if (*e) return false;
        i++;
        return true;
    }
    bool integer(int &v) {
        if (i >= t.size()) return false;
        char *e;
        long q = std::strtol(t[i].c_str(), &e, 10);
        if (*e) return false;
        v = (int)q;
        i++;
        return true;
    }
};
static bool load_mesh(const char *file, real a, const V3 &b, int type, std::vector<Tri> &out) {
    Tok tk;
    if (!tk.load(file)) return true;  // missing file => freopen fails => empty mesh
    auto xf = [&](real x, real y, real z) { return V3(x, y, -z) * a + b; };  // objects.h:348,365,384
    if (type == 0) {
        while (tk.more()) {
            real v[9];
            if (!tk.lit("begin")) return false;
            for (int k = 0; k < 3; k++) {
                if (!tk.lit("vertex")) return false;
                for (int c = 0; c < 3; c++) if (!tk.num(v[3 * k + c])) return false;
            }
            if (!tk.lit("end")) return false;
            out.push_back(Tri{xf(v[0], v[1], v[2]), xf(v[3], v[4], v[5]), xf(v[6], v[7], v[8])});
        }
        return true;
    }
    int num;
    if (!tk.integer(num)) return false;
    std::vector<V3> verts;
    for (int i = 0; i < num; i++) {
        real x, y, z;
        if (!tk.lit("v") || !tk.num(x) || !tk.num(y) || !tk.num(z)) return false;
        verts.push_back(V3(x, y, -z));
    }
    if (type == 2) {  // optional vn / vt blocks (objects.h:387-392)
        while (tk.lit("vn")) { real q; tk.num(q); tk.num(q); tk.num(q); }
        while (tk.lit("vt")) { real q; tk.num(q); tk.num(q); }
    }
    if (!tk.integer(num)) return false;
    for (int i = 0; i < num; i++) {
        int id[3];
        if (!tk.lit("f")) return false;
        for (int k = 0; k < 3; k++) {
            if (type == 1) {
                if (!tk.integer(id[k])) return false;
            } else {
                if (!tk.more()) return false;
                id[k] = std::atoi(tk.t[tk.i++].c_str());  // "a/b/c": leading integer
            }
            if (id[k] < 1 || id[k] > (int)verts.size()) return false;
        }
        out.push_back(Tri{verts[id[0] - 1] * a + b, verts[id[1] - 1] * a + b, verts[id[2] - 1] * a + b});
    }
    return true;
}

}  // namespace orc

using namespace orc;
static V3 v3(const double *p) { return V3(p[0], p[1], p[2]); }
static int g_threads = 1;

extern "C" {

uint64_t orc_debug_jitter_events(void) { return orc::g_jitter_events; }
void orc_set_threads(int n) { g_threads = n < 1 ? 1 : n; }
void *orc_scene_new(void) { return new Scene(); }
void orc_scene_free(void *p) { delete (Scene *)p; }

int orc_add_sphere(void *sp, const double *c, double r, const double *col, double refl, double transp) {
    Scene *s = (Scene *)sp;
    Obj *o = new Obj();
    o->kind = SPHERE; o->center = v3(c); o->radius = r; o->radius2 = r * r;  // objects.h:35
    o->color = v3(col); o->refl = refl; o->transp = transp;
    s->objs.push_back(o);
    return (int)s->objs.size() - 1;
}
int orc_add_texture(void *sp, const uint8_t *rgb, int rows, int cols, const double *n, const double *p, double lx,
                    double ly, int bump) {
    Scene *s = (Scene *)sp;
    Texture *t = new Texture();
    t->rows = rows; t->cols = cols;
    t->rgb.assign(rgb, rgb + (size_t)rows * cols * 3);
    t->normal = v3(n); t->position = v3(p); t->lenx = lx; t->leny = ly; t->isbump = bump != 0;
    t->make_height();
    s->textures.push_back(t);
    return (int)s->textures.size() - 1;
}
int orc_add_plane(void *sp, const double *p, const double *n, const double *col, double refl, double transp,
                  int tex_id) {
    Scene *s = (Scene *)sp;
    Obj *o = new Obj();
    o->kind = PLANE; o->position = v3(p); o->normal = v3(n); o->color = v3(col);
    o->refl = refl; o->transp = transp; o->tex = tex_id;
    if (tex_id >= 0) {
        const Texture &tx = *s->textures[tex_id];
        if (std::fabs(o->normal.y - 1.0) < 1e-5 && tx.isbump) {  // objects.h:482-503 (Q10)
            const int step = 3;
            const int R = tx.rows, C = tx.cols;
            for (int i = 0; i < R / step - 1; i++)
                for (int j = 0; j < C / step - 1; j++) {
                    double x1 = tx.position.x + tx.lenx * j * step / C;
                    double x2 = tx.position.x + tx.lenx * (j + 1) * step / C;
                    double y1 = tx.position.z + tx.leny * i * step / R;
                    double y2 = tx.position.z + tx.leny * (i + 1) * step / R;
                    auto H = [&](int a, int b) { return tx.height[(size_t)a * C + b]; };
                    V3 a(x1, H(i * step, j * step) + o->position.y, y1);
                    V3 b(x2, H(i * step, (j + 1) * step) + o->position.y, y1);
                    V3 c(x1, H((i + 1) * step, j * step) + o->position.y, y2);
                    V3 d(x2, H((i + 1) * step, (j + 1) * step) + o->position.y, y2);
                    o->bump.tris.push_back(Tri{a, b, c});
                    o->bump.tris.push_back(Tri{d, b, c});
                }
            o->bump.build_all();
            o->has_bump_tree = true;
        }
    }
    s->plane_ids.push_back((int)s->objs.size());
    s->objs.push_back(o);
    return (int)s->objs.size() - 1;
}
static int add_mesh(Scene *s, std::vector<Tri> &tris, const double *col, double refl, double transp, int type) {
    Obj *o = new Obj();
    o->kind = MESH; o->color = v3(col); o->refl = refl; o->transp = transp; o->objtype = type;
    o->tree.tris.swap(tris);
    o->tree.build_all();
    s->mesh_ids.push_back((int)s->objs.size());
    s->objs.push_back(o);
    return (int)s->objs.size() - 1;
}
int orc_add_mesh_file(void *sp, const char *file, double a, const double *b, const double *col, double refl,
                      double transp, int typeofdata) {
    std::vector<Tri> tris;
    if (!load_mesh(file, a, v3(b), typeofdata, tris)) return -1;
    return add_mesh((Scene *)sp, tris, col, refl, transp, typeofdata);
}
int orc_add_mesh_tris(void *sp, const double *tri, int ntri, const double *col, double refl, double transp,
                      int typeofdata) {
    std::vector<Tri> tris;
    for (int i = 0; i < ntri; i++) tris.push_back(Tri{v3(tri + 9 * i), v3(tri + 9 * i + 3), v3(tri + 9 * i + 6)});
    return add_mesh((Scene *)sp, tris, col, refl, transp, typeofdata);
}
int orc_add_bezier(void *sp, const double *cp, int ncp, const double *pos, const double *col, double refl,
                   double transp) {
    Scene *s = (Scene *)sp;
    Obj *o = new Obj();
    o->kind = BEZIER; o->color = v3(col); o->refl = refl; o->transp = transp; o->position = v3(pos);
    for (int i = 0; i < ncp; i++) o->cp.push_back(v3(cp + 3 * i));
    double max_z = -INF, max_y = -INF, min_y = INF;  // bezier.h:50-69
    for (auto &c : o->cp) {
        if (c.z > max_z) max_z = c.z;
        if (c.y > max_y) max_y = c.y;
        if (c.y < min_y) min_y = c.y;
    }
    o->xmax = max_z + o->position.x; o->xmin = -max_z + o->position.x;
    o->ymax = max_y + o->position.y; o->ymin = min_y + o->position.y;
    o->zmax = max_z + o->position.z; o->zmin = -max_z + o->position.z;
    s->objs.push_back(o);
    return (int)s->objs.size() - 1;
}

// ---- introspection (same shapes as the ref_* functions) ----
static const Tree *pick_tree(void *sp, int kind, int idx) {
    Scene *s = (Scene *)sp;
    return kind == 0 ? &s->objs[s->mesh_ids[idx]]->tree : &s->objs[s->plane_ids[idx]]->bump;
}
int orc_mesh_ntris(void *sp, int mesh) { return (int)pick_tree(sp, 0, mesh)->tris.size(); }
static void dump_tris(const Tree &t, double *out) {
    for (size_t i = 0; i < t.tris.size(); i++) {
        const Tri &q = t.tris[i];
        double *o = out + 9 * i;
        o[0] = q.pa.x; o[1] = q.pa.y; o[2] = q.pa.z; o[3] = q.pb.x; o[4] = q.pb.y; o[5] = q.pb.z;
        o[6] = q.pc.x; o[7] = q.pc.y; o[8] = q.pc.z;
    }
}
void orc_mesh_tris(void *sp, int mesh, double *out) { dump_tris(*pick_tree(sp, 0, mesh), out); }
int orc_tree_nnodes(void *sp, int kind, int idx) { return (int)pick_tree(sp, kind, idx)->nodes.size(); }
int orc_tree_nleaftris(void *sp, int kind, int idx) {
    const Tree &t = *pick_tree(sp, kind, idx);
    size_t k = 0;
    for (auto &n : t.nodes) if ((int)n.ids.size() < MINKD) k += n.ids.size();
    return (int)k;
}
void orc_tree_dump(void *sp, int kind, int idx, int32_t *node_lr_size, int32_t *leaf_ids, double *bbox) {
    const Tree &t = *pick_tree(sp, kind, idx);
    size_t k = 0;
    for (size_t i = 0; i < t.nodes.size(); i++) {
        const Node &nd = t.nodes[i];
        if (node_lr_size) {
            node_lr_size[3 * i] = nd.left; node_lr_size[3 * i + 1] = nd.right; node_lr_size[3 * i + 2] = (int32_t)nd.ids.size();
        }
        if (bbox) {
            bbox[6 * i] = nd.xmin; bbox[6 * i + 1] = nd.xmax; bbox[6 * i + 2] = nd.ymin;
            bbox[6 * i + 3] = nd.ymax; bbox[6 * i + 4] = nd.zmin; bbox[6 * i + 5] = nd.zmax;
        }
        if ((int)nd.ids.size() < MINKD && leaf_ids) for (int id : nd.ids) leaf_ids[k++] = id;
    }
}
int orc_plane_bump_ntris(void *sp, int idx) { return (int)pick_tree(sp, 1, idx)->tris.size(); }
void orc_plane_bump_tris(void *sp, int idx, double *out) { dump_tris(*pick_tree(sp, 1, idx), out); }

// ---- function-level probes ----
void orc_intersect_batch(void *sp, int obj, const double *org, const double *dir, const uint64_t *keys, int n,
                         int32_t *hit, double *len, double *normal) {
    Scene *s = (Scene *)sp;
    for (int i = 0; i < n; i++) {
        const Obj &ob = *s->objs[obj];
        real l = 0;
        V3 nv;
        bool h = false;
        V3 o = v3(org + 3 * i), d = v3(dir + 3 * i);
        switch (ob.kind) {
            case SPHERE: h = sphere_intersect(ob, o, d, l, nv); break;
            case PLANE: h = plane_intersect(*s, ob, o, d, l, nv, nullptr); break;
            case MESH: h = mesh_intersect(ob, o, d, l, nv, nullptr); break;
            case BEZIER: { Rng r{keys ? keys[i] : 0, 0}; h = bezier_intersect(ob, o, d, l, nv, r); } break;
        }
        hit[i] = h ? 1 : 0;
        len[i] = l;
        normal[3 * i] = nv.x; normal[3 * i + 1] = nv.y; normal[3 * i + 2] = nv.z;
    }
}
void orc_surface_color_batch(void *sp, int obj, const double *pts, int n, double *out) {
    Scene *s = (Scene *)sp;
    const Obj &ob = *s->objs[obj];
    for (int i = 0; i < n; i++) {
        V3 f = ob.color, c;
        if (ob.kind == PLANE && ob.tex >= 0 && s->textures[ob.tex]->color(v3(pts + 3 * i), c)) f = c;
        out[3 * i] = f.x; out[3 * i + 1] = f.y; out[3 * i + 2] = f.z;
    }
}
void orc_lens_samples(uint64_t seed, const int64_t *pix, const int32_t *smp, int n, double radius, double *out) {
    for (int i = 0; i < n; i++) {
        Rng r{cgrt_key(seed, (uint64_t)pix[i], (uint64_t)smp[i], 0), 0};
        V3 v = lens_sample(r, radius);
        out[3 * i] = v.x; out[3 * i + 1] = v.y; out[3 * i + 2] = v.z;
    }
}

// ---- the eye pass (main.cpp:185-219).  Same signature as ref_trace_grid; `hashsize` is ignored.
// stats (optional, 2 x uint64): node tests, triangle tests.  Hitpoint capture forces one thread.
double orc_trace_grid(void *sp, const orc_camera *cam, const orc_grid *g, int hashsize, double *acc, uint32_t *nhit,
                      uint64_t *nrays, double *hp, int64_t *hp_pix, uint64_t hp_cap, uint64_t *hp_count) {
    (void)hashsize;
    Scene *s = (Scene *)sp;
    const int W = g->W, H = g->H;
    V3 camorg = v3(cam->cam);
    uint64_t total_rays = 0, total_hp = 0;
    int threads = hp ? 1 : g_threads;
    auto t0 = std::chrono::steady_clock::now();
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads) reduction(+ : total_rays, total_hp)
    for (int h = g->row0; h < g->row0 + g->nrows; h++) {
        Sink sink;
        sink.hp = as_real(hp); sink.hp_pix = hp_pix; sink.hp_cap = hp_cap;
        if (hp) sink.hp_n = total_hp;  // single-threaded in this mode
        for (int w = 0; w < W; w++) {
            size_t pix = (size_t)(h - g->row0) * W + w;
            real x = (2.0 * (real((double)w) / W) - 1) * cam->half_width;           // main.cpp:188
            real y = (2.0 * (real((double)h) / H) - 1) * cam->half_width * H / W;   // main.cpp:189
            V3 dir = normalized(V3(x, y, 0) - camorg);                               // main.cpp:198
            V3 pof = dir * ((cam->focus_plane - camorg.z) / dir.z) + camorg;         // main.cpp:203
            sink.acc = as_real(acc + 3 * pix);
            sink.nhit = nhit + pix;
            for (int j = g->sample0; j < g->sample0 + g->spp; j++) {
                RayCtx rc{g->seed, (uint64_t)h * (uint64_t)W + (uint64_t)w, (uint64_t)j};
                sink.label = ((int64_t)(j - g->sample0) << 32) | (int64_t)pix;
                if (cam->lens_radius > 0) {
                    Rng r{cgrt_key(rc.seed, rc.pixel, rc.sample, 0), 0};
                    V3 neworg = camorg + lens_sample(r, cam->lens_radius);           // main.cpp:205
                    V3 newdir = normalized(pof - neworg);                            // main.cpp:206
                    trace(*s, neworg, newdir, V3(1, 1, 1), g->depth, 1, rc, sink);  // main.cpp:207
                } else {
                    trace(*s, camorg, dir, V3(1, 1, 1), g->depth, 1, rc, sink);     // main.cpp:209
                }
            }
        }
        total_rays += sink.nrays;
        if (hp) total_hp = sink.hp_n; else total_hp += sink.hp_n;
    }
    auto t1 = std::chrono::steady_clock::now();
    if (nrays) *nrays = total_rays;
    if (hp_count) *hp_count = total_hp;
    return std::chrono::duration<double>(t1 - t0).count();
}

// diffuse photon hits of photons [first, first+count) in serial order: 10 doubles each (photon, P, n, flux)
uint64_t orc_photon_events(void *sp, const orc_photons *ph, int depth, int64_t first, int64_t count, double *out,
                           uint64_t cap) {
    Scene *s = (Scene *)sp;
    HashGrid ht(1, 200.0 / 768);
    Sink sink;
    double dummy_acc[3] = {0, 0, 0};
    uint32_t dummy_hit = 0;
    sink.acc = as_real(dummy_acc);
    sink.nhit = &dummy_hit;
    EventLog log;
    log.out = as_real(out);
    log.cap = cap;
    g_evlog = &log;
    V3 light = v3(ph->light);
    for (int64_t i = first; i < first + count; i++) {
        log.photon = i;
        Rng rng{cgrt_key(ph->seed, (uint64_t)i, 0, CGRT_PURPOSE_PHOTON), 0};
        double a = rng.u01() * (2 * ph->jitter) - ph->jitter;
        double b = rng.u01() * (2 * ph->jitter) - ph->jitter;
        V3 dir = sample_sphere(rng);
        trace_photon(*s, light + V3(a, 0, b), dir, V3(ph->power, ph->power, ph->power) * (PI_REF * 4.0), V3(1, 1, 1), depth, rng, ht,
                     ph->alpha, sink);
    }
    g_evlog = nullptr;
    return log.n;
}

// eye pass + serial photon pass + final gather; same signature and record layout as ref_ppm (cgrt_testapi.h)
int64_t orc_ppm(void *sp, const orc_camera *cam, const orc_grid *g, const orc_photons *ph, double *hp_out,
                uint64_t hp_cap, double *image_out) {
    Scene *s = (Scene *)sp;
    const int W = g->W, H = g->H;
    V3 camorg = v3(cam->cam);
    const double r0 = 200.0 / 768;  // main.cpp:84,183 with the reference's compile-time height
    HashGrid ht(ph->hashsize, r0);
    Sink sink;
    double dummy_acc[3] = {0, 0, 0};
    uint32_t dummy_hit = 0;
    sink.acc = as_real(dummy_acc);
    sink.nhit = &dummy_hit;
    for (int h = g->row0; h < g->row0 + g->nrows; h++)
        for (int w = 0; w < W; w++) {
            double x = (2.0 * ((double)w / W) - 1) * cam->half_width;
            double y = (2.0 * ((double)h / H) - 1) * cam->half_width * H / W;
            V3 dir = normalized(V3(x, y, 0) - camorg);
            V3 pof = dir * ((cam->focus_plane - camorg.z) / dir.z) + camorg;
            for (int j = g->sample0; j < g->sample0 + g->spp; j++) {
                RayCtx rc{g->seed, (uint64_t)h * (uint64_t)W + (uint64_t)w, (uint64_t)j};
                const int lw = w + W * (j - g->sample0), lh = h - g->row0;
                if (cam->lens_radius > 0) {
                    Rng r{cgrt_key(rc.seed, rc.pixel, rc.sample, 0), 0};
                    V3 neworg = camorg + lens_sample(r, cam->lens_radius);
                    V3 newdir = normalized(pof - neworg);
                    trace_store(*s, neworg, newdir, V3(1, 1, 1), g->depth, 1, rc, ht, lw, lh, r0, sink);
                } else {
                    trace_store(*s, camorg, dir, V3(1, 1, 1), g->depth, 1, rc, ht, lw, lh, r0, sink);
                }
            }
        }
    V3 light = v3(ph->light);
    for (int64_t i = 0; i < ph->nphotons; i++) {  // main.cpp:231-248, serial
        Rng rng{cgrt_key(ph->seed, (uint64_t)i, 0, CGRT_PURPOSE_PHOTON), 0};
        double a = rng.u01() * (2 * ph->jitter) - ph->jitter;
        double b = rng.u01() * (2 * ph->jitter) - ph->jitter;
        V3 dir = sample_sphere(rng);
        trace_photon(*s, light + V3(a, 0, b), dir, V3(ph->power, ph->power, ph->power) * (PI_REF * 4.0), V3(1, 1, 1), g->depth, rng,
                     ht, ph->alpha, sink);
    }
    int64_t k = 0;
    for (size_t b = 0; b < ht.buckets.size(); b++)
        for (size_t i = 0; i < ht.buckets[b].size(); i++) {
            const Hitpt &q = ht.buckets[b][i];
            const int col = q.w % W, smp = q.w / W;
            const size_t pix = (size_t)q.h * W + col;
            if (image_out) {
                V3 c = q.flux * (1.0 / (PI_REF * q.r2 * (double)ph->nphotons * g->spp));
                image_out[3 * pix] += c.x; image_out[3 * pix + 1] += c.y; image_out[3 * pix + 2] += c.z;
            }
            if (hp_out && (uint64_t)k < hp_cap) {
                double *o = hp_out + 16 * k;
                o[0] = (double)pix; o[1] = (double)smp;
                o[2] = q.f.x; o[3] = q.f.y; o[4] = q.f.z; o[5] = q.pos.x; o[6] = q.pos.y; o[7] = q.pos.z;
                o[8] = q.normal.x; o[9] = q.normal.y; o[10] = q.normal.z;
                o[11] = q.flux.x; o[12] = q.flux.y; o[13] = q.flux.z; o[14] = q.r2; o[15] = (double)q.n;
            }
            k++;
        }
    return k;
}

// tone map + flip, main.cpp:403-411 with gammaCorr (util.h:45-47): int(pow(1-exp(-x),1/2.2)*255+.5) stored to a byte.
// image: H*W*3 doubles, row 0 = bottom; out: H*W*3 bytes, top row first.
void orc_tonemap(const double *image, int W, int H, uint8_t *out) {
    size_t counter = 0;
    for (int i = 0; i < H; i++)
        for (int j = 0; j < W; j++) {
            const double *px = image + ((size_t)(H - i - 1) * W + j) * 3;
            for (int k = 0; k < 3; k++) out[3 * counter + k] = (uint8_t)(int)(std::pow(1 - std::exp(-px[k]), 1 / 2.2) * 255 + .5);
            counter++;
        }
}

#ifdef ORC_COUNT_FLOPS
// ---- instrumented build only (liborc_flops.so): the counters of cgrt_flopcount.h.  Counting runs use ONE thread
// (orc_set_threads(1), the default): the counters are thread-local and these two calls read the calling thread's. ----
void orc_flop_reset(void) { std::memset(&g_fc, 0, sizeof(g_fc)); }
// out[0..4] = additions/subtractions, multiplications, divisions, square roots, transcendental calls (pow sin cos atan exp)
void orc_flop_counts(uint64_t *out) {
    for (int k = 0; k < FC_N; k++) out[k] = g_fc.c[k];
}
#endif

}  // extern "C"
