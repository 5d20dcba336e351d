// libcgrt.so -- gfx950 (MI355X, CDNA4) kernels and the C ABI of include/cgrt.h.
//
// One kernel, trace_grid_kernel, replaces the reference's serial pixel/sample loop (main.cpp:185-219) and
// the recursive trace() under it (main.cpp:42-100,129-157):
//   * a workgroup = 4 wavefronts = a 32x8 pixel tile; each wave owns a 16x4 sub-tile, one pixel per lane;
//   * the top-level object list (`objs`, main.cpp:277) is staged once per workgroup in LDS and walked by all
//     64 lanes in lockstep (kind is wave-uniform, so the type dispatch is a scalar branch, not a virtual call);
//   * recursion is an explicit per-lane stack of pending refracted rays (<= 4 entries: depth budget 5) and a
//     lane that finishes a sample's ray tree immediately starts its next sample (persistent-lane loop);
//   * all geometry is fp64 with FMA contraction disabled, so hit decisions are the reference's decisions;
//   * the per-pixel accumulator is summed in fp64 in the reference's own order, scaled by 1/spp, rounded once
//     to fp32 and written through an LDS transpose as 384-byte contiguous row segments.
// Compile: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (see __graft_entry__.build()).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/cgrt.h"
#include "cgrt_build.h"
#include "cgrt_rng.hpp"
#include "cgrt_types.h"

using namespace cgrt;

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
// vec3.h:36-44
__device__ __forceinline__ V3 normalized(V3 a) {
    double len = sqrt(a.x * a.x + a.y * a.y + a.z * a.z);
    if (len > 0) {
        double r = 1 / len;
        a.x *= r;
        a.y *= r;
        a.z *= r;
    }
    return a;
}
// vec3.h:95-97, Sarrus with the reference's association
__device__ __forceinline__ double det3(V3 a, V3 b, V3 c) {
    return (a.x * b.y * c.z + b.x * c.y * a.z + c.x * a.y * b.z - a.x * c.y * b.z - b.x * a.y * c.z -
            c.x * b.y * a.z);
}
__device__ __forceinline__ V3 ld3(const double *p) { return mk(p[0], p[1], p[2]); }

// =====================================================================================================
// kernel parameters
// =====================================================================================================
struct GridParams {
    int32_t W, H, rows, row_offset, stripe_rows, stripe_rank, stripe_nranks;
    int32_t spp, sample_offset, max_depth;
    int32_t accumulate;  // CGRT_GRID_ACCUMULATE: rgb += this pass (nhit is overwritten)
    int32_t xcd_tiles;   // block -> tile mapping: 1 = XCD-aware super-tiles, 0 = row-major (see tile_of_block)
    double inv_spp_total;
    uint64_t seed;
    double cam[3], half_width, focus_plane, lens_radius;
};

static constexpr int kTileW = 32, kTileH = 8, kThreads = 256;
static constexpr int kMaxObjs = 96;  // top-level objects staged in LDS (12 KiB)
// Pending refracted rays (main.cpp:157) of a lane, newest last:
//   * a glass hit whose children are leaves of the recursion (depth_left == 2) keeps the refracted child in
//     REGISTERS (it is consumed right after the reflected child, before any other push) -- in a full glass tree
//     that is 8 of the 15 pushes;
//   * the first two other levels live in LDS: per level 9 doubles + one packed (depth, path) word per thread,
//     layout [level][field][thread] (conflict-free), 2 x 19 456 B = 38 912 B per workgroup;
//   * a third level (three nested glass hits with all siblings waiting) spills to scratch memory.
// The output tile aliases the stack (dead by then), so stack + objs stays under 40 KiB and FOUR workgroups fit a
// CU's 160 KiB: occupancy 4 waves/SIMD instead of 3, worth ~10 % on C2.
static constexpr int kPendDoubles = 9;
static constexpr int kLdsLevels = 2;
static constexpr size_t kLevelBytes = (size_t)kThreads * (kPendDoubles * sizeof(double) + sizeof(uint32_t));
static constexpr size_t kStackBytes = (size_t)kLdsLevels * kLevelBytes;
static constexpr size_t kTileBytes = (size_t)8 * 32 * 3 * sizeof(float);

// local row -> global row (cgrt.h: block-cyclic stripes)
__device__ __forceinline__ int global_row(const GridParams &g, int j) {
    if (g.stripe_nranks > 1) {
        int S = g.stripe_rows;
        return ((j / S) * g.stripe_nranks + g.stripe_rank) * S + (j % S);
    }
    return g.row_offset + j;
}

// blockIdx -> tile, XCD-aware.  Workgroups are dealt round-robin to the 8 XCDs, each with a private 4 MiB L2, so the
// blocks b, b+8, b+16, ... share an L2.  Tiles are grouped in super-tiles of kSuperW x kSuperH tiles (128 x 32 pixels);
// the blocks of one XCD group walk one super-tile after another, so the tiles an L2 serves at any moment are neighbours in
// the image and want the same tree nodes, triangles and texels -- while successive super-tiles alternate between the XCD
// groups, which keeps the expensive part of a frame (a mesh in one corner) spread over all of them.  Only placement
// changes: every tile is still rendered exactly once by exactly one workgroup.
// Measured (MI355X): C3 (glass bunny) 50.3 -> 46.6 ms, C4 (dragon) 205.5 -> 197.2 ms; but C2 4.05 -> 4.47 ms and the
// Bezier vase 8.1 -> 9.0 ms -- scenes with no tree to share, whose expensive tiles (glass sphere, vase) then sit on
// one or two XCDs.  So the launch picks it for scenes with meshes and no Bezier object, row-major otherwise.
static constexpr int kXcds = 8, kSuperW = 4, kSuperH = 4, kSuperTiles = kSuperW * kSuperH;
__host__ __device__ inline int tile_grid_blocks(int W, int rows, bool xcd_tiles) {
    const int tiles_x = (W + kTileW - 1) / kTileW, tiles_y = (rows + kTileH - 1) / kTileH;
    if (!xcd_tiles) return tiles_x * tiles_y;
    const int sx = (tiles_x + kSuperW - 1) / kSuperW, sy = (tiles_y + kSuperH - 1) / kSuperH;
    const int nsuper = sx * sy;
    return ((nsuper + kXcds - 1) / kXcds) * kXcds * kSuperTiles;
}
// false: this block has no tile (edge of the super-tile grid)
__device__ __forceinline__ bool tile_of_block(const GridParams &g, int &tile_x, int &tile_y) {
    const int tiles_x = (g.W + kTileW - 1) / kTileW, tiles_y = (g.rows + kTileH - 1) / kTileH;
    const int b = (int)blockIdx.x;
    if (!g.xcd_tiles) {
        tile_x = b % tiles_x;
        tile_y = b / tiles_x;
        return true;
    }
    const int sx = (tiles_x + kSuperW - 1) / kSuperW;
    const int group = b % kXcds, q = b / kXcds;
    const int super = (q / kSuperTiles) * kXcds + group, t = q % kSuperTiles;
    tile_x = (super % sx) * kSuperW + t % kSuperW;
    tile_y = (super / sx) * kSuperH + t / kSuperW;
    return tile_x < tiles_x && tile_y < tiles_y;
}

// =====================================================================================================
// tree traversal: KDTree::intersect_subtree / KDTree::intersect (objects.h:269-332)
// =====================================================================================================
// The reference visits BOTH children of every inner node whose box the ray touches, scans every leaf it
// reaches, and returns (a) the nearest triangle hit, ties resolved "first triangle inside a leaf, LAST leaf
// across leaves" (strict < at objects.h:281 and 297) and (b) the total number of times a leaf's running minimum
// improved, whose parity picks the normal's sign (objects.h:321-327).  Node numbering is preorder, so the
// recursion is a linear scan with skip links; no stack is needed.
//
// Box test.  KDNode::intersect (objects.h:166-200) intersects the ray with each face plane of the box and
// accepts when the crossing point lies within the face rectangle grown by 1e-4.  Whenever that accepts, the
// ray touches the box grown by 1e-4 on all sides at some t > 0, so a slab test on the grown box accepts too.
// Conversely a triangle of the node can only be hit at a point inside the un-grown box, where the reference's
// exit-face test accepts with 1e-4 of margin.  So the slab test visits a superset of the reference's nodes
// and the extra nodes cannot contain a hit: (len, triangle, counter) are identical.  The box is pre-grown by
// kBoxPad on the host; 1/d is computed once per ray.
//
// Triangle test.  Triangle::intersect (objects.h:96-111) divides four determinants and compares the
// quotients with 0 and 1.  For finite non-zero det1 the sign and "<= 1" tests on correctly rounded quotients
// are equivalent to the sign / magnitude tests below (DESIGN.md "triangle test"), so only an accepted hit pays
// for a division (len = det2/det1, the same correctly rounded quotient).
struct TreeHit {
    double len;
    int tri;      // absolute index into tris[]
    int counter;  // improvements (Q5)
};

//
// PRUNE (opaque objects only).  The improvement counter only decides the SIGN of the returned normal, and trace()
// re-orients the normal against the ray for every material (main.cpp:73-76); diffuse and mirror shading use
// nothing else of it, so for an object whose transparency is < eps the counter cannot influence the image or
// the Hitpoint records (the one exception is a ray exactly tangent to the winning triangle,
// n.d == 0, where no re-orientation happens).  Then only the nearest hit matters, and a subtree whose box the
// ray enters beyond `bound` -- the nearest hit known so far, in this tree or among the objects tested before
// it -- can be skipped: a hit inside it would lose the strict `len < nearest` tests (objects.h:281,297;
// main.cpp:57).  Entry distances come from the grown, outward-rounded boxes, so they never exceed the true
// ones and a leaf holding a triangle that ties with the bound is still scanned: (len, triangle) stay exact.
//
// TRI (implies PRUNE): the hierarchy goes down to groups of <= 4 single triangles (`otris`, each carrying its place in
// the reference's leaf order); with no leaf-wide scan order to rely on, every accepted hit is compared on
// (len, reference leaf, index) directly -- the same winner as "first inside a leaf, last leaf across leaves".
template <bool STATS, bool PRUNE, bool TRI = false>
__device__ __forceinline__ TreeHit tree_intersect(const NodeRec *__restrict__ nodes, const TriRec *__restrict__ tris,
                                                  int nnodes, V3 o, V3 d, V3 inv, double bound, uint32_t &n_node,
                                                  uint32_t &n_tri, const OTriRec *__restrict__ otris = nullptr) {
    TreeHit r;
    r.len = kInf;
    r.tri = -1;
    r.counter = 0;
    int r_leaf = -1;  // first triangle index of the leaf holding r.tri (grows with the reference's leaf sequence)
    int i = 0;
    // "while-while" traversal: every lane first walks inner nodes until it stands on a leaf its ray touches
    // (or runs out of nodes); only then does the wave scan leaves, so the ~800-instruction leaf scan runs once
    // per leaf-visit round with many lanes active instead of once per node step with one or two.
    while (true) {
        int leaf_begin = 0, leaf_cnt_tris = -1;
        while (i < nnodes) {
            // one 32-byte record = two 16-byte loads
            const float4 q0 = reinterpret_cast<const float4 *>(nodes + i)[0];  // lo.x lo.y lo.z hi.x
            const float4 q1 = reinterpret_cast<const float4 *>(nodes + i)[1];  // hi.y hi.z skip leaf
            if (STATS) n_node++;
            double t1, t2, tn, tf;
            t1 = ((double)q0.x - o.x) * inv.x;
            t2 = ((double)q0.w - o.x) * inv.x;
            tn = fmin(t1, t2);
            tf = fmax(t1, t2);
            t1 = ((double)q0.y - o.y) * inv.y;
            t2 = ((double)q1.x - o.y) * inv.y;
            tn = fmax(tn, fmin(t1, t2));
            tf = fmin(tf, fmax(t1, t2));
            t1 = ((double)q0.z - o.z) * inv.z;
            t2 = ((double)q1.y - o.z) * inv.z;
            tn = fmax(tn, fmin(t1, t2));
            tf = fmin(tf, fmax(t1, t2));
            const bool touch = (tf > 0.0) && (tn <= tf) && !(PRUNE && tn > bound);
            const int leaf = __float_as_int(q1.w);
            if (!touch) {
                i = __float_as_int(q1.z);
                continue;
            }
            i = i + 1;  // inner: left child is next in preorder; leaf: its skip is i+1 too
            if (leaf < 0) continue;
            leaf_begin = leaf >> 4;
            leaf_cnt_tris = leaf & 15;
            break;
        }
        if (leaf_cnt_tris < 0) break;  // no further leaf for this lane
        if (TRI) {
            const OTriRec *tp = otris + leaf_begin;
            for (int k = 0; k < leaf_cnt_tris; k++) {
                if (STATS) n_tri++;
                const V3 pa = ld3(tp[k].t.pa), e1 = ld3(tp[k].t.e1), e2 = ld3(tp[k].t.e2);
                const int2 rank = *reinterpret_cast<const int2 *>(&tp[k].k);  // k, leaf
                const V3 s = pa - o;
                const double det1 = det3(d, e1, e2);
                const double det2 = det3(s, e1, e2);
                const double det3_ = det3(d, s, e2);
                const double det4 = det3(d, e1, s);
                const double sg = det1 > 0.0 ? 1.0 : -1.0;
                const double a1 = det1 * sg;
                const bool ok = (det1 != 0.0) && (det2 * sg > 0.0) && (det3_ * sg >= 0.0) && (det4 * sg >= 0.0) &&
                                ((det3_ + det4) * sg <= a1);
                if (ok) {
                    const double len = det2 / det1;
                    if (len < r.len || (len == r.len && (rank.y > r_leaf || (rank.y == r_leaf && rank.x < r.tri)))) {
                        r.len = len;
                        r.tri = rank.x;
                        r_leaf = rank.y;
                        r.counter = 1;
                    }
                }
            }
            if (r.len < bound) bound = r.len;
            continue;
        }
        // leaf scan, objects.h:273-289
        double leaf_len = kInf;
        int leaf_tri = -1, leaf_cnt = 0;
        const TriRec *tp = tris + leaf_begin;
        // one triangle ahead: the next record is requested before the current one is tested (every request is used
        // except the repeat of the last one, so this adds no traffic)
        V3 pa = ld3(tp[0].pa), e1 = ld3(tp[0].e1), e2 = ld3(tp[0].e2);
        for (int k = 0; k < leaf_cnt_tris; k++) {
            if (STATS) n_tri++;
            const int kn = (k + 1 < leaf_cnt_tris) ? k + 1 : k;
            const V3 npa = ld3(tp[kn].pa), ne1 = ld3(tp[kn].e1), ne2 = ld3(tp[kn].e2);
            const V3 s = pa - o;
            const double det1 = det3(d, e1, e2);
            const double det2 = det3(s, e1, e2);
            const double det3_ = det3(d, s, e2);
            const double det4 = det3(d, e1, s);
            const double sg = det1 > 0.0 ? 1.0 : -1.0;
            const double a1 = det1 * sg;
            const bool ok = (det1 != 0.0) && (det2 * sg > 0.0) && (det3_ * sg >= 0.0) && (det4 * sg >= 0.0) &&
                            ((det3_ + det4) * sg <= a1);
            if (ok) {
                const double len = det2 / det1;
                if (len < leaf_len) {
                    leaf_len = len;
                    leaf_tri = leaf_begin + k;
                    leaf_cnt++;
                }
            }
            pa = npa;
            e1 = ne1;
            e2 = ne2;
        }
        if (leaf_cnt > 0) {
            // objects.h:295-313: the left result survives only if strictly nearer => the LATER leaf of the reference's
            // sequence wins ties.  Spelled out on the leaf's position, because the hierarchy above the leaves need not
            // visit them in the reference's order.
            if (r.counter == 0 || leaf_len < r.len || (leaf_len == r.len && leaf_begin > r_leaf)) {
                r.len = leaf_len;
                r.tri = leaf_tri;
                r_leaf = leaf_begin;
            }
            r.counter += leaf_cnt;
            if (PRUNE && r.len < bound) bound = r.len;
        }
    }
    return r;
}

// =====================================================================================================
// Opaque bump floors: the displacement mesh as a height field
// =====================================================================================================
// The reference pushes every floor-bound ray through its object-median tree over the bump mesh -- for the stone
// floor 192 node tests and 179 triangle tests per ray on average, because both children are always visited and
// preorder is not front to back.  The mesh is a regular grid of quads in x-z (objects.h:485-497), so for an OPAQUE
// floor (where only the nearest hit matters, see PRUNE above) the ray is clipped to the slab of heights the mesh
// occupies and walked column by column along its major horizontal axis; the two triangles of every cell the ray's
// footprint touches are tested with the SAME triangle records and the SAME test as the tree's leaves, so an accepted
// hit has the same `len`, bit for bit.  A hit point lies inside its triangle, hence over its cell, so the walk --
// padded by kHfPad on every side against rounding -- meets every triangle the ray can hit; columns are visited in
// ray order and the walk stops at the first column that begins beyond the nearest hit.  Exact ties (a ray through
// a shared edge) are resolved as the tree resolves them: the triangle in the LATER leaf wins, inside a leaf the
// EARLIER one (objects.h:281,297) -- each cell record carries its triangles' leaf number and leaf-order index.
static constexpr double kHfPad = 1e-9;

template <bool STATS>
__device__ __forceinline__ TreeHit hfield_intersect(const HFieldRec &H, const HCellRec *__restrict__ cells, V3 o, V3 d,
                                                    V3 inv, double bound, uint32_t &n_node, uint32_t &n_tri) {
    TreeHit r;
    r.len = kInf;
    r.tri = -1;
    r.counter = 0;
    int best_leaf = -1;
    // clip the ray's parameter range (0, bound] to the padded box of the mesh
    double ta = 0.0, tb = bound;
    const double lo[3] = {H.x0 - kHfPad, H.ylo - kHfPad, H.z0 - kHfPad};
    const double hi[3] = {H.x0 + H.hx * H.nx + kHfPad, H.yhi + kHfPad, H.z0 + H.hz * H.nz + kHfPad};
    const double oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z}, ii[3] = {inv.x, inv.y, inv.z};
#pragma unroll
    for (int k = 0; k < 3; k++) {
        if (dd[k] != 0.0) {
            const double t1 = (lo[k] - oo[k]) * ii[k], t2 = (hi[k] - oo[k]) * ii[k];
            ta = fmax(ta, fmin(t1, t2));
            tb = fmin(tb, fmax(t1, t2));
        } else if (oo[k] < lo[k] || oo[k] > hi[k]) {
            tb = -1.0;
        }
    }
    if (!(ta <= tb)) return r;
    // major axis: the one along which the ray crosses more cells
    const bool xmaj = fabs(d.x) * H.hz >= fabs(d.z) * H.hx;
    const double oM = xmaj ? o.x : o.z, dM = xmaj ? d.x : d.z, iM = xmaj ? inv.x : inv.z;
    const double om = xmaj ? o.z : o.x, dm = xmaj ? d.z : d.x;
    const double M0 = xmaj ? H.x0 : H.z0, hM = xmaj ? H.hx : H.hz, m0 = xmaj ? H.z0 : H.x0, hm = xmaj ? H.hz : H.hx;
    const int nM = xmaj ? H.nx : H.nz, nm = xmaj ? H.nz : H.nx;
    const double ihM = 1.0 / hM, ihm = 1.0 / hm;
    const double Ma = oM + dM * ta, Mb = oM + dM * tb;
    int j0 = (int)floor((fmin(Ma, Mb) - kHfPad - M0) * ihM), j1 = (int)floor((fmax(Ma, Mb) + kHfPad - M0) * ihM);
    j0 = j0 < 0 ? 0 : j0;
    j1 = j1 > nM - 1 ? nM - 1 : j1;
    const int step = dM >= 0.0 ? 1 : -1;
    const int jn = j1 - j0 + 1;  // columns to visit (<= 0: none)
    int j = step > 0 ? j0 : j1;
    for (int c = 0; c < jn; c++, j += step) {
        // parameter range of the padded column, within [ta, tb]
        double s0 = ta, s1 = tb;
        if (dM != 0.0) {
            const double t1 = (M0 + hM * j - kHfPad - oM) * iM, t2 = (M0 + hM * (j + 1) + kHfPad - oM) * iM;
            s0 = fmax(s0, fmin(t1, t2));
            s1 = fmin(s1, fmax(t1, t2));
        }
        if (s0 > r.len) break;   // this column and all later ones begin beyond the nearest hit
        if (!(s0 <= s1)) continue;
        const double ma = om + dm * s0, mb = om + dm * s1;
        int i0 = (int)floor((fmin(ma, mb) - kHfPad - m0) * ihm), i1 = (int)floor((fmax(ma, mb) + kHfPad - m0) * ihm);
        i0 = i0 < 0 ? 0 : i0;
        i1 = i1 > nm - 1 ? nm - 1 : i1;
        for (int i = i0; i <= i1; i++) {
            if (STATS) n_node++;
            const HCellRec *cell = cells + (xmaj ? (size_t)i * H.nx + j : (size_t)j * H.nx + i);
            const int4 ids = *reinterpret_cast<const int4 *>(cell->k);  // k0 k1 leaf0 leaf1
#pragma unroll
            for (int q = 0; q < 2; q++) {
                if (STATS) n_tri++;
                const V3 pa = ld3(cell->t[q].pa), e1 = ld3(cell->t[q].e1), e2 = ld3(cell->t[q].e2);
                const V3 s = pa - o;
                const double det1 = det3(d, e1, e2);
                const double det2 = det3(s, e1, e2);
                const double det3_ = det3(d, s, e2);
                const double det4 = det3(d, e1, s);
                const double sg = det1 > 0.0 ? 1.0 : -1.0;
                const double a1 = det1 * sg;
                const bool ok = (det1 != 0.0) && (det2 * sg > 0.0) && (det3_ * sg >= 0.0) && (det4 * sg >= 0.0) &&
                                ((det3_ + det4) * sg <= a1);
                if (ok) {
                    const double len = det2 / det1;
                    const int k = q ? ids.y : ids.x, leaf = q ? ids.w : ids.z;
                    if (len < r.len || (len == r.len && (leaf > best_leaf || (leaf == best_leaf && k < r.tri)))) {
                        r.len = len;
                        r.tri = k;
                        best_leaf = leaf;
                        r.counter = 1;
                    }
                }
            }
        }
    }
    return r;
}

// normal of the winning triangle, oriented by the improvement-counter parity (objects.h:107,321-327)
__device__ __forceinline__ V3 tree_normal(const TriRec *__restrict__ tris, const TreeHit &h, V3 d) {
    const V3 e1 = ld3(tris[h.tri].e1), e2 = ld3(tris[h.tri].e2);
    V3 n = normalized(cross(e1, e2));
    const bool facing = dot(n, d) < 0;
    if ((h.counter & 1) == 0) {
        if (!facing) n = -n;  // origin outside: normal against the ray
    } else {
        if (facing) n = -n;  // origin inside: normal along the ray
    }
    return n;
}

// =====================================================================================================
// Texture::color, texture.h:39-72 (nearest texel, three axis-aligned orientations, d.x tested first)
// =====================================================================================================
__device__ __forceinline__ bool texture_color(const TexRec &t, const uint8_t *__restrict__ texels, V3 point, V3 &out) {
    V3 d = point - ld3(t.p);
    const V3 n = ld3(t.n);
    d = d - n * dot(d, n);
    const double te = 1e-2;  // texture.h:12
    const int rows = t.rows, cols = t.cols;
    int r, c;
    if (d.x < te && d.x > -te) {
        if (!(0 < d.y && d.y < t.lenx && 0 < d.z && d.z < t.leny)) return false;
        r = (int)floor(d.y / t.lenx * rows);
        c = (int)floor(d.z / t.leny * cols);
    } else if (d.y < te && d.y > -te) {
        if (!(0 < d.x && d.x < t.lenx && 0 < d.z && d.z < t.leny)) return false;
        c = (int)floor(d.x / t.lenx * cols);
        r = (int)floor(d.z / t.leny * rows);
    } else if (d.z < te && d.z > -te) {
        if (!(0 < d.x && d.x < t.lenx && 0 < d.y && d.y < t.leny)) return false;
        c = (int)floor(d.x / t.lenx * cols);
        r = rows - 1 - (int)floor(d.y / t.leny * rows);
    } else {
        return false;
    }
    // the reference indexes unchecked; an index equal to rows/cols can only arise from rounding at the far edge
    r = r < 0 ? 0 : (r >= rows ? rows - 1 : r);
    c = c < 0 ? 0 : (c >= cols ? cols - 1 : c);
    const uint8_t *px = texels + t.texel_begin + 3 * ((int64_t)r * cols + c);
    out = mk((double)px[0] / 256.0, (double)px[1] / 256.0, (double)px[2] / 256.0);  // main.cpp:307-311
    return true;
}


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
__device__ __forceinline__ double norm3(V3 v) { return sqrt(v.x * v.x + v.y * v.y + v.z * v.z); }

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
    const bool ordinary = fabs(n0) < 1e300 && fabs(n1) < 1e300 && fabs(n2) < 1e300 && fabs(n3) < 1e300 && fabs(n4) < 1e300 &&
                          fabs(n5) < 1e300 && fabs(n6) < 1e300 && fabs(n7) < 1e300 && fabs(n8) < 1e300;
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
    return norm3(st.fv) < 1e-4 && st.res.x > 0 && st.res.y <= 1 && st.res.y >= 0;
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
        while (norm3(st.fv) > 1e-6 && st.counter < 100) {
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
__device__ bool bezier_wave(const BezierRec &b, V3 pos, double cap_r, bool on, V3 o, V3 d, uint64_t key, uint32_t &n0,
                            double &len, V3 &n, volatile BezLds *L) {
    const int lane = threadIdx.x & 63;
    const bool want = on && bez_box(b, o, d);
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
                if (norm3(st.fv) > 1e-6 && st.counter < 100) {
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

// =====================================================================================================
// nearest hit over objs (main.cpp:55-63) -- all lanes walk the LDS-resident list in lockstep
// =====================================================================================================
struct SceneHit {
    double t;
    int id;  // -1: miss
    V3 n;    // geometric normal as the object's intersect() returns it (before main.cpp:73-76)
};

// identifies a ray for the keyed Bezier stream: purpose_key(k_smp, (path << 16) | (object + 1)), or, for the
// function-level probe, an explicit key
struct RayKey {
    uint64_t k;  // the sample's key k_smp (cgrt_rng.hpp), or the explicit stream key
    uint32_t path;
    bool explicit_key;
    uint32_t n0;  // explicit key only: position in the stream; advanced by the draws consumed (photon pass)
};

// Sphere::intersect, objects.h:45-68: the hit distance, or +inf-like kInf (never < nearest) on a miss
__device__ __forceinline__ double sphere_len(const ObjRec &ob, V3 o, V3 d) {
    const V3 l = ld3(ob.a) - o;
    const double tca = dot(l, d);
    const double l2 = dot(l, l);
    const double r2 = ob.s0;
    double len = kInf;
    if (!(tca < 0 && l2 > r2)) {
        const double d2 = l2 - tca * tca;
        if (!(d2 > r2)) {
            const double thc = sqrt(r2 - d2);
            const double t0 = tca - thc, t1 = tca + thc;
            len = (t0 < 0) ? t1 : t0;
        }
    }
    return len;
}

// One small tree (<= kNodeCache nodes: the bunny's 255, a coarse bump floor) is staged whole in LDS by every workgroup:
// traversal is latency-bound on dependent node fetches, and an LDS read costs ~100 cycles against ~500-800 for L1/L2.
static constexpr int kNodeCache = 256;  // 8 KiB of 32-byte nodes

// per-workgroup LDS resources handed down to the scene walk
struct LdsAux {
    volatile BezLds *bl;    // this wave's Bezier scratch (BEZ variants) or nullptr
    const NodeRec *lnodes;  // LDS copy of tree sc.cached_tree's nodes, or nullptr
};

// tree traversal entry; `on` = this lane really has a ray for this tree (all lanes of the wave call it).
// A wave-synchronous variant (one shared node sequence, records fetched through the scalar cache) was measured
// and dropped: with sub-pixel triangles the union of 64 rays' leaf sets approaches their sum.
// `opaque` (wave-uniform): the owning object's transparency is < eps, so the pruned traversal applies with
// `bound` = nearest hit distance already known for this ray.
template <bool STATS>
__device__ __forceinline__ TreeHit tree_hit(const DeviceScene &sc, const LdsAux &aux, int tr, bool opaque, double bound,
                                            bool on, V3 o, V3 d, V3 inv, uint32_t &n_node, uint32_t &n_tri) {
    const TreeRec T = sc.trees[tr];
    TreeHit none;
    none.len = kInf;
    none.tri = -1;
    none.counter = 0;
    if (!on) return none;
    if (opaque && T.hfield >= 0) {  // bump floor: walk the grid instead of the tree (same triangles, same test)
        const HFieldRec H = sc.hfields[T.hfield];
        return hfield_intersect<STATS>(H, sc.hcells + H.cell_begin, o, d, inv, bound, n_node, n_tri);
    }
    const bool cached = aux.lnodes != nullptr && tr == sc.cached_tree;
    // the copy of the hierarchy whose children are ordered near-to-far for this ray's direction octant (only worth a
    // per-lane base address where order matters, i.e. for the pruned traversal)
    const int oct = (T.noct == 8 && opaque) ? ((d.x < 0 ? 1 : 0) | (d.y < 0 ? 2 : 0) | (d.z < 0 ? 4 : 0)) : 0;
    const NodeRec *nodes = sc.nodes + T.node_begin + (size_t)oct * (size_t)T.nnodes;
    const TriRec *tris = sc.tris + T.tri_begin;
    if (opaque) {
        if (T.tri_level) {
            const OTriRec *ot = sc.otris + T.otri_begin;
            if (cached) return tree_intersect<STATS, true, true>(aux.lnodes, tris, T.nnodes, o, d, inv, bound, n_node, n_tri, ot);
            return tree_intersect<STATS, true, true>(nodes, tris, T.nnodes, o, d, inv, bound, n_node, n_tri, ot);
        }
        if (cached) return tree_intersect<STATS, true>(aux.lnodes, tris, T.nnodes, o, d, inv, bound, n_node, n_tri);
        return tree_intersect<STATS, true>(nodes, tris, T.nnodes, o, d, inv, bound, n_node, n_tri);
    }
    if (cached) return tree_intersect<STATS, false>(aux.lnodes, tris, T.nnodes, o, d, inv, bound, n_node, n_tri);
    return tree_intersect<STATS, false>(nodes, tris, T.nnodes, o, d, inv, bound, n_node, n_tri);
}

template <bool TREES, bool BEZ, bool SPH, bool STATS>
__device__ __forceinline__ SceneHit intersect_scene(const ObjRec *__restrict__ objs, int n_objs, const DeviceScene &sc,
                                                    V3 o, V3 d, RayKey &rk, bool on, const LdsAux &aux,
                                                    uint32_t &n_node, uint32_t &n_tri) {
    SceneHit best;
    best.t = kInf;  // `nearest = INF`, main.cpp:54
    best.id = -1;
    best.n = mk(0, 0, 0);
    int nsrc = 0;  // 0: sphere (normal derived after the loop), 1: stored in best.n
    if (SPH) {
        // scenes made of spheres only: no kind dispatch, nothing but (t, id) carried round the loop
        for (int i = 0; i < n_objs; i++) {
            const double len = sphere_len(objs[i], o, d);
            if (len < best.t) {
                best.t = len;
                best.id = i;
            }
        }
        if (best.id >= 0) best.n = normalized((o + d * best.t) - ld3(objs[best.id].a));  // objects.h:65-66
        return best;
    }
    V3 inv = mk(0, 0, 0);
    if (TREES) inv = mk(1.0 / d.x, 1.0 / d.y, 1.0 / d.z);
    for (int i = 0; i < n_objs; i++) {
        const ObjRec &ob = objs[i];
        const int kind = __builtin_amdgcn_readfirstlane(ob.kind);
        if (kind == KIND_SPHERE) {
            const double len = sphere_len(ob, o, d);
            if (len < best.t) {
                best.t = len;
                best.id = i;
                nsrc = 0;
            }
        } else if (kind == KIND_PLANE) {
            // Plane::intersect, objects.h:505-524
            const V3 pn = ld3(ob.b);
            const V3 dd = ld3(ob.a) - o;
            double len = dot(dd, pn) / dot(d, pn);
            const bool ph = len > 0;
            V3 nrm = pn;
            if (TREES) {
                const int tr = __builtin_amdgcn_readfirstlane(ob.tree);
                const bool want = on && ph;  // the bump tree is only consulted when the plane is hit (objects.h:508-513)
                if (tr >= 0 && __ballot(want) != 0ull) {
                    // a bump hit only counts if it is nearer than the plane itself (objects.h:514) and, to matter,
                    // nearer than the nearest object so far
                    const bool opaque = __builtin_amdgcn_readfirstlane((int)(ob.transp < kEps)) != 0;
                    const TreeHit h = tree_hit<STATS>(sc, aux, tr, opaque, fmin(len, best.t), want, o, d, inv, n_node, n_tri);
                    if (want && h.counter > 0 && h.len < len && h.len > 0) {
                        len = h.len;
                        nrm = tree_normal(sc.tris + sc.trees[tr].tri_begin, h, d);
                    }
                }
            }
            if (ph && len < best.t) {
                best.t = len;
                best.id = i;
                best.n = nrm;
                nsrc = 1;
            }
        } else if (TREES && kind == KIND_MESH) {
            // TriangleMesh::intersect, objects.h:405-455
            const int tr = __builtin_amdgcn_readfirstlane(ob.tree);
            if (__ballot(on) != 0ull) {
                const bool opaque = __builtin_amdgcn_readfirstlane((int)(ob.transp < kEps)) != 0;
                const TreeHit h = tree_hit<STATS>(sc, aux, tr, opaque, best.t, on, o, d, inv, n_node, n_tri);
                if (on && h.counter > 0 && h.len < best.t) {
                    V3 nrm = tree_normal(sc.tris + sc.trees[tr].tri_begin, h, d);
                    if (ob.aux == 2) nrm = (nrm.y > 0) ? nrm : -nrm;  // objects.h:434-436
                    best.t = h.len;
                    best.id = i;
                    best.n = nrm;
                    nsrc = 1;
                }
            }
        } else if (BEZ && kind == KIND_BEZIER) {
            const BezierRec &bz = sc.beziers[__builtin_amdgcn_readfirstlane(ob.aux)];
            const uint64_t key = rk.explicit_key ? rk.k : purpose_key(rk.k, ((uint64_t)rk.path << 16) | (uint64_t)(i + 1));
            double len = 0;
            V3 nrm = best.n;  // the reference passes its running `temp` normal (main.cpp:53,56)
            uint32_t n0 = rk.explicit_key ? rk.n0 : 0u;
            const bool bh = bezier_wave(bz, ld3(ob.a), ob.b[0], on, o, d, key, n0, len, nrm, aux.bl);
            if (rk.explicit_key) rk.n0 = n0;
            if (bh) {
                if (len < best.t) {
                    best.t = len;
                    best.id = i;
                    best.n = nrm;
                    nsrc = 1;
                }
            }
        }
    }
    if (best.id >= 0 && nsrc == 0) {
        const V3 p = o + d * best.t;  // objects.h:65-66
        best.n = normalized(p - ld3(objs[best.id].a));
    }
    return best;
}

// =====================================================================================================
// the eye pass
// =====================================================================================================
struct Pending {  // a refracted child waiting for its turn (main.cpp:157)
    V3 o, d, adj;
    int32_t depth_left;
    uint32_t path;
};

// GLASS: the scene contains a transparent object, so refracted children can be pending; without it the
// pending-ray storage (LDS levels, sibling registers) is compiled out and occupancy goes up.
// SPH: every object is a sphere (C1/C2-type scenes): specialised object loop.
// HPS: additionally append every Hitpoint {f, pos, normal} (hitpoints.h:6-20, main.cpp:87-98) to a global stream.
struct HitpointSink {
    double *rec;                // cap x 10 doubles: f(3) pos(3) normal(3) label
    unsigned long long *count;  // appended so far (may exceed cap: then the tail was dropped)
    unsigned long long cap;
};

template <bool TREES, bool BEZ, bool DOF, bool GLASS, bool SPH, bool STATS, bool HPS = false>
__global__ __launch_bounds__(kThreads, BEZ ? 3 : ((GLASS && TREES) ? 3 : 4)) void trace_grid_kernel(DeviceScene sc, GridParams g, float *__restrict__ rgb,
                                                             uint32_t *__restrict__ nhit_out,
                                                             unsigned long long *__restrict__ counters,
                                                             HitpointSink hps = HitpointSink{nullptr, nullptr, 0}) {
    // LDS carve-up: [ pending-ray levels (GLASS) -- aliased by the output tile at the end | objs ]
    extern __shared__ __align__(16) unsigned char lds_raw[];
    float *ltile = reinterpret_cast<float *>(lds_raw);
    ObjRec *lobjs = reinterpret_cast<ObjRec *>(lds_raw + (GLASS ? kStackBytes : kTileBytes));  // n_objs records
    // BEZ: one BezLds per wave behind the object list (16-byte aligned: ObjRec is 128 B)
    unsigned char *lrest = reinterpret_cast<unsigned char *>(lobjs + sc.n_objs);
    LdsAux aux;
    aux.bl = BEZ ? reinterpret_cast<volatile BezLds *>(lrest) + (threadIdx.x >> 6) : nullptr;
    if (BEZ) lrest += (kThreads / 64) * sizeof(BezLds);
    // TREES: node cache behind that (32-byte records, region is 16-byte aligned)
    NodeRec *lnodes = reinterpret_cast<NodeRec *>(lrest);
    aux.lnodes = (TREES && sc.cached_tree >= 0) ? lnodes : nullptr;
    if (TREES && sc.cached_tree >= 0) {
        const uint4 *src = reinterpret_cast<const uint4 *>(sc.nodes + sc.trees[sc.cached_tree].node_begin);
        uint4 *dst = reinterpret_cast<uint4 *>(lnodes);
        const int n16 = sc.cached_nodes * (int)(sizeof(NodeRec) / 16);
        for (int k = threadIdx.x; k < n16; k += kThreads) dst[k] = src[k];
    }

    // stage the primitive list in LDS (128 B records, copied as 16-byte pieces)
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(sc.objs);
        uint4 *dst = reinterpret_cast<uint4 *>(lobjs);
        const int n16 = sc.n_objs * (int)(sizeof(ObjRec) / 16);
        for (int k = threadIdx.x; k < n16; k += kThreads) dst[k] = src[k];
    }
    __syncthreads();

    int tile_x, tile_y;
    if (!tile_of_block(g, tile_x, tile_y)) return;  // whole workgroup (after the barrier above; no further barrier is missed)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // wave = 16x4 pixels, 2x2 waves per workgroup (8x8 per wave measured: meshes equal, C2 7 % slower)
    const int lx = (wave & 1) * 16 + (lane & 15);
    const int ly = (wave >> 1) * 4 + (lane >> 4);
    const int w = tile_x * kTileW + lx;
    const int j = tile_y * kTileH + ly;  // local row
    const int h = global_row(g, j);
    const bool live = (w < g.W) && (j < g.rows) && (h < g.H);

    const V3 camorg = mk(g.cam[0], g.cam[1], g.cam[2]);
    // main.cpp:188-189,198,203
    const double px = (2.0 * ((double)w / g.W) - 1) * g.half_width;
    const double py = (2.0 * ((double)h / g.H) - 1) * g.half_width * g.H / g.W;
    const V3 pdir = normalized(mk(px, py, 0) - camorg);
    const V3 pof = pdir * ((g.focus_plane - camorg.z) / pdir.z) + camorg;
    const uint64_t k_pix = pixel_key(g.seed, (uint64_t)h * (uint64_t)g.W + (uint64_t)w);
    uint64_t k_smp = 0;  // key of the sample whose ray tree this lane is tracing

    double acc_r = 0, acc_g = 0, acc_b = 0;
    uint32_t my_hits = 0, my_rays = 0, my_nodes = 0, my_tris = 0, wave_iters = 0;
    uint32_t hp_seq = 0;  // HPS: index of the next Hitpoint within the current sample's ray tree (emission order)

    Pending deep[2];   // third stack level (scratch; indexed dynamically so that it stays out of registers)
    Pending sib;       // refracted sibling of a leaf-level glass hit (registers)
    bool sib_valid = false;
    unsigned char *lslot = lds_raw;  // level L, field f of this thread: lslot + L*kLevelBytes + (f*256 + tid)*8
    int sp = 0;
    int s = 0;  // next sample to start
    bool have = false;
    V3 o = camorg, d = pdir, adj = mk(1, 1, 1);
    int depth_left = 0;
    uint32_t path = 1;

    while (true) {
        if (!have) {
            if (live && s < g.spp) {
                // start the next sample of this lane's pixel (main.cpp:204-209)
                k_smp = sample_key(k_pix, (uint64_t)(g.sample_offset + s));
                if (DOF) {
                    Stream rs(k_smp);  // purpose 0: the lens stream's key is the sample key
                    double sx, sy;
                    while (true) {  // uniform_sampling_circle, sampling.h:35-43
                        double ux, uy;
                        rs.pair(ux, uy);
                        sx = ux * 2.0 - 1;
                        sy = uy * 2.0 - 1;
                        if (sx * sx + sy * sy < 1) break;
                    }
                    o = camorg + mk(sx, sy, 0) * g.lens_radius;
                    d = normalized(pof - o);
                } else {
                    o = camorg;
                    d = pdir;
                }
                adj = mk(1, 1, 1);
                depth_left = g.max_depth;
                path = 1;
                s++;
                hp_seq = 0;
                have = true;
            }
        }
        if (__ballot(have) == 0ull) break;  // every lane of the wave has drained its pixel
        wave_iters++;
        // All 64 lanes enter the scene walk together (lanes without a ray carry on == false): the object list
        // is wave-uniform, so its control flow stays scalar.
        RayKey rk{k_smp, path, false, 0u};
        const SceneHit hit =
            intersect_scene<TREES, BEZ, SPH, STATS>(lobjs, sc.n_objs, sc, o, d, rk, have, aux, my_nodes, my_tris);
        if (have) {
            my_rays++;
            have = false;
            if (hit.id >= 0) {
                const ObjRec &ob = lobjs[hit.id];
                const V3 P = o + d * hit.t;  // main.cpp:68
                V3 n = hit.n;
                const V3 n_old = n;
                bool into = true;
                if (dot(n, d) > 0) {  // main.cpp:73-76
                    n = -n;
                    into = false;
                }
                V3 f = ld3(ob.col);  // getSurfaceColor
                if (ob.kind == KIND_PLANE && ob.tex >= 0) {
                    V3 c;
                    if (texture_color(sc.texs[ob.tex], sc.texels, P, c)) f = c;  // objects.h:533-539
                }
                const double refl = ob.refl, transp = ob.transp;
                if (refl < kEps && transp < kEps) {
                    // diffuse: the reference stores Hitpoint{f*adj,...} (main.cpp:85-100); we accumulate it
                    const V3 hf = mulv(f, adj);
                    acc_r += hf.x;
                    acc_g += hf.y;
                    acc_b += hf.z;
                    my_hits++;
                    if (HPS) {
                        const unsigned long long k = atomicAdd(hps.count, 1ull);
                        if (k < hps.cap) {
                            double *q = hps.rec + 10 * k;
                            q[0] = hf.x; q[1] = hf.y; q[2] = hf.z;
                            q[3] = P.x; q[4] = P.y; q[5] = P.z;
                            q[6] = n.x; q[7] = n.y; q[8] = n.z;
                            // label: (sample, local pixel) like Hitpoint::w/h (main.cpp:91-92), times 16, plus the
                            // hitpoint's position in the sample's emission order (<= 16 per tree)
                            q[9] = (double)((((unsigned long long)(s - 1) * (unsigned long long)g.W * g.rows +
                                              (unsigned long long)j * g.W + w) << 4) | (unsigned long long)hp_seq);
                        }
                        hp_seq++;
                    }
                } else if (depth_left > 1) {
                    if (transp < kEps) {
                        // mirror, main.cpp:129-134
                        const V3 nd = d - n * 2.0 * dot(n, d);
                        adj = mulv(f, adj) * refl;
                        o = P + n * kEps;
                        d = nd;
                        depth_left--;
                        path = path * 2;
                        have = true;
                    } else if (GLASS) {
                        // glass, main.cpp:135-157
                        const double nc = 1.0, nt = 1.33;
                        const double nnt = into ? nc / nt : nt / nc;
                        const double ddn = dot(d, n);
                        const V3 refl_dir = d - n_old * 2.0 * dot(n_old, d);
                        const double cos2t = 1 - nnt * nnt * (1 - ddn * ddn);
                        if (cos2t < 0) {
                            // total internal reflection keeps adj (main.cpp:144)
                            o = P + n * kEps;
                            d = refl_dir;
                        } else {
                            const V3 refr_dir =
                                normalized(d * nnt - n_old * ((into ? 1 : -1) * (ddn * nnt + sqrt(cos2t))));
                            const double a = nt - nc, b = nt + nc, R0 = a * a / (b * b);
                            const double c = 1 - (into ? -ddn : dot(refr_dir, n_old));
                            const double Re = R0 + (1 - R0) * c * c * c * c * c;
                            const V3 fa = mulv(f, adj);
                            Pending pe;
                            pe.o = P - n * kEps;
                            pe.d = refr_dir;
                            pe.adj = fa * (1 - Re);
                            pe.depth_left = depth_left - 1;
                            pe.path = path * 2 + 1;
                            if (depth_left == 2) {
                                sib = pe;
                                sib_valid = true;
                            } else {
                                if (sp < kLdsLevels) {
                                    double *q = reinterpret_cast<double *>(lslot + sp * kLevelBytes) + threadIdx.x;
                                    q[0 * kThreads] = pe.o.x; q[1 * kThreads] = pe.o.y; q[2 * kThreads] = pe.o.z;
                                    q[3 * kThreads] = pe.d.x; q[4 * kThreads] = pe.d.y; q[5 * kThreads] = pe.d.z;
                                    q[6 * kThreads] = pe.adj.x; q[7 * kThreads] = pe.adj.y; q[8 * kThreads] = pe.adj.z;
                                    // depth_left <= 4 and path < 32: one word
                                    reinterpret_cast<uint32_t *>(lslot + sp * kLevelBytes +
                                                                 kPendDoubles * kThreads * sizeof(double))[threadIdx.x] =
                                        ((uint32_t)pe.depth_left << 8) | pe.path;
                                } else {
                                    deep[sp - kLdsLevels] = pe;
                                }
                                sp++;
                            }
                            o = P + n * kEps;
                            d = refl_dir;
                            adj = fa * Re;
                        }
                        depth_left--;
                        path = path * 2;
                        have = true;
                    }
                }
            }
            if (GLASS && !have && sib_valid) {
                o = sib.o;
                d = sib.d;
                adj = sib.adj;
                depth_left = sib.depth_left;
                path = sib.path;
                sib_valid = false;
                have = true;
            }
            if (GLASS && !have && sp > 0) {
                --sp;
                if (sp < kLdsLevels) {
                    const double *q = reinterpret_cast<const double *>(lslot + sp * kLevelBytes) + threadIdx.x;
                    o = mk(q[0 * kThreads], q[1 * kThreads], q[2 * kThreads]);
                    d = mk(q[3 * kThreads], q[4 * kThreads], q[5 * kThreads]);
                    adj = mk(q[6 * kThreads], q[7 * kThreads], q[8 * kThreads]);
                    const uint32_t meta = reinterpret_cast<const uint32_t *>(
                        lslot + sp * kLevelBytes + kPendDoubles * kThreads * sizeof(double))[threadIdx.x];
                    depth_left = (int)(meta >> 8);
                    path = meta & 0xffu;
                } else {
                    const Pending &pe = deep[sp - kLdsLevels];
                    o = pe.o;
                    d = pe.d;
                    adj = pe.adj;
                    depth_left = pe.depth_left;
                    path = pe.path;
                }
                have = true;
            }
        }
    }

    // ---- coalesced store through LDS: 32 px x 3 floats = 384 contiguous bytes per tile row ----
    if (GLASS) __syncthreads();  // every wave is done with the pending-ray levels the tile aliases
    ltile[ly * (kTileW * 3) + lx * 3 + 0] = (float)(acc_r * g.inv_spp_total);
    ltile[ly * (kTileW * 3) + lx * 3 + 1] = (float)(acc_g * g.inv_spp_total);
    ltile[ly * (kTileW * 3) + lx * 3 + 2] = (float)(acc_b * g.inv_spp_total);
    __syncthreads();
    for (int k = threadIdx.x; k < kTileH * kTileW * 3; k += kThreads) {
        const int row = k / (kTileW * 3), col = k % (kTileW * 3);
        const int jj = tile_y * kTileH + row;
        const int ww = tile_x * kTileW + col / 3;
        if (jj < g.rows && ww < g.W) {
            float *dst = rgb + ((size_t)jj * g.W + tile_x * kTileW) * 3 + col;
            *dst = g.accumulate ? *dst + ltile[k] : ltile[k];  // progressive passes add into the fp32 frame
        }
    }
    if (nhit_out && (w < g.W) && (j < g.rows)) nhit_out[(size_t)j * g.W + w] = my_hits;

    if (counters) {
        // wave reduction, then one atomic per wave and counter
        unsigned long long r = my_rays, hh = my_hits, nn = my_nodes, tt = my_tris;
        for (int off = 32; off > 0; off >>= 1) {
            r += __shfl_xor(r, off);
            hh += __shfl_xor(hh, off);
            if (STATS) {
                nn += __shfl_xor(nn, off);
                tt += __shfl_xor(tt, off);
            }
        }
        if (lane == 0) {
            atomicAdd(&counters[CGRT_CNT_RAYS], r);
            atomicAdd(&counters[CGRT_CNT_HITPOINTS], hh);
            atomicAdd(&counters[CGRT_CNT_WAVE_ITERS], (unsigned long long)wave_iters);
            if (STATS) {
                atomicAdd(&counters[CGRT_CNT_NODE_TESTS], nn);
                atomicAdd(&counters[CGRT_CNT_TRI_TESTS], tt);
            }
        }
    }
}

// function-level probe: one object, n rays (cgrt_intersect_rays)
__global__ void intersect_rays_kernel(DeviceScene sc, int obj, const double *__restrict__ org,
                                      const double *__restrict__ dir, const unsigned long long *__restrict__ keys,
                                      int n, int32_t *__restrict__ hit,
                                      double *__restrict__ len, double *__restrict__ nrm) {
    __shared__ BezLds bl;  // blockDim.x == 64: one wave per block
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool on = i < n;
    const int ii = on ? i : 0;
    uint32_t a = 0, b = 0;
    const V3 o = ld3(org + 3 * ii), d = ld3(dir + 3 * ii);
    DeviceScene one = sc;
    RayKey rk{keys ? keys[ii] : 0ull, 1, true, 0u};
    const LdsAux aux{&bl, nullptr};
    SceneHit h = intersect_scene<true, true, false, false>(sc.objs + obj, 1, one, o, d, rk, on, aux, a, b);
    if (!on) return;
    hit[i] = h.id >= 0 ? 1 : 0;
    len[i] = h.t;
    nrm[3 * i] = h.n.x;
    nrm[3 * i + 1] = h.n.y;
    nrm[3 * i + 2] = h.n.z;
}

// =====================================================================================================
// host side: scene handle, upload, C ABI
// =====================================================================================================
struct cgrt_scene {
    HostScene host;
    bool committed = false;
    int device = -1;
    DeviceScene dev{};
    std::vector<void *> allocs;
    int64_t device_bytes = 0;
    std::vector<int> tree_of;  // flat tree list (object order)
};

static thread_local std::string g_err;
static int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}
#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(CGRT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

template <class T>
static int upload(cgrt_scene *s, const std::vector<T> &v, const T **out) {
    *out = nullptr;
    size_t bytes = v.size() * sizeof(T);
    if (bytes == 0) bytes = sizeof(T);  // keep pointers non-null
    void *p = nullptr;
    HIP_TRY(hipMalloc(&p, bytes));
    s->allocs.push_back(p);
    s->device_bytes += (int64_t)bytes;
    if (!v.empty()) HIP_TRY(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = reinterpret_cast<const T *>(p);
    return CGRT_OK;
}

extern "C" {

int cgrt_version(void) { return CGRT_VERSION; }
const char *cgrt_last_error(void) { return g_err.c_str(); }

int cgrt_scene_create(cgrt_scene **out) {
    if (!out) return fail(CGRT_ERR_INVALID, "cgrt_scene_create: null out");
    *out = new (std::nothrow) cgrt_scene();
    return *out ? CGRT_OK : fail(CGRT_ERR_INVALID, "out of memory");
}

void cgrt_scene_destroy(cgrt_scene *s) {
    if (!s) return;
    if (!s->allocs.empty()) {
        int cur = 0;
        if (hipGetDevice(&cur) == hipSuccess) {
            (void)hipSetDevice(s->device);
            for (void *p : s->allocs) (void)hipFree(p);
            (void)hipSetDevice(cur);
        }
    }
    delete s;
}

#define NEED_OPEN(s)                                                          \
    if (!(s)) return fail(CGRT_ERR_INVALID, "null scene");                    \
    if ((s)->committed) return fail(CGRT_ERR_INVALID, "scene already committed")

static int added(cgrt_scene *s, int r) {
    if (r == -2) return fail(CGRT_ERR_IO, s->host.error);
    if (r < 0) return fail(CGRT_ERR_INVALID, s->host.error);
    if ((int)s->host.objs.size() > kMaxObjs) return fail(CGRT_ERR_LIMIT, "more than 96 top-level objects");
    return r;
}

int cgrt_scene_add_sphere(cgrt_scene *s, const double c[3], double r, const double sc[3], double refl, double transp) {
    NEED_OPEN(s);
    if (!c || !sc) return fail(CGRT_ERR_INVALID, "null argument");
    return added(s, s->host.add_sphere(c, r, sc, refl, transp));
}
int cgrt_scene_add_texture(cgrt_scene *s, const uint8_t *rgb, int rows, int cols, const double n[3], const double p[3],
                           double lx, double ly, int isbump) {
    NEED_OPEN(s);
    if (!n || !p) return fail(CGRT_ERR_INVALID, "null argument");
    int r = s->host.add_texture(rgb, rows, cols, n, p, lx, ly, isbump);
    return r < 0 ? fail(CGRT_ERR_INVALID, s->host.error) : r;
}
int cgrt_scene_add_plane(cgrt_scene *s, const double p[3], const double n[3], const double sc[3], double refl,
                         double transp, int tex_id) {
    NEED_OPEN(s);
    if (!p || !n || !sc) return fail(CGRT_ERR_INVALID, "null argument");
    return added(s, s->host.add_plane(p, n, sc, refl, transp, tex_id));
}
int cgrt_scene_add_mesh_file(cgrt_scene *s, const char *filename, double a, const double b[3], const double sc[3],
                             double refl, double transp, int typeofdata) {
    NEED_OPEN(s);
    if (!filename || !b || !sc) return fail(CGRT_ERR_INVALID, "null argument");
    return added(s, s->host.add_mesh_file(filename, a, b, sc, refl, transp, typeofdata));
}
int cgrt_scene_add_mesh_triangles(cgrt_scene *s, const double *tri9, int ntri, const double sc[3], double refl,
                                  double transp, int typeofdata) {
    NEED_OPEN(s);
    if (!sc) return fail(CGRT_ERR_INVALID, "null argument");
    return added(s, s->host.add_mesh_triangles(tri9, ntri, sc, refl, transp, typeofdata));
}
int cgrt_scene_add_bezier(cgrt_scene *s, const double *cp3, int ncp, const double pos[3], const double sc[3],
                          double refl, double transp) {
    NEED_OPEN(s);
    if (!pos || !sc) return fail(CGRT_ERR_INVALID, "null argument");
    return added(s, s->host.add_bezier(cp3, ncp, pos, sc, refl, transp));
}

int cgrt_scene_commit(cgrt_scene *s, int device) {
    NEED_OPEN(s);
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(CGRT_ERR_DEVICE, "no such HIP device");
    HIP_TRY(hipSetDevice(device));
    s->device = device;
    HostScene &H = s->host;
    // flatten trees
    std::vector<NodeRec> nodes;
    std::vector<TriRec> tris;
    std::vector<TreeRec> trees;
    std::vector<HFieldRec> hfields;
    std::vector<HCellRec> hcells;
    std::vector<OTriRec> otris;
    for (auto &t : H.trees) {
        TreeRec tr;
        tr.node_begin = (int64_t)nodes.size();
        tr.tri_begin = (int64_t)tris.size();
        // CGRT_TREE=ref (measurement aid): traverse the reference's own inner nodes instead of the SAH hierarchy
        const char *tree_env = std::getenv("CGRT_TREE");
        const bool ref_order = tree_env && std::strcmp(tree_env, "ref") == 0;
        const std::vector<NodeRec> &dev_nodes = ref_order ? t.nodes : t.bvh;
        tr.nnodes = ref_order ? (int32_t)t.nodes.size() : t.bvh_nodes;
        tr.noct = ref_order ? 1 : 8;
        tr.tri_level = (!ref_order && t.tri_level) ? 1 : 0;
        tr.pad = 0;
        tr.otri_begin = (int64_t)otris.size();
        if (tr.tri_level) otris.insert(otris.end(), t.otris.begin(), t.otris.end());
        tr.ntris = (int32_t)t.tris.size();
        tr.hfield = -1;
        if (t.is_hfield) {
            HFieldRec hf = t.hfield;
            hf.cell_begin = (int64_t)hcells.size();
            tr.hfield = (int32_t)hfields.size();
            hfields.push_back(hf);
            hcells.insert(hcells.end(), t.hcells.begin(), t.hcells.end());
        }
        nodes.insert(nodes.end(), dev_nodes.begin(), dev_nodes.end());
        tris.insert(tris.end(), t.tris.begin(), t.tris.end());
        trees.push_back(tr);
    }
    std::vector<TexRec> texs;
    std::vector<uint8_t> texels;
    for (auto &t : H.textures) {
        TexRec tr;
        std::memset(&tr, 0, sizeof(tr));
        tr.texel_begin = (int64_t)texels.size();
        tr.rows = t.rows;
        tr.cols = t.cols;
        for (int k = 0; k < 3; k++) {
            tr.n[k] = t.n[k];
            tr.p[k] = t.p[k];
        }
        tr.lenx = t.lenx;
        tr.leny = t.leny;
        tr.isbump = t.isbump ? 1 : 0;
        texels.insert(texels.end(), t.rgb.begin(), t.rgb.end());
        texs.push_back(tr);
    }
    DeviceScene d{};
    int rc;
    if ((rc = upload(s, H.objs, &d.objs))) return rc;
    if ((rc = upload(s, nodes, &d.nodes))) return rc;
    if ((rc = upload(s, tris, &d.tris))) return rc;
    if ((rc = upload(s, trees, &d.trees))) return rc;
    if ((rc = upload(s, texs, &d.texs))) return rc;
    if ((rc = upload(s, texels, &d.texels))) return rc;
    if ((rc = upload(s, H.beziers, &d.beziers))) return rc;
    if ((rc = upload(s, hfields, &d.hfields))) return rc;
    if ((rc = upload(s, hcells, &d.hcells))) return rc;
    if ((rc = upload(s, otris, &d.otris))) return rc;
    d.n_objs = (int32_t)H.objs.size();
    d.n_trees = (int32_t)trees.size();
    d.n_texs = (int32_t)texs.size();
    d.n_beziers = (int32_t)H.beziers.size();
    d.has_mesh = trees.empty() ? 0 : 1;
    d.has_bezier = H.beziers.empty() ? 0 : 1;
    d.cached_tree = -1;
    d.cached_nodes = 0;
    for (size_t t = 0; t < trees.size(); t++)
        if (trees[t].nnodes > 0 && trees[t].nnodes <= kNodeCache) {  // first tree small enough to live in LDS
            d.cached_tree = (int32_t)t;
            d.cached_nodes = trees[t].nnodes;
            break;
        }
    d.all_spheres = 1;
    d.has_glass = 0;
    for (auto &o : H.objs) {
        if (o.kind != KIND_SPHERE) d.all_spheres = 0;
        if (!(o.transp < kEps)) d.has_glass = 1;  // main.cpp:129: the glass branch is `!(transparency < eps)`
    }
    s->dev = d;
    s->committed = true;
    return CGRT_OK;
}

int cgrt_scene_get_stats(const cgrt_scene *s, cgrt_scene_stats *out) {
    if (!s || !out) return fail(CGRT_ERR_INVALID, "null argument");
    std::memset(out, 0, sizeof(*out));
    const HostScene &H = s->host;
    out->n_objects = (int32_t)H.objs.size();
    int64_t bytes = 0;
    for (auto &o : H.objs) {
        if (o.kind == KIND_SPHERE) { out->n_spheres++; bytes += 88; }
        if (o.kind == KIND_PLANE) { out->n_planes++; bytes += 100; }
        if (o.kind == KIND_MESH) out->n_meshes++;
        if (o.kind == KIND_BEZIER) { out->n_beziers++; bytes += 24 * 6 + 100; }
    }
    out->n_textures = (int32_t)H.textures.size();
    out->n_trees = (int32_t)H.trees.size();
    for (auto &t : H.trees) {
        out->n_triangles += (int64_t)t.tris.size();
        out->n_nodes += (int64_t)t.nodes.size();
    }
    bytes += 56 * out->n_nodes + 72 * out->n_triangles;
    for (auto &t : H.textures) bytes += 3 * (int64_t)t.rows * t.cols;
    out->scene_bytes_fp64 = bytes;
    out->device_bytes = s->device_bytes;
    out->committed = s->committed ? 1 : 0;
    return CGRT_OK;
}

int cgrt_scene_tree_sizes(const cgrt_scene *s, int t, int32_t *nnodes, int32_t *nleaftris, int32_t *ntris) {
    if (!s || t < 0 || t >= (int)s->host.trees.size()) return fail(CGRT_ERR_INVALID, "bad tree index");
    const HostTree &T = s->host.trees[t];
    if (nnodes) *nnodes = (int32_t)T.nodes.size();  // the reference's tree (fingerprints); the device hierarchy is T.bvh
    if (nleaftris) *nleaftris = (int32_t)T.leaf_ids.size();
    if (ntris) *ntris = (int32_t)(T.tri9.size() / 9);
    return CGRT_OK;
}
int cgrt_scene_bvh_dump(const cgrt_scene *s, int t, int32_t *nnodes, float *box6, int32_t *skip_leaf2) {
    if (!s || t < 0 || t >= (int)s->host.trees.size()) return fail(CGRT_ERR_INVALID, "bad tree index");
    const HostTree &T = s->host.trees[t];
    if (nnodes) *nnodes = T.bvh_nodes;
    for (size_t i = 0; i < T.bvh.size(); i++) {
        if (box6) {
            for (int k = 0; k < 3; k++) {
                box6[6 * i + k] = T.bvh[i].lo[k];
                box6[6 * i + 3 + k] = T.bvh[i].hi[k];
            }
        }
        if (skip_leaf2) {
            skip_leaf2[2 * i] = T.bvh[i].skip;
            skip_leaf2[2 * i + 1] = T.bvh[i].leaf;
        }
    }
    return CGRT_OK;
}
int cgrt_scene_bvh_order(const cgrt_scene *s, int t, int32_t *tri_level, int32_t *order) {
    if (!s || t < 0 || t >= (int)s->host.trees.size()) return fail(CGRT_ERR_INVALID, "bad tree index");
    const HostTree &T = s->host.trees[t];
    if (tri_level) *tri_level = T.tri_level ? 1 : 0;
    if (order)
        for (size_t j = 0; j < T.otris.size(); j++) order[j] = T.otris[j].k;
    return CGRT_OK;
}
int cgrt_scene_tree_dump(const cgrt_scene *s, int t, int32_t *node_lr_size, int32_t *leaf_ids, double *bbox,
                         double *tri9) {
    if (!s || t < 0 || t >= (int)s->host.trees.size()) return fail(CGRT_ERR_INVALID, "bad tree index");
    const HostTree &T = s->host.trees[t];
    if (node_lr_size) std::memcpy(node_lr_size, T.node_lr_size.data(), T.node_lr_size.size() * sizeof(int32_t));
    if (leaf_ids) std::memcpy(leaf_ids, T.leaf_ids.data(), T.leaf_ids.size() * sizeof(int32_t));
    if (bbox) std::memcpy(bbox, T.bbox.data(), T.bbox.size() * sizeof(double));
    if (tri9) std::memcpy(tri9, T.tri9.data(), T.tri9.size() * sizeof(double));
    return CGRT_OK;
}

static int check_grid(const cgrt_scene *s, const cgrt_camera *cam, const cgrt_grid *g) {
    if (!s || !cam || !g) return fail(CGRT_ERR_INVALID, "null argument");
    if (!s->committed) return fail(CGRT_ERR_INVALID, "scene not committed");
    if (g->width <= 0 || g->height <= 0 || g->rows <= 0) return fail(CGRT_ERR_INVALID, "empty grid");
    if (g->spp <= 0 || g->spp_total <= 0 || g->sample_offset < 0) return fail(CGRT_ERR_INVALID, "bad sample range");
    if (g->max_depth < 1 || g->max_depth > kMaxDepth) return fail(CGRT_ERR_INVALID, "max_depth must be 1..5");
    if (g->stripe_nranks > 1) {
        if (g->stripe_rows <= 0 || g->stripe_rows % kTileH != 0)
            return fail(CGRT_ERR_INVALID, "stripe_rows must be a positive multiple of 8");
        if (g->stripe_rank < 0 || g->stripe_rank >= g->stripe_nranks) return fail(CGRT_ERR_INVALID, "bad stripe_rank");
    } else if (g->row_offset < 0) {
        return fail(CGRT_ERR_INVALID, "bad row_offset");
    }
    if (!(cam->lens_radius >= 0)) return fail(CGRT_ERR_INVALID, "lens_radius must be >= 0");
    return CGRT_OK;
}

int cgrt_trace_grid(const cgrt_scene *s, const cgrt_camera *cam, const cgrt_grid *grid, float *rgb, uint32_t *nhit,
                    uint64_t *counters, void *stream) {
    int rc = check_grid(s, cam, grid);
    if (rc) return rc;
    if (!rgb) return fail(CGRT_ERR_INVALID, "null rgb");
    GridParams g;
    g.W = grid->width;
    g.H = grid->height;
    g.rows = grid->rows;
    g.row_offset = grid->row_offset;
    g.stripe_rows = grid->stripe_rows;
    g.stripe_rank = grid->stripe_rank;
    g.stripe_nranks = grid->stripe_nranks;
    g.spp = grid->spp;
    g.sample_offset = grid->sample_offset;
    g.max_depth = grid->max_depth;
    g.accumulate = (grid->flags & CGRT_GRID_ACCUMULATE) ? 1 : 0;
    g.inv_spp_total = 1.0 / (double)grid->spp_total;
    g.seed = grid->seed;
    for (int k = 0; k < 3; k++) g.cam[k] = cam->cam[k];
    g.half_width = cam->half_width;
    g.focus_plane = cam->focus_plane;
    g.lens_radius = cam->lens_radius;

    g.xcd_tiles = (s->dev.has_mesh && !s->dev.has_bezier) ? 1 : 0;
    const dim3 grid_dim((unsigned)tile_grid_blocks(g.W, g.rows, g.xcd_tiles != 0)), block(kThreads);
    size_t lds = (size_t)s->dev.n_objs * sizeof(ObjRec);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    auto *cnt = reinterpret_cast<unsigned long long *>(counters);
    // launch on the scene's device whatever the caller's current device is (one host thread may drive several GPUs)
    int caller_dev = s->device;
    HIP_TRY(hipGetDevice(&caller_dev));
    if (caller_dev != s->device) HIP_TRY(hipSetDevice(s->device));
    const bool trees = s->dev.has_mesh != 0, dof = cam->lens_radius > 0, bez = s->dev.has_bezier != 0;
    const bool glass = s->dev.has_glass != 0 && grid->max_depth > 1;
    const bool stats = (grid->flags & CGRT_GRID_STATS) != 0 && trees && !bez;
    lds += glass ? kStackBytes : kTileBytes;
    if (bez) lds += (kThreads / 64) * sizeof(BezLds);
    if (trees && s->dev.cached_tree >= 0) lds += (size_t)s->dev.cached_nodes * sizeof(NodeRec);
#define LAUNCH(T, B, D, G, P, S) \
    hipLaunchKernelGGL((trace_grid_kernel<T, B, D, G, P, S>), grid_dim, block, lds, st, s->dev, g, rgb, nhit, cnt)
#define LAUNCH_DG(T, B, P, S)                                      \
    do {                                                           \
        if (dof) { if (glass) LAUNCH(T, B, true, true, P, S); else LAUNCH(T, B, true, false, P, S); }   \
        else     { if (glass) LAUNCH(T, B, false, true, P, S); else LAUNCH(T, B, false, false, P, S); } \
    } while (0)
    if (bez) {  // Bezier scenes share the tree-capable variants (the tree code is skipped when there is no tree)
        LAUNCH_DG(true, true, false, false);
    } else if (trees) {
        if (stats) LAUNCH_DG(true, false, false, true); else LAUNCH_DG(true, false, false, false);
    } else if (s->dev.all_spheres) {
        LAUNCH_DG(false, false, true, false);
    } else {
        LAUNCH_DG(false, false, false, false);
    }
#undef LAUNCH_DG
#undef LAUNCH
    const hipError_t launch_err = hipGetLastError();
    if (caller_dev != s->device) (void)hipSetDevice(caller_dev);
    if (launch_err != hipSuccess) return fail(CGRT_ERR_DEVICE, std::string("kernel launch: ") + hipGetErrorString(launch_err));
    return CGRT_OK;
}

}  // extern "C"

struct DevBuf {  // RAII for device temporaries: freed on every return path
    void *p = nullptr;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
    void *release() { void *q = p; p = nullptr; return q; }
    template <class T> T *as() { return reinterpret_cast<T *>(p); }
};

// Eye pass with Hitpoint capture into a device buffer of `cap` records (10 doubles each); *count = hitpoints produced.
// *d_rec_out is hipMalloc'ed here (caller frees) unless cap == 0.
static int hitpoints_device(const cgrt_scene *s, const cgrt_camera *cam, const cgrt_grid *grid, uint64_t cap,
                            double **d_rec_out, uint64_t *count) {
    GridParams g;
    g.W = grid->width; g.H = grid->height; g.rows = grid->rows; g.row_offset = grid->row_offset;
    g.stripe_rows = grid->stripe_rows; g.stripe_rank = grid->stripe_rank; g.stripe_nranks = grid->stripe_nranks;
    g.spp = grid->spp; g.sample_offset = grid->sample_offset; g.max_depth = grid->max_depth;
    g.accumulate = 0;
    g.inv_spp_total = 1.0 / (double)grid->spp_total;
    g.seed = grid->seed;
    for (int k = 0; k < 3; k++) g.cam[k] = cam->cam[k];
    g.half_width = cam->half_width; g.focus_plane = cam->focus_plane; g.lens_radius = cam->lens_radius;
    const size_t npx = (size_t)grid->rows * grid->width;
    DevBuf b_rgb, b_rec, b_cnt;
    HIP_TRY(b_rgb.alloc(npx * 3 * sizeof(float)));
    HIP_TRY(b_rec.alloc((cap ? cap : 1) * 10 * sizeof(double)));
    HIP_TRY(b_cnt.alloc(sizeof(unsigned long long)));
    float *d_rgb = b_rgb.as<float>();
    double *d_rec = b_rec.as<double>();
    unsigned long long *d_cnt = b_cnt.as<unsigned long long>();
    HIP_TRY(hipMemset(d_cnt, 0, sizeof(unsigned long long)));
    g.xcd_tiles = (s->dev.has_mesh && !s->dev.has_bezier) ? 1 : 0;
    const dim3 grid_dim((unsigned)tile_grid_blocks(g.W, g.rows, g.xcd_tiles != 0)), block(kThreads);
    const size_t lds = (size_t)s->dev.n_objs * sizeof(ObjRec) + kStackBytes + (kThreads / 64) * sizeof(BezLds) +
                       (s->dev.cached_tree >= 0 ? (size_t)s->dev.cached_nodes * sizeof(NodeRec) : 0);
    HitpointSink sink{d_rec, d_cnt, (unsigned long long)cap};
    // the most general variant serves every scene; capture is a verification / hand-off path, not the hot path
    if (cam->lens_radius > 0)
        hipLaunchKernelGGL((trace_grid_kernel<true, true, true, true, false, false, true>), grid_dim, block, lds, 0,
                           s->dev, g, d_rgb, (uint32_t *)nullptr, (unsigned long long *)nullptr, sink);
    else
        hipLaunchKernelGGL((trace_grid_kernel<true, true, false, true, false, false, true>), grid_dim, block, lds, 0,
                           s->dev, g, d_rgb, (uint32_t *)nullptr, (unsigned long long *)nullptr, sink);
    int rc = CGRT_OK;
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) rc = fail(CGRT_ERR_DEVICE, std::string("hitpoint kernel: ") + hipGetErrorString(e));
    unsigned long long n = 0;
    if (rc == CGRT_OK && hipMemcpy(&n, d_cnt, sizeof(n), hipMemcpyDeviceToHost) != hipSuccess)
        rc = fail(CGRT_ERR_DEVICE, "hitpoint count copy");
    *count = n;
    if (rc == CGRT_OK && cap && d_rec_out) *d_rec_out = reinterpret_cast<double *>(b_rec.release());
    return rc;
}

extern "C" {

int cgrt_trace_grid_hitpoints(const cgrt_scene *s, const cgrt_camera *cam, const cgrt_grid *grid, double *hp10,
                              uint64_t cap, uint64_t *count) {
    int rc = check_grid(s, cam, grid);
    if (rc) return rc;
    if (!count || (cap > 0 && !hp10)) return fail(CGRT_ERR_INVALID, "null output");
    HIP_TRY(hipSetDevice(s->device));
    double *d_rec = nullptr;
    uint64_t n = 0;
    rc = hitpoints_device(s, cam, grid, cap, &d_rec, &n);
    if (rc == CGRT_OK) {
        *count = n;
        const uint64_t m = n < cap ? n : cap;
        if (m && hipMemcpy(hp10, d_rec, (size_t)m * 10 * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
            rc = fail(CGRT_ERR_DEVICE, "hitpoint copy");
    }
    if (d_rec) (void)hipFree(d_rec);
    return rc;
}

int cgrt_trace_grid_host(const cgrt_scene *s, const cgrt_camera *cam, const cgrt_grid *grid, float *rgb,
                         uint32_t *nhit, uint64_t *counters) {
    int rc = check_grid(s, cam, grid);
    if (rc) return rc;
    if (!rgb) return fail(CGRT_ERR_INVALID, "null rgb");
    HIP_TRY(hipSetDevice(s->device));
    const size_t npx = (size_t)grid->rows * grid->width;
    DevBuf b_rgb, b_nhit, b_cnt;
    HIP_TRY(b_rgb.alloc(npx * 3 * sizeof(float)));
    HIP_TRY(b_nhit.alloc(npx * sizeof(uint32_t)));
    HIP_TRY(b_cnt.alloc(CGRT_NCOUNTERS * sizeof(uint64_t)));
    float *d_rgb = b_rgb.as<float>();
    uint32_t *d_nhit = b_nhit.as<uint32_t>();
    uint64_t *d_cnt = b_cnt.as<uint64_t>();
    HIP_TRY(hipMemset(d_rgb, 0, npx * 3 * sizeof(float)));
    HIP_TRY(hipMemset(d_nhit, 0, npx * sizeof(uint32_t)));
    HIP_TRY(hipMemset(d_cnt, 0, CGRT_NCOUNTERS * sizeof(uint64_t)));
    rc = cgrt_trace_grid(s, cam, grid, d_rgb, d_nhit, d_cnt, nullptr);
    if (rc == CGRT_OK) {
        hipError_t e = hipDeviceSynchronize();
        if (e != hipSuccess) rc = fail(CGRT_ERR_DEVICE, std::string("kernel: ") + hipGetErrorString(e));
    }
    if (rc == CGRT_OK) {
        HIP_TRY(hipMemcpy(rgb, d_rgb, npx * 3 * sizeof(float), hipMemcpyDeviceToHost));
        if (nhit) HIP_TRY(hipMemcpy(nhit, d_nhit, npx * sizeof(uint32_t), hipMemcpyDeviceToHost));
        if (counters) HIP_TRY(hipMemcpy(counters, d_cnt, CGRT_NCOUNTERS * sizeof(uint64_t), hipMemcpyDeviceToHost));
    }
    return rc;
}

int cgrt_intersect_rays(const cgrt_scene *s, int obj, const double *org3, const double *dir3, const uint64_t *keys,
                        int n, int32_t *hit, double *len, double *normal3) {
    if (!s || !s->committed) return fail(CGRT_ERR_INVALID, "scene not committed");
    if (obj < 0 || obj >= s->dev.n_objs || n < 0 || !org3 || !dir3 || !hit || !len || !normal3)
        return fail(CGRT_ERR_INVALID, "bad argument");
    if (n == 0) return CGRT_OK;
    HIP_TRY(hipSetDevice(s->device));
    DevBuf b_o, b_d, b_len, b_n, b_hit, b_keys;
    HIP_TRY(b_o.alloc((size_t)n * 24));
    HIP_TRY(b_d.alloc((size_t)n * 24));
    HIP_TRY(b_len.alloc((size_t)n * 8));
    HIP_TRY(b_n.alloc((size_t)n * 24));
    HIP_TRY(b_hit.alloc((size_t)n * 4));
    double *d_o = b_o.as<double>(), *d_d = b_d.as<double>(), *d_len = b_len.as<double>(), *d_n = b_n.as<double>();
    int32_t *d_hit = b_hit.as<int32_t>();
    HIP_TRY(hipMemcpy(d_o, org3, (size_t)n * 24, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_d, dir3, (size_t)n * 24, hipMemcpyHostToDevice));
    unsigned long long *d_keys = nullptr;
    if (keys) {
        HIP_TRY(b_keys.alloc((size_t)n * 8));
        d_keys = b_keys.as<unsigned long long>();
        HIP_TRY(hipMemcpy(d_keys, keys, (size_t)n * 8, hipMemcpyHostToDevice));
    }
    hipLaunchKernelGGL(intersect_rays_kernel, dim3((n + 63) / 64), dim3(64), 0, 0, s->dev, obj, d_o, d_d, d_keys, n,
                       d_hit, d_len, d_n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(hit, d_hit, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(len, d_len, (size_t)n * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(normal3, d_n, (size_t)n * 24, hipMemcpyDeviceToHost));
    return CGRT_OK;
}

int cgrt_lens_samples(uint64_t seed, const int64_t *pixel, const int32_t *sample, int n, double radius, double *out3) {
    if (n < 0 || (n > 0 && (!pixel || !sample || !out3))) return fail(CGRT_ERR_INVALID, "bad argument");
    for (int i = 0; i < n; i++) {
        Stream rs(stream_key(seed, (uint64_t)pixel[i], (uint64_t)sample[i], 0));
        double sx, sy;
        while (true) {
            double ux, uy;
            rs.pair(ux, uy);
            sx = ux * 2.0 - 1;
            sy = uy * 2.0 - 1;
            if (sx * sx + sy * sy < 1) break;
        }
        out3[3 * i] = sx * radius;
        out3[3 * i + 1] = sy * radius;
        out3[3 * i + 2] = 0 * radius;
    }
    return CGRT_OK;
}

}  // extern "C"

#include "cgrt_photon.inc"
