// Mesh traversal on the device: the stackless hierarchy walk with the reference's leaf scan, and the height-field walk
// for opaque bump floors.  Part of libcgrt.so (cgrt_hip.hip).
#ifndef CGRT_TRAVERSE_HPP
#define CGRT_TRAVERSE_HPP
#include "cgrt_device_math.hpp"
#include "cgrt_grid.hpp"

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
// ---- the box test in single precision --------------------------------------------------------------------------------
// A box test only has to accept a SUPERSET of the boxes the ray really touches (above), and a ray's entry distance into a box
// is only ever used as a LOWER bound (pruning).  The boxes are fp32 already; evaluating the slabs in fp32 too halves the
// instructions of a node visit (no conversions, two-wide packed subtract / multiply) -- provided every rounding error is
// covered.  For one face at coordinate b the computed crossing is
//     t = fl32( fl32(b - o32) * i32 ),   o32 = fl32(o),  i32 = fl32(1/d)
// = (b_eq - o) / d for a plane at b_eq with |b_eq - b| <= u (3.01 |b| + 4.02 |o|), u = 2^-24 (expand the three roundings and
// the two conversions; 1/d itself is the correctly rounded fp64 quotient).  So the test is run against boxes grown, per ray
// and per axis, by e = 8 u (|o| + bmax) >= twice that bound, bmax >= |b| for every face of the tree (TreeRec::bmax): the
// lower faces see the origin o32 + e, the upper faces o32 - e, at no cost per box.  Then every computed slab interval
// CONTAINS the true one of the stored (1e-4-grown) box: accepted boxes are a superset, computed entry distances lower
// bounds.  d_k = 0 gives i32 = +-inf and intervals (-inf, +inf) or empty exactly as in fp64 (0 * inf = NaN arises only for
// an origin that the growth puts exactly on a face, i.e. a true origin outside the stored box: NaN loses every min / max,
// the box is rejected, correctly).  Coordinates are assumed to stay below ~1e30 in magnitude (no fp32 overflow).
struct Ray32 {
    float olo[3], ohi[3], inv[3];
};
__device__ __forceinline__ Ray32 make_ray32(V3 o, V3 inv, float bmax) {
    Ray32 r;
    const double oo[3] = {o.x, o.y, o.z}, ii[3] = {inv.x, inv.y, inv.z};
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float o32 = (float)oo[k];
        const float e = (fabsf(o32) + bmax) * 4.8e-7f + 1e-30f;  // 8 * 2^-24 = 4.77e-7
        r.olo[k] = o32 + e;
        r.ohi[k] = o32 - e;
        r.inv[k] = (float)ii[k];
    }
    return r;
}
// entry / exit of the grown box {lo, hi}: tn <= true entry distance, tf >= true exit distance
__device__ __forceinline__ void slab32(const Ray32 &r, float lx, float ly, float lz, float hx, float hy, float hz, float &tn, float &tf) {
    const float ax = (lx - r.olo[0]) * r.inv[0], bx = (hx - r.ohi[0]) * r.inv[0];
    const float ay = (ly - r.olo[1]) * r.inv[1], by = (hy - r.ohi[1]) * r.inv[1];
    const float az = (lz - r.olo[2]) * r.inv[2], bz = (hz - r.ohi[2]) * r.inv[2];
    tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
    tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
}

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
// An opaque owner's own hierarchy goes further: it reaches down to groups of <= 4 single triangles and is walked in a 4-wide
// form (tree_intersect_wide below), a transparent owner's is walked by tree_intersect_lq; this routine serves what is left: opaque
// owners whose stackless hierarchy is walked as it is (CGRT_TREE=ref; a bump floor without a grid).
template <bool STATS, bool PRUNE>
__device__ __forceinline__ TreeHit tree_intersect(const NodeRec *__restrict__ nodes, const TriRec *__restrict__ tris,
                                                  int nnodes, V3 o, V3 d, const Ray32 &r32, double bound, uint32_t &n_node,
                                                  uint32_t &n_tri) {
    float bound32 = PRUNE ? __double2float_ru(bound) : 0.f;  // >= bound: an entry distance above it is above bound
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
            UTIL(2);
            // one 32-byte record = two 16-byte loads
            const float4 q0 = reinterpret_cast<const float4 *>(nodes + i)[0];  // lo.x lo.y lo.z hi.x
            const float4 q1 = reinterpret_cast<const float4 *>(nodes + i)[1];  // hi.y hi.z skip leaf
            if (STATS) n_node++;
            float tn, tf;
            slab32(r32, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, tn, tf);
            const bool touch = (tf > 0.f) && (tn <= tf) && !(PRUNE && tn > bound32);
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
        UTIL(3);
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
            if (PRUNE && r.len < bound) {
                bound = r.len;
                bound32 = __double2float_ru(bound);
            }
        }
    }
    return r;
}

// =====================================================================================================
// Transparent owners: the same walk with the touched leaves queued
// =====================================================================================================
// Every leaf the ray touches must be scanned, in order, for the improvement counter.  The reference's leaves are loose (7.5
// triangles under one box, a ray that touches the box misses most of them) and a triangle test is ~110 fp64 instructions on 72
// bytes, so each triangle is first tested against its own grown fp32 box (`tboxes`: 32 bytes, ~25 instructions, a superset test
// like the nodes') and only triangles whose box the ray touches get the exact test; hits, their order and the counter are unchanged.
//
// tree_intersect above alternates "every lane walks until it stands on a leaf" and "scan the leaves".  Measured on the glass bunny
// (tools/util_probe.py): a lane needs 4.4 node steps to its next leaf on average, the slowest lane of the wave 12.6 -- 19 of 64
// lanes are active in a node step, and the node steps (a chain of dependent LDS fetches) are what the walk's time is made of.
// Here a lane does not stop at a leaf: it puts the leaf into a four-entry queue and walks on; all lanes take kLqSteps node steps
// together, and one queued leaf per lane is scanned when a quarter of the wave's lanes hold one, or nobody can walk.  A leaf's
// scan and the merge of the leaves are tree_intersect's (leaves are merged on their position, not on the order of visiting), so
// (len, triangle, counter) are the same.  Measured: node-step executions per frame -43 % at 34.6 instead of 19.6 lanes, C3
// 13.4 -> 12.8-13.0 ms -- the walk's node steps are a sixth of that frame's issue cycles, not more.
#ifndef CGRT_LQ_STEPS
#define CGRT_LQ_STEPS 8
#endif
#ifndef CGRT_LQ_NUM
#define CGRT_LQ_NUM 1
#define CGRT_LQ_DEN 4
#endif
static constexpr int kLqSteps = CGRT_LQ_STEPS;  // node steps between two looks at the queues (measured 2 / 4 / 8 / 16 / 32: C3 13.4 / 13.1 / 12.9 / 13.3 / 13.6 ms)
// leaves are scanned when kLqNum / kLqDen of the lanes hold one (1/4: 12.8, 1/2: 13.0, 3/4: 13.1 ms)
static constexpr int kLqNum = CGRT_LQ_NUM, kLqDen = CGRT_LQ_DEN;

template <bool STATS>
__device__ __forceinline__ TreeHit tree_intersect_lq(const NodeRec *__restrict__ nodes, const TriRec *__restrict__ tris,
                                                     int nnodes, V3 o, V3 d, const Ray32 &r32, uint32_t &n_node,
                                                     uint32_t &n_tri, const NodeRec *__restrict__ tboxes) {
    TreeHit r;
    r.len = kInf;
    r.tri = -1;
    r.counter = 0;
    int r_leaf = -1;
    int i = 0;
    int lq0 = 0, lq1 = 0, lq2 = 0, lq3 = 0, nl = 0;  // queued leaves (NodeRec::leaf), oldest in lq0
    const int n_active = (int)__popcll(__ballot(true));
    while (true) {
        for (int st = 0; st < kLqSteps; st++) {
            const bool can = i < nnodes && nl < 4;
            if (__ballot(can) == 0ull) break;
            if (can) {
                UTIL(2);
                const float4 q0 = reinterpret_cast<const float4 *>(nodes + i)[0];  // lo.x lo.y lo.z hi.x
                const float4 q1 = reinterpret_cast<const float4 *>(nodes + i)[1];  // hi.y hi.z skip leaf
                if (STATS) n_node++;
                float tn, tf;
                slab32(r32, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, tn, tf);
                const bool touch = (tf > 0.f) && (tn <= tf);
                const int leaf = __float_as_int(q1.w);
                if (!touch) {
                    i = __float_as_int(q1.z);
                } else {
                    i = i + 1;
                    if (leaf >= 0) {
                        if (nl == 0) lq0 = leaf;
                        else if (nl == 1) lq1 = leaf;
                        else if (nl == 2) lq2 = leaf;
                        else lq3 = leaf;
                        nl++;
                    }
                }
            }
        }
        const unsigned long long walkers = __ballot(i < nnodes && nl < 4);
        const unsigned long long holders = __ballot(nl > 0);
        if (walkers == 0ull && holders == 0ull) break;
        if (walkers != 0ull && kLqDen * (int)__popcll(holders) < kLqNum * n_active) continue;
        if (nl > 0) {
            UTIL(3);
            const int leaf_begin = lq0 >> 4, leaf_cnt_tris = lq0 & 15;
            lq0 = lq1;
            lq1 = lq2;
            lq2 = lq3;
            nl--;
            double leaf_len = kInf;
            int leaf_tri = -1, leaf_cnt = 0;
            const TriRec *tp = tris + leaf_begin;
            const NodeRec *bp = tboxes + leaf_begin;
            unsigned cand = 0;
            for (int k = 0; k < leaf_cnt_tris; k++) {
                UTIL(4);
                const float4 q0 = reinterpret_cast<const float4 *>(bp + k)[0];  // lo.x lo.y lo.z hi.x
                const float2 q1 = reinterpret_cast<const float2 *>(bp + k)[2];  // hi.y hi.z
                float tn, tf;
                slab32(r32, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, tn, tf);
                if ((tf > 0.f) && (tn <= tf)) cand |= 1u << k;
            }
            while (cand != 0u) {
                const int k = __ffs((int)cand) - 1;
                cand &= cand - 1u;
                UTIL(5);
                if (STATS) n_tri++;
                const V3 pa = ld3(tp[k].pa), e1 = ld3(tp[k].e1), e2 = ld3(tp[k].e2);
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
            }
            if (leaf_cnt > 0) {  // objects.h:295-313, see tree_intersect
                if (r.counter == 0 || leaf_len < r.len || (leaf_len == r.len && leaf_begin > r_leaf)) {
                    r.len = leaf_len;
                    r.tri = leaf_tri;
                    r_leaf = leaf_begin;
                }
                r.counter += leaf_cnt;
            }
        }
    }
    return r;
}

// =====================================================================================================
// Opaque meshes: the 4-wide form of the triangle-level hierarchy (WideNodeRec)
// =====================================================================================================
// Only the nearest hit of an opaque object matters (PRUNE above), so its hierarchy need not stop at the reference's leaves:
// it goes down to groups of <= 4 single triangles (`otris`, each carrying its place in the reference's leaf order), and with
// no leaf-wide scan order to rely on every accepted hit is compared on (len, reference leaf, index) directly -- the same
// winner as "first inside a leaf, last leaf across leaves" (objects.h:281,297).  Same grown boxes and triangle test as
// tree_intersect; what differs is how a ray gets from box to box.  One step fetches a node's four child boxes (seven independent 16-byte loads of one 128-byte line), tests
// them all, goes on with the NEAREST child it touches and leaves the others on a per-lane stack with their entry distances
// (fp32, rounded down: an entry is dropped at pop time only if it provably begins beyond the nearest hit).  A walk is bound by
// the latency of its dependent fetches -- the kernel runs at three waves per SIMD --, and this form has ~2.5x fewer of them
// than one box per node (measured on C4: the same 132 M box tests per 67 M rays, frame 33.2 -> 31.7 ms); the near-to-far
// order comes from the entry distances, so one copy of the nodes serves every octant.  The box arithmetic per child is
// tree_intersect's, so the superset argument is unchanged, and any order of visiting leaves gives
// the same (len, triangle).
template <bool STATS>
__device__ __forceinline__ TreeHit tree_intersect_wide(const WideNodeRec *__restrict__ wn, const OTriRec *__restrict__ otris, V3 o,
                                                       V3 d, const Ray32 &r32, double bound, uint32_t &n_node, uint32_t &n_tri,
                                                       uint2 *lstack = nullptr) {
    float bound32 = __double2float_ru(bound);  // >= bound: an entry distance above it is above bound
    TreeHit r;
    r.len = kInf;
    r.tri = -1;
    r.counter = 0;
    int r_leaf = -1;
    // the stack: {ref, entry distance as float bits}; its first kWideLdsDepth entries in LDS when the workgroup has room for
    // them (lstack: [entry][thread]), the rest -- or all of it -- in scratch
    uint2 stk[kWideStack];
    int sp = 0;
    const int nt = blockDim.x, tid = threadIdx.x;
    auto push = [&](uint2 e) {
        if (lstack && sp < kWideLdsDepth) lstack[sp * nt + tid] = e;
        else stk[sp] = e;
        sp++;
    };
    auto pop = [&]() -> int32_t {
        while (sp > 0) {
            --sp;
            const uint2 e = (lstack && sp < kWideLdsDepth) ? lstack[sp * nt + tid] : stk[sp];
            if (!(__uint_as_float(e.y) > bound32)) return (int32_t)e.x;
        }
        return kWideNone;
    };
    int32_t nxt = ~0;  // the root
    while (true) {
        while (nxt < 0 && nxt != kWideNone) {  // inner nodes, all lanes together, until this lane stands on a leaf or has nothing left
            const float4 *q = reinterpret_cast<const float4 *>(wn + (~nxt));
            const float4 lox = q[0], loy = q[1], loz = q[2], hix = q[3], hiy = q[4], hiz = q[5];
            const int4 ref = reinterpret_cast<const int4 *>(q)[6];
            const float lx[4] = {lox.x, lox.y, lox.z, lox.w}, ly[4] = {loy.x, loy.y, loy.z, loy.w}, lz[4] = {loz.x, loz.y, loz.z, loz.w};
            const float hx[4] = {hix.x, hix.y, hix.z, hix.w}, hy[4] = {hiy.x, hiy.y, hiy.z, hiy.w}, hz[4] = {hiz.x, hiz.y, hiz.z, hiz.w};
            const int32_t rf[4] = {ref.x, ref.y, ref.z, ref.w};
            float tn4[4];
            bool hit4[4];
            float best_tn = __int_as_float(0x7f800000);  // +inf
            int32_t best_ref = kWideNone;
            int best = -1;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                float tn, tf;
                slab32(r32, lx[k], ly[k], lz[k], hx[k], hy[k], hz[k], tn, tf);
                const bool touch = (rf[k] != kWideNone) && (tf > 0.f) && (tn <= tf) && !(tn > bound32);
                if (STATS && rf[k] != kWideNone) n_node++;
                tn4[k] = tn;
                hit4[k] = touch;
                if (touch && tn < best_tn) {
                    best_tn = tn;
                    best_ref = rf[k];
                    best = k;
                }
            }
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (hit4[k] && k != best) push(make_uint2((uint32_t)rf[k], __float_as_uint(tn4[k])));
            nxt = best >= 0 ? best_ref : pop();
        }
        if (nxt == kWideNone) break;
        {
            const OTriRec *tp = otris + (nxt >> 4);
            const int cnt = nxt & 15;
            for (int k = 0; k < cnt; k++) {
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
            if (r.len < bound) {
                bound = r.len;
                bound32 = __double2float_ru(bound);
            }
        }
        nxt = pop();
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
__device__ __forceinline__ TreeHit hfield_intersect(const HFieldRec &H, const HCellRec *__restrict__ cells,
                                                    const HCellY *__restrict__ celly, V3 o, V3 d,
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
    const double ihM = xmaj ? H.ihx : H.ihz, ihm = xmaj ? H.ihz : H.ihx;  // = 1.0 / hM, 1.0 / hm (computed once, at commit)
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
        // heights the ray passes through while it is over this column (a hit point lies over its cell, hence within the
        // column's parameter range [s0, s1], and y is linear in the parameter)
        const double ya = o.y + d.y * s0, yb = o.y + d.y * s1;
        const double ymin = fmin(ya, yb) - kHfPad, ymax = fmax(ya, yb) + kHfPad;
        for (int i = i0; i <= i1; i++) {
            const size_t ci = xmaj ? (size_t)i * H.nx + j : (size_t)j * H.nx + i;
            // a cell whose four vertices all lie below or all above that range cannot be hit: 8 bytes decide, the 160-byte
            // record and the two triangle tests are skipped (the stored range is rounded outward)
            const HCellY cy = celly[ci];
            if (ymin > (double)cy.hi || ymax < (double)cy.lo) continue;
            if (STATS) n_node++;
            const HCellRec *cell = cells + ci;
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

#endif
