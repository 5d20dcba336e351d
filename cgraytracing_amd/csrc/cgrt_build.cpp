// Host-side scene assembly for libcgrt.so (see cgrt_build.h).  Citations: /root/reference/<file>:<line>.
#include "cgrt_build.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <future>
#include <thread>
#include <unordered_map>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace cgrt {

namespace {

inline double hi3(double a, double b, double c) { return (a > b && a > c) ? a : (b > c ? b : c); }  // util.h:16-27
inline double lo3(double a, double b, double c) { return (a < b && a < c) ? a : (b < c ? b : c); }  // util.h:33-42
// nearest floats not above / not below v (outward rounding of the padded box)
inline float round_down(double v) {
    float f = (float)v;
    return ((double)f > v) ? std::nextafterf(f, -INFINITY) : f;
}
inline float round_up(double v) {
    float f = (float)v;
    return ((double)f < v) ? std::nextafterf(f, INFINITY) : f;
}

// ---- tree build -------------------------------------------------------------------------------
// The reference builds an object-median tree (it calls it a KD-tree): every node takes the triangle
// list it was handed, records its bounding box, and -- when it holds 10 or more triangles -- sorts
// the list by the per-triangle MAXIMUM coordinate on axis depth%3 with std::sort, gives the lower
// half (n/2) to the left child and the rest to the right (objects.h:217-267).  Nodes are numbered
// in the order they are created, which is preorder.
//
// Which side of a glass mesh a ray is on is decided by the parity of a counter that depends on the
// order of triangles inside the leaves (objects.h:281-286,321-327), so the leaf order -- including
// how std::sort happens to arrange equal keys -- is part of the observable behaviour.  We therefore
// run the same std::sort over the same sequence: sorting a slice of one id array in place visits
// the same comparisons as sorting the reference's private copy of that slice.
struct Builder {
    const std::vector<double> &T;  // 9 doubles per triangle
    std::vector<int32_t> ids;
    HostTree &out;

    double key(int32_t id, int axis) const {
        const double *t = &T[9 * (size_t)id];
        return hi3(t[axis], t[3 + axis], t[6 + axis]);
    }
    // Node numbers and leaf positions follow from the triangle counts alone (a node of n >= 10 triangles has children of
    // n/2 and n - n/2; the leaves partition the id array in order), so every node knows its number before its subtree
    // exists: `me` for itself, me + 1 for the left child, me + 1 + nodes(n/2) for the right one, and a leaf over ids[b, e)
    // owns tris[b, e).  That makes the two halves of a node independent pieces of work -- the sorts touch disjoint slices
    // -- and the top of the tree is built by several threads; the output is the serial build's, element for element.
    std::unordered_map<size_t, size_t> node_count;
    size_t count_nodes(size_t n) {  // fills the memo (call once from one thread before emit)
        auto it = node_count.find(n);
        if (it != node_count.end()) return it->second;
        const size_t c = ((int)n < kMinKd) ? 1 : 1 + count_nodes(n / 2) + count_nodes(n - n / 2);
        node_count[n] = c;
        return c;
    }
    size_t nodes_of(size_t n) const { return node_count.at(n); }
    void emit(size_t b, size_t e, int axis, int32_t me, int depth) {
        const size_t n = e - b;
        double mx[3] = {-kInf, -kInf, -kInf}, mn[3] = {kInf, kInf, kInf};  // objects.h:227-232
        for (size_t i = b; i < e; i++) {
            const double *t = &T[9 * (size_t)ids[i]];
            for (int k = 0; k < 3; k++) {
                double h = hi3(t[k], t[3 + k], t[6 + k]), l = lo3(t[k], t[3 + k], t[6 + k]);
                if (mx[k] < h) mx[k] = h;
                if (mn[k] > l) mn[k] = l;
            }
        }
        NodeRec nr;
        for (int k = 0; k < 3; k++) {
            nr.lo[k] = round_down(mn[k] - kBoxPad);
            nr.hi[k] = round_up(mx[k] + kBoxPad);
        }
        nr.skip = me + (int32_t)nodes_of(n);
        nr.leaf = -1;
        out.node_lr_size[3 * (size_t)me + 0] = -1;
        out.node_lr_size[3 * (size_t)me + 1] = -1;
        out.node_lr_size[3 * (size_t)me + 2] = (int32_t)n;
        for (int k = 0; k < 3; k++) {
            out.bbox[6 * (size_t)me + 2 * (size_t)k] = mn[k];
            out.bbox[6 * (size_t)me + 2 * (size_t)k + 1] = mx[k];
        }
        if ((int)n < kMinKd) {  // leaf (objects.h:251, 273)
            nr.leaf = (int32_t)(((uint32_t)b << 4) | (uint32_t)n);
            out.nodes[(size_t)me] = nr;
            for (size_t i = b; i < e; i++) {
                const double *t = &T[9 * (size_t)ids[i]];
                TriRec tr;
                for (int k = 0; k < 3; k++) {
                    tr.pa[k] = t[k];
                    tr.e1[k] = t[k] - t[3 + k];  // pa - pb (objects.h:98)
                    tr.e2[k] = t[k] - t[6 + k];  // pa - pc (objects.h:99)
                }
                out.tris[i] = tr;
                out.leaf_ids[i] = ids[i];
            }
            return;
        }
        out.nodes[(size_t)me] = nr;
        std::sort(ids.begin() + b, ids.begin() + e,
                  [this, axis](int32_t p, int32_t q) { return key(p, axis) < key(q, axis); });
        const size_t mid = b + n / 2;
        const int next_axis = (axis + 1) % 3;
        const int32_t left = me + 1, right = me + 1 + (int32_t)nodes_of(n / 2);
        out.node_lr_size[3 * (size_t)me + 0] = left;
        out.node_lr_size[3 * (size_t)me + 1] = right;
        if (depth < 3 && n > 16384) {
            std::future<void> fl = std::async(std::launch::async, [=] { emit(b, mid, next_axis, left, depth + 1); });
            emit(mid, e, next_axis, right, depth + 1);
            fl.get();
        } else {
            emit(b, mid, next_axis, left, depth + 1);
            emit(mid, e, next_axis, right, depth + 1);
        }
    }
};

void set3(double *d, const double *s) {
    d[0] = s[0];
    d[1] = s[1];
    d[2] = s[2];
}

ObjRec blank_obj(int kind, const double sc[3], double refl, double transp) {
    ObjRec o;
    std::memset(&o, 0, sizeof(o));
    o.kind = kind;
    set3(o.col, sc);
    o.refl = refl;
    o.transp = transp;
    o.tree = -1;
    o.tex = -1;
    o.axis = -1;
    return o;
}

// ---- text scanning for the three mesh formats (objects.h:343-400) ------------------------------
struct Scanner {
    std::vector<char> buf;
    const char *p = nullptr, *end = nullptr;
    bool open(const char *file) {
        FILE *f = std::fopen(file, "rb");
        if (!f) return false;
        std::fseek(f, 0, SEEK_END);
        long n = std::ftell(f);
        std::fseek(f, 0, SEEK_SET);
        buf.resize((size_t)(n > 0 ? n : 0) + 1);
        size_t got = n > 0 ? std::fread(buf.data(), 1, (size_t)n, f) : 0;
        std::fclose(f);
        buf[got] = 0;
        p = buf.data();
        end = p + got;
        return true;
    }
    void ws() {
        while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r' || *p == '\f' || *p == '\v')) p++;
    }
    bool at_end() {
        ws();
        return p >= end;
    }
    bool word(const char *w) {  // literal token followed by whitespace/end
        ws();
        size_t n = std::strlen(w);
        if ((size_t)(end - p) < n || std::strncmp(p, w, n) != 0) return false;
        const char *q = p + n;
        if (q < end && !(*q == ' ' || *q == '\t' || *q == '\n' || *q == '\r')) return false;
        p = q;
        return true;
    }
    bool real(double &v) {
        ws();
        char *e = nullptr;
        v = std::strtod(p, &e);
        if (e == p) return false;
        p = e;
        return true;
    }
    bool integer(long &v) {
        ws();
        char *e = nullptr;
        v = std::strtol(p, &e, 10);
        if (e == p) return false;
        p = e;
        return true;
    }
    void skip_token() {
        ws();
        while (p < end && !(*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) p++;
    }
};

}  // namespace

void HostTree::build(bool opaque, bool with_bvh) {
    nodes.clear();
    tris.clear();
    node_lr_size.clear();
    bbox.clear();
    leaf_ids.clear();
    Builder b{tri9, {}, *this, {}};
    const size_t n = tri9.size() / 9;
    b.ids.resize(n);
    for (size_t i = 0; i < n; i++) b.ids[i] = (int32_t)i;
    const size_t nn = b.count_nodes(n);
    nodes.resize(nn);
    node_lr_size.resize(3 * nn);
    bbox.resize(6 * nn);
    tris.resize(n);
    leaf_ids.resize(n);
    b.emit(0, n, 0, 0, 0);
    // per-triangle boxes in leaf order (cgrt_traverse.hpp, leaf scan): a triangle can only be hit at a point of its own
    // box, so the grown box is a superset test exactly like the nodes' 
    tboxes.resize(n);
    for (size_t k = 0; k < n; k++) {
        const double *t = &tri9[9 * (size_t)leaf_ids[k]];
        NodeRec nr;
        for (int c = 0; c < 3; c++) {
            nr.lo[c] = round_down(lo3(t[c], t[3 + c], t[6 + c]) - kBoxPad);
            nr.hi[c] = round_up(hi3(t[c], t[3 + c], t[6 + c]) + kBoxPad);
        }
        nr.skip = 0;
        nr.leaf = 0;
        tboxes[k] = nr;
    }
    if (with_bvh) build_bvh(opaque);
}

// ---- traversal hierarchy over the reference's leaves ---------------------------------------------------------------
// What the reference's tree contributes to the RESULT is its leaves: which triangles share a leaf and in which order
// (quirk Q5), and the order of the leaves themselves (ties between leaves, objects.h:297).  The inner nodes only decide
// which leaves a ray reaches, and any set of boxes that contains every leaf a ray can hit serves (DESIGN.md section
// 4.2).  The reference's inner nodes are poor at that job: object-median splits on the triangles' MAXIMUM coordinate,
// heavily overlapping, and visited in an order unrelated to the ray.  So the device traverses a binned-SAH hierarchy
// over the same leaves.  It stays stackless (preorder + skip links, like the reference-order form) and still visits
// children front to back, because it is laid out eight times, once per octant of ray directions, each time with the
// children of every node ordered near-to-far along that node's split axis -- 32 bytes per node and octant.
namespace {
struct BvhItem {
    double lo[3], hi[3];
    int32_t leafref;  // NodeRec::leaf encoding of the reference leaf
    int32_t weight;   // triangles in it
};
struct BvhTmp {
    double lo[3], hi[3];
    int32_t left = -1, right = -1, axis = 0, leafref = -1;
    bool left_is_lower = true;
};
struct BvhBuilder {
    std::vector<BvhItem> items;
    std::vector<int32_t> idx;
    std::vector<BvhTmp> nodes;          // preallocated (2 * items); slots are claimed through `next`, so node NUMBERS depend on
    std::atomic<int32_t> next{0};       // thread timing -- the tree's shape, and everything emitted from it, does not
    int sah_depth = 48;   // binned SAH down to this depth, object-median splits (balanced) below
    size_t max_leaf = 1;  // 1: one item (a reference leaf) per leaf, leafref = the item's; > 1: items are triangles and
                          // a leaf is a run of idx, leafref = (first position << 4) | count

    static double half_area(const double *lo, const double *hi) {
        const double x = hi[0] - lo[0], y = hi[1] - lo[1], z = hi[2] - lo[2];
        return x * y + y * z + z * x;
    }
    int32_t build(size_t b, size_t e, int depth) {
        BvhTmp nd;
        for (int k = 0; k < 3; k++) { nd.lo[k] = kInf; nd.hi[k] = -kInf; }
        double clo[3] = {kInf, kInf, kInf}, chi[3] = {-kInf, -kInf, -kInf};
        for (size_t i = b; i < e; i++) {
            const BvhItem &it = items[(size_t)idx[i]];
            for (int k = 0; k < 3; k++) {
                nd.lo[k] = std::min(nd.lo[k], it.lo[k]);
                nd.hi[k] = std::max(nd.hi[k], it.hi[k]);
                const double c = 0.5 * (it.lo[k] + it.hi[k]);
                clo[k] = std::min(clo[k], c);
                chi[k] = std::max(chi[k], c);
            }
        }
        const int32_t me = next.fetch_add(1);
        nodes[(size_t)me] = nd;
        if (e - b <= max_leaf) {
            nodes[(size_t)me].leafref = max_leaf == 1 ? items[(size_t)idx[b]].leafref : (int32_t)((b << 4) | (e - b));
            return me;
        }
        // binned SAH over the three axes (16 bins on the centroid range); cost = area * triangles
        constexpr int NB = 16;
        int best_axis = -1, best_bin = -1;
        double best_cost = 1e300;
        for (int ax = 0; ax < 3 && depth < sah_depth; ax++) {
            const double ext = chi[ax] - clo[ax];
            if (!(ext > 0)) continue;
            double blo[NB][3], bhi[NB][3];
            int64_t bw[NB];
            for (int q = 0; q < NB; q++) {
                bw[q] = 0;
                for (int k = 0; k < 3; k++) { blo[q][k] = kInf; bhi[q][k] = -kInf; }
            }
            for (size_t i = b; i < e; i++) {
                const BvhItem &it = items[(size_t)idx[i]];
                int q = (int)((0.5 * (it.lo[ax] + it.hi[ax]) - clo[ax]) / ext * NB);
                q = q < 0 ? 0 : (q >= NB ? NB - 1 : q);
                bw[q] += it.weight;
                for (int k = 0; k < 3; k++) {
                    blo[q][k] = std::min(blo[q][k], it.lo[k]);
                    bhi[q][k] = std::max(bhi[q][k], it.hi[k]);
                }
            }
            double rlo[NB][3], rhi[NB][3];
            int64_t rw[NB];
            double alo[3] = {kInf, kInf, kInf}, ahi[3] = {-kInf, -kInf, -kInf};
            int64_t aw = 0;
            for (int q = NB - 1; q >= 1; q--) {  // suffix unions: bins q..NB-1
                aw += bw[q];
                for (int k = 0; k < 3; k++) { alo[k] = std::min(alo[k], blo[q][k]); ahi[k] = std::max(ahi[k], bhi[q][k]); }
                rw[q] = aw;
                for (int k = 0; k < 3; k++) { rlo[q][k] = alo[k]; rhi[q][k] = ahi[k]; }
            }
            double llo[3] = {kInf, kInf, kInf}, lhi[3] = {-kInf, -kInf, -kInf};
            int64_t lw = 0;
            for (int q = 0; q + 1 < NB; q++) {  // split after bin q
                lw += bw[q];
                for (int k = 0; k < 3; k++) { llo[k] = std::min(llo[k], blo[q][k]); lhi[k] = std::max(lhi[k], bhi[q][k]); }
                if (lw == 0 || rw[q + 1] == 0) continue;
                const double cost = half_area(llo, lhi) * (double)lw + half_area(rlo[q + 1], rhi[q + 1]) * (double)rw[q + 1];
                if (cost < best_cost) { best_cost = cost; best_axis = ax; best_bin = q; }
            }
        }
        size_t mid;
        int axis;
        if (best_axis >= 0) {
            axis = best_axis;
            const double ext = chi[axis] - clo[axis], c0 = clo[axis];
            const int bin = best_bin;
            auto it = std::partition(idx.begin() + (long)b, idx.begin() + (long)e, [&](int32_t id) {
                const BvhItem &t = items[(size_t)id];
                int q = (int)((0.5 * (t.lo[axis] + t.hi[axis]) - c0) / ext * NB);
                q = q < 0 ? 0 : (q >= NB ? NB - 1 : q);
                return q <= bin;
            });
            mid = (size_t)(it - idx.begin());
        } else {  // coincident centroids or a very deep branch: object median on the widest axis
            axis = 0;
            for (int k = 1; k < 3; k++)
                if (chi[k] - clo[k] > chi[axis] - clo[axis]) axis = k;
            mid = b + (e - b) / 2;
            std::nth_element(idx.begin() + (long)b, idx.begin() + (long)mid, idx.begin() + (long)e, [&](int32_t p, int32_t q) {
                const BvhItem &P = items[(size_t)p], &Q = items[(size_t)q];
                const double cp = P.lo[axis] + P.hi[axis], cq = Q.lo[axis] + Q.hi[axis];
                return cp < cq || (cp == cq && p < q);
            });
        }
        if (mid == b || mid == e) mid = b + (e - b) / 2;
        int32_t l, r;
        if (depth < 3 && e - b > 16384) {  // the two halves touch disjoint slices of idx: build them side by side
            std::future<int32_t> fl = std::async(std::launch::async, [this, b, mid, depth] { return build(b, mid, depth + 1); });
            r = build(mid, e, depth + 1);
            l = fl.get();
        } else {
            l = build(b, mid, depth + 1);
            r = build(mid, e, depth + 1);
        }
        BvhTmp &m = nodes[(size_t)me];
        m.left = l;
        m.right = r;
        m.axis = axis;
        const BvhTmp &L = nodes[(size_t)l], &R = nodes[(size_t)r];
        m.left_is_lower = (L.lo[axis] + L.hi[axis]) <= (R.lo[axis] + R.hi[axis]);
        return me;
    }
    // The 4-wide form (WideNodeRec): a node's children are its two binary children, the larger-surfaced inner one of which is
    // replaced by ITS children until there are four or only leaves.  Returns the wide node's index; `need` = how deep the
    // walk's stack can get below it (descending into one child leaves at most the others waiting).
    int32_t emit_wide(int32_t n, std::vector<WideNodeRec> &out, int32_t &need) const {
        int32_t kids[4], nk = 0;
        if (nodes[(size_t)n].leafref >= 0) {
            kids[nk++] = n;  // a tree that is one leaf: a root with that one child
        } else {
            kids[nk++] = nodes[(size_t)n].left;
            kids[nk++] = nodes[(size_t)n].right;
            while (nk < 4) {
                int best = -1;
                double best_area = -1;
                for (int k = 0; k < nk; k++) {
                    const BvhTmp &c = nodes[(size_t)kids[k]];
                    if (c.leafref >= 0) continue;
                    const double a = half_area(c.lo, c.hi);
                    if (a > best_area) { best_area = a; best = k; }
                }
                if (best < 0) break;
                const BvhTmp &c = nodes[(size_t)kids[best]];
                kids[best] = c.left;
                kids[nk++] = c.right;
            }
        }
        const int32_t me = (int32_t)out.size();
        out.emplace_back();
        WideNodeRec w;
        std::memset(&w, 0, sizeof(w));
        need = 0;
        for (int k = 0; k < 4; k++) {
            if (k >= nk) { w.ref[k] = kWideNone; continue; }
            const BvhTmp &c = nodes[(size_t)kids[k]];
            w.lox[k] = round_down(c.lo[0] - kBoxPad); w.loy[k] = round_down(c.lo[1] - kBoxPad); w.loz[k] = round_down(c.lo[2] - kBoxPad);
            w.hix[k] = round_up(c.hi[0] + kBoxPad);   w.hiy[k] = round_up(c.hi[1] + kBoxPad);   w.hiz[k] = round_up(c.hi[2] + kBoxPad);
            if (c.leafref >= 0) {
                w.ref[k] = c.leafref;
                need = std::max(need, nk - 1);
            } else {
                int32_t below = 0;
                const int32_t child = emit_wide(kids[k], out, below);
                w.ref[k] = ~child;
                need = std::max(need, nk - 1 + below);
            }
        }
        out[(size_t)me] = w;
        return me;
    }
    // preorder with skip links into out[0..): `cur` is the next free slot of this octant's array
    void emit(int32_t n, int oct, NodeRec *out, int32_t &cur) const {
        const BvhTmp &t = nodes[(size_t)n];
        const int32_t me = cur++;
        NodeRec nr;
        for (int k = 0; k < 3; k++) {
            nr.lo[k] = round_down(t.lo[k] - kBoxPad);
            nr.hi[k] = round_up(t.hi[k] + kBoxPad);
        }
        nr.skip = 0;
        nr.leaf = t.leafref;
        out[me] = nr;
        if (t.leafref < 0) {
            const bool positive = ((oct >> t.axis) & 1) == 0;  // octant bit set = direction component negative
            const bool left_first = (positive == t.left_is_lower);
            emit(left_first ? t.left : t.right, oct, out, cur);
            emit(left_first ? t.right : t.left, oct, out, cur);
        }
        out[me].skip = cur;
    }
};
}  // namespace

void HostTree::build_bvh(bool opaque) {
    bvh.clear();
    otris.clear();
    bvh_nodes = 0;
    tri_level = opaque;
    BvhBuilder B;
    std::vector<int32_t> leaf_first;  // tri_level: first leaf-order index of the reference leaf of triangle k
    if (opaque) {
        B.max_leaf = 4;
        leaf_first.resize(tris.size());
        for (const NodeRec &nd : nodes)
            if (nd.leaf >= 0)
                for (int32_t k = nd.leaf >> 4; k < (nd.leaf >> 4) + (nd.leaf & 15); k++) leaf_first[(size_t)k] = nd.leaf >> 4;
        for (size_t k = 0; k < tris.size(); k++) {  // triangles in the reference's leaf order
            const double *t = &tri9[9 * (size_t)leaf_ids[k]];
            BvhItem it;
            for (int c = 0; c < 3; c++) {
                it.lo[c] = lo3(t[c], t[3 + c], t[6 + c]);
                it.hi[c] = hi3(t[c], t[3 + c], t[6 + c]);
            }
            it.leafref = -1;
            it.weight = 1;
            B.items.push_back(it);
        }
    } else {
        for (size_t n = 0; n < nodes.size(); n++)
            if (nodes[n].leaf >= 0 && (nodes[n].leaf & 15) > 0) {  // the reference's leaves, in their own order
                BvhItem it;
                for (int k = 0; k < 3; k++) {
                    it.lo[k] = bbox[6 * n + 2 * (size_t)k];
                    it.hi[k] = bbox[6 * n + 2 * (size_t)k + 1];
                }
                it.leafref = nodes[n].leaf;
                it.weight = nodes[n].leaf & 15;
                B.items.push_back(it);
            }
    }
    if (B.items.empty()) return;
    B.idx.resize(B.items.size());
    B.nodes.resize(2 * B.items.size());
    wide.clear();
    wide_stack = 0;
    int32_t root = 0;
    // An opaque owner's hierarchy is walked in its 4-wide form with a stack of kWideStack entries.  SAH splits can chain
    // (a long sliver next to many small triangles peels one item per level); if the stack bound is missed the build is
    // repeated with SAH confined to the top levels -- object-median splits below are balanced, and with none at all the
    // bound is 1.5 log2(n) < 48.
    for (int cap : {48, 8, 0}) {
        for (size_t i = 0; i < B.idx.size(); i++) B.idx[i] = (int32_t)i;
        B.next.store(0);
        B.sah_depth = cap;
        root = B.build(0, B.items.size(), 0);
        if (!opaque) break;
        wide.clear();
        B.emit_wide(root, wide, wide_stack);
        if (wide_stack < kWideStack) break;
    }
    bvh_nodes = B.next.load();
    bvh.resize((size_t)bvh_nodes * 8);
    {  // the eight layouts are independent: one thread each
        std::vector<std::thread> th;
        for (int oct = 0; oct < 8; oct++)
            th.emplace_back([&, oct] {
                int32_t cur = 0;
                B.emit(root, oct, bvh.data() + (size_t)oct * (size_t)bvh_nodes, cur);
            });
        for (auto &t : th) t.join();
    }
    if (opaque) {  // triangle records in the hierarchy's own order, each with its rank for ties
        otris.resize(tris.size());
        for (size_t j = 0; j < otris.size(); j++) {
            const int32_t k = B.idx[j];
            otris[j].t = tris[(size_t)k];
            otris[j].k = k;
            otris[j].leaf = leaf_first[(size_t)k];
        }
    }
}

// Grid-ordered view of a bump floor's triangles (call after build()).
void HostTree::build_hfield(int nx, int nz, double x0, double z0, double hx, double hz) {
    const size_t ntri = tri9.size() / 9;
    // a grid walk needs a proper grid: positive, finite pitch (a texture with lenx <= 0 keeps the tree)
    if (nx < 1 || nz < 1 || ntri != (size_t)nx * nz * 2 || tris.size() != ntri) return;
    if (!(hx > 0) || !(hz > 0) || !std::isfinite(hx) || !std::isfinite(hz) || !std::isfinite(x0) || !std::isfinite(z0)) return;
    std::vector<int32_t> pos(ntri), leaf_of(ntri);
    for (size_t k = 0; k < ntri; k++) pos[(size_t)leaf_ids[k]] = (int32_t)k;
    for (const NodeRec &nd : nodes)
        if (nd.leaf >= 0) {  // a leaf is identified by its first triangle's index, which grows with the leaf sequence
            const int32_t first = nd.leaf >> 4, cnt = nd.leaf & 15;
            for (int32_t k = first; k < first + cnt; k++) leaf_of[(size_t)k] = first;
        }
    hcells.resize((size_t)nx * nz);
    hcell_y.resize((size_t)nx * nz);
    double ylo = kInf, yhi = -kInf;
    for (size_t c = 0; c < hcells.size(); c++) {
        double clo = kInf, chi = -kInf;
        for (int q = 0; q < 2; q++)
            for (int v = 0; v < 3; v++) {
                const double y = tri9[9 * (2 * c + q) + 3 * v + 1];
                clo = std::min(clo, y);
                chi = std::max(chi, y);
            }
        hcell_y[c].lo = round_down(clo);
        hcell_y[c].hi = round_up(chi);
    }
    for (size_t c = 0; c < hcells.size(); c++)
        for (int q = 0; q < 2; q++) {
            const int32_t k = pos[2 * c + q];
            hcells[c].t[q] = tris[(size_t)k];
            hcells[c].k[q] = k;
            hcells[c].leaf[q] = leaf_of[(size_t)k];
            const double *t = &tri9[9 * (2 * c + q)];
            for (int v = 0; v < 3; v++) {
                if (t[3 * v + 1] < ylo) ylo = t[3 * v + 1];
                if (t[3 * v + 1] > yhi) yhi = t[3 * v + 1];
            }
        }
    hfield.cell_begin = 0;
    hfield.nx = nx; hfield.nz = nz;
    hfield.x0 = x0; hfield.z0 = z0;
    hfield.hx = hx; hfield.hz = hz;
    hfield.ihx = 1.0 / hx; hfield.ihz = 1.0 / hz;
    hfield.ylo = ylo; hfield.yhi = yhi;
    is_hfield = true;
}

int64_t ref_node_count(int64_t n) {
    return n < kMinKd ? 1 : 1 + ref_node_count(n / 2) + ref_node_count(n - n / 2);
}

namespace {
struct BuildTimer {  // adds the enclosing scope's wall-clock to HostScene::host_build_ms
    double &acc;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    explicit BuildTimer(double &a) : acc(a) {}
    ~BuildTimer() { acc += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};
}  // namespace

bool load_mesh_file(const char *file, double a, const double b[3], int type, std::vector<double> &tri9,
                    std::string &err) {
    tri9.clear();
    Scanner sc;
    if (!sc.open(file)) return true;  // freopen() fails silently in the reference => empty mesh
    auto put = [&](const double *v) {  // (x, y, -z) * a + b   (objects.h:348,365,384)
        tri9.push_back(v[0] * a + b[0]);
        tri9.push_back(v[1] * a + b[1]);
        tri9.push_back(-v[2] * a + b[2]);
    };
    if (type == 0) {  // begin / vertex x y z (x3) / end   (objects.h:346)
        while (!sc.at_end()) {
            double v[9];
            if (!sc.word("begin")) { err = "type-0 mesh: expected 'begin'"; return false; }
            for (int k = 0; k < 3; k++) {
                if (!sc.word("vertex")) { err = "type-0 mesh: expected 'vertex'"; return false; }
                for (int c = 0; c < 3; c++)
                    if (!sc.real(v[3 * k + c])) { err = "type-0 mesh: bad coordinate"; return false; }
            }
            if (!sc.word("end")) { err = "type-0 mesh: expected 'end'"; return false; }
            put(v); put(v + 3); put(v + 6);
        }
        return true;
    }
    if (type != 1 && type != 2) { err = "unknown typeofdata"; return false; }
    long nv = 0, nf = 0;
    if (!sc.integer(nv) || nv < 0) { err = "mesh: bad vertex count"; return false; }
    std::vector<double> verts((size_t)nv * 3);
    for (long i = 0; i < nv; i++) {
        if (!sc.word("v")) { err = "mesh: expected 'v'"; return false; }
        for (int c = 0; c < 3; c++)
            if (!sc.real(verts[3 * (size_t)i + c])) { err = "mesh: bad vertex"; return false; }
    }
    if (type == 2) {  // optional vn / vt records (objects.h:387-392)
        double q;
        while (sc.word("vn")) { if (!sc.real(q) || !sc.real(q) || !sc.real(q)) { err = "mesh: bad vn"; return false; } }
        while (sc.word("vt")) { if (!sc.real(q) || !sc.real(q)) { err = "mesh: bad vt"; return false; } }
    }
    if (!sc.integer(nf) || nf < 0) { err = "mesh: bad face count"; return false; }
    for (long i = 0; i < nf; i++) {
        if (!sc.word("f")) { err = "mesh: expected 'f'"; return false; }
        long id[3];
        for (int k = 0; k < 3; k++) {
            if (!sc.integer(id[k])) { err = "mesh: bad face index"; return false; }
            if (type == 2) sc.skip_token();  // the "/b/c" tail of "a/b/c" (objects.h:397)
            if (id[k] < 1 || id[k] > nv) { err = "mesh: face index out of range"; return false; }
        }
        for (int k = 0; k < 3; k++) put(&verts[3 * (size_t)(id[k] - 1)]);
    }
    return true;
}

int HostScene::add_sphere(const double c[3], double r, const double sc[3], double refl, double transp) {
    ObjRec o = blank_obj(KIND_SPHERE, sc, refl, transp);
    set3(o.a, c);
    o.s0 = r * r;  // radius2(r * r), objects.h:35
    objs.push_back(o);
    return (int)objs.size() - 1;
}

int HostScene::add_texture(const uint8_t *rgb, int rows, int cols, const double n[3], const double p[3], double lx,
                           double ly, int isbump) {
    if (!rgb || rows <= 0 || cols <= 0) { error = "texture: empty image"; return -1; }
    HostTexture t;
    t.rgb.assign(rgb, rgb + (size_t)rows * cols * 3);
    t.rows = rows;
    t.cols = cols;
    set3(t.n, n);
    set3(t.p, p);
    t.lenx = lx;
    t.leny = ly;
    t.isbump = isbump != 0;
    textures.push_back(std::move(t));
    return (int)textures.size() - 1;
}

int HostScene::add_plane(const double p[3], const double n[3], const double sc[3], double refl, double transp,
                         int tex) {
    if (tex >= (int)textures.size()) { error = "plane: unknown texture id"; return -1; }
    ObjRec o = blank_obj(KIND_PLANE, sc, refl, transp);
    set3(o.a, p);
    set3(o.b, n);
    o.tex = tex < 0 ? -1 : tex;
    for (int k = 0; k < 3; k++)  // exactly axis-aligned, exactly unit: the fast path of the scene walk
        if ((n[k] == 1.0 || n[k] == -1.0) && n[(k + 1) % 3] == 0.0 && n[(k + 2) % 3] == 0.0) o.axis = k;
    if (tex >= 0 && textures[tex].isbump && std::fabs(n[1] - 1.0) < 1e-5) {
        // Displacement mesh of a bump-mapped floor (objects.h:482-503): one quad per 3x3 texel block,
        // split into triangles (a,b,c) and (d,b,c); heights 0.5*(1-exp(-3.3*luma)) (texture.h:28-35).
        const HostTexture &tx = textures[tex];
        const int R = tx.rows, C = tx.cols, step = 3;
        const bool grid_ok = C / step - 1 >= 1 && R / step - 1 >= 1 && tx.lenx * step / C > 0 && tx.leny * step / R > 0 &&
                             std::isfinite(tx.lenx * step / C) && std::isfinite(tx.leny * step / R) && std::isfinite(tx.p[0]) &&
                             std::isfinite(tx.p[2]);
        if (build_mode == 1 && transp < kEps && grid_ok) {
            // row f3: heights, vertices and grid cells are generated on the device at commit (cgrt_devbuild.hpp)
            HostTree tree;
            tree.dev_kind = 2;
            tree.dev_tex = tex;
            tree.dev_plane_y = p[1];
            tree.dev_ntri = 2 * (int64_t)(C / step - 1) * (R / step - 1);
            tree.is_hfield = true;
            tree.hfield.cell_begin = 0;
            tree.hfield.nx = C / step - 1;
            tree.hfield.nz = R / step - 1;
            tree.hfield.x0 = tx.p[0];
            tree.hfield.z0 = tx.p[2];
            tree.hfield.hx = tx.lenx * step / C;
            tree.hfield.hz = tx.leny * step / R;
            tree.hfield.ihx = 1.0 / tree.hfield.hx;
            tree.hfield.ihz = 1.0 / tree.hfield.hz;
            trees.push_back(std::move(tree));
            o.tree = (int)trees.size() - 1;
            objs.push_back(o);
            return (int)objs.size() - 1;
        }
        BuildTimer timer(host_build_ms);
        std::vector<double> height((size_t)R * C);
        for (int i = 0; i < R; i++)
            for (int j = 0; j < C; j++) {
                const uint8_t *q = &tx.rgb[3 * ((size_t)i * C + j)];
                double luma = (0.299 * ((double)q[0] / 256.0) + 0.587 * ((double)q[1] / 256.0) +
                               0.114 * ((double)q[2] / 256.0));
                double h = 1 - std::exp(-3.3 * luma);
                height[(size_t)i * C + j] = h * 0.5;
            }
        HostTree tree;
        for (int i = 0; i < R / step - 1; i++)
            for (int j = 0; j < C / step - 1; j++) {
                const double x1 = tx.p[0] + tx.lenx * j * step / C;
                const double x2 = tx.p[0] + tx.lenx * (j + 1) * step / C;
                const double z1 = tx.p[2] + tx.leny * i * step / R;
                const double z2 = tx.p[2] + tx.leny * (i + 1) * step / R;
                const double ya = height[(size_t)(i * step) * C + j * step] + p[1];
                const double yb = height[(size_t)(i * step) * C + (j + 1) * step] + p[1];
                const double yc = height[(size_t)((i + 1) * step) * C + j * step] + p[1];
                const double yd = height[(size_t)((i + 1) * step) * C + (j + 1) * step] + p[1];
                const double A[3] = {x1, ya, z1}, B[3] = {x2, yb, z1}, Cc[3] = {x1, yc, z2}, D[3] = {x2, yd, z2};
                for (const double *v : {A, B, Cc}) tree.tri9.insert(tree.tri9.end(), v, v + 3);
                for (const double *v : {D, B, Cc}) tree.tri9.insert(tree.tri9.end(), v, v + 3);
            }
        tree.build(transp < kEps, false);
        tree.build_hfield(C / step - 1, R / step - 1, tx.p[0], tx.p[2], tx.lenx * step / C, tx.leny * step / R);
        if (!(tree.is_hfield && transp < kEps)) tree.build_bvh(transp < kEps);  // an opaque floor is walked as a grid
        trees.push_back(std::move(tree));
        o.tree = (int)trees.size() - 1;
    }
    objs.push_back(o);
    return (int)objs.size() - 1;
}

// Spheres around groups of triangles: the set is halved `depth` times at the median centroid along the longest axis of
// the group's vertices; each group gives (centre of its vertices' box, largest vertex distance + 1e-3).  out: 4 doubles each.
static constexpr int kCoverDepth = 6;
static void cover_spheres(const std::vector<double> &tri9, int depth, std::vector<double> &out) {
    const size_t n = tri9.size() / 9;
    if (n == 0) return;
    std::vector<uint32_t> idx(n);
    for (size_t i = 0; i < n; i++) idx[i] = (uint32_t)i;
    struct Range { size_t b, e; int d; };
    std::vector<Range> todo{{0, n, depth}};
    while (!todo.empty()) {
        const Range r = todo.back();
        todo.pop_back();
        double lo[3] = {kInf, kInf, kInf}, hi[3] = {-kInf, -kInf, -kInf};
        for (size_t i = r.b; i < r.e; i++) {
            const double *t = &tri9[9 * (size_t)idx[i]];
            for (int v = 0; v < 9; v++) {
                lo[v % 3] = std::min(lo[v % 3], t[v]);
                hi[v % 3] = std::max(hi[v % 3], t[v]);
            }
        }
        if (r.d > 0 && r.e - r.b >= 128) {
            int ax = 0;
            for (int k = 1; k < 3; k++)
                if (hi[k] - lo[k] > hi[ax] - lo[ax]) ax = k;
            const size_t mid = r.b + (r.e - r.b) / 2;
            std::nth_element(idx.begin() + r.b, idx.begin() + mid, idx.begin() + r.e, [&](uint32_t a, uint32_t b) {
                const double *ta = &tri9[9 * (size_t)a], *tb = &tri9[9 * (size_t)b];
                return ta[ax] + ta[3 + ax] + ta[6 + ax] < tb[ax] + tb[3 + ax] + tb[6 + ax];
            });
            todo.push_back({r.b, mid, r.d - 1});
            todo.push_back({mid, r.e, r.d - 1});
            continue;
        }
        const double c[3] = {0.5 * (lo[0] + hi[0]), 0.5 * (lo[1] + hi[1]), 0.5 * (lo[2] + hi[2])};
        double r2 = 0;
        for (size_t i = r.b; i < r.e; i++) {
            const double *t = &tri9[9 * (size_t)idx[i]];
            for (int v = 0; v < 3; v++) {
                const double dx = t[3 * v] - c[0], dy = t[3 * v + 1] - c[1], dz = t[3 * v + 2] - c[2];
                r2 = std::max(r2, dx * dx + dy * dy + dz * dz);
            }
        }
        out.insert(out.end(), {c[0], c[1], c[2], std::sqrt(r2) + 1e-3});
    }
}

int HostScene::add_mesh_triangles(const double *tri9, int ntri, const double sc[3], double refl, double transp,
                                  int type) {
    if (ntri < 0 || (ntri > 0 && !tri9)) { error = "mesh: bad triangle array"; return -1; }
    ObjRec o = blank_obj(KIND_MESH, sc, refl, transp);
    o.aux = type;
    HostTree tree;
    tree.tri9.assign(tri9, tri9 + (size_t)ntri * 9);
    if (build_mode == 1 && transp < kEps && ntri > 0) {
        // row f3: hierarchy, bounding sphere and cover spheres are built on the device at commit (cgrt_devbuild.hpp)
        tree.dev_kind = 1;
        tree.dev_ntri = ntri;
        tree.dev_obj = (int)objs.size();
        tree.tri_level = true;
        o.a[0] = o.a[1] = o.a[2] = 0;
        o.s0 = -1.0;  // until the commit has the sphere
        trees.push_back(std::move(tree));
        o.tree = (int)trees.size() - 1;
        objs.push_back(o);
        return (int)objs.size() - 1;
    }
    BuildTimer timer(host_build_ms);
    tree.build(transp < kEps);  // objects.h:402; an opaque owner gets the triangle-level hierarchy
    {   // bounding sphere of the mesh (a = centre, s0 = radius^2): the scene walk's division-free early-out.  A triangle can
        // only be hit at one of its own points, so a sphere around every vertex (plus 1e-3, far above the triangle test's
        // rounding) is enough: a ray that misses it cannot hit, whatever boxes it crosses.
        const size_t first = cover.size();
        cover_spheres(tree.tri9, 0, cover);
        if (ntri == 0) {
            o.a[0] = o.a[1] = o.a[2] = 0;
            o.s0 = -1.0;  // empty mesh: nothing to hit
        } else {
            for (int k = 0; k < 3; k++) o.a[k] = cover[first + k];
            o.s0 = cover[first + 3] * cover[first + 3] * (1 + 1e-9);
            cover.resize(first);
            // the same bound in pieces (<= 2^kCoverDepth spheres over median-split groups of triangles), for classify_kernel:
            // a long thin mesh fills a small part of its one sphere's silhouette
            cover_spheres(tree.tri9, kCoverDepth, cover);
        }
    }
    trees.push_back(std::move(tree));
    o.tree = (int)trees.size() - 1;
    objs.push_back(o);
    return (int)objs.size() - 1;
}

int HostScene::add_mesh_file(const char *file, double a, const double b[3], const double sc[3], double refl,
                             double transp, int type) {
    std::vector<double> t9;
    if (!file || !load_mesh_file(file, a, b, type, t9, error)) return -2;
    return add_mesh_triangles(t9.data(), (int)(t9.size() / 9), sc, refl, transp, type);
}

int HostScene::add_bezier(const double *cp3, int ncp, const double pos[3], const double sc[3], double refl,
                          double transp) {
    if (!cp3 || ncp < 1 || ncp > 6) { error = "bezier: 1..6 control points"; return -1; }  // bezier.h:46
    ObjRec o = blank_obj(KIND_BEZIER, sc, refl, transp);
    set3(o.a, pos);
    BezierRec b;
    std::memset(&b, 0, sizeof(b));
    b.ncp = ncp;
    double max_z = -kInf, max_y = -kInf, min_y = kInf;  // bezier.h:50-63
    for (int i = 0; i < ncp; i++) {
        for (int k = 0; k < 3; k++) b.cp[i][k] = cp3[3 * i + k];
        if (b.cp[i][2] > max_z) max_z = b.cp[i][2];
        if (b.cp[i][1] > max_y) max_y = b.cp[i][1];
        if (b.cp[i][1] < min_y) min_y = b.cp[i][1];
    }
    b.box[0] = -max_z + pos[0];  // xmin
    b.box[1] = max_z + pos[0];   // xmax
    b.box[2] = min_y + pos[1];
    b.box[3] = max_y + pos[1];
    b.box[4] = -max_z + pos[2];
    b.box[5] = max_z + pos[2];
    o.b[0] = b.cp[ncp - 1][2];  // radius of the cap disc (bezier.h:277)
    // bounds of the surface piece by piece (BezSlabRec): the control points of the curve restricted to [u0, u1] by two de
    // Casteljau subdivisions; a Bezier curve lies in the hull of its control points
    for (int k = 0; k < kBezSlabs; k++) {
        const double u0 = (double)k / kBezSlabs, u1 = (double)(k + 1) / kBezSlabs;
        double q[6][3];
        for (int i = 0; i < ncp; i++)
            for (int c = 0; c < 3; c++) q[i][c] = b.cp[i][c];
        auto split_left = [&](double t) {  // q := control points of the part [0, t]
            double w[6][3], out[6][3];
            std::memcpy(w, q, sizeof(w));
            for (int lvl = 0; lvl < ncp; lvl++) {
                for (int c = 0; c < 3; c++) out[lvl][c] = w[0][c];
                for (int i = 0; i + 1 < ncp - lvl; i++)
                    for (int c = 0; c < 3; c++) w[i][c] = (1 - t) * w[i][c] + t * w[i + 1][c];
            }
            std::memcpy(q, out, sizeof(out));
        };
        auto split_right = [&](double t) {  // q := control points of the part [t, 1]
            double w[6][3], out[6][3];
            std::memcpy(w, q, sizeof(w));
            for (int lvl = 0; lvl < ncp; lvl++) {
                for (int c = 0; c < 3; c++) out[ncp - 1 - lvl][c] = w[ncp - 1 - lvl][c];
                for (int i = 0; i + 1 < ncp - lvl; i++)
                    for (int c = 0; c < 3; c++) w[i][c] = (1 - t) * w[i][c] + t * w[i + 1][c];
            }
            std::memcpy(q, out, sizeof(out));
        };
        split_left(u1);
        if (u0 > 0) split_right(u0 / u1);
        BezSlabRec sl;
        sl.ylo = kInf; sl.yhi = -kInf;
        double rmax = 0;
        for (int i = 0; i < ncp; i++) {
            sl.ylo = std::min(sl.ylo, q[i][1]);
            sl.yhi = std::max(sl.yhi, q[i][1]);
            rmax = std::max(rmax, std::fabs(q[i][2]));
        }
        const double span = std::max(1.0, std::max(rmax, std::max(std::fabs(sl.ylo), std::fabs(sl.yhi))));
        const double grow = 1e-3 * span;  // far above the 1e-4 acceptance radius and the rounding of the subdivision
        sl.ylo -= grow;
        sl.yhi += grow;
        sl.r2 = (rmax + grow) * (rmax + grow);
        sl.pad = 0;
        bez_slabs.push_back(sl);
    }
    beziers.push_back(b);
    o.aux = (int)beziers.size() - 1;
    objs.push_back(o);
    return (int)objs.size() - 1;
}

}  // namespace cgrt
