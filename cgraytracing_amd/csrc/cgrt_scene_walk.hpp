// Nearest hit over the top-level object list (main.cpp:55-63) and Texture::color.  Part of libcgrt.so (cgrt_hip.hip).
#ifndef CGRT_SCENE_WALK_HPP
#define CGRT_SCENE_WALK_HPP
#include "cgrt_bezier.hpp"
#include "cgrt_traverse.hpp"

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
    // PRE instantiations only (the unit-queue body): this lane's hit in object pre_obj is already known (primary_walk_kernel)
    bool pre_valid = false;
    int32_t pre_obj = -1, pre_tri = -1;
    double pre_len = 0;
    int32_t pre_counter = 0;  // improvements (a transparent owner's normal depends on their parity)
};

// TriangleMesh::intersect's early-out (see intersect_scene): false = the ray cannot hit a triangle of the mesh bounded by the
// sphere (ob.a, ob.s0)
__device__ __forceinline__ bool mesh_may_hit(const ObjRec &ob, V3 o, V3 d) {
    const V3 lc = ld3(ob.a) - o;
    const double tca = dot(lc, d), l2 = dot(lc, lc), dd = dot(d, d), r2 = ob.s0;
    return !(tca < 0 && l2 > r2) && !(l2 * dd - tca * tca > r2 * dd);
}

// Sphere::intersect, objects.h:45-68: the hit distance, or +inf-like kInf (never < nearest) on a miss
__device__ __forceinline__ double sphere_len(V3 centre, double r2, V3 o, V3 d) {
    const V3 l = centre - o;
    const double tca = dot(l, d);
    const double l2 = dot(l, l);
    double len = kInf;
    if (!(tca < 0 && l2 > r2)) {
        const double d2 = l2 - tca * tca;
        if (!(d2 > r2)) {
            const double thc = sqrt_cr(r2 - d2);
            const double t0 = tca - thc, t1 = tca + thc;
            len = (t0 < 0) ? t1 : t0;
        }
    }
    return len;
}
__device__ __forceinline__ double sphere_len(const ObjRec &ob, V3 o, V3 d) { return sphere_len(ld3(ob.a), ob.s0, o, d); }

// Plane::intersect's distance, objects.h:505-507: len = ((p - o) . n) / (d . n).  For a normal that is exactly +-e_k the two dot
// products are +-(p_k - o_k) and +-d_k to the bit -- the other products are +-0 and adding +-0 to a non-zero double changes
// nothing -- and (-x) / (-y) rounds like x / y, so len is (p_k - o_k) / d_k: one subtraction and the division instead of three
// subtractions and two dot products (five planes a ray: ~60 of a plane-bound ray's instructions).  Only when that numerator or
// denominator is ZERO does the sign of the zero the general expression produces matter (+-inf, NaN): those lanes -- a ray
// exactly parallel to the plane, an origin exactly in it -- take the general expression.  Called by all lanes of the wave.
__device__ __forceinline__ double plane_len(const ObjRec &ob, V3 pn, V3 o, V3 d) {
    const int ax = __builtin_amdgcn_readfirstlane(ob.axis);
    double len;
    if (ax >= 0) {
        const double ok = ax == 0 ? o.x : (ax == 1 ? o.y : o.z), dk = ax == 0 ? d.x : (ax == 1 ? d.y : d.z);
        const double num = ob.a[ax] - ok;
        len = num / dk;
        const bool zero = (num == 0.0) || (dk == 0.0);
        if (__ballot(zero) != 0ull) {
            const V3 dd = ld3(ob.a) - o;
            const double general = dot(dd, pn) / dot(d, pn);
            if (zero) len = general;
        }
    } else {
        const V3 dd = ld3(ob.a) - o;
        len = dot(dd, pn) / dot(d, pn);
    }
    return len;
}

// A run of axis-aligned planes (DeviceScene::prun_begin/end), tested as a group.  Each plane's distance is the correctly rounded
// quotient (p_k - o_k) / d_k (plane_len above): five planes, five fp64 divisions of ~28 instructions, of which only the smallest
// positive one is used.  Here every plane first gets an APPROXIMATE distance t~ = fl32(p_k - o_k) * rcp32(d_k), relative error
// below 2^-21 (two conversions, the hardware reciprocal's 1 ulp, one product): if the smallest positive t~ is below the second
// smallest by a factor of more than 1 + 2^-14, the exact quotient of that plane is smaller than every other plane's by more than
// 1 + 2^-15, so its ROUNDED quotient is strictly the smallest too -- it is what the loop over the planes would have kept, whatever
// the order -- and one division yields it.  Lanes for which that does not hold (two planes within 2^-14 of each other: a ray
// through an edge of the room; a zero or tiny numerator or denominator, where plane_len takes its general form; distances beyond
// 1e9, near kInf) are told apart (`unsure`) and go through the planes one by one as before.  Called by all lanes of the wave.
struct PlaneRunHit {
    double len;   // the winner's distance (only if id >= 0)
    int id;       // the plane of the run with the smallest positive distance, or -1
    bool unsure;  // this lane must take the planes one by one
};
__device__ __forceinline__ PlaneRunHit plane_run(const ObjRec *__restrict__ objs, int begin, int end, V3 o, V3 d) {
    const float r32[3] = {__builtin_amdgcn_rcpf((float)d.x), __builtin_amdgcn_rcpf((float)d.y), __builtin_amdgcn_rcpf((float)d.z)};
    const float inf32 = __int_as_float(0x7f800000);
    float t1 = inf32, t2 = inf32;  // smallest and second smallest positive approximate distance
    double wnum = 0.0;
    int wax = 0;
    PlaneRunHit h;
    h.id = -1;
    // zero or tiny operands (plane_len's general form; fp32 underflow), huge or non-finite distances: not decided here
    h.unsure = !(fabs(d.x) > 1e-30) || !(fabs(d.y) > 1e-30) || !(fabs(d.z) > 1e-30);
    for (int j = begin; j < end; j++) {
        const int ax = __builtin_amdgcn_readfirstlane(objs[j].axis);
        const double p = objs[j].a[ax];
        double num;
        float t;
        if (ax == 0) { num = p - o.x; t = (float)num * r32[0]; }       // (wave-uniform branches)
        else if (ax == 1) { num = p - o.y; t = (float)num * r32[1]; }
        else { num = p - o.z; t = (float)num * r32[2]; }
        h.unsure = h.unsure || !(fabs(num) > 1e-30) || !(fabsf(t) < 1e9f);
        if (t > 0.f) {
            if (t < t1) {
                t2 = t1;
                t1 = t;
                h.id = j;
                wnum = num;
                wax = ax;
            } else if (t < t2) {
                t2 = t;
            }
        }
    }
    h.unsure = h.unsure || (h.id >= 0 && !(t2 > t1 * 1.00006103515625f));  // 1 + 2^-14
    h.len = wnum / (wax == 0 ? d.x : (wax == 1 ? d.y : d.z));
    return h;
}

// The first 56 bytes of an ObjRec -- what the sphere loop needs of an object that is not in LDS (one scalar load)
struct ObjHead {
    double a[3], b[3], s0;
};
// What shading needs of the object a lane has hit: from LDS, or for an object beyond the LDS list from the uploaded array
struct ObjMat {
    V3 col;
    double refl, transp;
    int32_t kind, tex;
};
template <bool SPILL>
__device__ __forceinline__ ObjMat load_mat(const ObjRec *__restrict__ lobjs, int n_lds, const ObjRec *__restrict__ gobjs, int id) {
    ObjMat m;
    if (!SPILL || id < n_lds) {
        const ObjRec &r = lobjs[id];
        m.col = ld3(r.col); m.refl = r.refl; m.transp = r.transp; m.kind = r.kind; m.tex = r.tex;
    } else {
        const ObjRec &r = gobjs[id];
        m.col = ld3(r.col); m.refl = r.refl; m.transp = r.transp; m.kind = r.kind; m.tex = r.tex;
    }
    return m;
}
template <bool SPILL>
__device__ __forceinline__ V3 load_centre(const ObjRec *__restrict__ lobjs, int n_lds, const ObjRec *__restrict__ gobjs, int id) {
    if (!SPILL || id < n_lds) return ld3(lobjs[id].a);
    return ld3(gobjs[id].a);
}

// bytes of LDS the object list takes in a workgroup: the resident records and, when some objects are not resident, one staging
// record per wave
__host__ __device__ inline size_t obj_list_lds(const DeviceScene &sc, int waves) {
    return ((size_t)sc.n_lds + (sc.n_objs > sc.n_lds ? (size_t)waves : 0)) * sizeof(ObjRec);
}

// One small tree (<= kNodeCache nodes: the bunny's 255, a coarse bump floor) is staged whole in LDS by every workgroup:
// traversal is latency-bound on dependent node fetches, and an LDS read costs ~100 cycles against ~500-800 for L1/L2.
static constexpr int kNodeCache = 256;  // 8 KiB of 32-byte nodes

// per-workgroup LDS resources handed down to the scene walk
struct LdsAux {
    volatile BezLds *bl;    // this wave's Bezier scratch (BEZ variants) or nullptr
    const NodeRec *lnodes;  // LDS copy of tree sc.cached_tree's nodes, or nullptr
    uint2 *wstack = nullptr;  // LDS part of the 4-wide walk's stack ([entry][thread]), or nullptr
    ObjRec *spill = nullptr;  // this wave's staging record for objects beyond the LDS list (scenes with more than kLdsObjsMax objects)
};

// tree traversal entry; `on` = this lane really has a ray for this tree (all lanes of the wave call it).
// A wave-synchronous variant (one shared node sequence, records fetched through the scalar cache) was measured
// and dropped: with sub-pixel triangles the union of 64 rays' leaf sets approaches their sum.
// `opaque` (wave-uniform): the owning object's transparency is < eps, so the pruned traversal applies with
// `bound` = nearest hit distance already known for this ray.
// HFONLY: the light-tile variant of scenes whose only reachable trees are opaque bump floors walked as grids
// (DeviceScene::light_hf_only): none of the hierarchy-walk code -- and none of its registers and scratch -- exists in it.
template <bool STATS, bool HFONLY = false>
__device__ __forceinline__ TreeHit tree_hit(const DeviceScene &sc, const LdsAux &aux, int tr, bool opaque, double bound,
                                            bool on, V3 o, V3 d, V3 inv, uint32_t &n_node, uint32_t &n_tri) {
    const TreeRec T = load_uniform(sc.trees + tr);
    TreeHit none;
    none.len = kInf;
    none.tri = -1;
    none.counter = 0;
    UTILP(1, on);
    if (!on) return none;
    if (HFONLY || (opaque && T.hfield >= 0)) {  // bump floor: walk the grid instead of the tree (same triangles, same test)
        const HFieldRec H = load_uniform(sc.hfields + T.hfield);
        return hfield_intersect<STATS>(H, sc.hcells + H.cell_begin, sc.hcell_y + H.cell_begin, o, d, inv, bound, n_node, n_tri);
    }
    const bool cached = aux.lnodes != nullptr && tr == sc.cached_tree;
    const Ray32 r32 = make_ray32(o, inv, T.bmax);
    // the copy of the hierarchy whose children are ordered near-to-far for this ray's direction octant (only worth a
    // per-lane base address where order matters, i.e. for the pruned traversal)
    const int oct = (T.noct == 8 && opaque) ? ((d.x < 0 ? 1 : 0) | (d.y < 0 ? 2 : 0) | (d.z < 0 ? 4 : 0)) : 0;
    const NodeRec *nodes = sc.nodes + T.node_begin + (size_t)oct * (size_t)T.nnodes;
    const TriRec *tris = sc.tris + T.tri_begin;
    if (opaque) {
        if (T.tri_level) {  // an opaque mesh: the 4-wide triangle-level hierarchy (an empty mesh has no nodes at all)
            if (T.nwide == 0) return none;
            return tree_intersect_wide<STATS>(sc.wnodes + T.wnode_begin, sc.otris + T.otri_begin, o, d, r32, bound, n_node, n_tri,
                                              aux.wstack);
        }
        if (cached) return tree_intersect<STATS, true>(aux.lnodes, tris, T.nnodes, o, d, r32, bound, n_node, n_tri);
        return tree_intersect<STATS, true>(nodes, tris, T.nnodes, o, d, r32, bound, n_node, n_tri);
    }
    const NodeRec *tb = sc.tboxes + T.tbox_begin;
    if (cached) return tree_intersect_lq<STATS>(aux.lnodes, tris, T.nnodes, o, d, r32, n_node, n_tri, tb);
    return tree_intersect_lq<STATS>(nodes, tris, T.nnodes, o, d, r32, n_node, n_tri, tb);
}

// objs[0 .. n_lds): the LDS-resident list; objects n_lds .. n_objs-1 (scenes with more than kLdsObjsMax objects) come from
// sc.objs, in the same order, so ties still go to the earlier object (main.cpp:57).
// SPILL: the kernel variants for such scenes (n_objs > n_lds); without it the list is the whole scene and none of the code for
// the others exists.
template <bool TREES, bool BEZ, bool SPH, bool STATS, bool SPILL = false, bool PRE = false, bool HFONLY = false>
__device__ __forceinline__ SceneHit intersect_scene(const ObjRec *__restrict__ objs, int n_lds, int n_objs, const DeviceScene &sc,
                                                    V3 o, V3 d, RayKey &rk, bool on, const LdsAux &aux,
                                                    uint32_t &n_node, uint32_t &n_tri) {
    SceneHit best;
    best.t = kInf;  // `nearest = INF`, main.cpp:54
    best.id = -1;
    best.n = mk(0, 0, 0);
    int nsrc = 0;  // 0: sphere (normal derived after the loop), 1: stored in best.n
    if (SPH) {
        // scenes made of spheres only: no kind dispatch, nothing but (t, id) carried round the loop
        for (int i = 0; i < n_lds; i++) {
            const double len = sphere_len(objs[i], o, d);
            if (len < best.t) {
                best.t = len;
                best.id = i;
            }
        }
        for (int i = n_lds; SPILL && i < n_objs; i++) {  // beyond the LDS list: one scalar load per object and wave (s_load_dwordx16)
            const ObjHead h = load_uniform(reinterpret_cast<const ObjHead *>(sc.objs + i));
            const double len = sphere_len(mk(h.a[0], h.a[1], h.a[2]), h.s0, o, d);
            if (len < best.t) {
                best.t = len;
                best.id = i;
            }
        }
        if (best.id >= 0) best.n = normalized((o + d * best.t) - load_centre<SPILL>(objs, n_lds, sc.objs, best.id));  // objects.h:65-66
        return best;
    }
    // 1/d for the box tests: three fp64 divisions (~100 instructions), paid only by waves that reach a tree
    V3 inv = mk(0, 0, 0);
    bool inv_ready = false;  // wave-uniform
    // The run of axis-aligned planes (plane_run above), before the loop.  When every lane of the wave is sure of its result the
    // loop passes over the run (and starts behind it if the run leads the list); otherwise the run's planes count for the unsure
    // lanes only.  Objects in FRONT of the run (planes with a bump tree: C5's floor) are then tested after the run's winner is
    // known, so a tie goes to them explicitly -- in the list's order they would have been there first (main.cpp:57) -- and
    // their bump walk starts with the nearest wall as its bound.
    // A run behind other planes (FRONT) is compiled into the HFONLY and the Bezier variants only -- the two launches of C5, whose
    // floor carries a bump tree (share at spp 256 219.7 -> 212.0 ms) --: in the other variants its two tests per object cost the
    // runs that lead the list 2 % (C4 full size 86.7 -> 88.6 ms), so there such a run is left to the loop.
    constexpr bool FRONT = HFONLY || BEZ;
    bool run_unsure = false, run_skip = false;
    int run_begin = 0, run_end = 0, i_first = 0;
    if (!SPH && sc.prun_end > 0 && (FRONT || sc.prun_begin == 0)) {  // (wave-uniform)
        run_begin = FRONT ? sc.prun_begin : 0;
        run_end = sc.prun_end;
        const PlaneRunHit ph = plane_run(objs, run_begin, run_end, o, d);
        run_unsure = ph.unsure;
        if (!ph.unsure && ph.id >= 0 && ph.len > 0) {  // best.t == kInf here: `len < nearest` is the loop's own test
            if (ph.len < best.t) {
                best.t = ph.len;
                best.id = ph.id;
                best.n = ld3(objs[ph.id].b);
                nsrc = 1;
            }
        }
        run_skip = __ballot(run_unsure) == 0ull;
        if (run_skip && run_begin == 0) i_first = run_end;
    }
    for (int i = i_first; i < (SPILL ? n_objs : n_lds); i++) {
        const ObjRec *obp = objs + i;
        if (SPILL && i >= n_lds) {
            // beyond the LDS list: the record is staged in this wave's LDS slot (eight 16-byte pieces) and read from there,
            // so that the body below exists once.  LDS operations of one wave execute in order; the fence keeps the
            // compiler from moving the reads of other lanes above the writes.
            if ((threadIdx.x & 63) < 8)
                reinterpret_cast<uint4 *>(aux.spill)[threadIdx.x & 63] = reinterpret_cast<const uint4 *>(sc.objs + i)[threadIdx.x & 63];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            obp = aux.spill;
        }
        const ObjRec &ob = *obp;
        if (FRONT && run_skip && i >= run_begin && i < run_end) continue;  // (uniform; only met when the run does not lead the list)
        // inside the plane run (some lane was unsure): only the unsure lanes take a plane's result from here
        const bool planes_one_by_one = !((!FRONT || i >= run_begin) && i < run_end) || run_unsure;
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
            const double len_plane = plane_len(ob, pn, o, d);
            double len = len_plane;
            const bool ph = len > 0;
            V3 nrm = pn;
            if (TREES) {
                const int tr = __builtin_amdgcn_readfirstlane(ob.tree);
                const bool want = on && ph;  // the bump tree is only consulted when the plane is hit (objects.h:508-513)
                if (tr >= 0 && __ballot(want) != 0ull) {
                    if (!inv_ready) {
                        inv = mk(1.0 / d.x, 1.0 / d.y, 1.0 / d.z);
                        inv_ready = true;
                    }
                    // a bump hit only counts if it is nearer than the plane itself (objects.h:514) and, to matter,
                    // nearer than the nearest object so far
                    const bool opaque = __builtin_amdgcn_readfirstlane((int)(ob.transp < kEps)) != 0;
                    const TreeHit h = tree_hit<STATS, HFONLY>(sc, aux, tr, opaque, fmin(len, best.t), want, o, d, inv, n_node, n_tri);
                    if (want && h.counter > 0 && h.len < len && h.len > 0) {
                        len = h.len;
                        nrm = tree_normal(sc.tris + load_uniform(&sc.trees[tr].tri_begin), h, d);
                    }
                }
            }
            // (a plane in front of the run meets the run's winner, not the other way round: it keeps a tie)
            if (planes_one_by_one && ph && (len < best.t || (FRONT && i < run_begin && len == best.t && best.id >= run_begin))) {
                best.t = len;
                best.id = i;
                best.n = nrm;
                nsrc = 1;
            }
        } else if (TREES && !HFONLY && kind == KIND_MESH) {
            // TriangleMesh::intersect, objects.h:405-455
            const int tr = __builtin_amdgcn_readfirstlane(ob.tree);
            // Early-out without a division: a ray whose LINE misses the sphere around the mesh's vertices (ObjRec.a, s0, with
            // 1e-3 to spare), or that points away from it from outside, cannot hit a triangle, whatever boxes it crosses: the
            // tree would return nothing that counts.  Most waves of a frame never come near the mesh and skip the
            // tree call, its three divisions and its root fetch altogether (measured: 7.7 of C4's 42.7 ms at spp 64).
            const bool may = on && mesh_may_hit(ob, o, d);
            // PRE: a lane whose hit in this object is already known (a fresh unit's primary ray, cgrt_primwalk.hpp) does not walk
            const bool known = PRE && rk.pre_valid && rk.pre_obj == i;
            const bool walk = may && !known;
            if (__ballot(PRE ? (walk || (may && known)) : may) != 0ull) {
                TreeHit h;
                h.len = kInf;
                h.tri = -1;
                h.counter = 0;
                if (__ballot(walk) != 0ull) {
                    if (!inv_ready) {
                        inv = mk(1.0 / d.x, 1.0 / d.y, 1.0 / d.z);
                        inv_ready = true;
                    }
                    const bool opaque = __builtin_amdgcn_readfirstlane((int)(ob.transp < kEps)) != 0;
                    h = tree_hit<STATS>(sc, aux, tr, opaque, best.t, walk, o, d, inv, n_node, n_tri);
                }
                if (PRE && known) {
                    h.len = rk.pre_len;
                    h.tri = rk.pre_tri;
                    h.counter = rk.pre_counter;
                }
                if (may && h.counter > 0 && h.len < best.t) {
                    V3 nrm = tree_normal(sc.tris + load_uniform(&sc.trees[tr].tri_begin), h, d);
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
            const int bi = __builtin_amdgcn_readfirstlane(ob.aux);
            const bool bh = bezier_wave(bz, ld3(ob.a), ob.b[0], on, o, d, key, n0, len, nrm, aux.bl,
                                        (rk.explicit_key || !sc.bez_slabs) ? nullptr : sc.bez_slabs + (size_t)bi * kBezSlabs);
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
        best.n = normalized(p - load_centre<SPILL>(objs, n_lds, sc.objs, best.id));
    }
    return best;
}

#endif
