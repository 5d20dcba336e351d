// The primary-ray mesh walk as a kernel of its own (round 3).  Part of libcgrt.so (cgrt_hip.hip).
//
// What was wrong.  Inside a heavy tile (cgrt_eye.hpp, unit-queue body) nearly every ray enters the mesh's hierarchy, but their
// walks are 3 to 60 node visits long, and a lane whose walk is over waits -- with its whole ray state, shading code and
// pending-ray machinery around it -- until the longest walk of the wave ends: 0.30 lanes active per VALU instruction on C4
// (profiles/r02_c4_dragon_pmc.json), > 90 % of that kernel's instructions being the walk.  Slicing the walk inside that kernel
// was measured and lost (the resumable state spills).
//
// What this kernel does.  Its WHOLE state is a walk: unit -> primary ray -> (len, triangle) of that ray's nearest hit in the
// one opaque mesh of the scene, written to prim_len / prim_tri at the unit's slot; nothing else.  A lane whose walk is over is
// refilled with the next unit of the same item queue the unit-queue body uses -- as soon as a quarter of the wave is idle, so
// that the ray set-up runs with many lanes --, and walks proceed in rounds of a few inner-node steps followed by the leaf
// tests of the lanes that stand on a leaf.  The unit-queue body then takes a fresh unit's primary mesh hit from the table
// instead of walking (cgrt_scene_walk.hpp, PRE): the same (len, triangle) by construction -- the walk below is
// tree_intersect_wide's arithmetic (slab32 box tests, the determinant triangle test, the (len, leaf, index) tie rule), and a
// nearest hit does not depend on the order in which leaves are visited nor on the pruning bound, as long as that bound is not
// below the hit finally accepted: here the walk starts unbounded (the scene walk compares the hit with the other objects'
// as always).
//
// Finishing units here (pw.finish, scenes whose planes carry no bump tree).  When a unit's walk is over, the rest of its primary
// ray's scene walk is a handful of sphere / plane tests; if the nearest object is DIFFUSE the ray tree ends there with one
// Hitpoint whose value is the surface colour (adj = (1,1,1), main.cpp:85-100) -- the unit is complete: its value is parked
// (dvals, dcnt = 1, or dcnt = 0 for a ray that hits nothing) and the unit-queue body skips it.  Only units whose nearest object
// reflects or refracts are left to that body (dcnt = 255 and the mesh hit in the table).  On C4 every heavy unit completes here,
// and the scheduled kernel -- which would generate the ray a second time, walk the planes, merge and park -- has nothing left
// to do for them.  The object loop is intersect_scene's: objects in `objs` order, strict <, the mesh at its own position.
#ifndef CGRT_PRIMWALK_HPP
#define CGRT_PRIMWALK_HPP
#include "cgrt_eye.hpp"

struct PrimWalkArgs {
    double *len;        // [heavy rank][sample][pixel]: distance of the primary ray's nearest hit in the mesh, kInf if none
    int32_t *tri;       //   ... its triangle (leaf-order index, TreeHit::tri), -1 if none
    int32_t obj, tree;  // the mesh: index in objs, tree index
    int32_t finish;     // complete units whose nearest object is diffuse here (see above); 0: fill the table only
    int32_t pad_;
    unsigned long long *counters;  // finish: rays / Hitpoints of the units completed here (CGRT_CNT_*)
};

template <bool DOF>
__global__ __launch_bounds__(kThreads, 4) void primary_walk_kernel(DeviceScene sc, GridParams g, PrimWalkArgs pw) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    ObjRec *lobjs = reinterpret_cast<ObjRec *>(lds_raw);
    const int n_stage = pw.finish ? sc.n_objs : pw.obj + 1;  // the mesh's record; all of them when units are finished here
    uint2 *lstack = reinterpret_cast<uint2 *>(lobjs + n_stage);  // [entry][thread], kWideLdsDepth entries; deeper ones in scratch
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(sc.objs);
        uint4 *dst = reinterpret_cast<uint4 *>(lobjs);
        const int n16 = n_stage * (int)(sizeof(ObjRec) / 16);
        for (int k = threadIdx.x; k < n16; k += kThreads) dst[k] = src[k];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, tid = threadIdx.x;
    const unsigned long long lanes_below = (1ull << lane) - 1ull;
    const int wtiles_x = (g.W + kWaveTileW - 1) / kWaveTileW;
    const V3 camorg = mk(g.cam[0], g.cam[1], g.cam[2]);
    const TreeRec T = load_uniform(sc.trees + pw.tree);
    const WideNodeRec *wn = sc.wnodes + T.wnode_begin;
    const OTriRec *otris = sc.otris + T.otri_begin;
    const ObjRec &mesh = lobjs[pw.obj];

    // the item queue (its own head, plan[5]; same items as the unit-queue body's)
    const unsigned n_items_total = load_uniform(g.plan) * (unsigned)g.items_per_tile;
    unsigned item_ahead = 0;
    if (lane == 0) item_ahead = atomicAdd(&g.plan[5], 1u);
    bool queue_empty = false;
    int unit_next = 0, unit_end = 0, hrank = 0;

    // the lane's walk
    bool active = false, pending = false;  // pending: the walk is over (or was not needed), the unit awaits its finish
    uint32_t my_rays = 0, my_hits = 0;
    int unit_rank_l = 0, unit_smp_l = 0, unit_pix_l = 0;
    size_t slot = 0;
    V3 o = camorg, d = mk(0, 0, 1);
    Ray32 r32 = make_ray32(o, mk(1, 1, 1), 0.f);
    double bound = kInf, best_len = kInf;
    float bound32 = 0.f;
    int best_tri = -1, best_leaf = -1;
    int32_t nxt = kWideNone;
    uint2 stk[kWideStack];
    int sp = 0;
    auto push = [&](uint2 e) {
        if (sp < kWideLdsDepth) lstack[sp * kThreads + tid] = e;
        else stk[sp] = e;
        sp++;
    };
    auto pop = [&]() -> int32_t {
        while (sp > 0) {
            --sp;
            const uint2 e = (sp < kWideLdsDepth) ? lstack[sp * kThreads + tid] : stk[sp];
            if (!(__uint_as_float(e.y) > bound32)) return (int32_t)e.x;
        }
        return kWideNone;
    };

    while (true) {
        // ---- (1) refill: once a quarter of the wave is not walking (or nothing is), the lanes whose walk is over finish their
        // unit (together), then every lane that is not walking draws a new unit ----
        const unsigned long long notwalking = __ballot(!active);
        const bool fresh_left = !(queue_empty && unit_next >= unit_end);
        const bool any_pending = __ballot(pending) != 0ull;
        if (notwalking != 0ull && (fresh_left || any_pending) && (__popcll(notwalking) >= g.pw_refill || notwalking == ~0ull)) {
            if (any_pending) {
                UTILP(19, pending);
                // the rest of the scene walk (main.cpp:55-63): every object in order, the mesh's hit at the mesh's position
                double bt = kInf;
                int bid = -1;
                // the leading run of axis-aligned planes as a group, as in intersect_scene (cgrt_scene_walk.hpp plane_run)
                int i_first = 0, run_end = 0;
                bool run_unsure = false;
                if (sc.prun_end > 0 && sc.prun_begin == 0 && pw.obj >= sc.prun_end) {
                    run_end = sc.prun_end;
                    const PlaneRunHit ph = plane_run(lobjs, 0, run_end, o, d);
                    run_unsure = ph.unsure;
                    if (!ph.unsure && ph.id >= 0 && ph.len > 0 && ph.len < bt) {
                        bt = ph.len;
                        bid = ph.id;
                    }
                    if (__ballot(run_unsure) == 0ull) i_first = run_end;
                }
                for (int i = i_first; i < sc.n_objs; i++) {
                    const ObjRec &ob = lobjs[i];
                    const int kind = __builtin_amdgcn_readfirstlane(ob.kind);
                    double len = kInf;
                    if (i == pw.obj) {
                        if (best_tri >= 0) len = best_len;
                    } else if (kind == KIND_SPHERE) {
                        len = sphere_len(ob, o, d);
                    } else if (kind == KIND_PLANE) {
                        const double l = plane_len(ob, ld3(ob.b), o, d);
                        if (l > 0) len = l;  // (no plane of such a scene carries a bump tree)
                    }
                    if ((!(i < run_end) || run_unsure) && len < bt) {
                        bt = len;
                        bid = i;
                    }
                }
                if (pending) {
                    const size_t us = ((size_t)unit_rank_l * g.spp + unit_smp_l) * 64 + unit_pix_l;
                    bool complete = true;
                    unsigned char cnt = 0;
                    if (bid >= 0) {
                        const ObjRec &ob = lobjs[bid];
                        if (ob.refl < kEps && ob.transp < kEps) {  // diffuse: Hitpoint{f * adj}, adj = (1,1,1)
                            V3 f = ld3(ob.col);
                            if (ob.kind == KIND_PLANE && ob.tex >= 0) {
                                V3 c;
                                if (texture_color(sc.texs[ob.tex], sc.texels, o + d * bt, c)) f = c;
                            }
                            const V3 hf = mulv(f, mk(1, 1, 1));
                            double *q = g.dvals + ((((size_t)unit_rank_l * g.spp + unit_smp_l) * g.maxhp + 0) * 64 + unit_pix_l) * 3;
                            q[0] = hf.x;
                            q[1] = hf.y;
                            q[2] = hf.z;
                            cnt = 1;
                            my_hits++;
                        } else {
                            complete = false;  // a mirror or glass object: the unit-queue body takes it from here
                        }
                    }
                    if (complete) {
                        g.dcnt[us] = cnt;
                        my_rays++;
                    } else {
                        g.dcnt[us] = 255;
                        pw.len[us] = best_len;
                        pw.tri[us] = best_tri;
                    }
                    pending = false;
                }
            }
            const unsigned long long idle = __ballot(!active);
            if (unit_next >= unit_end && !queue_empty) {
                const unsigned item = (unsigned)__builtin_amdgcn_readfirstlane((int)item_ahead);
                if (lane == 0) item_ahead = atomicAdd(&g.plan[5], 1u);
                if (item >= n_items_total) {
                    queue_empty = true;
                } else {
                    hrank = (int)(item / (unsigned)g.items_per_tile);
                    unit_next = (int)(item % (unsigned)g.items_per_tile) * g.units_per_item;
                    unit_end = unit_next + g.units_per_item;
                    if (unit_end > 64 * g.spp) unit_end = 64 * g.spp;
                }
            }
            const int u = unit_next + (int)__popcll(idle & lanes_below);
            unit_next += (int)__popcll(idle);
            bool fresh = !active && u < unit_end && !queue_empty;
            UTILP(20, fresh);
            if (fresh) {
                const int unit_pix = u & 63, unit_smp = u >> 6;
                const uint32_t wt = load_uniform(g.order + hrank);
                const int w = (int)(wt % (uint32_t)wtiles_x) * kWaveTileW + (unit_pix & 15);
                const int j = (int)(wt / (uint32_t)wtiles_x) * kWaveTileH + (unit_pix >> 4);
                const int h = global_row(g, j);
                slot = ((size_t)hrank * g.spp + unit_smp) * 64 + unit_pix;
                unit_rank_l = hrank;
                unit_smp_l = unit_smp;
                unit_pix_l = unit_pix;
                if (!((w < g.W) && (j < g.rows) && (h < g.H))) {
                    fresh = false;  // a pixel outside the image: the unit-queue body never opens it
                } else {
                    // the unit's primary ray: the expressions of trace_grid_body::start_sample (main.cpp:204-207)
                    const double *pc = g.pconst + (size_t)hrank * (7 * 64) + unit_pix;
                    const V3 pdir = mk(pc[0 * 64], pc[1 * 64], pc[2 * 64]);
                    const V3 pof = mk(pc[3 * 64], pc[4 * 64], pc[5 * 64]);
                    const uint64_t k_pix = (uint64_t)__double_as_longlong(pc[6 * 64]);
                    if (DOF) {
                        double sx, sy;
                        lens_disc(sample_key(k_pix, (uint64_t)(g.sample_offset + unit_smp)), sx, sy);
                        o = camorg + mk(sx, sy, 0) * g.lens_radius;
                        d = normalized(pof - o);
                    } else {
                        o = camorg;
                        d = pdir;
                    }
                }
            }
            if (fresh) {
                // the scene walk's early-out (cgrt_scene_walk.hpp, KIND_MESH): a ray that misses the sphere around the vertices
                const V3 lc = ld3(mesh.a) - o;
                const double tca = dot(lc, d), l2 = dot(lc, lc), dd2 = dot(d, d), r2 = mesh.s0;
                const bool may = !(tca < 0 && l2 > r2) && !(l2 * dd2 - tca * tca > r2 * dd2);
                if (may && T.nwide > 0) {
                    const V3 inv = mk(1.0 / d.x, 1.0 / d.y, 1.0 / d.z);
                    r32 = make_ray32(o, inv, T.bmax);
                    bound = kInf;  // (pruning with the nearest hit among the objects before the mesh was measured: the planes of a
                    bound32 = __double2float_ru(bound);  //  room lie behind the mesh for nearly every ray that enters it, 117.8 vs 115.0 ms)
                    best_len = kInf;
                    best_tri = -1;
                    best_leaf = -1;
                    sp = 0;
                    nxt = ~0;  // the root
                    active = true;
                } else if (pw.finish) {
                    best_len = kInf;
                    best_tri = -1;
                    pending = true;  // no walk needed: finished with the next batch
                } else {
                    pw.len[slot] = kInf;
                    pw.tri[slot] = -1;
                }
            }
        }
        if (__ballot(active) == 0ull) {
            if (queue_empty && unit_next >= unit_end && __ballot(pending) == 0ull) break;
            continue;
        }
        // ---- (2) a few rounds of inner-node steps: the lanes that stand on an inner node ----
        for (int round = 0; round < g.pw_rounds; round++) {
            const bool inner = active && nxt < 0 && nxt != kWideNone;
            if (__ballot(inner) == 0ull) break;
            if (inner) {
                UTIL(16);
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
        }
        // ---- (3) the lanes that stand on a leaf: its <= 4 triangles (tree_intersect_wide's test and tie rule) ----
        const bool leaf = active && nxt >= 0;
        if (__ballot(leaf) != 0ull) {
            if (leaf) {
                UTIL(17);
                const OTriRec *tp = otris + (nxt >> 4);
                const int cnt = nxt & 15;
                for (int k = 0; k < cnt; k++) {
                    UTIL(18);
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
                        if (len < best_len || (len == best_len && (rank.y > best_leaf || (rank.y == best_leaf && rank.x < best_tri)))) {
                            best_len = len;
                            best_tri = rank.x;
                            best_leaf = rank.y;
                        }
                    }
                }
                if (best_len < bound) {
                    bound = best_len;
                    bound32 = __double2float_ru(bound);
                }
                nxt = pop();
            }
        }
        // ---- (4) walks that are over ----
        if (active && nxt == kWideNone) {
            active = false;
            if (pw.finish) {
                pending = true;
            } else {
                pw.len[slot] = best_len;
                pw.tri[slot] = best_tri;
            }
        }
    }
    if (pw.finish && pw.counters) {  // the units completed here: one ray each, one Hitpoint where something diffuse was hit
        unsigned long long r = my_rays, hh = my_hits;
        for (int off = 32; off > 0; off >>= 1) {
            r += __shfl_xor(r, off);
            hh += __shfl_xor(hh, off);
        }
        if (lane == 0 && r != 0ull) {
            atomicAdd(&pw.counters[CGRT_CNT_RAYS], r);
            atomicAdd(&pw.counters[CGRT_CNT_HITPOINTS], hh);
        }
    }
}

#endif
