// The eye pass: trace_grid_kernel and the function-level probe kernel.  Part of libcgrt.so (cgrt_hip.hip).
#ifndef CGRT_EYE_HPP
#define CGRT_EYE_HPP
#include "cgrt_scene_walk.hpp"

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
#ifndef CGRT_BEZ_WAVES
#define CGRT_BEZ_WAVES 2
#endif
static constexpr int kBezWaves = CGRT_BEZ_WAVES;  // waves per SIMD the Bezier variants are compiled for
#ifndef CGRT_TREE_WAVES
#define CGRT_TREE_WAVES 3
#endif
#ifndef CGRT_SCHED_TREE_WAVES
#define CGRT_SCHED_TREE_WAVES CGRT_TREE_WAVES
#endif
static constexpr int kTreeWaves = CGRT_TREE_WAVES;             // ... the tree-capable tile kernels
static constexpr int kSchedTreeWaves = CGRT_SCHED_TREE_WAVES;  // ... the tree-capable scheduled kernels (unit queue + tile queue)

// uniform_sampling_circle (sampling.h:35-43) on a sample's lens stream (cgrt_rng.hpp): attempt j takes the two draws of the
// splitmix output z_j = fin64(key + (j + 1) G) -- what Stream::pair returns at position 2j, with the 64-bit multiply of the
// counter replaced by a running addition (the same integers mod 2^64).  Shared by every kernel that starts a primary ray.
__device__ __forceinline__ void lens_disc(uint64_t k_smp, double &sx, double &sy) {
    uint64_t ctr = k_smp;
    while (true) {
        ctr += kGolden;
        const uint64_t z = fin64(ctr);
        const double ux = div_rand_max((uint32_t)(z >> 33)), uy = div_rand_max((uint32_t)((z >> 2) & 0x7fffffffu));
        sx = ux * 2.0 - 1;
        sy = uy * 2.0 - 1;
        if (sx * sx + sy * sy < 1) break;
    }
}

struct HitpointSink {
    double *rec;                // cap x 10 doubles: f(3) pos(3) normal(3) label
    unsigned long long *count;  // appended so far (may exceed cap: then the tail was dropped)
    unsigned long long cap;
};

// Whether this wave of a tile workgroup has a wave tile to render in this launch -- the decisions trace_grid_body makes below,
// ahead of everything else: a workgroup none of whose waves has one (tiles of the other launch of a light / full pair, heavy
// tiles, tiles outside the image: three quarters of a probe launch's workgroups on C3) leaves before it stages the object list
// and the node cache in LDS.
template <int NT>
__device__ __forceinline__ bool wave_has_tile(const GridParams &g, int tile_block, int tile_grid) {
    using TG = TileGeom<NT>;
    const int wave = threadIdx.x >> 6;
    const int wtiles_x = (g.W + kWaveTileW - 1) / kWaveTileW, wtiles_y = (g.rows + kWaveTileH - 1) / kWaveTileH;
    int tile_x, tile_y;
    if (tile_grid < 0) {
        const int tiles_x = (g.W + TG::W - 1) / TG::W;
        tile_x = tile_block % tiles_x;
        tile_y = tile_block / tiles_x;
    } else {
        const int tile_blocks = tile_grid / g.chunks;
        if (!tile_of_block(g, tile_block % tile_blocks, tile_x, tile_y, TG::W, TG::H)) return false;
    }
    const int wx = tile_x * (TG::W / kWaveTileW) + (NT == 256 ? (wave & 1) : 0);
    const int wy = tile_y * (TG::H / kWaveTileH) + (NT == 256 ? (wave >> 1) : 0);
    if (!(wx < wtiles_x && wy < wtiles_y)) return false;
    if (g.hidx && g.hidx[wy * wtiles_x + wx] >= 0) return false;
    if (g.light && (g.light[wy * wtiles_x + wx] != 0) != (g.light_mode != 0)) return false;
    return true;
}

// The body of the eye pass for one workgroup.  HEAVY selects how the waves get their work (see GridParams): false -- each
// wave owns one wave tile (16x4 pixels, one per lane) and every lane runs its pixel's samples; true -- the waves serve the
// queue of heavy-tile items, lanes drawing (pixel, sample) units.  tile_block / tile_grid: this workgroup's index among the
// tile workgroups and their number (the launch may put heavy workgroups in front of them).
template <bool TREES, bool BEZ, bool DOF, bool GLASS, bool SPH, bool STATS, bool HPS, int NT, bool HEAVY, bool SPILL = false, bool HFONLY = false>
__device__ __forceinline__ void trace_grid_body(const DeviceScene &sc, const GridParams &g, float *__restrict__ rgb,
                                                uint32_t *__restrict__ nhit_out, unsigned long long *__restrict__ counters,
                                                const HitpointSink &hps, int tile_block, int tile_grid) {
    using TG = TileGeom<NT>;
    const long long tl_t0 = g.timeline ? wall_clock64() : 0;
    if (!HEAVY && !g.timeline) {  // (the timeline, a development aid, wants a record from every workgroup)
        if (!__syncthreads_or((int)wave_has_tile<NT>(g, tile_block, tile_grid))) return;
    }
    // LDS carve-up: [ pending-ray levels (GLASS) | objs | Bezier scratch | node cache ]
    extern __shared__ __align__(16) unsigned char lds_raw[];
    ObjRec *lobjs = reinterpret_cast<ObjRec *>(lds_raw + (GLASS ? TG::stack_bytes : 0));  // n_objs records
    // BEZ: one BezLds per wave behind the object list (16-byte aligned: ObjRec is 128 B)
    unsigned char *lrest = reinterpret_cast<unsigned char *>(lobjs + sc.n_lds);
    LdsAux aux;
    if (SPILL) {  // one staging record per wave for the objects beyond the LDS list
        aux.spill = reinterpret_cast<ObjRec *>(lrest) + (threadIdx.x >> 6);
        lrest += (NT / 64) * sizeof(ObjRec);
    }
    aux.bl = BEZ ? reinterpret_cast<volatile BezLds *>(lrest) + (threadIdx.x >> 6) : nullptr;
    if (BEZ) lrest += (NT / 64) * sizeof(BezLds);
    // TREES: node cache behind that (32-byte records, region is 16-byte aligned)
    NodeRec *lnodes = reinterpret_cast<NodeRec *>(lrest);
    aux.lnodes = (TREES && sc.cached_tree >= 0) ? lnodes : nullptr;
    // TREES without glass or Bezier (their LDS is spoken for): the first entries of the wide walk's stack, behind the node cache
    aux.wstack = (TREES && !GLASS && !BEZ && sc.has_wide)
                     ? reinterpret_cast<uint2 *>(lrest + ((sc.cached_tree >= 0 ? (size_t)sc.cached_nodes * sizeof(NodeRec) : 0)))
                     : nullptr;
    if (TREES && sc.cached_tree >= 0) {
        const uint4 *src = reinterpret_cast<const uint4 *>(sc.nodes + sc.trees[sc.cached_tree].node_begin);
        uint4 *dst = reinterpret_cast<uint4 *>(lnodes);
        const int n16 = sc.cached_nodes * (int)(sizeof(NodeRec) / 16);
        for (int k = threadIdx.x; k < n16; k += NT) dst[k] = src[k];
    }

    // stage the primitive list in LDS (128 B records, copied as 16-byte pieces)
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(sc.objs);
        uint4 *dst = reinterpret_cast<uint4 *>(lobjs);
        const int n16 = sc.n_lds * (int)(sizeof(ObjRec) / 16);
        for (int k = threadIdx.x; k < n16; k += NT) dst[k] = src[k];
    }
    __syncthreads();
    const long long cost_t0 = g.probe ? (long long)clock64() : 0;

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wtiles_x = (g.W + kWaveTileW - 1) / kWaveTileW, wtiles_y = (g.rows + kWaveTileH - 1) / kWaveTileH;
    const V3 camorg = mk(g.cam[0], g.cam[1], g.cam[2]);
    // HEAVY: this launch's waves serve the queue of heavy-tile items instead of owning a tile each (its own kernel variant:
    // the per-unit pixel state costs registers the tile variant does not have to spare).
    constexpr bool heavy = HEAVY;
    const unsigned long long lanes_below = (1ull << lane) - 1ull;

    // ---- the pixel a lane is working on: fixed for a tile's wave, set per unit for a heavy wave ----
    int wx = 0, wy = 0, chunk = 0, s_end = 0;
    bool have_tile = false;
    int w = 0, j = 0, h = 0;
    bool live = false;
    V3 pdir = mk(0, 0, 1), pof = mk(0, 0, 0);
    uint64_t k_pix = 0;
    auto set_pixel = [&](int lx, int ly) {  // lane-local pixel (lx, ly) of wave tile (wx, wy); main.cpp:188-189,198,203
        w = wx * kWaveTileW + lx;
        j = wy * kWaveTileH + ly;  // local row
        h = global_row(g, j);
        live = have_tile && (w < g.W) && (j < g.rows) && (h < g.H);
        const double px = (2.0 * ((double)w / g.W) - 1) * g.half_width;
        const double py = (2.0 * ((double)h / g.H) - 1) * g.half_width * g.H / g.W;
        pdir = normalized(mk(px, py, 0) - camorg);
        pof = pdir * ((g.focus_plane - camorg.z) / pdir.z) + camorg;
        k_pix = pixel_key(g.seed, (uint64_t)h * (uint64_t)g.W + (uint64_t)w);
    };
    if (!heavy) {
        // Natural order: a workgroup's waves are the 2x2 wave tiles of a 32x8 tile (tile_of_block: XCD-aware or row-major);
        // 8x8 per wave measured: meshes equal, C2 7 % slower.  When a tile's samples are split over several workgroups
        // (chunks > 1) the chunk index comes from the block index too.
        const int nb = tile_block;
        int tile_x, tile_y;
        if (tile_grid < 0) {  // an entry of the tile queue: the tile's number itself
            const int tiles_x = (g.W + TG::W - 1) / TG::W;
            tile_x = nb % tiles_x;
            tile_y = nb / tiles_x;
        } else {
            const int tile_blocks = tile_grid / g.chunks;
            chunk = nb / tile_blocks;
            if (!tile_of_block(g, nb % tile_blocks, tile_x, tile_y, TG::W, TG::H)) return;  // whole workgroup (no later barrier is missed)
        }
        wx = tile_x * (TG::W / kWaveTileW) + (NT == 256 ? (wave & 1) : 0);
        wy = tile_y * (TG::H / kWaveTileH) + (NT == 256 ? (wave >> 1) : 0);
        have_tile = wx < wtiles_x && wy < wtiles_y;
        // a heavy tile is rendered by the heavy workgroups: its wave stands down here; so does the wave of a tile that
        // belongs to the other launch of a light / full pair (the probe leaves the light tiles out: they are never heavy)
        if (have_tile && g.hidx && g.hidx[wy * wtiles_x + wx] >= 0) have_tile = false;
        if (have_tile && g.light && (g.light[wy * wtiles_x + wx] != 0) != (g.light_mode != 0)) have_tile = false;
        s_end = (g.chunks > 1) ? ((chunk + 1) * g.chunk_spp < g.spp ? (chunk + 1) * g.chunk_spp : g.spp) : g.spp;
        set_pixel(lane & 15, lane >> 4);
    }
    uint64_t k_smp = 0;  // key of the sample whose ray tree this lane is tracing

    double acc_r = 0, acc_g = 0, acc_b = 0;
    uint32_t my_hits = 0, my_rays = 0, my_nodes = 0, my_tris = 0, wave_iters = 0;
    uint32_t hp_seq = 0;  // index of the next Hitpoint within the current sample's ray tree (emission order)

    Pending deep[2];   // third stack level (scratch; indexed dynamically so that it stays out of registers)
    Pending sib;       // refracted sibling of a leaf-level glass hit (registers)
    bool sib_valid = false;
    unsigned char *lslot = lds_raw;  // level L, field f of this thread: lslot + L*TG::level_bytes + (f*NT + tid)*8
    int sp = 0;
    int s = (g.chunks > 1) ? chunk * g.chunk_spp : 0;  // next sample to start
    bool have = false;
    V3 o = camorg, d = pdir, adj = mk(1, 1, 1);
    int depth_left = 0;
    uint32_t path = 1;
    // heavy waves: the item units are being drawn from (wave-uniform) and the lane's own unit
    int unit_next = 0, unit_end = 0, hrank = 0;
    int unit_pix = 0, unit_smp = 0, unit_rank = 0;
    bool unit_open = false;
    // PRE: the primary ray of a fresh unit finds its mesh hit in the table primary_walk_kernel filled (cgrt_primwalk.hpp)
    constexpr bool PRE = HEAVY && TREES && !BEZ && !STATS && !SPILL;
    bool pre_valid = false;
    double pre_len = 0;
    int32_t pre_tri = -1;

    auto start_sample = [&](int smp) {  // main.cpp:204-209
        k_smp = sample_key(k_pix, (uint64_t)(g.sample_offset + smp));
        if (DOF) {
            double sx, sy;
            lens_disc(k_smp, sx, sy);  // purpose 0: the lens stream's key is the sample key
            o = camorg + mk(sx, sy, 0) * g.lens_radius;
            d = normalized(pof - o);
        } else {
            o = camorg;
            d = pdir;
        }
        adj = mk(1, 1, 1);
        depth_left = g.max_depth;
        path = 1;
        hp_seq = 0;
        have = true;
    };

    // heavy waves: `hrank`, wx, wy and [unit_next, unit_end) describe the item the wave is currently drawing units from
    // (wave-uniform); a lane remembers the item of the unit it is tracing in unit_rank (the wave may have moved on)
    bool queue_empty = false;
    unsigned item_ahead = 0, n_items_total = 0;
    if (heavy) {
        n_items_total = load_uniform(g.plan) * (unsigned)g.items_per_tile;
        if (lane == 0) item_ahead = atomicAdd(&g.plan[2], 1u);
    }
    while (true) {
        if (heavy) {
            // A lane without a ray closes its unit (how many Hitpoints it produced) and takes the next one: unit u of an item
            // is sample u / 64 of pixel u % 64 of the item's tile, so the lanes of a wave keep working on neighbouring pixels
            // of one sample.  When the item runs dry the wave takes the next item from the queue at once -- lanes still
            // tracing units of the old item carry on -- so lanes idle only when the whole queue is empty.
            const bool want = !have;
            const unsigned long long m = __ballot(want);
            if (m != 0ull) {
                if (want && unit_open) {
                    g.dcnt[((size_t)unit_rank * g.spp + unit_smp) * 64 + unit_pix] = (unsigned char)hp_seq;
                    unit_open = false;
                }
                if (unit_next >= unit_end && !queue_empty) {
                    // the item fetched ahead (its atomic has been in flight while the previous item was traced) ...
                    const unsigned item = (unsigned)__builtin_amdgcn_readfirstlane((int)item_ahead);
                    // ... and the one after it is requested now
                    if (lane == 0) item_ahead = atomicAdd(&g.plan[2], 1u);
                    if (item >= n_items_total) {
                        queue_empty = true;
                    } else {
                        hrank = (int)(item / (unsigned)g.items_per_tile);
                        const uint32_t wt = load_uniform(g.order + hrank);
                        wx = (int)(wt % (uint32_t)wtiles_x);
                        wy = (int)(wt / (uint32_t)wtiles_x);
                        have_tile = true;
                        unit_next = (int)(item % (unsigned)g.items_per_tile) * g.units_per_item;
                        unit_end = unit_next + g.units_per_item;
                        if (unit_end > 64 * g.spp) unit_end = 64 * g.spp;
                    }
                }
                const int u = unit_next + (int)__popcll(m & lanes_below);
                unit_next += (int)__popcll(m);  // lanes that drew beyond unit_end draw again from the next item
                if (want && u < unit_end) {
                    unit_pix = u & 63;
                    unit_smp = u >> 6;
                    unit_rank = hrank;
                    // the pixel's camera constants come from the table pixel_const_kernel filled (five fp64 divisions, a square
                    // root and two hash rounds per pixel: a third of a sphere-scene sample if redone for every unit)
                    w = wx * kWaveTileW + (unit_pix & 15);
                    j = wy * kWaveTileH + (unit_pix >> 4);
                    h = global_row(g, j);
                    live = (w < g.W) && (j < g.rows) && (h < g.H);
                    // PRE: a unit primary_walk_kernel has already completed (its nearest object was diffuse) is not opened
                    if (PRE && live && g.prim_len && g.prim_done &&
                        g.dcnt[((size_t)unit_rank * g.spp + unit_smp) * 64 + unit_pix] != 255)
                        live = false;
                    if (live) {
                        const double *pc = g.pconst + (size_t)unit_rank * (7 * 64) + unit_pix;
                        pdir = mk(pc[0 * 64], pc[1 * 64], pc[2 * 64]);
                        pof = mk(pc[3 * 64], pc[4 * 64], pc[5 * 64]);
                        k_pix = (uint64_t)__double_as_longlong(pc[6 * 64]);
                        start_sample(unit_smp);
                        unit_open = true;
                        if (PRE && g.prim_len) {
                            const size_t us = ((size_t)unit_rank * g.spp + unit_smp) * 64 + unit_pix;
                            pre_len = g.prim_len[us];
                            pre_tri = g.prim_tri[us];
                            pre_valid = true;
                        }
                    }
                }
            }
            if (__ballot(have) == 0ull) {
                if (queue_empty && unit_next >= unit_end) break;  // nothing left anywhere
                continue;                                         // drew nothing traceable (item boundary, pixels outside the image)
            }
        } else {
            if (!have && live && s < s_end) {
                // start the next sample of this lane's pixel
                start_sample(s);
                s++;
            }
            if (__ballot(have) == 0ull) break;  // every lane of the wave has drained its pixel
        }
        wave_iters++;
        UTILP(heavy ? 0 : 11, have);
        // All 64 lanes enter the scene walk together (lanes without a ray carry on == false): the object list
        // is wave-uniform, so its control flow stays scalar.
        RayKey rk{k_smp, path, false, 0u};
        if (PRE) {
            rk.pre_valid = have && pre_valid;  // only the ray a unit starts with
            rk.pre_obj = g.prim_obj;
            rk.pre_len = pre_len;
            rk.pre_tri = pre_tri;
            rk.pre_counter = pre_tri >= 0 ? 1 : 0;
            pre_valid = false;
        }
        const SceneHit hit =
            intersect_scene<TREES, BEZ, SPH, STATS, SPILL, PRE, HFONLY>(lobjs, sc.n_lds, sc.n_objs, sc, o, d, rk, have, aux, my_nodes, my_tris);
        if (have) {
            my_rays++;
            have = false;
            if (hit.id >= 0) {
                const ObjMat ob = load_mat<SPILL>(lobjs, sc.n_lds, sc.objs, hit.id);
                const V3 P = o + d * hit.t;  // main.cpp:68
                V3 n = hit.n;
                const V3 n_old = n;
                bool into = true;
                if (dot(n, d) > 0) {  // main.cpp:73-76
                    n = -n;
                    into = false;
                }
                V3 f = ob.col;  // getSurfaceColor
                if (ob.kind == KIND_PLANE && ob.tex >= 0) {
                    V3 c;
                    if (texture_color(sc.texs[ob.tex], sc.texels, P, c)) f = c;  // objects.h:533-539
                }
                const double refl = ob.refl, transp = ob.transp;
                if (refl < kEps && transp < kEps) {
                    UTIL(heavy ? 7 : 12);
                    // diffuse: the reference stores Hitpoint{f*adj,...} (main.cpp:85-100); we accumulate it
                    const V3 hf = mulv(f, adj);
                    if (heavy) {
                        // deferred: the value is added in deferred_sum_kernel, in this (sample, emission) position
                        double *q = g.dvals + ((((size_t)unit_rank * g.spp + unit_smp) * g.maxhp + hp_seq) * 64 + unit_pix) * 3;
                        q[0] = hf.x;
                        q[1] = hf.y;
                        q[2] = hf.z;
                    } else {
                        acc_r += hf.x;
                        acc_g += hf.y;
                        acc_b += hf.z;
                    }
                    my_hits++;
                    if (HPS) {
                        // one atomic per wave, not per lane (same-address atomics are served one after the other): the lanes
                        // that are here take consecutive places behind the count their first lane fetched
                        const unsigned long long here = __ballot(true);
                        unsigned long long base = 0ull;
                        if (lane == (int)__ffsll((long long)here) - 1) base = atomicAdd(hps.count, (unsigned long long)__popcll(here));
                        base = (unsigned long long)__shfl((long long)base, (int)__ffsll((long long)here) - 1);
                        const unsigned long long k = base + (unsigned long long)__popcll(here & lanes_below);
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
                    }
                    if (HPS || heavy) hp_seq++;
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
                        UTIL(heavy ? 8 : 13);
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
                                normalized(d * nnt - n_old * ((into ? 1 : -1) * (ddn * nnt + sqrt_cr(cos2t))));
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
                                    double *q = reinterpret_cast<double *>(lslot + sp * TG::level_bytes) + threadIdx.x;
                                    q[0 * NT] = pe.o.x; q[1 * NT] = pe.o.y; q[2 * NT] = pe.o.z;
                                    q[3 * NT] = pe.d.x; q[4 * NT] = pe.d.y; q[5 * NT] = pe.d.z;
                                    q[6 * NT] = pe.adj.x; q[7 * NT] = pe.adj.y; q[8 * NT] = pe.adj.z;
                                    // depth_left <= 4 and path < 32: one word
                                    reinterpret_cast<uint32_t *>(lslot + sp * TG::level_bytes +
                                                                 kPendDoubles * NT * sizeof(double))[threadIdx.x] =
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
                UTIL(heavy ? 10 : 14);
                --sp;
                if (sp < kLdsLevels) {
                    const double *q = reinterpret_cast<const double *>(lslot + sp * TG::level_bytes) + threadIdx.x;
                    o = mk(q[0 * NT], q[1 * NT], q[2 * NT]);
                    d = mk(q[3 * NT], q[4 * NT], q[5 * NT]);
                    adj = mk(q[6 * NT], q[7 * NT], q[8 * NT]);
                    const uint32_t meta = reinterpret_cast<const uint32_t *>(
                        lslot + sp * TG::level_bytes + kPendDoubles * NT * sizeof(double))[threadIdx.x];
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
    }  // ray loop

    if (g.probe) {
        // cost probe: nothing of the image is stored; the wave leaves what its tile cost and the tile's number
        if (lane == 0 && have_tile) {
            const long long dt = (long long)clock64() - cost_t0;
            const uint32_t wt = (uint32_t)(wy * wtiles_x + wx);
            g.cost[wt] = dt > 0 ? (dt < 0xffffffffll ? (uint32_t)dt : 0xffffffffu) : 1u;
        }
        return;  // no counters: the render launch that follows counts these rays
    }
    if (heavy) {
        // nothing to store: deferred_sum_kernel writes the heavy tiles' pixels
    } else if (g.chunks > 1) {
        // split samples: this wave's raw fp64 sums; finalize_chunks_kernel adds the chunks in order
        if (have_tile && (w < g.W) && (j < g.rows)) {  // rows past the image (stripe padding) carry zero sums
            const size_t px = ((size_t)chunk * g.rows + (size_t)j) * g.W + (size_t)w;
            g.partial[3 * px + 0] = acc_r;
            g.partial[3 * px + 1] = acc_g;
            g.partial[3 * px + 2] = acc_b;
            if (g.partial_nhit) g.partial_nhit[px] = my_hits;
        }
    } else {
        // ---- store: the wave's 16x4 pixels are 4 rows of 48 contiguous floats; lane l of pass k writes float k*64 + l of
        // the tile, fetched from the lane that owns that pixel (three ds_bpermute shuffles per pass, no LDS, no barrier).
        // A store instruction then covers 256 contiguous bytes of a row (192 + 64 of the next).
        const float v0 = (float)(acc_r * g.inv_spp_total), v1 = (float)(acc_g * g.inv_spp_total),
                    v2 = (float)(acc_b * g.inv_spp_total);
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int idx = k * 64 + lane;
            const int row = idx / (kWaveTileW * 3), c = idx % (kWaveTileW * 3);
            const int src = row * kWaveTileW + c / 3, ch = c % 3;
            const float a0 = __shfl(v0, src), a1 = __shfl(v1, src), a2 = __shfl(v2, src);
            const float v = ch == 0 ? a0 : (ch == 1 ? a1 : a2);
            const int jj = wy * kWaveTileH + row, ww = wx * kWaveTileW + c / 3;
            if (have_tile && jj < g.rows && ww < g.W) {
                float *dst = rgb + ((size_t)jj * g.W + ww) * 3 + ch;
                *dst = g.accumulate ? *dst + v : v;  // progressive passes add into the fp32 frame
            }
        }
        if (nhit_out && have_tile && (w < g.W) && (j < g.rows)) nhit_out[(size_t)j * g.W + w] = my_hits;
    }

    if (g.timeline) {  // development aid: when and where this workgroup ran
        __shared__ unsigned int tl_rays;
        if (threadIdx.x == 0) tl_rays = 0;
        __syncthreads();
        atomicAdd(&tl_rays, my_rays);
        __syncthreads();
        if (threadIdx.x == 0) {  // a workgroup that serves queues passes here once per tile / per queue: first start, last end, all rays
            unsigned long long *q = g.timeline + 4 * (size_t)blockIdx.x;
            const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);
            if (q[0] == 0ull) q[0] = (unsigned long long)tl_t0;
            q[1] = (unsigned long long)wall_clock64();
            q[2] = (unsigned long long)hw | ((unsigned long long)xcc << 32);
            q[3] = ((unsigned long long)(wx & 0xffff) | ((unsigned long long)(wy & 0xffff) << 16)) + ((q[3] >> 32) << 32) + ((unsigned long long)tl_rays << 32);
        }
    }
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
        // (`counters` is the workgroup's LDS array, see wg_counters_begin) a wave that traced nothing adds nothing
        if (lane == 0 && (r | (unsigned long long)wave_iters) != 0ull) {
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

// Counters.  A wave adds its sums to counters that all waves of the launch share; same-address atomics are served one after the
// other, ~12 ns each, by the memory side -- measured: a 4096x4096 frame of plane-only rays took 9.5 ms whatever its sample count
// (1, 4 or 16), 262 144 waves x 3 atomics.  So a workgroup collects its waves' sums in LDS (wg_counters_begin/end around the
// body or bodies it runs: the body's `counters` IS that LDS array) and adds them to the launch's counters once, when it ends:
// a quarter of the atomics for one-tile workgroups, a few thousand per launch for the persistent ones.
__device__ __forceinline__ unsigned long long *wg_counters_begin(unsigned long long *wg, const unsigned long long *counters) {
    if (!counters) return nullptr;  // (a kernel argument: uniform)
    if (threadIdx.x < CGRT_NCOUNTERS) wg[threadIdx.x] = 0ull;
    __syncthreads();
    return wg;
}
__device__ __forceinline__ void wg_counters_end(const unsigned long long *wg, unsigned long long *counters) {
    if (!counters) return;
    __syncthreads();  // every wave of the workgroup has left its last body
    if (threadIdx.x < CGRT_NCOUNTERS && wg[threadIdx.x] != 0ull) atomicAdd(&counters[threadIdx.x], wg[threadIdx.x]);
}

// One launch = tile workgroups only (probe, image order, Hitpoint capture) ...
template <bool TREES, bool BEZ, bool DOF, bool GLASS, bool SPH, bool STATS, bool HPS = false, int NT = 256, bool SPILL = false, bool HFONLY = false>
__global__ __launch_bounds__(NT, BEZ ? kBezWaves : ((TREES && !HFONLY) ? kTreeWaves : 4)) void trace_grid_kernel(DeviceScene sc, GridParams g, float *__restrict__ rgb,
                                                             uint32_t *__restrict__ nhit_out,
                                                             unsigned long long *__restrict__ counters,
                                                             HitpointSink hps = HitpointSink{nullptr, nullptr, 0}) {
    __shared__ unsigned long long wg_cnt[CGRT_NCOUNTERS];
    unsigned long long *wc = wg_counters_begin(wg_cnt, counters);
    trace_grid_body<TREES, BEZ, DOF, GLASS, SPH, STATS, HPS, NT, false, SPILL, HFONLY>(sc, g, rgb, nhit_out, wc, hps, (int)blockIdx.x,
                                                                                       (int)gridDim.x);
    wg_counters_end(wg_cnt, counters);
}
// ... or the scheduled form: the first g.heavy_blocks workgroups serve the heavy tiles' unit queue, the others are the tile
// workgroups.  Two bodies in one kernel: the dispatcher starts workgroups in index order, so the heavy work starts first and
// tile workgroups take over the slots as the heavy waves retire -- no seam between two launches.  Each body keeps its own
// register allocation (the paths are disjoint); the kernel's register and scratch sizes are the larger of the two.
template <bool TREES, bool BEZ, bool DOF, bool GLASS, bool SPH, bool STATS, int NT = 256>
__global__ __launch_bounds__(NT, BEZ ? kBezWaves : (TREES ? kSchedTreeWaves : 4)) void trace_grid_sched_kernel(DeviceScene sc, GridParams g,
                                                                                         float *__restrict__ rgb,
                                                                                         uint32_t *__restrict__ nhit_out,
                                                                                         unsigned long long *__restrict__ counters) {
    const HitpointSink none{nullptr, nullptr, 0};
    const bool heavy_first = (int)blockIdx.x < g.heavy_blocks;
    __shared__ unsigned long long wg_cnt[CGRT_NCOUNTERS];
    unsigned long long *wc = wg_counters_begin(wg_cnt, counters);
    if (NT == 64) {
        // one-wave workgroups (Bezier scenes): one workgroup per 16x4 tile behind the heavy ones -- a slot is handed on the
        // moment its wave ends, and measured on a C5 band the queue form bought nothing (34.6 vs 34.9 ms) while its larger
        // kernel cost 4 %
        if (heavy_first)
            trace_grid_body<TREES, BEZ, DOF, GLASS, SPH, STATS, false, NT, true>(sc, g, rgb, nhit_out, wc, none, 0, 1);
        else
            trace_grid_body<TREES, BEZ, DOF, GLASS, SPH, STATS, false, NT, false>(sc, g, rgb, nhit_out, wc, none,
                                                                                  (int)blockIdx.x - g.heavy_blocks,
                                                                                  (int)gridDim.x - g.heavy_blocks);
        wg_counters_end(wg_cnt, counters);
        return;
    }
    // With a tile queue (g.border) both kinds of work come from queues: a workgroup serves its own kind until that queue is
    // empty, then the other.  Without one (split samples) a workgroup behind the heavy ones renders the one tile its index names.
    // Each body appears once in the code: the loop runs them in either order.
    const bool queued = g.border != nullptr;
    __shared__ unsigned tile_entry;
    for (int phase = 0; phase < 2; phase++) {
        if ((phase == 0) == heavy_first) {
            if (queued || heavy_first)
                trace_grid_body<TREES, BEZ, DOF, GLASS, SPH, STATS, false, NT, true>(sc, g, rgb, nhit_out, wc, none, 0, 1);
        } else {
            const unsigned n_entries = queued ? load_uniform(g.plan + 3) : (heavy_first ? 0u : 1u);
            unsigned served = 0;
            while (true) {
                unsigned e = served++;
                if (queued) {
                    __syncthreads();  // every wave is done with the previous tile (and has read tile_entry)
                    if (threadIdx.x == 0) tile_entry = atomicAdd(&g.plan[4], 1u);
                    __syncthreads();
                    e = tile_entry;
                }
                if (e >= n_entries) break;
                trace_grid_body<TREES, BEZ, DOF, GLASS, SPH, STATS, false, NT, false>(
                    sc, g, rgb, nhit_out, wc, none, queued ? (int)g.border[e] : (int)blockIdx.x - g.heavy_blocks,
                    queued ? -1 : (int)gridDim.x - g.heavy_blocks);
            }
        }
    }
    wg_counters_end(wg_cnt, counters);
}

// CGRT_GRID_SPLIT_SAMPLES, second step: chunk sums added in chunk order, scaled, rounded once to fp32.
__global__ void finalize_chunks_kernel(GridParams g, float *__restrict__ rgb, uint32_t *__restrict__ nhit_out) {
    const size_t npx = (size_t)g.rows * g.W;
    const size_t px = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (px >= npx) return;
    if (g.hidx) {  // pixels of heavy tiles are written by deferred_sum_kernel (their chunk sums do not exist)
        const int wtiles_x = (g.W + kWaveTileW - 1) / kWaveTileW;
        if (g.hidx[(int)((px / g.W) / kWaveTileH) * wtiles_x + (int)((px % g.W) / kWaveTileW)] >= 0) return;
    }
    double r = 0, gg = 0, b = 0;
    uint32_t hits = 0;
    for (int c = 0; c < g.chunks; c++) {
        const size_t q = (size_t)c * npx + px;
        r += g.partial[3 * q + 0];
        gg += g.partial[3 * q + 1];
        b += g.partial[3 * q + 2];
        if (g.partial_nhit) hits += g.partial_nhit[q];
    }
    const float fr = (float)(r * g.inv_spp_total), fg = (float)(gg * g.inv_spp_total), fb = (float)(b * g.inv_spp_total);
    float *dst = rgb + 3 * px;
    if (g.accumulate) {
        dst[0] += fr; dst[1] += fg; dst[2] += fb;
    } else {
        dst[0] = fr; dst[1] = fg; dst[2] = fb;
    }
    if (nhit_out) nhit_out[px] = hits;
}

// ---- light tiles -------------------------------------------------------------------------------------------------------
// One thread per wave tile (16x4 pixels).  The tile is LIGHT when no primary ray of it -- any pixel, any lens sample -- can
// touch the bounding sphere of a "special" object: a mesh or Bezier object (trees, Newton) or a sphere that reflects or
// refracts (secondary rays go anywhere).  With every plane diffuse and un-bumped (DeviceScene::light_ok) such a tile's rays
// end on a diffuse sphere or plane after one scene walk, whatever the sample: the variant without tree / Bezier / pending-ray
// code renders it exactly.  Conservative bound: the tile's pinhole directions lie in a cone around its centre direction
// (half-angle = 1.5 x the largest corner deviation + 1e-6); a thin-lens ray deviates from its pinhole ray, at depth z, by
// lens_radius * |1 - (z - cam.z) / (focus_plane - cam.z)| sideways, so the object's sphere is grown by the largest such
// deviation over its depth range.  A mesh is bounded by its cover spheres (DeviceScene::cover: up to 64 spheres over
// median-split groups of its triangles -- a long thin mesh fills little of one sphere around all of it).  Anything doubtful
// (object behind or around the camera, rows beyond the image) is FULL.
__global__ void classify_kernel(DeviceScene sc, GridParams g, unsigned char *__restrict__ light, int n_wt) {
    const int wt = blockIdx.x * blockDim.x + threadIdx.x;
    if (wt >= n_wt) return;
    const int wtiles_x = (g.W + kWaveTileW - 1) / kWaveTileW;
    const int wx = wt % wtiles_x, wy = wt / wtiles_x;
    const V3 cam = mk(g.cam[0], g.cam[1], g.cam[2]);
    auto dir_of = [&](int w, int j) {
        const int h = global_row(g, j);
        const double px = (2.0 * ((double)w / g.W) - 1) * g.half_width;
        const double py = (2.0 * ((double)h / g.H) - 1) * g.half_width * g.H / g.W;
        return normalized(mk(px, py, 0) - cam);
    };
    const int w0 = wx * kWaveTileW, j0 = wy * kWaveTileH;
    // corners one pixel beyond the tile on every side (pixels are sampled at their lower-left corner; the margin also covers
    // the curvature of the angle function along the edges)
    const V3 c00 = dir_of(w0 - 1, j0 - 1), c10 = dir_of(w0 + kWaveTileW, j0 - 1), c01 = dir_of(w0 - 1, j0 + kWaveTileH),
             c11 = dir_of(w0 + kWaveTileW, j0 + kWaveTileH);
    const V3 dc = normalized((c00 + c10) + (c01 + c11));
    double cmin = fmin(fmin(dot(dc, c00), dot(dc, c10)), fmin(dot(dc, c01), dot(dc, c11)));
    cmin = fmin(1.0, fmax(-1.0, cmin));
    const double alpha = 1.5 * acos(cmin) + 1e-6;
    bool is_light = true;
    // the stripe mapping keeps a wave tile's four rows adjacent (stripes are multiples of 8 rows), so the corners bound it
    // false: the sphere (c, r) may be touched by a primary ray of the tile
    auto clear_of = [&](V3 c, double r) {
        r = r * (1 + 1e-9) + 1e-6;
        if (g.lens_radius > 0) {
            const double f = g.focus_plane - cam.z;
            const double s_lo = (c.z - r - cam.z) / f, s_hi = (c.z + r - cam.z) / f;
            if (!(f > 0) || !(s_lo > 0)) return false;  // object reaches the lens plane or behind it
            r += g.lens_radius * fmax(fabs(1 - s_lo), fabs(1 - s_hi));
        }
        const V3 v = c - cam;
        const double dist = sqrt(dot(v, v));
        if (!(dist > r)) return false;
        double ct = dot(dc, v) / dist;
        ct = fmin(1.0, fmax(-1.0, ct));
        return !(acos(ct) <= alpha + asin(r / dist) + 1e-6);
    };
    for (int i = 0; i < sc.n_objs && is_light; i++) {
        const ObjRec &ob = sc.objs[i];
        if (ob.kind == KIND_SPHERE) {
            if (ob.refl < kEps && ob.transp < kEps) continue;  // diffuse: no secondary rays
            is_light = clear_of(ld3(ob.a), sqrt(ob.s0));
        } else if (ob.kind == KIND_BEZIER) {
            const BezierRec &bz = sc.beziers[ob.aux];
            const V3 hw = mk(0.5 * (bz.box[1] - bz.box[0]), 0.5 * (bz.box[3] - bz.box[2]), 0.5 * (bz.box[5] - bz.box[4]));
            is_light = clear_of(mk(0.5 * (bz.box[0] + bz.box[1]), 0.5 * (bz.box[2] + bz.box[3]), 0.5 * (bz.box[4] + bz.box[5])),
                                sqrt(dot(hw, hw)) + 1e-3);
        }
        // planes: vetted by light_ok; meshes: the cover spheres below
    }
    for (int i = 0; i < sc.n_cover && is_light; i++)
        is_light = clear_of(mk(sc.cover[4 * i], sc.cover[4 * i + 1], sc.cover[4 * i + 2]), sc.cover[4 * i + 3]);
    light[wt] = is_light ? 1 : 0;
}

// ---- cost-aware scheduling: plan and ordered sum (GridParams, "Cost-aware scheduling") ------------------------------------
// One workgroup.  cost[]: the wave tiles' probe costs.  A wave tile is HEAVY when its cost exceeds total / divisor --
// divisor = wave slots of the chip x a constant, i.e. when the tile alone would occupy a wave slot for more than 1/constant
// of the frame's ideal duration.  At most kmax tiles can be heavy (what the deferred buffers hold): if more exceed the
// threshold it is raised, on a quarter-octave histogram of the costs, until the count fits -- the heaviest stay.
// Output: order[0..K) = the heavy wave tiles, heaviest histogram bin first, hidx[wave tile] = rank or -1, plan[0] = K,
// plan[1] = threshold, plan[2] = 0 (the item queue's head); with `border`: the tile queue (GridParams::border), plan[3] = its
// length, plan[4] = 0 (its head).
__global__ __launch_bounds__(1024) void plan_kernel(const uint32_t *__restrict__ cost_in, const unsigned char *__restrict__ light, int n,
                                                    unsigned kmax, unsigned long long divisor, uint32_t *__restrict__ plan,
                                                    uint32_t *__restrict__ order, int32_t *__restrict__ hidx,
                                                    uint32_t *__restrict__ border, int wtiles_x, int wtiles_y, int tw, int th) {
    // a light tile (rendered by the other launch) counts for nothing here
    auto cost = [&](int i) -> uint32_t { return (light && light[i]) ? 0u : cost_in[i]; };
    __shared__ unsigned long long part[1024];
    __shared__ unsigned base_s, hist[128];
    __shared__ unsigned long long thr_s;
    auto bin_of = [](uint32_t c) { return c < 4u ? (int)c : (31 - __clz((int)c)) * 4 + (int)((c >> (29 - __clz((int)c))) & 3u); };
    if (threadIdx.x < 128) hist[threadIdx.x] = 0;
    __syncthreads();
    unsigned long long t = 0;
    for (int i = threadIdx.x; i < n; i += 1024) {
        const uint32_t c = cost(i);
        t += c;
        atomicAdd(&hist[bin_of(c)], 1u);
    }
    part[threadIdx.x] = t;
    __syncthreads();
    for (int off = 512; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) part[threadIdx.x] += part[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        unsigned long long thr = part[0] / divisor;
        if (thr < 1) thr = 1;
        // raise the threshold to the lower edge of the lowest histogram bin such that all bins above it hold <= kmax tiles
        unsigned above = 0;
        int b = 127;
        for (; b >= 0; b--) {
            if (above + hist[b] > kmax) break;
            above += hist[b];
        }
        if (b >= 0) {  // bins 0..b must stay out: threshold = the largest cost of bin b
            const unsigned long long edge = b < 4 ? (unsigned long long)b
                                                  : (((4ull + (unsigned long long)(b & 3) + 1ull) << (b / 4)) >> 2) - 1ull;
            if (edge > thr) thr = edge;
        }
        thr_s = thr;
        base_s = 0;
    }
    __syncthreads();
    const unsigned long long thr = thr_s;
    // Counting sort of the heavy tiles by histogram bin, heaviest bin first (longest-processing-time-first at quarter-octave
    // resolution; the order inside a bin is whatever the atomics give -- it affects neither the image nor the counters).
    // hist[] is reused: bins wholly above the threshold get their start offset, the others nothing.
    if (threadIdx.x == 0) {
        unsigned off = 0;
        for (int b = 127; b >= 0; b--) {
            const unsigned long long lo = b < 4 ? (unsigned long long)b : ((4ull + (unsigned long long)(b & 3)) << (b / 4)) >> 2;  // smallest cost of bin b
            const unsigned c = hist[b];
            if (lo > thr) {
                hist[b] = off;
                off += c;
            } else {
                hist[b] = 0xffffffffu;  // the bin holding the threshold (its costs may lie on either side) and all below: not heavy
            }
        }
        base_s = off < kmax ? off : kmax;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 1024) {
        const uint32_t c = cost(i);
        const int b = bin_of(c);
        int32_t r = -1;
        if (hist[b] != 0xffffffffu) {
            const unsigned slot = atomicAdd(&hist[b], 1u);
            if (slot < kmax) {
                r = (int32_t)slot;
                order[slot] = (uint32_t)i;
            }
        }
        hidx[i] = r;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        plan[0] = base_s;
        plan[1] = (uint32_t)(thr > 0xffffffffull ? 0xffffffffull : thr);
        plan[2] = 0;
        plan[3] = 0;
        plan[4] = 0;
        plan[5] = 0;  // primary_walk_kernel's head of the item queue
    }
    if (!border) return;
    // The tile queue: tiles of tw x th wave tiles (one workgroup's worth) with something left to render -- a wave tile that is
    // neither light nor heavy --, costliest histogram bin first.
    const int tiles_x = (wtiles_x + tw - 1) / tw, tiles_y = (wtiles_y + th - 1) / th, n_tiles = tiles_x * tiles_y;
    auto tile_cost = [&](int t, bool &any) -> uint32_t {
        const int tx = t % tiles_x, ty = t / tiles_x;
        unsigned long long c = 0;
        any = false;
        for (int b = 0; b < th; b++)
            for (int a = 0; a < tw; a++) {
                const int wx = tx * tw + a, wy = ty * th + b;
                if (wx >= wtiles_x || wy >= wtiles_y) continue;
                const int i = wy * wtiles_x + wx;
                if ((light && light[i]) || hidx[i] >= 0) continue;
                any = true;
                c += cost_in[i];
            }
        return c > 0xffffffffull ? 0xffffffffu : (uint32_t)c;
    };
    if (threadIdx.x < 128) hist[threadIdx.x] = 0;
    __syncthreads();
    for (int t2 = threadIdx.x; t2 < n_tiles; t2 += 1024) {
        bool any;
        const uint32_t c = tile_cost(t2, any);
        if (any) atomicAdd(&hist[bin_of(c)], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned off = 0;
        for (int b = 127; b >= 0; b--) {
            const unsigned c = hist[b];
            hist[b] = off;
            off += c;
        }
        plan[3] = off;
    }
    __syncthreads();
    for (int t2 = threadIdx.x; t2 < n_tiles; t2 += 1024) {
        bool any;
        const uint32_t c = tile_cost(t2, any);
        if (any) border[atomicAdd(&hist[bin_of(c)], 1u)] = (uint32_t)t2;
    }
}

// The per-pixel camera constants of the heavy tiles (what set_pixel computes), one 64-thread workgroup per heavy tile.
__global__ __launch_bounds__(64) void pixel_const_kernel(GridParams g) {
    const unsigned hr = blockIdx.x;
    if (hr >= g.plan[0]) return;
    const int wtiles_x = (g.W + kWaveTileW - 1) / kWaveTileW;
    const uint32_t wt = g.order[hr];
    const int p = threadIdx.x;
    const int w = (int)(wt % (uint32_t)wtiles_x) * kWaveTileW + (p & 15), j = (int)(wt / (uint32_t)wtiles_x) * kWaveTileH + (p >> 4);
    const int h = global_row(g, j);
    const V3 camorg = mk(g.cam[0], g.cam[1], g.cam[2]);
    // main.cpp:188-189,198,203 -- the same expressions as set_pixel
    const double px = (2.0 * ((double)w / g.W) - 1) * g.half_width;
    const double py = (2.0 * ((double)h / g.H) - 1) * g.half_width * g.H / g.W;
    const V3 pdir = normalized(mk(px, py, 0) - camorg);
    const V3 pof = pdir * ((g.focus_plane - camorg.z) / pdir.z) + camorg;
    const uint64_t k_pix = pixel_key(g.seed, (uint64_t)h * (uint64_t)g.W + (uint64_t)w);
    double *pc = g.pconst + (size_t)hr * (7 * 64) + p;
    pc[0 * 64] = pdir.x; pc[1 * 64] = pdir.y; pc[2 * 64] = pdir.z;
    pc[3 * 64] = pof.x;  pc[4 * 64] = pof.y;  pc[5 * 64] = pof.z;
    pc[6 * 64] = __longlong_as_double((long long)k_pix);
}

// One 64-thread workgroup per heavy tile, one thread per pixel: the pixel's Hitpoint values added sample by sample, in
// emission order within a sample -- the order of the reference's serial loop (main.cpp:204-209 around main.cpp:85-100).
// The additions are a serial chain but the loads are not: the counts and the first value of eight samples are requested
// together (a sample rarely has more than one Hitpoint outside glass), further values of a sample all at once.
__global__ __launch_bounds__(64) void deferred_sum_kernel(GridParams g, float *__restrict__ rgb, uint32_t *__restrict__ nhit_out) {
    const unsigned hr = blockIdx.x;
    if (hr >= g.plan[0]) return;
    const int wtiles_x = (g.W + kWaveTileW - 1) / kWaveTileW;
    const uint32_t wt = g.order[hr];
    const int p = threadIdx.x;
    const int w = (int)(wt % (uint32_t)wtiles_x) * kWaveTileW + (p & 15), j = (int)(wt / (uint32_t)wtiles_x) * kWaveTileH + (p >> 4);
    if (w >= g.W || j >= g.rows) return;
    double r = 0, gg = 0, b = 0;
    uint32_t hits = 0;
    if (global_row(g, j) < g.H) {  // rows past the image (stripe padding) had no units: zero
        constexpr int B = 8;
        for (int s0 = 0; s0 < g.spp; s0 += B) {
            int c[B];
            double v[B][3];
#pragma unroll
            for (int i = 0; i < B; i++) {
                const size_t us = (size_t)hr * g.spp + (s0 + i < g.spp ? s0 + i : g.spp - 1);
                c[i] = s0 + i < g.spp ? (int)g.dcnt[us * 64 + p] : 0;
                const double *q = g.dvals + (us * g.maxhp * 64 + p) * 3;
                v[i][0] = q[0]; v[i][1] = q[1]; v[i][2] = q[2];  // slot 0 always exists (read even when the count is 0)
            }
#pragma unroll
            for (int i = 0; i < B; i++) {
                if (c[i] > 0) {
                    r += v[i][0];
                    gg += v[i][1];
                    b += v[i][2];
                    const size_t us = (size_t)hr * g.spp + s0 + i;
                    for (int q = 1; q < c[i]; q++) {
                        const double *x = g.dvals + ((us * g.maxhp + q) * 64 + p) * 3;
                        r += x[0];
                        gg += x[1];
                        b += x[2];
                    }
                    hits += (uint32_t)c[i];
                }
            }
        }
    }
    const float fr = (float)(r * g.inv_spp_total), fg = (float)(gg * g.inv_spp_total), fb = (float)(b * g.inv_spp_total);
    float *dst = rgb + ((size_t)j * g.W + w) * 3;
    if (g.accumulate) {
        dst[0] += fr; dst[1] += fg; dst[2] += fb;
    } else {
        dst[0] = fr; dst[1] = fg; dst[2] = fb;
    }
    if (nhit_out) nhit_out[(size_t)j * g.W + w] = hits;
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
    one.prun_begin = one.prun_end = 0;  // the list handed down is this one object
    RayKey rk{keys ? keys[ii] : 0ull, 1, true, 0u};
    const LdsAux aux{&bl, nullptr};
    SceneHit h = intersect_scene<true, true, false, false>(sc.objs + obj, 1, 1, one, o, d, rk, on, aux, a, b);
    if (!on) return;
    hit[i] = h.id >= 0 ? 1 : 0;
    len[i] = h.t;
    nrm[3 * i] = h.n.x;
    nrm[3 * i + 1] = h.n.y;
    nrm[3 * i + 2] = h.n.z;
}

// function-level probe: objs[obj]->getSurfaceColor(P) (objects.h:84-86,533-539 -> Texture::color) for n points
__global__ void surface_colors_kernel(DeviceScene sc, int obj, const double *__restrict__ pts, int n, double *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const ObjRec &ob = sc.objs[obj];
    V3 f = ld3(ob.col);
    if (ob.kind == KIND_PLANE && ob.tex >= 0) {
        V3 c;
        if (texture_color(sc.texs[ob.tex], sc.texels, ld3(pts + 3 * i), c)) f = c;
    }
    out[3 * i] = f.x;
    out[3 * i + 1] = f.y;
    out[3 * i + 2] = f.z;
}

#endif
