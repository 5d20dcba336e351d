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
struct HitpointSink {
    double *rec;                // cap x 10 doubles: f(3) pos(3) normal(3) label
    unsigned long long *count;  // appended so far (may exceed cap: then the tail was dropped)
    unsigned long long cap;
};

template <bool TREES, bool BEZ, bool DOF, bool GLASS, bool SPH, bool STATS, bool HPS = false, int NT = 256>
__global__ __launch_bounds__(NT, BEZ ? 3 : ((GLASS && TREES) ? 3 : 4)) void trace_grid_kernel(DeviceScene sc, GridParams g, float *__restrict__ rgb,
                                                             uint32_t *__restrict__ nhit_out,
                                                             unsigned long long *__restrict__ counters,
                                                             HitpointSink hps = HitpointSink{nullptr, nullptr, 0}) {
    using TG = TileGeom<NT>;
    // LDS carve-up: [ pending-ray levels (GLASS) -- aliased by the output tile at the end | objs ]
    extern __shared__ __align__(16) unsigned char lds_raw[];
    float *ltile = reinterpret_cast<float *>(lds_raw);
    ObjRec *lobjs = reinterpret_cast<ObjRec *>(lds_raw + (GLASS ? TG::stack_bytes : TG::tile_bytes));  // n_objs records
    // BEZ: one BezLds per wave behind the object list (16-byte aligned: ObjRec is 128 B)
    unsigned char *lrest = reinterpret_cast<unsigned char *>(lobjs + sc.n_objs);
    LdsAux aux;
    aux.bl = BEZ ? reinterpret_cast<volatile BezLds *>(lrest) + (threadIdx.x >> 6) : nullptr;
    if (BEZ) lrest += (NT / 64) * sizeof(BezLds);
    // TREES: node cache behind that (32-byte records, region is 16-byte aligned)
    NodeRec *lnodes = reinterpret_cast<NodeRec *>(lrest);
    aux.lnodes = (TREES && sc.cached_tree >= 0) ? lnodes : nullptr;
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
        const int n16 = sc.n_objs * (int)(sizeof(ObjRec) / 16);
        for (int k = threadIdx.x; k < n16; k += NT) dst[k] = src[k];
    }
    __syncthreads();

    // which tile, and -- when a tile's samples are split over several workgroups -- which chunk of its samples
    const int tile_blocks = (int)gridDim.x / g.chunks;
    const int chunk = (int)blockIdx.x / tile_blocks;
    int tile_x, tile_y;
    if (!tile_of_block(g, (int)blockIdx.x % tile_blocks, tile_x, tile_y, TG::W, TG::H)) return;  // whole workgroup (no later barrier is missed)
    const int s_end = (g.chunks > 1) ? ((chunk + 1) * g.chunk_spp < g.spp ? (chunk + 1) * g.chunk_spp : g.spp) : g.spp;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // wave = 16x4 pixels, 2x2 waves per workgroup (8x8 per wave measured: meshes equal, C2 7 % slower)
    const int lx = (wave & 1) * 16 + (lane & 15);
    const int ly = (wave >> 1) * 4 + (lane >> 4);
    const int w = tile_x * TG::W + lx;
    const int j = tile_y * TG::H + ly;  // local row
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
    unsigned char *lslot = lds_raw;  // level L, field f of this thread: lslot + L*TG::level_bytes + (f*256 + tid)*8
    int sp = 0;
    int s = (g.chunks > 1) ? chunk * g.chunk_spp : 0;  // next sample to start
    bool have = false;
    V3 o = camorg, d = pdir, adj = mk(1, 1, 1);
    int depth_left = 0;
    uint32_t path = 1;

    while (true) {
        if (!have) {
            if (live && s < s_end) {
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
    }

    if (g.chunks > 1) {
        // split samples: this workgroup's raw fp64 sums; finalize_chunks_kernel adds the chunks in order
        if ((w < g.W) && (j < g.rows)) {  // rows past the image (stripe padding) carry zero sums
            const size_t px = ((size_t)chunk * g.rows + (size_t)j) * g.W + (size_t)w;
            g.partial[3 * px + 0] = acc_r;
            g.partial[3 * px + 1] = acc_g;
            g.partial[3 * px + 2] = acc_b;
            if (g.partial_nhit) g.partial_nhit[px] = my_hits;
        }
    } else {
    // ---- coalesced store through LDS: 32 px x 3 floats = 384 contiguous bytes per tile row ----
    if (GLASS) __syncthreads();  // every wave is done with the pending-ray levels the tile aliases
    ltile[ly * (TG::W * 3) + lx * 3 + 0] = (float)(acc_r * g.inv_spp_total);
    ltile[ly * (TG::W * 3) + lx * 3 + 1] = (float)(acc_g * g.inv_spp_total);
    ltile[ly * (TG::W * 3) + lx * 3 + 2] = (float)(acc_b * g.inv_spp_total);
    __syncthreads();
    for (int k = threadIdx.x; k < TG::H * TG::W * 3; k += NT) {
        const int row = k / (TG::W * 3), col = k % (TG::W * 3);
        const int jj = tile_y * TG::H + row;
        const int ww = tile_x * TG::W + col / 3;
        if (jj < g.rows && ww < g.W) {
            float *dst = rgb + ((size_t)jj * g.W + tile_x * TG::W) * 3 + col;
            *dst = g.accumulate ? *dst + ltile[k] : ltile[k];  // progressive passes add into the fp32 frame
        }
    }
    if (nhit_out && (w < g.W) && (j < g.rows)) nhit_out[(size_t)j * g.W + w] = my_hits;
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

// CGRT_GRID_SPLIT_SAMPLES, second step: chunk sums added in chunk order, scaled, rounded once to fp32.
__global__ void finalize_chunks_kernel(GridParams g, float *__restrict__ rgb, uint32_t *__restrict__ nhit_out) {
    const size_t npx = (size_t)g.rows * g.W;
    const size_t px = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (px >= npx) return;
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
