/*
 * cgrt.h -- C ABI of libcgrt.so, the MI355X (gfx950) implementation of the image-grid eye-ray pass of
 * haoyuzhao123/CGRayTracing.
 *
 * The reference has no plugin / FFI seam; the boundary this library replaces is the call
 *
 *     trace(camorg, dir, objs, Vec3(), Vec3(1,1,1), true, 0, htable, w, h);        main.cpp:209 (DoF form :207)
 *
 * inside the pixel/sample loop of render() (main.cpp:185-219), together with everything that call reaches:
 * trace() main.cpp:42-100,129-157 and the intersect()/getSurfaceColor() virtuals of headers/objects.h,
 * headers/bezier.h and headers/texture.h.  One cgrt_trace_grid() call replaces the whole loop nest for a
 * set of image rows.  The scene-building entry points mirror the reference constructors one to one so that
 * a host program can keep main()'s scene code shape.
 *
 * Conventions: plain C, no C++ or torch types.  Every function returns CGRT_OK (0) or a negative error code
 * and never exits or throws; cgrt_last_error() gives the message for the calling thread.  Object handles are
 * not thread-safe; distinct scenes may be used concurrently from different host threads / devices.
 * Threading: launches on ONE scene handle must be ordered by the caller (same stream, or events between streams) --
 * a handle owns device scratch that consecutive launches reuse (CGRT_GRID_SPLIT_SAMPLES chunk sums).
 * All geometry is IEEE double, like the reference (Vec3 = 3 x double, vec3.h:11-30).
 */
#ifndef CGRT_H
#define CGRT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CGRT_VERSION 112 /* 110: cgrt_photons.initial_radius / .pair_cap, cgrt_ppm_result.n_batch_halvings,
                            cgrt_surface_colors, cgrt_trace_grid_variant; 111: cgrt_scene_wide_dump;
                            112: cgrt_scene_set_build, cgrt_scene_build_info (row f3: structures built on the device) */

enum {
    CGRT_OK = 0,
    CGRT_ERR_INVALID = -1,     /* bad argument / bad handle state                                       */
    CGRT_ERR_IO = -2,          /* mesh file unreadable or malformed                                     */
    CGRT_ERR_DEVICE = -3,      /* HIP runtime error (no GPU, allocation or launch failure)              */
    CGRT_ERR_UNSUPPORTED = -4, /* feature combination not available in this build                        */
    CGRT_ERR_LIMIT = -5        /* scene exceeds a documented limit (e.g. top-level objects per scene)    */
};

typedef struct cgrt_scene cgrt_scene; /* opaque; owns host copies and device buffers */

/* Camera and lens constants of render(), main.cpp:178-181,188-206. */
typedef struct cgrt_camera {
    double cam[3];      /* camorg, main.cpp:181: (0,0,-10)                                              */
    double half_width;  /* the literal 10.0 of main.cpp:188-189: image plane z=0 spans x in [-hw,hw)    */
    double focus_plane; /* main.cpp:178: 20.0                                                            */
    double lens_radius; /* main.cpp:179: 1.5.  0 selects the pinhole call main.cpp:209, >0 the thin-lens
                           call main.cpp:207 with neworg = cam + disc(radius) (sampling.h:35-43)         */
} cgrt_camera;

/* Which part of the W x H x spp grid one call renders.  Compile-time constants of the reference
 * (width,height main.cpp:28-29; MAX_DEPTH :35; num_of_samples :177) become fields here. */
typedef struct cgrt_grid {
    int32_t width, height;  /* global image; row 0 is the BOTTOM row (main.cpp:185,189)                  */
    int32_t rows;           /* local rows rendered by this call = rows of the output buffers             */
    int32_t row_offset;     /* contiguous mode (stripe_nranks <= 1): first global row                    */
    int32_t stripe_rows;    /* block-cyclic mode: stripe height in rows (multiple of 8)                  */
    int32_t stripe_rank;    /*   local row j is global row ((j/S)*nranks + rank)*S + j%S                 */
    int32_t stripe_nranks;  /*   rows that fall beyond `height` are left zero                            */
    int32_t spp;            /* samples traced by this call                                               */
    int32_t sample_offset;  /* index of the first sample (keys the lens stream)                          */
    int32_t spp_total;      /* normaliser: output = sum / spp_total (== spp for a single pass)           */
    int32_t max_depth;      /* MAX_DEPTH, main.cpp:35 (1..5)                                             */
    int32_t flags;          /* bit set of CGRT_GRID_* below                                              */
    uint64_t seed;          /* seed of the counter-based lens / Bezier streams                           */
} cgrt_grid;

enum {
    CGRT_GRID_STATS = 1,      /* also count tree-node and triangle tests (CGRT_CNT_NODE_TESTS / _TRI_TESTS)          */
    CGRT_GRID_ACCUMULATE = 2, /* rgb += this pass instead of rgb = this pass: progressive multi-pass rendering with
                                 sample_offset / spp_total, the fp32 replacement for the reference's average.cpp,
                                 which averages nine uint8 images with truncating division (average.cpp:21-64)     */
    CGRT_GRID_NO_REORDER = 8, /* render every tile in image order by its own workgroup.  By default a launch of >= 4 samples
                                 per pixel on a scene with a mesh, bump floor or Bezier object is cost-scheduled: one sample of
                                 every 16x4-pixel wave tile is traced first to measure it; tiles that alone would hold a wave
                                 slot for more than 1/32 of the frame's ideal duration are HEAVY and are rendered through a
                                 queue of (pixel, sample) units that any free lane of any heavy wave takes, their Hitpoint
                                 values parked in HBM and added per pixel afterwards in the reference's order (sample by
                                 sample, emission order inside a sample); the other tiles follow in image order.  Image, hit
                                 counts and counters are bit-identical either way; this flag exists to measure the difference
                                 and to test both paths.  Device memory: up to 12 GiB (at most 1/8 of the device) of parked
                                 values per scene handle, sized by the largest launch (CGRT_DEFER_BYTES overrides)    */
    CGRT_GRID_FORCE_REORDER = 16, /* cost-schedule sphere-only scenes too (off by default: measured no gain on them)    */
    CGRT_GRID_SPLIT_SAMPLES = 4 /* let several workgroups share a tile's samples: each sums a contiguous chunk of the
                                 samples in fp64 and the chunk sums are added in chunk order by a second kernel.
                                 Reproducible, but the fp64 summation ORDER differs from the sample-by-sample sum
                                 (the last bit of a sum may differ before the rounding to fp32).  It fills the GPU
                                 when a few tiles carry most of the work and exactness of the last fp64 bit is not needed.
                                 (Round 1 forced it on for Bezier scenes; the cost scheduler above now balances them
                                 with the exact summation order, so it is purely opt-in.)                            */
};

/* indices into the uint64 counters[CGRT_NCOUNTERS] array written by cgrt_trace_grid (added to, not reset) */
enum {
    CGRT_CNT_RAYS = 0,      /* trace() invocations past the depth test (main.cpp:46)                     */
    CGRT_CNT_HITPOINTS = 1, /* Hitpoints the reference would have inserted (main.cpp:98)                 */
    CGRT_CNT_WAVE_ITERS = 2,/* wavefront loop iterations (x64 = lane slots; lane utilisation = rays/slots)*/
    CGRT_CNT_NODE_TESTS = 3,/* tree nodes visited (lane granularity)                                     */
    CGRT_CNT_TRI_TESTS = 4, /* triangle tests                                                            */
    CGRT_NCOUNTERS = 8
};

typedef struct cgrt_scene_stats {
    int32_t n_objects, n_spheres, n_planes, n_meshes, n_beziers, n_textures, n_trees, committed;
    int64_t n_triangles;   /* leaf triangles over all trees (mesh + bump)                                */
    int64_t n_nodes;
    int64_t device_bytes;  /* bytes resident in HBM for this scene                                       */
    int64_t scene_bytes_fp64; /* S_scene of SURVEY.md section 8d: 88/sphere, 100/plane, 56/node + 72/tri, 3/texel */
} cgrt_scene_stats;

int cgrt_version(void);
const char *cgrt_last_error(void);

/* ---- scene construction; each add_* returns the object's position in `objs` (>= 0) or an error (< 0).
 *      Order matters exactly as in vector<Object*> objs (main.cpp:355-366): on equal hit distance the
 *      earlier object wins (main.cpp:57). */
int cgrt_scene_create(cgrt_scene **out);
void cgrt_scene_destroy(cgrt_scene *s);

/* ---- row f3 of SURVEY.md section 8: WHERE the acceleration structures of OPAQUE owners are built -------------------------
 * CGRT_BUILD_HOST (default): KDTree::buildKdTree's leaf order (objects.h:217-267, the same std::sort over the same
 * sequence), the bump mesh of objects.h:480-504 and the height field of texture.h:19-38 are produced on the host by the add_*
 * calls, bit for bit the reference's; the device hierarchies are derived from them.
 * CGRT_BUILD_DEVICE: for objects added AFTER this call whose transparency is < 1e-4, cgrt_scene_commit builds them on the
 * GPU instead -- a bump floor's heights, vertices and grid cells from the texture bytes; an opaque mesh's triangle-level
 * hierarchy from the triangle soup (Morton codes, radix sort, binary radix tree, 4-wide collapse), its bounding and cover
 * spheres.  Transparent owners keep the host build (their normals depend on the reference's leaf ORDER, quirk Q5).
 * Parity class of this mode: tolerance, not bit-exact -- every ray with a unique nearest hit gets the same hit, bit for
 * bit; exact ties (a ray through a shared edge or vertex) go to the lower construction index instead of the reference's
 * leaf-order rule, and heights use a correctly rounded exp where the reference's libm is only within 0.52 ulp.  The
 * verification dumps that describe the reference's tree (cgrt_scene_tree_dump, cgrt_scene_bvh_dump) are empty for a
 * device-built tree; cgrt_scene_wide_dump / cgrt_scene_bvh_order read the device's records back.
 * The environment variable CGRT_BUILD=device makes CGRT_BUILD_DEVICE the default of scenes created afterwards. */
enum { CGRT_BUILD_HOST = 0, CGRT_BUILD_DEVICE = 1 };
int cgrt_scene_set_build(cgrt_scene *s, int mode);

typedef struct cgrt_build_info {
    int32_t mode;            /* CGRT_BUILD_*                                                                              */
    int32_t n_device_trees;  /* trees the commit built on the device                                                      */
    double ms_host_build;    /* wall-clock the add_* calls spent building trees / bump meshes on the host                 */
    double ms_device_build;  /* wall-clock of the device builds inside cgrt_scene_commit (uploads of their inputs included) */
    double ms_commit;        /* wall-clock of cgrt_scene_commit as a whole                                                */
} cgrt_build_info;
int cgrt_scene_build_info(const cgrt_scene *s, cgrt_build_info *out);

/* Sphere(c, r, sc, refl, transp)                                                   objects.h:28-38 */
int cgrt_scene_add_sphere(cgrt_scene *s, const double c[3], double r, const double sc[3], double refl, double transp);

/* Texture(data, n, p, lx, ly, bump)   texture.h:19-38.  rgb = rows*cols*3 bytes as decoded (texel = byte/256,
 * main.cpp:303-316).  Returns a texture id for cgrt_scene_add_plane. */
int cgrt_scene_add_texture(cgrt_scene *s, const uint8_t *rgb, int rows, int cols, const double n[3],
                           const double p[3], double lx, double ly, int isbump);

/* Plane(p, n, sc, refl, transp, tx)   objects.h:480-504.  tex_id < 0: no texture.  A bump texture on a
 * plane with |n.y-1| < 1e-5 builds the displacement mesh and its tree (objects.h:482-503). */
int cgrt_scene_add_plane(cgrt_scene *s, const double p[3], const double n[3], const double sc[3], double refl,
                         double transp, int tex_id);

/* TriangleMesh(filename, a, b, sc, refl, transp, typeofdata)   objects.h:338-403: the three text formats,
 * vertex -> (x, y, -z) * a + b.  A missing file yields an empty mesh like the reference; a malformed file
 * is CGRT_ERR_IO (the reference's behaviour there is undefined). */
int cgrt_scene_add_mesh_file(cgrt_scene *s, const char *filename, double a, const double b[3], const double sc[3],
                             double refl, double transp, int typeofdata);
/* Same object from ntri*9 doubles (pa, pb, pc per triangle, already transformed). */
int cgrt_scene_add_mesh_triangles(cgrt_scene *s, const double *tri9, int ntri, const double sc[3], double refl,
                                  double transp, int typeofdata);

/* Bezier(points, pos, sc, refl, transp)   bezier.h:44-71; 1..6 control points. */
int cgrt_scene_add_bezier(cgrt_scene *s, const double *cp3, int ncp, const double pos[3], const double sc[3],
                          double refl, double transp);

/* Builds the trees (KDTree::buildKdTree, objects.h:217-267, same leaf order), flattens the scene and uploads
 * it to HIP device `device`.  Must be called once before tracing; the scene is immutable afterwards. */
int cgrt_scene_commit(cgrt_scene *s, int device);
int cgrt_scene_get_stats(const cgrt_scene *s, cgrt_scene_stats *out);

/* ---- host-side prerequisites exposed for verification (tree fingerprints, loader output) ----
 * tree index t: 0..n_trees-1 in object order (a mesh's tree, or a bump plane's tree).
 * node_lr_size: nnodes*3 int32 (left, right, triangle count) in the reference's node numbering;
 * leaf_ids: triangle ids of all leaves in node order; bbox: nnodes*6 (xmin,xmax,ymin,ymax,zmin,zmax).
 * tri9: the tree's triangles in construction order.  Any pointer may be NULL. */
int cgrt_scene_tree_sizes(const cgrt_scene *s, int t, int32_t *nnodes, int32_t *nleaftris, int32_t *ntris);
int cgrt_scene_tree_dump(const cgrt_scene *s, int t, int32_t *node_lr_size, int32_t *leaf_ids, double *bbox,
                         double *tri9);

/* The hierarchy the device actually traverses over the reference's leaves (a binned-SAH tree stored once per ray-direction
 * octant, children near-to-far): *nnodes = nodes per octant; box6 = 8*nnodes x {lo(3), hi(3)}
 * (grown and rounded outward to fp32); skip_leaf2 = 8*nnodes x {skip link relative to the octant's first node,
 * leaf = -1 for inner nodes else (first triangle in leaf order << 4) | count}.  Either array may be NULL
 * (call once with both NULL to size them). */
int cgrt_scene_bvh_dump(const cgrt_scene *s, int tree, int32_t *nnodes, float *box6, int32_t *skip_leaf2);
/* *tri_level = 1 when the owner is opaque and the hierarchy goes down to single triangles: leaf references of
 * cgrt_scene_bvh_dump then index the hierarchy's own triangle order, and order[j] (ntris entries, may be NULL) is the
 * leaf-order index of the triangle at position j; 0: leaf references index the reference's leaf order directly. */
int cgrt_scene_bvh_order(const cgrt_scene *s, int tree, int32_t *tri_level, int32_t *order);
/* The 4-wide form of a triangle-level hierarchy (what the device walks for an opaque owner; *nwide = 0 for other trees):
 * node i has up to four children, box24[(4*i + k)*6 ..] = lo[3], hi[3] of child k and ref4[4*i + k] = its reference --
 * >= 0 a leaf ((first triangle of the hierarchy's order << 4) | count), INT32_MIN an empty slot, otherwise ~(child node).
 * *stack_need = the deepest the walk's stack can get (always below 64).  Two-call protocol: box24 / ref4 may be NULL. */
int cgrt_scene_wide_dump(const cgrt_scene *s, int tree, int32_t *nwide, int32_t *stack_need, float *box24, int32_t *ref4);

/* ---- the hot path -------------------------------------------------------------------------------------
 * Renders grid->rows rows: for every pixel and sample it runs the reference's trace(flag=true) recursion
 * (iteratively) and writes
 *     rgb [rows*width*3] float : (1/spp_total) * sum over samples and diffuse hits of f*adj  (main.cpp:88)
 *     nhit[rows*width]   uint32: number of Hitpoints of the pixel (may be NULL)
 *     counters[CGRT_NCOUNTERS] uint64: totals, ADDED to the existing values (may be NULL)
 * rgb, nhit and counters are DEVICE pointers on the scene's device; the launch is asynchronous on `stream`
 * (a hipStream_t passed as void*, NULL = default stream).  Arithmetic is fp64 on the device; the only fp32
 * rounding is the final store. */
int cgrt_trace_grid(const cgrt_scene *s, const cgrt_camera *cam, const cgrt_grid *grid, float *rgb,
                    uint32_t *nhit, uint64_t *counters, void *stream);

/* Row e of SURVEY.md section 8: the un-permute that follows the framebuffer gather.  shares = n_present buffers
 * [rows_local][width][channels] float, share-major (share r's local row j is global row ((j / stripe_rows) * nshares + r) *
 * stripe_rows + j % stripe_rows, as in cgrt_grid); frame = [height][width][channels].  Rows of shares >= n_present are
 * written as zero.  DEVICE pointers on the current device; asynchronous on `stream`. */
int cgrt_unpermute_stripes(const float *shares, int n_present, int nshares, int width, int height, int stripe_rows,
                           int rows_local, int channels, float *frame, void *stream);

/* Convenience form with HOST output buffers: allocates device scratch, runs, synchronises and copies back
 * (counters are overwritten, not added to). */
int cgrt_trace_grid_host(const cgrt_scene *s, const cgrt_camera *cam, const cgrt_grid *grid, float *rgb,
                         uint32_t *nhit, uint64_t *counters);

/* The Hitpoint stream the reference would have inserted into its hash table (main.cpp:87-98, hash.h:43-54), for
 * the same grid: up to `cap` records of 10 doubles {f(3) = surface colour * adj, pos(3), normal(3), label} are written
 * to the HOST buffer hp10 in no particular order; label = ((sample_index * (rows*width) + local pixel index) << 4) |
 * position of the hitpoint in that sample's emission order.  *count
 * receives the number of hitpoints produced (if > cap the excess was dropped).  This is the hand-off a photon pass
 * would consume and what parity tests compare with the reference's own records. */
int cgrt_trace_grid_hitpoints(const cgrt_scene *s, const cgrt_camera *cam, const cgrt_grid *grid, double *hp10,
                              uint64_t cap, uint64_t *count);

/* ---- row f1 of SURVEY.md section 8: the photon pass and final gather, render() main.cpp:223-258 ----------------
 * Constants of the reference as fields.  The reference races eight OpenMP threads over time-seeded rand(); what is
 * implemented is its SERIAL meaning (photons in index order, photon i on the keyed stream (seed, i)), which the
 * compiled reference reproduces on one thread and which golden vectors pin bit for bit. */
typedef struct cgrt_photons {
    double light[3];   /* lightorg, main.cpp:180: (0, 19.999, 20)                                                  */
    double jitter;     /* main.cpp:240-241: emitter half extent 2.0 (a, b = u*4-2)                                  */
    double power;      /* main.cpp:246: 700 (flux = power * 4*PI per channel)                                      */
    double alpha;      /* main.cpp:36: 0.7                                                                         */
    int64_t nphotons;  /* photons in total (reference: num_photon * num_threads = 20 480 000, main.cpp:223-224)     */
    int32_t hashsize;  /* main.cpp:184: 1000001 (bucket collisions are part of the semantics, hash.h:32-37)         */
    int32_t batch;     /* photons traced per batch (0 = default 1048576); does not change the result               */
    uint64_t seed;
    double initial_radius; /* radius every Hitpoint starts with and, through it, the hash cell length (hash.h:25-26).
                            * The reference ties both to its compile-time `height`: r = 200.0/height (main.cpp:84,183),
                            * 200/768 as committed.  0 selects 200/768; a host mirroring a reference built for another
                            * height passes that build's 200.0/height                                               */
    int64_t pair_cap;  /* capacity of the per-batch (hitpoint, photon hit) pair buffer; 0 = automatic (128 per hitpoint,
                        * between 4 M and 128 M).  A batch whose pairs do not fit is redone in halves -- the result does
                        * not depend on it; tests set it low to drive that path                                     */
} cgrt_photons;

/* What cgrt_ppm_render hands back; every pointer is a HOST buffer owned by the caller and may be NULL. */
typedef struct cgrt_ppm_result {
    double *image;     /* rows*width*3: image[h][w] = sum over the pixel's hitpoints of flux / (PI * r2 * nphotons * spp),
                        * row 0 = bottom (main.cpp:252-258)                                                        */
    uint8_t *rgb8;     /* rows*width*3: the PNG pixels of main.cpp:403-412 -- top row first, gammaCorr() per channel
                        * (row f2; same values as cgrt_tonemap_rgb8(image))                                        */
    double *hp16;      /* hp_cap x 16: per hitpoint {pixel*spp+sample, emission index, f(3), pos(3), normal(3), flux(3),
                        * r2, n} in the reference's table order (bucket, then insertion order)                     */
    uint64_t hp_cap;
    uint64_t hp_count; /* out: hitpoints the eye pass produced                                                     */
    uint64_t n_events; /* out: diffuse photon hits processed (main.cpp:103-125 executions)                         */
    uint64_t n_pairs;  /* out: (hitpoint, photon hit) candidate pairs replayed                                      */
    uint64_t n_batch_halvings; /* out: times a batch overflowed the pair buffer and was redone in halves            */
    double ms_eye, ms_table, ms_photons, ms_gather; /* out: device time of the four stages, milliseconds            */
} cgrt_ppm_result;

/* Eye pass + photon pass + final gather (+ tone map) for grid: the whole of render(), main.cpp:169-258, and the pixel
 * loop of main(), main.cpp:403-412.
 * Sharding (multi-GPU): a hitpoint's history depends only on the ordered photon hits that reach it, never on other
 * hitpoints, so a rank may own just the rows of its grid (a band, or block-cyclic stripes as in cgrt_trace_grid):
 * it traces ALL photons but keeps, searches and updates only its own hitpoints, and its image rows are bit-identical
 * to the same rows of a single full-frame call.  rgb8 needs contiguous rows (it flips them); hp16[0] is
 * local_pixel*spp + sample. */
int cgrt_ppm_render(const cgrt_scene *s, const cgrt_camera *cam, const cgrt_grid *grid, const cgrt_photons *ph,
                    cgrt_ppm_result *out);

/* ---- row f2 of SURVEY.md section 8: tone map, flip, PNG (util.h:45-47, main.cpp:403-412) ----------------------
 * rgb8[(height-1-h)*width + w][c] = int(pow(1 - exp(-image[h][w][c]), 1/2.2) * 255 + .5) computed on `device`;
 * HOST buffers.  NaN and negative inputs give 0 (the reference's int(NaN) is undefined). */
int cgrt_tonemap_rgb8(int device, const double *image, int width, int height, uint8_t *rgb8);

/* 8-bit RGB PNG, rows top to bottom, like stbi_write_png("test.png", width, height, 3, data, width*3) at main.cpp:412.
 * Self-contained encoder (stored deflate blocks; no zlib dependency).  Host only. */
int cgrt_write_png(const char *path, int width, int height, const uint8_t *rgb8);

/* Verification probe for photon paths: the diffuse hits (the events the serial loop of main.cpp:103-125 processes) of
 * photons [first, first+count): events9 = count*8 slots of 9 doubles {P(3), n(3), flux(3)}, slot = (photon-first)*8 +
 * path segment; valid = one byte per slot.  HOST buffers. */
int cgrt_photon_events(const cgrt_scene *s, const cgrt_photons *ph, int max_depth, int64_t first, int32_t count,
                       double *events9, uint8_t *valid);

/* Function-level probe used by parity tests: objs[obj]->intersect(org, dir, len, normal) for n rays on the
 * device (host pointers; keys: per-ray stream key for Bezier draws, may be NULL). */
int cgrt_intersect_rays(const cgrt_scene *s, int obj, const double *org3, const double *dir3, const uint64_t *keys,
                        int n, int32_t *hit, double *len, double *normal3);

/* Function-level probe: objs[obj]->getSurfaceColor(P) for n points on the device -- the flat colour, or for a textured
 * plane Texture::color (texture.h:39-72: three axis-aligned orientations, nearest texel) with the flat colour outside
 * the texture rectangle (objects.h:533-539).  HOST buffers, n*3 doubles each. */
int cgrt_surface_colors(const cgrt_scene *s, int obj, const double *points3, int n, double *colors3);

/* Name of the trace_grid_kernel instantiation cgrt_trace_grid would launch for (scene, cam, grid) -- the kernel name a
 * rocprofv3 kernel trace shows; written NUL-terminated into name[cap]. */
int cgrt_trace_grid_variant(const cgrt_scene *s, const cgrt_camera *cam, const cgrt_grid *grid, char *name, size_t cap);

/* Host evaluation of the lens stream (cgrt_rng.hpp, the same inline code the kernel runs): writes
 * uniform_sampling_circle(radius) (sampling.h:35-43) for n (pixel, sample) pairs as 3 doubles each.  Lets CPU-only
 * tests pin the stream against the reference's sampler without a GPU. */
int cgrt_lens_samples(uint64_t seed, const int64_t *pixel, const int32_t *sample, int n, double radius, double *out3);

#ifdef __cplusplus
}
#endif
#endif /* CGRT_H */
