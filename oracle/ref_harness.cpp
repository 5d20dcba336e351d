// TEST INFRASTRUCTURE — builds oracle/_ref/libcgrt_ref.so.  Never shipped, never on the product path.
//
// This translation unit is OUR code.  It compiles the UNMODIFIED reference
// (`#include "main.cpp"`, found through -I/root/reference at build time; no reference
// source is copied into this repository) and exposes its hot path -- trace(),
// main.cpp:42-167, with the scene classes of headers/objects.h / bezier.h / texture.h --
// through a plain C interface so that tests can (a) generate golden vectors,
// (b) validate the CPU restatement in oracle/cgrt_oracle.cpp, (c) time the reference's
// own eye pass on the host cores (bench.py cpu_baseline kind = "reference").
//
// Harness techniques:
//   * `#define main` renames the reference's main(); the scene is built here through the
//     reference's own constructors.
//   * `#define rand()` redirects libc rand() (sampling.h:31-43, bezier.h:183,236,239) to the keyed
//     counter stream of cgrt_rng.h, so the reference's OWN uniform_sampling_circle() draws the
//     same lens samples as the oracle and the GPU.
//   * `#define private public` gives access to TriangleMesh/Plane internals for tree
//     fingerprints and for building a mesh from an in-memory triangle list.
//   * rays are counted by a never-hit Object appended to objs (called once per ray, main.cpp:55-56).
//   * Hashtable(1, r) makes bucket 0 the emission-ordered hitpoint list (hash.h:36-38).
//   * the pixel loop/camera formulas of render() (main.cpp:185-209) are restated here because
//     render() hard-codes width/height/spp; trace() itself only reads `height` for r^2 (main.cpp:84).
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <vector>
#include <ctime>
#include <algorithm>
#include <utility>
#include <string>
#include <chrono>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <assert.h>
#include <stdarg.h>
#include <stddef.h>
#include <limits.h>

#include "cgrt_rng.h"
#include "cgrt_testapi.h"

static uint64_t g_rng_key = 0;
static uint32_t g_rng_ctr = 0;
extern "C" int cgrt_ref_rand(void) { return (int)cgrt_rand31(g_rng_key, g_rng_ctr++); }

#define rand() cgrt_ref_rand()
#define private public
#define main cgrt_ref_unused_main
#include "main.cpp"
#undef main
#undef private
#undef rand

namespace {

struct Probe : public Object {
    mutable uint64_t n = 0;
    bool intersect(const Vec3 &, const Vec3 &, double &, Vec3 &) const { n++; return false; }
    double getTransparency() const { return 0; }
    double getReflection() const { return 0; }
    Vec3 getSurfaceColor(const Vec3 &) const { return Vec3(); }
};

struct RefScene {
    std::vector<Object *> objs;          // reference order (main.cpp:355-366 semantics: first wins ties)
    std::vector<Texture *> textures;
    std::vector<TriangleMesh *> meshes;
    std::vector<Plane *> planes;
    Probe probe;
};

Vec3 v3(const double *p) { return Vec3(p[0], p[1], p[2]); }

void dump_tree(const KDTree &t, int32_t *node_lr_size, int32_t *leaf_ids, double *bbox) {
    size_t k = 0;
    for (size_t i = 0; i < t.kdnodes.size(); i++) {
        const KDNode &nd = t.kdnodes[i];
        if (node_lr_size) {
            node_lr_size[3 * i + 0] = nd.left;
            node_lr_size[3 * i + 1] = nd.right;
            node_lr_size[3 * i + 2] = (int32_t)nd.triangleList.size();
        }
        if (bbox) {
            bbox[6 * i + 0] = nd.xmin; bbox[6 * i + 1] = nd.xmax;
            bbox[6 * i + 2] = nd.ymin; bbox[6 * i + 3] = nd.ymax;
            bbox[6 * i + 4] = nd.zmin; bbox[6 * i + 5] = nd.zmax;
        }
        if ((int)nd.triangleList.size() < Minkdsize && leaf_ids) {
            for (size_t j = 0; j < nd.triangleList.size(); j++) leaf_ids[k++] = nd.triangleList[j].first;
        }
    }
}
size_t count_leaf_tris(const KDTree &t) {
    size_t k = 0;
    for (size_t i = 0; i < t.kdnodes.size(); i++)
        if ((int)t.kdnodes[i].triangleList.size() < Minkdsize) k += t.kdnodes[i].triangleList.size();
    return k;
}

}  // namespace

extern "C" {

void *ref_scene_new(void) { return new RefScene(); }
void ref_scene_free(void *p) { delete (RefScene *)p; }  // objects intentionally leaked (test process)

int ref_add_sphere(void *sp, const double *c, double r, const double *col, double refl, double transp) {
    RefScene *s = (RefScene *)sp;
    s->objs.push_back(new Sphere(v3(c), r, v3(col), refl, transp));
    return (int)s->objs.size() - 1;
}

// rgb: rows*cols*3 bytes, row-major as stbi_load returns them; texel = byte/256 (main.cpp:303-316)
int ref_add_texture(void *sp, const uint8_t *rgb, int rows, int cols, const double *n, const double *p,
                    double lx, double ly, int bump) {
    RefScene *s = (RefScene *)sp;
    vector<vector<Vec3> > tdata;
    int ctr = 0;
    for (int i = 0; i < rows; i++) {
        vector<Vec3> v;
        for (int j = 0; j < cols; j++) {
            Vec3 col = Vec3();
            col.x = (double)rgb[ctr] / (double)256; ctr++;
            col.y = (double)rgb[ctr] / (double)256; ctr++;
            col.z = (double)rgb[ctr] / (double)256; ctr++;
            v.push_back(col);
        }
        tdata.push_back(v);
    }
    s->textures.push_back(new Texture(tdata, v3(n), v3(p), lx, ly, bump != 0));
    return (int)s->textures.size() - 1;
}

int ref_add_plane(void *sp, const double *p, const double *n, const double *col, double refl, double transp,
                  int tex_id) {
    RefScene *s = (RefScene *)sp;
    Plane *pl;
    if (tex_id >= 0) {
        pl = new Plane(v3(p), v3(n), v3(col), refl, transp, *s->textures[tex_id]);
    } else {
        // Q4: Texture() leaves isbump uninitialised (texture.h:16-18) while Plane::intersect reads it
        // (objects.h:513).  Pass a default Texture whose isbump is forced false.
        Texture t;
        t.isbump = false;
        pl = new Plane(v3(p), v3(n), v3(col), refl, transp, t);
    }
    s->planes.push_back(pl);
    s->objs.push_back(pl);
    return (int)s->objs.size() - 1;
}

// Reference loader + tree build (objects.h:338-403).  Q9: uses freopen(stdin); call at most once per process.
int ref_add_mesh_file(void *sp, const char *file, double a, const double *b, const double *col, double refl,
                      double transp, int typeofdata) {
    RefScene *s = (RefScene *)sp;
    TriangleMesh *m = new TriangleMesh((char *)file, a, v3(b), v3(col), refl, transp, typeofdata);
    s->meshes.push_back(m);
    s->objs.push_back(m);
    return (int)s->objs.size() - 1;
}

// Mesh from an in-memory triangle list (ntri*9 doubles, already transformed): default ctor + the
// reference's own buildKdTree (objects.h:217-267,402).
int ref_add_mesh_tris(void *sp, const double *tri, int ntri, const double *col, double refl, double transp,
                      int typeofdata) {
    RefScene *s = (RefScene *)sp;
    TriangleMesh *m = new TriangleMesh();
    m->surfaceColor = v3(col);
    m->transparency = transp;
    m->reflection = refl;
    m->objtype = typeofdata;
    for (int i = 0; i < ntri; i++) {
        Triangle t(v3(tri + 9 * i), v3(tri + 9 * i + 3), v3(tri + 9 * i + 6));
        m->triangles.push_back(pair<int, Triangle>((int)m->triangles.size(), t));
    }
    m->kdtree.buildKdTree(m->triangles, 0, false, 0, true);
    s->meshes.push_back(m);
    s->objs.push_back(m);
    return (int)s->objs.size() - 1;
}

int ref_add_bezier(void *sp, const double *cp, int ncp, const double *pos, const double *col, double refl,
                   double transp) {
    RefScene *s = (RefScene *)sp;
    vector<Vec3> pts;
    for (int i = 0; i < ncp; i++) pts.push_back(v3(cp + 3 * i));
    s->objs.push_back(new Bezier(pts, v3(pos), v3(col), refl, transp));
    return (int)s->objs.size() - 1;
}

// ---- introspection for loader / tree-build pinning -------------------------------------------
int ref_mesh_ntris(void *sp, int mesh) { return (int)((RefScene *)sp)->meshes[mesh]->triangles.size(); }
void ref_mesh_tris(void *sp, int mesh, double *out) {
    TriangleMesh *m = ((RefScene *)sp)->meshes[mesh];
    for (size_t i = 0; i < m->triangles.size(); i++) {
        const Triangle &t = m->triangles[i].second;
        double *o = out + 9 * i;
        o[0] = t.pa.x; o[1] = t.pa.y; o[2] = t.pa.z;
        o[3] = t.pb.x; o[4] = t.pb.y; o[5] = t.pb.z;
        o[6] = t.pc.x; o[7] = t.pc.y; o[8] = t.pc.z;
    }
}
static const KDTree *pick_tree(void *sp, int kind, int idx) {
    RefScene *s = (RefScene *)sp;
    return kind == 0 ? &s->meshes[idx]->kdtree : &s->planes[idx]->bumpmapping;
}
// kind 0: mesh idx; kind 1: plane idx (bump tree)
int ref_tree_nnodes(void *sp, int kind, int idx) { return (int)pick_tree(sp, kind, idx)->kdnodes.size(); }
int ref_tree_nleaftris(void *sp, int kind, int idx) { return (int)count_leaf_tris(*pick_tree(sp, kind, idx)); }
void ref_tree_dump(void *sp, int kind, int idx, int32_t *node_lr_size, int32_t *leaf_ids, double *bbox) {
    dump_tree(*pick_tree(sp, kind, idx), node_lr_size, leaf_ids, bbox);
}
// root triangle list of a bump tree = the bump mesh in construction order (objects.h:485-497)
int ref_plane_bump_ntris(void *sp, int idx) {
    const KDTree &t = ((RefScene *)sp)->planes[idx]->bumpmapping;
    return t.kdnodes.empty() ? 0 : (int)t.kdnodes[0].triangleList.size();
}
void ref_plane_bump_tris(void *sp, int idx, double *out) {
    const KDTree &t = ((RefScene *)sp)->planes[idx]->bumpmapping;
    if (t.kdnodes.empty()) return;
    const vector<pair<int, Triangle> > &l = t.kdnodes[0].triangleList;
    for (size_t i = 0; i < l.size(); i++) {
        const Triangle &q = l[i].second;
        double *o = out + 9 * i;
        o[0] = q.pa.x; o[1] = q.pa.y; o[2] = q.pa.z;
        o[3] = q.pb.x; o[4] = q.pb.y; o[5] = q.pb.z;
        o[6] = q.pc.x; o[7] = q.pc.y; o[8] = q.pc.z;
    }
}

// ---- function-level probes --------------------------------------------------------------------
// objs[obj]->intersect on a batch of rays; Bezier draws come from stream keys[i].
void ref_intersect_batch(void *sp, int obj, const double *org, const double *dir, const uint64_t *keys, int n,
                         int32_t *hit, double *len, double *normal) {
    RefScene *s = (RefScene *)sp;
    for (int i = 0; i < n; i++) {
        g_rng_key = keys ? keys[i] : 0;
        g_rng_ctr = 0;
        double l = 0;
        Vec3 nv;
        bool h = s->objs[obj]->intersect(v3(org + 3 * i), v3(dir + 3 * i), l, nv);
        hit[i] = h ? 1 : 0;
        len[i] = l;
        normal[3 * i] = nv.x; normal[3 * i + 1] = nv.y; normal[3 * i + 2] = nv.z;
    }
}
void ref_surface_color_batch(void *sp, int obj, const double *pts, int n, double *out) {
    RefScene *s = (RefScene *)sp;
    for (int i = 0; i < n; i++) {
        Vec3 c = s->objs[obj]->getSurfaceColor(v3(pts + 3 * i));
        out[3 * i] = c.x; out[3 * i + 1] = c.y; out[3 * i + 2] = c.z;
    }
}
// the reference's own lens sampler on the keyed stream (sampling.h:35-43)
void ref_lens_samples(uint64_t seed, const int64_t *pix, const int32_t *smp, int n, double radius, double *out) {
    for (int i = 0; i < n; i++) {
        g_rng_key = cgrt_key(seed, (uint64_t)pix[i], (uint64_t)smp[i], 0);
        g_rng_ctr = 0;
        Vec3 v = uniform_sampling_circle(radius);
        out[3 * i] = v.x; out[3 * i + 1] = v.y; out[3 * i + 2] = v.z;
    }
}

// ---- the eye pass: render()'s grid loop (main.cpp:185-219) around the reference trace() ----------
// hashsize: 1 => emission-ordered bucket (fixtures); 1000001 => the reference's own table (timing).
// Returns seconds spent in the pixel loop (Hashtable construction and result gathering excluded).
double ref_trace_grid(void *sp, const orc_camera *cam, const orc_grid *g, int hashsize, double *acc,
                      uint32_t *nhit, uint64_t *nrays, double *hp, int64_t *hp_pix, uint64_t hp_cap,
                      uint64_t *hp_count) {
    RefScene *s = (RefScene *)sp;
    std::vector<Object *> objs = s->objs;
    s->probe.n = 0;
    objs.push_back(&s->probe);
    const int W = g->W, H = g->H;
    Vec3 camorg = v3(cam->cam);
    double r = 200.0 / height;
    Hashtable htable = Hashtable(hashsize, r);
    const int start_depth = MAX_DEPTH - g->depth;
    auto t0 = std::chrono::steady_clock::now();
    for (int h = g->row0; h < g->row0 + g->nrows; h++) {
        for (int w = 0; w < W; w++) {
            double x = (2.0 * ((double)w / W) - 1) * cam->half_width;
            double y = (2.0 * ((double)h / H) - 1) * cam->half_width * H / W;
            Vec3 dir = (Vec3(x, y, 0) - camorg).normalize();
            Vec3 point_on_focus = dir * ((cam->focus_plane - camorg.z) / dir.z) + camorg;
            for (int j = g->sample0; j < g->sample0 + g->spp; j++) {
                int lw = w + W * (j - g->sample0);  // label: pixel column + W*sample (Hitpoint::w, main.cpp:91)
                int lh = h - g->row0;
                if (cam->lens_radius > 0) {
                    g_rng_key = cgrt_key(g->seed, (uint64_t)h * (uint64_t)W + (uint64_t)w, (uint64_t)j, 0);
                    g_rng_ctr = 0;
                    Vec3 neworg = camorg + uniform_sampling_circle(cam->lens_radius);
                    Vec3 newdir = (point_on_focus - neworg).normalize();
                    // Bezier draws inside trace() continue on a per-sample stream (statistical parity only)
                    g_rng_key = cgrt_key(g->seed, (uint64_t)h * (uint64_t)W + (uint64_t)w, (uint64_t)j, 1);
                    g_rng_ctr = 0;
                    trace(neworg, newdir, objs, Vec3(), Vec3(1, 1, 1), true, start_depth, htable, lw, lh);
                } else {
                    g_rng_key = cgrt_key(g->seed, (uint64_t)h * (uint64_t)W + (uint64_t)w, (uint64_t)j, 1);
                    g_rng_ctr = 0;
                    trace(camorg, dir, objs, Vec3(), Vec3(1, 1, 1), true, start_depth, htable, lw, lh);
                }
            }
        }
    }
    auto t1 = std::chrono::steady_clock::now();
    if (nrays) *nrays = s->probe.n;
    // result gathering, the analogue of main.cpp:252-258 with hp.f in place of hp.flux
    uint64_t k = 0;
    for (size_t b = 0; b < htable.hashtable.size(); b++) {
        for (size_t i = 0; i < htable.hashtable[b].size(); i++) {
            const Hitpoint &q = htable.hashtable[b][i];
            int col = q.w % W, smp = q.w / W;
            size_t pix = (size_t)q.h * W + col;
            if (acc) { acc[3 * pix] += q.f.x; acc[3 * pix + 1] += q.f.y; acc[3 * pix + 2] += q.f.z; }
            if (nhit) nhit[pix]++;
            if (hp && k < hp_cap) {
                double *o = hp + 9 * k;
                o[0] = q.f.x; o[1] = q.f.y; o[2] = q.f.z;
                o[3] = q.pos.x; o[4] = q.pos.y; o[5] = q.pos.z;
                o[6] = q.normal.x; o[7] = q.normal.y; o[8] = q.normal.z;
                hp_pix[k] = ((int64_t)smp << 32) | (int64_t)pix;
            }
            k++;
        }
    }
    if (hp_count) *hp_count = k;
    return std::chrono::duration<double>(t1 - t0).count();
}

// ---- eye pass + photon pass + final gather: render() (main.cpp:169-266) with its constants as arguments -------
// The eye pass fills the reference's own Hashtable (real hash size, so bucket collisions behave as in the
// reference); the photon loop (main.cpp:231-248) runs SERIALLY in photon-index order -- the reference's OpenMP
// region is racy and time-seeded, this is its deterministic single-thread meaning -- with rand() redirected to
// the photon's keyed stream; trace(flag=false) and the samplers are the reference's own code.
// hp_out: 16 doubles per hitpoint (cgrt_testapi.h) in bucket order; image: nrows*W*3, main.cpp:252-258.
int64_t ref_ppm(void *sp, const orc_camera *cam, const orc_grid *g, const orc_photons *ph, double *hp_out,
                uint64_t hp_cap, double *image_out) {
    RefScene *s = (RefScene *)sp;
    std::vector<Object *> objs = s->objs;
    const int W = g->W, H = g->H;
    Vec3 camorg = v3(cam->cam);
    double r = 200.0 / height;  // main.cpp:183 (the global `height` = 768 that trace() also uses, main.cpp:84)
    Hashtable htable = Hashtable(ph->hashsize, r);
    const int start_depth = MAX_DEPTH - g->depth;
    for (int h = g->row0; h < g->row0 + g->nrows; h++) {
        for (int w = 0; w < W; w++) {
            double x = (2.0 * ((double)w / W) - 1) * cam->half_width;
            double y = (2.0 * ((double)h / H) - 1) * cam->half_width * H / W;
            Vec3 dir = (Vec3(x, y, 0) - camorg).normalize();
            Vec3 point_on_focus = dir * ((cam->focus_plane - camorg.z) / dir.z) + camorg;
            for (int j = g->sample0; j < g->sample0 + g->spp; j++) {
                int lw = w + W * (j - g->sample0), lh = h - g->row0;
                g_rng_key = cgrt_key(g->seed, (uint64_t)h * (uint64_t)W + (uint64_t)w, (uint64_t)j, 0);
                g_rng_ctr = 0;
                if (cam->lens_radius > 0) {
                    Vec3 neworg = camorg + uniform_sampling_circle(cam->lens_radius);
                    Vec3 newdir = (point_on_focus - neworg).normalize();
                    trace(neworg, newdir, objs, Vec3(), Vec3(1, 1, 1), true, start_depth, htable, lw, lh);
                } else {
                    trace(camorg, dir, objs, Vec3(), Vec3(1, 1, 1), true, start_depth, htable, lw, lh);
                }
            }
        }
    }
    Vec3 lightorg = v3(ph->light);
    for (int64_t i = 0; i < ph->nphotons; i++) {  // main.cpp:231-248
        g_rng_key = cgrt_key(ph->seed, (uint64_t)i, 0, CGRT_PURPOSE_PHOTON);
        g_rng_ctr = 0;
        double a = uniform_sampling_zeroone() * (2 * ph->jitter) - ph->jitter;
        double b = uniform_sampling_zeroone() * (2 * ph->jitter) - ph->jitter;
        Vec3 disturbance = Vec3(a, 0, b);
        Vec3 dir = uniform_sampling_sphere();
        trace(lightorg + disturbance, dir, objs, Vec3(ph->power, ph->power, ph->power) * (PI * 4.0), Vec3(1, 1, 1), false,
              start_depth, htable, 0, 0);
    }
    int64_t k = 0;
    for (size_t bkt = 0; bkt < htable.hashtable.size(); bkt++) {
        for (size_t i = 0; i < htable.hashtable[bkt].size(); i++) {
            const Hitpoint &q = htable.hashtable[bkt][i];
            const int col = q.w % W, smp = q.w / W;
            const size_t pix = (size_t)q.h * W + col;
            if (image_out) {  // main.cpp:256
                Vec3 c = q.flux * (1.0 / (PI * q.r2 * (double)ph->nphotons * g->spp));
                image_out[3 * pix] += c.x; image_out[3 * pix + 1] += c.y; image_out[3 * pix + 2] += c.z;
            }
            if (hp_out && (uint64_t)k < hp_cap) {
                double *o = hp_out + 16 * k;
                o[0] = (double)pix; o[1] = (double)smp;
                o[2] = q.f.x; o[3] = q.f.y; o[4] = q.f.z;
                o[5] = q.pos.x; o[6] = q.pos.y; o[7] = q.pos.z;
                o[8] = q.normal.x; o[9] = q.normal.y; o[10] = q.normal.z;
                o[11] = q.flux.x; o[12] = q.flux.y; o[13] = q.flux.z;
                o[14] = q.r2; o[15] = (double)q.n;
            }
            k++;
        }
    }
    return k;
}

// ---- tone map + flip: the PNG pixel loop of main(), main.cpp:403-411, with the reference's own gammaCorr (util.h:45-47)
// image: H*W*3 doubles, row 0 = bottom; out: H*W*3 bytes, top row first
void ref_tonemap(const double *image, int W, int H, uint8_t *out) {
    size_t counter = 0;
    for (int i = 0; i < H; i++)
        for (int j = 0; j < W; j++) {
            const double *px = image + ((size_t)(H - i - 1) * W + j) * 3;
            out[3 * counter] = gammaCorr(px[0]);
            out[3 * counter + 1] = gammaCorr(px[1]);
            out[3 * counter + 2] = gammaCorr(px[2]);
            counter++;
        }
}

// ---- texture decoding with the reference's own vendored decoder (stbi_load(..., 3), main.cpp:300): used once, by
// tests/golden/make_assets.py, to turn texture/stone.jpg into the byte asset the reference's main() would see.
// out may be NULL (size query).  Returns 0 on success.
int ref_decode_image(const char *path, int *w, int *h, uint8_t *out, uint64_t cap) {
    int ww = 0, hh = 0, bpp = 0;
    unsigned char *px = stbi_load(path, &ww, &hh, &bpp, 3);
    if (!px) return -1;
    if (w) *w = ww;
    if (h) *h = hh;
    const uint64_t need = (uint64_t)ww * hh * 3;
    if (out && cap >= need) memcpy(out, px, need);
    stbi_image_free(px);
    return 0;
}

}  // extern "C"
