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

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/cgrt.h"
#include "cgrt_build.h"
#include "cgrt_rng.hpp"
#include "cgrt_types.h"

// device code, in dependency order
#include "cgrt_device_math.hpp"
#include "cgrt_grid.hpp"
#include "cgrt_traverse.hpp"
#include "cgrt_bezier.hpp"
#include "cgrt_scene_walk.hpp"
#include "cgrt_eye.hpp"
#include "cgrt_primwalk.hpp"

using namespace cgrt;

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
    // CGRT_GRID_SPLIT_SAMPLES: chunk sums between the two kernels; grown on demand, reused by later launches on this handle
    // (launches on one handle are ordered by the caller: cgrt.h, "Threading")
    mutable void *scratch = nullptr;
    mutable size_t scratch_bytes = 0;
    mutable size_t scratch_refused = 0;  // smallest scratch size this device has refused (0: none yet): not asked for again
    size_t mem_total = 0;                // memory of the scene's device (read at commit; bounds the deferred-value budget)
    // second stream + fork/join events for the light-tile launch that runs beside the full one (created at commit)
    hipStream_t aux_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    cgrt_build_info build_info{};  // filled by cgrt_scene_commit
    std::vector<TreeRec> tree_recs;  // host copy of dev.trees (where a device-built tree's records live)
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

// Every entry point that works on a scene's device switches to it for the duration of the call only: one host thread may
// drive several GPUs (or run under a framework with its own current device) and must find its device unchanged afterwards.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int dev) {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != dev) {
            err = hipSetDevice(dev);
            switched = (err == hipSuccess);
        }
    }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
};
#define ON_DEVICE(dev)          \
    DeviceGuard dev_guard_(dev); \
    if (dev_guard_.err != hipSuccess) return fail(CGRT_ERR_DEVICE, std::string("hipSetDevice: ") + hipGetErrorString(dev_guard_.err))

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

template <class T>
static int upload(cgrt_scene *s, const std::vector<T> &v, const T **out, size_t extra = 0) {  // extra: records of room behind v
    *out = nullptr;
    size_t bytes = (v.size() + extra) * sizeof(T);
    if (bytes == 0) bytes = sizeof(T);  // keep pointers non-null
    void *p = nullptr;
    HIP_TRY(hipMalloc(&p, bytes));
    s->allocs.push_back(p);
    s->device_bytes += (int64_t)bytes;
    if (!v.empty()) HIP_TRY(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = reinterpret_cast<const T *>(p);
    return CGRT_OK;
}

#include "cgrt_devbuild.hpp"

extern "C" {

int cgrt_version(void) { return CGRT_VERSION; }
const char *cgrt_last_error(void) { return g_err.c_str(); }

int cgrt_scene_create(cgrt_scene **out) {
    if (!out) return fail(CGRT_ERR_INVALID, "cgrt_scene_create: null out");
    *out = new (std::nothrow) cgrt_scene();
    if (!*out) return fail(CGRT_ERR_INVALID, "out of memory");
    // CGRT_BUILD=device: the default of cgrt_scene_set_build for scenes created from now on (row f3)
    if (const char *e = std::getenv("CGRT_BUILD")) (*out)->host.build_mode = std::strcmp(e, "device") == 0 ? CGRT_BUILD_DEVICE : CGRT_BUILD_HOST;
    return CGRT_OK;
}

void cgrt_scene_destroy(cgrt_scene *s) {
    if (!s) return;
    if (!s->allocs.empty() || s->scratch || s->aux_stream) {
        DeviceGuard g(s->device);
        if (g.err == hipSuccess) {
            for (void *p : s->allocs) (void)hipFree(p);
            if (s->scratch) (void)hipFree(s->scratch);
            if (s->ev_fork) (void)hipEventDestroy(s->ev_fork);
            if (s->ev_join) (void)hipEventDestroy(s->ev_join);
            if (s->aux_stream) (void)hipStreamDestroy(s->aux_stream);
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
    if ((int)s->host.objs.size() > kMaxObjs) return fail(CGRT_ERR_LIMIT, "more than 2^20 top-level objects");
    return r;
}

int cgrt_scene_set_build(cgrt_scene *s, int mode) {
    NEED_OPEN(s);
    if (mode != CGRT_BUILD_HOST && mode != CGRT_BUILD_DEVICE) return fail(CGRT_ERR_INVALID, "unknown build mode");
    s->host.build_mode = mode;
    return CGRT_OK;
}
int cgrt_scene_build_info(const cgrt_scene *s, cgrt_build_info *out) {
    if (!s || !out) return fail(CGRT_ERR_INVALID, "null argument");
    *out = s->build_info;
    out->mode = s->host.build_mode;
    out->ms_host_build = s->host.host_build_ms;
    return CGRT_OK;
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
    try {  // the tree build allocates and starts threads: nothing may unwind through the C ABI
        return added(s, s->host.add_plane(p, n, sc, refl, transp, tex_id));
    } catch (const std::exception &e) {
        return fail(CGRT_ERR_LIMIT, std::string("plane: ") + e.what());
    }
}
int cgrt_scene_add_mesh_file(cgrt_scene *s, const char *filename, double a, const double b[3], const double sc[3],
                             double refl, double transp, int typeofdata) {
    NEED_OPEN(s);
    if (!filename || !b || !sc) return fail(CGRT_ERR_INVALID, "null argument");
    try {
        return added(s, s->host.add_mesh_file(filename, a, b, sc, refl, transp, typeofdata));
    } catch (const std::exception &e) {
        return fail(CGRT_ERR_LIMIT, std::string("mesh: ") + e.what());
    }
}
int cgrt_scene_add_mesh_triangles(cgrt_scene *s, const double *tri9, int ntri, const double sc[3], double refl,
                                  double transp, int typeofdata) {
    NEED_OPEN(s);
    if (!sc) return fail(CGRT_ERR_INVALID, "null argument");
    try {
        return added(s, s->host.add_mesh_triangles(tri9, ntri, sc, refl, transp, typeofdata));
    } catch (const std::exception &e) {
        return fail(CGRT_ERR_LIMIT, std::string("mesh: ") + e.what());
    }
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
    ON_DEVICE(device);
    s->device = device;
    HostScene &H = s->host;
    const auto t_commit0 = std::chrono::steady_clock::now();
    auto ms_since = [](std::chrono::steady_clock::time_point t0) {
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    };
    // flatten trees
    std::vector<NodeRec> nodes;
    std::vector<TriRec> tris;
    std::vector<TreeRec> trees;
    std::vector<HFieldRec> hfields;
    std::vector<HCellRec> hcells;
    std::vector<HCellY> hcell_y;
    std::vector<OTriRec> otris;
    std::vector<NodeRec> tboxes;
    std::vector<WideNodeRec> wnodes;
    // row f3: records a device build will produce go BEHIND the host-built ones of the same array (room only, no host copy)
    size_t dev_tris = 0, dev_otris = 0, dev_wnodes = 0, dev_hcells = 0;
    for (auto &t : H.trees) {
        TreeRec tr;
        if (t.dev_kind) {  // offsets are assigned once the host parts' sizes are known (second loop below)
            std::memset(&tr, 0, sizeof(tr));
            trees.push_back(tr);
            continue;
        }
        tr.node_begin = (int64_t)nodes.size();
        tr.tri_begin = (int64_t)tris.size();
        // CGRT_TREE=ref (measurement aid): traverse the reference's own inner nodes instead of the SAH hierarchy
        const char *tree_env = std::getenv("CGRT_TREE");
        const bool ref_order = tree_env && std::strcmp(tree_env, "ref") == 0;
        const std::vector<NodeRec> &dev_nodes = ref_order ? t.nodes : t.bvh;
        tr.nnodes = ref_order ? (int32_t)t.nodes.size() : t.bvh_nodes;
        tr.noct = ref_order ? 1 : 8;
        tr.tri_level = (!ref_order && t.tri_level) ? 1 : 0;
        {   // bound on |coordinate| of every box face (the fp32 box test's error term, cgrt_traverse.hpp Ray32): the vertices'
            // largest magnitude, the growth of the boxes and a little more
            double m = 0;
            for (double v : t.tri9) m = std::max(m, std::fabs(v));
            tr.bmax = (float)((m + 2 * kBoxPad) * (1 + 1e-6)) * (1.f + 1e-6f);
        }
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
            hcell_y.insert(hcell_y.end(), t.hcell_y.begin(), t.hcell_y.end());
        }
        tr.tbox_begin = (int64_t)tboxes.size();
        tr.wnode_begin = (int64_t)wnodes.size();
        tr.nwide = tr.tri_level ? (int32_t)t.wide.size() : 0;
        tr.pad2 = 0;
        if (tr.nwide > 0) wnodes.insert(wnodes.end(), t.wide.begin(), t.wide.end());
        tboxes.insert(tboxes.end(), t.tboxes.begin(), t.tboxes.end());
        if (tr.nwide > 0) {  // the device walks the wide form: the one-box-per-node copies stay on the host (cgrt_scene_bvh_dump)
            tr.nnodes = 0;
            tr.noct = 1;
        } else {
            nodes.insert(nodes.end(), dev_nodes.begin(), dev_nodes.end());
        }
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
    // device-built trees: their place behind the host-built records
    int n_dev_trees = 0;
    for (size_t ti = 0; ti < H.trees.size(); ti++) {
        const HostTree &t = H.trees[ti];
        if (!t.dev_kind) continue;
        n_dev_trees++;
        TreeRec &tr = trees[ti];
        tr.hfield = -1;
        tr.noct = 1;
        tr.ntris = (int32_t)t.dev_ntri;
        tr.node_begin = (int64_t)nodes.size();
        tr.tbox_begin = (int64_t)tboxes.size();
        tr.tri_begin = (int64_t)(tris.size() + dev_tris);
        tr.otri_begin = (int64_t)(otris.size() + dev_otris);
        tr.wnode_begin = (int64_t)(wnodes.size() + dev_wnodes);
        dev_tris += (size_t)t.dev_ntri;
        if (t.dev_kind == 1) {  // opaque mesh: triangle-level hierarchy, 4-wide (nwide is known after the build)
            tr.tri_level = 1;
            dev_otris += (size_t)t.dev_ntri;
            dev_wnodes += (size_t)t.dev_ntri;
        } else {  // bump floor: grid cells
            HFieldRec hf = t.hfield;
            hf.cell_begin = (int64_t)(hcells.size() + dev_hcells);
            dev_hcells += (size_t)hf.nx * (size_t)hf.nz;
            tr.hfield = (int32_t)hfields.size();
            hfields.push_back(hf);  // ylo / yhi come from the build
        }
    }
    DeviceScene d{};
    int rc = CGRT_OK;
    double ms_device_build = 0;
    const size_t cover_host = H.cover.size();  // device builds append their cover spheres; a failed commit takes them back
    // a commit that fails part-way leaves nothing behind: the scene stays open and a later commit starts from scratch
    auto all_uploads = [&]() -> int {
        // the large arrays first: the device builds write into the room behind the host-built records
        if ((rc = upload(s, nodes, &d.nodes))) return rc;
        if ((rc = upload(s, tris, &d.tris, dev_tris))) return rc;
        if ((rc = upload(s, texels, &d.texels))) return rc;
        if ((rc = upload(s, hcells, &d.hcells, dev_hcells))) return rc;
        if ((rc = upload(s, hcell_y, &d.hcell_y, dev_hcells))) return rc;
        if ((rc = upload(s, otris, &d.otris, dev_otris))) return rc;
        if ((rc = upload(s, tboxes, &d.tboxes))) return rc;
        if ((rc = upload(s, wnodes, &d.wnodes, dev_wnodes))) return rc;
        if (n_dev_trees > 0) {
            const auto t_build0 = std::chrono::steady_clock::now();
            for (size_t ti = 0; ti < H.trees.size(); ti++) {
                HostTree &t = H.trees[ti];
                TreeRec &tr = trees[ti];
                if (t.dev_kind == 2) {
                    const HostTexture &tx = H.textures[(size_t)t.dev_tex];
                    HFieldRec &hf = hfields[(size_t)tr.hfield];
                    devbuild::BumpResult br;
                    if ((rc = devbuild::build_bump_floor(d.texels + texs[(size_t)t.dev_tex].texel_begin, tx.rows, tx.cols, tx.p, tx.lenx,
                                                         tx.leny, t.dev_plane_y, const_cast<HCellRec *>(d.hcells) + hf.cell_begin,
                                                         const_cast<HCellY *>(d.hcell_y) + hf.cell_begin,
                                                         const_cast<TriRec *>(d.tris) + tr.tri_begin, br)))
                        return rc;
                    hf.ylo = br.ylo;
                    hf.yhi = br.yhi;
                    t.hfield.ylo = br.ylo;
                    t.hfield.yhi = br.yhi;
                } else if (t.dev_kind == 1) {
                    devbuild::MeshResult mr;
                    if ((rc = devbuild::build_mesh_hierarchy(t.tri9.data(), (int)t.dev_ntri, const_cast<TriRec *>(d.tris) + tr.tri_begin,
                                                             const_cast<OTriRec *>(d.otris) + tr.otri_begin,
                                                             const_cast<WideNodeRec *>(d.wnodes) + tr.wnode_begin, mr)))
                        return rc;
                    tr.nwide = mr.nwide;
                    tr.bmax = (float)((mr.bmax + 2 * kBoxPad) * (1 + 1e-6)) * (1.f + 1e-6f);
                    t.dev_nwide = mr.nwide;
                    t.wide_stack = mr.stack_need;
                    t.dev_balanced = mr.balanced;
                    ObjRec &ob = H.objs[(size_t)t.dev_obj];
                    for (int k = 0; k < 3; k++) ob.a[k] = mr.centre[k];
                    ob.s0 = mr.r2;
                    t.dev_cover_at = H.cover.size();
                    H.cover.insert(H.cover.end(), mr.cover.begin(), mr.cover.end());
                }
            }
            ms_device_build = ms_since(t_build0);
        }
        if ((rc = upload(s, H.objs, &d.objs))) return rc;
        if ((rc = upload(s, trees, &d.trees))) return rc;
        if ((rc = upload(s, texs, &d.texs))) return rc;
        if ((rc = upload(s, H.beziers, &d.beziers))) return rc;
        if ((rc = upload(s, H.bez_slabs, &d.bez_slabs))) return rc;
        // CGRT_NO_BEZIER_CULL=1 (measurement / test aid, read at every commit): run every solve of every ray that enters the box
        if (const char *e = std::getenv("CGRT_NO_BEZIER_CULL")) if (*e && *e != '0') d.bez_slabs = nullptr;
        if ((rc = upload(s, hfields, &d.hfields))) return rc;
        if ((rc = upload(s, H.cover, &d.cover))) return rc;
        return CGRT_OK;
    };
    if (all_uploads() != CGRT_OK) {
        for (void *p : s->allocs) (void)hipFree(p);
        s->allocs.clear();
        s->device_bytes = 0;
        s->device = -1;
        H.cover.resize(cover_host);
        return rc;
    }
    d.n_objs = (int32_t)H.objs.size();
    // objects resident in LDS: all of them up to kLdsObjsMax (CGRT_LDS_OBJS, read at every commit, lowers that for measurements
    // and tests of the spill path)
    d.n_lds = std::min(d.n_objs, kLdsObjsMax);
    if (const char *e = std::getenv("CGRT_LDS_OBJS")) d.n_lds = std::max(0, std::min(d.n_lds, std::atoi(e)));
    d.n_trees = (int32_t)trees.size();
    d.n_texs = (int32_t)texs.size();
    d.n_beziers = (int32_t)H.beziers.size();
    d.n_cover = (int32_t)(H.cover.size() / 4);
    d.has_wide = 0;
    for (const TreeRec &tr : trees)
        if (tr.nwide > 0) d.has_wide = 1;
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
    {   // the one opaque mesh whose primary-ray walks may run as their own kernel (cgrt_primwalk.hpp): no other mesh, no Bezier object
        int n_mesh = 0, at = -1;
        for (size_t i = 0; i < H.objs.size(); i++)
            if (H.objs[i].kind == KIND_MESH) { n_mesh++; at = (int)i; }
        d.prim_obj = -1;
        if (n_mesh == 1 && H.beziers.empty() && H.objs[(size_t)at].transp < kEps && H.objs[(size_t)at].tree >= 0 &&
            trees[(size_t)H.objs[(size_t)at].tree].nwide > 0 && at < d.n_lds)
            d.prim_obj = at;
        d.prim_finish = d.prim_obj >= 0 ? 1 : 0;
        for (auto &o : H.objs)
            if (o.kind == KIND_PLANE && o.tree >= 0) d.prim_finish = 0;
    }
    // light tiles (classify_kernel): possible when planes are plain diffuse surfaces and something else is not
    {
        bool planes_plain = true, special = false, plane_trees = false;
        for (auto &o : H.objs) {
            const bool diffuse = o.refl < kEps && o.transp < kEps;
            if (o.kind == KIND_PLANE && !diffuse) planes_plain = false;  // a bump map is fine: a diffuse bumped floor still ends the path
            if (o.kind == KIND_PLANE && o.tree >= 0) plane_trees = true;
            if (o.kind == KIND_MESH || o.kind == KIND_BEZIER || (o.kind == KIND_SPHERE && !diffuse)) special = true;
        }
        d.light_ok = (planes_plain && special && !d.all_spheres) ? 1 : 0;
        d.single_ray = (planes_plain && !special) ? 1 : 0;
        d.light_trees = plane_trees ? 1 : 0;  // the light variant then needs the tree / height-field code (not Bezier, not glass)
        d.light_hf_only = plane_trees ? 1 : 0;
        // the first run of >= 3 axis-aligned planes without a bump tree, with nothing but other planes in front of it, all in the LDS list
        d.prun_begin = d.prun_end = 0;
        static const bool env_no_plane_run = [] { const char *e = std::getenv("CGRT_NO_PLANE_RUN"); return e && *e && *e != '0'; }();
        if (!env_no_plane_run) {
            auto eligible = [&](int j) { return H.objs[(size_t)j].kind == KIND_PLANE && H.objs[(size_t)j].axis >= 0 && H.objs[(size_t)j].tree < 0; };
            int b = 0;  // planes of any sort may stand in front of the run (a bump floor), nothing else
            while (b < d.n_lds && H.objs[(size_t)b].kind == KIND_PLANE && !eligible(b)) b++;
            int j = b;
            while (j < d.n_lds && eligible(j)) j++;
            if (j - b >= 3) {
                d.prun_begin = b;
                d.prun_end = j;
            }
        }
        for (auto &o : H.objs)
            if (o.kind == KIND_PLANE && o.tree >= 0 && !(o.transp < kEps && trees[(size_t)o.tree].hfield >= 0)) d.light_hf_only = 0;
        if (d.light_ok && !s->aux_stream) {
            // the light launch yields to the scheduled one: lowest stream priority
            int prio_least = 0, prio_greatest = 0;
            (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
            const char *pe = std::getenv("CGRT_AUX_PRIORITY");
            const int prio = (pe && std::strcmp(pe, "same") == 0) ? 0 : (pe && std::strcmp(pe, "high") == 0) ? prio_greatest : prio_least;
            if (hipStreamCreateWithPriority(&s->aux_stream, hipStreamNonBlocking, prio) != hipSuccess ||
                hipEventCreateWithFlags(&s->ev_fork, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&s->ev_join, hipEventDisableTiming) != hipSuccess)
                d.light_ok = 0;  // no second stream: render everything with the full variant
        }
    }
    s->dev = d;
    s->committed = true;
    s->tree_recs = trees;
    {
        size_t fr = 0, tot = 0;
        s->mem_total = hipMemGetInfo(&fr, &tot) == hipSuccess ? tot : ((size_t)32 << 30);
    }
    s->build_info.n_device_trees = n_dev_trees;
    s->build_info.ms_device_build = ms_device_build;
    s->build_info.ms_commit = ms_since(t_commit0);
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
        if (t.dev_kind) {  // never built on the host: the counts the reference's tree over these triangles would have
            out->n_triangles += t.dev_ntri;
            out->n_nodes += ref_node_count(t.dev_ntri);
            continue;
        }
        out->n_triangles += (int64_t)t.tris.size();
        out->n_nodes += (int64_t)t.nodes.size();
    }
    bytes += 56 * out->n_nodes + 72 * out->n_triangles;
    for (auto &t : H.textures) bytes += 3 * (int64_t)t.rows * t.cols;
    out->scene_bytes_fp64 = bytes;
    out->device_bytes = s->device_bytes + (int64_t)s->scratch_bytes;  // uploaded scene + the handle's launch scratch
    out->committed = s->committed ? 1 : 0;
    return CGRT_OK;
}

int cgrt_scene_tree_sizes(const cgrt_scene *s, int t, int32_t *nnodes, int32_t *nleaftris, int32_t *ntris) {
    if (!s || t < 0 || t >= (int)s->host.trees.size()) return fail(CGRT_ERR_INVALID, "bad tree index");
    const HostTree &T = s->host.trees[t];
    if (nnodes) *nnodes = (int32_t)T.nodes.size();  // the reference's tree (fingerprints); the device hierarchy is T.bvh
    if (nleaftris) *nleaftris = (int32_t)T.leaf_ids.size();
    if (ntris) *ntris = T.dev_kind ? (int32_t)T.dev_ntri : (int32_t)(T.tri9.size() / 9);  // device-built: no reference tree exists (0 nodes)
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
int cgrt_scene_wide_dump(const cgrt_scene *s, int t, int32_t *nwide, int32_t *stack_need, float *box24, int32_t *ref4) {
    if (!s || t < 0 || t >= (int)s->host.trees.size()) return fail(CGRT_ERR_INVALID, "bad tree index");
    const HostTree &T = s->host.trees[t];
    std::vector<WideNodeRec> from_device;
    if (T.dev_kind == 1 && s->committed && (box24 || ref4)) {  // built on the device (row f3): read the records back
        ON_DEVICE(s->device);
        from_device.resize((size_t)T.dev_nwide);
        if (T.dev_nwide > 0)
            HIP_TRY(hipMemcpy(from_device.data(), s->dev.wnodes + s->tree_recs[(size_t)t].wnode_begin,
                              from_device.size() * sizeof(WideNodeRec), hipMemcpyDeviceToHost));
    }
    const std::vector<WideNodeRec> &wide = T.dev_kind == 1 ? from_device : T.wide;
    if (nwide) *nwide = T.dev_kind == 1 ? T.dev_nwide : (int32_t)T.wide.size();
    if (stack_need) *stack_need = T.wide_stack;
    for (size_t i = 0; i < wide.size(); i++) {
        const WideNodeRec &w = wide[i];
        for (int k = 0; k < 4; k++) {
            if (box24) {
                float *q = box24 + (4 * i + (size_t)k) * 6;
                q[0] = w.lox[k]; q[1] = w.loy[k]; q[2] = w.loz[k];
                q[3] = w.hix[k]; q[4] = w.hiy[k]; q[5] = w.hiz[k];
            }
            if (ref4) ref4[4 * i + (size_t)k] = w.ref[k];
        }
    }
    return CGRT_OK;
}
int cgrt_scene_bvh_order(const cgrt_scene *s, int t, int32_t *tri_level, int32_t *order) {
    if (!s || t < 0 || t >= (int)s->host.trees.size()) return fail(CGRT_ERR_INVALID, "bad tree index");
    const HostTree &T = s->host.trees[t];
    if (tri_level) *tri_level = T.tri_level ? 1 : 0;
    if (order && T.dev_kind == 1 && s->committed) {  // built on the device: k = the triangle's construction index
        ON_DEVICE(s->device);
        std::vector<OTriRec> ot((size_t)T.dev_ntri);
        if (!ot.empty())
            HIP_TRY(hipMemcpy(ot.data(), s->dev.otris + s->tree_recs[(size_t)t].otri_begin, ot.size() * sizeof(OTriRec), hipMemcpyDeviceToHost));
        for (size_t j = 0; j < ot.size(); j++) order[j] = ot[j].k;
        return CGRT_OK;
    }
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

// A launch whose dynamic LDS exceeds the default allowance asks for it first (a CU has 160 KiB; scenes with hundreds of
// top-level objects).  The attribute is per kernel; setting it again is harmless.
#define BIG_LDS(kernel, bytes)                                                                                                 \
    do {                                                                                                                      \
        if ((bytes) > ((size_t)64 << 10)) {                                                                                    \
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(&kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)) != hipSuccess) \
                (void)hipGetLastError(); /* not needed / not supported by this runtime: the launch itself will tell */          \
        }                                                                                                                     \
    } while (0)

// Which instantiation of trace_grid_kernel a launch uses (chosen from the scene's materials and the camera).
struct GridVariant {
    bool trees, bez, dof, glass, sph, stats;
    bool sched;  // the scheduled form (trace_grid_sched_kernel): heavy-tile unit queue in front of the tile workgroups
    int nt;  // threads per workgroup: 256 (32x8-pixel tiles) or 64 (Bezier scenes: one-wave workgroups on 16x4 tiles)
};
static GridVariant grid_variant(const cgrt_scene *s, const cgrt_camera *cam, const cgrt_grid *grid) {
    GridVariant v;
    v.bez = s->dev.has_bezier != 0;
    v.trees = s->dev.has_mesh != 0 || v.bez;  // Bezier scenes share the tree-capable variants
    v.dof = cam->lens_radius > 0;
    v.glass = s->dev.has_glass != 0 && grid->max_depth > 1;
    v.sph = !v.trees && s->dev.all_spheres != 0;
    v.stats = (grid->flags & CGRT_GRID_STATS) != 0 && s->dev.has_mesh != 0 && !v.bez;
    v.nt = v.bez ? 64 : kThreads;
    v.sched = grid->spp >= 4 && !(grid->flags & CGRT_GRID_NO_REORDER) && (!((s->dev.has_mesh == 0 && s->dev.has_bezier == 0) || s->dev.single_ray != 0) || (grid->flags & CGRT_GRID_FORCE_REORDER)) &&  // (CGRT_FORCE_REORDER, a measurement aid, is not reflected here)
              ((size_t)((grid->width + kWaveTileW - 1) / kWaveTileW) * ((grid->rows + kWaveTileH - 1) / kWaveTileH)) > 1;
    return v;
}

static bool glass_possible(const cgrt_scene *s, const cgrt_grid *grid) { return s->dev.has_glass != 0 && grid->max_depth > 1; }

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

int cgrt_trace_grid_variant(const cgrt_scene *s, const cgrt_camera *cam, const cgrt_grid *grid, char *name, size_t cap) {
    int rc = check_grid(s, cam, grid);
    if (rc) return rc;
    if (!name || cap == 0) return fail(CGRT_ERR_INVALID, "null name buffer");
    const GridVariant v = grid_variant(s, cam, grid);
    if (s->dev.n_objs > s->dev.n_lds) {
        if (s->dev.all_spheres)
            std::snprintf(name, cap, "trace_grid_kernel<TREES=0,BEZ=0,DOF=%d,GLASS=%d,SPH=1,STATS=0,HPS=0,NT=256,SPILL=1>", (int)v.dof, (int)v.glass);
        else
            std::snprintf(name, cap, "trace_grid_kernel<TREES=1,BEZ=1,DOF=%d,GLASS=1,SPH=0,STATS=0,HPS=0,NT=256,SPILL=1>", (int)v.dof);
        return CGRT_OK;
    }
    std::snprintf(name, cap, "trace_grid_%skernel<TREES=%d,BEZ=%d,DOF=%d,GLASS=%d,SPH=%d,STATS=%d,%sNT=%d>",
                  v.sched ? "sched_" : "", (int)v.trees, (int)v.bez, (int)v.dof, (int)v.glass, (int)v.sph, (int)v.stats,
                  v.sched ? "" : "HPS=0,", v.nt);
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
    // Split a tile's samples over several workgroups (CGRT_GRID_SPLIT_SAMPLES, opt-in for every scene since the cost
    // scheduler balances Bezier scenes too): chunks of >= 16 samples, at most 16 chunks, at most 4 GiB of chunk sums.
    g.chunks = 1;
    g.chunk_spp = grid->spp;
    g.partial = nullptr;
    g.partial_nhit = nullptr;
    g.timeline = nullptr;
    const size_t npx_all = (size_t)grid->rows * grid->width;
    const bool spill = s->dev.n_objs > s->dev.n_lds;  // image order, whole tiles (see the SPILL launch below)
    if ((grid->flags & CGRT_GRID_SPLIT_SAMPLES) && grid->spp >= 32 && !spill) {
        int chunks = grid->spp / 16;
        if (chunks > 16) chunks = 16;
        while (chunks > 1 && (size_t)chunks * npx_all * 28 > ((size_t)4 << 30)) chunks--;
        if (chunks > 1) {
            g.chunk_spp = (grid->spp + chunks - 1) / chunks;
            g.chunks = (grid->spp + g.chunk_spp - 1) / g.chunk_spp;
        }
    }
    // Bezier scenes run one-wave workgroups on 16x4 tiles (TileGeom<64>): waves over the vase outlast their neighbours ~100x
    const bool one_wave = s->dev.has_bezier != 0;
    const int waves_per_block = one_wave ? 1 : kThreads / 64;
    const int tile_blocks = one_wave ? tile_grid_blocks(g.W, g.rows, false, TileGeom<64>::W, TileGeom<64>::H)
                                     : tile_grid_blocks(g.W, g.rows, g.xcd_tiles != 0);
    const dim3 natural_dim((unsigned)(tile_blocks * g.chunks)), block(one_wave ? 64 : kThreads);
    size_t lds = obj_list_lds(s->dev, waves_per_block);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    auto *cnt = reinterpret_cast<unsigned long long *>(counters);
    // launch on the scene's device whatever the caller's current device is (one host thread may drive several GPUs)
    ON_DEVICE(s->device);
    const GridVariant gv = grid_variant(s, cam, grid);
    const bool trees = s->dev.has_mesh != 0, dof = gv.dof, bez = gv.bez, glass = gv.glass, stats = gv.stats;
    // Cost-aware scheduling ("classify -> probe -> plan -> render -> ordered sum"; DESIGN.md sections 4.6-4.7).  A frame's cost
    // is concentrated in a few tiles (a glass mesh: one 32x8 tile ran 38 of the frame's 46 ms while four of the eight XCDs
    // were idle after 6 ms), the hardware hands out workgroups in block-index order, and a tile is bound to one wave per
    // 64 pixels.  So: (0) classify_kernel marks the LIGHT wave tiles (no primary ray can reach a mesh, a Bezier object or a
    // reflecting / refracting sphere), which a lighter kernel variant renders on a second stream; (1) the full variant traces
    // ONE sample of every other wave tile (16x4 pixels) storing nothing but the shader-clock ticks it took; (2) plan_kernel
    // marks as HEAVY the tiles that alone would hold a wave slot for more than 1/heavy_div of the frame's ideal duration,
    // orders them heaviest first, and lists the remaining tiles with something to render, costliest first; (3) the render
    // launch serves the heavy tiles through a queue of (pixel, sample) units that any lane of any wave may take, and the
    // other tiles through a queue of tiles, with persistent workgroups serving both (GridParams); (4) deferred_sum_kernel
    // adds the heavy tiles' Hitpoint values in the reference's order.  The image does not depend on any of this -- every
    // Hitpoint value is added to its pixel in sample order, emission order within a sample --: identical bits and counters;
    // the probe costs 1/spp of the frame and the whole scheme is skipped below 4 samples per pixel or on request
    // (CGRT_GRID_NO_REORDER).
    const int wtiles_x = (g.W + kWaveTileW - 1) / kWaveTileW, wtiles_y = (g.rows + kWaveTileH - 1) / kWaveTileH;
    const size_t n_wt = (size_t)wtiles_x * wtiles_y;
    static const bool env_force_reorder = [] { const char *e = std::getenv("CGRT_FORCE_REORDER"); return e && *e && *e != '0'; }();
    // Scenes of spheres and plain planes are left in image order: without a tree or a Newton solve behind a ray, a tile's cost
    // varies only with the size of its ray trees (<= 31 rays per sample), the per-lane sample loop already keeps 97 % of the
    // lanes busy, and measured on C2 the unit queue costs 13 % more VALU instructions (at 93 % VALU busy) and 0.9 GB of deferred
    // values per frame for a gain within the noise (4.0-4.2 ms either way); a room of five planes, 4096x4096 spp 4: 2.2 ms
    // scheduled (probe and plan for nothing), 1.0 ms in image order.
    // The same holds when every ray tree is a single ray (diffuse planes, bump-mapped or not, and diffuse spheres): a room with the
    // stone floor, 8192 x 512 rows at spp 16: floor band 7.7 ms scheduled (every tile of a uniform frame counts as heavy), 5.4 ms
    // in image order; wall band 4.0 and 1.7 ms.
    const bool plain_scene = (s->dev.has_mesh == 0 && s->dev.has_bezier == 0) || s->dev.single_ray != 0;  // has_mesh: any tree, a bump floor's included
    const bool reorder = !spill && grid->spp >= 4 && !(grid->flags & CGRT_GRID_NO_REORDER) && n_wt > 1 && n_wt < (1u << 30) &&
                         (!plain_scene || (grid->flags & CGRT_GRID_FORCE_REORDER) || env_force_reorder);
    static const long long env_defer_bytes = [] { const char *e = std::getenv("CGRT_DEFER_BYTES"); return e ? std::atoll(e) : 0ll; }();
    static const int env_heavy_div = [] { const char *e = std::getenv("CGRT_HEAVY_DIV"); return e ? std::atoi(e) : 0; }();
    static const int env_units = [] { const char *e = std::getenv("CGRT_UNITS_PER_ITEM"); return e ? std::atoi(e) : 0; }();
    // deferred Hitpoint values: up to 12 GiB, at most an eighth of THIS scene's device (MI355X: 288 GB; read at commit -- a
    // process may drive devices of different sizes); allocated once per scene handle, as large as the biggest launch needed it
    const size_t defer_budget = env_defer_bytes > 0 ? (size_t)env_defer_bytes : std::min((size_t)12 << 30, s->mem_total / 8);
    const int heavy_div = env_heavy_div > 0 ? env_heavy_div : 32;
    const int units_per_item = env_units > 0 ? ((env_units + 63) / 64) * 64 : 256;
    const int maxhp = glass_possible(s, grid) ? 16 : 1;  // Hitpoints per sample: a mirror chain ends in one, a glass tree of depth 5 in <= 16
    const size_t tile_vals = (size_t)grid->spp * 64 * (size_t)maxhp * 3 * sizeof(double), tile_cnt = (size_t)grid->spp * 64;
    const size_t tile_pconst = 7 * 64 * sizeof(double);
    // primary-ray mesh hits of the heavy tiles' units (cgrt_primwalk.hpp): a double and an int per unit
    static const bool env_no_primwalk = [] { const char *e = std::getenv("CGRT_NO_PRIMWALK"); return e && *e && *e != '0'; }();
    const bool use_prim = s->dev.prim_obj >= 0 && !env_no_primwalk && !(grid->flags & CGRT_GRID_STATS);
    const size_t tile_prim = use_prim ? (size_t)grid->spp * 64 * (sizeof(double) + sizeof(int32_t)) : 0;
    size_t kmax = 0;
    const size_t sched_pad = 8;
    size_t sched_bytes = 0, defer_bytes = 0;
    if (reorder) {
        sched_bytes = (4 * (n_wt + sched_pad) + 64) * sizeof(uint32_t) + ((n_wt + 255) & ~(size_t)255);  // cost, order, hidx, border, plan, light
        sched_bytes = (sched_bytes + 255) & ~(size_t)255;
        kmax = defer_budget / (tile_vals + ((tile_cnt + 7) & ~(size_t)7) + tile_pconst + tile_prim);
        if (kmax > n_wt) kmax = n_wt;
        defer_bytes = kmax * (tile_vals + ((tile_cnt + 7) & ~(size_t)7) + tile_pconst + tile_prim) + 256;
    }
    size_t chunk_bytes = 0;
    if (g.chunks > 1) chunk_bytes = (size_t)g.chunks * npx_all * (3 * sizeof(double) + sizeof(uint32_t));
    const size_t chunk_bytes_al = (chunk_bytes + 255) & ~(size_t)255;
    const size_t per_tile = tile_vals + ((tile_cnt + 7) & ~(size_t)7) + tile_pconst + tile_prim;
    size_t scratch_need = chunk_bytes_al + sched_bytes + defer_bytes;
    // A size this device has refused before is not asked for again (every attempt is a synchronous hipFree plus failing
    // hipMallocs): the deferred buffer shrinks -- fewer heavy tiles, same image -- until the need lies below it.
    while (s->scratch_refused && scratch_need >= s->scratch_refused && kmax > 0) {
        kmax /= 2;
        defer_bytes = kmax ? kmax * per_tile + 256 : 0;
        scratch_need = chunk_bytes_al + sched_bytes + defer_bytes;
    }
    if (scratch_need > 0 && s->scratch_bytes < scratch_need) {
        if (s->scratch) (void)hipFree(s->scratch);
        s->scratch = nullptr;
        s->scratch_bytes = 0;
        // a device short of memory gets a smaller deferred buffer before it gets an error
        while (hipMalloc(&s->scratch, scratch_need) != hipSuccess) {
            (void)hipGetLastError();
            s->scratch = nullptr;
            if (!s->scratch_refused || scratch_need < s->scratch_refused) s->scratch_refused = scratch_need;
            if (kmax == 0) return fail(CGRT_ERR_DEVICE, "cannot allocate launch scratch (chunk sums / schedule / deferred Hitpoint values)");
            kmax /= 2;
            defer_bytes = kmax ? kmax * per_tile + 256 : 0;
            scratch_need = chunk_bytes_al + sched_bytes + defer_bytes;
        }
        s->scratch_bytes = scratch_need;
    }
    if (g.chunks > 1) {
        g.partial = reinterpret_cast<double *>(s->scratch);
        g.partial_nhit = nhit ? reinterpret_cast<uint32_t *>(g.partial + (size_t)g.chunks * npx_all * 3) : nullptr;
    }
    g.light = nullptr;
    g.light_mode = 0;
    g.pad_light_ = 0;
    g.order = nullptr;
    g.border = nullptr;
    g.cost = nullptr;
    g.hidx = nullptr;
    g.plan = nullptr;
    g.dvals = nullptr;
    g.dcnt = nullptr;
    g.pconst = nullptr;
    g.probe = 0;
    g.heavy_blocks = 0;
    g.items_per_tile = 1;
    g.units_per_item = units_per_item;
    g.maxhp = maxhp;
    g.prim_len = nullptr;
    g.prim_tri = nullptr;
    g.prim_obj = -1;
    g.prim_done = 0;
    static const int env_pw_refill = [] { const char *e = std::getenv("CGRT_PW_REFILL"); return e ? std::atoi(e) : 0; }();
    static const int env_pw_rounds = [] { const char *e = std::getenv("CGRT_PW_ROUNDS"); return e ? std::atoi(e) : 0; }();
    g.pw_refill = env_pw_refill > 0 ? env_pw_refill : 16;
    g.pw_rounds = env_pw_rounds > 0 ? env_pw_rounds : 8;
    if (one_wave) lds += (glass ? TileGeom<64>::stack_bytes : 0) + sizeof(BezLds);
    else lds += glass ? kStackBytes : 0;
    if (trees && s->dev.cached_tree >= 0) lds += (size_t)s->dev.cached_nodes * sizeof(NodeRec);
    if (trees && !glass && !bez && s->dev.has_wide) lds += (size_t)kThreads * kWideLdsDepth * sizeof(uint2);  // wide walk's stack
    static const int env_lds_pad = [] { const char *e = std::getenv("CGRT_LDS_PAD"); return e ? std::atoi(e) : 0; }();
    lds += (size_t)env_lds_pad;
    auto launch_mode = [&](auto sched_tag, const GridParams &gp, dim3 gd, float *rgb_, uint32_t *nhit_, unsigned long long *cnt_) {
        constexpr bool SCHED = decltype(sched_tag)::value;
#define LAUNCH(T, B, D, G, P, S)                                                                                              \
    do {                                                                                                                      \
        if (SCHED) { BIG_LDS((trace_grid_sched_kernel<T, B, D, G, P, S, 256>), lds); hipLaunchKernelGGL((trace_grid_sched_kernel<T, B, D, G, P, S, 256>), gd, block, lds, st, s->dev, gp, rgb_, nhit_, cnt_); } \
        else { BIG_LDS((trace_grid_kernel<T, B, D, G, P, S>), lds); hipLaunchKernelGGL((trace_grid_kernel<T, B, D, G, P, S>), gd, block, lds, st, s->dev, gp, rgb_, nhit_, cnt_); }     \
    } while (0)
#define LAUNCH_DG(T, B, P, S)                                      \
    do {                                                           \
        if (dof) { if (glass) LAUNCH(T, B, true, true, P, S); else LAUNCH(T, B, true, false, P, S); }   \
        else     { if (glass) LAUNCH(T, B, false, true, P, S); else LAUNCH(T, B, false, false, P, S); } \
    } while (0)
        if (bez) {  // Bezier scenes share the tree-capable variants (the tree code is skipped when there is no tree)
#define LAUNCH1(D, G)                                                                                                          \
    do {                                                                                                                       \
        if (SCHED) { BIG_LDS((trace_grid_sched_kernel<true, true, D, G, false, false, 64>), lds); hipLaunchKernelGGL((trace_grid_sched_kernel<true, true, D, G, false, false, 64>), gd, block, lds, st, s->dev, gp, rgb_, nhit_, cnt_); } \
        else { BIG_LDS((trace_grid_kernel<true, true, D, G, false, false, false, 64>), lds); hipLaunchKernelGGL((trace_grid_kernel<true, true, D, G, false, false, false, 64>), gd, block, lds, st, s->dev, gp, rgb_, nhit_, cnt_); }    \
    } while (0)
            if (dof) { if (glass) LAUNCH1(true, true); else LAUNCH1(true, false); }
            else     { if (glass) LAUNCH1(false, true); else LAUNCH1(false, false); }
#undef LAUNCH1
        } else if (trees) {
            if (stats) LAUNCH_DG(true, false, false, true); else LAUNCH_DG(true, false, false, false);
        } else if (s->dev.all_spheres) {
            LAUNCH_DG(false, false, true, false);
        } else {
            LAUNCH_DG(false, false, false, false);
        }
#undef LAUNCH_DG
#undef LAUNCH
    };
    auto launch = [&](const GridParams &gp, dim3 gd, float *rgb_, uint32_t *nhit_, unsigned long long *cnt_) {
        launch_mode(std::false_type{}, gp, gd, rgb_, nhit_, cnt_);
    };
    if (spill) {
        // More top-level objects than the LDS list holds (kLdsObjsMax): the SPILL variants, which read the others from the
        // uploaded array, in image order -- the sphere loop for sphere-only scenes, else the most general body (trees, Bezier,
        // pending rays), as the Hitpoint capture does.  Such scenes are bound by their object loop, not by tile imbalance.
        const dim3 gd((unsigned)tile_grid_blocks(g.W, g.rows, false)), blk(kThreads);
        GridParams gs = g;
        gs.xcd_tiles = 0;
        if (s->dev.all_spheres) {
            const size_t l = obj_list_lds(s->dev, kThreads / 64) + (glass ? kStackBytes : 0);
#define SPILL_SPH(D, G)                                                                                                       \
    do {                                                                                                                      \
        BIG_LDS((trace_grid_kernel<false, false, D, G, true, false, false, 256, true>), l);                                    \
        hipLaunchKernelGGL((trace_grid_kernel<false, false, D, G, true, false, false, 256, true>), gd, blk, l, st, s->dev, gs, rgb, nhit, cnt); \
    } while (0)
            if (dof) { if (glass) SPILL_SPH(true, true); else SPILL_SPH(true, false); }
            else     { if (glass) SPILL_SPH(false, true); else SPILL_SPH(false, false); }
#undef SPILL_SPH
        } else {
            const size_t l = obj_list_lds(s->dev, kThreads / 64) + kStackBytes + (kThreads / 64) * sizeof(BezLds) +
                             (s->dev.cached_tree >= 0 ? (size_t)s->dev.cached_nodes * sizeof(NodeRec) : 0);
#define SPILL_GEN(D)                                                                                                          \
    do {                                                                                                                      \
        BIG_LDS((trace_grid_kernel<true, true, D, true, false, false, false, 256, true>), l);                                  \
        hipLaunchKernelGGL((trace_grid_kernel<true, true, D, true, false, false, false, 256, true>), gd, blk, l, st, s->dev, gs, rgb, nhit, cnt); \
    } while (0)
            if (dof) SPILL_GEN(true); else SPILL_GEN(false);
#undef SPILL_GEN
        }
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return fail(CGRT_ERR_DEVICE, std::string("kernel launch: ") + hipGetErrorString(e));
        return CGRT_OK;
    }
    dim3 grid_dim = natural_dim;
    if (reorder && kmax > 0) {
        unsigned char *base = reinterpret_cast<unsigned char *>(s->scratch) + chunk_bytes_al;
        uint32_t *sb = reinterpret_cast<uint32_t *>(base);
        const size_t np = n_wt + sched_pad;
        uint32_t *cost = sb, *order = sb + np;
        int32_t *hidx = reinterpret_cast<int32_t *>(sb + 2 * np);
        uint32_t *border = sb + 3 * np;
        uint32_t *plan = sb + 4 * np;
        unsigned char *light = reinterpret_cast<unsigned char *>(sb + 4 * np + 64);
        unsigned char *dbase = base + sched_bytes;
        const bool split_light = s->dev.light_ok != 0 && g.chunks == 1 && !stats;
        if (split_light)
            hipLaunchKernelGGL(classify_kernel, dim3((unsigned)((n_wt + 255) / 256)), dim3(256), 0, st, s->dev, g, light, (int)n_wt);
        GridParams gp = g;  // the probe: this launch's first sample, natural order, one workgroup per tile, nothing stored
        gp.spp = 1;
        gp.chunks = 1;
        gp.chunk_spp = 1;
        gp.partial = nullptr;
        gp.partial_nhit = nullptr;
        gp.probe = 1;
        gp.cost = cost;
        gp.timeline = nullptr;
        if (split_light) {  // light tiles are not probed (plan_kernel counts them as zero); their cost entries stay unwritten
            gp.light = light;
            gp.light_mode = 0;
        }
        launch(gp, dim3((unsigned)tile_blocks), nullptr, nullptr, nullptr);
        if (split_light) {
            // The light tiles: the variant without tree / Bezier / pending-ray code on the second stream, beside everything that
            // follows here.  It needs nothing but the classification and starts as soon as the probe is through, beside the
            // plan (one workgroup, 0.1-0.3 ms).  Started before the probe it perturbs the measured costs and delays the
            // probe's workgroups: C3 14.4 -> 14.1 ms but C4 (spp 64) 33.6 -> 35.3 ms.
            GridParams gl = g;
            gl.light = light;
            gl.light_mode = 1;
            gl.timeline = nullptr;
            HIP_TRY(hipEventRecord(s->ev_fork, st));
            HIP_TRY(hipStreamWaitEvent(s->aux_stream, s->ev_fork, 0));
            size_t lds_light = obj_list_lds(s->dev, kThreads / 64);
            const bool ltrees = s->dev.light_trees != 0;  // bump-mapped planes: the tree-capable variant (same LDS carve-up as the main launch's)
            if (ltrees && s->dev.cached_tree >= 0) lds_light += (size_t)s->dev.cached_nodes * sizeof(NodeRec);
            if (ltrees && s->dev.has_wide) lds_light += (size_t)kThreads * kWideLdsDepth * sizeof(uint2);
            const dim3 gd_light((unsigned)tile_grid_blocks(g.W, g.rows, false));
            gl.xcd_tiles = 0;
#define LIGHT(T, D)                                                                                                           \
    do {                                                                                                                      \
        BIG_LDS((trace_grid_kernel<T, false, D, false, false, false>), lds_light);                                             \
        hipLaunchKernelGGL((trace_grid_kernel<T, false, D, false, false, false>), gd_light, dim3(kThreads), lds_light, s->aux_stream, \
                           s->dev, gl, rgb, nhit, cnt);                                                                       \
    } while (0)
#define LIGHT_HF(D)                                                                                                           \
    do {                                                                                                                      \
        BIG_LDS((trace_grid_kernel<true, false, D, false, false, false, false, 256, false, true>), lds_hf);                    \
        hipLaunchKernelGGL((trace_grid_kernel<true, false, D, false, false, false, false, 256, false, true>), gd_light, dim3(kThreads), lds_hf, \
                           s->aux_stream, s->dev, gl, rgb, nhit, cnt);                                                         \
    } while (0)
            const size_t lds_hf = obj_list_lds(s->dev, kThreads / 64);  // no node cache, no walk stack
            static const bool env_no_hfonly = [] { const char *e = std::getenv("CGRT_NO_HFONLY"); return e && *e && *e != '0'; }();
            if (ltrees && s->dev.light_hf_only && !env_no_hfonly) { if (dof) LIGHT_HF(true); else LIGHT_HF(false); }
            else if (ltrees) { if (dof) LIGHT(true, true); else LIGHT(true, false); }
            else        { if (dof) LIGHT(false, true); else LIGHT(false, false); }
#undef LIGHT_HF
#undef LIGHT
            HIP_TRY(hipEventRecord(s->ev_join, s->aux_stream));
        }
        // heavy: cost x spp > (total cost x spp / wave slots) / heavy_div
        int n_cu = 256;
        (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, s->device);
        const int wave_slots = n_cu * 4 * (one_wave ? kBezWaves : (trees ? kSchedTreeWaves : 4));
        // tiles through a queue too (GridParams::border) unless their samples are split over workgroups or the workgroups are
        // single waves (trace_grid_sched_kernel)
        static const bool env_no_tile_queue = [] { const char *e = std::getenv("CGRT_NO_TILE_QUEUE"); return e && *e && *e != '0'; }();
        const bool tile_queue = g.chunks == 1 && !one_wave && !env_no_tile_queue;
        hipLaunchKernelGGL(plan_kernel, dim3(1), dim3(1024), 0, st, cost, split_light ? light : nullptr, (int)n_wt, (unsigned)kmax,
                           (unsigned long long)wave_slots * (unsigned long long)heavy_div, plan, order, hidx,
                           tile_queue ? border : nullptr, wtiles_x, wtiles_y, one_wave ? 1 : kTileW / kWaveTileW,
                           one_wave ? 1 : kTileH / kWaveTileH);
        if (split_light) g.light = light;
        g.order = order;
        g.hidx = hidx;
        g.plan = plan;
        g.dvals = reinterpret_cast<double *>(dbase);
        g.dcnt = dbase + kmax * tile_vals;
        g.pconst = reinterpret_cast<double *>(dbase + kmax * (tile_vals + ((tile_cnt + 7) & ~(size_t)7)));
        if (use_prim) {
            unsigned char *pb = dbase + kmax * (tile_vals + ((tile_cnt + 7) & ~(size_t)7) + tile_pconst);
            g.prim_len = reinterpret_cast<const double *>(pb);
            g.prim_tri = reinterpret_cast<const int32_t *>(pb + kmax * (size_t)grid->spp * 64 * sizeof(double));
            g.prim_obj = s->dev.prim_obj;
            static const bool env_no_finish = [] { const char *e = std::getenv("CGRT_PW_NO_FINISH"); return e && *e && *e != '0'; }();
            g.prim_done = (s->dev.prim_finish && !env_no_finish) ? 1 : 0;
        }
        g.items_per_tile = (int)(((size_t)grid->spp * 64 + units_per_item - 1) / units_per_item);
        // enough heavy workgroups to fill the chip once: they loop over the item queue until it is empty
        size_t hb = (kmax * (size_t)g.items_per_tile + waves_per_block - 1) / waves_per_block;
        const size_t fill = (size_t)wave_slots / waves_per_block;
        if (hb > fill) hb = fill;
        g.heavy_blocks = (int)hb;
        if (tile_queue) {
            g.border = border;
            grid_dim = dim3((unsigned)fill);  // tile workgroups: a chip's worth, each loops over the tile queue
        }
    }
    // development aid: CGRT_TIMELINE_FILE=path makes this launch synchronous and dumps, per workgroup, when and where it ran
    DevBuf timeline;
    const char *timeline_file = std::getenv("CGRT_TIMELINE_FILE");
    if (timeline_file && *timeline_file) {
        HIP_TRY(timeline.alloc(((size_t)grid_dim.x + g.heavy_blocks) * 32));
        HIP_TRY(hipMemsetAsync(timeline.p, 0, ((size_t)grid_dim.x + g.heavy_blocks) * 32, st));
        g.timeline = timeline.as<unsigned long long>();
    }
    if (g.heavy_blocks > 0) {  // heavy workgroups in front, the tile workgroups behind them, one launch
        hipLaunchKernelGGL(pixel_const_kernel, dim3((unsigned)kmax), dim3(64), 0, st, g);
        if (g.prim_len) {
            // the heavy units' primary rays against the mesh, as a walk-only kernel with lane refill (cgrt_primwalk.hpp): a
            // chip's worth of persistent workgroups at 4 waves per SIMD
            int n_cu2 = 256;
            (void)hipDeviceGetAttribute(&n_cu2, hipDeviceAttributeMultiprocessorCount, s->device);
            PrimWalkArgs pw;
            pw.len = const_cast<double *>(g.prim_len);
            pw.tri = const_cast<int32_t *>(g.prim_tri);
            pw.obj = s->dev.prim_obj;
            pw.tree = s->host.objs[(size_t)s->dev.prim_obj].tree;
            pw.finish = g.prim_done;
            pw.pad_ = 0;
            pw.counters = cnt;
            const size_t lds_pw = (size_t)(pw.finish ? s->dev.n_objs : pw.obj + 1) * sizeof(ObjRec) + (size_t)kThreads * kWideLdsDepth * sizeof(uint2);
            if (dof) hipLaunchKernelGGL(primary_walk_kernel<true>, dim3((unsigned)n_cu2 * 4), dim3(kThreads), lds_pw, st, s->dev, g, pw);
            else hipLaunchKernelGGL(primary_walk_kernel<false>, dim3((unsigned)n_cu2 * 4), dim3(kThreads), lds_pw, st, s->dev, g, pw);
        }
        launch_mode(std::true_type{}, g, dim3((unsigned)g.heavy_blocks + grid_dim.x), rgb, nhit, cnt);
    } else {
        launch(g, grid_dim, rgb, nhit, cnt);
    }
    if (g.chunks > 1)
        hipLaunchKernelGGL(finalize_chunks_kernel, dim3((unsigned)((npx_all + 255) / 256)), dim3(256), 0, st, g, rgb, nhit);
    if (g.heavy_blocks > 0) hipLaunchKernelGGL(deferred_sum_kernel, dim3((unsigned)kmax), dim3(64), 0, st, g, rgb, nhit);
    if (g.light) HIP_TRY(hipStreamWaitEvent(st, s->ev_join, 0));  // the caller's stream continues when both launches are done
    const hipError_t launch_err = hipGetLastError();
    if (g.timeline && launch_err == hipSuccess) {
        std::vector<unsigned long long> tl(((size_t)grid_dim.x + g.heavy_blocks) * 4);
        HIP_TRY(hipStreamSynchronize(st));
        HIP_TRY(hipMemcpy(tl.data(), timeline.p, tl.size() * 8, hipMemcpyDeviceToHost));
        if (FILE *f = std::fopen(timeline_file, "wb")) {
            const unsigned long long head[4] = {(unsigned long long)grid_dim.x + g.heavy_blocks, block.x, (unsigned long long)g.chunks, (unsigned long long)g.xcd_tiles};
            std::fwrite(head, 8, 4, f);
            std::fwrite(tl.data(), 8, tl.size(), f);
            std::fclose(f);
        }
    }
    if (launch_err != hipSuccess) return fail(CGRT_ERR_DEVICE, std::string("kernel launch: ") + hipGetErrorString(launch_err));
    // development aid: CGRT_PLAN_DUMP=1 makes the launch synchronous and prints what the planner decided
    static const bool env_plan_dump = [] { const char *e = std::getenv("CGRT_PLAN_DUMP"); return e && *e && *e != '0'; }();
    if (env_plan_dump && g.plan) {
        uint32_t pl[8] = {0};
        HIP_TRY(hipStreamSynchronize(st));
        HIP_TRY(hipMemcpy(pl, g.plan, sizeof(pl), hipMemcpyDeviceToHost));
        std::fprintf(stderr, "cgrt plan: wave tiles %zu, heavy %u (capacity %zu), cost threshold %u, tile-queue entries %u, items per tile %d, prim walk %s\n",
                     n_wt, pl[0], kmax, pl[1], pl[3], g.items_per_tile, g.prim_len ? "on" : "off");
    }
    return CGRT_OK;
}

}  // extern "C"


// Row e: global row h of the frame = local row (stripe / nshares) * S + h % S of share stripe % nshares (the inverse of
// global_row()); one workgroup column per 256 floats of a row, blockIdx.y = the row.  Rows of absent shares are zeroed.
__global__ void unpermute_stripes_kernel(const float *__restrict__ shares, int n_present, int nshares, int row_floats, int S,
                                         int rows_local, float *__restrict__ frame) {
    const int h = blockIdx.y, x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= row_floats) return;
    const int stripe = h / S, share = stripe % nshares, j = (stripe / nshares) * S + h % S;
    float v = 0.f;
    if (share < n_present && j < rows_local) v = shares[((size_t)share * rows_local + (size_t)j) * row_floats + x];
    frame[(size_t)h * row_floats + x] = v;
}

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
    g.chunks = 1; g.chunk_spp = grid->spp; g.partial = nullptr; g.partial_nhit = nullptr;  // capture keeps one workgroup per tile
    g.timeline = nullptr;
    g.light = nullptr; g.light_mode = 0; g.pad_light_ = 0; g.order = nullptr; g.border = nullptr; g.cost = nullptr; g.hidx = nullptr; g.plan = nullptr; g.dvals = nullptr; g.dcnt = nullptr; g.pconst = nullptr;
    g.probe = 0; g.heavy_blocks = 0; g.items_per_tile = 1; g.units_per_item = 256; g.maxhp = 16;
    g.prim_len = nullptr; g.prim_tri = nullptr; g.prim_obj = -1; g.prim_done = 0; g.pw_refill = 16; g.pw_rounds = 8;
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
    const size_t lds = obj_list_lds(s->dev, kThreads / 64) + kStackBytes + (kThreads / 64) * sizeof(BezLds) +
                       (s->dev.cached_tree >= 0 ? (size_t)s->dev.cached_nodes * sizeof(NodeRec) : 0);
    HitpointSink sink{d_rec, d_cnt, (unsigned long long)cap};
    // the most general variant serves every scene; capture is a verification / hand-off path, not the hot path
#define CAPTURE(D, SP)                                                                                                         \
    do {                                                                                                                      \
        BIG_LDS((trace_grid_kernel<true, true, D, true, false, false, true, 256, SP>), lds);                                   \
        hipLaunchKernelGGL((trace_grid_kernel<true, true, D, true, false, false, true, 256, SP>), grid_dim, block, lds, 0, s->dev, g, d_rgb, \
                           (uint32_t *)nullptr, (unsigned long long *)nullptr, sink);                                          \
    } while (0)
    const bool spill = s->dev.n_objs > s->dev.n_lds;
    if (cam->lens_radius > 0) { if (spill) CAPTURE(true, true); else CAPTURE(true, false); }
    else                      { if (spill) CAPTURE(false, true); else CAPTURE(false, false); }
#undef CAPTURE
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
    ON_DEVICE(s->device);
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

int cgrt_unpermute_stripes(const float *shares, int n_present, int nshares, int width, int height, int stripe_rows,
                           int rows_local, int channels, float *frame, void *stream) {
    if (!shares || !frame || width <= 0 || height <= 0 || channels <= 0 || nshares < 1 || n_present < 1 || n_present > nshares ||
        stripe_rows <= 0 || rows_local <= 0 || rows_local % stripe_rows)
        return fail(CGRT_ERR_INVALID, "cgrt_unpermute_stripes: bad argument");
    const int row_floats = width * channels;
    const dim3 grid((unsigned)((row_floats + 255) / 256), (unsigned)height);
    hipLaunchKernelGGL(unpermute_stripes_kernel, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), shares, n_present, nshares,
                       row_floats, stripe_rows, rows_local, frame);
    HIP_TRY(hipGetLastError());
    return CGRT_OK;
}

int cgrt_trace_grid_host(const cgrt_scene *s, const cgrt_camera *cam, const cgrt_grid *grid, float *rgb,
                         uint32_t *nhit, uint64_t *counters) {
    int rc = check_grid(s, cam, grid);
    if (rc) return rc;
    if (!rgb) return fail(CGRT_ERR_INVALID, "null rgb");
    ON_DEVICE(s->device);
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
    ON_DEVICE(s->device);
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

int cgrt_surface_colors(const cgrt_scene *s, int obj, const double *points3, int n, double *colors3) {
    if (!s || !s->committed) return fail(CGRT_ERR_INVALID, "scene not committed");
    if (obj < 0 || obj >= s->dev.n_objs || n < 0 || (n > 0 && (!points3 || !colors3))) return fail(CGRT_ERR_INVALID, "bad argument");
    if (n == 0) return CGRT_OK;
    ON_DEVICE(s->device);
    DevBuf b_p, b_c;
    HIP_TRY(b_p.alloc((size_t)n * 24));
    HIP_TRY(b_c.alloc((size_t)n * 24));
    HIP_TRY(hipMemcpy(b_p.p, points3, (size_t)n * 24, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(surface_colors_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, s->dev, obj, b_p.as<double>(), n,
                       b_c.as<double>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(colors3, b_c.p, (size_t)n * 24, hipMemcpyDeviceToHost));
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

#include "cgrt_photon.hpp"

#ifdef CGRT_UTIL
// development aid (make exp NAME=util DEFS=-DCGRT_UTIL; tools/util_probe.py): lane-utilisation probes, see UTILP in cgrt_device_math.hpp
extern "C" void cgrt_util_dump(unsigned long long *out) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_util), sizeof(unsigned long long) * 64);
    unsigned long long z[64] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_util), z, sizeof(z));
}
#endif
