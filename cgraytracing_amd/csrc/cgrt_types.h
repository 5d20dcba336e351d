// POD layouts shared by the host-side scene builder (cgrt_build.cpp) and the gfx950 kernels (cgrt_hip.hip).
// Everything the kernels read lives in these structs; the flattened scene is uploaded once at commit.
#ifndef CGRT_TYPES_H
#define CGRT_TYPES_H
#include <stdint.h>

namespace cgrt {

// Reference constants (main.cpp:24-25, objects.h:15,143-144).
static constexpr double kEps = 1e-4;       // main.cpp:24  ray-origin offset and material thresholds
static constexpr double kInf = 1e10;       // main.cpp:25 / objects.h:15
static constexpr int kMinKd = 10;          // objects.h:143  leaf iff triangle count < 10
static constexpr double kBoxPad = 1.001e-4;// objects.h:144 pads boxes by 1e-4; see DESIGN.md "box test" for the slack
static constexpr int kMaxDepth = 5;        // main.cpp:35

enum ObjKind : int32_t { KIND_SPHERE = 0, KIND_PLANE = 1, KIND_MESH = 2, KIND_BEZIER = 3 };

// One entry per top-level object, in `objs` order.  Staged in LDS by every workgroup.
// 16 doubles = 128 B.
struct ObjRec {
    // sphere: a = centre, s0 = radius^2                       (objects.h:83-88)
    // plane : a = position, b = normal                         (objects.h:541-542)
    // mesh  : a = centre, s0 = radius^2 of a sphere around the mesh (objects.h:470-475 hold nothing; the scene walk's early-out)
    // bezier: a = position, b.x = cp[last].z, index -> BezierRec (bezier.h:303-313)
    double a[3];
    double b[3];
    double s0;
    double col[3];        // surfaceColor
    double refl, transp;  // reflection, transparency
    int32_t kind;
    int32_t tree;         // mesh: tree index; plane: bump tree index or -1
    int32_t tex;          // plane: texture index or -1
    int32_t aux;          // mesh: objtype (objects.h:434); bezier: index into beziers[]
    int32_t axis;         // plane: k = 0, 1, 2 when the normal is exactly +-e_k (two components +-0, one +-1), else -1 -- such a
                          // plane's `len` is (a_k - o_k) / d_k, the same double the general expression gives (cgrt_scene_walk.hpp)
    int32_t pad1, pad2, pad3;
};
static_assert(sizeof(ObjRec) == 128, "ObjRec layout");

// Tree node in the reference's preorder numbering (objects.h:217-226), with a skip link instead of child
// indices: next node when the subtree is abandoned.  32 B = two 16-byte loads per lane (tree traversal is bound by
// the number of vector-memory requests, not by arithmetic).  The box is the reference's box grown by kBoxPad and
// then rounded OUTWARD to fp32: the box test only has to accept a superset of what KDNode::intersect accepts
// (DESIGN.md section 4.2), so a slightly larger box is still exact for (len, triangle, counter).
struct NodeRec {
    float lo[3];
    float hi[3];
    int32_t skip;  // index of the first node after this subtree
    int32_t leaf;  // inner node: -1; leaf: (first triangle in TriRec order << 4) | triangle count (0..9)
};
static_assert(sizeof(NodeRec) == 32, "NodeRec layout");

// Leaf triangle, stored in leaf order: pa and the two edge vectors the reference recomputes per test
// (e1 = pa-pb, e2 = pa-pc, objects.h:98-99; same doubles).  72 B.
struct TriRec {
    double pa[3];
    double e1[3];
    double e2[3];
};

// Opaque meshes: a triangle of the triangle-level hierarchy -- the same TriRec as in tris[], plus its place in the
// reference's leaf order, which decides exact ties (k: index in leaf order; leaf: first index of its reference leaf).
struct OTriRec {
    TriRec t;
    int32_t k;
    int32_t leaf;
};
static_assert(sizeof(OTriRec) == 80, "OTriRec layout");

// Opaque meshes: node of the 4-wide form of the triangle-level hierarchy -- the boxes of up to four children, one coordinate
// of all four per 16-byte load, so that one fetch round (seven independent loads of one 128-byte line) decides four boxes
// and the walk needs ~2.5x fewer DEPENDENT fetches than with one box per node.  ref: >= 0 a leaf ((first otri << 4) | count),
// kWideNone an empty slot, otherwise ~(index of the child node, relative to the tree's first wide node).  Boxes are grown
// and rounded outward exactly like NodeRec's.
struct WideNodeRec {
    float lox[4], loy[4], loz[4], hix[4], hiy[4], hiz[4];
    int32_t ref[4];
    int32_t pad[4];
};
static_assert(sizeof(WideNodeRec) == 128, "WideNodeRec layout");
static constexpr int32_t kWideNone = INT32_MIN;
static constexpr int kWideLdsDepth = 16;  // of them in LDS where the workgroup has room (8 bytes per entry and thread)
static constexpr int kWideStack = 64;  // entries of the walk's per-lane stack; cgrt_build.cpp guarantees a tree needs fewer

struct TreeRec {
    int64_t node_begin;  // into nodes[]
    int64_t tri_begin;   // into tris[]
    int64_t otri_begin;  // tri_level only: into otris[]
    int32_t nnodes;
    int32_t ntris;
    int32_t hfield;      // bump floors: index into hfields[] (the same triangles as a regular grid), else -1
    int32_t noct;        // copies of the node array at node_begin, nnodes apart: 8 (one per ray-direction octant,
                         // children near-to-far) for the SAH hierarchy, 1 for the reference-order tree
    int32_t tri_level;   // 1: the hierarchy is over single triangles (opaque owner), leaves index otris[]; 0: over the
                         // reference's leaves, leaves index tris[]
    float bmax;          // >= |coordinate| of every box face of this tree (the single-precision box test's error bound, Ray32)
    int64_t tbox_begin;  // into tboxes[]: one box per triangle of tris[], same order
    int64_t wnode_begin; // tri_level only: into wnodes[] (nwide records, root first); nwide == 0: walk nodes[] instead
    int32_t nwide;
    int32_t pad2;
};

// A bump-mapped floor's displacement mesh (objects.h:482-503) is a height field over a regular x-z grid: one quad per
// 3x3 texel block, split into triangles (a,b,c) and (d,b,c).  Besides the reference's tree, the triangles are kept in
// GRID order so that an opaque floor can be traversed cell by cell along the ray: the same
// triangle records and the same triangle test, hence the same (len, triangle); `k` and `leaf` carry each triangle's
// place in the reference's leaf order, which decides exact ties (objects.h:281,297).
struct HCellRec {
    TriRec t[2];
    int32_t k[2];     // index in the tree's leaf-ordered tris[] (relative to tri_begin)
    int32_t leaf[2];  // first triangle index of the leaf holding it (grows with the leaf's sequence number)
};
static_assert(sizeof(HCellRec) == 160, "HCellRec layout");
struct HFieldRec {
    int64_t cell_begin;  // into hcells[] and hcell_y[]: cell (i, j) at cell_begin + i * nx + j
    int32_t nx, nz;      // cells along x (cols/3 - 1) and z (rows/3 - 1)
    double x0, z0;       // texture position (vertex (0,0))
    double hx, hz;       // cell pitch: lenx*3/cols, leny*3/rows
    double ylo, yhi;     // range of vertex heights (plane y included)
    double ihx, ihz;     // 1.0 / hx, 1.0 / hz (the same IEEE quotients the walk used to compute per ray)
};
// Heights of a cell's four vertices, rounded outward to fp32: a ray that stays above or below them over the cell's column cannot
// hit its triangles (hfield_intersect's cull; 8 bytes instead of the 160-byte cell record for the cells a ray merely flies over)
struct HCellY {
    float lo, hi;
};

struct TexRec {
    int64_t texel_begin;  // into texels[] (3 bytes per texel, row-major)
    int32_t rows, cols;
    double n[3];
    double p[3];
    double lenx, leny;
    int32_t isbump;
    int32_t pad;
};

struct BezierRec {
    double cp[6][3];
    int32_t ncp;
    int32_t pad;
    double box[6];  // xmin,xmax,ymin,ymax,zmin,zmax (bezier.h:64-69)
};

// A Bezier surface of revolution in kBezSlabs pieces of its parameter range: the piece u in [k, k+1] / kBezSlabs lies between
// the heights ylo..yhi (relative to the object's position) and within radius sqrt(r2) of the axis -- hull of the piece's own
// control points, grown by 1e-3.  A ray that stays outside every piece's cylinder within that piece's heights cannot come
// within 1e-4 of the surface, so none of Bezier::intersect's solves could be accepted (bezier.h:257).
struct BezSlabRec {
    double ylo, yhi, r2, pad;
};
static constexpr int kBezSlabs = 64;

// Top-level objects kept in LDS per workgroup: 768 x 128 B = 96 KiB, which leaves every kernel variant's other LDS (pending-ray
// levels 38 KiB, node cache 8 KiB, wide-walk stack 32 KiB, Bezier scratch) inside a CU's 160 KiB.  `vector<Object*> objs`
// (main.cpp:277) has no bound: objects beyond that are read from HBM / L2 -- through the scalar cache in the sphere loop, through a
// per-wave LDS staging record in the general loop.
static constexpr int kLdsObjsMax = 768;
static constexpr int kMaxObjs = 1 << 20;  // sanity limit of cgrt_scene_add_* (CGRT_ERR_LIMIT beyond it)

// Kernel argument block.
struct DeviceScene {
    const ObjRec *objs;
    const NodeRec *nodes;
    const TriRec *tris;
    const TreeRec *trees;
    const TexRec *texs;
    const uint8_t *texels;
    const BezierRec *beziers;
    const HFieldRec *hfields;
    const HCellRec *hcells;
    const HCellY *hcell_y;  // parallel to hcells
    const OTriRec *otris;
    const NodeRec *tboxes;
    const WideNodeRec *wnodes;
    const BezSlabRec *bez_slabs;  // n_beziers x kBezSlabs
    const double *cover;  // n_cover x (cx, cy, cz, r): spheres that together contain every mesh triangle
    int32_t n_objs, n_trees, n_texs, n_beziers;
    int32_t n_lds;       // objects 0 .. n_lds-1 are staged in LDS by every workgroup (all of them up to kLdsObjsMax); the rest are
                         // read from `objs` where the walk needs them (cgrt_scene_walk.hpp)
    int32_t prim_obj;    // the scene's ONE opaque mesh with a 4-wide hierarchy when it has no other mesh and no Bezier object (its
                         // primary-ray walks can run as a kernel of their own, cgrt_primwalk.hpp), else -1
    int32_t has_mesh;    // any tree to traverse (mesh or bump plane)
    int32_t has_bezier;
    int32_t all_spheres; // fast path selector
    int32_t has_glass;   // some object takes the refraction branch (main.cpp:135)
    int32_t cached_tree; // tree whose nodes every workgroup stages in LDS (-1: none)
    int32_t cached_nodes;
    int32_t has_wide;    // some tree is walked in its 4-wide form (the eye kernels' LDS stack)
    int32_t light_trees; // light_ok and some plane is bump-mapped: the light variant is the tree-capable one
    int32_t light_ok;    // every plane is diffuse (bump-mapped or not) and some object is "special" (mesh, Bezier, mirror or glass
                         // sphere): tiles whose primary rays provably stay clear of the special objects' bounding spheres see
                         // diffuse spheres and planes only and may be rendered by the light kernel variant (cgrt_hip.hip)
    int32_t n_cover;
    int32_t prim_finish;    // prim_obj >= 0 and no plane carries a bump tree: primary_walk_kernel may complete units (cgrt_primwalk.hpp)
    int32_t prun_begin, prun_end;  // objects [prun_begin, prun_end): >= 3 axis-aligned planes without a bump tree, tested as a group
                                   // (cgrt_scene_walk.hpp plane_run), only planes in front of them; prun_end == 0: none
    int32_t single_ray;     // every plane and sphere is diffuse and there is no mesh and no Bezier object: every ray tree is one ray
    int32_t light_hf_only;  // light_trees and every plane's tree is an opaque bump floor with a grid (hfield): the light variant needs the
                            // height-field walk and nothing else of the tree code (HFONLY)
};

}  // namespace cgrt
#endif
