// Host-side scene assembly: mesh loaders, bump-mesh construction, tree build, flattening.
// These are the reference's pre-pass (objects.h:217-267,338-403,480-504; texture.h:19-38), run once per
// scene on the CPU; the per-ray work is all in cgrt_hip.hip.
#ifndef CGRT_BUILD_H
#define CGRT_BUILD_H
#include <cstdint>
#include <string>
#include <vector>

#include "cgrt_types.h"

namespace cgrt {

struct HostTree {
    std::vector<double> tri9;        // construction order, 9 doubles per triangle
    // reference-numbered view (for verification dumps)
    std::vector<int32_t> node_lr_size;  // left, right, count
    std::vector<double> bbox;           // xmin,xmax,ymin,ymax,zmin,zmax (unpadded)
    std::vector<int32_t> leaf_ids;      // triangle ids of leaves, node order
    // device view: the reference's leaves (tris, in leaf order) and two hierarchies over them --
    // `nodes`: the reference's own tree in preorder with skip links;
    // `bvh`  : a surface-area-heuristic hierarchy over the SAME leaves, stored once per ray-direction octant
    //          (8 x bvh_nodes records, each a preorder with skip links whose children are ordered near-to-far
    //          for rays of that octant), see build_bvh()
    std::vector<NodeRec> nodes;
    std::vector<TriRec> tris;
    std::vector<NodeRec> tboxes;  // one grown, outward-rounded fp32 box per triangle of tris[] (the leaf scan's cheap pre-test)
    std::vector<NodeRec> bvh;
    int32_t bvh_nodes = 0;
    // opaque owner (transparency < eps): only the nearest hit matters, so the hierarchy goes down to small groups of
    // TRIANGLES (<= 4) instead of stopping at the reference's leaves; its leaves index otris
    bool tri_level = false;
    std::vector<OTriRec> otris;
    std::vector<WideNodeRec> wide;  // tri_level: the same hierarchy four children to a node (device walk); root = wide[0]
    int32_t wide_stack = 0;         // deepest the walk's stack can get on it (< kWideStack, or `wide` is left empty)
    void build_bvh(bool opaque);
    // bump floors only: the same triangles in grid order (construction order is cell-major: quad (i,j) = triangles
    // 2*(i*nx+j) and +1), see HCellRec
    bool is_hfield = false;
    HFieldRec hfield{};
    std::vector<HCellRec> hcells;
    std::vector<HCellY> hcell_y;  // per cell: its vertices' height range, rounded outward
    void build(bool opaque = false, bool with_bvh = true);
    void build_hfield(int nx, int nz, double x0, double z0, double hx, double hz);
    // Row f3 (cgrt_devbuild.hpp): an opaque owner's structure is built on the device at commit; nothing above is filled on the
    // host then except tri9 (a mesh's input triangles).  dev_kind: 0 = host build, 1 = mesh (4-wide triangle-level hierarchy),
    // 2 = bump floor (grid cells generated from the texture bytes).
    int dev_kind = 0;
    int64_t dev_ntri = 0;       // triangles the device build will produce / was given
    int dev_tex = -1;           // bump floor: texture index
    double dev_plane_y = 0;     // bump floor: the plane's y (objects.h:489-492)
    int dev_obj = -1;           // mesh: the owning object (its bounding sphere is computed by the device build)
    size_t dev_cover_at = 0;    // mesh: where its cover spheres go in HostScene::cover (appended at commit)
    // filled by the commit for the verification dumps (cgrt_scene_wide_dump on a device-built tree reads the device arrays)
    int32_t dev_nwide = 0;
    bool dev_balanced = false;
};
// nodes of the reference's tree over n triangles (objects.h:217-267: leaf below 10, halves n/2 and n - n/2): what
// cgrt_scene_stats reports for a tree that was never built on the host
int64_t ref_node_count(int64_t n);

struct HostTexture {
    std::vector<uint8_t> rgb;
    int rows = 0, cols = 0;
    double n[3], p[3], lenx = 0, leny = 0;
    bool isbump = false;
};

struct HostScene {
    std::vector<ObjRec> objs;
    std::vector<HostTree> trees;
    std::vector<HostTexture> textures;
    std::vector<BezierRec> beziers;
    std::vector<BezSlabRec> bez_slabs;  // kBezSlabs per Bezier object (see BezSlabRec)
    std::vector<double> cover;  // (cx, cy, cz, r) spheres that together contain every mesh triangle (classify_kernel)
    std::string error;
    int build_mode = 0;      // CGRT_BUILD_HOST / CGRT_BUILD_DEVICE (cgrt.h): where opaque owners' structures are built
    double host_build_ms = 0;  // wall-clock spent in HostTree::build* by the add_* calls

    int add_sphere(const double c[3], double r, const double sc[3], double refl, double transp);
    int add_texture(const uint8_t *rgb, int rows, int cols, const double n[3], const double p[3], double lx,
                    double ly, int isbump);
    int add_plane(const double p[3], const double n[3], const double sc[3], double refl, double transp, int tex);
    int add_mesh_file(const char *file, double a, const double b[3], const double sc[3], double refl,
                      double transp, int type);
    int add_mesh_triangles(const double *tri9, int ntri, const double sc[3], double refl, double transp, int type);
    int add_bezier(const double *cp3, int ncp, const double pos[3], const double sc[3], double refl, double transp);
};

// returns false and sets err on malformed input; a missing file gives an empty list and true
bool load_mesh_file(const char *file, double a, const double b[3], int type, std::vector<double> &tri9,
                    std::string &err);

}  // namespace cgrt
#endif
