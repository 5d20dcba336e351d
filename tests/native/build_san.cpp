// Host-side scene assembly (cgrt_build.cpp: loaders, bump mesh, tree build) under AddressSanitizer and
// UBSan -- CPU build only; driven by tests/test_capi_host.py with the golden assets.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "cgrt_build.h"

using namespace cgrt;

// A mesh large enough that the multi-threaded parts of the build run (halves of more than 16 384 triangles side by side,
// the eight octant layouts): a wavy 150 x 150 grid = 44 402 triangles, built as a glass and as an opaque object.
static int big_mesh() {
    const int N = 150;
    std::vector<double> t9;
    auto vtx = [&](int i, int j, double *v) {
        v[0] = -10.0 + 20.0 * i / N;
        v[2] = 25.0 + 20.0 * j / N;
        v[1] = -12.0 + 1.5 * std::sin(0.37 * i) * std::cos(0.23 * j) + 0.001 * ((i * 31 + j * 17) % 13);
    };
    for (int i = 0; i < N - 1; i++)
        for (int j = 0; j < N - 1; j++) {
            double a[3], b[3], c[3], d[3];
            vtx(i, j, a); vtx(i + 1, j, b); vtx(i, j + 1, c); vtx(i + 1, j + 1, d);
            for (const double *v : {a, b, c}) t9.insert(t9.end(), v, v + 3);
            for (const double *v : {d, b, c}) t9.insert(t9.end(), v, v + 3);
        }
    const double col[3] = {0.5, 0.5, 0.5};
    HostScene a, b;
    a.add_mesh_triangles(t9.data(), (int)(t9.size() / 9), col, 0.8, 0.5, 0);  // glass: hierarchy over reference leaves
    b.add_mesh_triangles(t9.data(), (int)(t9.size() / 9), col, 0.0, 0.0, 0);  // opaque: triangle-level hierarchy
    const HostTree &A = a.trees[0], &B = b.trees[0];
    // the reference-order part must not depend on the owner's material (nor on thread timing)
    if (A.leaf_ids != B.leaf_ids || A.node_lr_size != B.node_lr_size || A.bbox != B.bbox) return 5;
    if (A.bvh.size() != 8u * (size_t)A.bvh_nodes || B.bvh.size() != 8u * (size_t)B.bvh_nodes) return 6;
    if (!B.tri_level || B.otris.size() != B.tris.size() || A.tri_level) return 7;
    std::printf("big: tris=%zu ref_nodes=%zu leaf_bvh=%d tri_bvh=%d\n", A.tris.size(), A.nodes.size(), A.bvh_nodes, B.bvh_nodes);
    return 0;
}

int main(int argc, char **argv) {
    if (argc == 2 && std::strcmp(argv[1], "--big") == 0) return big_mesh();
    if (argc < 2) return 2;
    HostScene sc;
    const double b[3] = {1.0, -4.0, 30.0}, col[3] = {0.6, 0.7, 0.9}, n[3] = {0, 1, 0}, p[3] = {-21, 0, 0};
    // every mesh file given: (path, type)
    for (int i = 1; i + 1 < argc; i += 2) {
        int r = sc.add_mesh_file(argv[i], 3.0, b, col, 0.8, 0.5, std::atoi(argv[i + 1]));
        std::printf("%s -> %d (%s) trees=%zu\n", argv[i], r, sc.error.c_str(), sc.trees.size());
    }
    // a bump floor from a synthetic 90x60 texture
    std::vector<uint8_t> rgb(90 * 60 * 3);
    for (size_t k = 0; k < rgb.size(); k++) rgb[k] = (uint8_t)((k * 37u + (k >> 3)) & 255u);
    int t = sc.add_texture(rgb.data(), 60, 90, n, p, 42, 40, 1);
    const double pp[3] = {0, -20, 0};
    sc.add_plane(pp, n, col, 0, 0, t);
    // degenerate inputs
    sc.add_mesh_triangles(nullptr, 0, col, 0, 0, 0);
    double one[9] = {0, 0, 30, 1, 0, 30, 0, 1, 30};
    sc.add_mesh_triangles(one, 1, col, 0, 0, 2);
    size_t nodes = 0, tris = 0;
    for (auto &tr : sc.trees) {
        nodes += tr.nodes.size();
        tris += tr.tris.size();
        for (size_t k = 0; k < tr.nodes.size(); k++) {
            const NodeRec &nd = tr.nodes[k];
            if (nd.skip <= (int)k || nd.skip > (int)tr.nodes.size()) return 3;  // skip links move forward
            if (nd.leaf >= 0 && (size_t)((nd.leaf >> 4) + (nd.leaf & 15)) > tr.tris.size()) return 4;
        }
    }
    std::printf("objs=%zu trees=%zu nodes=%zu leaf_tris=%zu\n", sc.objs.size(), sc.trees.size(), nodes, tris);
    return 0;
}
