// Host-side scene assembly (cgrt_build.cpp: loaders, bump mesh, tree build) under AddressSanitizer and
// UBSan -- CPU build only; driven by tests/test_capi_host.py with the golden assets.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "cgrt_build.h"

using namespace cgrt;

int main(int argc, char **argv) {
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
