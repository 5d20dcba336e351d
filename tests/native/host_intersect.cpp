// Object::intersect / intersect_batch of include/cgrt_host.hpp (the reference's virtual, objects.h:20) from a plain C++
// program: rays read from stdin ("ox oy oz dx dy dz" per line), one object selected by argv[1], results printed with
// full precision.  Driven by tests/test_gpu_parity.py, which compares them with the oracle bit for bit.
//   host_intersect sphere|plane|mesh FILE|vase
#include <cstdio>
#include <cstring>
#include <vector>

#include "cgrt_host.hpp"

using namespace cgrt_host;

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    Sphere sph(Vec3(-8.0, -13.0, 25), 7, Vec3(1.0, 1.0, 1.0), 0.8, 0.5);
    Plane pln(Vec3(0.0, -20, 0), Vec3(0, 1, 0), Vec3(0.15, 0.15, 0.15), 0.0, 0.0);
    std::vector<Vec3> cp;
    cp.push_back(Vec3(0, -10, 4)); cp.push_back(Vec3(0, 2, 4)); cp.push_back(Vec3(0, -2, 0)); cp.push_back(Vec3(0, 10, 2));
    Bezier vase(cp, Vec3(15, -10.1, 35), Vec3(1.0, 1.0, 1.0), 0.5, 0.0);
    TriangleMesh *tm = nullptr;
    Object *obj = nullptr;
    if (!std::strcmp(argv[1], "sphere")) obj = &sph;
    else if (!std::strcmp(argv[1], "plane")) obj = &pln;
    else if (!std::strcmp(argv[1], "vase")) obj = &vase;
    else if (!std::strcmp(argv[1], "mesh") && argc > 2) obj = tm = new TriangleMesh(argv[2], 3.0, Vec3(1.0, -4.0, 30.0), Vec3(0.6, 0.7, 0.9), 0.8, 0.5, 0);
    if (!obj) return 2;
    std::vector<double> o, d;
    double v[6];
    while (std::scanf("%lf %lf %lf %lf %lf %lf", v, v + 1, v + 2, v + 3, v + 4, v + 5) == 6) {
        o.insert(o.end(), v, v + 3);
        d.insert(d.end(), v + 3, v + 6);
    }
    const int n = (int)(o.size() / 3);
    try {
        // the first few one at a time through the reference's signature, then everything as a batch; both must agree
        std::vector<int32_t> hit(n);
        std::vector<double> len(n), nrm(3 * (size_t)n);
        obj->intersect_batch(o.data(), d.data(), n, hit.data(), len.data(), nrm.data());
        for (int i = 0; i < n && i < 8; i++) {
            double l = -1;
            Vec3 nv(9, 9, 9);
            const bool h = obj->intersect(Vec3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), Vec3(d[3 * i], d[3 * i + 1], d[3 * i + 2]), l, nv);
            if (h != (hit[i] != 0)) return 3;
            if (h && (l != len[i] || nv.x != nrm[3 * i] || nv.y != nrm[3 * i + 1] || nv.z != nrm[3 * i + 2])) return 4;
            if (!h && (l != -1 || nv.x != 9)) return 5;  // outputs untouched on a miss
        }
        for (int i = 0; i < n; i++)
            std::printf("%d %.17g %.17g %.17g %.17g\n", hit[i], hit[i] ? len[i] : 0.0, hit[i] ? nrm[3 * i] : 0.0,
                        hit[i] ? nrm[3 * i + 1] : 0.0, hit[i] ? nrm[3 * i + 2] : 0.0);
    } catch (const Error &e) {
        std::fprintf(stderr, "error %d: %s\n", e.code, e.what());
        return 1;
    }
    delete tm;
    return 0;
}
