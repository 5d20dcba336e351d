// Host program in the reference's style, built on include/cgrt_host.hpp: the scene is assembled with the
// reference's constructor signatures, pushed into `vector<Object*> objs` in the reference's order (spheres,
// planes, meshes, Bezier; main.cpp:355-378) and handed to render(objs).  The per-pixel work runs on the GPU.
//
//   cgrt_main [--scene c2|planes|chess|vase] [--mesh FILE TYPE] [--width W] [--height H] [--spp N] [--dof] [--depth D]
//             [--raw out.f32] [--ppm out.ppm]
//             [--photons N --png test.png]   the whole of render() + main(): photon pass, gather, tone map, PNG
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "cgrt_host.hpp"

using namespace cgrt_host;
using std::vector;

int main(int argc, char *argv[]) {
    RenderParams rp;
    rp.width = 256;
    rp.height = 192;
    std::string scene = "c2", raw, ppm, mesh_file, png;
    long long photons = 0;
    int mesh_type = 0;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto next = [&]() -> const char * { return (i + 1 < argc) ? argv[++i] : ""; };
        if (a == "--scene") scene = next();
        else if (a == "--mesh") { mesh_file = next(); mesh_type = std::atoi(next()); }
        else if (a == "--width") rp.width = std::atoi(next());
        else if (a == "--height") rp.height = std::atoi(next());
        else if (a == "--spp") rp.num_of_samples = std::atoi(next());
        else if (a == "--depth") rp.max_depth = std::atoi(next());
        else if (a == "--dof") rp.depth_of_field = true;
        else if (a == "--raw") raw = next();
        else if (a == "--ppm") ppm = next();
        else if (a == "--png") png = next();
        else if (a == "--photons") photons = std::atoll(next());
    }

    vector<Object *> objs;
    vector<Sphere> sphs;
    vector<Plane> plns;
    if (scene == "c2") {
        sphs.push_back(Sphere(Vec3(0.0, -10020, 0), 10000, Vec3(0.25, 0.25, 0.25), 0.0, 0.0));
        sphs.push_back(Sphere(Vec3(10020, 0.0, 0), 10000, Vec3(0.25, 0.75, 0.25), 0.0, 0.0));
        sphs.push_back(Sphere(Vec3(-10020, 0.0, 0), 10000, Vec3(0.75, 0.25, 0.25), 0.0, 0.0));
        sphs.push_back(Sphere(Vec3(0.0, 0.0, 10040), 10000, Vec3(0.25, 0.25, 0.25), 0.0, 0.0));
        sphs.push_back(Sphere(Vec3(0.0, 10020, 0), 10000, Vec3(0.25, 0.25, 0.25), 0.0, 0.0));
        sphs.push_back(Sphere(Vec3(-15.0, -20.0, 60), 10, Vec3(0.3, 0.3, 0.3), 0.0, 0.0));
        sphs.push_back(Sphere(Vec3(10.0, -13.0, 30), 7, Vec3(1.0, 1.0, 1.0), 0.8, 0.0));
        sphs.push_back(Sphere(Vec3(-8.0, -13.0, 25), 7, Vec3(1.0, 1.0, 1.0), 0.8, 0.5));
    } else {
        Texture tex;  // default: untextured floor (main.cpp:349-353 style planes)
        if (scene == "chess") {
            // a procedural 64x64 checker decoded the way main.cpp:303-316 decodes an image: byte / 256 per channel
            vector<vector<Vec3> > tdata;
            for (int i = 0; i < 64; i++) {
                vector<Vec3> row;
                for (int j = 0; j < 64; j++) {
                    const unsigned char v = (((i / 8) + (j / 8)) & 1) ? 230 : 25;
                    Vec3 col;
                    col.x = (double)v / (double)256;
                    col.y = (double)(unsigned char)(v / 2 + 60) / (double)256;
                    col.z = (double)(unsigned char)(255 - v) / (double)256;
                    row.push_back(col);
                }
                tdata.push_back(row);
            }
            tex = Texture(tdata, Vec3(0, 1, 0), Vec3(-21, 0, 0), 42, 40, true);  // bump-mapped, like main.cpp:320
        }
        plns.push_back(Plane(Vec3(0.0, -20, 0), Vec3(0, 1, 0), Vec3(0.15, 0.15, 0.15), 0.0, 0.0, tex));
        plns.push_back(Plane(Vec3(20, 0.0, 0), Vec3(-1, 0, 0), Vec3(0.15, 0.50, 0.15), 0.0, 0.0));
        plns.push_back(Plane(Vec3(-20, 0.0, 0), Vec3(1, 0, 0), Vec3(0.50, 0.15, 0.15), 0.0, 0.0));
        plns.push_back(Plane(Vec3(0.0, 0.0, 40), Vec3(0, 0, -1), Vec3(0.15, 0.15, 0.15), 0.0, 0.0));
        plns.push_back(Plane(Vec3(0.0, 20, 0), Vec3(0, -1, 0), Vec3(0.15, 0.15, 0.15), 0.0, 0.0));
    }
    for (size_t i = 0; i < sphs.size(); i++) objs.push_back(&sphs[i]);
    for (size_t i = 0; i < plns.size(); i++) objs.push_back(&plns[i]);
    TriangleMesh *tm = nullptr;
    if (!mesh_file.empty()) {
        tm = new TriangleMesh(mesh_file.c_str(), 3.0, Vec3(1.0, -4.0, 30.0), Vec3(0.6, 0.7, 0.9), 0.8, 0.5, mesh_type);
        objs.push_back(tm);
    }
    vector<Vec3> cp;
    cp.push_back(Vec3(0, -10, 4));
    cp.push_back(Vec3(0, 2, 4));
    cp.push_back(Vec3(0, -2, 0));
    cp.push_back(Vec3(0, 10, 2));
    Bezier vase(cp, Vec3(15, -10.1, 35), Vec3(1.0, 1.0, 1.0), 0.5, 0.0);
    if (scene == "vase") objs.push_back(&vase);

    if (photons > 0) {  // render() as the reference runs it, then main()'s PNG loop (main.cpp:403-412)
        PhotonParams pp;
        pp.num_photon = photons;
        pp.num_threads = 1;
        vector<double> img;
        vector<unsigned char> image_data;
        PpmStats ps;
        try {
            render_ppm(objs, rp, pp, img, image_data, &ps);
            if (!png.empty()) write_png(png.c_str(), rp.width, rp.height, image_data);
        } catch (const Error &e) {
            std::fprintf(stderr, "render failed (%d): %s\n", e.code, e.what());
            return 1;
        }
        std::printf("\nhitpoints: %llu\n", (unsigned long long)ps.hitpoints);  // main.cpp:265
        std::printf("photon events: %llu; eye %.2f ms, table %.2f ms, photons %.2f ms, gather %.2f ms\n",
                    (unsigned long long)ps.photon_events, ps.ms_eye, ps.ms_table, ps.ms_photons, ps.ms_gather);
        delete tm;
        return 0;
    }
    vector<float> image;
    RenderStats st;
    try {
        render(objs, rp, image, &st);
    } catch (const Error &e) {
        std::fprintf(stderr, "render failed (%d): %s\n", e.code, e.what());
        return 1;
    }
    std::printf("rays: %llu hitpoints: %llu\n", (unsigned long long)st.rays, (unsigned long long)st.hitpoints);
    if (!raw.empty()) {
        FILE *f = std::fopen(raw.c_str(), "wb");
        std::fwrite(image.data(), sizeof(float), image.size(), f);
        std::fclose(f);
    }
    if (!ppm.empty()) {  // top row first, like main.cpp:403-411; simple gamma for viewing
        FILE *f = std::fopen(ppm.c_str(), "wb");
        std::fprintf(f, "P6\n%d %d\n255\n", rp.width, rp.height);
        for (int i = rp.height - 1; i >= 0; i--)
            for (int j = 0; j < rp.width; j++)
                for (int k = 0; k < 3; k++) {
                    double v = image[((size_t)i * rp.width + j) * 3 + k];
                    int b = (int)(std::pow(v < 0 ? 0 : (v > 1 ? 1 : v), 1 / 2.2) * 255 + .5);
                    std::fputc(b, f);
                }
        std::fclose(f);
    }
    delete tm;
    return 0;
}
