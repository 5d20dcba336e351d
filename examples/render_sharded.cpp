// The eye pass of the reference's render(objs) across the GPUs of one node from one C++ process
// (include/cgrt_host_sharded.hpp: a host thread per GPU, block-cyclic stripes, ncclSend/ncclRecv gather to GPU 0).
//
//   cgrt_sharded [--gpus N] [--width W] [--height H] [--spp S] [--dof] [--stripe ROWS] [--raw out.f32] [--emulate SHARES]
//
// --emulate SHARES (one-GPU machines): deal the stripes to SHARES shares, render them one after another on GPU 0 and
// assemble the frame with the same device un-permute the N > 1 path uses -- everything but the RCCL transfers.
//
// Scene: C2 (BASELINE.json configs[1]: the five wall spheres of main.cpp:281-285 + diffuse, mirror and glass spheres).
// Built by hipcc (HIP runtime + RCCL); run on this project's one-GPU box with N = 1 only -- the N > 1 path (communicator
// over all devices, grouped send/recv) is compiled and exercised by the same code on a multi-GPU node, which this work
// never had: treat it as unverified there.
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "cgrt_host_sharded.hpp"

using namespace cgrt_host;

int main(int argc, char *argv[]) {
    RenderParams rp;
    rp.width = 1920;
    rp.height = 1080;
    rp.num_of_samples = 64;
    int gpus = 0, stripe = 8, emulate = 0;
    std::string raw;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto next = [&]() -> const char * { return (i + 1 < argc) ? argv[++i] : ""; };
        if (a == "--gpus") gpus = std::atoi(next());
        else if (a == "--width") rp.width = std::atoi(next());
        else if (a == "--height") rp.height = std::atoi(next());
        else if (a == "--spp") rp.num_of_samples = std::atoi(next());
        else if (a == "--stripe") stripe = std::atoi(next());
        else if (a == "--emulate") emulate = std::atoi(next());
        else if (a == "--dof") rp.depth_of_field = true;
        else if (a == "--raw") raw = next();
    }
    std::vector<Sphere> sphs;
    sphs.push_back(Sphere(Vec3(0.0, -10020, 0), 10000, Vec3(0.25, 0.25, 0.25), 0.0, 0.0));
    sphs.push_back(Sphere(Vec3(10020, 0.0, 0), 10000, Vec3(0.25, 0.75, 0.25), 0.0, 0.0));
    sphs.push_back(Sphere(Vec3(-10020, 0.0, 0), 10000, Vec3(0.75, 0.25, 0.25), 0.0, 0.0));
    sphs.push_back(Sphere(Vec3(0.0, 0.0, 10040), 10000, Vec3(0.25, 0.25, 0.25), 0.0, 0.0));
    sphs.push_back(Sphere(Vec3(0.0, 10020, 0), 10000, Vec3(0.25, 0.25, 0.25), 0.0, 0.0));
    sphs.push_back(Sphere(Vec3(-15.0, -20.0, 60), 10, Vec3(0.3, 0.3, 0.3), 0.0, 0.0));
    sphs.push_back(Sphere(Vec3(10.0, -13.0, 30), 7, Vec3(1.0, 1.0, 1.0), 0.8, 0.0));
    sphs.push_back(Sphere(Vec3(-8.0, -13.0, 25), 7, Vec3(1.0, 1.0, 1.0), 0.8, 0.5));
    std::vector<Object *> objs;
    for (size_t i = 0; i < sphs.size(); i++) objs.push_back(&sphs[i]);
    std::vector<float> image;
    ShardedStats st;
    try {
        render_sharded(objs, rp, image, gpus, stripe, &st, emulate);
    } catch (const Error &e) {
        std::fprintf(stderr, "render_sharded failed (%d): %s\n", e.code, e.what());
        return 1;
    }
    std::printf("gpus: %d rays: %llu hitpoints: %llu slowest share: %.3f ms gather: %.3f ms\n", st.n_gpus, (unsigned long long)st.rays,
                (unsigned long long)st.hitpoints, st.ms_render_slowest, st.ms_gather);
    if (!raw.empty()) {
        FILE *f = std::fopen(raw.c_str(), "wb");
        if (!f) return 2;
        std::fwrite(image.data(), sizeof(float), image.size(), f);
        std::fclose(f);
    }
    return 0;
}
