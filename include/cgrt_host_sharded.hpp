// cgrt_host_sharded.hpp -- the eye pass of render(objs) across all GPUs of one node from ONE C++ process:
// one host thread per GPU, block-cyclic row stripes, framebuffer gathered to GPU 0 with grouped
// ncclSend / ncclRecv (RCCL over xGMI), un-permuted into image[h][w].
//
// What it replaces in the reference: the loop nest main.cpp:185-219 (as cgrt_host.hpp's render() does), sharded by
// rows as SURVEY.md section 8e lays out -- rows are independent once Hashtable::insert (hash.h:43-54) has become a
// per-pixel sum, the scene is replicated, and the only exchange is the frame gather.  Needs the HIP runtime and RCCL
// headers (hipcc, or g++ with -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include), unlike cgrt_host.hpp which is plain C++.
//
// Stripe s (stripe_rows rows, a multiple of 8) belongs to GPU s % N; each GPU's buffer is its stripes concatenated
// (cgrt_grid.stripe_rows / stripe_rank / stripe_nranks), so GPU 0 posts N-1 receives, one per xGMI link, and every
// other GPU one send of W * rows_local * 12 bytes.
#ifndef CGRT_HOST_SHARDED_HPP
#define CGRT_HOST_SHARDED_HPP
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <thread>

#include "cgrt_host.hpp"

namespace cgrt_host {

struct ShardedStats {
    int n_gpus = 0;
    uint64_t rays = 0, hitpoints = 0;
    double ms_render_slowest = 0, ms_gather = 0;
};

#define CGRT_HIP_OK(expr)                                                                               \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) throw Error(CGRT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)
#define CGRT_NCCL_OK(expr)                                                                                \
    do {                                                                                                  \
        ncclResult_t r_ = (expr);                                                                         \
        if (r_ != ncclSuccess) throw Error(CGRT_ERR_DEVICE, std::string(#expr) + ": " + ncclGetErrorString(r_)); \
    } while (0)

// image: [height][width][3] float, row 0 = bottom, as render() fills it.  n_gpus <= 0: every visible device.
inline void render_sharded(const std::vector<Object *> &objs, const RenderParams &rp, std::vector<float> &image, int n_gpus = 0,
                           int stripe_rows = 8, ShardedStats *stats = nullptr) {
    int ndev = 0;
    CGRT_HIP_OK(hipGetDeviceCount(&ndev));
    const int N = (n_gpus > 0 && n_gpus < ndev) ? n_gpus : ndev;
    if (N < 1) throw Error(CGRT_ERR_DEVICE, "no HIP device");
    if (stripe_rows <= 0 || stripe_rows % 8) throw Error(CGRT_ERR_INVALID, "stripe_rows must be a positive multiple of 8");
    const int W = rp.width, H = rp.height;
    const int nstripes = (H + stripe_rows - 1) / stripe_rows, per_rank = (nstripes + N - 1) / N;
    const int rows_local = N > 1 ? per_rank * stripe_rows : H;
    const size_t share = (size_t)rows_local * W * 3;

    std::vector<ncclComm_t> comms((size_t)N);
    if (N > 1) {
        std::vector<int> devs((size_t)N);
        for (int r = 0; r < N; r++) devs[(size_t)r] = r;
        CGRT_NCCL_OK(ncclCommInitAll(comms.data(), N, devs.data()));
    }
    std::vector<float *> d_rgb((size_t)N, nullptr);
    float *d_all = nullptr;  // GPU 0: the N shares, rank-major
    std::vector<hipStream_t> streams((size_t)N, nullptr);
    std::vector<std::string> errors((size_t)N);
    std::vector<double> ms((size_t)N, 0.0);
    std::vector<uint64_t> rays((size_t)N, 0), hps((size_t)N, 0);
    hipEvent_t g0 = nullptr, g1 = nullptr;

    auto worker = [&](int r) {  // one host thread per GPU; everything it creates lives on device r
        try {
            CGRT_HIP_OK(hipSetDevice(r));
            CGRT_HIP_OK(hipStreamCreate(&streams[(size_t)r]));
            SceneBuilder sb;  // the scene is replicated: every GPU builds and uploads its own copy
            for (const Object *o : objs) o->add_to(sb);
            check(cgrt_scene_commit(sb.scene, r));
            cgrt_camera cam;
            rp.camorg.get(cam.cam);
            cam.half_width = 10.0;
            cam.focus_plane = rp.focus_plane;
            cam.lens_radius = rp.depth_of_field ? rp.radius : 0.0;
            cgrt_grid g{};
            g.width = W;
            g.height = H;
            g.rows = rows_local;
            g.stripe_rows = N > 1 ? stripe_rows : 0;
            g.stripe_rank = r;
            g.stripe_nranks = N;
            g.spp = g.spp_total = rp.num_of_samples;
            g.max_depth = rp.max_depth;
            g.seed = rp.seed;
            uint64_t *d_cnt = nullptr;
            CGRT_HIP_OK(hipMalloc(&d_rgb[(size_t)r], share * sizeof(float)));
            CGRT_HIP_OK(hipMalloc(&d_cnt, CGRT_NCOUNTERS * sizeof(uint64_t)));
            CGRT_HIP_OK(hipMemsetAsync(d_cnt, 0, CGRT_NCOUNTERS * sizeof(uint64_t), streams[(size_t)r]));
            if (r == 0 && N > 1) CGRT_HIP_OK(hipMalloc(&d_all, share * sizeof(float) * (size_t)N));
            hipEvent_t e0, e1;
            CGRT_HIP_OK(hipEventCreate(&e0));
            CGRT_HIP_OK(hipEventCreate(&e1));
            CGRT_HIP_OK(hipEventRecord(e0, streams[(size_t)r]));
            check(cgrt_trace_grid(sb.scene, &cam, &g, d_rgb[(size_t)r], nullptr, d_cnt, streams[(size_t)r]));
            CGRT_HIP_OK(hipEventRecord(e1, streams[(size_t)r]));
            CGRT_HIP_OK(hipStreamSynchronize(streams[(size_t)r]));
            float t = 0;
            CGRT_HIP_OK(hipEventElapsedTime(&t, e0, e1));
            ms[(size_t)r] = t;
            uint64_t cnt[CGRT_NCOUNTERS];
            CGRT_HIP_OK(hipMemcpy(cnt, d_cnt, sizeof(cnt), hipMemcpyDeviceToHost));
            rays[(size_t)r] = cnt[CGRT_CNT_RAYS];
            hps[(size_t)r] = cnt[CGRT_CNT_HITPOINTS];
            (void)hipFree(d_cnt);
            (void)hipEventDestroy(e0);
            (void)hipEventDestroy(e1);
        } catch (const Error &e) {
            errors[(size_t)r] = e.what();
        }
    };
    {
        std::vector<std::thread> th;
        for (int r = 0; r < N; r++) th.emplace_back(worker, r);
        for (auto &t : th) t.join();
    }
    for (int r = 0; r < N; r++)
        if (!errors[(size_t)r].empty()) throw Error(CGRT_ERR_DEVICE, "GPU " + std::to_string(r) + ": " + errors[(size_t)r]);

    // ---- the gather: one group, N-1 receives on GPU 0 (N-1 distinct xGMI links), one send from every other GPU ----
    image.assign((size_t)W * H * 3, 0.f);
    std::vector<float> host_all;
    double ms_gather = 0;
    if (N > 1) {
        CGRT_HIP_OK(hipSetDevice(0));
        CGRT_HIP_OK(hipEventCreate(&g0));
        CGRT_HIP_OK(hipEventCreate(&g1));
        CGRT_HIP_OK(hipEventRecord(g0, streams[0]));
        CGRT_NCCL_OK(ncclGroupStart());
        for (int r = 1; r < N; r++) {
            CGRT_NCCL_OK(ncclRecv(d_all + share * (size_t)r, share, ncclFloat, r, comms[0], streams[0]));
            CGRT_NCCL_OK(ncclSend(d_rgb[(size_t)r], share, ncclFloat, 0, comms[(size_t)r], streams[(size_t)r]));
        }
        CGRT_NCCL_OK(ncclGroupEnd());
        CGRT_HIP_OK(hipMemcpyAsync(d_all, d_rgb[0], share * sizeof(float), hipMemcpyDeviceToDevice, streams[0]));
        CGRT_HIP_OK(hipEventRecord(g1, streams[0]));
        for (int r = 0; r < N; r++) {
            CGRT_HIP_OK(hipSetDevice(r));
            CGRT_HIP_OK(hipStreamSynchronize(streams[(size_t)r]));
        }
        float t = 0;
        CGRT_HIP_OK(hipSetDevice(0));
        CGRT_HIP_OK(hipEventElapsedTime(&t, g0, g1));
        ms_gather = t;
        host_all.resize(share * (size_t)N);
        CGRT_HIP_OK(hipMemcpy(host_all.data(), d_all, host_all.size() * sizeof(float), hipMemcpyDeviceToHost));
        // un-permute: local row j of rank r is global row ((j / S) * N + r) * S + j % S (cgrt.h)
        for (int r = 0; r < N; r++)
            for (int j = 0; j < rows_local; j++) {
                const int gr = ((j / stripe_rows) * N + r) * stripe_rows + j % stripe_rows;
                if (gr < H)
                    std::copy(host_all.begin() + (long)(share * (size_t)r + (size_t)j * W * 3),
                              host_all.begin() + (long)(share * (size_t)r + (size_t)(j + 1) * W * 3), image.begin() + (long)((size_t)gr * W * 3));
            }
    } else {
        CGRT_HIP_OK(hipSetDevice(0));
        CGRT_HIP_OK(hipMemcpy(image.data(), d_rgb[0], image.size() * sizeof(float), hipMemcpyDeviceToHost));
    }
    for (int r = 0; r < N; r++) {
        (void)hipSetDevice(r);
        if (d_rgb[(size_t)r]) (void)hipFree(d_rgb[(size_t)r]);
        if (streams[(size_t)r]) (void)hipStreamDestroy(streams[(size_t)r]);
        if (N > 1) (void)ncclCommDestroy(comms[(size_t)r]);
    }
    (void)hipSetDevice(0);
    if (d_all) (void)hipFree(d_all);
    if (g0) (void)hipEventDestroy(g0);
    if (g1) (void)hipEventDestroy(g1);
    if (stats) {
        stats->n_gpus = N;
        stats->ms_gather = ms_gather;
        for (int r = 0; r < N; r++) {
            stats->rays += rays[(size_t)r];
            stats->hitpoints += hps[(size_t)r];
            if (ms[(size_t)r] > stats->ms_render_slowest) stats->ms_render_slowest = ms[(size_t)r];
        }
    }
}

}  // namespace cgrt_host
#endif
