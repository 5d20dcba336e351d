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

// Everything render_sharded creates on the GPUs, released on every path out of it (an error on one GPU must not leak the
// buffers, streams and communicators of the others).
struct ShardedResources {
    std::vector<ncclComm_t> comms;
    std::vector<float *> d_rgb;        // [share]: on GPU share (emulation: all on GPU 0)
    std::vector<int> dev_of;           // device that owns d_rgb[share] / streams[share]
    std::vector<hipStream_t> streams;
    float *d_all = nullptr, *d_frame = nullptr;  // GPU 0: the N shares rank-major; the un-permuted frame
    hipEvent_t g0 = nullptr, g1 = nullptr;
    bool have_comms = false;
    ~ShardedResources() {
        for (size_t r = 0; r < d_rgb.size(); r++) {
            (void)hipSetDevice(dev_of[r]);
            if (d_rgb[r]) (void)hipFree(d_rgb[r]);
            if (r < streams.size() && streams[r]) (void)hipStreamDestroy(streams[r]);
            if (have_comms && r < comms.size() && comms[r]) (void)ncclCommDestroy(comms[r]);
        }
        (void)hipSetDevice(0);
        if (d_all) (void)hipFree(d_all);
        if (d_frame) (void)hipFree(d_frame);
        if (g0) (void)hipEventDestroy(g0);
        if (g1) (void)hipEventDestroy(g1);
    }
};

// image: [height][width][3] float, row 0 = bottom, as render() fills it.  n_gpus <= 0: every visible device.
// emulate_shares > 1 (verification aid for one-GPU machines): deal the stripes to that many shares, render them one after
// another on GPU 0 and run the same device un-permute -- everything of the N > 1 path except the RCCL transfers.
inline void render_sharded(const std::vector<Object *> &objs, const RenderParams &rp, std::vector<float> &image, int n_gpus = 0,
                           int stripe_rows = 8, ShardedStats *stats = nullptr, int emulate_shares = 0) {
    int ndev = 0;
    CGRT_HIP_OK(hipGetDeviceCount(&ndev));
    const bool emulate = emulate_shares > 1;
    const int N = emulate ? emulate_shares : ((n_gpus > 0 && n_gpus < ndev) ? n_gpus : ndev);
    if (N < 1 || ndev < 1) throw Error(CGRT_ERR_DEVICE, "no HIP device");
    if (stripe_rows <= 0 || stripe_rows % 8) throw Error(CGRT_ERR_INVALID, "stripe_rows must be a positive multiple of 8");
    const int W = rp.width, H = rp.height;
    const int nstripes = (H + stripe_rows - 1) / stripe_rows, per_rank = (nstripes + N - 1) / N;
    const int rows_local = N > 1 ? per_rank * stripe_rows : H;
    const size_t share = (size_t)rows_local * W * 3;

    ShardedResources R;
    R.comms.assign((size_t)N, nullptr);
    R.d_rgb.assign((size_t)N, nullptr);
    R.streams.assign((size_t)N, nullptr);
    R.dev_of.assign((size_t)N, 0);
    for (int r = 0; r < N; r++) R.dev_of[(size_t)r] = emulate ? 0 : r;
    if (N > 1 && !emulate) {
        std::vector<int> devs((size_t)N);
        for (int r = 0; r < N; r++) devs[(size_t)r] = r;
        CGRT_NCCL_OK(ncclCommInitAll(R.comms.data(), N, devs.data()));
        R.have_comms = true;
    }
    std::vector<std::string> errors((size_t)N);
    std::vector<double> ms((size_t)N, 0.0);
    std::vector<uint64_t> rays((size_t)N, 0), hps((size_t)N, 0);

    auto worker = [&](int r) {  // one host thread per GPU; everything it creates lives on that GPU
        try {
            const int dev = R.dev_of[(size_t)r];
            CGRT_HIP_OK(hipSetDevice(dev));
            CGRT_HIP_OK(hipStreamCreate(&R.streams[(size_t)r]));
            SceneBuilder sb;  // the scene is replicated: every GPU builds and uploads its own copy
            for (const Object *o : objs) o->add_to(sb);
            check(cgrt_scene_commit(sb.scene, dev));
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
            struct Tmp {  // this worker's temporaries, released on its error paths too
                uint64_t *d_cnt = nullptr;
                hipEvent_t e0 = nullptr, e1 = nullptr;
                ~Tmp() {
                    if (d_cnt) (void)hipFree(d_cnt);
                    if (e0) (void)hipEventDestroy(e0);
                    if (e1) (void)hipEventDestroy(e1);
                }
            } t;
            hipStream_t st = R.streams[(size_t)r];
            CGRT_HIP_OK(hipMalloc(&R.d_rgb[(size_t)r], share * sizeof(float)));
            CGRT_HIP_OK(hipMalloc(&t.d_cnt, CGRT_NCOUNTERS * sizeof(uint64_t)));
            CGRT_HIP_OK(hipMemsetAsync(t.d_cnt, 0, CGRT_NCOUNTERS * sizeof(uint64_t), st));
            CGRT_HIP_OK(hipEventCreate(&t.e0));
            CGRT_HIP_OK(hipEventCreate(&t.e1));
            CGRT_HIP_OK(hipEventRecord(t.e0, st));
            check(cgrt_trace_grid(sb.scene, &cam, &g, R.d_rgb[(size_t)r], nullptr, t.d_cnt, st));
            CGRT_HIP_OK(hipEventRecord(t.e1, st));
            CGRT_HIP_OK(hipStreamSynchronize(st));
            float el = 0;
            CGRT_HIP_OK(hipEventElapsedTime(&el, t.e0, t.e1));
            ms[(size_t)r] = el;
            uint64_t cnt[CGRT_NCOUNTERS];
            CGRT_HIP_OK(hipMemcpy(cnt, t.d_cnt, sizeof(cnt), hipMemcpyDeviceToHost));
            rays[(size_t)r] = cnt[CGRT_CNT_RAYS];
            hps[(size_t)r] = cnt[CGRT_CNT_HITPOINTS];
        } catch (const Error &e) {
            errors[(size_t)r] = e.what();
        }
    };
    if (emulate) {
        for (int r = 0; r < N; r++) worker(r);  // one GPU: the shares one after another
    } else {
        std::vector<std::thread> th;
        for (int r = 0; r < N; r++) th.emplace_back(worker, r);
        for (auto &t : th) t.join();
    }
    for (int r = 0; r < N; r++)
        if (!errors[(size_t)r].empty()) throw Error(CGRT_ERR_DEVICE, "GPU " + std::to_string(r) + ": " + errors[(size_t)r]);

    // ---- the gather: one group, N-1 receives on GPU 0 (N-1 distinct xGMI links), one send from every other GPU; then the
    // un-permute on GPU 0 (cgrt_unpermute_stripes: local row j of share r is global row ((j / S) * N + r) * S + j % S) and
    // ONE copy of the finished frame to the host ----
    image.assign((size_t)W * H * 3, 0.f);
    double ms_gather = 0;
    CGRT_HIP_OK(hipSetDevice(0));
    if (N > 1) {
        CGRT_HIP_OK(hipMalloc(&R.d_all, share * sizeof(float) * (size_t)N));
        CGRT_HIP_OK(hipMalloc(&R.d_frame, image.size() * sizeof(float)));
        CGRT_HIP_OK(hipEventCreate(&R.g0));
        CGRT_HIP_OK(hipEventCreate(&R.g1));
        CGRT_HIP_OK(hipEventRecord(R.g0, R.streams[0]));
        if (emulate) {
            for (int r = 1; r < N; r++)
                CGRT_HIP_OK(hipMemcpyAsync(R.d_all + share * (size_t)r, R.d_rgb[(size_t)r], share * sizeof(float), hipMemcpyDeviceToDevice,
                                           R.streams[0]));
        } else {
            CGRT_NCCL_OK(ncclGroupStart());
            for (int r = 1; r < N; r++) {
                CGRT_NCCL_OK(ncclRecv(R.d_all + share * (size_t)r, share, ncclFloat, r, R.comms[0], R.streams[0]));
                CGRT_NCCL_OK(ncclSend(R.d_rgb[(size_t)r], share, ncclFloat, 0, R.comms[(size_t)r], R.streams[(size_t)r]));
            }
            CGRT_NCCL_OK(ncclGroupEnd());
        }
        CGRT_HIP_OK(hipMemcpyAsync(R.d_all, R.d_rgb[0], share * sizeof(float), hipMemcpyDeviceToDevice, R.streams[0]));
        check(cgrt_unpermute_stripes(R.d_all, N, N, W, H, stripe_rows, rows_local, 3, R.d_frame, R.streams[0]));
        CGRT_HIP_OK(hipEventRecord(R.g1, R.streams[0]));
        for (int r = 0; r < N; r++) {
            CGRT_HIP_OK(hipSetDevice(R.dev_of[(size_t)r]));
            CGRT_HIP_OK(hipStreamSynchronize(R.streams[(size_t)r]));
        }
        float t = 0;
        CGRT_HIP_OK(hipSetDevice(0));
        CGRT_HIP_OK(hipEventElapsedTime(&t, R.g0, R.g1));
        ms_gather = t;
        CGRT_HIP_OK(hipMemcpy(image.data(), R.d_frame, image.size() * sizeof(float), hipMemcpyDeviceToHost));
    } else {
        CGRT_HIP_OK(hipMemcpy(image.data(), R.d_rgb[0], image.size() * sizeof(float), hipMemcpyDeviceToHost));
    }
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
