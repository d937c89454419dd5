// raytrace_hip -- the RayTrace program on top of the C ABIs: read a scene file in the
// reference's format, build the kd-tree on the host, upload the scene to every GPU, render
// interleaved tiles (one host thread and one kernel launch per GPU), gather the GPUs' packed
// shares on GPU 0 over RCCL, tone-map and write a BMP.
//
// It takes the place of the reference's main (raytrace.cpp:97-248): the scene-file grammar, the
// camera, the per-pixel arithmetic and drawPixel are the reference's; the MPI row-band split with
// its MPI_Isend / MPI_Recv exchange (raytrace.cpp:179, 211, 219-227) became disjoint tiles and one
// grouped RCCL exchange (include/rt_gather.h), and the OpenCV window is gone (output is headless).
//
// usage: raytrace_hip [scene.txt] [-o output.bmp] [--samps N] [--depth N] [--rr 0|1] [--seed N]
//                     [--gpus N] [--tile N] [--gather rccl|copy] [--obj-flags N] [--lights 0|1]
//   --lights 1      explicit light sampling (rt_params.light_mode; the reference has no such mode)
//   --gather copy   collects the shares with one device-to-host copy per GPU instead of the RCCL exchange
//   --obj-flags     RTH_OBJ_* bits for the OBJ reader (1: triangulate polygons, 2: negative indices)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "rt_gather.h"
#include "rt_hip.h"
#include "rt_host.h"

namespace {
// All GPU threads meet here before the exchange: a collective that one rank never joins would hang the others.
struct Meet {
    std::mutex mu;
    std::condition_variable cv;
    int waiting = 0, round = 0, n = 0;
    bool failed = false;
    bool arrive(bool ok) { // returns true if every thread arrived without a failure
        std::unique_lock<std::mutex> lk(mu);
        if (!ok) failed = true;
        const int my = round;
        if (++waiting == n) { waiting = 0; round++; cv.notify_all(); }
        else cv.wait(lk, [&] { return round != my; });
        return !failed;
    }
};
} // namespace

int main(int argc, char **argv) {
    std::string scene_path = "test.txt", out_path = "output.bmp"; // raytrace.cpp:106, 234
    int samps = -1, depth = 5, rr = 1, gpus = -1, tile = 64;
    unsigned obj_flags = 0;
    int light_mode = 0;
    bool use_rccl = true;
    unsigned long long seed = 1;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto next = [&](const char *what) -> const char * {
            if (i + 1 >= argc) { fprintf(stderr, "%s needs a value\n", what); exit(2); }
            return argv[++i];
        };
        if (a == "-o") out_path = next("-o");
        else if (a == "--samps") samps = atoi(next("--samps"));
        else if (a == "--depth") depth = atoi(next("--depth"));
        else if (a == "--rr") rr = atoi(next("--rr"));
        else if (a == "--seed") seed = strtoull(next("--seed"), nullptr, 10);
        else if (a == "--gpus") gpus = atoi(next("--gpus"));
        else if (a == "--tile") tile = atoi(next("--tile"));
        else if (a == "--gather") use_rccl = std::string(next("--gather")) != "copy";
        else if (a == "--obj-flags") obj_flags = (unsigned)atoi(next("--obj-flags"));
        else if (a == "--lights") light_mode = atoi(next("--lights"));
        else scene_path = a;
    }
    const auto t0 = std::chrono::steady_clock::now();
    rth_scene *hs = rth_scene_new();
    rth_settings st;
    if (rth_load_scene_file_ex(hs, scene_path.c_str(), &st, obj_flags) < 0 || rth_build(hs) < 0) {
        fprintf(stderr, "%s\n", rth_last_error());
        return 1;
    }
    const rt_scene_desc *desc = rth_desc(hs);
    int ndev = rt_device_count();
    if (ndev < 1) { fprintf(stderr, "no GPU: %s\n", rt_last_error()); return 1; }
    if (gpus > 0 && gpus < ndev) ndev = gpus;
    if (samps < 1) samps = st.g_samps > 0 ? st.g_samps : st.c_samps; // raytrace.cpp:160-161
    printf("%d triangles, %d spheres, %d tree nodes (depth %d); %d x %d, %d samples per sub-pixel, %d GPU(s)\n", desc->n_tris,
           desc->n_spheres, desc->n_nodes, desc->tree_depth, st.sizex, st.sizey, samps, ndev);

    rt_camera cam;
    for (int k = 0; k < 3; k++) { cam.o[k] = st.view[k]; cam.d[k] = st.dir[k]; }
    cam.sizex = st.sizex; cam.sizey = st.sizey;
    rt_params prm = {samps, depth, rr, 2, 100000, 1, seed, light_mode, 0};

    std::vector<rt_tile> tiles;
    for (int x0 = 0; x0 < st.sizex; x0 += tile)
        for (int y0 = 0; y0 < st.sizey; y0 += tile)
            tiles.push_back(rt_tile{x0, y0, std::min(tile, st.sizex - x0), std::min(tile, st.sizey - y0)});

    std::vector<double> frame((size_t)st.sizex * st.sizey * 3, 0.0);
    std::vector<rt_stats> stats((size_t)ndev);
    std::vector<int> status((size_t)ndev, 0);
    std::vector<std::string> errors((size_t)ndev);
    // tile i belongs to GPU i mod ndev; a GPU's share is its tiles back to back
    std::vector<std::vector<rt_tile>> share((size_t)ndev);
    std::vector<uint64_t> counts((size_t)ndev, 0); // doubles per share
    for (size_t i = 0; i < tiles.size(); i++) share[i % (size_t)ndev].push_back(tiles[i]);
    uint64_t total = 0;
    for (int g = 0; g < ndev; g++) {
        for (const rt_tile &t : share[g]) counts[g] += (uint64_t)t.rows * t.cols * 3;
        total += counts[g];
    }
    // the device layout (culling hierarchies and all) is built once, on the host, and uploaded by every GPU's thread
    rt_prepared *prep = nullptr;
    if (rt_prepare_scene(desc, &prep) < 0) { fprintf(stderr, "%s\n", rt_last_error()); return 1; }
    rt_gather *xg = nullptr;
    if (use_rccl) { // raytrace.cpp:99-104: the group of ranks
        std::vector<int32_t> devs((size_t)ndev);
        for (int g = 0; g < ndev; g++) devs[g] = g;
        if (rt_gather_create(ndev, devs.data(), &xg) < 0) { fprintf(stderr, "%s\n", rt_gather_last_error()); return 1; }
    }
    std::vector<double> gathered(use_rccl ? (size_t)total : 0); // all shares in rank order, as they arrive on GPU 0
    std::vector<std::vector<double>> copied((size_t)ndev);       // --gather copy: one host buffer per GPU
    Meet meet;
    meet.n = ndev;
    std::vector<std::thread> workers;
    for (int g = 0; g < ndev; g++)
        workers.emplace_back([&, g]() {
            char info[512];
            if (rt_device_info(g, info, sizeof(info)) == 0) printf("%s\n", info); // raytrace.cpp:164
            rt_scene *ds = nullptr;
            double *d_rgb = nullptr, *d_all = nullptr;
            int rc = rt_upload_prepared(g, prep, &ds);
            if (rc == 0 && hipMalloc((void **)&d_rgb, std::max<uint64_t>(counts[g], 1) * sizeof(double)) != hipSuccess) rc = RT_E_NOMEM;
            if (rc == 0 && use_rccl && g == 0 && hipMalloc((void **)&d_all, total * sizeof(double)) != hipSuccess) rc = RT_E_NOMEM;
            if (rc == 0 && !share[g].empty())
                rc = rt_render_tiles_device(ds, &cam, share[g].data(), (int)share[g].size(), &prm, d_rgb, nullptr, nullptr);
            if (rc == 0) rc = rt_get_stats(ds, &stats[g]);
            if (rc != 0) { status[g] = rc; errors[g] = rt_last_error(); }
            const bool all_ok = meet.arrive(rc == 0);
            if (all_ok && use_rccl) { // raytrace.cpp:211, 219-227: every rank's share to rank 0
                rc = rt_gather_shares(xg, g, d_rgb, d_all, counts.data(), nullptr);
                if (rc != 0) { status[g] = rc; errors[g] = rt_gather_last_error(); }
                // A rank whose half of the exchange was never queued leaves the others waiting for it for ever:
                // all meet again, and if any failed the process ends here, naming the rank, instead of hanging in
                // a synchronisation on a collective that cannot complete.
                if (!meet.arrive(rc == 0)) {
                    if (rc != 0) fprintf(stderr, "GPU %d failed in the exchange: %s\n", g, errors[g].c_str());
                    fflush(stderr);
                    std::_Exit(1);
                }
                if (hipDeviceSynchronize() != hipSuccess) rc = RT_E_DEVICE;
                if (rc == 0 && g == 0 && hipMemcpy(gathered.data(), d_all, total * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) rc = RT_E_DEVICE;
                if (rc != 0) { status[g] = rc; errors[g] = rc == RT_E_DEVICE && !*rt_gather_last_error() ? "HIP error after the exchange" : rt_gather_last_error(); }
            } else if (all_ok) {
                copied[g].resize((size_t)counts[g]);
                if (counts[g] && hipMemcpy(copied[g].data(), d_rgb, counts[g] * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) {
                    status[g] = RT_E_DEVICE; errors[g] = "device-to-host copy failed";
                }
            }
            if (d_rgb) (void)hipFree(d_rgb);
            if (d_all) (void)hipFree(d_all);
            rt_free_scene(ds);
        });
    for (auto &w : workers) w.join();
    rt_free_prepared(prep);
    if (xg) rt_gather_destroy(xg);
    for (int g = 0; g < ndev; g++)
        if (status[g]) { fprintf(stderr, "GPU %d failed: %s\n", g, errors[g].c_str()); return 1; }
    { // un-permute: rank order, tiles back to back, rows inside a tile
        size_t off = 0;
        for (int g = 0; g < ndev; g++) {
            const double *src = use_rccl ? gathered.data() + off : copied[g].data();
            size_t at = 0;
            for (const rt_tile &t : share[g])
                for (int r = 0; r < t.rows; r++) {
                    memcpy(&frame[((size_t)(t.x0 + r) * st.sizey + t.y0) * 3], src + at, (size_t)t.cols * 3 * sizeof(double));
                    at += (size_t)t.cols * 3;
                }
            off += (size_t)counts[g];
        }
    }
    unsigned long long rays = 0;
    for (int g = 0; g < ndev; g++) rays += stats[g].rays;
    std::vector<uint8_t> bgr((size_t)st.sizex * st.sizey * 3);
    rth_tonemap(frame.data(), (size_t)st.sizex * st.sizey, bgr.data());
    printf("Save image to %s\n", out_path.c_str());
    if (rth_write_bmp(out_path.c_str(), bgr.data(), st.sizex, st.sizey) < 0) { fprintf(stderr, "%s\n", rth_last_error()); return 1; }
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("%llu rays, %lfs\n", rays, secs); // raytrace.cpp:235
    rth_scene_free(hs);
    return 0;
}
