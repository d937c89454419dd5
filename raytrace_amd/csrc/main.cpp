// raytrace_hip -- the RayTrace program on top of the two C ABIs: read a scene file in the
// reference's format, build the kd-tree on the host, upload the scene to every GPU, render
// interleaved tiles (one host thread and one kernel launch per GPU), tone-map and write a BMP.
//
// It takes the place of the reference's main (raytrace.cpp:97-248): the scene-file grammar, the
// camera, the per-pixel arithmetic and drawPixel are the reference's; the MPI row-band split and
// the OpenCV window are gone (tiles are disjoint, output is headless).  This single-process
// driver collects the GPUs' tile buffers with device-to-host copies; the one-process-per-GPU
// driver that assembles the frame with an RCCL gather is raytrace_amd/frame.py.
//
// usage: raytrace_hip [scene.txt] [-o output.bmp] [--samps N] [--depth N] [--rr 0|1] [--seed N]
//                     [--gpus N] [--tile N]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "rt_hip.h"
#include "rt_host.h"

int main(int argc, char **argv) {
    std::string scene_path = "test.txt", out_path = "output.bmp"; // raytrace.cpp:106, 234
    int samps = -1, depth = 5, rr = 1, gpus = -1, tile = 64;
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
        else scene_path = a;
    }
    const auto t0 = std::chrono::steady_clock::now();
    rth_scene *hs = rth_scene_new();
    rth_settings st;
    if (rth_load_scene_file(hs, scene_path.c_str(), &st) < 0 || rth_build(hs) < 0) {
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
    rt_params prm = {samps, depth, rr, 2, 100000, 1, seed};

    std::vector<rt_tile> tiles;
    for (int x0 = 0; x0 < st.sizex; x0 += tile)
        for (int y0 = 0; y0 < st.sizey; y0 += tile)
            tiles.push_back(rt_tile{x0, y0, std::min(tile, st.sizex - x0), std::min(tile, st.sizey - y0)});

    std::vector<double> frame((size_t)st.sizex * st.sizey * 3, 0.0);
    std::vector<rt_stats> stats((size_t)ndev);
    std::vector<int> status((size_t)ndev, 0);
    std::vector<std::string> errors((size_t)ndev);
    std::vector<std::thread> workers;
    for (int g = 0; g < ndev; g++)
        workers.emplace_back([&, g]() {
            char info[512];
            if (rt_device_info(g, info, sizeof(info)) == 0) printf("%s\n", info); // raytrace.cpp:164
            std::vector<rt_tile> mine;
            for (size_t i = g; i < tiles.size(); i += (size_t)ndev) mine.push_back(tiles[i]);
            if (mine.empty()) return;
            size_t npix = 0;
            for (const rt_tile &t : mine) npix += (size_t)t.rows * t.cols;
            rt_scene *ds = nullptr;
            double *d_rgb = nullptr;
            std::vector<double> packed(npix * 3);
            int rc = rt_upload_scene(g, desc, &ds);
            if (rc == 0 && hipMalloc((void **)&d_rgb, npix * 3 * sizeof(double)) != hipSuccess) rc = RT_E_NOMEM;
            if (rc == 0) rc = rt_render_tiles_device(ds, &cam, mine.data(), (int)mine.size(), &prm, d_rgb, nullptr, nullptr);
            if (rc == 0) rc = rt_get_stats(ds, &stats[g]);
            if (rc == 0 && hipMemcpy(packed.data(), d_rgb, npix * 3 * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) rc = RT_E_DEVICE;
            if (rc != 0) { status[g] = rc; errors[g] = rt_last_error(); }
            else {
                size_t off = 0;
                for (const rt_tile &t : mine)
                    for (int r = 0; r < t.rows; r++) {
                        memcpy(&frame[((size_t)(t.x0 + r) * st.sizey + t.y0) * 3], &packed[off], (size_t)t.cols * 3 * sizeof(double));
                        off += (size_t)t.cols * 3;
                    }
            }
            if (d_rgb) (void)hipFree(d_rgb);
            rt_free_scene(ds);
        });
    for (auto &w : workers) w.join();
    unsigned long long rays = 0;
    for (int g = 0; g < ndev; g++) {
        if (status[g]) { fprintf(stderr, "GPU %d failed: %s\n", g, errors[g].c_str()); return 1; }
        rays += stats[g].rays;
    }
    std::vector<uint8_t> bgr((size_t)st.sizex * st.sizey * 3);
    rth_tonemap(frame.data(), (size_t)st.sizex * st.sizey, bgr.data());
    printf("Save image to %s\n", out_path.c_str());
    if (rth_write_bmp(out_path.c_str(), bgr.data(), st.sizex, st.sizey) < 0) { fprintf(stderr, "%s\n", rth_last_error()); return 1; }
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("%llu rays, %lfs\n", rays, secs); // raytrace.cpp:235
    rth_scene_free(hs);
    return 0;
}
