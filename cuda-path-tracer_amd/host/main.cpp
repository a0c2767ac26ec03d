// Command line of the path tracer, same arguments and progress lines as the reference (src/main.cu:14-51):
//     pathtracer scene_file x_res y_res samples_per_pixel image_file
// Writes image_file + ".png".  Optional environment variables: PT_SEED (default 0), PT_DEVICE (default 0),
// PT_DEVICES = a count ("8": devices 0..7) or a list ("0,2,5"): the samples are split evenly over these GPUs, one host
// thread and one context each, and the films are summed on the first (pt_film_reduce); the reference renders on one GPU.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>
#include <thread>
#include <vector>

#include "pt_api.h"
#include "pt_host.h"

#define EXPOSURE 2.0f   // reference src/main.cu:11
#define GAMMA 2.2f      // reference src/main.cu:12

int main(int argc, char **argv)
{
    if (argc != 6) {
        std::cout << "usage: ./pathtracer scene_file x_res y_res samples_per_pixel image_file\n";
        return 0;
    }
    const std::string scene_file = argv[1];
    const size_t width = std::stol(argv[2]);
    const size_t height = std::stol(argv[3]);
    const size_t spp = std::stol(argv[4]);
    const std::string image_file = argv[5];
    const uint64_t seed = std::getenv("PT_SEED") ? std::strtoull(std::getenv("PT_SEED"), nullptr, 0) : 0;
    std::vector<int> devices;
    if (const char *e = std::getenv("PT_DEVICES")) {
        const std::string list = e;
        if (list.find(',') == std::string::npos) {
            for (int i = 0; i < std::atoi(e); i++) devices.push_back(i);
        } else {
            for (size_t at = 0; at < list.size();) {
                const size_t comma = std::min(list.find(',', at), list.size());
                if (comma > at) devices.push_back(std::atoi(list.substr(at, comma - at).c_str()));
                at = comma + 1;
            }
        }
    }
    if (devices.empty()) devices.push_back(std::getenv("PT_DEVICE") ? std::atoi(std::getenv("PT_DEVICE")) : 0);
    const size_t G = devices.size();

    pth_scene *scene = nullptr;
    if (pth_scene_load(scene_file.c_str(), (uint32_t)width, (uint32_t)height, &scene) != 0) {
        std::cerr << "scene error: " << pth_last_error() << "\n";
        return 1;
    }
    std::vector<pt_ctx *> ctxs(G, nullptr);
    for (size_t g = 0; g < G; g++) {
        if (pt_ctx_create(devices[g], &ctxs[g]) != PT_OK) {
            std::cerr << "GPU error: " << pt_last_error(nullptr) << "\n";
            return 2;
        }
    }
    pt_ctx *ctx = ctxs[0];
    auto check_on = [&](pt_ctx *c, int rc) {
        if (rc != PT_OK) {
            std::cerr << "GPU error: " << pt_last_error(c) << "\n";
            std::exit(2);
        }
    };
    auto check = [&](int rc) { check_on(ctx, rc); };
    for (pt_ctx *c : ctxs) {
        check_on(c, pt_scene_upload(c, pth_scene_desc(scene)));
        check_on(c, pt_film_bind(c, (uint32_t)width, (uint32_t)height, nullptr));
    }

    std::cout << "Rendering image...\n";
    auto t1 = std::chrono::high_resolution_clock::now();
    // GPU g renders samples [g * spp / G, (g + 1) * spp / G) of every pixel, already weighted 1 / spp
    std::vector<pt_stats> stats(G);
    std::vector<int> status(G, PT_OK);
    std::vector<std::thread> workers;
    for (size_t g = 1; g < G; g++)
        workers.emplace_back([&, g] {
            status[g] = pt_render(ctxs[g], spp * g / G, spp * (g + 1) / G - spp * g / G, spp, seed, &stats[g]);
        });
    status[0] = pt_render(ctx, 0, spp / G, spp, seed, &stats[0]);
    for (auto &w : workers) w.join();
    for (size_t g = 0; g < G; g++) check_on(ctxs[g], status[g]);
    if (G > 1) check(pt_film_reduce(ctxs.data(), (int)G, 0));
    auto t2 = std::chrono::high_resolution_clock::now();
    auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(t2 - t1);
    std::cout << "\rRendered image in " << ms.count() << " ms\n";
    pt_stats st{};
    for (const pt_stats &r : stats) {
        st.rays += r.rays; st.paths += r.paths; st.iterations += r.iterations;
        st.ms_total = std::max(st.ms_total, r.ms_total);
    }
    std::printf("{\"gpus\": %zu, \"rays\": %llu, \"paths\": %llu, \"iterations\": %llu, \"ms\": %.3f, \"mrays_per_s\": %.1f, "
                "\"render_mode\": %d}\n",
                G, (unsigned long long)st.rays, (unsigned long long)st.paths, (unsigned long long)st.iterations, st.ms_total,
                st.ms_total > 0 ? st.rays / st.ms_total / 1e3 : 0.0, pt_render_mode_used(ctx));

    if (const char *e = std::getenv("PT_PFM")) {        // PT_PFM=1: also dump the linear film as image_file.pfm
        if (e[0] == '1') {
            std::vector<float> film(width * height * 3);
            check(pt_film_download(ctx, film.data()));
            if (pth_pfm_write((image_file + ".pfm").c_str(), film.data(), (uint32_t)width, (uint32_t)height) != 0)
                std::cerr << "Error writing PFM: " << pth_last_error() << "\n";
        }
    }

    std::cout << "Saving image...\n";
    t1 = std::chrono::high_resolution_clock::now();
    std::vector<uint16_t> rgb16(width * height * 3);
    check(pt_film_tonemap16(ctx, EXPOSURE, GAMMA, rgb16.data()));      // tone map on the device, 6 B per pixel come back
    if (pth_png16_write((image_file + ".png").c_str(), rgb16.data(), (uint32_t)width, (uint32_t)height) != 0)
        std::cerr << "Error encoding PNG: " << pth_last_error() << "\n";
    t2 = std::chrono::high_resolution_clock::now();
    ms = std::chrono::duration_cast<std::chrono::milliseconds>(t2 - t1);
    std::cout << "\rSaved image in " << ms.count() << " ms\n";

    for (pt_ctx *c : ctxs) pt_ctx_destroy(c);
    pth_scene_free(scene);
    return 0;
}
