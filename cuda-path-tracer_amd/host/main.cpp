// Command line of the path tracer, same arguments and progress lines as the reference (src/main.cu:14-51):
//     pathtracer scene_file x_res y_res samples_per_pixel image_file
// Writes image_file + ".png".  Optional environment variables: PT_SEED (default 0), PT_DEVICE (default 0).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>
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
    const int device = std::getenv("PT_DEVICE") ? std::atoi(std::getenv("PT_DEVICE")) : 0;

    pth_scene *scene = nullptr;
    if (pth_scene_load(scene_file.c_str(), (uint32_t)width, (uint32_t)height, &scene) != 0) {
        std::cerr << "scene error: " << pth_last_error() << "\n";
        return 1;
    }
    pt_ctx *ctx = nullptr;
    if (pt_ctx_create(device, &ctx) != PT_OK) {
        std::cerr << "GPU error: " << pt_last_error(nullptr) << "\n";
        return 2;
    }
    auto check = [&](int rc) {
        if (rc != PT_OK) {
            std::cerr << "GPU error: " << pt_last_error(ctx) << "\n";
            std::exit(2);
        }
    };
    check(pt_scene_upload(ctx, pth_scene_desc(scene)));
    check(pt_film_bind(ctx, (uint32_t)width, (uint32_t)height, nullptr));

    std::cout << "Rendering image...\n";
    auto t1 = std::chrono::high_resolution_clock::now();
    pt_stats st{};
    check(pt_render(ctx, 0, spp, spp, seed, &st));
    auto t2 = std::chrono::high_resolution_clock::now();
    auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(t2 - t1);
    std::cout << "\rRendered image in " << ms.count() << " ms\n";
    std::printf("{\"rays\": %llu, \"paths\": %llu, \"iterations\": %llu, \"ms\": %.3f, \"mrays_per_s\": %.1f}\n",
                (unsigned long long)st.rays, (unsigned long long)st.paths, (unsigned long long)st.iterations, st.ms_total,
                st.ms_total > 0 ? st.rays / st.ms_total / 1e3 : 0.0);

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

    pt_ctx_destroy(ctx);
    pth_scene_free(scene);
    return 0;
}
