// Internal declarations shared by the host-side sources of libpt_host.so.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "pt_host.h"

namespace pth {

// Records the message for pth_last_error() and returns -1.
int set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));

int bvh_build(const float *verts, uint64_t ntris, int threads, pth_bvh *out);

int obj_load(const std::string &path, std::vector<float> &verts, uint64_t &ntris);

int hdr_load(const std::string &path, std::vector<float> &rgba, uint32_t &w, uint32_t &h);
int hdr_write(const std::string &path, const float *px, uint32_t w, uint32_t h, int channels);
int env_build(float *texels, uint32_t w, uint32_t h, uint64_t *marginal, uint64_t *conditional);

void camera_make(const float pos[3], const float look_at[3], float fov, float aspect, float aperture, float focus_dist,
                 pt_camera *out);

void tonemap16(const float *film, uint64_t npixels, float exposure, float gamma, uint16_t *out);
int png16_write(const std::string &path, const uint16_t *rgb, uint32_t w, uint32_t h);

int asset_generate(const std::string &kind, const std::string &path);

}   // namespace pth

// Owns every buffer a pt_scene_desc points to.
struct pth_scene {
    pt_scene_desc desc{};
    pth_bvh bvh{};
    std::vector<pt_plane> planes;
    std::vector<pt_sphere> spheres;
    std::vector<pt_mesh> meshes;
    std::vector<float> env;
    std::vector<uint64_t> marginal, conditional;
};
