/*
 * pt_host.h -- C ABI of libpt_host.so: the host-side scene assembly that feeds the device hot path.
 *
 * Everything here runs on the CPU and produces exactly the buffers pt_scene_upload (pt_api.h) consumes.  Each
 * entry point names the reference code (SuperMario210/cuda-path-tracer) it stands in for.  All functions return 0
 * on success and a negative value on failure; pth_last_error() gives the message of the calling thread's last
 * failure.  Buffers returned through out-pointers are released with the matching *_free call.
 */
#ifndef PT_HOST_H
#define PT_HOST_H

#include <stddef.h>
#include <stdint.h>

#include "pt_api.h"

#ifdef __cplusplus
extern "C" {
#endif

const char *pth_last_error(void);

/* ---- scene files ------------------------------------------------------------------------------------------ */
typedef struct pth_scene pth_scene;

/* Parses a .scene file, loads its .obj and .hdr, builds the BVH and the environment tables: the host half of
 * Scene::Scene (reference src/scene.cu:138-221).  Asset paths are tried relative to the working directory first
 * (as the reference does) and then relative to the scene file. */
int  pth_scene_load(const char *scene_file, uint32_t width, uint32_t height, pth_scene **out);
void pth_scene_free(pth_scene *scene);
const pt_scene_desc *pth_scene_desc(const pth_scene *scene);

/* ---- meshes ----------------------------------------------------------------------------------------------- */
/* v / f records of a Wavefront .obj (reference src/scene.cu:15-55); out_verts = ntris*9 floats. */
int  pth_obj_load(const char *path, float **out_verts, uint64_t *out_ntris);
void pth_free(void *p);

typedef struct {
    float *nodes;        uint64_t num_nodes;        /* GPUBVHNode[num_nodes], 16 floats each (src/bvh.cu:36-41) */
    float *tris;         uint64_t num_tri_float4;   /* Woop float4 stream (src/bvh.cu:214-227,240,275-277) */
    uint32_t *prim_ids;  uint64_t num_tris;         /* [num_tri_float4]: OBJ face index at each triangle's first float4 */
    uint64_t num_leaves;
    uint32_t max_depth;
} pth_bvh;

/* Binned-SAH build, flatten and Woop transform: BVH::build + BVH::send_to_device without the upload (reference
 * src/bvh.cu:100-294).  verts = ntris*9 floats in face order.  threads <= 1 builds on the calling thread. */
int  pth_bvh_build(const float *verts, uint64_t ntris, int threads, pth_bvh *out);
void pth_bvh_free(pth_bvh *bvh);

/* ---- environment maps ------------------------------------------------------------------------------------- */
/* Radiance .hdr (RGBE, run-length or flat) -> float4 rgba, alpha = 1: what stbi_loadf(..., 4) returns at
 * reference src/environment_map.cu:24. */
int  pth_hdr_load(const char *path, float **out_rgba, uint32_t *out_w, uint32_t *out_h);
int  pth_hdr_write(const char *path, const float *rgb, uint32_t w, uint32_t h, int channels);
/* EnvironmentMap::build_distribution (reference src/environment_map.cu:44-125): writes the pdf into texels[..].w
 * and fills the inverse-CDF lookup tables (marginal[w], conditional[w*h]). */
int  pth_env_build(float *texels_rgba, uint32_t w, uint32_t h, uint64_t *marginal, uint64_t *conditional);

/* ---- camera ----------------------------------------------------------------------------------------------- */
/* Camera::Camera (reference src/camera.cuh:21-37). */
void pth_camera_make(const float pos[3], const float look_at[3], float fov, float aspect, float aperture,
                     float focus_dist, pt_camera *out);

/* ---- image output ----------------------------------------------------------------------------------------- */
/* Image::tone_map + the 16-bit quantisation of Image::save_png (reference src/image.cpp:12-33). */
void pth_tonemap16(const float *film_rgb, uint64_t npixels, float exposure, float gamma, uint16_t *out_rgb);
/* 16-bit RGB PNG (reference src/image.cpp:35-50 writes the same pixel format through lodepng). */
int  pth_png16_write(const char *path, const uint16_t *rgb, uint32_t w, uint32_t h);
/* The linear film as a little-endian colour PFM ("PF", rows bottom to top): the exact floats, for parity tooling.
 * No reference counterpart (the reference only writes the tone-mapped PNG). */
int  pth_pfm_write(const char *path, const float *film_rgb, uint32_t w, uint32_t h);

/* ---- procedural stand-ins for assets the reference repository does not ship -------------------------------- */
/* kind: "studio.hdr", "sunset.hdr", "forest.hdr" (2048x1024 lat-long maps) or "dragon.obj" (868,352-triangle
 * displaced torus knot sized for scene/dragon.scene), "knot_small.obj" (a 9,216-triangle version for tests).
 * Deterministic: the same bytes on every run. */
int  pth_asset_generate(const char *kind, const char *path);

#ifdef __cplusplus
}
#endif
#endif
