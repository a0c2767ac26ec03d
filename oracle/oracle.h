/*
 * CPU ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A host restatement of the reference path tracer's hot path (BVH build and
 * flatten, closest-hit traversal, camera, environment map, shading, film
 * accumulation).  It exists only as the checker for the HIP path: tests, the
 * smoke check and the benchmark's cpu_baseline leg may load it; nothing that
 * ships may link, import or call it.
 *
 * Parity pins: (1) the random generator against the published Philox4x32-10
 * known-answer vectors; (2) the BVH builder/flattener against the node,
 * triangle-buffer and leaf counts recorded from the reference's own builder
 * for its four shipped meshes (reference src/bvh.cu:298,302) -- the FNV-1a-64
 * hashes recorded with those counts could NOT be reproduced; the buffer
 * contents are pinned instead by a six-triangle tree worked out by hand from
 * the text of reference src/bvh.cu:100-207,234-294 (tests/golden/
 * bvh_handworked.npz, derivation in tools/make_golden_bvh_handworked.py), which
 * the builder here reproduces byte for byte; (3) traversal + triangle test against an
 * independent float64 brute force and against closest hits worked out by
 * hand (tests/scenes.py, handworked_hits); (4) the RGBE pixel decode against the
 * reference's own vendored reader (stb_image, compiled from the reference
 * tree into oracle/_ref by `make ref`; frozen in tests/golden/hdr_decode.npz);
 * (5) the shading estimator against radiometry and optics: films whose
 * converged value is known in closed form or by float64 quadrature (a
 * Lambertian plane under a non-uniform sky, convex Lambertian / mirror /
 * glossy / clear-glass bodies in a constant sky, a camera between two mirrors,
 * a glass plane whose pixels follow Schlick's reflectance and Snell's angle:
 * tests/scenes.py furnace_cases, fresnel_case), a film, the environment
 * lookup tables of a 4 x 2 map and a camera frame worked out by hand.
 * sin / cos / atan2 / acos / x^5 of the shading path are evaluated by fixed polynomial kernels (stated here and,
 * separately, in the device code) instead of a math library's, so that host and device take every random decision
 * alike; their error against double precision is held to 3 ulp by the tests.
 * Everything else on the device side
 * (visit order details, individual shading decisions, the texture filter) is "parity unpinned":
 * the reference ships no tests, fixtures or golden outputs for it and cannot
 * be built here (CUDA toolkit, cuRAND, inline PTX), so it is restated from
 * the source text alone.
 *
 * Every struct here is laid out exactly like its pt_* twin in include/pt_api.h
 * so one ctypes definition can feed both libraries the same bytes.
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* material_type flag values: reference src/render.cuh:17-20 */
#define ORC_DIFFUSE 0x4u
#define ORC_MIRROR  0x8u
#define ORC_GLOSSY  0x10u
#define ORC_GLASS   0x20u

/* primitive id encoding of a closest hit */
#define ORC_PRIM_MISS   0xFFFFFFFFu
#define ORC_PRIM_PLANE  0x80000000u /* | plane index  */
#define ORC_PRIM_SPHERE 0x40000000u /* | sphere index */

typedef struct {            /* reference src/object.cuh:47-51 */
    float position[3];
    float normal[3];        /* already normalised (object.cuh:55) */
    float material[4];
    uint32_t material_type;
} orc_plane;

typedef struct {            /* reference src/object.cuh:13-18 */
    float position[3];
    float radius;
    float material[4];
    uint32_t material_type;
} orc_sphere;

typedef struct {            /* reference src/camera.cuh:14-18 */
    float origin[3];
    float bottom_left[3];
    float right[3];
    float up[3];
    float u[3];
    float v[3];
    float w[3];
    float lens_radius;
} orc_camera;

typedef struct {            /* one OBJ block; a superset of the reference's single mesh (src/scene.cu:186-200) */
    float material[4];
    uint32_t material_type;
    uint32_t first_face;    /* prim id of the mesh's first triangle, ascending over the array */
} orc_mesh;

typedef struct {
    const float *bvh_nodes;          /* 16 floats per node, reference src/bvh.cu:36-41 */
    uint64_t num_nodes;
    const float *bvh_tris;           /* float4 stream, reference src/bvh.cu:224-226,240,275-277 */
    uint64_t num_tri_float4;
    const uint32_t *bvh_prim_ids;    /* [num_tri_float4]: id of the triangle whose first float4 sits at that index */
    uint64_t num_tris;
    float mesh_material[4];
    uint32_t mesh_material_type;
    uint32_t reserved0;
    const orc_plane *planes;
    uint64_t num_planes;
    const orc_sphere *spheres;
    uint64_t num_spheres;
    orc_camera camera;
    uint32_t env_width;
    uint32_t env_height;
    const float *env_texels;         /* float4 [height][width], row 0 = top, .w = solid-angle pdf */
    const uint64_t *env_marginal;    /* [width] */
    const uint64_t *env_conditional; /* [width*height], index x*height + y, y = 0 is the bottom row */
    const orc_mesh *meshes;          /* NULL: one mesh with mesh_material; else a triangle with prim id p has the material */
    uint64_t num_meshes;             /* of the last mesh whose first_face <= p */
} orc_scene_desc;

typedef struct {
    float *nodes;        uint64_t num_nodes;       /* 16 floats each */
    float *tris;         uint64_t num_tri_float4;  /* 4 floats each, 2 float4 of zero padding allocated past the end */
    uint32_t *prim_ids;  uint64_t num_tris;        /* [num_tri_float4] */
    uint64_t num_leaves;
    uint32_t max_depth;
} orc_bvh;

typedef struct {
    uint64_t rays;       /* closest-hit queries */
    uint64_t nodes;      /* inner nodes visited */
    uint64_t tris;       /* triangles tested */
    uint64_t leaves;     /* leaves entered */
    uint64_t paths;      /* camera paths started */
    double   ms;         /* wall time of the call */
} orc_stats;

uint64_t orc_fnv1a64(const void *data, uint64_t nbytes);

/* verts: ntris*9 floats (v0 v1 v2 per triangle, OBJ face order).  Follows reference src/bvh.cu:100-294. */
int  orc_bvh_build(const float *verts, uint64_t ntris, orc_bvh *out);
void orc_bvh_free(orc_bvh *bvh);

/* texels: float4 [H][W]; .w is overwritten with the pdf.  Follows reference src/environment_map.cu:44-125. */
int  orc_env_build(float *texels, uint32_t W, uint32_t H, uint64_t *marginal, uint64_t *conditional);

/* Follows reference src/camera.cuh:21-37. */
void orc_camera_make(const float pos[3], const float look_at[3], float fov, float aspect, float aperture,
                     float focus_dist, orc_camera *out);

/* Philox4x32-10 block, for pinning the generator against its published test vectors. */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

/* Closest hit for n rays.  org/dir: n*3 floats.  normal_out (n*3, normalised) and stats may be NULL.
 * Follows reference src/render.cu:11-44, src/object.cuh:27-75, src/bvh.cuh:75-189. */
int  orc_intersect(const orc_scene_desc *scene, uint64_t n, const float *org, const float *dir,
                   float *t_out, uint32_t *prim_out, float *normal_out, orc_stats *stats);

/* Camera rays exactly as the render loop generates them: pixel-major order, sample index s. */
int  orc_primary_rays(const orc_scene_desc *scene, uint32_t W, uint32_t H, uint64_t sample, uint64_t seed,
                      float *org, float *dir);

/* Accumulates samples [spp_begin, spp_begin+spp_count) of spp_total into film (W*H*3 floats, row 0 = top).
 * Follows reference src/render.cu:49-242 with the counter-based generator in place of cuRAND. */
int  orc_render(const orc_scene_desc *scene, uint32_t W, uint32_t H, uint64_t spp_begin, uint64_t spp_count,
                uint64_t spp_total, uint64_t seed, float *film, int nthreads, orc_stats *stats);

/* Follows reference src/image.cpp:12-33: 1-exp(-c*exposure), ^(1/gamma), truncation to 16 bits. */
void orc_tonemap16(const float *film, uint64_t npixels, float exposure, float gamma, uint16_t *out_rgb);

/* Radiance RGBE pixel -> float, as stb_image's stbi__hdr_convert does (reference include/stb_image.h:6930-6955). */
void orc_rgbe_to_float(const uint8_t *rgbe, uint64_t npixels, float *out_rgba);

/* The shading path's transcendental functions (fixed polynomial kernels, see oracle.cpp), for checking their error:
 * which = 0 sin and cos of a[i] in [0, 2 pi] -> out0, out1; 1 atan2(a[i], b[i]) -> out0; 2 acos(a[i]) -> out0;
 * 3 a[i]^5 -> out0. */
void orc_math(int which, uint64_t n, const float *a, const float *b, float *out0, float *out1);

#ifdef __cplusplus
}
#endif
#endif
