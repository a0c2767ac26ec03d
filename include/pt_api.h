/*
 * pt_api.h -- C ABI of libpt_hip.so, the MI355X (gfx950) wavefront path-tracing hot path.
 *
 * Plain C: opaque handle, plain pointers and sizes, int status returns (0 = ok, negative = error; the text is
 * available from pt_last_error).  The caller owns every host buffer; the library owns every device buffer
 * except a film the caller binds explicitly.  One pt_ctx per device; calls on one ctx must be serialised by
 * the caller, different ctxs may be driven from different host threads.
 *
 * Each entry point names the reference interface (SuperMario210/cuda-path-tracer) it stands in for.
 */
#ifndef PT_API_H
#define PT_API_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* status codes */
#define PT_OK             0
#define PT_ERR_ARG       (-1)   /* bad argument / scene the traversal cannot represent */
#define PT_ERR_HIP       (-2)   /* a HIP runtime call failed */
#define PT_ERR_STATE     (-3)   /* call order (no scene uploaded, no film bound, ...) */
#define PT_ERR_NOMEM     (-4)

/* material_type values: reference src/render.cuh:17-20 (IS_DIFFUSE .. IS_GLASS) */
#define PT_DIFFUSE 0x4u
#define PT_MIRROR  0x8u
#define PT_GLOSSY  0x10u
#define PT_GLASS   0x20u

/* Primitive id of a closest hit.  Triangles report the id given in bvh_prim_ids (the OBJ face index). */
#define PT_PRIM_MISS   0xFFFFFFFFu
#define PT_PRIM_PLANE  0x80000000u   /* | index into planes  */
#define PT_PRIM_SPHERE 0x40000000u   /* | index into spheres */

typedef struct {                /* reference src/object.cuh:47-51 (class Plane) */
    float position[3];
    float normal[3];            /* unit length, as Plane's constructor leaves it (object.cuh:55) */
    float material[4];          /* albedo rgb, w = IOR / glossyness (src/scene.cu:96-136) */
    uint32_t material_type;
} pt_plane;

typedef struct {                /* reference src/object.cuh:13-18 (class Sphere) */
    float position[3];
    float radius;
    float material[4];
    uint32_t material_type;
} pt_sphere;

typedef struct {                /* reference src/camera.cuh:14-18 (class Camera, as uploaded at src/scene.cu:160-162) */
    float origin[3];
    float bottom_left[3];
    float right[3];
    float up[3];
    float u[3];
    float v[3];
    float w[3];
    float lens_radius;
} pt_camera;

typedef struct {                /* one OBJ block of a .scene file.  The reference allows a single one per scene (src/scene.cu:
                                   186-200, one material per BVH: src/bvh.cuh:64-65); several are a superset of it. */
    float material[4];
    uint32_t material_type;
    uint32_t first_face;        /* bvh_prim_ids value of the mesh's first triangle; meshes are listed in ascending order */
} pt_mesh;

/*
 * Host-side description of a scene: exactly the buffers the reference's Scene constructor uploads
 * (src/scene.cu:138-221), as host pointers.
 *   bvh_nodes        GPUBVHNode[num_nodes], 64 B each (src/bvh.cu:36-41), as memcpy'd at src/bvh.cu:302
 *   bvh_tris         float4[num_tri_float4] Woop rows with 0x80000000 leaf terminators and the zero float4 at
 *                    index 0 (src/bvh.cu:224-226,240,275-277), as memcpy'd at src/bvh.cu:298
 *   bvh_prim_ids     uint32[num_tri_float4] or NULL: id reported for the triangle whose first float4 sits at that
 *                    index (NULL: the float4 index itself is reported)
 *   env_texels       float4[env_height][env_width], row 0 = top of the .hdr, .w = solid-angle pdf
 *                    (src/environment_map.cu:88-104,142-147)
 *   env_marginal     size_t[env_width]; env_conditional size_t[env_width*env_height], index x*height+y
 *                    (src/environment_map.cu:107-134)
 *   meshes           pt_mesh[num_meshes] or NULL.  NULL / 0: every triangle has mesh_material / mesh_material_type (the
 *                    reference's case).  Otherwise ONE BVH over the triangles of all meshes; a triangle with prim id p has
 *                    the material of the last mesh whose first_face <= p (meshes[0].first_face must be 0), and
 *                    mesh_material / mesh_material_type are ignored.  Needs bvh_prim_ids.
 * A scene without a mesh passes bvh_nodes = NULL, num_nodes = 0.
 */
typedef struct {
    const float *bvh_nodes;
    uint64_t num_nodes;
    const float *bvh_tris;
    uint64_t num_tri_float4;
    const uint32_t *bvh_prim_ids;
    uint64_t num_tris;
    float mesh_material[4];
    uint32_t mesh_material_type;
    uint32_t reserved0;
    const pt_plane *planes;
    uint64_t num_planes;
    const pt_sphere *spheres;
    uint64_t num_spheres;
    pt_camera camera;
    uint32_t env_width;
    uint32_t env_height;
    const float *env_texels;
    const uint64_t *env_marginal;
    const uint64_t *env_conditional;
    const pt_mesh *meshes;
    uint64_t num_meshes;
} pt_scene_desc;

typedef struct {
    uint64_t rays;              /* closest-hit queries: active slots summed over extend launches + rays_escaped */
    uint64_t paths;             /* camera paths started */
    uint64_t iterations;        /* wavefront iterations (one extend launch each) */
    uint64_t extend_launches;   /* extend launches that had at least one ray */
    double ms_total;            /* HIP-event time of the whole call on the ctx stream */
    double ms_extend;           /* sum of HIP-event times of the extend launches */
    uint64_t nodes;             /* traversal counters; filled only when PT_OPT_COUNT_TRAVERSAL is set */
    uint64_t tris;
    uint64_t leaves;
    double ms_extend_busy;      /* time during which at least one extend launch was running (extend launches of
                                   different streams overlap, so this is below ms_extend when PT_OPT_STREAMS > 1) */
    uint64_t rays_escaped;      /* queries answered where the ray was created, by the traversal's own first step: rays that
                                   miss both child boxes of the BVH root (given their plane/sphere hit, if any).  Those
                                   that hit nothing end their path at once; the others carry their hit and skip extend.
                                   Neither takes a slot in an extend launch. */
} pt_stats;

typedef struct pt_ctx pt_ctx;

/* Options for pt_set_option. */
#define PT_OPT_POOL_SIZE        1   /* path slots in flight; the reference fixes 0x200000 (src/render.cuh:14) */
#define PT_OPT_COUNT_TRAVERSAL  2   /* 1: extend counts nodes/triangles/leaves (slower; for roofline bookkeeping) */
#define PT_OPT_TIME_EXTEND      3   /* 1 (default): bracket every extend launch with HIP events */
#define PT_OPT_EXTEND_MODE      5   /* 2 (default): persistent waves, one traversal step per lane and iteration, finished ray
                                       slots refilled; 1: one 64-ray batch per wave step, loop-per-phase traversal */
#define PT_OPT_TUNE_REFILL_IDLE 6   /* extend tuning: refill a wave's ray slots once this many lanes are empty */
#define PT_OPT_TUNE_LDS_SLOTS   7   /* extend tuning: traversal stack slots per lane kept in LDS (default 26) */
#define PT_OPT_TUNE_EXTEND_BLOCKS 10 /* extend tuning: cap on resident extend blocks per CU (1..8, default 8) */
#define PT_OPT_STREAMS          9   /* wavefront loops run side by side on streams of their own, each rendering a share of
                                       the samples (1..4, default 3); the film does not depend on it */
#define PT_OPT_ENV_PACK         8   /* 1 (default): keep environment texels as shared-exponent word + pdf (8 B, tiled)
                                       when that reproduces every texel exactly; 0: always 16 B float texels.
                                       Read by pt_scene_upload. */
#define PT_OPT_TUNE_XCD_GROUP   12  /* extend tuning: 1 = the blocks that share an XCD (and its L2) work on one contiguous eighth
                                       of the ray buffer instead of chunks from all over it (default 0) */
#define PT_OPT_NODE_LAYOUT      11  /* 1 (default): inner nodes are re-ordered at upload so that a node and the child a ray most
                                       likely visits next share a 128 B cache line; 0: the order given.  Hits do not depend on
                                       it.  Read by pt_scene_upload. */
#define PT_OPT_RENDER_MODE      13  /* 1: wavefront iterations over path buffers (bounce / generate / extend kernels on
                                       PT_OPT_STREAMS streams); 2: every camera path is traced from first ray to end by one
                                       lane of a persistent kernel (traversal steps, logic and shading in the same wave; no
                                       path buffers); 0 (default): chosen for the scene and film in place by the first
                                       pt_render call after pt_scene_upload / pt_film_bind, from its first sample: mode 2 when
                                       fewer than 42 % of that sample's rays are settled where they are made (nothing below
                                       the BVH root can stop them), else mode 1.  The film does not depend on the mode. */
#define PT_OPT_TUNE_PATH_WAVES  14  /* render mode 2 tuning: cap on resident waves per CU (0 = as many as fit, default) */
#define PT_OPT_TUNE_PATH_REFILL 15  /* render mode 2 tuning: a wave shades / restarts its lanes once this many wait (default 52) */
#define PT_OPT_TUNE_PATH_LDS_SLOTS 16 /* render mode 2 tuning: traversal stack slots per lane kept in LDS (default 28); deeper
                                       entries go to a global array */
#define PT_OPT_SLAB_BYTES       4   /* device bytes the per-path contribution slab may take (default 32 GiB);
                                       samples beyond it are rendered slab after slab */

/* No reference counterpart (the reference uses process-global device symbols, src/render.cu:47,91). */
int  pt_ctx_create(int device_ordinal, pt_ctx **out);
void pt_ctx_destroy(pt_ctx *ctx);
const char *pt_last_error(const pt_ctx *ctx);          /* ctx may be NULL: error of a failed pt_ctx_create */
int  pt_set_option(pt_ctx *ctx, int option, uint64_t value);

/* Replaces the cudaMalloc/cudaMemcpy half of Scene::Scene, BVH::send_to_device and
 * EnvironmentMap::send_to_device (reference src/scene.cu:160-214, src/bvh.cu:296-313,
 * src/environment_map.cu:131-134,142-166).  A scene that is refused (unknown material type, malformed BVH buffers,
 * table entries out of range, no memory: negative return, pt_last_error says what) leaves the context with the scene it
 * had; a complete upload replaces and frees that one. */
int  pt_scene_upload(pt_ctx *ctx, const pt_scene_desc *scene);

/* Closest hit on a batch of rays given as host arrays of n*3 floats: the body of intersect_scene
 * (reference src/render.cu:11-44) exposed for parity checks.  normal_out (n*3, unit length) may be NULL. */
int  pt_intersect(pt_ctx *ctx, uint64_t n, const float *org, const float *dir,
                  float *t_out, uint32_t *prim_out, float *normal_out);

/* Same kernel on device-resident rays (float4 per origin/direction, xyz used; hit_out float4 = unit normal xyz +
 * prim id bits, t_out float).  Used to time traversal alone.  stats may be NULL. */
int  pt_intersect_device(pt_ctx *ctx, uint64_t n, const void *d_org4, const void *d_dir4,
                         void *d_hit4, void *d_t, pt_stats *stats);

/* pt_intersect_device for callers that cannot allocate device memory (tight serving loops, harnesses that forbid hipMalloc,
 * code under HIP stream capture).  No counterpart in the reference.  Same arguments, same results bit for bit, but the
 * call makes no hipMalloc / hipFree: its hit-id scratch and counters live in one static device array per GPU (16 MiB +
 * 256 B, held by every GPU the library runs on until the process ends) and its traversal stacks in LDS only.
 * Two behaviours, chosen by `stats`:
 *   stats != NULL  synchronous, like pt_intersect_device: counters reset, HIP events around the traversal, the stream
 *                  waited for, *stats filled.  Not usable while the stream (pt_stream) is being captured.
 *   stats == NULL  the traversal and the hit-expansion kernel are only QUEUED on the context's stream and the call
 *                  returns: no memset, no event, no copy, no synchronisation -- nothing a stream capture refuses.  The
 *                  results are complete once the stream has passed that point.
 * Limits: n <= 4,194,304 rays per call and a tree of at most 47 levels (48 stack slots); beyond either the call returns
 * PT_ERR_ARG.  All calls on one GPU share that GPU's array: calls that overlap in time on one GPU race, whatever their
 * contexts or streams. */
int  pt_intersect_device_noalloc(pt_ctx *ctx, uint64_t n, const void *d_org4, const void *d_dir4,
                                 void *d_hit4, void *d_t, pt_stats *stats);

/* Camera rays of one sample index exactly as the generate stage produces them (reference src/render.cu:93-127),
 * written to host arrays of width*height*3 floats indexed by film pixel.  For parity checks. */
int  pt_primary_rays(pt_ctx *ctx, uint32_t width, uint32_t height, uint64_t sample, uint64_t seed,
                     float *org_out, float *dir_out);

/* The shading path's transcendental functions as the DEVICE evaluates them, on host arrays of n floats, for parity checks:
 * which = 0 sin and cos of a[i] in [0, 2 pi] -> out0, out1; 1 atan2(a[i], b[i]) -> out0; 2 acos(a[i]), a in [-1, 1] -> out0;
 * 3 a[i]^5 -> out0.  They stand where the reference calls the CUDA math library (src/material.cuh:13-21,34-39,
 * src/environment_map.cuh:32-49) and are fixed polynomial kernels whose bits do not depend on where they run. */
int  pt_math_eval(pt_ctx *ctx, int which, uint64_t n, const float *a, const float *b, float *out0, float *out1);

/* The film: float3[width*height], row 0 = top (reference Scene::image_data, src/scene.cu:216, src/render.cu:121).
 * device_film = NULL lets the library allocate it; otherwise the caller's device buffer is accumulated into
 * (e.g. a tensor that is reduced across GPUs afterwards). */
int  pt_film_bind(pt_ctx *ctx, uint32_t width, uint32_t height, void *device_film);
int  pt_film_clear(pt_ctx *ctx);
int  pt_film_download(pt_ctx *ctx, float *rgb_out);    /* replaces the cudaMemcpy at reference src/main.cu:40 */

/* Image::tone_map and the 16-bit quantisation of Image::save_png (reference src/image.cpp:12-33) on the device:
 * rgb_out[width*height*3] = (uint16)(65535 * (1 - exp(-c * exposure)) ^ (1 / gamma)), host buffer.  Half the bytes of
 * pt_film_download; the device's expf/powf may differ from the host's in the last place, so a sample can differ from
 * pth_tonemap16's by one step of 65535. */
int  pt_film_tonemap16(pt_ctx *ctx, float exposure, float gamma, uint16_t *rgb_out);

/* Film reduce for a process that drives several GPUs: adds the films of the other contexts to the film of ctxs[root] and
 * returns when the sum is complete.  Contexts on distinct GPUs: ONE RCCL reduce (ncclReduce, sum, float32, width*height*3
 * elements) over xGMI, every rank on its context's stream; librccl is bound at the first such call, not linked, and the
 * communicators are kept until the last context of the process is destroyed.  Contexts that share a GPU (RCCL takes one
 * rank per device): copies into a staging buffer on the root's device and adds, in index order.  All films must have the
 * root's size; the other films are left as they are.  The sum equals the sum of the films up to float addition order.
 * No reference counterpart (the reference renders on one GPU).  With one process per GPU the same reduce is one
 * RCCL call on the bound film buffers instead (see pt_film_bind).  PT_FILM_REDUCE_RCCL=1 in the environment sends a
 * single context through RCCL too (a one-rank reduce that changes nothing), to check the binding on a one-GPU machine. */
int  pt_film_reduce(pt_ctx *const *ctxs, int n, int root);

/* Replaces render_scene (reference src/render.cuh:50, src/render.cu:244-280): accumulates samples
 * [spp_begin, spp_begin+spp_count) of every pixel, each weighted 1/spp_total, into the bound film and returns
 * when the device is done.  Splitting [0, spp_total) over several calls, contexts or GPUs and summing the films
 * gives the same image as one call (up to float addition order).  Needs width*height*16 bytes of device memory per
 * sample per pixel rendered at a time (see PT_OPT_SLAB_BYTES). */
int  pt_render(pt_ctx *ctx, uint64_t spp_begin, uint64_t spp_count, uint64_t spp_total, uint64_t seed,
               pt_stats *stats);

/* Render mode (1 or 2, see PT_OPT_RENDER_MODE) the last pt_render call of the context ran in; 0 before the first.  No reference
 * counterpart. */
int  pt_render_mode_used(const pt_ctx *ctx);

/* 1 when the uploaded scene keeps its environment texels in the packed form (see PT_OPT_ENV_PACK), 0 when not,
 * negative without a scene.  No reference counterpart. */
int  pt_env_is_packed(const pt_ctx *ctx);

/* Stream the ctx launches on (a hipStream_t), for callers that want to order their own work after it. */
void *pt_stream(pt_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif
