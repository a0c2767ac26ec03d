// Whole camera paths in one persistent kernel (reference src/render.cu:244-280: the loop over generate, the four material
// kernels, intersect_scene and logic_kernel, with every stage's work done by the wave that owns the path).
//
// Each lane of a persistent wave owns ONE camera path from its first ray to its end.  A lane is in one of three states:
//   A  no path          (waits for a new camera path)
//   B  hit known        (its ray's closest hit (t, hid) is known: the path wants its logic + shade step)
//   C  traversing       (its ray is on its way through the tree, one 64 B record per step)
// The wave's loop takes one traversal step for every lane in C, exactly like k_extend_stream, until enough lanes have left
// C; then ONE service phase runs, in this order and each part only for the lanes that need it:
//   step    lanes in B: logic (miss / depth limit / Russian roulette, src/render.cu:49-89) and shade (src/render.cu:129-242);
//           the path ends (its slab entry is written, the lane goes to A) or continues with a new ray
//   camera  lanes in A: the next camera paths of the wave's chunk (src/render.cu:93-127)
//   setup   lanes with a new ray from either part: plane and sphere tests, slab-test constants, and the slab test of the
//           root's two child boxes.  A ray nothing below the root can stop is settled at once (state B, or A when a
//           continued ray leaves the scene); the others enter the tree (state C).
// Nothing of a path ever leaves the registers of its lane except its one slab entry when it ends: no path buffers, no
// compaction, no per-iteration launches or counters, and a ray's hit normal is taken from the triangle record while the
// traversal has it in registers.  Which wave or lane traces a path changes nothing in it (random numbers are keyed on
// pixel, sample and bounce), so the film is bit-identical to the wavefront loop's.
//
// Work is handed out in chunks of kPathChunk consecutive path indices of one sample (sample-major order, 8x4 pixel
// tiles), claimed from one device-wide cursor; a wave claims its next chunk while it still works on the current one, so
// the atomic's round trip is never waited for, and the frame's tail is as long as one path, not one chunk.
#pragma once

#include "pt_wavefront.h"

namespace ptd {
namespace {

constexpr int kPathBlock = 64;            // one wave per block: the waves share nothing, so nothing needs a barrier
constexpr uint32_t kPathChunk = 256;      // camera paths per claim
constexpr uint32_t kNoPath = 0xFFFFFFFFu;

struct PathCounters {
    unsigned int cursor;                  // next chunk to hand out
    unsigned int pad;
    unsigned long long rays;              // closest-hit queries: every ray created, settled at creation or not
    unsigned long long settled;           // of those, rays settled where they were created (no tree walk)
    unsigned long long paths;             // camera paths started
    unsigned long long trav[3];           // nodes, triangles, leaves (COUNT only)
};

struct PathArgs {
    float4 *slab;                 // [sample - slab_first_sample][pixel]: what the path adds to its pixel, times spp_total
    uint32_t slab_first_sample;   // global sample index of the slab's first row
    uint32_t W, H, npix;
    FastDiv row_div;              // pixel_of's divisor
    uint32_t chunks_per_sample;   // ceil(npix / kPathChunk)
    FastDiv cps_div;
    uint32_t total_chunks;        // chunks_per_sample * samples of this launch
    uint32_t seed0, seed1;
    PathCounters *cnt;
    int refill_idle;              // service once this many lanes wait for it
    int lds_slots;
    int *gstk;                    // OVERFLOW only
};

// logic + shade of a path whose ray's hit is known (the first half of a wavefront iteration for one path).  Returns true
// when the path continues: (o, d) is then its new ray, th its throughput, sk its sample | bounce << 27.  In either case esc
// holds what the path adds to its pixel if it ends with the ray it has now.
__device__ __forceinline__ bool path_logic_shade(const DScene &sc, f3 &o, f3 &d, float t, uint32_t hid, f3 hn, uint32_t pixel,
                                                 uint32_t &sk, f3 &th, f3 &esc, uint32_t seed0, uint32_t seed1)
{
    const uint32_t sample = sk & 0x07FFFFFFu, k = sk >> 27;
    bool alive = false;
    const bool late_miss = t == kFltMax && k == 0;             // a camera ray that hit nothing: no escape value yet
    if (!(t == kFltMax || 2 * k >= kMaxDepth)) {
        alive = true;
        if (2 * k + 1 >= kMinDepth) {                           // Russian roulette (src/render.cu:78-86)
            Rng rng(pixel, sample, k, STAGE_LOGIC, seed0, seed1);
            const float p = fmaxf(th.x, fmaxf(th.y, th.z));
            if (rng.next() > p) {
                alive = false;
                esc = mk3(0.0f, 0.0f, 0.0f);
            } else {
                const float ip = 1 / p;
                th.x *= ip; th.y *= ip; th.z *= ip;
            }
        }
    }
    f3 out = d, hitp = o, n3 = o;
    float4 mat = make_float4(0, 0, 0, 0);
    float4 th4 = make_float4(th.x, th.y, th.z, 0.0f);
    bool lambert = false;
    if (alive) {
        uint32_t mat_type;
        material_of(sc, hid, mat, mat_type);
        const int slot = mat_slot(mat_type);
        // unit normal (src/render.cu:41): of a triangle from the cross product the traversal kept, else from the primitive
        if (hid & (PT_PRIM_PLANE | PT_PRIM_SPHERE)) {
            uint32_t prim;
            n3 = hit_normal(sc, hid, o, d, t, prim);
        } else {
            n3 = normalize3(hn);
        }
        hitp = mk3(__fmaf_rn(d.x, t, o.x), __fmaf_rn(d.y, t, o.y), __fmaf_rn(d.z, t, o.z));
        Rng rng(pixel, sample, k, STAGE_SHADE, seed0, seed1);
        out = shade_dir(sc, slot, d, n3, mat, rng, th4, lambert);
    }
    if (alive || late_miss) {
        // one environment lookup serves both: the new ray's direction, or the direction of the camera ray that missed
        const float4 e = env_lookup(sc, out);
        if (lambert) lambert_weight(n3, out, mat, e, th4);
        const float4 ev = escape_value(th4, e);
        esc = mk3(ev.x, ev.y, ev.z);
    }
    if (!alive) return false;
    th = mk3(th4.x, th4.y, th4.z);
    sk = sample | ((k + 1) << 27);
    o = hitp;
    d = out;
    return true;
}

extern __shared__ int g_path_stack_lds[];      // [lds_slots][kPathBlock], sized at launch

// 90 VGPRs, five waves per SIMD.  Measured on MI355X (868 k-triangle frame): 4 / 8 / 12 / 16 / 20 waves per CU = 850 /
// 457 / 328 / 268 / 235 ms, so the kernel lives on resident waves; a sixth wave per SIMD (80 VGPRs) costs 7 to 11 spilled
// registers and gains nothing (238.6 against 236.4 ms), and parking the registers only traversing lanes need in LDS
// during a service phase means a 16-slot LDS stack with global overflow, which alone costs 9 %.
template <bool OVERFLOW, bool COUNT>
__global__ __launch_bounds__(kPathBlock) __attribute__((amdgpu_waves_per_eu(5, 5))) void k_path_stream(DScene sc, PathArgs a)
{
    const Stack<OVERFLOW, kPathBlock> stk = make_stack<OVERFLOW, kPathBlock>(g_path_stack_lds, a.lds_slots, a.gstk);
    const uint32_t lane = threadIdx.x;

    RayState R;
    R.c_nodes = R.c_tris = R.c_leaves = 0;
    R.node = kSentinel;
    R.sp = 1;
    R.t = kFltMax;
    R.hid = PT_PRIM_MISS;
    R.o = R.d = R.inv = R.nood = R.hn = mk3(0, 0, 0);
    f3 th = mk3(0, 0, 0), esc = mk3(0, 0, 0);
    uint32_t pixel = kNoPath;            // kNoPath: state A
    uint32_t sk = 0;
    uint32_t n_rays = 0, n_settled = 0;  // wave-uniform counts (ballots), added to the device counters once at the end
    bool canon = false;                  // wave-uniform: some ray in flight needs the NaN-rewriting slab test

    // the wave's chunk: path indices [next, end) of local sample chunk_s; the chunk after it is claimed ahead (pf, valid in
    // lane 0)
    uint32_t next = 0, end = 0, chunk_s = 0;
    uint32_t n_paths = 0;
    uint32_t pf = 0;
    if (lane == 0) pf = atomicAdd(&a.cnt->cursor, 1u);
    bool more_paths = true;

    int stuck = 0;           // wave-uniform: lanes without a path that no service can give one (no camera paths are left)
    for (;;) {
        const bool traversing = R.node != kSentinel;
        const unsigned long long cmask = __ballot(traversing);
        // lanes a service would do something for: every lane that is not traversing, less the ones that stay empty
        const int waiting = 64 - __popcll(cmask) - stuck;
        if (cmask == 0ull || waiting >= a.refill_idle) {
            if (waiting == 0) break;                  // nothing in flight, nothing to shade, nothing left to start
            bool newray = false;
            // ---- step: lanes whose ray's hit is known ----------------------------------------------------------------
            if (pixel != kNoPath && !traversing) {
                newray = path_logic_shade(sc, R.o, R.d, R.t, R.hid, R.hn, pixel, sk, th, esc, a.seed0, a.seed1);
                if (!newray) {
                    const uint32_t sample = sk & 0x07FFFFFFu;
                    st_stream(a.slab + (size_t)(sample - a.slab_first_sample) * a.npix + pixel,
                              make_float4(esc.x, esc.y, esc.z, 0.0f));
                    pixel = kNoPath;
                }
            }
            // ---- camera: new paths for the lanes without one ---------------------------------------------------------
            if (more_paths) {
                if (next >= end) {
                    const uint32_t c = __builtin_amdgcn_readfirstlane(pf);
                    if (c < a.total_chunks) {
                        chunk_s = fastdiv(c, a.cps_div);
                        next = (c - chunk_s * a.chunks_per_sample) * kPathChunk;
                        end = min(a.npix, next + kPathChunk);
                        if (lane == 0) pf = atomicAdd(&a.cnt->cursor, 1u);
                    } else {
                        more_paths = false;
                    }
                }
                const bool empty = pixel == kNoPath;
                const unsigned long long emask = __ballot(empty);
                const uint32_t take = min(end - next, (uint32_t)__popcll(emask));
                const uint32_t rank =
                    __builtin_amdgcn_mbcnt_hi((uint32_t)(emask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)emask, 0));
                if (empty && rank < take) {
                    const uint32_t sample = a.slab_first_sample + chunk_s;
                    uint32_t x, y;
                    pixel_of(next + rank, a.W, a.H, a.row_div, x, y);
                    pixel = (a.H - y - 1) * a.W + x;
                    Rng rng(pixel, sample, 0, STAGE_CAMERA, a.seed0, a.seed1);
                    camera_ray(sc.cam, a.W, a.H, x, y, rng, R.o, R.d);
                    th = mk3(1.0f, 1.0f, 1.0f);
                    sk = sample;
                    newray = true;
                }
                next += take;
                n_paths += take;
            }
            // ---- setup: the traversal's first step on every new ray (src/render.cu:25-33, src/bvh.cuh:133-137) -----------
            n_rays += (uint32_t)__popcll(__ballot(newray));
            bool settled = false;
            f3 ninv = mk3(0, 0, 0), nnood = ninv;
            if (newray) {
                ray_setup(sc, R.o, R.d, R.t, R.hid, ninv, nnood);
                settled = true;
                if (sc.has_bvh) {
                    const float4 r0 = sc.recs[0], r1 = sc.recs[1], r2 = sc.recs[2];
                    float t0;
                    const bool hl = hit_aabb<true>(ninv, nnood, r0.x, r0.y, r0.z, r0.w, r2.x, r2.y, R.t, t0);
                    const bool hr = hit_aabb<true>(ninv, nnood, r1.x, r1.y, r1.z, r1.w, r2.z, r2.w, R.t, t0);
                    settled = !(hl || hr);
                }
                if (settled) {
                    if (COUNT && sc.has_bvh) R.c_nodes++;        // the root visit that settled it
                    // a continued ray that leaves the scene ends its path here: the escape value is what the path adds
                    if (R.hid == PT_PRIM_MISS && (sk >> 27) != 0) {
                        const uint32_t sample = sk & 0x07FFFFFFu;
                        st_stream(a.slab + (size_t)(sample - a.slab_first_sample) * a.npix + pixel,
                                  make_float4(esc.x, esc.y, esc.z, 0.0f));
                        pixel = kNoPath;
                    }
                } else {
                    stk.put(0, kSentinel);
                    R.node = 0;
                }
            }
            if (newray) { R.inv = ninv; R.nood = nnood; R.sp = 1; }
            n_settled += (uint32_t)__popcll(__ballot(settled));
            canon = __any(R.node != kSentinel && slab_degenerate(R.d));
            stuck = more_paths ? 0 : __popcll(__ballot(pixel == kNoPath));
        }
        if (canon) {
#pragma unroll
            for (int s = 0; s < kStepsPerCheck; s++) stream_step<COUNT, true, true>(sc, R, stk);
        } else {
#pragma unroll
            for (int s = 0; s < kStepsPerCheck; s++) stream_step<COUNT, false, true>(sc, R, stk);
        }
    }
    // the wave's counts: one atomic each
    if (lane == 0) {
        atomicAdd(&a.cnt->rays, (unsigned long long)n_rays);
        atomicAdd(&a.cnt->settled, (unsigned long long)n_settled);
        atomicAdd(&a.cnt->paths, (unsigned long long)n_paths);
    }
    flush_counts<COUNT>(R, a.cnt->trav);
}

}   // namespace
}   // namespace ptd
