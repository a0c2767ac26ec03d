// extend: closest hit of every ray of a buffer (reference src/render.cu:11-44, src/bvh.cuh:130-189).
//
// The BVH is one array of 64 B records: inner nodes first (two child boxes + two child references), triangle records
// after them (three Woop rows + id/flags).  A child reference is the index of a record; references >= first_tri are
// leaves (runs of triangle records ended by a "last" flag).  Every traversal step, inner node or triangle, therefore is
// "fetch 64 B at recs + 64 * ref, then compute", and a lane's traversal stack is a column of an LDS array whose bottom
// slot holds the sentinel that ends the traversal.
//
// Two kernels produce bit-identical answers:
//   k_extend_batch   one 64-ray batch per wave step, loop-per-phase traversal as in the reference;
//   k_extend_stream  persistent waves: each lane owns a ray slot, every loop iteration every lane with a ray takes ONE
//                    step (inner node or triangle) through the same four load instructions, and slots whose ray has
//                    finished are refilled from the wave's share of the ray buffer.  With 64 incoherent rays per wave
//                    the loop-per-phase form leaves most lanes idle (measured 19 % active lanes on MI355X); the
//                    single-step form keeps every ray in flight moving and measured 1.5x faster end to end.
#pragma once

#include "pt_device.h"

namespace ptd {

constexpr int kBlock = 256;
constexpr uint32_t kNoRay = 0xFFFFFFFFu;
constexpr uint32_t kChunk = 128;           // rays per chunk; chunk c belongs to wave c mod #waves
constexpr int kRefillIdle = 24;            // stream kernel: refill once this many lanes have no ray

struct ExtendArgs {
    const float4 *ro;          // ray origins (xyz)
    const float4 *rd;          // ray directions (xyz)
    const uint32_t *n_ptr;     // ray count on the device, or NULL
    uint32_t n_fixed;          // ray count when n_ptr is NULL
    float4 *hit_out;           // unit normal xyz, w = prim id bits
    float *t_out;              // hit distance, FLT_MAX on a miss
    unsigned long long *cnt3;  // nodes, triangles, leaves (COUNT only)
    int refill_idle;
};

// Per-lane traversal state.  The hit is kept as (t, hid): hid = PT_PRIM_PLANE|i, PT_PRIM_SPHERE|i, the record index of
// the triangle, or PT_PRIM_MISS; normal and reported primitive id are derived from it once, when the ray retires.
struct RayState {
    f3 o, d, inv, nood;
    float t;
    uint32_t hid;
    int node, sp;
    uint32_t c_nodes, c_tris, c_leaves;
};

__device__ __forceinline__ void ray_begin(const DScene &sc, RayState &R, f3 o, f3 d, int *stk_col)
{
    R.o = o;
    R.d = d;
    Hit h;
    h.t = kFltMax; h.prim = PT_PRIM_MISS; h.n = mk3(0, 0, 0);
    for (uint32_t k = 0; k < sc.num_planes; k++) hit_plane(sc.planes[k], k, o, d, h);
    for (uint32_t k = 0; k < sc.num_spheres; k++) hit_sphere(sc.spheres[k], k, o, d, h);
    R.t = h.t;
    R.hid = h.prim;
    R.inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    const f3 ood = {R.inv.x * o.x, R.inv.y * o.y, R.inv.z * o.z};
    R.nood = mk3(ood.x != ood.x ? fl(0x7FC00000) : -ood.x, ood.y != ood.y ? fl(0x7FC00000) : -ood.y,
                 ood.z != ood.z ? fl(0x7FC00000) : -ood.z);
    stk_col[0] = kSentinel;                 // bottom slot: popping it ends the traversal
    R.sp = 1;
    R.node = sc.has_bvh ? 0 : kSentinel;
}

// Writes the hit record of a finished ray: unit normal (reference src/render.cu:41) + reported primitive id, and t.
__device__ __forceinline__ void ray_retire(const DScene &sc, const RayState &R, uint32_t ray, float4 *hit_out, float *t_out)
{
    f3 n = mk3(0, 0, 0);
    uint32_t prim = R.hid;
    if (R.hid != PT_PRIM_MISS) {
        if (R.hid & PT_PRIM_PLANE) {
            n = xyz(sc.planes[R.hid & 0x3FFFFFFFu].nrm);
        } else if (R.hid & PT_PRIM_SPHERE) {
            const float4 pr = sc.spheres[R.hid & 0x3FFFFFFFu].pos_r;     // reference src/object.cuh:39
            const f3 p = {__fmaf_rn(R.d.x, R.t, R.o.x), __fmaf_rn(R.d.y, R.t, R.o.y), __fmaf_rn(R.d.z, R.t, R.o.z)};
            n = (p - xyz(pr)) / pr.w;
        } else {
            const float4 *rec = sc.recs + (size_t)R.hid * 4;               // reference src/bvh.cuh:114
            n = cross3(xyz(rec[1]), xyz(rec[2]));
            prim = __float_as_uint(rec[3].x);
        }
        n = normalize3(n);
    }
    hit_out[ray] = make_float4(n.x, n.y, n.z, __uint_as_float(prim));
    t_out[ray] = R.t;
}

// One inner-node step (reference src/bvh.cuh:141-168): test both child boxes, descend into the nearer hit child
// (ties: left), push the farther one, or pop when neither is hit.
template <bool COUNT, bool CANON>
__device__ __forceinline__ void node_step(const DScene &sc, RayState &R, int *stk_col, float4 r0, float4 r1, float4 r2,
                                          float4 r3)
{
    if (COUNT) R.c_nodes++;
    const int cl = fi(r3.x), cr = fi(r3.y);
    float t0l, t0r;
    const bool hl = hit_aabb<CANON>(R.inv, R.nood, r0.x, r0.y, r0.z, r0.w, r2.x, r2.y, R.t, t0l);
    const bool hr = hit_aabb<CANON>(R.inv, R.nood, r1.x, r1.y, r1.z, r1.w, r2.z, r2.w, R.t, t0r);
    const bool both = hl && hr;
    const bool right_first = both && (t0r < t0l);
    int next = hl ? cl : cr;
    next = right_first ? cr : next;
    const int far_child = right_first ? cl : cr;
    if (both) {
        stk_col[R.sp * kBlock] = far_child;
        R.sp++;
    }
    if (!(hl || hr)) {
        R.sp--;
        next = stk_col[R.sp * kBlock];
    }
    if (COUNT && next != kSentinel && (uint32_t)next >= sc.first_tri) R.c_leaves++;
    R.node = next;
}

// One triangle step (reference src/bvh.cuh:171-187, 93-120): test the record, go on to the next record of the leaf or
// pop after the last one.  All of the test is evaluated unconditionally; the outcome is the same as the reference's
// nested early-outs because nothing in it has side effects.
template <bool COUNT>
__device__ __forceinline__ void tri_step(const DScene &sc, RayState &R, int *stk_col, float4 v00, float4 v11, float4 v22,
                                         float4 meta)
{
    const uint32_t flags = __float_as_uint(meta.y);
    if (COUNT && !(flags & 2u)) R.c_tris++;
    const f3 o = R.o, d = R.d;
    const float Oz = __fmaf_rn(-o.z, v00.z, __fmaf_rn(-o.y, v00.y, __fmaf_rn(-o.x, v00.x, v00.w)));
    const float Dz = __fmaf_rn(d.z, v00.z, __fmaf_rn(d.y, v00.y, d.x * v00.x));
    const float invDz = 1.0f / Dz;
    const float t = Oz * invDz;
    const float Ox = __fmaf_rn(o.z, v11.z, __fmaf_rn(o.y, v11.y, __fmaf_rn(o.x, v11.x, v11.w)));
    const float Dx = __fmaf_rn(d.z, v11.z, __fmaf_rn(d.y, v11.y, d.x * v11.x));
    const float u = __fmaf_rn(t, Dx, Ox);
    const float Oy = __fmaf_rn(o.z, v22.z, __fmaf_rn(o.y, v22.y, __fmaf_rn(o.x, v22.x, v22.w)));
    const float Dy = __fmaf_rn(d.z, v22.z, __fmaf_rn(d.y, v22.y, d.x * v22.x));
    const float v = __fmaf_rn(t, Dy, Oy);
    const bool hit = (t > kEps) & (t < R.t) & (u >= 0.0f) & (v >= 0.0f) & (u + v <= 1.0f);
    R.t = hit ? t : R.t;
    R.hid = hit ? (uint32_t)R.node : R.hid;
    int next = R.node + 1;
    if (flags & 1u) {
        R.sp--;
        next = stk_col[R.sp * kBlock];
        if (COUNT && next != kSentinel && (uint32_t)next >= sc.first_tri) R.c_leaves++;
    }
    R.node = next;
}

template <bool COUNT>
__device__ __forceinline__ void flush_counts(const RayState &R, unsigned long long *cnt3)
{
    if (COUNT) {
        atomicAdd(&cnt3[0], (unsigned long long)R.c_nodes);
        atomicAdd(&cnt3[1], (unsigned long long)R.c_tris);
        atomicAdd(&cnt3[2], (unsigned long long)R.c_leaves);
    }
}

// ---- loop-per-phase traversal of one ray (reference src/bvh.cuh:139-188) -------------------------------------------
template <bool COUNT, bool CANON>
__device__ __forceinline__ void traverse_phases(const DScene &sc, RayState &R, int *stk_col)
{
    while (R.node != kSentinel) {
        while ((uint32_t)R.node < sc.first_tri) {
            const float4 *p = sc.recs + (size_t)R.node * 4;
            node_step<COUNT, CANON>(sc, R, stk_col, p[0], p[1], p[2], p[3]);
        }
        while (R.node != kSentinel && (uint32_t)R.node >= sc.first_tri) {
            const float4 *p = sc.recs + (size_t)R.node * 4;
            tri_step<COUNT>(sc, R, stk_col, p[0], p[1], p[2], p[3]);
        }
    }
}

template <int STACK, bool COUNT>
__global__ __launch_bounds__(kBlock) void k_extend_batch(DScene sc, ExtendArgs a)
{
    __shared__ int stk[STACK][kBlock];
    int *stk_col = &stk[0][threadIdx.x];
    const uint32_t n = a.n_ptr ? *a.n_ptr : a.n_fixed;
    const uint32_t stride = gridDim.x * kBlock;
    RayState R;
    R.c_nodes = R.c_tris = R.c_leaves = 0;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        ray_begin(sc, R, xyz(a.ro[i]), xyz(a.rd[i]), stk_col);
        // wave-uniform choice: the NaN-rewriting slab test only when some ray of this wave needs it
        if (__any(slab_degenerate(R.d))) traverse_phases<COUNT, true>(sc, R, stk_col);
        else traverse_phases<COUNT, false>(sc, R, stk_col);
        ray_retire(sc, R, i, a.hit_out, a.t_out);
    }
    flush_counts<COUNT>(R, a.cnt3);
}

// ---- persistent single-step traversal ---------------------------------------------------------------------------------
template <bool COUNT, bool CANON>
__device__ __forceinline__ void stream_step(const DScene &sc, RayState &R, int *stk_col)
{
    if (R.node == kSentinel) return;
    const float4 *p = sc.recs + (size_t)R.node * 4;
    const float4 r0 = p[0], r1 = p[1], r2 = p[2], r3 = p[3];
    if ((uint32_t)R.node < sc.first_tri) node_step<COUNT, CANON>(sc, R, stk_col, r0, r1, r2, r3);
    else tri_step<COUNT>(sc, R, stk_col, r0, r1, r2, r3);
}

template <int STACK, bool COUNT>
__global__ __launch_bounds__(kBlock) void k_extend_stream(DScene sc, ExtendArgs a)
{
    __shared__ int stk[STACK][kBlock];
    int *stk_col = &stk[0][threadIdx.x];
    const uint32_t n = a.n_ptr ? *a.n_ptr : a.n_fixed;
    const uint32_t nwaves = (gridDim.x * kBlock) >> 6;
    // wave-uniform cursor over this wave's chunks of the ray buffer
    uint32_t chunk = __builtin_amdgcn_readfirstlane((blockIdx.x * kBlock + threadIdx.x) >> 6);
    uint32_t next = chunk * kChunk;
    uint32_t end = next < n ? min(n, next + kChunk) : next;

    RayState R;
    R.c_nodes = R.c_tris = R.c_leaves = 0;
    R.node = kSentinel;
    R.sp = 1;
    R.t = kFltMax;
    R.hid = PT_PRIM_MISS;
    R.o = R.d = R.inv = R.nood = mk3(0, 0, 0);
    uint32_t ray = kNoRay;
    bool degenerate = false;
    bool canon = false;          // wave-uniform: some ray in flight needs the NaN-rewriting slab test
    bool more_work = true;

    for (;;) {
        const int nidle = __popcll(__ballot(R.node == kSentinel));
        if (nidle == 64 || (more_work && nidle >= a.refill_idle)) {
            // retire finished rays, refill their slots
            if (ray != kNoRay && R.node == kSentinel) {
                ray_retire(sc, R, ray, a.hit_out, a.t_out);
                ray = kNoRay;
                degenerate = false;
            }
            const bool empty = ray == kNoRay;
            const unsigned long long emask = __ballot(empty);
            if (next >= end) {
                const uint64_t c2 = (uint64_t)chunk + nwaves;
                if (c2 * kChunk < n) {
                    chunk = (uint32_t)c2;
                    next = chunk * kChunk;
                    end = min(n, next + kChunk);
                }
            }
            const uint32_t take = min(end - next, (uint32_t)__popcll(emask));
            const uint32_t rank =
                __builtin_amdgcn_mbcnt_hi((uint32_t)(emask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)emask, 0));
            if (empty && rank < take) {
                ray = next + rank;
                ray_begin(sc, R, xyz(a.ro[ray]), xyz(a.rd[ray]), stk_col);
                degenerate = slab_degenerate(R.d);
            }
            next += take;
            more_work = next < end || ((uint64_t)chunk + nwaves) * kChunk < n;
            canon = __any(degenerate);
            if (!__any(ray != kNoRay)) break;            // nothing in flight and nothing left to fetch
        }
        if (canon) stream_step<COUNT, true>(sc, R, stk_col);
        else stream_step<COUNT, false>(sc, R, stk_col);
    }
    flush_counts<COUNT>(R, a.cnt3);
}

}   // namespace ptd
