// extend: closest hit of every ray of a buffer (reference src/render.cu:11-44, src/bvh.cuh:130-189).
//
// The BVH is one array of 64 B records: inner nodes (two child boxes + two child references) and triangle records (three
// Woop rows + id/flags), in an order chosen at upload.  A child reference is the index of a record, with the sign bit set
// when it points at a leaf (a run of consecutive triangle records ended by a "last" flag).  Every traversal step, inner
// node or triangle, therefore is "fetch 64 B at recs + 64 * index, then compute", and a lane's traversal stack is a
// column of an LDS array whose bottom slot holds the sentinel that ends the traversal.
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
constexpr uint32_t kRefIndex = 0x3FFFFFFFu;      // record index of a reference (references to leaves are negative)
constexpr int kStepsPerCheck = 2;                // traversal steps between two looks at the refill condition
constexpr uint32_t kChunk = 128;                 // rays per chunk; chunk c belongs to wave c mod #waves
// stream kernel: refill once this many lanes have no ray.  Measured on the 868 k-triangle stand-in, 78 M rays per launch
// (ms per frame): 8 -> 356, 16 -> 342, 24 -> 333, 32 -> 327, 44 -> 320, 56 -> 340.  Idle lanes cost no memory traffic,
// which is what bounds the kernel; every refill is issue time the wave's live rays stand still for.
constexpr int kRefillIdle = 44;

struct ExtendArgs {
    const float4 *ro;          // ray origins (xyz)
    const float4 *rd;          // ray directions (xyz)
    const uint32_t *n_ptr;     // ray count on the device, or NULL
    uint32_t n_fixed;          // ray count when n_ptr is NULL
    uint32_t *hid_out;         // hit id: PT_PRIM_PLANE|i, PT_PRIM_SPHERE|i, triangle record index, or PT_PRIM_MISS
    float *t_out;              // hit distance, FLT_MAX on a miss
    unsigned long long *cnt3;  // nodes, triangles, leaves (COUNT only)
    int refill_idle;
    int lds_slots;             // traversal stack slots per lane kept in LDS
    int *gstk;                 // stack slots beyond lds_slots: [slot - lds_slots][global thread]  (OVERFLOW only)
    int xcd_group;             // stream kernel: 1 = the blocks of an XCD share one contiguous eighth of the ray buffer
};

// Per-lane traversal state.  The hit is kept as (t, hid): hid = PT_PRIM_PLANE|i, PT_PRIM_SPHERE|i, the record index of
// the triangle, or PT_PRIM_MISS; normal and reported primitive id are derived from it once, when the ray retires.
struct RayState {
    f3 o, d, inv, nood;
    float t;
    uint32_t hid;
    int node, sp;
    f3 hn;                     // un-normalised normal of the triangle hit (kept only by kernels that shade, KEEPN)
    uint32_t c_nodes, c_tris, c_leaves;
};

// A lane's traversal stack: slot s lives at lds[s * BLOCK] (a column of the block's LDS array) while s < lds_slots and,
// in kernels built with OVERFLOW, in a global array beyond that.  Trees shallow enough for LDS alone use the kernels
// without the overflow test.  BLOCK = threads per block of the kernel that owns the array.
template <bool OVERFLOW, int BLOCK = kBlock>
struct Stack {
    int *lds;
    int *glob;
    int lds_slots;
    uint32_t gstride;
    __device__ __forceinline__ void put(int slot, int v) const
    {
        if (!OVERFLOW || slot < lds_slots) lds[slot * BLOCK] = v;
        else __builtin_nontemporal_store(v, glob + (size_t)(slot - lds_slots) * gstride);
    }
    __device__ __forceinline__ int get(int slot) const
    {
        if (!OVERFLOW || slot < lds_slots) return lds[slot * BLOCK];
        return __builtin_nontemporal_load(glob + (size_t)(slot - lds_slots) * gstride);
    }
};

template <bool OVERFLOW, int BLOCK = kBlock>
__device__ __forceinline__ Stack<OVERFLOW, BLOCK> make_stack(int *lds_base, int lds_slots, int *gstk)
{
    Stack<OVERFLOW, BLOCK> s;
    s.lds = lds_base + threadIdx.x;
    s.gstride = gridDim.x * BLOCK;
    s.glob = OVERFLOW ? gstk + (blockIdx.x * BLOCK + threadIdx.x) : nullptr;
    s.lds_slots = lds_slots;
    return s;
}

// The traversal stack of a ray: all entries in the lane's column, R.sp = entries on the stack; the bottom entry is the
// sentinel whose pop ends the traversal.
template <class STK>
__device__ __forceinline__ void stack_reset(RayState &R, const STK &stk)
{
    stk.put(0, kSentinel);
    R.sp = 1;
}

template <class STK>
__device__ __forceinline__ void stack_push(RayState &R, const STK &stk, int v)
{
    stk.put(R.sp, v);
    R.sp++;
}

template <class STK>
__device__ __forceinline__ int stack_pop(RayState &R, const STK &stk)
{
    R.sp--;
    return stk.get(R.sp);
}

// What every ray needs before the tree: its closest hit among the planes and spheres (reference src/render.cu:25-33)
// and the slab-test constants 1/d and -o/d (reference src/bvh.cuh:133-137), NaNs made positive.
__device__ __forceinline__ void ray_setup(const DScene &sc, f3 o, f3 d, float &t, uint32_t &hid, f3 &inv, f3 &nood)
{
    Hit h;
    h.t = kFltMax; h.prim = PT_PRIM_MISS; h.n = mk3(0, 0, 0);
    for (uint32_t k = 0; k < sc.num_planes; k++) hit_plane(sc.planes[k], k, o, d, h);
    for (uint32_t k = 0; k < sc.num_spheres; k++) hit_sphere(sc.spheres[k], k, o, d, h);
    t = h.t;
    hid = h.prim;
    inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    const f3 ood = {inv.x * o.x, inv.y * o.y, inv.z * o.z};
    nood = mk3(ood.x != ood.x ? fl(0x7FC00000) : -ood.x, ood.y != ood.y ? fl(0x7FC00000) : -ood.y,
               ood.z != ood.z ? fl(0x7FC00000) : -ood.z);
}

template <class STK>
__device__ __forceinline__ void ray_begin(const DScene &sc, RayState &R, f3 o, f3 d, const STK &stk)
{
    R.o = o;
    R.d = d;
    ray_setup(sc, o, d, R.t, R.hid, R.inv, R.nood);
    stack_reset(R, stk);
    R.node = sc.has_bvh ? 0 : kSentinel;
}

// Runs the traversal's own first step on a ray that has just been created: the plane and sphere tests, then the slab
// test of the root's two child boxes against the hit found so far -- same tests, same arithmetic, the NaN-rewriting slab
// test for every ray.  Returns true when nothing below the root can be hit (or there is no mesh): (t, hid) then IS the
// ray's closest hit, exactly what the extend kernel would report.  hid == PT_PRIM_MISS in that case means the ray
// leaves the scene.  On the 868 k-triangle stand-in 53 % of all rays are settled this way, 37 % of all rays escape.
__device__ __forceinline__ bool ray_settled_at_root(const DScene &sc, f3 o, f3 d, float &t, uint32_t &hid)
{
    Hit h;
    h.t = kFltMax; h.prim = PT_PRIM_MISS; h.n = mk3(0, 0, 0);
    for (uint32_t k = 0; k < sc.num_planes; k++) hit_plane(sc.planes[k], k, o, d, h);
    for (uint32_t k = 0; k < sc.num_spheres; k++) hit_sphere(sc.spheres[k], k, o, d, h);
    t = h.t;
    hid = h.prim;
    if (!sc.has_bvh) return true;
    // the slab-test constants exactly as ray_setup forms them
    const f3 inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    const f3 ood = {inv.x * o.x, inv.y * o.y, inv.z * o.z};
    const f3 nood = mk3(ood.x != ood.x ? fl(0x7FC00000) : -ood.x, ood.y != ood.y ? fl(0x7FC00000) : -ood.y,
                        ood.z != ood.z ? fl(0x7FC00000) : -ood.z);
    const float4 r0 = sc.recs[0], r1 = sc.recs[1], r2 = sc.recs[2];
    float t0;
    const bool hl = hit_aabb<true>(inv, nood, r0.x, r0.y, r0.z, r0.w, r2.x, r2.y, t, t0);
    const bool hr = hit_aabb<true>(inv, nood, r1.x, r1.y, r1.z, r1.w, r2.z, r2.w, t, t0);
    return !(hl || hr);
}

// Unit normal (reference src/render.cu:41) and reported primitive id of a hit, from the hit id extend stores.  The
// triangle normal is the cross product of two Woop rows (reference src/bvh.cuh:114), the sphere normal is (p - c) / r
// at the hit point (reference src/object.cuh:39); both are re-derived here so that traversal carries no normal around.
__device__ __forceinline__ f3 hit_normal(const DScene &sc, uint32_t hid, f3 o, f3 d, float t, uint32_t &prim)
{
    f3 n = mk3(0, 0, 0);
    prim = hid;
    if (hid == PT_PRIM_MISS) return n;
    if (hid & PT_PRIM_PLANE) {
        n = xyz(sc.planes[hid & 0x3FFFFFFFu].nrm);
    } else if (hid & PT_PRIM_SPHERE) {
        const float4 pr = sc.spheres[hid & 0x3FFFFFFFu].pos_r;
        const f3 p = {__fmaf_rn(d.x, t, o.x), __fmaf_rn(d.y, t, o.y), __fmaf_rn(d.z, t, o.z)};
        n = (p - xyz(pr)) / pr.w;
    } else {
        const float4 *rec = sc.recs + (size_t)hid * 4;
        n = cross3(xyz(rec[1]), xyz(rec[2]));
        prim = __float_as_uint(rec[3].x);
    }
    return normalize3(n);
}

__device__ __forceinline__ void ray_retire(const RayState &R, uint32_t ray, uint32_t *hid_out, float *t_out)
{
    st_stream(hid_out + ray, R.hid);
    st_stream(t_out + ray, R.t);
}

// One inner-node step (reference src/bvh.cuh:141-168): test both child boxes, descend into the nearer hit child
// (ties: left), push the farther one, or pop when neither is hit.
template <bool COUNT, bool CANON, class STK>
__device__ __forceinline__ void node_step(const DScene &sc, RayState &R, const STK &stk, float4 r0, float4 r1, float4 r2,
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
    if (both) stack_push(R, stk, far_child);
    if (!(hl || hr)) next = stack_pop(R, stk);
    if (COUNT && next < 0) R.c_leaves++;
    R.node = next;
}

// One triangle step (reference src/bvh.cuh:171-187, 93-120): test the record, go on to the next record of the leaf or
// pop after the last one.
template <bool COUNT, bool KEEPN, class STK>
__device__ __forceinline__ void tri_step(const DScene &sc, RayState &R, const STK &stk, float4 v00, float4 v11, float4 v22,
                                         float4 meta)
{
    const uint32_t flags = __float_as_uint(meta.y);
    if (COUNT && !(flags & 2u)) R.c_tris++;
    const f3 o = R.o, d = R.d;
    const float Oz = __fmaf_rn(-o.z, v00.z, __fmaf_rn(-o.y, v00.y, __fmaf_rn(-o.x, v00.x, v00.w)));
    const float Dz = __fmaf_rn(d.z, v00.z, __fmaf_rn(d.y, v00.y, d.x * v00.x));
    const float invDz = 1.0f / Dz;
    const float t = Oz * invDz;
    if (t > kEps && t < R.t) {
        const float Ox = __fmaf_rn(o.z, v11.z, __fmaf_rn(o.y, v11.y, __fmaf_rn(o.x, v11.x, v11.w)));
        const float Dx = __fmaf_rn(d.z, v11.z, __fmaf_rn(d.y, v11.y, d.x * v11.x));
        const float u = __fmaf_rn(t, Dx, Ox);
        const float Oy = __fmaf_rn(o.z, v22.z, __fmaf_rn(o.y, v22.y, __fmaf_rn(o.x, v22.x, v22.w)));
        const float Dy = __fmaf_rn(d.z, v22.z, __fmaf_rn(d.y, v22.y, d.x * v22.x));
        const float v = __fmaf_rn(t, Dy, Oy);
        if ((u >= 0.0f) & (v >= 0.0f) & (u + v <= 1.0f)) {
            R.t = t;
            R.hid = (uint32_t)R.node & kRefIndex;
            if (KEEPN) R.hn = cross3(xyz(v11), xyz(v22));      // reference src/bvh.cuh:114
        }
    }
    int next = R.node + 1;
    if (flags & 1u) {
        next = stack_pop(R, stk);
        if (COUNT && next < 0) R.c_leaves++;
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
template <bool COUNT, bool CANON, class STK>
__device__ __forceinline__ void traverse_phases(const DScene &sc, RayState &R, const STK &stk)
{
    while (R.node != kSentinel) {
        while (R.node >= 0 && R.node != kSentinel) {
            const float4 *p = sc.recs + (size_t)R.node * 4;
            node_step<COUNT, CANON>(sc, R, stk, p[0], p[1], p[2], p[3]);
        }
        while (R.node < 0) {
            const float4 *p = sc.recs + (size_t)((uint32_t)R.node & kRefIndex) * 4;
            tri_step<COUNT, false>(sc, R, stk, p[0], p[1], p[2], p[3]);
        }
    }
}

extern __shared__ int g_stack_lds[];      // [lds_slots][kBlock], sized at launch

template <bool OVERFLOW, bool COUNT>
__global__ __launch_bounds__(kBlock) void k_extend_batch(DScene sc, ExtendArgs a)
{
    const Stack<OVERFLOW> stk = make_stack<OVERFLOW>(g_stack_lds, a.lds_slots, a.gstk);
    const uint32_t n = a.n_ptr ? *a.n_ptr : a.n_fixed;
    const uint32_t stride = gridDim.x * kBlock;
    RayState R;
    R.c_nodes = R.c_tris = R.c_leaves = 0;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        ray_begin(sc, R, xyz(a.ro[i]), xyz(a.rd[i]), stk);
        // wave-uniform choice: the NaN-rewriting slab test only when some ray of this wave needs it
        if (__any(slab_degenerate(R.d))) traverse_phases<COUNT, true>(sc, R, stk);
        else traverse_phases<COUNT, false>(sc, R, stk);
        ray_retire(R, i, a.hid_out, a.t_out);
    }
    flush_counts<COUNT>(R, a.cnt3);
}

// ---- persistent single-step traversal ---------------------------------------------------------------------------------
// One step of a lane that has a ray: fetch the 64 B record the ray stands at, then an inner-node or a triangle step.
template <bool COUNT, bool CANON, bool KEEPN, class STK>
__device__ __forceinline__ void stream_step(const DScene &sc, RayState &R, const STK &stk)
{
    if (R.node == kSentinel) return;
    const float4 *p = sc.recs + (size_t)((uint32_t)R.node & kRefIndex) * 4;
    const float4 r0 = p[0], r1 = p[1], r2 = p[2];
    float4 r3 = p[3];
    // Keep the whole record one batch of loads issued here: without this the compiler sinks the loads of the child
    // references into the node branch, a second memory round trip in every step.
    asm volatile("" : "+v"(r3.x), "+v"(r3.y));
    if (R.node >= 0) node_step<COUNT, CANON>(sc, R, stk, r0, r1, r2, r3);
    else tri_step<COUNT, KEEPN>(sc, R, stk, r0, r1, r2, r3);
    // Keep the record address in registers of its own until the step is over.  When the allocator lets it share
    // registers with one of the four loaded values, the next step's address computation has to wait for every
    // outstanding memory operation first (s_waitcnt vmcnt(0)) -- the streaming stores of rays retired at the last
    // refill included -- which cost 18 % of the kernel's time in a build where that happened.
    asm volatile("" ::"v"(p));
}

template <bool OVERFLOW, bool COUNT>
__global__ __launch_bounds__(kBlock) void k_extend_stream(DScene sc, ExtendArgs a)
{
    const Stack<OVERFLOW> stk = make_stack<OVERFLOW>(g_stack_lds, a.lds_slots, a.gstk);
    const uint32_t n = a.n_ptr ? *a.n_ptr : a.n_fixed;
    // wave-uniform cursor over this wave's chunks of the ray buffer: chunks first, first + step, ... below last
    uint32_t chunk = __builtin_amdgcn_readfirstlane((blockIdx.x * kBlock + threadIdx.x) >> 6);
    uint32_t chunk_step = (gridDim.x * kBlock) >> 6;
    uint32_t chunk_last = (uint32_t)(((uint64_t)n + kChunk - 1) / kChunk);
    if (a.xcd_group && (gridDim.x & 7u) == 0) {
        // Blocks are dealt round-robin over the 8 XCDs, so blocks b and b + 8 share an L2.  Give each XCD one contiguous
        // eighth of the buffer: rays that are neighbours in the buffer start close together in space, and an L2 then
        // holds the subtrees around its own eighth instead of a share of all of them.  (Speed only: nothing depends on
        // where a block really runs.)
        const uint32_t xcd = blockIdx.x & 7u;
        const uint32_t first = (uint32_t)((uint64_t)chunk_last * xcd / 8), last = (uint32_t)((uint64_t)chunk_last * (xcd + 1) / 8);
        chunk_step = (gridDim.x >> 3) * (kBlock >> 6);
        chunk = first + __builtin_amdgcn_readfirstlane((blockIdx.x >> 3) * (kBlock >> 6) + (threadIdx.x >> 6));
        chunk_last = last;
    }
    uint32_t next = chunk < chunk_last ? chunk * kChunk : 0;
    uint32_t end = chunk < chunk_last ? min(n, next + kChunk) : next;

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
                ray_retire(R, ray, a.hid_out, a.t_out);
                ray = kNoRay;
                degenerate = false;
            }
            const bool empty = ray == kNoRay;
            const unsigned long long emask = __ballot(empty);
            if (next >= end) {
                const uint64_t c2 = (uint64_t)chunk + chunk_step;
                if (c2 < chunk_last) {
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
                ray_begin(sc, R, xyz(ld_stream(a.ro + ray)), xyz(ld_stream(a.rd + ray)), stk);
                degenerate = slab_degenerate(R.d);
            }
            next += take;
            more_work = next < end || (uint64_t)chunk + chunk_step < chunk_last;
            canon = __any(degenerate);
            if (!__any(ray != kNoRay)) break;            // nothing in flight and nothing left to fetch
        }
        // kStepsPerCheck traversal steps between two looks at the refill condition (measured on MI355X, 1/2/3/4 steps:
        // 280.6/279.1/279.2/280.1 ms per frame)
        if (canon) {
#pragma unroll
            for (int s = 0; s < kStepsPerCheck; s++) stream_step<COUNT, true, false>(sc, R, stk);
        } else {
#pragma unroll
            for (int s = 0; s < kStepsPerCheck; s++) stream_step<COUNT, false, false>(sc, R, stk);
        }
    }
    flush_counts<COUNT>(R, a.cnt3);
}

}   // namespace ptd
