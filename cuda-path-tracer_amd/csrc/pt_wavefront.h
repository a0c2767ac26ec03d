// The wavefront kernels of libpt_hip.so other than the traversal: path state, device-side counters, shading, and the
// bounce / take / generate / advance / film kernels (reference src/render.cu:49-242).  Private to pt_hip.hip, which
// launches them; the traversal they call into for a new ray's first step is pt_extend.h.
//
// One wavefront iteration advances every live path by one segment:
//   bounce   (logic + shade: ends paths on miss / depth limit / Russian roulette, shades the survivors whatever their
//             material and appends their continued rays, in input order, to the other path buffer; one atomic per
//             block)
//   generate (fills the rest of that buffer with new camera paths)
//   advance  (one thread: rolls the device-side counters over)
//   extend   (closest hit for every ray of the new buffer)
// Path state ping-pongs between two SoA buffers, so every state access is a coalesced 16 B-per-lane stream.  All
// counters live on the device; the host only polls them (lagging) to learn when to stop.
//
// Film accumulation uses no atomics.  A path adds to its pixel exactly once, when it ends, and what it adds then is
// known as soon as its last ray is created (throughput x environment in that ray's direction, or zero after Russian
// roulette).  So each path owns one float4 of a "sample slab" indexed [sample][pixel]; the kernel that creates a ray
// stores the ray's escape value there, overwriting the previous bounce's, and ending a path needs no lookup at all.
// Scattered float atomics run an order of magnitude below plain stores on this chip, and HBM is large enough to hold
// a slab of hundreds of samples per pixel.  When a slab's samples have all ended, one streaming kernel sums them per
// pixel in sample order and adds the sum to the film, so the film is reproducible bit for bit from run to run and
// independent of pool size and kernel scheduling.
#pragma once

#include "pt_api.h"
#include "pt_device.h"
#include "pt_extend.h"

namespace ptd {
namespace {

struct PathBuf {
    float4 *o;      // origin xyz
    float4 *d;      // direction xyz, w = film pixel index bits
    float4 *thr;    // throughput rgb, w = sample index | bounce << 27
    uint32_t *hid;  // hit id: plane / sphere / triangle record / miss  (written by extend)
    float *t;       // hit distance, FLT_MAX on a miss            (written by extend)
};

struct Counters {
    uint32_t n_in;                  // rays at the front of the current buffer: the ones extend has to trace
    uint32_t n_back_in;             // rays at the back of the current buffer, slots [P - n_back_in, P): settled at creation,
                                    // hit already stored, not for extend
    unsigned long long n_out;       // rays appended to the other buffer so far (bounce, then generate): low word = at its
                                    // front, high word = at its back.  One word so that a block claims both with ONE atomic.
    uint32_t n_escaped;             // rays of this iteration that were created and found to leave the scene at once
    uint32_t take;                  // camera paths the generate kernel starts in this iteration
    unsigned long long gen_first;   // global index of the first of them
    unsigned long long next_sample; // next global sample index to hand out
    unsigned long long total_samples;
    unsigned long long rays_total;
    unsigned long long paths_total;
    unsigned long long iter;
    unsigned long long extend_nonempty;
    unsigned long long escaped_total; // rays ended where they were created (counted in rays_total too)
    unsigned long long trav[3];     // nodes, triangles, leaves (only with COUNT)
};

__device__ __forceinline__ int mat_slot(uint32_t type)
{
    return type == PT_DIFFUSE ? 0 : (type == PT_GLOSSY ? 1 : (type == PT_MIRROR ? 2 : (type == PT_GLASS ? 3 : 0)));
}

// Material of a hit (reference src/render.cu:21-23,33-38: intersect_scene hands material and type on with the hit).
// hid: PT_PRIM_PLANE | i, PT_PRIM_SPHERE | i, or the record index of a triangle.  A scene with one mesh keeps its
// material in the scene constants; with several meshes the triangle record names its mesh.
__device__ __forceinline__ void material_of(const DScene &sc, uint32_t hid, float4 &mat, uint32_t &type)
{
    if (hid & PT_PRIM_PLANE) {
        mat = sc.planes[hid & 0x3FFFFFFFu].mat;
        type = sc.planes[hid & 0x3FFFFFFFu].type;
    } else if (hid & PT_PRIM_SPHERE) {
        mat = sc.spheres[hid & 0x3FFFFFFFu].mat;
        type = sc.spheres[hid & 0x3FFFFFFFu].type;
    } else if (sc.num_meshes <= 1) {
        mat = sc.mesh_mat;
        type = sc.mesh_type;
    } else {
        const DMesh &m = sc.meshes[__float_as_uint(sc.recs[(size_t)hid * 4 + 3].z)];
        mat = m.mat;
        type = m.type;
    }
}

// What a path adds to its pixel if its current ray leaves the scene or is cut at the depth limit: throughput times the
// environment in the ray's direction over the sample count (reference src/render.cu:62-72).  Every kernel that creates
// a ray stores this value in the path's slab entry right away; a later bounce of the same path overwrites it, so the
// entry holds the right value whenever the path ends, and ending a path costs no environment lookup.
// The division by the sample count of src/render.cu:68 is applied when the slab is summed (k_film_reduce), on the very
// same product, so the quotient is the same float; here it would be 33 instructions in kernels that have none to spare.
__device__ __forceinline__ float4 escape_value(float4 th, float4 e)
{
    return make_float4(th.x * e.x, th.y * e.y, th.z * e.z, 0.0f);
}

// --------------------------------------------------------------------------------------------------------------
// shade: reference src/render.cu:129-242.  Returns the new direction and scales the throughput, all but the Lambertian
// mixture weight: that needs the environment texel in the new direction, which the caller looks up (it stores the path's
// escape value from the same texel) and hands to lambert_weight.
// One body for all four materials, chosen per lane by `slot` (0 Lambertian, 1 glossy, 2 mirror, 3 glass): the lanes of a
// wave hit different materials, and with a body per material the wave ran the Lambertian sampling and the environment
// lookup once for each material present.  Each lane still draws its random numbers in the order its material's
// reference kernel does.
// --------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ f3 shade_dir(const DScene &sc, int slot, f3 d, f3 n3, float4 mat, Rng &rng, float4 &th,
                                        bool &lambert)
{
    f3 out = mk3(0, 0, 0);
    lambert = slot == 0;                                // Lambertian (src/render.cu:129-162)
    bool mirror = slot == 2;                            // mirror (src/render.cu:195-211)
    f3 nn = n3;                                         // the normal a reflection uses
    if (slot == 3) {                                    // glass (src/render.cu:213-242)
        const bool external = dot3(d, n3) < 0;
        nn = external ? n3 : -n3;
        const float ref_idx = external ? (1 / mat.w) : mat.w;
        const float cos_t = fminf(dot3(-d, nn), 1.0f);
        const float sin_t = sqrtf(__fmaf_rn(-cos_t, cos_t, 1.0f));
        if (ref_idx * sin_t > 1.0f || schlick(cos_t, ref_idx) > rng.next()) mirror = true;
        else out = refract_dir(d, nn, ref_idx);
    } else if (slot == 1) {                             // glossy (src/render.cu:164-193)
        if (rng.next() < 0.05f) mirror = true;          // throughput untouched
        else lambert = true;
    }
    if (mirror) out = reflect3(d, nn);
    if (slot >= 2) { th.x *= mat.x; th.y *= mat.y; th.z *= mat.z; }
    if (lambert) out = (rng.next() < 0.5f) ? diffuse_dir(n3, rng) : env_sample_lights(sc, rng);
    return out;
}

// second half of the Lambertian case (reference src/render.cu:145-151): the one-sample mixture weight, from the
// environment texel in the new direction
__device__ __forceinline__ void lambert_weight(f3 n3, f3 out, float4 mat, float4 env_out, float4 &th)
{
    const float env_pdf = env_out.w;
    const float diff_pdf = fmaxf(dot3(n3, out) / kPi, 0.0f);
    const float mixed_pdf = (env_pdf + diff_pdf) * 0.5f;
    const float s = diff_pdf / mixed_pdf;
    th.x *= mat.x * s; th.y *= mat.y * s; th.z *= mat.z * s;
}

// Shades the hit (t, hid) of the ray (o, d) of path (pixel, sample) at bounce k: material and normal of the hit, the hit
// point, the new direction (everything of shade_dir; the Lambertian weight is left to the caller, see lambert_weight).
__device__ __forceinline__ void shade_hit(const DScene &sc, f3 o, f3 d, float t, uint32_t hid, uint32_t pixel, uint32_t sample,
                                          uint32_t k, uint32_t seed0, uint32_t seed1, float4 &th, f3 &out, f3 &hitp, f3 &n3,
                                          float4 &mat, bool &lambert)
{
    uint32_t mat_type;
    material_of(sc, hid, mat, mat_type);
    const int slot = mat_slot(mat_type);
    uint32_t prim;
    n3 = hit_normal(sc, hid, o, d, t, prim);
    hitp = mk3(__fmaf_rn(d.x, t, o.x), __fmaf_rn(d.y, t, o.y), __fmaf_rn(d.z, t, o.z));
    Rng rng(pixel, sample, k, STAGE_SHADE, seed0, seed1);
    out = shade_dir(sc, slot, d, n3, mat, rng, th, lambert);
}

// --------------------------------------------------------------------------------------------------------------
// bounce = logic (reference src/render.cu:49-89) + shade of the surviving paths, in one pass over the ray buffer.
//
// The reference sorts surviving paths into one queue per material and shades each queue with its own kernel.  Here a
// path is decided and shaded by the same thread: the four material branches are short next to the memory traffic, and
// shading from queues meant reading every survivor's state a second time through an index (sparse 16 B reads that
// fetched four times the bytes they used).  Survivors are appended to the other buffer in input order; a block of
// kBounceBlock threads claims its output range with ONE atomic, because every block adds to the same counter and
// same-address atomics are what this kernel would otherwise wait for.
// --------------------------------------------------------------------------------------------------------------
// --------------------------------------------------------------------------------------------------------------
// A ray whose closest hit is known, and one step of its path.
struct PathRay {
    f3 o, d;
    float t;            // hit distance, FLT_MAX on a miss
    uint32_t hid;       // hit id as extend stores it
    float4 th;          // throughput rgb, w = sample | bounce << 27
};

enum { RAY_DROP = 0, RAY_FRONT = 1, RAY_BACK = 2, RAY_ESCAPED = 3 };

// logic (reference src/render.cu:49-89) + shade (src/render.cu:129-242) of ONE ray of a path: ends the path (miss, depth
// limit, Russian roulette) or shades the hit, stores the escape value of the ray the path continues with in the path's
// slab entry, and runs the traversal's first step on that ray.  Returns what became of the path: RAY_DROP it ended
// (its slab entry holds what it adds to the film); RAY_ESCAPED the new ray leaves the scene at once, so the path ended
// too; RAY_BACK the new ray's closest hit is a plane or sphere in front of the mesh and is known (R.t, R.hid);
// RAY_FRONT the new ray needs the tree.  In the last two cases R is the new ray.
// A camera ray (bounce 0) that missed has no escape value in the slab yet (see k_generate): it is written here.
__device__ __forceinline__ int path_step(const DScene &sc, PathRay &R, uint32_t pixel, float4 *__restrict__ slab_entry,
                                         uint32_t seed0, uint32_t seed1)
{
    const uint32_t sk = __float_as_uint(R.th.w);
    const uint32_t sample = sk & 0x07FFFFFFu, k = sk >> 27;
    float4 th = R.th;
    bool alive = false;
    const bool late_miss = R.t == kFltMax && k == 0;
    if (!(R.t == kFltMax || 2 * k >= kMaxDepth)) {
        alive = true;
        if (2 * k + 1 >= kMinDepth) {                       // Russian roulette (src/render.cu:78-86)
            Rng rng(pixel, sample, k, STAGE_LOGIC, seed0, seed1);
            const float p = fmaxf(th.x, fmaxf(th.y, th.z));
            if (rng.next() > p) {
                alive = false;
                st_stream(slab_entry, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
            } else {
                const float ip = 1 / p;
                th.x *= ip; th.y *= ip; th.z *= ip;
            }
        }
    }
    f3 out = R.d, hitp = R.o, n3 = R.o;
    float4 mat = make_float4(0, 0, 0, 0);
    bool lambert = false;
    if (alive) shade_hit(sc, R.o, R.d, R.t, R.hid, pixel, sample, k, seed0, seed1, th, out, hitp, n3, mat, lambert);
    if (alive || late_miss) {
        // one environment lookup serves both: the new ray's direction, or the direction of the camera ray that missed
        const float4 e = env_lookup(sc, out);
        if (lambert) lambert_weight(n3, out, mat, e, th);
        st_stream(slab_entry, escape_value(th, e));
    }
    if (!alive) return RAY_DROP;
    th.w = __uint_as_float(sample | ((k + 1) << 27));
    // The traversal's first step, here: a new ray that leaves the scene at once ends its path (its escape value is
    // already in the slab; the ray needs no slot, no traversal and no pass through the next bounce), one that hits a
    // plane or sphere in front of the mesh has its hit known now and skips extend.
    float nt;
    uint32_t nhid;
    const bool settled = ray_settled_at_root(sc, hitp, out, nt, nhid);
    R.o = hitp; R.d = out; R.t = nt; R.hid = nhid; R.th = th;
    return !settled ? RAY_FRONT : (nhid == PT_PRIM_MISS ? RAY_ESCAPED : RAY_BACK);
}

// threads per block = rays per output claim.  The threads of a block move in lock step (two barriers per batch), so a
// smaller block lets more of the gathers of different batches overlap; a larger one needs fewer of the same-address
// atomics, which all blocks of all kernels of a stream share.  Measured ms per frame (868 k-triangle stand-in / spheres
// only) with generate at 512: bounce 64 -> 345 / 130, 128 -> 271 / 87, 256 -> 251 / 72.6, 512 -> 260 / 75.8,
// 1024 -> 265 / 80; with bounce at 256, generate 256 -> 249 / 77.4.
constexpr int kBounceBlock = 256;

// slot of the i-th ray of a buffer whose front holds n_front rays and whose back holds the rest
__device__ __forceinline__ uint32_t buffer_slot(uint32_t i, uint32_t n_front, uint32_t n_total, uint32_t P)
{
    return i < n_front ? i : P - (n_total - n_front) + (i - n_front);
}

struct BounceIn { float4 th, d4, o4; float t; uint32_t hid; };

// the whole state in one batch of loads: a path that ends here wastes 36 of the 56 bytes, but its neighbours in the
// same cache lines need them anyway and nothing later waits for a second round trip
__device__ __forceinline__ BounceIn bounce_load(const PathBuf &A, uint32_t i, uint32_t n, uint32_t n_front, uint32_t P)
{
    BounceIn s;
    s.th = make_float4(0, 0, 0, 0); s.d4 = s.th; s.o4 = s.th;
    s.t = kFltMax;
    s.hid = PT_PRIM_MISS;
    if (i < n) {
        i = buffer_slot(i, n_front, n, P);
        s.t = ld_stream(A.t + i);
        s.th = ld_stream(A.thr + i);
        s.d4 = ld_stream(A.d + i);
        s.o4 = ld_stream(A.o + i);
        s.hid = ld_stream(A.hid + i);
    }
    return s;
}

// Appends the rays a block wants to keep to the other buffer and returns each keeper's slot.  Rays that still need
// the tree go to the front of the buffer (slots 0, 1, ...), rays whose hit was settled where they were made to its back
// (slots P-1, P-2, ...), so that extend gets a dense range of exactly the rays it has work for.  Every block adds to the
// same counter, and same-address atomics are what these kernels wait for (generate ran 1.8x faster with half as many):
// one thread claims the block's front and back ranges with ONE 64-bit atomic, and escaped rays are only counted in a
// wave-uniform register that flush_escaped adds once, when the kernel ends.
// Two barriers; s_wave/s_base are double-buffered by `par` so that consecutive calls need no third one.
template <int BLOCK>
__device__ __forceinline__ uint32_t block_append(int cls, Counters *cnt, uint32_t P, uint32_t (&s_wave)[2][BLOCK / 64],
                                                 unsigned long long (&s_base)[2], int par, uint32_t &escaped)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned long long fmask = __ballot(cls == RAY_FRONT), bmask = __ballot(cls == RAY_BACK);
    escaped += (uint32_t)__popcll(__ballot(cls == RAY_ESCAPED));
    // two counts of at most 64 each, 16 bits apiece (block sums stay below 65536)
    if (lane == 0) s_wave[par][wave] = (uint32_t)__popcll(fmask) | ((uint32_t)__popcll(bmask) << 16);
    __syncthreads();
    uint32_t before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; w++) {
        const uint32_t c = s_wave[par][w];
        before += (uint32_t)w < wave ? c : 0u;
        total += c;
    }
    if (threadIdx.x == 0) {
        const unsigned long long add = (unsigned long long)(total & 0xFFFFu) | ((unsigned long long)(total >> 16) << 32);
        s_base[par] = add ? atomicAdd(&cnt->n_out, add) : 0ull;
    }
    __syncthreads();
    const unsigned long long base = s_base[par];
    if (cls == RAY_FRONT)
        return (uint32_t)base + (before & 0xFFFFu) +
               __builtin_amdgcn_mbcnt_hi((uint32_t)(fmask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fmask, 0));
    return P - 1u - ((uint32_t)(base >> 32) + (before >> 16) +
                     __builtin_amdgcn_mbcnt_hi((uint32_t)(bmask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bmask, 0)));
}

// adds a wave's count of rays that were settled without taking a slot to the iteration's counter (once per wave and
// kernel): `escaped` is the same in every lane (block_append counts with ballots), `consumed` is each lane's own count
__device__ __forceinline__ void flush_escaped(Counters *cnt, uint32_t escaped, uint32_t consumed)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) consumed += __shfl_xor(consumed, off, 64);
    if ((threadIdx.x & 63u) == 0 && escaped + consumed) atomicAdd(&cnt->n_escaped, escaped + consumed);
}

// waves_per_eu(5): 96 VGPRs instead of 103, a fifth wave per SIMD where the kernel has the CU to itself (no spills;
// 1.8 % on a frame without a mesh, nothing on the others; at 6 the compiler spills)
__global__ __launch_bounds__(kBounceBlock) __attribute__((amdgpu_waves_per_eu(5, 8))) void k_bounce(DScene sc, PathBuf A, PathBuf B, Counters *cnt,
                                                         uint32_t P, float4 *__restrict__ slab,
                                                         uint32_t slab_first_sample, uint32_t npix, uint32_t seed0,
                                                         uint32_t seed1)
{
    __shared__ uint32_t s_wave[2][kBounceBlock / 64];
    __shared__ unsigned long long s_base[2];
    uint32_t escaped = 0;
    const uint32_t n_front = cnt->n_in;
    const uint32_t n = n_front + cnt->n_back_in;
    const uint32_t stride = gridDim.x * kBounceBlock;
    int par = 0;
    BounceIn nxt = bounce_load(A, blockIdx.x * kBounceBlock + threadIdx.x, n, n_front, P);
    // the trip count is the same for every thread of a block (barriers inside)
    for (uint32_t first = blockIdx.x * kBounceBlock; first < n; first += stride, par ^= 1) {
        const uint32_t i = first + threadIdx.x;
        // the state of the block's next batch of rays is requested before this batch is shaded, so that the streaming loads
        // of one batch overlap the gathers of the one before (3 % of a frame without triangles, nothing with)
        const BounceIn in = nxt;
        nxt = bounce_load(A, i + stride, (uint64_t)first + stride < n ? n : 0u, n_front, P);
        PathRay R;
        R.o = xyz(in.o4); R.d = xyz(in.d4); R.t = in.t; R.hid = in.hid; R.th = in.th;
        const uint32_t pixel = __float_as_uint(in.d4.w);
        int cls = RAY_DROP;
        if (i < n) {
            const uint32_t sample = __float_as_uint(in.th.w) & 0x07FFFFFFu;
            cls = path_step(sc, R, pixel, slab + (size_t)(sample - slab_first_sample) * npix + pixel, seed0, seed1);
        }
        const uint32_t dst = block_append<kBounceBlock>(cls, cnt, P, s_wave, s_base, par, escaped);
        if (cls == RAY_FRONT || cls == RAY_BACK) {
            st_stream(B.o + dst, make_float4(R.o.x, R.o.y, R.o.z, 0.0f));
            st_stream(B.d + dst, make_float4(R.d.x, R.d.y, R.d.z, in.d4.w));
            st_stream(B.thr + dst, R.th);
            if (cls == RAY_BACK) {
                st_stream(B.hid + dst, R.hid);
                st_stream(B.t + dst, R.t);
            }
        }
    }
    flush_escaped(cnt, escaped, 0u);
}

// --------------------------------------------------------------------------------------------------------------
// generate: reference src/render.cu:93-127
// --------------------------------------------------------------------------------------------------------------
// one thread, before generate: how many camera paths this iteration starts (as many as the buffer has room for and
// samples are left, at most `cap`)
__global__ void k_take(Counters *cnt, uint32_t P, uint32_t cap)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    const unsigned long long remaining = cnt->total_samples - cnt->next_sample;
    unsigned long long room = P - (uint32_t)cnt->n_out - (uint32_t)(cnt->n_out >> 32);
    if (room > cap) room = cap;
    const uint32_t take = (uint32_t)(remaining < room ? remaining : room);
    cnt->take = take;
    cnt->gen_first = cnt->next_sample;
    cnt->next_sample += take;
    cnt->paths_total += take;
}

// threads per block = paths per output claim.  512 while the kernel only made camera rays; since it also shades the camera
// hits that are settled here (gathers, like bounce) 256: 234.6 ms against 238.8 (512), 243.7 (1024), 244.6 (128) on the
// 868 k-triangle frame, 51.0 against 55.3 on the spheres-only frame.
constexpr int kGenBlock = 256;
// waves_per_eu(6): 80 VGPRs (8 B of scratch per lane) instead of 98, a sixth wave per SIMD for the gathers of the camera
// hits that are shaded here.  Measured with bounce at 5 / 6 and generate at 5 / 6 waves: 240.9 ms (5, 5), 238.5 (5, 6),
// 243.1 (6, 5), 240.6 (6, 6), 249.3 (7, 7) on the 868 k-triangle frame.
__global__ __launch_bounds__(kGenBlock) __attribute__((amdgpu_waves_per_eu(6, 8))) void k_generate(DScene sc, PathBuf B, Counters *cnt, uint32_t P, uint32_t W,
                                                     uint32_t H, FastDiv row_div, uint32_t spp_begin, uint32_t seed0,
                                                     uint32_t seed1, float4 *__restrict__ slab)
{
    __shared__ uint32_t s_wave[2][kGenBlock / 64];
    __shared__ unsigned long long s_base[2];
    uint32_t escaped = 0, consumed = 0;
    const uint32_t take = cnt->take;
    const unsigned long long first = cnt->gen_first;
    const uint32_t npix = W * H;
    const uint32_t stride = gridDim.x * kGenBlock;
    // Path j of this launch is global sample index g = first + j = s_local * npix + r (sample-major: all pixels of a sample,
    // then the next sample).  The division is done once per thread; from one trip of the loop to the next g grows by the
    // stride, so (s_local, r) follow by an addition and at most a carry.
    uint32_t s_local, r;
    {
        const unsigned long long g0 = first + blockIdx.x * kGenBlock + threadIdx.x;
        s_local = (uint32_t)udiv_wide(g0, npix);
        r = (uint32_t)(g0 - (unsigned long long)s_local * npix);
    }
    const uint32_t stride_s = udiv_small(stride, npix), stride_r = stride - stride_s * npix;
    int par = 0;
    for (uint32_t base = blockIdx.x * kGenBlock; base < take; base += stride, par ^= 1) {
        const uint32_t j = base + threadIdx.x;
        int cls = RAY_DROP;
        float nt = kFltMax;
        uint32_t nhid = PT_PRIM_MISS;
        f3 o = mk3(0, 0, 0), d = o;
        float4 th = make_float4(0, 0, 0, 0);
        uint32_t pixel = 0;
        if (j < take) {
            const uint32_t sample = spp_begin + s_local;
            uint32_t x, y;
            pixel_of(r, W, H, row_div, x, y);
            pixel = (H - y - 1) * W + x;
            Rng rng(pixel, sample, 0, STAGE_CAMERA, seed0, seed1);
            camera_ray(sc.cam, W, H, x, y, rng, o, d);
            th = make_float4(1.0f, 1.0f, 1.0f, __uint_as_float(sample));
            // a camera ray into the sky completes its path here; one that hits a plane or sphere in front of the mesh
            // gets its hit stored now
            const bool settled = ray_settled_at_root(sc, o, d, nt, nhid);
            cls = !settled ? RAY_FRONT : (nhid == PT_PRIM_MISS ? RAY_ESCAPED : RAY_BACK);
            // A camera ray that is settled here is dealt with here, exactly as the bounce kernel would one iteration later.
            // One that leaves the scene: its path ends, path_step writes its slab entry.  One whose hit is a plane or sphere
            // in front of the mesh: it is shaded now, so the path leaves this kernel with its second ray and the camera ray's
            // trip through the path buffers is saved.  (Going on in registers while the next ray is settled too was
            // measured slower: the few lanes that go round hold their wave.)
            // A camera ray that needs the tree gets its slab entry from the bounce kernel (when it is shaded, or when the
            // tree reports a miss): waves whose pixels all show the mesh skip the environment lookup altogether.
            if (cls != RAY_FRONT) {
                PathRay R;
                R.o = o; R.d = d; R.t = nt; R.hid = nhid; R.th = th;
                const bool hit = cls == RAY_BACK;
                consumed += hit ? 1u : 0u;
                const int c = path_step(sc, R, pixel, slab + (size_t)s_local * npix + pixel, seed0, seed1);
                if (hit) {
                    cls = c;
                    o = R.o; d = R.d; th = R.th; nt = R.t; nhid = R.hid;
                }
            }
        }
        const uint32_t dst = block_append<kGenBlock>(cls, cnt, P, s_wave, s_base, par, escaped);
        if (cls == RAY_FRONT || cls == RAY_BACK) {
            st_stream(B.o + dst, make_float4(o.x, o.y, o.z, 0.0f));
            st_stream(B.d + dst, make_float4(d.x, d.y, d.z, __uint_as_float(pixel)));
            st_stream(B.thr + dst, th);
            if (cls == RAY_BACK) {
                st_stream(B.hid + dst, nhid);
                st_stream(B.t + dst, nt);
            }
        }
        r += stride_r;
        s_local += stride_s;
        if (r >= npix) { r -= npix; s_local++; }
    }
    flush_escaped(cnt, escaped, consumed);
}

// one thread, after generate: the rays appended at the front become the extend kernel's input.  A ray that was settled
// where it was made (it escaped, or it hit a plane or sphere in front of the mesh) still is a closest-hit query that
// was answered -- by the first step of the traversal -- so it counts as a ray, and as one node visit when the scene has
// a tree.
// `done` points into pinned host memory: set once the buffer is empty and no samples are left, which is all the host
// needs to know while a render runs (a copy of the counters per iteration would put a blit kernel on the stream, 0.2 to
// 1 ms each beside the other streams' kernels).
__global__ void k_advance(Counters *cnt, uint32_t count_root_visits, uint32_t *done)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    const uint32_t n_front = (uint32_t)cnt->n_out, n_back = (uint32_t)(cnt->n_out >> 32);
    const uint32_t settled = n_back + cnt->n_escaped;
    cnt->n_in = n_front;
    cnt->n_back_in = n_back;
    cnt->rays_total += (unsigned long long)n_front + settled;
    cnt->escaped_total += settled;
    if (count_root_visits) cnt->trav[0] += settled;
    cnt->iter += 1;
    if (n_front) cnt->extend_nonempty += 1;
    cnt->n_out = 0;
    cnt->n_escaped = 0;
    if (n_front == 0 && n_back == 0 && cnt->next_sample >= cnt->total_samples)
        __hip_atomic_store(done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Divides a pixel's slab entries by the sample count, sums them in sample order and adds the sum to the film (float3
// per pixel, row 0 = top).
__global__ __launch_bounds__(kBlock) void k_film_reduce(const float4 *__restrict__ slab, uint32_t nsamples, uint32_t npix,
                                                        float spp_total, float *__restrict__ film)
{
    for (uint32_t p = blockIdx.x * kBlock + threadIdx.x; p < npix; p += gridDim.x * kBlock) {
        float r = 0.0f, g = 0.0f, b = 0.0f;
        const float4 *src = slab + p;
        for (uint32_t s = 0; s < nsamples; s++) {
            const float4 v = ld_stream(src + (size_t)s * npix);
            r += v.x / spp_total; g += v.y / spp_total; b += v.z / spp_total;
        }
        float *px = film + 3 * (size_t)p;
        px[0] += r; px[1] += g; px[2] += b;
    }
}

// hit ids -> (unit normal, reported primitive id) for the closest-hit entry points
__global__ __launch_bounds__(kBlock) void k_hit_expand(DScene sc, const float4 *__restrict__ ro, const float4 *__restrict__ rd,
                                                       const uint32_t *__restrict__ hid, const float *__restrict__ t,
                                                       uint32_t n, float4 *__restrict__ hit4)
{
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        uint32_t prim;
        const f3 nn = hit_normal(sc, hid[i], xyz(ro[i]), xyz(rd[i]), t[i], prim);
        hit4[i] = make_float4(nn.x, nn.y, nn.z, __uint_as_float(prim));
    }
}

// dst += src, element by element (film reduce across contexts)
__global__ __launch_bounds__(kBlock) void k_film_add(float *__restrict__ dst, const float *__restrict__ src, uint64_t n)
{
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock) dst[i] += src[i];
}

// film -> 16-bit samples: c' = (1 - exp(-c * exposure)) ^ (1 / gamma), truncated (reference src/image.cpp:12-33)
__global__ __launch_bounds__(kBlock) void k_tonemap16(const float *__restrict__ film, uint64_t n, float exposure,
                                                      float inv_gamma, uint16_t *__restrict__ out)
{
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock) {
        const float c = powf(1.0f - expf(-film[i] * exposure), inv_gamma);
        out[i] = (uint16_t)(65535 * c);
    }
}

// camera rays of one sample index in film-pixel order (parity check of the generate stage)
__global__ __launch_bounds__(kBlock) void k_primary(DScene sc, uint32_t W, uint32_t H, FastDiv row_div, uint32_t sample,
                                                    uint32_t seed0, uint32_t seed1, float4 *__restrict__ ro,
                                                    float4 *__restrict__ rd)
{
    const uint32_t npix = W * H;
    for (uint32_t r = blockIdx.x * kBlock + threadIdx.x; r < npix; r += gridDim.x * kBlock) {
        uint32_t x, y;
        pixel_of(r, W, H, row_div, x, y);
        const uint32_t pixel = (H - y - 1) * W + x;
        Rng rng(pixel, sample, 0, STAGE_CAMERA, seed0, seed1);
        f3 o, d;
        camera_ray(sc.cam, W, H, x, y, rng, o, d);
        ro[pixel] = make_float4(o.x, o.y, o.z, 0.0f);
        rd[pixel] = make_float4(d.x, d.y, d.z, 0.0f);
    }
}

// the shading path's transcendental kernels (pt_math.h) as the device evaluates them, for parity checks
__global__ __launch_bounds__(kBlock) void k_math_eval(int which, uint32_t n, const float *__restrict__ a,
                                                      const float *__restrict__ b, float *__restrict__ out0,
                                                      float *__restrict__ out1)
{
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        float r0 = 0.0f, r1 = 0.0f;
        if (which == 0) ptm::sincos_turn(a[i], &r0, &r1);
        else if (which == 1) r0 = ptm::atan2_fixed(a[i], b[i]);
        else if (which == 2) r0 = ptm::acos_fixed(a[i]);
        else r0 = ptm::pow5(a[i]);
        out0[i] = r0;
        out1[i] = r1;
    }
}

}   // namespace
}   // namespace ptd
