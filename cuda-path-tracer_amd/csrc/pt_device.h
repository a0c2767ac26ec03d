// Device-side arithmetic of the wavefront path tracer (gfx950).  Compiled with -ffp-contract=off: the only fused
// operations are the explicit __fmaf_rn calls, so the floating-point sequence of every intersection test is fixed
// and a host restatement of it can reproduce hit ids bit for bit.  Division and sqrt are hipcc's default
// correctly rounded forms.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pt_api.h"
#include "pt_math.h"

namespace ptd {

constexpr float kPi = 3.14159265358979323846264338327950288f;   // reference src/common.h:16
constexpr float kEps = 0.001f;                                   // reference src/common.h:17
constexpr uint32_t kMinDepth = 3;                                // reference src/render.cu:8
constexpr uint32_t kMaxDepth = 16;                               // reference src/render.cu:9
constexpr float kFltMax = 3.402823466e+38f;
constexpr int kSentinel = 0x76543210;                            // bottom-of-stack marker, never a node index
constexpr int kMaxTreeDepth = 63;                                // reference src/bvh.cuh:12 (64 entries, one sentinel)

struct f3 { float x, y, z; };

__device__ __forceinline__ f3 mk3(float x, float y, float z) { return {x, y, z}; }
__device__ __forceinline__ f3 ld3(const float *p) { return {p[0], p[1], p[2]}; }
__device__ __forceinline__ f3 xyz(float4 v) { return {v.x, v.y, v.z}; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ f3 operator-(f3 a) { return {-a.x, -a.y, -a.z}; }
__device__ __forceinline__ f3 operator*(f3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ f3 operator/(f3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }

__device__ __forceinline__ float dot3(f3 a, f3 b) { return __fmaf_rn(a.z, b.z, __fmaf_rn(a.y, b.y, a.x * b.x)); }
__device__ __forceinline__ f3 cross3(f3 a, f3 b)
{
    return {__fmaf_rn(a.y, b.z, -(a.z * b.y)), __fmaf_rn(a.z, b.x, -(a.x * b.z)), __fmaf_rn(a.x, b.y, -(a.y * b.x))};
}
// reference include/cutil_math.cuh:1187-1191 with the host form of rsqrtf (1/sqrt)
__device__ __forceinline__ f3 normalize3(f3 v) { float s = 1.0f / sqrtf(dot3(v, v)); return v * s; }
// reference include/cutil_math.cuh:1289-1292
__device__ __forceinline__ f3 reflect3(f3 i, f3 n)
{
    float k = 2.0f * dot3(n, i);
    return {__fmaf_rn(-k, n.x, i.x), __fmaf_rn(-k, n.y, i.y), __fmaf_rn(-k, n.z, i.z)};
}

__device__ __forceinline__ int fi(float f) { return __float_as_int(f); }
__device__ __forceinline__ float fl(int i) { return __int_as_float(i); }

// ---- streaming accesses ------------------------------------------------------------------------------------------
// Path state, hit records and slab entries are read or written once per kernel and never again before other data has
// passed through the caches; marking those accesses non-temporal keeps them from pushing the BVH's upper levels and
// the environment map out of the 4 MB L2 of each XCD.
typedef float v4f_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld_stream(const float4 *p)
{
    const v4f_t v = __builtin_nontemporal_load(reinterpret_cast<const v4f_t *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float ld_stream(const float *p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ uint32_t ld_stream(const uint32_t *p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void st_stream(float4 *p, float4 v)
{
    v4f_t w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, reinterpret_cast<v4f_t *>(p));
}
__device__ __forceinline__ void st_stream(float *p, float v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ void st_stream(uint32_t *p, uint32_t v) { __builtin_nontemporal_store(v, p); }

// ---- scene as the kernels see it ----------------------------------------------------------------------------
struct DPlane { float4 pos; float4 nrm; float4 mat; uint32_t type; uint32_t pad[3]; };    // 64 B
struct DSphere { float4 pos_r; float4 mat; uint32_t type; uint32_t pad[3]; };            // 48 B
struct DMesh { float4 mat; uint32_t type; uint32_t pad[3]; };                             // 32 B

struct DScene {
    // 64 B records, record 0 the root.  Inner node = left xy box, right xy box, both z ranges, (left ref, right ref, 0, 0);
    // triangle = Woop rows z, x, y and (prim id, flags: 1 = last of its leaf, 2 = null record, index of the triangle's
    // mesh, 0).  A ref is a record index (30 bits); refs to leaves have the sign bit set.  Which record sits where is
    // decided at upload (see pt_scene_upload).
    const float4 *recs;
    uint32_t stack_slots;    // traversal stack slots a ray can need: tree depth + 1
    const DPlane *planes;
    const DSphere *spheres;
    uint32_t num_planes, num_spheres, has_bvh, mesh_type;
    float4 mesh_mat;         // material of the mesh when the scene has one mesh (num_meshes <= 1)
    const DMesh *meshes;     // per-mesh materials when it has several, indexed by the triangle record's mesh word
    uint32_t num_meshes;
    const float4 *env;       // [envH][envW] rgb + pdf, or NULL when envp is used
    // Packed form of the same texels: rgb as one shared-exponent word (exact: chosen only when every texel of the map
    // is a value such a word can hold, as maps decoded from Radiance .hdr files are) + the pdf, 8 B per texel in 4x4
    // tiles of 128 B.  The 2x2 footprint of a bilinear fetch then touches 1.6 lines on average instead of 2.25.
    const uint2 *envp;
    uint32_t env_tiles_x;    // tiles per tile row
    const uint16_t *marg;    // [envW]
    const uint16_t *cond;    // [envW*envH]
    uint32_t envW, envH;
    pt_camera cam;
};

struct Hit { float t; uint32_t prim; f3 n; };

// ---- counter-based generator: Philox4x32-10, counter (pixel, sample, bounce*4+stage, block), key = seed ------
enum { STAGE_CAMERA = 0, STAGE_LOGIC = 1, STAGE_SHADE = 2 };

struct Rng {
    uint32_t c0, c1, c2, c3, k0, k1;
    uint32_t b0, b1, b2, b3;
    int have;
    __device__ __forceinline__ Rng(uint32_t pixel, uint32_t sample, uint32_t bounce, uint32_t stage, uint32_t s0,
                                   uint32_t s1)
        : c0(pixel), c1(sample), c2(bounce * 4u + stage), c3(0), k0(s0), k1(s1), b0(0), b1(0), b2(0), b3(0), have(0)
    {
    }
    __device__ __forceinline__ void refill()
    {
        uint32_t x0 = c0, x1 = c1, x2 = c2, x3 = c3, a = k0, b = k1;
#pragma unroll
        for (int r = 0; r < 10; r++) {
            // one 32x32->64 multiply per product (v_mad_u64_u32) instead of separate high and low halves: 32-bit
            // integer multiplies are quarter rate here and this generator is most of the shading kernels' arithmetic
            const unsigned long long p0 = (unsigned long long)0xD2511F53u * x0, p1 = (unsigned long long)0xCD9E8D57u * x2;
            uint32_t h0 = (uint32_t)(p0 >> 32), l0 = (uint32_t)p0;
            uint32_t h1 = (uint32_t)(p1 >> 32), l1 = (uint32_t)p1;
            x0 = h1 ^ x1 ^ a; x1 = l1; x2 = h0 ^ x3 ^ b; x3 = l0;
            a += 0x9E3779B9u; b += 0xBB67AE85u;
        }
        b0 = x0; b1 = x1; b2 = x2; b3 = x3;
        c3++;
        have = 4;
    }
    // uniform in (0,1], like curand_uniform
    __device__ __forceinline__ float next()
    {
        if (have == 0) refill();
        uint32_t x = have == 4 ? b0 : (have == 3 ? b1 : (have == 2 ? b2 : b3));
        have--;
        return (float)((x >> 8) + 1u) * (1.0f / 16777216.0f);
    }
};

// ---- primitives -----------------------------------------------------------------------------------------------
// reference src/object.cuh:62-75
__device__ __forceinline__ void hit_plane(const DPlane &pl, uint32_t idx, f3 o, f3 d, Hit &h)
{
    f3 n = xyz(pl.nrm);
    float det = dot3(d, n);
    if (det < -kEps || det > kEps) {
        float t = dot3(xyz(pl.pos) - o, n) / det;
        if (t > kEps && t < h.t) { h.t = t; h.n = n; h.prim = PT_PRIM_PLANE | idx; }
    }
}

// reference src/object.cuh:27-44
__device__ __forceinline__ void hit_sphere(const DSphere &sp, uint32_t idx, f3 o, f3 d, Hit &h)
{
    f3 c = xyz(sp.pos_r);
    float rad = sp.pos_r.w;
    f3 op = c - o;
    float b = dot3(op, d);
    float disc = __fmaf_rn(rad, rad, __fmaf_rn(b, b, -dot3(op, op)));
    if (disc > 0) {
        disc = sqrtf(disc);
        float t = (b - disc) >= 0 ? (b - disc) : (b + disc);
        if (t > kEps && t < h.t) {
            h.t = t;
            f3 p = {__fmaf_rn(d.x, t, o.x), __fmaf_rn(d.y, t, o.y), __fmaf_rn(d.z, t, o.z)};
            h.n = (p - c) / rad;
            h.prim = PT_PRIM_SPHERE | idx;
        }
    }
}

// reference src/bvh.cuh:75-88; the integer compares of src/bvh.cuh:18-27 become v_min/max(3)_i32.
// The reference's hardware turns every NaN into the positive pattern 0x7FFFFFFF, which as a signed integer is above
// every number; gfx950 generates the negative pattern 0xFFC00000, which orders differently.  NaNs appear only for rays
// with a zero (or denormal-small) direction component, so CANON = true (chosen per wave when any of its rays is such
// a ray) rewrites NaNs to a positive pattern before they are compared; CANON = false is the same arithmetic without
// that step.  nood = -(o/d), with NaNs already positive.
__device__ __forceinline__ float pos_nan(float x) { return (x != x) ? fl(0x7FC00000) : x; }

template <bool CANON>
__device__ __forceinline__ bool hit_aabb(f3 inv, f3 nood, float x0, float x1, float y0, float y1, float z0, float z1,
                                         float t_max, float &t0)
{
    float lox = __fmaf_rn(x0, inv.x, nood.x);
    float hix = __fmaf_rn(x1, inv.x, nood.x);
    float loy = __fmaf_rn(y0, inv.y, nood.y);
    float hiy = __fmaf_rn(y1, inv.y, nood.y);
    float loz = __fmaf_rn(z0, inv.z, nood.z);
    float hiz = __fmaf_rn(z1, inv.z, nood.z);
    if (CANON) {
        lox = pos_nan(lox); hix = pos_nan(hix); loy = pos_nan(loy); hiy = pos_nan(hiy); loz = pos_nan(loz); hiz = pos_nan(hiz);
    }
    int b = max(max(fi(fminf(lox, hix)), fi(fminf(loy, hiy))), max(min(fi(loz), fi(hiz)), fi(kEps)));
    int e = min(min(fi(fmaxf(lox, hix)), fi(fmaxf(loy, hiy))), min(max(fi(loz), fi(hiz)), fi(t_max)));
    t0 = fl(b);
    return fl(e) >= t0;
}

// True for rays whose slab arithmetic can produce a NaN or an infinity: a direction component that is zero or so small
// that 1/d times a coordinate below 1e18 could overflow.
__device__ __forceinline__ bool slab_degenerate(f3 d)
{
    const float tiny = 1e-18f;
    return !(fabsf(d.x) >= tiny && fabsf(d.y) >= tiny && fabsf(d.z) >= tiny);
}

// reference src/bvh.cuh:93-120
__device__ __forceinline__ void hit_triangle(float4 v00, float4 v11, float4 v22, uint32_t prim, f3 o, f3 d, Hit &h)
{
    float Oz = __fmaf_rn(-o.z, v00.z, __fmaf_rn(-o.y, v00.y, __fmaf_rn(-o.x, v00.x, v00.w)));
    float Dz = __fmaf_rn(d.z, v00.z, __fmaf_rn(d.y, v00.y, d.x * v00.x));
    float invDz = 1.0f / Dz;
    float t = Oz * invDz;
    if (t > kEps && t < h.t) {
        float Ox = __fmaf_rn(o.z, v11.z, __fmaf_rn(o.y, v11.y, __fmaf_rn(o.x, v11.x, v11.w)));
        float Dx = __fmaf_rn(d.z, v11.z, __fmaf_rn(d.y, v11.y, d.x * v11.x));
        float u = __fmaf_rn(t, Dx, Ox);
        if (u >= 0.0f) {
            float Oy = __fmaf_rn(o.z, v22.z, __fmaf_rn(o.y, v22.y, __fmaf_rn(o.x, v22.x, v22.w)));
            float Dy = __fmaf_rn(d.z, v22.z, __fmaf_rn(d.y, v22.y, d.x * v22.x));
            float v = __fmaf_rn(t, Dy, Oy);
            if (v >= 0.0f && u + v <= 1.0f) {
                h.t = t;
                h.n = cross3(xyz(v11), xyz(v22));
                h.prim = prim;
            }
        }
    }
}

// ---- environment map --------------------------------------------------------------------------------------------
__device__ __forceinline__ float clamp1(float x) { return x < -1.0f ? -1.0f : (x > 1.0f ? 1.0f : x); }

__device__ __forceinline__ float4 env_texel(const DScene &sc, int x, int y)
{
    if (sc.envp) {
        const uint2 v = sc.envp[((size_t)(y >> 2) * sc.env_tiles_x + (x >> 2)) * 16 + (y & 3) * 4 + (x & 3)];
        // mantissa * 2^(E - 136); the all-zero word is black
        const float scale = v.x ? __uint_as_float(((v.x >> 24) - 9u) << 23) : 0.0f;
        return make_float4((float)(v.x & 255u) * scale, (float)((v.x >> 8) & 255u) * scale,
                           (float)((v.x >> 16) & 255u) * scale, __uint_as_float(v.y));
    }
    return sc.env[(size_t)y * sc.envW + x];
}

__device__ __forceinline__ int wrap_index(int i, int n)
{
    if (i >= -n && i < 2 * n) return i < 0 ? i + n : (i >= n ? i - n : i);
    return ((i % n) + n) % n;
}

// fp32 bilinear fetch with wrap addressing at normalised coordinates (reference src/environment_map.cu:156-162)
__device__ __forceinline__ float4 env_fetch(const DScene &sc, float phi, float theta)
{
    const int W = (int)sc.envW, H = (int)sc.envH;
    float x = __fmaf_rn(phi, (float)W, -0.5f);
    float y = __fmaf_rn(theta, (float)H, -0.5f);
    float fx0 = floorf(x), fy0 = floorf(y);
    float fx = x - fx0, fy = y - fy0;
    int ix = (int)fx0, iy = (int)fy0;
    // wrap addressing.  Coordinates in [0,1] give indices in [-1, size]; one conditional add or subtract is the modulo
    // there, and integer division (35 instructions each on this chip) is left to indices outside [-size, 2 size).
    int x0 = wrap_index(ix, W), x1 = x0 + 1 == W ? 0 : x0 + 1;
    int y0 = wrap_index(iy, H), y1 = y0 + 1 == H ? 0 : y0 + 1;
    const float4 t00 = env_texel(sc, x0, y0), t10 = env_texel(sc, x1, y0);
    const float4 t01 = env_texel(sc, x0, y1), t11 = env_texel(sc, x1, y1);
    float4 r;
#define PT_BILERP(c) { float top = __fmaf_rn(fx, t10.c - t00.c, t00.c); float bot = __fmaf_rn(fx, t11.c - t01.c, t01.c); \
                       r.c = __fmaf_rn(fy, bot - top, top); }
    PT_BILERP(x) PT_BILERP(y) PT_BILERP(z) PT_BILERP(w)
#undef PT_BILERP
    return r;
}

// reference src/environment_map.cuh:32-37,51-56
__device__ __forceinline__ float4 env_lookup(const DScene &sc, f3 dir)
{
    float phi = __fmaf_rn(ptm::atan2_fixed(dir.z, dir.x), 0.5f / kPi, 0.5f);
    float theta = ptm::acos_fixed(clamp1(dir.y)) / kPi;
    return env_fetch(sc, phi, theta);
}

// reference src/environment_map.cuh:39-49
__device__ __forceinline__ f3 env_sample_lights(const DScene &sc, Rng &rng)
{
    const uint32_t W = sc.envW, H = sc.envH;
    float f1 = rng.next() * (float)W - 0.5f;
    uint32_t i1 = f1 <= 0.0f ? 0u : (uint32_t)f1;
    if (i1 >= W) i1 = W - 1;
    uint32_t x = sc.marg[i1];
    float f2 = rng.next() * (float)H - 0.5f;
    uint32_t i2 = f2 <= 0.0f ? 0u : (uint32_t)f2;
    if (i2 >= H) i2 = H - 1;
    uint32_t y = sc.cond[(size_t)x * H + i2];
    float phi = 2.0f * (float)x * kPi / (float)W;
    float theta = (float)y * kPi / (float)H;
    float st, ct, sp, cp;
    ptm::sincos_turn(theta, &st, &ct);
    ptm::sincos_turn(phi, &sp, &cp);
    float sin_t = -st;
    return {sin_t * cp, -ct, sin_t * sp};
}

// ---- materials --------------------------------------------------------------------------------------------------
// reference src/material.cuh:11-24
__device__ __forceinline__ f3 diffuse_dir(f3 n, Rng &rng)
{
    float z = __fmaf_rn(rng.next(), 2.0f, -1.0f);
    float a = rng.next() * 2.0f * kPi;
    float r = sqrtf(__fmaf_rn(-z, z, 1.0f));
    float sa, ca;
    ptm::sincos_turn(a, &sa, &ca);
    float x = r * ca;
    float y = r * sa;
    f3 dir = mk3(x, y, z) + n;
    if (dir.x < kEps && dir.y < kEps && dir.z < kEps) return n;
    return normalize3(dir);
}

// reference src/material.cuh:26-32
__device__ __forceinline__ f3 refract_dir(f3 v, f3 n, float ref_idx)
{
    float cos_t = fminf(dot3(-v, n), 1.0f);
    f3 perp = mk3(__fmaf_rn(n.x, cos_t, v.x), __fmaf_rn(n.y, cos_t, v.y), __fmaf_rn(n.z, cos_t, v.z)) * ref_idx;
    float k = -sqrtf(fabsf(1.0f - dot3(perp, perp)));
    f3 parallel = n * k;
    return normalize3(perp + parallel);
}

// reference src/material.cuh:34-39
__device__ __forceinline__ float schlick(float cos_t, float ref_idx)
{
    float r0 = (1 - ref_idx) / (1 + ref_idx);
    r0 = r0 * r0;
    return __fmaf_rn(1 - r0, ptm::pow5(1 - cos_t), r0);
}

// ---- camera -------------------------------------------------------------------------------------------------------
// reference src/camera.cuh:39-53 and src/render.cu:114-123
__device__ __forceinline__ void camera_ray(const pt_camera &c, uint32_t W, uint32_t H, uint32_t x, uint32_t y, Rng &rng,
                                           f3 &o, f3 &d)
{
    float u = (float(x) + rng.next() - .5f) / float(W - 1);
    float v = (float(y) + rng.next() - .5f) / float(H - 1);
    f3 bl = ld3(c.bottom_left), right = ld3(c.right), up = ld3(c.up);
    f3 target = {__fmaf_rn(up.x, v, __fmaf_rn(right.x, u, bl.x)), __fmaf_rn(up.y, v, __fmaf_rn(right.y, u, bl.y)),
                 __fmaf_rn(up.z, v, __fmaf_rn(right.z, u, bl.z))};
    if (c.lens_radius < kEps) {
        o = ld3(c.origin);
        d = normalize3(target);
    } else {
        float rx, ry;
        do {
            rx = __fmaf_rn(rng.next(), 2.0f, -1.0f);
            ry = __fmaf_rn(rng.next(), 2.0f, -1.0f);
        } while (__fmaf_rn(ry, ry, rx * rx) >= 1);
        rx *= c.lens_radius;
        ry *= c.lens_radius;
        f3 cu = ld3(c.u), cv = ld3(c.v);
        f3 off = {__fmaf_rn(cv.x, ry, cu.x * rx), __fmaf_rn(cv.y, ry, cu.y * rx), __fmaf_rn(cv.z, ry, cu.z * rx)};
        o = ld3(c.origin) + off;
        d = normalize3(target - off);
    }
}

// floor(a / b) for a < 2^53, b < 2^32, b != 0 through one double-precision multiply and a one-step correction: the integer
// division sequences of this chip are 40 (32-bit) to 150 (64-bit) instructions, and the generate kernel is bound by its
// instruction count.
__device__ __forceinline__ unsigned long long udiv_wide(unsigned long long a, uint32_t b)
{
    unsigned long long q = (unsigned long long)((double)a * (1.0 / (double)b));
    // the estimate is off by at most one either way
    if (q * b > a) q--;
    else if ((q + 1) * b <= a) q++;
    return q;
}
__device__ __forceinline__ uint32_t udiv_small(uint32_t a, uint32_t b) { return (uint32_t)udiv_wide(a, b); }

// Division of a 32-bit number by a divisor that is the same for a whole launch: the host prepares a multiplier and a
// shift (round-up method: q = (t + ((n - t) >> 1)) >> shift with t = mulhi(n, mul); exact for every 32-bit n), the
// device then divides in four instructions instead of the 20 to 40 of a real division.
struct FastDiv { uint32_t mul, shift, divisor; };

inline FastDiv make_fastdiv(uint32_t d)
{
    FastDiv f{0u, 0u, d};
    if (d < 2) return f;
    uint32_t l = 0;
    while ((1ull << l) < d) l++;                               // ceil(log2 d), >= 1
    f.mul = (uint32_t)((((1ull << l) - d) << 32) / d + 1);
    f.shift = l - 1;
    return f;
}

__device__ __forceinline__ uint32_t fastdiv(uint32_t n, FastDiv f)
{
    if (f.divisor < 2) return n;                               // the same for every lane
    const uint32_t t = __umulhi(n, f.mul);
    return (t + ((n - t) >> 1)) >> f.shift;
}

// Sample-major enumeration of the film: index r in [0, W*H) -> pixel.  When the film tiles into 8x4 blocks a wave's
// 64 consecutive indices cover a 16x4 patch, which keeps primary rays of a wave close together.
// row_div divides by the tiles per row (W / 8) when the film tiles, else by W: see pixel_divisor.
inline uint32_t pixel_divisor(uint32_t W, uint32_t H) { return ((W & 7u) == 0 && (H & 3u) == 0) ? (W >> 3) : W; }

__device__ __forceinline__ void pixel_of(uint32_t r, uint32_t W, uint32_t H, FastDiv row_div, uint32_t &x, uint32_t &y)
{
    if ((W & 7u) == 0 && (H & 3u) == 0) {
        uint32_t tile = r >> 5, in = r & 31u, tpr = W >> 3;
        uint32_t ty = fastdiv(tile, row_div), tx = tile - ty * tpr;
        x = tx * 8 + (in & 7u);
        y = ty * 4 + (in >> 3);
    } else {
        y = fastdiv(r, row_div);
        x = r - y * W;
    }
}

}   // namespace ptd
