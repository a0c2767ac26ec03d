/*
 * CPU ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see oracle.h).
 *
 * Plain C++ restatement of the reference's algorithms.  Build with
 * -ffp-contract=off: every fused multiply-add below is an explicit fmaf so the
 * floating-point sequence is the same on x86 and on the GPU.  The BVH builder
 * uses NO fused operations (the reference's host builder is plain x86 code).
 *
 * "parity unpinned" except for the pins listed in oracle.h.
 */
#include "oracle.h"

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace {

constexpr float kPi = 3.14159265358979323846264338327950288f;   // reference src/common.h:16
constexpr float kEps = 0.001f;                                   // reference src/common.h:17
constexpr uint32_t kMinDepth = 3;                                // reference src/render.cu:8
constexpr uint32_t kMaxDepth = 16;                               // reference src/render.cu:9
constexpr int kStackSize = 64;                                   // reference src/bvh.cuh:12

struct V3 { float x, y, z; };

inline V3 v3(float x, float y, float z) { return {x, y, z}; }
inline V3 ld3(const float *p) { return {p[0], p[1], p[2]}; }
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator/(V3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }

inline uint32_t f2u(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
inline int32_t f2i(float f) { int32_t u; std::memcpy(&u, &f, 4); return u; }
inline float u2f(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

// ---- arithmetic shared (by specification) with the device code -------------------------------------------
// dot = fma(a.z,b.z, fma(a.y,b.y, a.x*b.x))
inline float dot3(V3 a, V3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
// cross component = fma(p,q, -(r*s))
inline V3 cross3(V3 a, V3 b)
{
    return {fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x))};
}
// normalize = v * (1/sqrt(dot)) : the host fallback of rsqrtf, reference include/cutil_math.cuh:55-58,1187-1191
inline V3 normalize3(V3 v) { float s = 1.0f / sqrtf(dot3(v, v)); return v * s; }
// reflect = i - 2*n*dot(n,i), reference include/cutil_math.cuh:1289-1292
inline V3 reflect3(V3 i, V3 n) { float k = 2.0f * dot3(n, i); return {fmaf(-k, n.x, i.x), fmaf(-k, n.y, i.y), fmaf(-k, n.z, i.z)}; }

// Float min/max with the semantics of the GPU's v_min_f32/v_max_f32 (and of PTX min/max.f32):
// a NaN operand yields the other operand, and -0 orders below +0.
inline float gmin(float a, float b)
{
    if (a != a) return b;
    if (b != b) return a;
    if (a == 0.0f && b == 0.0f) return (f2u(a) | f2u(b)) & 0x80000000u ? -0.0f : 0.0f;
    return a < b ? a : b;
}
inline float gmax(float a, float b)
{
    if (a != a) return b;
    if (b != b) return a;
    if (a == 0.0f && b == 0.0f) return (f2u(a) & f2u(b)) & 0x80000000u ? -0.0f : 0.0f;
    return a > b ? a : b;
}
// The reference compares float bit patterns as signed integers (src/bvh.cuh:18-27).  A NaN produced on the GPU
// is a positive quiet NaN, which as an integer is above every non-NaN pattern; x86 produces a negative one, so
// NaNs are made positive before any integer comparison.
inline int32_t ibits(float f) { return (f != f) ? 0x7FFFFFFF : f2i(f); }
inline float ifl(int32_t i) { return u2f((uint32_t)i); }
inline float fmin_fmax(float a, float b, float c) { return ifl(std::max(std::min(ibits(a), ibits(b)), ibits(c))); }
inline float fmax_fmin(float a, float b, float c) { return ifl(std::min(std::max(ibits(a), ibits(b)), ibits(c))); }
inline float fmax_fmax(float a, float b, float c) { return ifl(std::max(std::max(ibits(a), ibits(b)), ibits(c))); }
inline float fmin_fmin(float a, float b, float c) { return ifl(std::min(std::min(ibits(a), ibits(b)), ibits(c))); }

// ---- counter-based generator -------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11).  Counter = (pixel, sample, bounce*4 + stage, block), key = seed.
inline void philox_round(uint32_t c[4], const uint32_t k[2])
{
    uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k[0];
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k[1];
    uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
inline void philox10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]};
    uint32_t k[2] = {key[0], key[1]};
    for (int r = 0; r < 10; r++) {
        philox_round(c, k);
        k[0] += 0x9E3779B9u;
        k[1] += 0xBB67AE85u;
    }
    out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}

enum { STAGE_CAMERA = 0, STAGE_LOGIC = 1, STAGE_SHADE = 2 };

struct Rng {
    uint32_t ctr[4];
    uint32_t key[2];
    uint32_t buf[4];
    int have = 0;
    Rng(uint32_t pixel, uint32_t sample, uint32_t bounce, uint32_t stage, uint64_t seed)
    {
        ctr[0] = pixel; ctr[1] = sample; ctr[2] = bounce * 4u + stage; ctr[3] = 0;
        key[0] = (uint32_t)seed; key[1] = (uint32_t)(seed >> 32);
    }
    // uniform in (0,1], like curand_uniform
    float next()
    {
        if (have == 0) { philox10(ctr, key, buf); ctr[3]++; have = 4; }
        uint32_t x = buf[4 - have];
        have--;
        return (float)((x >> 8) + 1u) * (1.0f / 16777216.0f);
    }
};

// ---- scene access ------------------------------------------------------------------------------------------
struct Hit {
    float t;
    uint32_t prim;
    V3 n;             // un-normalised
    const float *mat; // 4 floats
    uint32_t mat_type;
};

struct Counters { uint64_t nodes = 0, tris = 0, leaves = 0; };

// reference src/object.cuh:62-75
inline void hit_plane(const orc_plane &pl, uint32_t idx, V3 o, V3 d, Hit &h)
{
    V3 n = ld3(pl.normal);
    float det = dot3(d, n);
    if (det < -kEps || det > kEps) {
        float t = dot3(ld3(pl.position) - o, n) / det;
        if (t > kEps && t < h.t) {
            h.t = t; h.n = n; h.mat = pl.material; h.mat_type = pl.material_type; h.prim = ORC_PRIM_PLANE | idx;
        }
    }
}

// reference src/object.cuh:27-44
inline void hit_sphere(const orc_sphere &sp, uint32_t idx, V3 o, V3 d, Hit &h)
{
    V3 c = ld3(sp.position);
    V3 op = c - o;
    float b = dot3(op, d);
    float disc = fmaf(sp.radius, sp.radius, fmaf(b, b, -dot3(op, op)));
    if (disc > 0) {
        disc = sqrtf(disc);
        float t = (b - disc) >= 0 ? (b - disc) : (b + disc);
        if (t > kEps && t < h.t) {
            h.t = t;
            V3 p = {fmaf(d.x, t, o.x), fmaf(d.y, t, o.y), fmaf(d.z, t, o.z)};
            h.n = (p - c) / sp.radius;
            h.mat = sp.material; h.mat_type = sp.material_type; h.prim = ORC_PRIM_SPHERE | idx;
        }
    }
}

// reference src/bvh.cuh:75-88 with the integer-compare min/max of src/bvh.cuh:18-27
inline bool hit_aabb(V3 inv, V3 nood, float x0, float x1, float y0, float y1, float z0, float z1, float t_max,
                     float &t0)
{
    float lox = fmaf(x0, inv.x, nood.x);
    float hix = fmaf(x1, inv.x, nood.x);
    float loy = fmaf(y0, inv.y, nood.y);
    float hiy = fmaf(y1, inv.y, nood.y);
    float loz = fmaf(z0, inv.z, nood.z);
    float hiz = fmaf(z1, inv.z, nood.z);
    t0 = fmax_fmax(gmin(lox, hix), gmin(loy, hiy), fmin_fmax(loz, hiz, kEps));
    float t1 = fmin_fmin(gmax(lox, hix), gmax(loy, hiy), fmax_fmin(loz, hiz, t_max));
    return t1 >= t0;
}

// reference src/bvh.cuh:93-120
inline void hit_triangle(const float *v00, const float *v11, const float *v22, V3 o, V3 d, uint32_t prim,
                         const orc_scene_desc *sc, Hit &h)
{
    float Oz = fmaf(-o.z, v00[2], fmaf(-o.y, v00[1], fmaf(-o.x, v00[0], v00[3])));
    float Dz = fmaf(d.z, v00[2], fmaf(d.y, v00[1], d.x * v00[0]));
    float invDz = 1.0f / Dz;
    float t = Oz * invDz;
    if (t > kEps && t < h.t) {
        float Ox = fmaf(o.z, v11[2], fmaf(o.y, v11[1], fmaf(o.x, v11[0], v11[3])));
        float Dx = fmaf(d.z, v11[2], fmaf(d.y, v11[1], d.x * v11[0]));
        float u = fmaf(t, Dx, Ox);
        if (u >= 0.0f) {
            float Oy = fmaf(o.z, v22[2], fmaf(o.y, v22[1], fmaf(o.x, v22[0], v22[3])));
            float Dy = fmaf(d.z, v22[2], fmaf(d.y, v22[1], d.x * v22[0]));
            float v = fmaf(t, Dy, Oy);
            if (v >= 0.0f && u + v <= 1.0f) {
                h.t = t;
                h.n = cross3(ld3(v11), ld3(v22));
                h.prim = prim;
                if (sc->meshes && sc->num_meshes > 0) {
                    // several OBJ blocks: the material of the mesh this face belongs to
                    uint64_t k = sc->num_meshes - 1;
                    while (k > 0 && sc->meshes[k].first_face > prim) k--;
                    h.mat = sc->meshes[k].material; h.mat_type = sc->meshes[k].material_type;
                } else {
                    h.mat = sc->mesh_material; h.mat_type = sc->mesh_material_type;
                }
            }
        }
    }
}

// reference src/bvh.cuh:130-189
void hit_bvh(const orc_scene_desc *sc, V3 o, V3 d, Hit &h, Counters *cnt)
{
    int stack[kStackSize];
    int sp = 0;
    int node = 0;
    V3 inv = {1.0f / d.x, 1.0f / d.y, 1.0f / d.z};
    V3 ood = {inv.x * o.x, inv.y * o.y, inv.z * o.z};
    // the device code carries -ood with NaNs made positive, so the slab fma is fma(x, inv, nood)
    V3 nood = {ood.x != ood.x ? u2f(0x7FC00000u) : -ood.x, ood.y != ood.y ? u2f(0x7FC00000u) : -ood.y,
               ood.z != ood.z ? u2f(0x7FC00000u) : -ood.z};
    const float *nodes = sc->bvh_nodes;
    const float *tris = sc->bvh_tris;

    while (sp >= 0) {
        while (node >= 0 && sp >= 0) {
            const float *nd = nodes + (size_t)node * 16;
            int32_t cl = f2i(nd[12]), cr = f2i(nd[13]);
            if (cnt) cnt->nodes++;
            float t0l, t0r;
            bool hl = hit_aabb(inv, nood, nd[0], nd[1], nd[2], nd[3], nd[8], nd[9], h.t, t0l);
            bool hr = hit_aabb(inv, nood, nd[4], nd[5], nd[6], nd[7], nd[10], nd[11], h.t, t0r);
            if (hl || hr) {
                node = hl ? cl : cr;
                if (hl && hr) {
                    if (t0r < t0l) std::swap(node, cr);
                    if (sp + 1 >= kStackSize) return;   // the reference would overrun its stack here
                    stack[++sp] = cr;
                }
            } else {
                node = stack[sp--];
            }
        }
        while (node < 0 && sp >= 0) {
            if (cnt) cnt->leaves++;
            for (int a = ~node;; a += 3) {
                const float *v00 = tris + (size_t)a * 4;
                if (f2u(v00[0]) == 0x80000000u) break;
                if (cnt) cnt->tris++;
                hit_triangle(v00, v00 + 4, v00 + 8, o, d, sc->bvh_prim_ids ? sc->bvh_prim_ids[a] : (uint32_t)a, sc, h);
            }
            node = stack[sp--];
        }
    }
}

// reference src/render.cu:11-44
inline Hit closest_hit(const orc_scene_desc *sc, V3 o, V3 d, Counters *cnt)
{
    Hit h;
    h.t = FLT_MAX; h.prim = ORC_PRIM_MISS; h.n = {0, 0, 0}; h.mat = nullptr; h.mat_type = 0;
    for (uint64_t i = 0; i < sc->num_planes; i++) hit_plane(sc->planes[i], (uint32_t)i, o, d, h);
    for (uint64_t i = 0; i < sc->num_spheres; i++) hit_sphere(sc->spheres[i], (uint32_t)i, o, d, h);
    if (sc->bvh_nodes && sc->num_nodes > 0) hit_bvh(sc, o, d, h, cnt);
    return h;
}

// ---- environment map ---------------------------------------------------------------------------------------
inline float clamp1(float x) { return x < -1.0f ? -1.0f : (x > 1.0f ? 1.0f : x); }

// Bilinear, wrap-addressed, normalised-coordinate fetch (reference src/environment_map.cu:156-162), restated in
// fp32; the reference's texture unit filters with 9-bit fixed-point weights, so this boundary is unpinned.
inline void env_fetch(const orc_scene_desc *sc, float phi, float theta, float out[4])
{
    const int W = (int)sc->env_width, H = (int)sc->env_height;
    float x = fmaf(phi, (float)W, -0.5f);
    float y = fmaf(theta, (float)H, -0.5f);
    float fx0 = floorf(x), fy0 = floorf(y);
    float fx = x - fx0, fy = y - fy0;
    int ix = (int)fx0, iy = (int)fy0;
    int x0 = ((ix % W) + W) % W, x1 = (x0 + 1) % W;
    int y0 = ((iy % H) + H) % H, y1 = (y0 + 1) % H;
    const float *t00 = sc->env_texels + 4 * ((size_t)y0 * W + x0);
    const float *t10 = sc->env_texels + 4 * ((size_t)y0 * W + x1);
    const float *t01 = sc->env_texels + 4 * ((size_t)y1 * W + x0);
    const float *t11 = sc->env_texels + 4 * ((size_t)y1 * W + x1);
    for (int c = 0; c < 4; c++) {
        float top = fmaf(fx, t10[c] - t00[c], t00[c]);
        float bot = fmaf(fx, t11[c] - t01[c], t01[c]);
        out[c] = fmaf(fy, bot - top, top);
    }
}

// ---- transcendental functions with a fixed operation sequence ------------------------------------------------
// The reference calls the CUDA math library's sinf / cosf / atan2f / acosf / powf (src/material.cuh:13-21,34-39,
// src/environment_map.cuh:32-49); its roundings cannot be had here, and the C library's are not the device's either.
// Both sides therefore evaluate the same single-precision Cephes kernels (sinf/cosf, atanf, asinf) over correctly
// rounded operations only, so that every random decision of a path comes out the same on the host and on the device.
// This is the oracle's own statement of them; the device's is csrc/pt_math.h.
inline void sincos_turn(float x, float *s, float *c)        // x in [0, 2 pi]
{
    const float pio2_hi = 1.57079637050628662109375f, pio2_lo = -4.37113900018624283e-8f;
    const float k = rintf(x * 0.636619772367581343f);
    float r = fmaf(-k, pio2_hi, x);
    r = fmaf(-k, pio2_lo, r);
    const float z = r * r;
    float sp = fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
    sp = fmaf(z, sp, -1.6666654611e-1f);
    const float sn = fmaf(r * z, sp, r);
    float cp = fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
    cp = fmaf(z, cp, 4.166664568298827e-2f);
    const float cs = fmaf(z * z, cp, fmaf(z, -0.5f, 1.0f));
    const int q = (int)k & 3;
    const float s0 = (q & 1) ? cs : sn, c0 = (q & 1) ? sn : cs;
    *s = (q & 2) ? -s0 : s0;
    *c = ((q + 1) & 2) ? -c0 : c0;
}

inline float atan2_fixed(float y, float x)
{
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = ax > ay ? ax : ay, mn = ax > ay ? ay : ax;
    const float a = mx > 0.0f ? mn / mx : 0.0f;
    const bool upper = a > 0.4142135679721832275390625f;
    const float t = upper ? (a - 1.0f) / (a + 1.0f) : a;
    const float z = t * t;
    float p = fmaf(z, 8.05374449538e-2f, -1.38776856032e-1f);
    p = fmaf(z, p, 1.99777106478e-1f);
    p = fmaf(z, p, -3.33329491539e-1f);
    float r = fmaf(t * z, p, t);
    if (upper) r += 0.785398185253143310546875f;
    if (ay > ax) r = 1.57079637050628662109375f - r;
    // signs by sign bit, so that the zeros behave as atan2f's: atan2(-0, x < 0) = -pi, atan2(+-0, -0) = +-pi
    if (std::signbit(x)) r = 3.1415927410125732421875f - r;
    return std::signbit(y) ? -r : r;
}

inline float asin_half(float x)
{
    const float z = x * x;
    float p = fmaf(z, 4.2163199048e-2f, 2.4181311049e-2f);
    p = fmaf(z, p, 4.5470025998e-2f);
    p = fmaf(z, p, 7.4953002686e-2f);
    p = fmaf(z, p, 1.6666752422e-1f);
    return fmaf(x * z, p, x);
}

inline float acos_fixed(float x)                              // x in [-1, 1]
{
    if (x > 0.5f) return 2.0f * asin_half(sqrtf(fmaf(-0.5f, x, 0.5f)));
    if (x < -0.5f) return 3.1415927410125732421875f - 2.0f * asin_half(sqrtf(fmaf(0.5f, x, 0.5f)));
    return 1.57079637050628662109375f - asin_half(x);
}

inline float pow5(float x)
{
    const float x2 = x * x;
    return x2 * x2 * x;
}

// reference src/environment_map.cuh:32-37,51-56 (acos argument clamped: the reference leaves |y|>1 undefined)
inline void env_lookup(const orc_scene_desc *sc, V3 dir, float out[4])
{
    float phi = fmaf(atan2_fixed(dir.z, dir.x), 0.5f / kPi, 0.5f);
    float theta = acos_fixed(clamp1(dir.y)) / kPi;
    env_fetch(sc, phi, theta, out);
}

// reference src/environment_map.cuh:39-49
inline V3 env_sample_lights(const orc_scene_desc *sc, Rng &rng)
{
    const uint32_t W = sc->env_width, H = sc->env_height;
    float f1 = rng.next() * (float)W - 0.5f;
    uint32_t i1 = f1 <= 0.0f ? 0u : (uint32_t)f1;
    if (i1 >= W) i1 = W - 1;
    uint64_t x = sc->env_marginal[i1];
    float f2 = rng.next() * (float)H - 0.5f;
    uint32_t i2 = f2 <= 0.0f ? 0u : (uint32_t)f2;
    if (i2 >= H) i2 = H - 1;
    uint64_t y = sc->env_conditional[x * H + i2];
    float phi = 2.0f * (float)x * kPi / (float)W;
    float theta = (float)y * kPi / (float)H;
    float st, ct, sp, cp;
    sincos_turn(theta, &st, &ct);
    sincos_turn(phi, &sp, &cp);
    float sin_t = -st;
    return {sin_t * cp, -ct, sin_t * sp};
}

// ---- materials ---------------------------------------------------------------------------------------------
// reference src/material.cuh:11-24
inline V3 diffuse_dir(V3 n, Rng &rng)
{
    float z = fmaf(rng.next(), 2.0f, -1.0f);
    float a = rng.next() * 2.0f * kPi;
    float r = sqrtf(fmaf(-z, z, 1.0f));
    float sa, ca;
    sincos_turn(a, &sa, &ca);
    float x = r * ca;
    float y = r * sa;
    V3 dir = v3(x, y, z) + n;
    if (dir.x < kEps && dir.y < kEps && dir.z < kEps) return n;
    return normalize3(dir);
}

// reference src/material.cuh:26-32
inline V3 refract_dir(V3 v, V3 n, float ref_idx)
{
    float cos_t = fminf(dot3(-v, n), 1.0f);
    V3 perp = v3(fmaf(n.x, cos_t, v.x), fmaf(n.y, cos_t, v.y), fmaf(n.z, cos_t, v.z)) * ref_idx;
    float k = -sqrtf(fabsf(1.0f - dot3(perp, perp)));
    V3 parallel = n * k;
    return normalize3(perp + parallel);
}

// reference src/material.cuh:34-39
inline float schlick(float cos_t, float ref_idx)
{
    float r0 = (1 - ref_idx) / (1 + ref_idx);
    r0 = r0 * r0;
    return fmaf(1 - r0, pow5(1 - cos_t), r0);
}

// reference src/camera.cuh:39-53 and src/render.cu:114-123
inline void camera_ray(const orc_scene_desc *sc, uint32_t W, uint32_t H, uint32_t x, uint32_t y, Rng &rng, V3 &o,
                       V3 &d)
{
    const orc_camera &c = sc->camera;
    float u = (float(x) + rng.next() - .5f) / float(W - 1);
    float v = (float(y) + rng.next() - .5f) / float(H - 1);
    V3 bl = ld3(c.bottom_left), right = ld3(c.right), up = ld3(c.up);
    V3 target = {fmaf(up.x, v, fmaf(right.x, u, bl.x)), fmaf(up.y, v, fmaf(right.y, u, bl.y)),
                 fmaf(up.z, v, fmaf(right.z, u, bl.z))};
    if (c.lens_radius < kEps) {
        o = ld3(c.origin);
        d = normalize3(target);
    } else {
        float rx, ry;
        do {
            rx = fmaf(rng.next(), 2.0f, -1.0f);
            ry = fmaf(rng.next(), 2.0f, -1.0f);
        } while (fmaf(ry, ry, rx * rx) >= 1);
        rx *= c.lens_radius;
        ry *= c.lens_radius;
        V3 cu = ld3(c.u), cv = ld3(c.v);
        V3 off = {fmaf(cv.x, ry, cu.x * rx), fmaf(cv.y, ry, cu.y * rx), fmaf(cv.z, ry, cu.z * rx)};
        o = ld3(c.origin) + off;
        d = normalize3(target - off);
    }
}

// One camera path; returns the radiance contribution (already divided by spp_total) for its pixel.
// reference src/render.cu:49-242 collapsed to one path: shade -> intersect -> logic per bounce.
void trace_path(const orc_scene_desc *sc, uint32_t W, uint32_t H, uint32_t x, uint32_t y, uint32_t pixel_index,
                uint32_t sample, uint64_t spp_total, uint64_t seed, float rgb[3], orc_stats &st, Counters &cnt)
{
    rgb[0] = rgb[1] = rgb[2] = 0.0f;
    V3 o, d;
    {
        Rng rng(pixel_index, sample, 0, STAGE_CAMERA, seed);
        camera_ray(sc, W, H, x, y, rng, o, d);
    }
    float thr[3] = {1.0f, 1.0f, 1.0f};
    const float inv_spp = (float)spp_total;
    for (uint32_t k = 0;; k++) {     // k = surface hits survived so far; the reference's depth counter is 2k here
        Hit h = closest_hit(sc, o, d, &cnt);
        st.rays++;
        // src/render.cu:62-73 : miss or depth limit -> add throughput * env(dir) / spp, path ends
        if (h.t == FLT_MAX || 2 * k >= kMaxDepth) {
            float e[4];
            env_lookup(sc, d, e);
            rgb[0] = thr[0] * e[0] / inv_spp;
            rgb[1] = thr[1] * e[1] / inv_spp;
            rgb[2] = thr[2] * e[2] / inv_spp;
            return;
        }
        // src/render.cu:78-86 : Russian roulette once the (double-incremented) depth reaches MIN_DEPTH
        if (2 * k + 1 >= kMinDepth) {
            Rng rng(pixel_index, sample, k, STAGE_LOGIC, seed);
            float p = fmaxf(thr[0], fmaxf(thr[1], thr[2]));
            if (rng.next() > p) return;
            float ip = 1 / p;
            thr[0] *= ip; thr[1] *= ip; thr[2] *= ip;
        }
        // shade: src/render.cu:129-242
        Rng rng(pixel_index, sample, k, STAGE_SHADE, seed);
        V3 n = normalize3(h.n);      // src/render.cu:41
        V3 hitp = {fmaf(d.x, h.t, o.x), fmaf(d.y, h.t, o.y), fmaf(d.z, h.t, o.z)};
        const float *mat = h.mat;
        V3 out;
        bool lambert = false;
        switch (h.mat_type) {
        case ORC_MIRROR:
            out = reflect3(d, n);
            thr[0] *= mat[0]; thr[1] *= mat[1]; thr[2] *= mat[2];
            break;
        case ORC_GLASS: {
            bool external = dot3(d, n) < 0;
            V3 nn = external ? n : -n;
            float ref_idx = external ? (1 / mat[3]) : mat[3];
            float cos_t = fminf(dot3(-d, nn), 1.0f);
            float sin_t = sqrtf(fmaf(-cos_t, cos_t, 1.0f));
            if (ref_idx * sin_t > 1.0f || schlick(cos_t, ref_idx) > rng.next()) out = reflect3(d, nn);
            else out = refract_dir(d, nn, ref_idx);
            thr[0] *= mat[0]; thr[1] *= mat[1]; thr[2] *= mat[2];
            break;
        }
        case ORC_GLOSSY:
            if (rng.next() < 0.05f) { out = reflect3(d, n); break; }   // throughput untouched, src/render.cu:178-182
            lambert = true;
            break;
        case ORC_DIFFUSE:
        default:
            lambert = true;
            break;
        }
        if (lambert) {
            out = (rng.next() < 0.5f) ? diffuse_dir(n, rng) : env_sample_lights(sc, rng);
            float e[4];
            env_lookup(sc, out, e);
            float env_pdf = e[3];
            float diff_pdf = fmaxf(dot3(n, out) / kPi, 0.0f);
            float mixed_pdf = (env_pdf + diff_pdf) * 0.5f;
            float s = diff_pdf / mixed_pdf;
            thr[0] *= mat[0] * s; thr[1] *= mat[1] * s; thr[2] *= mat[2] * s;
        }
        o = hitp;
        d = out;
    }
}

// ---- BVH build (no fused arithmetic anywhere in this section) ----------------------------------------------
struct Box {
    V3 mn{FLT_MAX, FLT_MAX, FLT_MAX}, mx{-FLT_MAX, -FLT_MAX, -FLT_MAX};     // reference src/aabb.cpp:8
};
inline float hmin(float a, float b) { return a < b ? a : b; }   // reference include/cutil_math.cuh:35-43
inline float hmax(float a, float b) { return a > b ? a : b; }
inline void box_extend(Box &b, const Box &o)                     // reference src/aabb.cpp:56-65
{
    b.mn = {hmin(b.mn.x, o.mn.x), hmin(b.mn.y, o.mn.y), hmin(b.mn.z, o.mn.z)};
    b.mx = {hmax(b.mx.x, o.mx.x), hmax(b.mx.y, o.mx.y), hmax(b.mx.z, o.mx.z)};
}
inline void box_extend(Box &b, V3 p)                             // reference src/aabb.cpp:44-53
{
    b.mn = {hmin(b.mn.x, p.x), hmin(b.mn.y, p.y), hmin(b.mn.z, p.z)};
    b.mx = {hmax(b.mx.x, p.x), hmax(b.mx.y, p.y), hmax(b.mx.z, p.z)};
}
inline Box box_union(const Box &a, const Box &b)                 // reference src/aabb.cpp:10-19
{
    Box r;
    r.mn = {hmin(a.mn.x, b.mn.x), hmin(a.mn.y, b.mn.y), hmin(a.mn.z, b.mn.z)};
    r.mx = {hmax(a.mx.x, b.mx.x), hmax(a.mx.y, b.mx.y), hmax(a.mx.z, b.mx.z)};
    return r;
}
inline float box_area(const Box &b)                              // reference src/aabb.cpp:21-25
{
    V3 d = b.mx - b.mn;
    return 2.0f * (d.x * d.y + d.x * d.z + d.y * d.z);
}
inline V3 box_centroid(const Box &b) { return (b.mn + b.mx) / 2; }   // reference src/aabb.cpp:27-30
inline int box_max_extent(const Box &b)                          // reference src/aabb.cpp:32-42
{
    V3 d = b.mx - b.mn;
    if (d.x >= d.y && d.x >= d.z) return 0;
    if (d.y >= d.x && d.y >= d.z) return 1;
    return 2;
}
inline float axis_of(V3 v, int a) { return a == 0 ? v.x : (a == 1 ? v.y : v.z); }

struct Tri { V3 v0, v1, v2; uint32_t id; };

inline Box tri_box(const Tri &t)                                 // reference src/bvh.cuh:39-49
{
    Box b;
    b.mn = {hmin(hmin(t.v0.x, t.v1.x), t.v2.x), hmin(hmin(t.v0.y, t.v1.y), t.v2.y), hmin(hmin(t.v0.z, t.v1.z), t.v2.z)};
    b.mx = {hmax(hmax(t.v0.x, t.v1.x), t.v2.x), hmax(hmax(t.v0.y, t.v1.y), t.v2.y), hmax(hmax(t.v0.z, t.v1.z), t.v2.z)};
    if (b.mx.x - b.mn.x == 0) b.mx.x += kEps;
    if (b.mx.y - b.mn.y == 0) b.mx.y += kEps;
    if (b.mx.z - b.mn.z == 0) b.mx.z += kEps;
    return b;
}

struct BNode { Box box; size_t index = 0; uint8_t num_children = 0; uint8_t axis = 0; };   // src/bvh.cuh:52-57

constexpr int kBuckets = 12;                                     // reference src/bvh.cu:13

inline int bucket_of(const Box &box, const Box &bound, int axis)  // reference src/bvh.cu:87-90,142-143
{
    float c = axis_of(box_centroid(box), axis);
    float lo = axis_of(bound.mn, axis), hi = axis_of(bound.mx, axis);
    float f = kBuckets * (c - lo) / (hi - lo);
    int b = (int)(size_t)f;
    if (b == kBuckets) b--;
    return b;
}

// reference src/bvh.cu:100-207
void build_rec(size_t start, size_t end, size_t &index, std::vector<Tri> &tris, std::vector<BNode> &nodes,
               uint32_t depth, uint32_t &max_depth)
{
    BNode *node = &nodes[index++];
    if (depth > max_depth) max_depth = depth;
    Box cbounds, bounds;
    for (size_t i = start; i < end; i++) {
        Box b = tri_box(tris[i]);
        box_extend(bounds, b);
        box_extend(cbounds, box_centroid(b));
    }
    int axis = box_max_extent(cbounds);
    node->axis = (uint8_t)axis;
    size_t n = end - start;
    if (n == 1) {
        node->num_children = 1;
        node->index = start;
        node->box = tri_box(tris[start]);
    } else if (n == 2) {
        node->num_children = 2;
        float ca = axis_of(box_centroid(tri_box(tris[start])), axis);
        float cb = axis_of(box_centroid(tri_box(tris[start + 1])), axis);
        if (!(ca < cb)) std::swap(tris[start], tris[start + 1]);
        node->index = start;
        node->box = box_union(tri_box(tris[start]), tri_box(tris[start + 1]));
    } else {
        node->num_children = 0;
        size_t mid;
        float ext = axis_of(cbounds.mx, axis) - axis_of(cbounds.mn, axis);
        if (!(ext > 0.0f)) {
            // All centroids coincide on the chosen axis: the reference divides 0/0 here and never terminates.
            // Median split keeps the build defined; it cannot trigger on meshes the reference can build.
            mid = start + n / 2;
        } else {
            struct { Box box; size_t count = 0; } buckets[kBuckets];
            for (size_t i = start; i < end; i++) {
                Box b = tri_box(tris[i]);
                int k = bucket_of(b, cbounds, axis);
                buckets[k].count++;
                box_extend(buckets[k].box, b);
            }
            float costs[kBuckets - 1];
            for (int i = 0; i < kBuckets - 1; i++) {
                Box b0, b1;
                size_t n0 = 0, n1 = 0;
                for (int j = 0; j <= i; j++) { box_extend(b0, buckets[j].box); n0 += buckets[j].count; }
                for (int j = i + 1; j < kBuckets; j++) { box_extend(b1, buckets[j].box); n1 += buckets[j].count; }
                costs[i] = (n0 * box_area(b0) + n1 * box_area(b1)) / box_area(bounds) + 1;
            }
            float min_cost = costs[0];
            size_t min_idx = 0;
            for (int i = 1; i < kBuckets - 1; i++)
                if (costs[i] < min_cost) { min_cost = costs[i]; min_idx = i; }
            auto mid_it = std::partition(tris.begin() + start, tris.begin() + end, [&](const Tri &t) {
                return (size_t)bucket_of(tri_box(t), cbounds, axis) <= min_idx;
            });
            mid = mid_it - tris.begin();
            if (mid == start || mid == end) mid = start + n / 2;
        }
        size_t left = index;
        build_rec(start, mid, index, tris, nodes, depth + 1, max_depth);
        node = &nodes[left - 1];
        node->index = index;
        build_rec(mid, end, index, tris, nodes, depth + 1, max_depth);
        node->box = box_union(nodes[left].box, nodes[node->index].box);
    }
}

// General 4x4 inverse by cofactors with the reference's term order (include/linear_math.cuh:85-121); column
// major m[r + 4c].  Only rows 0..2 of the inverse are produced.  inv3x4[r*4 + c] = inverse(r, c).
#define T3(a, b, c) (m[a] * m[b] * m[c])
bool inverse_rows(const float m[16], float out[12])
{
    float inv[16];
    inv[0]  =  T3(5,10,15) - T3(5,11,14) - T3(9,6,15) + T3(9,7,14) + T3(13,6,11) - T3(13,7,10);
    inv[4]  = -m[4] * m[10] * m[15] + T3(4,11,14) + T3(8,6,15) - T3(8,7,14) - T3(12,6,11) + T3(12,7,10);
    inv[8]  =  T3(4,9,15) - T3(4,11,13) - T3(8,5,15) + T3(8,7,13) + T3(12,5,11) - T3(12,7,9);
    inv[12] = -m[4] * m[9] * m[14] + T3(4,10,13) + T3(8,5,14) - T3(8,6,13) - T3(12,5,10) + T3(12,6,9);
    inv[1]  = -m[1] * m[10] * m[15] + T3(1,11,14) + T3(9,2,15) - T3(9,3,14) - T3(13,2,11) + T3(13,3,10);
    inv[5]  =  T3(0,10,15) - T3(0,11,14) - T3(8,2,15) + T3(8,3,14) + T3(12,2,11) - T3(12,3,10);
    inv[9]  = -m[0] * m[9] * m[15] + T3(0,11,13) + T3(8,1,15) - T3(8,3,13) - T3(12,1,11) + T3(12,3,9);
    inv[13] =  T3(0,9,14) - T3(0,10,13) - T3(8,1,14) + T3(8,2,13) + T3(12,1,10) - T3(12,2,9);
    inv[2]  =  T3(1,6,15) - T3(1,7,14) - T3(5,2,15) + T3(5,3,14) + T3(13,2,7) - T3(13,3,6);
    inv[6]  = -m[0] * m[6] * m[15] + T3(0,7,14) + T3(4,2,15) - T3(4,3,14) - T3(12,2,7) + T3(12,3,6);
    inv[10] =  T3(0,5,15) - T3(0,7,13) - T3(4,1,15) + T3(4,3,13) + T3(12,1,7) - T3(12,3,5);
    inv[14] = -m[0] * m[5] * m[14] + T3(0,6,13) + T3(4,1,14) - T3(4,2,13) - T3(12,1,6) + T3(12,2,5);
    float det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    if (det == 0) return false;
    det = 1.f / det;
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 4; c++) out[r * 4 + c] = inv[r + 4 * c] * det;
    return true;
}
#undef T3

// reference src/bvh.cu:214-227
void woopify(const Tri &t, std::vector<float> &out)
{
    V3 e0 = t.v0 - t.v2, e1 = t.v1 - t.v2;
    V3 c = {e0.y * e1.z - e0.z * e1.y, e0.z * e1.x - e0.x * e1.z, e0.x * e1.y - e0.y * e1.x};   // cutil cross
    float m[16] = {e0.x, e0.y, e0.z, 0.0f, e1.x, e1.y, e1.z, 0.0f, c.x, c.y, c.z, 0.0f, t.v2.x, t.v2.y, t.v2.z, 1.0f};
    float r[12];
    if (!inverse_rows(m, r)) {
        const float id[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};   // singular -> identity, linear_math.cuh:112-113
        std::memcpy(r, id, sizeof r);
    }
    const float rows[12] = {r[8], r[9], r[10], -r[11], r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7]};
    out.insert(out.end(), rows, rows + 12);
}

}   // namespace

// ================================================================================================================
extern "C" {

uint64_t orc_fnv1a64(const void *data, uint64_t nbytes)
{
    const uint8_t *p = (const uint8_t *)data;
    uint64_t h = 0xcbf29ce484222325ull;
    for (uint64_t i = 0; i < nbytes; i++) { h ^= p[i]; h *= 0x100000001b3ull; }
    return h;
}

void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) { philox10(ctr, key, out); }

int orc_bvh_build(const float *verts, uint64_t ntris, orc_bvh *out)
{
    std::memset(out, 0, sizeof *out);
    if (ntris < 3) return -1;   // the reference needs an interior root (src/bvh.cu:248-250)
    std::vector<Tri> tris(ntris);
    for (uint64_t i = 0; i < ntris; i++) {
        const float *p = verts + i * 9;
        tris[i] = {ld3(p), ld3(p + 3), ld3(p + 6), (uint32_t)i};
    }
    std::vector<BNode> nodes(2 * ntris);
    size_t index = 0;
    uint32_t max_depth = 0;
    build_rec(0, ntris, index, tris, nodes, 0, max_depth);

    // flatten: reference src/bvh.cu:234-294
    struct Entry { size_t index, parent; bool is_left; };
    std::vector<float> gnodes;      // 16 floats per node
    std::vector<float> gtris(4, 0.0f);
    std::vector<uint32_t> gids(1, ORC_PRIM_MISS);
    std::vector<Entry> stack;
    stack.push_back({0, 0, false});
    uint64_t leaves = 0;
    auto emit_leaf = [&](const BNode *leaf) -> int32_t {
        int32_t ref = ~(int32_t)(gtris.size() / 4);
        for (size_t i = 0; i < leaf->num_children; i++) {
            const Tri &t = tris[leaf->index + i];
            gids.push_back(t.id); gids.push_back(ORC_PRIM_MISS); gids.push_back(ORC_PRIM_MISS);
            woopify(t, gtris);
        }
        for (int k = 0; k < 4; k++) gtris.push_back(u2f(0x80000000u));
        gids.push_back(ORC_PRIM_MISS);
        leaves++;
        return ref;
    };
    while (!stack.empty()) {
        Entry e = stack.back();
        stack.pop_back();
        const BNode *node = &nodes[e.index];
        const BNode *l = &nodes[e.index + 1];
        const BNode *r = &nodes[node->index];
        size_t me = gnodes.size() / 16;
        float g[16] = {l->box.mn.x, l->box.mx.x, l->box.mn.y, l->box.mx.y, r->box.mn.x, r->box.mx.x, r->box.mn.y,
                       r->box.mx.y, l->box.mn.z, l->box.mx.z, r->box.mn.z, r->box.mx.z, 0, 0, 0, 0};
        if (e.index != 0) gnodes[e.parent * 16 + (e.is_left ? 12 : 13)] = u2f((uint32_t)me);
        if (l->num_children == 0) stack.push_back({e.index + 1, me, true});
        else g[12] = u2f((uint32_t)emit_leaf(l));
        if (r->num_children == 0) stack.push_back({node->index, me, false});
        else g[13] = u2f((uint32_t)emit_leaf(r));
        gnodes.insert(gnodes.end(), g, g + 16);
    }

    out->num_nodes = gnodes.size() / 16;
    out->num_tri_float4 = gtris.size() / 4;
    out->num_tris = ntris;
    out->num_leaves = leaves;
    out->max_depth = max_depth;
    out->nodes = (float *)std::malloc(gnodes.size() * 4);
    out->tris = (float *)std::calloc(gtris.size() + 8, 4);    // the leaf loop reads two float4 past the last terminator
    out->prim_ids = (uint32_t *)std::malloc(gids.size() * 4);
    std::memcpy(out->nodes, gnodes.data(), gnodes.size() * 4);
    std::memcpy(out->tris, gtris.data(), gtris.size() * 4);
    std::memcpy(out->prim_ids, gids.data(), gids.size() * 4);
    return 0;
}

void orc_bvh_free(orc_bvh *b)
{
    std::free(b->nodes); std::free(b->tris); std::free(b->prim_ids);
    std::memset(b, 0, sizeof *b);
}

int orc_env_build(float *data, uint32_t W32, uint32_t H32, uint64_t *marginal_lookup, uint64_t *conditional_lookup)
{
    const size_t width = W32, height = H32;
    std::vector<float> mcdf(width), ccdf(width * height);
    // conditional CDF per column, reference src/environment_map.cu:55-85
    float row_sum = 0;
    for (size_t x = 0; x < width; x++) {
        float col_sum = 0;
        for (size_t y = 0; y < height; y++) {
            size_t i = 4 * ((height - y - 1) * width + x);
            float sin_theta = sinf((y + 0.5f) * kPi / height);
            float lum = 0.2126f * data[i + 0] + 0.7152f * data[i + 1] + 0.0722f * data[i + 2];
            float weight = lum * sin_theta;
            col_sum += weight;
            ccdf[x * height + y] = col_sum;
        }
        if (col_sum > 0)
            for (size_t y = 0; y < height; y++) ccdf[x * height + y] /= col_sum;
        row_sum += col_sum;
        mcdf[x] = row_sum;
    }
    if (row_sum > 0)
        for (size_t x = 0; x < width; x++) mcdf[x] /= row_sum;
    // pdf into alpha, reference src/environment_map.cu:88-104
    for (size_t x = 0; x < width; x++) {
        for (size_t y = 0; y < height; y++) {
            size_t i = 4 * ((height - y - 1) * width + x) + 3;
            float sin_theta = sinf(y * kPi / height);
            if (sin_theta == 0) { data[i] = 0; continue; }
            const float *cc = &ccdf[x * height];
            float pdf_x = (x == 0) ? mcdf[0] : (mcdf[x] - mcdf[x - 1]);
            float pdf_y = (y == 0) ? cc[0] : (cc[y] - cc[y - 1]);
            data[i] = static_cast<float>(width * height) * (pdf_x * pdf_y) / (sin_theta * 2.0f * kPi * kPi);
        }
    }
    // lookup tables, reference src/environment_map.cu:107-125 (bounds checked before the read)
    for (size_t x = 0; x < width; x++) {
        size_t i = 0;
        for (size_t y = 0; y < height; y++) {
            float r = (float)(y + 1) / height;
            while (i < height && ccdf[x * height + i] < r) i++;
            conditional_lookup[x * height + y] = std::min(i, height - 1);
        }
    }
    size_t i = 0;
    for (size_t x = 0; x < width; x++) {
        float r = (float)(x + 1) / width;
        while (i < width && mcdf[i] < r) i++;
        marginal_lookup[x] = std::min(i, width - 1);
    }
    return 0;
}

void orc_camera_make(const float pos[3], const float look_at[3], float fov, float aspect, float aperture,
                     float focus_dist, orc_camera *out)
{
    // host arithmetic, unfused, cutil helpers: reference src/camera.cuh:21-37
    auto hdot = [](V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; };
    auto hnorm = [&](V3 v) { float s = 1.0f / sqrtf(hdot(v, v)); return v * s; };
    auto hcross = [](V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; };
    float height = 2.0f * tanf(fov * kPi / 360.0f);
    float width = aspect * height;
    V3 origin = ld3(pos);
    V3 w = hnorm(origin - ld3(look_at));
    V3 u = hnorm(hcross(v3(0, 1, 0), w));
    V3 v = hcross(w, u);
    V3 right = u * focus_dist * width;
    V3 up = v * focus_dist * height;
    V3 bl = -right / 2 - up / 2 - w * focus_dist;
    auto st = [](float *d, V3 s) { d[0] = s.x; d[1] = s.y; d[2] = s.z; };
    st(out->origin, origin); st(out->bottom_left, bl); st(out->right, right); st(out->up, up);
    st(out->u, u); st(out->v, v); st(out->w, w);
    out->lens_radius = aperture / 2;
}

int orc_intersect(const orc_scene_desc *sc, uint64_t n, const float *org, const float *dir, float *t_out,
                  uint32_t *prim_out, float *normal_out, orc_stats *stats)
{
    Counters cnt;
    for (uint64_t i = 0; i < n; i++) {
        Hit h = closest_hit(sc, ld3(org + 3 * i), ld3(dir + 3 * i), &cnt);
        t_out[i] = h.t;
        prim_out[i] = h.prim;
        if (normal_out) {
            V3 nn = h.prim == ORC_PRIM_MISS ? v3(0, 0, 0) : normalize3(h.n);
            normal_out[3 * i] = nn.x; normal_out[3 * i + 1] = nn.y; normal_out[3 * i + 2] = nn.z;
        }
    }
    if (stats) {
        std::memset(stats, 0, sizeof *stats);
        stats->rays = n; stats->nodes = cnt.nodes; stats->tris = cnt.tris; stats->leaves = cnt.leaves;
    }
    return 0;
}

int orc_primary_rays(const orc_scene_desc *sc, uint32_t W, uint32_t H, uint64_t sample, uint64_t seed, float *org,
                     float *dir)
{
    for (uint32_t y = 0; y < H; y++)
        for (uint32_t x = 0; x < W; x++) {
            uint32_t pixel = (H - y - 1) * W + x;
            Rng rng(pixel, (uint32_t)sample, 0, STAGE_CAMERA, seed);
            V3 o, d;
            camera_ray(sc, W, H, x, y, rng, o, d);
            float *po = org + 3 * (size_t)pixel, *pd = dir + 3 * (size_t)pixel;
            po[0] = o.x; po[1] = o.y; po[2] = o.z;
            pd[0] = d.x; pd[1] = d.y; pd[2] = d.z;
        }
    return 0;
}

int orc_render(const orc_scene_desc *sc, uint32_t W, uint32_t H, uint64_t spp_begin, uint64_t spp_count,
               uint64_t spp_total, uint64_t seed, float *film, int nthreads, orc_stats *stats)
{
    auto t0 = std::chrono::steady_clock::now();
    if (nthreads < 1) nthreads = 1;
    std::vector<orc_stats> tstats(nthreads);
    std::vector<Counters> tcnt(nthreads);
    std::atomic<uint32_t> next_row{0};
    auto worker = [&](int tid) {
        orc_stats st{};
        Counters cnt;
        for (;;) {
            uint32_t y = next_row.fetch_add(1);
            if (y >= H) break;
            for (uint32_t x = 0; x < W; x++) {
                uint32_t pixel = (H - y - 1) * W + x;
                float acc[3] = {0, 0, 0};
                for (uint64_t s = spp_begin; s < spp_begin + spp_count; s++) {
                    float rgb[3];
                    trace_path(sc, W, H, x, y, pixel, (uint32_t)s, spp_total, seed, rgb, st, cnt);
                    st.paths++;
                    acc[0] += rgb[0]; acc[1] += rgb[1]; acc[2] += rgb[2];
                }
                float *f = film + 3 * (size_t)pixel;
                f[0] += acc[0]; f[1] += acc[1]; f[2] += acc[2];
            }
        }
        tstats[tid] = st;
        tcnt[tid] = cnt;
    };
    std::vector<std::thread> th;
    for (int i = 1; i < nthreads; i++) th.emplace_back(worker, i);
    worker(0);
    for (auto &t : th) t.join();
    if (stats) {
        std::memset(stats, 0, sizeof *stats);
        for (int i = 0; i < nthreads; i++) {
            stats->rays += tstats[i].rays; stats->paths += tstats[i].paths;
            stats->nodes += tcnt[i].nodes; stats->tris += tcnt[i].tris; stats->leaves += tcnt[i].leaves;
        }
        stats->ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    return 0;
}

void orc_tonemap16(const float *film, uint64_t npixels, float exposure, float gamma, uint16_t *out)
{
    for (uint64_t i = 0; i < npixels * 3; i++) {
        float c = 1.0f - expf(-film[i] * exposure);
        c = powf(c, 1.0f / gamma);
        out[i] = static_cast<uint16_t>(65535 * c);
    }
}

void orc_rgbe_to_float(const uint8_t *rgbe, uint64_t npixels, float *out)
{
    for (uint64_t i = 0; i < npixels; i++) {
        const uint8_t *p = rgbe + 4 * i;
        float *o = out + 4 * i;
        if (p[3] != 0) {
            float f1 = (float)ldexp(1.0f, p[3] - (int)(128 + 8));
            o[0] = p[0] * f1; o[1] = p[1] * f1; o[2] = p[2] * f1;
        } else {
            o[0] = o[1] = o[2] = 0;
        }
        o[3] = 1;
    }
}

void orc_math(int which, uint64_t n, const float *a, const float *b, float *out0, float *out1)
{
    for (uint64_t i = 0; i < n; i++) {
        switch (which) {
        case 0: sincos_turn(a[i], &out0[i], &out1[i]); break;
        case 1: out0[i] = atan2_fixed(a[i], b[i]); break;
        case 2: out0[i] = acos_fixed(a[i]); break;
        default: out0[i] = pow5(a[i]); break;
        }
    }
}

}   // extern "C"
