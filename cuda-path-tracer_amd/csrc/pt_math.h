// Transcendental functions of the shading path with a fixed operation sequence.
//
// The reference calls sinf / cosf / atan2f / acosf / powf of the CUDA math library (src/material.cuh:13-21,34-39,
// src/environment_map.cuh:32-49); those, the device library's here and the host C library's each round differently in
// the last place, and one such ulp flips a discrete decision (a texel boundary, reflect-or-refract) on about one path in
// 1,500.  Written out as polynomial kernels over +, *, fma, division, sqrt and round-to-nearest — every one of them
// correctly rounded on the device and on a host alike — the functions below return the same bits wherever they run,
// so a host restatement of the shading can be held against the device bit for bit.  Kernels and coefficients are the
// single-precision ones of the Cephes library (sinf/cosf, atanf, asinf); measured error against double precision over
// the ranges used: sin/cos below 1.5e-7 absolute, atan2 and acos within 3 ulp.
#pragma once

#include <math.h>

#if defined(__HIPCC__)
#define PTM_FN __host__ __device__ __forceinline__
#else
#define PTM_FN static inline
#endif

namespace ptm {

// float(pi/2), pi/2 - float(pi/2), float(pi/4), float(pi)
constexpr float kPio2Hi = 1.57079637050628662109375f;
constexpr float kPio2Lo = -4.37113900018624283e-8f;
constexpr float kPio4 = 0.785398185253143310546875f;
constexpr float kPiF = 3.1415927410125732421875f;

// sin and cos of x in [0, 2 pi] (the callers' angles: 2 pi xi, 2 pi x / W, pi y / H)
PTM_FN void sincos_turn(float x, float *s, float *c)
{
    const float k = rintf(x * 0.636619772367581343f);        // nearest multiple of pi/2: 0 .. 4
    float r = fmaf(-k, kPio2Hi, x);
    r = fmaf(-k, kPio2Lo, r);                                // |r| <= pi/4
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

// atan2(y, x) in [-pi, pi]; (0, 0) gives 0.  Signs are taken from the sign bits, so the zeros behave as atan2f's do:
// atan2(-0, x < 0) = -pi, atan2(+-0, -0) = +-pi.
PTM_FN float atan2_fixed(float y, float x)
{
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = ax > ay ? ax : ay, mn = ax > ay ? ay : ax;
    const float a = mx > 0.0f ? mn / mx : 0.0f;              // tangent of the angle to the nearer axis, in [0, 1]
    const bool upper = a > 0.4142135679721832275390625f;     // tan(pi/8): atan(a) = pi/4 + atan((a - 1) / (a + 1))
    const float t = upper ? (a - 1.0f) / (a + 1.0f) : a;
    const float z = t * t;
    float p = fmaf(z, 8.05374449538e-2f, -1.38776856032e-1f);
    p = fmaf(z, p, 1.99777106478e-1f);
    p = fmaf(z, p, -3.33329491539e-1f);
    float r = fmaf(t * z, p, t);
    if (upper) r += kPio4;
    if (ay > ax) r = kPio2Hi - r;
    if (signbit(x)) r = kPiF - r;
    return copysignf(r, y);
}

PTM_FN float asin_half(float x)                              // |x| <= 0.5
{
    const float z = x * x;
    float p = fmaf(z, 4.2163199048e-2f, 2.4181311049e-2f);
    p = fmaf(z, p, 4.5470025998e-2f);
    p = fmaf(z, p, 7.4953002686e-2f);
    p = fmaf(z, p, 1.6666752422e-1f);
    return fmaf(x * z, p, x);
}

// acos(x), x in [-1, 1]
PTM_FN float acos_fixed(float x)
{
    if (x > 0.5f) return 2.0f * asin_half(sqrtf(fmaf(-0.5f, x, 0.5f)));
    if (x < -0.5f) return kPiF - 2.0f * asin_half(sqrtf(fmaf(0.5f, x, 0.5f)));
    return kPio2Hi - asin_half(x);
}

// x^5 (Schlick's approximation, reference src/material.cuh:38 pow(1 - cosine, 5))
PTM_FN float pow5(float x)
{
    const float x2 = x * x;
    return x2 * x2 * x;
}

}   // namespace ptm
