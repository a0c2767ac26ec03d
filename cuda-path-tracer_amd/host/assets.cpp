// Procedural stand-ins for the assets the reference's scenes name but its repository does not ship
// (background/studio.hdr, sunset.hdr, forest.hdr, obj/dragon.obj).  Closed-form and seeded by integer hashes only.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "host_internal.h"

namespace pth {

namespace {

constexpr double kPiD = 3.14159265358979323846;

inline uint32_t hash32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
inline float hash01(uint32_t a, uint32_t b, uint32_t seed) { return (hash32(a * 0x9E3779B1u ^ hash32(b + seed)) >> 8) * (1.0f / 16777216.0f); }

inline float smooth(float e0, float e1, float x)
{
    float t = (x - e0) / (e1 - e0);
    t = t < 0 ? 0 : (t > 1 ? 1 : t);
    return t * t * (3 - 2 * t);
}
// 1 inside [lo,hi], falling to 0 over `soft` outside
inline float box1d(float x, float lo, float hi, float soft) { return smooth(lo - soft, lo, x) * (1 - smooth(hi, hi + soft, x)); }

// direction of the texel centre (u,v in [0,1]) under the lat-long mapping of reference src/environment_map.cuh:32-37
inline void texel_dir(float u, float v, double d[3])
{
    const double a = 2 * kPiD * (u - 0.5), th = kPiD * v;
    d[0] = std::sin(th) * std::cos(a);
    d[1] = std::cos(th);
    d[2] = std::sin(th) * std::sin(a);
}

int make_env(const std::string &kind, const std::string &path)
{
    const uint32_t W = 2048, H = 1024;
    std::vector<float> px((size_t)W * H * 3);
    for (uint32_t j = 0; j < H; j++) {
        for (uint32_t i = 0; i < W; i++) {
            const float u = (i + 0.5f) / W, v = (j + 0.5f) / H;
            float r, g, b;
            if (kind == "studio.hdr") {
                // dim ambient, darker floor, three soft boxes and an overhead panel
                const float amb = v < 0.5f ? 0.05f : 0.02f;
                r = amb; g = amb; b = amb * 1.1f;
                const float k1 = box1d(u, 0.15f, 0.35f, 0.01f) * box1d(v, 0.20f, 0.40f, 0.01f);
                const float k2 = box1d(u, 0.60f, 0.70f, 0.01f) * box1d(v, 0.25f, 0.50f, 0.01f);
                const float k3 = box1d(u, 0.85f, 0.95f, 0.01f) * box1d(v, 0.30f, 0.45f, 0.01f);
                const float top = 1 - smooth(0.06f, 0.09f, v);
                r += 30 * k1 + 20 * k2 * 1.0f + 25 * k3 * 0.8f + 25 * top;
                g += 30 * k1 + 20 * k2 * 0.9f + 25 * k3 * 0.9f + 25 * top;
                b += 30 * k1 + 20 * k2 * 0.7f + 25 * k3 * 1.0f + 25 * top;
            } else if (kind == "sunset.hdr") {
                // gradient sky, dark ground and a small very bright sun low over the horizon
                const float el = 0.5f - v;                       // +0.5 zenith .. -0.5 nadir
                if (el >= 0) {
                    const float t = smooth(0.0f, 0.35f, el);
                    r = 1.8f * (1 - t) + 0.05f * t; g = 0.8f * (1 - t) + 0.12f * t; b = 0.35f * (1 - t) + 0.45f * t;
                } else {
                    r = 0.06f; g = 0.05f; b = 0.04f;
                }
                double d[3], s[3];
                texel_dir(u, v, d);
                texel_dir(0.30f, 0.5f - 8.0f / 180.0f, s);
                const double c = d[0] * s[0] + d[1] * s[1] + d[2] * s[2];
                const double ang = std::acos(std::min(1.0, std::max(-1.0, c))) * 180.0 / kPiD;
                const float disc = 1 - smooth(1.3f, 1.7f, (float)ang);
                const float glow = 1 - smooth(0.0f, 12.0f, (float)ang);
                r += 5000 * disc + 6 * glow * glow; g += 3800 * disc + 3 * glow * glow; b += 2200 * disc + 1 * glow * glow;
            } else {
                // "forest": green/blue gradient modulated by a blocky high-frequency pattern with rare bright gaps
                const uint32_t cx = i / 8, cy = j / 8;
                const float n = hash01(cx, cy, 0x5EEDu);
                const float fleck = hash01(cx, cy, 0x5EEDu + 77u) > 0.99f && v < 0.45f ? 50.0f : 0.0f;
                const float k = 0.2f + 2.8f * n * n;
                const float sky = 1 - smooth(0.30f, 0.55f, v);
                r = k * (0.10f + 0.15f * sky) + fleck;
                g = k * (0.35f + 0.10f * sky) + fleck;
                b = k * (0.08f + 0.55f * sky) + fleck * 0.9f;
            }
            float *p = &px[((size_t)j * W + i) * 3];
            p[0] = r; p[1] = g; p[2] = b;
        }
    }
    return hdr_write(path, px.data(), W, H, 3);
}

// A (2,3) torus knot swept by a tube whose radius is displaced by sinusoids and an integer hash; nu x nv quads,
// two triangles each.  Scaled and placed to rest on the plane y = -0.283 of the reference's scene/dragon.scene.
// scale: size of the whole figure; tube: mean tube radius in the knot's own units (its strands are 1 to 2 apart);
// relief: factor on the radial displacement.
int make_knot(const std::string &path, uint32_t nu, uint32_t nv, double scale = 0.13, double tube = 0.45, double relief = 1.0)
{
    std::vector<double> P((size_t)nu * nv * 3);
    auto curve = [](double t, double c[3]) {
        const double r = 2 + std::cos(3 * t);
        c[0] = r * std::cos(2 * t); c[1] = std::sin(3 * t); c[2] = r * std::sin(2 * t);
    };
    double ymin = 1e30;
    for (uint32_t iu = 0; iu < nu; iu++) {
        const double t = 2 * kPiD * iu / nu, h = 1e-4;
        double c[3], c0[3], c1[3];
        curve(t, c); curve(t - h, c0); curve(t + h, c1);
        double T[3], A[3];
        for (int k = 0; k < 3; k++) { T[k] = (c1[k] - c0[k]) / (2 * h); A[k] = (c1[k] - 2 * c[k] + c0[k]) / (h * h); }
        const double tl = std::sqrt(T[0] * T[0] + T[1] * T[1] + T[2] * T[2]);
        for (int k = 0; k < 3; k++) T[k] /= tl;
        const double at = A[0] * T[0] + A[1] * T[1] + A[2] * T[2];
        double N[3] = {A[0] - at * T[0], A[1] - at * T[1], A[2] - at * T[2]};
        const double nl = std::sqrt(N[0] * N[0] + N[1] * N[1] + N[2] * N[2]);
        for (int k = 0; k < 3; k++) N[k] /= nl;
        const double B[3] = {T[1] * N[2] - T[2] * N[1], T[2] * N[0] - T[0] * N[2], T[0] * N[1] - T[1] * N[0]};
        for (uint32_t iv = 0; iv < nv; iv++) {
            const double a = 2 * kPiD * iv / nv;
            const double bump = 0.10 * std::sin(9 * t + 2 * a) * std::sin(5 * a) + 0.05 * std::sin(31 * t) +
                                0.02 * (hash01(iu, iv, 0xD4A60u) * 2.0 - 1.0);
            const double rad = tube + relief * bump;
            double *p = &P[((size_t)iu * nv + iv) * 3];
            for (int k = 0; k < 3; k++) p[k] = scale * (c[k] + rad * (std::cos(a) * N[k] + std::sin(a) * B[k]));
            ymin = std::min(ymin, p[1]);
        }
    }
    FILE *f = std::fopen(path.c_str(), "w");
    if (!f) return set_error("cannot create %s", path.c_str());
    std::vector<char> buf(1 << 20);
    std::setvbuf(f, buf.data(), _IOFBF, buf.size());
    const double lift = -0.283 - ymin;
    for (size_t i = 0; i < (size_t)nu * nv; i++)
        std::fprintf(f, "v %.6f %.6f %.6f\n", P[3 * i], P[3 * i + 1] + lift, P[3 * i + 2]);
    for (uint32_t iu = 0; iu < nu; iu++) {
        const uint32_t ju = (iu + 1) % nu;
        for (uint32_t iv = 0; iv < nv; iv++) {
            const uint32_t jv = (iv + 1) % nv;
            const uint32_t a = iu * nv + iv + 1, b = ju * nv + iv + 1, c = ju * nv + jv + 1, d = iu * nv + jv + 1;
            std::fprintf(f, "f %u %u %u\nf %u %u %u\n", a, c, b, a, d, c);      // counter-clockwise seen from outside
        }
    }
    const bool ok = std::fflush(f) == 0;
    std::fclose(f);
    return ok ? 0 : set_error("short write to %s", path.c_str());
}

}   // namespace

int asset_generate(const std::string &kind, const std::string &path)
{
    if (kind == "studio.hdr" || kind == "sunset.hdr" || kind == "forest.hdr") return make_env(kind, path);
    if (kind == "dragon.obj") return make_knot(path, 2048, 212);
    if (kind == "knot_small.obj") return make_knot(path, 96, 48);
    // The same knot 2.2 times as large with twice the relief: from the camera of scene/dragon.scene it fills the frame
    // (91 % of the camera rays hit it) and a ray visits 44 inner nodes on average where the stand-in above, which the
    // camera sees from afar, gives 15: the traversal-bound case a frame-filling 800 k-triangle mesh is.
    if (kind == "dragon_dense.obj") return make_knot(path, 2048, 212, 0.29, 0.45, 2.0);
    return set_error("pth_asset_generate: unknown asset kind '%s'", kind.c_str());
}

}   // namespace pth
