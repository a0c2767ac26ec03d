// Environment maps on the host: Radiance .hdr reader/writer and the importance-sampling tables.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "host_internal.h"

namespace pth {

namespace {

constexpr float kPi = 3.14159265358979323846264338327950288f;   // reference src/common.h:16

// One RGBE pixel -> float rgb, alpha 1 (reference include/stb_image.h:6930-6955 for req_comp == 4).
inline void rgbe_decode(const uint8_t *p, float *o)
{
    if (p[3] != 0) {
        const float scale = (float)std::ldexp(1.0f, (int)p[3] - (128 + 8));
        o[0] = p[0] * scale; o[1] = p[1] * scale; o[2] = p[2] * scale;
    } else {
        o[0] = o[1] = o[2] = 0.0f;
    }
    o[3] = 1.0f;
}

inline void rgbe_encode(const float *c, uint8_t *p)
{
    const float v = std::max(c[0], std::max(c[1], c[2]));
    if (!(v >= 1e-32f)) { p[0] = p[1] = p[2] = p[3] = 0; return; }
    int e;
    const float m = std::frexp(v, &e) * 256.0f / v;
    p[0] = (uint8_t)std::min(255.0f, std::max(0.0f, c[0] * m));
    p[1] = (uint8_t)std::min(255.0f, std::max(0.0f, c[1] * m));
    p[2] = (uint8_t)std::min(255.0f, std::max(0.0f, c[2] * m));
    p[3] = (uint8_t)(e + 128);
}

struct Reader {
    const std::vector<uint8_t> &d;
    size_t pos = 0;
    explicit Reader(const std::vector<uint8_t> &data) : d(data) {}
    bool eof() const { return pos >= d.size(); }
    int get() { return pos < d.size() ? d[pos++] : 0; }
    std::string line()
    {
        std::string s;
        while (!eof()) {
            const char c = (char)get();
            if (c == '\n') break;
            s.push_back(c);
        }
        return s;
    }
};

}   // namespace

int hdr_load(const std::string &path, std::vector<float> &rgba, uint32_t &w, uint32_t &h)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) return set_error("cannot open %s", path.c_str());
    std::vector<uint8_t> data((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    Reader r(data);
    const std::string magic = r.line();
    if (magic != "#?RADIANCE" && magic != "#?RGBE") return set_error("%s: not a Radiance HDR file", path.c_str());
    bool rle_rgbe = false;
    for (;;) {
        const std::string l = r.line();
        if (l.empty()) break;
        if (l == "FORMAT=32-bit_rle_rgbe") rle_rgbe = true;
        if (r.eof()) break;
    }
    if (!rle_rgbe) return set_error("%s: unsupported HDR format", path.c_str());
    const std::string dims = r.line();
    int hh = 0, ww = 0;
    if (std::sscanf(dims.c_str(), "-Y %d +X %d", &hh, &ww) != 2 || hh <= 0 || ww <= 0)
        return set_error("%s: unsupported HDR data layout '%s'", path.c_str(), dims.c_str());
    w = (uint32_t)ww;
    h = (uint32_t)hh;
    rgba.assign((size_t)w * h * 4, 0.0f);

    auto flat_from = [&](size_t first_pixel) -> int {
        for (size_t i = first_pixel; i < (size_t)w * h; i++) {
            if (r.pos + 4 > data.size()) return set_error("%s: truncated pixel data", path.c_str());
            rgbe_decode(&data[r.pos], &rgba[4 * i]);
            r.pos += 4;
        }
        return 0;
    };
    if (w < 8 || w >= 32768) return flat_from(0);

    std::vector<uint8_t> scan((size_t)w * 4);
    for (uint32_t j = 0; j < h; j++) {
        if (r.pos + 4 > data.size()) return set_error("%s: truncated scanline header", path.c_str());
        const int c1 = r.get(), c2 = r.get(), len_hi = r.get();
        if (c1 != 2 || c2 != 2 || (len_hi & 0x80)) {
            // not run-length encoded: these bytes are the first pixel of a flat image (stb_image.h:7033-7046)
            if (j != 0) return set_error("%s: mixed flat and run-length scanlines", path.c_str());
            const uint8_t px[4] = {(uint8_t)c1, (uint8_t)c2, (uint8_t)len_hi, (uint8_t)r.get()};
            rgbe_decode(px, &rgba[0]);
            return flat_from(1);
        }
        const int len = (len_hi << 8) | r.get();
        if (len != (int)w) return set_error("%s: invalid decoded scanline length", path.c_str());
        for (int k = 0; k < 4; k++) {
            uint32_t i = 0;
            while (i < w) {
                if (r.eof()) return set_error("%s: truncated run-length data", path.c_str());
                int count = r.get();
                if (count > 128) {
                    const uint8_t value = (uint8_t)r.get();
                    count -= 128;
                    if ((uint32_t)count > w - i) return set_error("%s: bad RLE data", path.c_str());
                    for (int z = 0; z < count; z++) scan[(size_t)(i++) * 4 + k] = value;
                } else {
                    if (count == 0 || (uint32_t)count > w - i) return set_error("%s: bad RLE data", path.c_str());
                    for (int z = 0; z < count; z++) scan[(size_t)(i++) * 4 + k] = (uint8_t)r.get();
                }
            }
        }
        for (uint32_t i = 0; i < w; i++) rgbe_decode(&scan[(size_t)i * 4], &rgba[4 * ((size_t)j * w + i)]);
    }
    return 0;
}

int hdr_write(const std::string &path, const float *px, uint32_t w, uint32_t h, int channels)
{
    if (channels != 3 && channels != 4) return set_error("pth_hdr_write: channels must be 3 or 4");
    if (w < 8 || w >= 32768) return set_error("pth_hdr_write: width must be in [8, 32768)");
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) return set_error("cannot create %s", path.c_str());
    std::fprintf(f, "#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %u +X %u\n", h, w);
    std::vector<uint8_t> scan((size_t)w * 4), out;
    for (uint32_t j = 0; j < h; j++) {
        for (uint32_t i = 0; i < w; i++) rgbe_encode(px + ((size_t)j * w + i) * channels, &scan[(size_t)i * 4]);
        out.clear();
        out.push_back(2); out.push_back(2); out.push_back((uint8_t)(w >> 8)); out.push_back((uint8_t)(w & 255));
        for (int k = 0; k < 4; k++) {
            uint32_t i = 0;
            while (i < w) {
                // run of equal bytes (at least 4 long) or a literal block of up to 128 bytes
                uint32_t run = 1;
                while (i + run < w && run < 127 && scan[(size_t)(i + run) * 4 + k] == scan[(size_t)i * 4 + k]) run++;
                if (run >= 4) {
                    out.push_back((uint8_t)(128 + run));
                    out.push_back(scan[(size_t)i * 4 + k]);
                    i += run;
                    continue;
                }
                uint32_t lit = 0;
                const uint32_t start = i;
                while (i < w && lit < 128) {
                    uint32_t r2 = 1;
                    while (i + r2 < w && r2 < 4 && scan[(size_t)(i + r2) * 4 + k] == scan[(size_t)i * 4 + k]) r2++;
                    if (r2 >= 4) break;
                    i++;
                    lit++;
                }
                out.push_back((uint8_t)lit);
                for (uint32_t z = 0; z < lit; z++) out.push_back(scan[(size_t)(start + z) * 4 + k]);
            }
        }
        if (std::fwrite(out.data(), 1, out.size(), f) != out.size()) {
            std::fclose(f);
            return set_error("short write to %s", path.c_str());
        }
    }
    std::fclose(f);
    return 0;
}

// Marginal (over columns) and conditional (over rows of a column, bottom row first) distributions of
// luminance * sin(theta), their pdf written to alpha, and inverse-CDF lookup tables: reference
// src/environment_map.cu:44-125.  The running sums are fp32 and accumulate in the reference's order; the sines depend
// on the row only and are tabulated once.
int env_build(float *data, uint32_t w32, uint32_t h32, uint64_t *marginal, uint64_t *conditional)
{
    if (!data || !marginal || !conditional || w32 == 0 || h32 == 0) return set_error("pth_env_build: bad argument");
    const size_t width = w32, height = h32;
    std::vector<float> sin_mid(height), sin_edge(height);
    for (size_t y = 0; y < height; y++) {
        sin_mid[y] = sinf((y + 0.5f) * kPi / height);
        sin_edge[y] = sinf(y * kPi / height);
    }
    std::vector<float> mcdf(width), ccdf(width * height);
    float total = 0;
    for (size_t x = 0; x < width; x++) {
        float *col = &ccdf[x * height];
        float acc = 0;
        for (size_t y = 0; y < height; y++) {
            const float *t = data + 4 * ((height - y - 1) * width + x);
            const float lum = 0.2126f * t[0] + 0.7152f * t[1] + 0.0722f * t[2];
            acc += lum * sin_mid[y];
            col[y] = acc;
        }
        if (acc > 0)
            for (size_t y = 0; y < height; y++) col[y] /= acc;
        total += acc;
        mcdf[x] = total;
    }
    if (total > 0)
        for (size_t x = 0; x < width; x++) mcdf[x] /= total;

    const float npix = static_cast<float>(width * height);
    for (size_t x = 0; x < width; x++) {
        const float *col = &ccdf[x * height];
        const float px = (x == 0) ? mcdf[0] : (mcdf[x] - mcdf[x - 1]);
        for (size_t y = 0; y < height; y++) {
            float *alpha = data + 4 * ((height - y - 1) * width + x) + 3;
            if (sin_edge[y] == 0) { *alpha = 0; continue; }
            const float py = (y == 0) ? col[0] : (col[y] - col[y - 1]);
            *alpha = npix * (px * py) / (sin_edge[y] * 2.0f * kPi * kPi);
        }
    }

    for (size_t x = 0; x < width; x++) {
        const float *col = &ccdf[x * height];
        size_t i = 0;
        for (size_t y = 0; y < height; y++) {
            const float r = (float)(y + 1) / height;
            while (i < height && col[i] < r) i++;
            conditional[x * height + y] = std::min(i, height - 1);
        }
    }
    size_t i = 0;
    for (size_t x = 0; x < width; x++) {
        const float r = (float)(x + 1) / width;
        while (i < width && mcdf[i] < r) i++;
        marginal[x] = std::min(i, width - 1);
    }
    return 0;
}

}   // namespace pth
