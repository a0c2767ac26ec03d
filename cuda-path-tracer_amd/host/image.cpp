// Film -> 16-bit RGB PNG (host).
#include <zlib.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "host_internal.h"

namespace pth {

// reference src/image.cpp:12-33: c' = (1 - exp(-c * exposure)) ^ (1 / gamma), then truncation to 16 bits
void tonemap16(const float *film, uint64_t npixels, float exposure, float gamma, uint16_t *out)
{
    const float inv_gamma = 1.0f / gamma;
    for (uint64_t i = 0; i < npixels * 3; i++) {
        float c = 1.0f - expf(-film[i] * exposure);
        c = powf(c, inv_gamma);
        out[i] = static_cast<uint16_t>(65535 * c);
    }
}

namespace {

void put_be32(std::vector<uint8_t> &v, uint32_t x)
{
    v.push_back((uint8_t)(x >> 24)); v.push_back((uint8_t)(x >> 16)); v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x);
}

void put_chunk(std::vector<uint8_t> &png, const char type[4], const uint8_t *data, size_t n)
{
    put_be32(png, (uint32_t)n);
    const size_t start = png.size();
    png.insert(png.end(), type, type + 4);
    if (n) png.insert(png.end(), data, data + n);
    put_be32(png, (uint32_t)crc32(0L, png.data() + start, (uInt)(n + 4)));
}

}   // namespace

// Colour type 2 (RGB), bit depth 16, big-endian samples, no interlace: the pixel format the reference asks lodepng
// for (src/image.cpp:36-41).  Scanlines use filter type 0 and one zlib stream.
int png16_write(const std::string &path, const uint16_t *rgb, uint32_t w, uint32_t h)
{
    std::vector<uint8_t> raw;
    raw.reserve((size_t)h * (1 + (size_t)w * 6));
    for (uint32_t y = 0; y < h; y++) {
        raw.push_back(0);
        for (uint32_t i = 0; i < w * 3; i++) {
            const uint16_t s = rgb[(size_t)y * w * 3 + i];
            raw.push_back((uint8_t)(s >> 8));
            raw.push_back((uint8_t)(s & 255));
        }
    }
    uLongf zlen = compressBound((uLong)raw.size());
    std::vector<uint8_t> z(zlen);
    if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 6) != Z_OK) return set_error("zlib compress failed");

    std::vector<uint8_t> png = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<uint8_t> ihdr;
    put_be32(ihdr, w);
    put_be32(ihdr, h);
    ihdr.push_back(16); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    put_chunk(png, "IHDR", ihdr.data(), ihdr.size());
    put_chunk(png, "IDAT", z.data(), zlen);
    put_chunk(png, "IEND", nullptr, 0);

    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) return set_error("cannot create %s", path.c_str());
    const size_t wr = std::fwrite(png.data(), 1, png.size(), f);
    std::fclose(f);
    if (wr != png.size()) return set_error("short write to %s", path.c_str());
    return 0;
}

}   // namespace pth
