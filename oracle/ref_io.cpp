/*
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Thin C entry points over two third-party sources the reference vendors and uses on either side of its hot path,
 * compiled FROM WHERE THEY LIE under /root/reference/include (nothing of them is copied into this repository):
 *   stb_image.h   stbi_loadf(file, .., 4): how the reference decodes its .hdr environment maps
 *                 (reference src/environment_map.cu:11,24)
 *   lodepng.cpp   lodepng::encode with the state the reference sets up, and the decoder
 *                 (reference src/image.cpp:35-50)
 * The result, oracle/_ref/libref_io.so, is the reference's own reader and writer: the tests hold this repository's
 * RGBE reader and PNG16 writer against it.  It exists only where /root/reference does; tests skip without it.
 * Build: oracle/Makefile, target _ref/libref_io.so.
 */
#define STB_IMAGE_IMPLEMENTATION
#include "stb_image.h"
#include "lodepng.h"

#include <cstdlib>
#include <cstring>
#include <vector>

extern "C" {

/* rgba float pixels exactly as EnvironmentMap's constructor gets them; release with ref_free */
float *ref_hdr_load(const char *path, int *w, int *h)
{
    int n = 0;
    return stbi_loadf(path, w, h, &n, 4);
}

void ref_free(void *p) { std::free(p); }

/* decodes a PNG file as 16-bit RGB (big-endian samples, as lodepng hands them out); 0 on success */
unsigned ref_png_decode_rgb16(const char *path, unsigned char **out, unsigned *w, unsigned *h)
{
    return lodepng_decode_file(out, w, h, path, LCT_RGB, 16);
}

/* encodes big-endian 16-bit RGB samples with the encoder state of reference src/image.cpp:36-43; 0 on success */
unsigned ref_png_encode_rgb16(const char *path, const unsigned char *be_rgb16, unsigned w, unsigned h)
{
    lodepng::State state;
    state.info_raw.colortype = LCT_RGB;
    state.info_raw.bitdepth = 16;
    state.info_png.color.colortype = LCT_RGB;
    state.info_png.color.bitdepth = 16;
    state.encoder.auto_convert = 0;
    std::vector<unsigned char> bytes;
    unsigned error = lodepng::encode(bytes, be_rgb16, w, h, state);
    if (error) return error;
    return lodepng::save_file(bytes, path);
}

}   /* extern "C" */
