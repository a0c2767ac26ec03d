// Driver of tools/host_sanitize.sh: the host library's BVH builder (1 thread against 8, outputs compared) and its
// scene / OBJ / HDR readers on well-formed and on hostile input, to be run under TSan and ASan + UBSan.
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "pt_host.h"

static void put(const char *path, const std::string &text)
{
    FILE *f = std::fopen(path, "wb");
    std::fwrite(text.data(), 1, text.size(), f);
    std::fclose(f);
}

static int builder()
{
    if (pth_asset_generate("knot_small.obj", "/tmp/pt_sanitize_knot.obj") != 0) { std::printf("gen: %s\n", pth_last_error()); return 1; }
    float *v = nullptr; uint64_t n = 0;
    if (pth_obj_load("/tmp/pt_sanitize_knot.obj", &v, &n) != 0) return 1;
    std::vector<float> big;
    for (int k = 0; k < 12; k++)
        for (uint64_t i = 0; i < n * 9; i++) big.push_back(v[i] + ((i % 3) == 0 ? 0.37f * k : 0.0f));
    pth_bvh a{}, b{};
    if (pth_bvh_build(big.data(), n * 12, 1, &a) != 0 || pth_bvh_build(big.data(), n * 12, 8, &b) != 0) { std::printf("build: %s\n", pth_last_error()); return 1; }
    const bool same = a.num_nodes == b.num_nodes && !std::memcmp(a.nodes, b.nodes, a.num_nodes * 64) &&
                      !std::memcmp(a.tris, b.tris, a.num_tri_float4 * 16) && !std::memcmp(a.prim_ids, b.prim_ids, a.num_tri_float4 * 4);
    std::printf("tris %llu nodes %llu identical %d\n", (unsigned long long)(n * 12), (unsigned long long)a.num_nodes, (int)same);
    pth_bvh_free(&a); pth_bvh_free(&b); pth_free(v);
    return same ? 0 : 2;
}

// a scene with two OBJ blocks, then the same readers on input that must be refused without a stray access
static int readers()
{
    const char *obj_a = "/tmp/pt_sanitize_a.obj", *obj_b = "/tmp/pt_sanitize_b.obj", *hdr = "/tmp/pt_sanitize.hdr";
    put(obj_a, "# four faces\nv 0 0 0\nv 1 0 0\nv 0 1 0\nv 0 0 1\nvn 0 0 1\nf 1 2 3\nf 1//1 2//1 4//1\nf -4 -2 -1\nf 1 2 3 4\n");
    put(obj_b, "v 2 0 0\nv 3 0 0\nv 2 1 0\nv 2 0 1\nf 1 2 3\nf 1 3 4\nf 2 3 4\n");
    if (pth_asset_generate("studio.hdr", hdr) != 0) { std::printf("hdr: %s\n", pth_last_error()); return 1; }
    const std::string head = std::string("CAMERA\nposition 0 1 5\nlook_at 0 0 0\nfov 35\naperture 0.02\nfocus_dist 5\n\n"
                                         "ENVIRONMENT\nfile_name ") + hdr + "\n\n";
    const std::string two = head + "OBJ\nmaterial_type LAMBERTIAN\nalbedo 0.5 0.5 0.5\nfile_name " + obj_a +
                            "\n\nOBJ\nmaterial_type GLASS\nalbedo 1 1 1\nrefraction_index 1.5\nfile_name " + obj_b +
                            "\n\nSPHERE\nmaterial_type MIRROR\nalbedo 0.6 0.6 0.6\nposition 0 0.5 0\nradius 0.5\n";
    put("/tmp/pt_sanitize_two.scene", two);
    pth_scene *sc = nullptr;
    if (pth_scene_load("/tmp/pt_sanitize_two.scene", 64, 48, &sc) != 0) { std::printf("scene: %s\n", pth_last_error()); return 1; }
    const pt_scene_desc *d = pth_scene_desc(sc);
    std::printf("two-OBJ scene: %llu triangles, %llu meshes\n", (unsigned long long)d->num_tris, (unsigned long long)d->num_meshes);
    const bool ok = d->num_meshes == 2 && d->num_tris == 5 + 3 && d->meshes[1].first_face == 5;
    pth_scene_free(sc);
    if (!ok) return 3;

    int refused = 0, cases = 0;
    const char *bad_objs[] = {
        "v 0 0 0\nv 1 0\nv 0 1 0\nf 1 2 3\n",            // short vertex record
        "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 9\n",          // index past the vertices
        "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 -7\n",         // negative index before the first vertex
        "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2\n",            // face with two corners
        "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3",            // no newline at the end: fine
        "f 1 2 3\n",                                     // faces without vertices
        "",                                              // empty file
        "v 1e999 0 0\nv nan 0 0\nv 0 1 0\nf 1 2 3\n",    // out-of-range and non-finite numbers
        "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 99999999999999999999 2 3\n",
    };
    for (const char *text : bad_objs) {
        put("/tmp/pt_sanitize_bad.obj", text);
        float *v = nullptr; uint64_t n = 0;
        const int rc = pth_obj_load("/tmp/pt_sanitize_bad.obj", &v, &n);
        cases++;
        if (rc != 0) refused++;
        else { std::printf("  accepted OBJ (%llu triangles): %.40s\n", (unsigned long long)n, text); pth_free(v); }
    }
    const std::string bad_scenes[] = {
        head + "OBJ\nmaterial_type PLASTIC\nalbedo 1 1 1\nfile_name " + obj_a + "\n",
        head + "OBJ\nmaterial_type GLASS\nalbedo 1 1\n",
        head + "OBJ\nmaterial_type GLASS\nalbedo 1 1 1\nfile_name /tmp/pt_sanitize_missing.obj\n",
        "CAMERA\nposition 0 1\n",
        "ENVIRONMENT\nfile_name /tmp/pt_sanitize_missing.hdr\n",
        head.substr(0, head.size() / 2),
        "",
    };
    for (const std::string &text : bad_scenes) {
        put("/tmp/pt_sanitize_bad.scene", text);
        pth_scene *s2 = nullptr;
        const int rc = pth_scene_load("/tmp/pt_sanitize_bad.scene", 32, 32, &s2);
        cases++;
        if (rc != 0) refused++;
        else { std::printf("  accepted scene: %.40s\n", text.c_str() + (text.size() > 60 ? text.size() - 60 : 0)); pth_scene_free(s2); }
    }
    put("/tmp/pt_sanitize_bad.hdr", "#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 4 +X 4\n\x02\x02\x00\x04\xff\xff");   // truncated scanline
    float *px = nullptr; uint32_t w = 0, h = 0;
    cases++;
    if (pth_hdr_load("/tmp/pt_sanitize_bad.hdr", &px, &w, &h) != 0) refused++;
    else pth_free(px);
    std::printf("hostile inputs: %d of %d refused, none crashed\n", refused, cases);
    return 0;
}

int main()
{
    int rc = builder();
    if (rc) return rc;
    return readers();
}
