// Driver of tools/device_sanitize.sh: the device library's HOST code (scene upload with its validation and re-layout
// passes, the entry points' argument checks, the render loop's bookkeeping) under AddressSanitizer + UBSan on a machine
// with a GPU.  Only host code is instrumented; the kernels are the ordinary ones.  Every malformed scene is held in
// heap buffers of exactly the stated sizes, so a read past what the descriptor promises is reported.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>
#include "pt_api.h"
#include "pt_host.h"

static void put(const char *path, const std::string &text)
{
    FILE *f = std::fopen(path, "wb");
    std::fwrite(text.data(), 1, text.size(), f);
    std::fclose(f);
}

#define MUST(call) do { int rc_ = (call); if (rc_ != 0) { std::printf("%s -> %d: %s\n", #call, rc_, pt_last_error(ctx)); return 1; } } while (0)

int main()
{
    const char *obj = "/tmp/pt_dsan_knot.obj", *hdr = "/tmp/pt_dsan.hdr";
    if (pth_asset_generate("knot_small.obj", obj) != 0 || pth_asset_generate("studio.hdr", hdr) != 0) {
        std::printf("assets: %s\n", pth_last_error());
        return 1;
    }
    put("/tmp/pt_dsan.scene", std::string("CAMERA\nposition 0 1.2 5\nlook_at 0 0.6 0\nfov 35\naperture 0.02\nfocus_dist 5\n\n"
        "ENVIRONMENT\nfile_name ") + hdr + "\n\nPLANE\nmaterial_type LAMBERTIAN\nalbedo 0.3 0.3 0.3\nposition 0 0 0\nnormal 0 1 0\n\n"
        "SPHERE\nmaterial_type GLASS\nalbedo 1 1 1\nrefraction_index 1.5\nposition 1.5 0.5 0\nradius 0.5\n\n"
        "OBJ\nmaterial_type GLOSSY\nalbedo 0.6 0.6 0.6\nspecular 1 1 1\nglossyness 0.05\nfile_name " + obj + "\n");
    pth_scene *hs = nullptr;
    if (pth_scene_load("/tmp/pt_dsan.scene", 96, 64, &hs) != 0) { std::printf("scene: %s\n", pth_last_error()); return 1; }
    const pt_scene_desc good = *pth_scene_desc(hs);

    pt_ctx *ctx = nullptr;
    if (pt_ctx_create(0, &ctx) != 0) { std::printf("ctx: %s\n", pt_last_error(nullptr)); return 1; }
    MUST(pt_scene_upload(ctx, &good));

    // a render and a batch of closest hits through every entry point that takes host buffers
    const uint32_t W = 96, H = 64;
    std::vector<float> org(W * H * 3), dir(W * H * 3), t(W * H), nrm(W * H * 3), film(W * H * 3);
    std::vector<uint32_t> prim(W * H);
    std::vector<uint16_t> rgb16(W * H * 3);
    MUST(pt_primary_rays(ctx, W, H, 0, 7, org.data(), dir.data()));
    MUST(pt_intersect(ctx, (uint64_t)W * H, org.data(), dir.data(), t.data(), prim.data(), nrm.data()));
    MUST(pt_intersect(ctx, 0, org.data(), dir.data(), t.data(), prim.data(), nullptr));
    MUST(pt_film_bind(ctx, W, H, nullptr));
    pt_stats st{};
    for (uint64_t streams = 1; streams <= 4; streams++) {
        MUST(pt_set_option(ctx, PT_OPT_STREAMS, streams));
        MUST(pt_set_option(ctx, PT_OPT_POOL_SIZE, streams == 2 ? 1024 : (1u << 20)));
        MUST(pt_render(ctx, 0, 5, 5, 3, &st));
    }
    // both render modes explicitly, then the mode chosen from the first sample (after an upload forgets the choice)
    for (uint64_t mode : {1, 2, 0}) {
        MUST(pt_set_option(ctx, PT_OPT_RENDER_MODE, mode));
        if (mode == 0) { MUST(pt_scene_upload(ctx, &good)); MUST(pt_film_bind(ctx, W, H, nullptr)); }
        MUST(pt_render(ctx, 0, 5, 5, 3, &st));
        MUST(pt_render(ctx, 2, 1, 5, 3, &st));
        std::printf("render mode %llu -> ran in mode %d\n", (unsigned long long)mode, pt_render_mode_used(ctx));
    }
    MUST(pt_set_option(ctx, PT_OPT_SLAB_BYTES, 1u << 20));          // several slabs
    MUST(pt_render(ctx, 0, 40, 40, 3, &st));
    MUST(pt_film_download(ctx, film.data()));
    MUST(pt_film_tonemap16(ctx, 1.0f, 2.2f, rgb16.data()));
    std::printf("rendered: %llu paths, %llu rays\n", (unsigned long long)st.paths, (unsigned long long)st.rays);

    // malformed scenes: exact-size copies of the buffers, one defect each; all must be refused
    const size_t nn = good.num_nodes, nt = good.num_tri_float4;
    int refused = 0, cases = 0;
    auto attempt = [&](const char *what, const std::function<void(pt_scene_desc &, std::vector<float> &, std::vector<float> &)> &edit) {
        std::vector<float> nodes(good.bvh_nodes, good.bvh_nodes + nn * 16), tris(good.bvh_tris, good.bvh_tris + nt * 4);
        nodes.shrink_to_fit(); tris.shrink_to_fit();
        pt_scene_desc d = good;
        d.bvh_nodes = nodes.data();
        d.bvh_tris = tris.data();
        edit(d, nodes, tris);
        const int rc = pt_scene_upload(ctx, &d);
        cases++;
        if (rc != 0) refused++;
        else std::printf("  ACCEPTED: %s\n", what);
    };
    auto ints = [](std::vector<float> &v) { return reinterpret_cast<int32_t *>(v.data()); };
    attempt("child out of range", [&](pt_scene_desc &, std::vector<float> &n, std::vector<float> &) { ints(n)[(nn / 2) * 16 + 12] = (int32_t)nn + 5; });
    attempt("leaf past the buffer", [&](pt_scene_desc &, std::vector<float> &n, std::vector<float> &) { ints(n)[(nn - 1) * 16 + 13] = ~(int32_t)(nt + 7); });
    attempt("run without terminator", [&](pt_scene_desc &, std::vector<float> &, std::vector<float> &tr) {
        for (size_t i = 0; i < nt; i++) if (reinterpret_cast<uint32_t *>(tr.data())[i * 4] == 0x80000000u) tr[i * 4] = 1.0f; });
    attempt("cycle", [&](pt_scene_desc &, std::vector<float> &n, std::vector<float> &) { ints(n)[(nn - 1) * 16 + 12] = 1; });
    attempt("node count beyond the buffer's meaning", [&](pt_scene_desc &d, std::vector<float> &, std::vector<float> &) { d.num_tri_float4 = nt / 2; });
    attempt("planes missing", [&](pt_scene_desc &d, std::vector<float> &, std::vector<float> &) { d.planes = nullptr; });
    attempt("unknown mesh material", [&](pt_scene_desc &d, std::vector<float> &, std::vector<float> &) { d.mesh_material_type = 0x40; });
    attempt("environment size without texels", [&](pt_scene_desc &d, std::vector<float> &, std::vector<float> &) { d.env_texels = nullptr; });
    {
        std::vector<uint64_t> marg(good.env_marginal, good.env_marginal + good.env_width);
        marg.shrink_to_fit();
        marg[good.env_width / 2] = good.env_width;
        attempt("marginal entry out of range", [&](pt_scene_desc &d, std::vector<float> &, std::vector<float> &) { d.env_marginal = marg.data(); });
    }
    {
        pt_mesh two[2] = {};
        two[0].material_type = PT_DIFFUSE; two[0].first_face = 5;       // the first mesh must start at face 0
        two[1].material_type = PT_GLASS; two[1].first_face = 3;
        attempt("meshes out of order", [&](pt_scene_desc &d, std::vector<float> &, std::vector<float> &) { d.meshes = two; d.num_meshes = 2; });
    }
    std::printf("malformed scenes: %d of %d refused\n", refused, cases);

    // the scene uploaded before them is still in place and usable
    MUST(pt_intersect(ctx, (uint64_t)W * H, org.data(), dir.data(), t.data(), prim.data(), nullptr));
    // argument checks of the render call
    const int r1 = pt_render(ctx, 3, 4, 5, 0, nullptr), r2 = pt_render(ctx, 0, 1, 0, 0, nullptr);
    std::printf("bad sample ranges refused: %d %d\n", r1 != 0, r2 != 0);
    pt_ctx_destroy(ctx);
    pth_scene_free(hs);
    return refused == cases && r1 != 0 && r2 != 0 ? 0 : 2;
}
