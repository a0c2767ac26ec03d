#include <cstdio>
#include <cstring>
#include <vector>
#include "pt_host.h"
int main()
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
