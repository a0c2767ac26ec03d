// Probe: dependent 64 B record fetches at the traversal kernel's cache mix (a share of the fetches from a set that stays in
// L1, a share from one that stays in L2, the rest from a set that only fits the Infinity Cache), with part of the lanes
// idle, issued two ways:
//   mode 0  every lane loads its own record with four 16 B loads
//   mode 1  the four lanes of a quad load each of the quad's records together (one 64 B segment per quad and load), then
//           a 4x4 transpose inside the quad gives every lane its record
//   mode 2  as mode 1 without the transpose: what the load side alone could gain (results meaningless)
// usage: gather_probe2 [active_lanes] [l1_share_percent] [l2_share_percent] [steps] [blocks_per_cu]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

extern __shared__ int lds[];

__device__ __forceinline__ uint32_t mix(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <int CTRL> __device__ __forceinline__ float dpp(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}

__device__ __forceinline__ void quad_transpose(float (&a)[4], uint32_t lane)
{
    const bool odd = lane & 1u, hi = lane & 2u;
    {
        const float s01 = odd ? a[0] : a[1], s23 = odd ? a[2] : a[3];
        const float r01 = dpp<0xB1>(s01), r23 = dpp<0xB1>(s23);
        if (odd) { a[0] = r01; a[2] = r23; } else { a[1] = r01; a[3] = r23; }
    }
    {
        const float s02 = hi ? a[0] : a[2], s13 = hi ? a[1] : a[3];
        const float r02 = dpp<0x4E>(s02), r13 = dpp<0x4E>(s13);
        if (hi) { a[0] = r02; a[1] = r13; } else { a[2] = r02; a[3] = r13; }
    }
}

constexpr uint32_t kHot = 256, kMid = 32768, kCold = 1u << 20;

template <int MODE>
__global__ __launch_bounds__(256) void k_probe(const float4 *__restrict__ recs, int steps, uint32_t *out, uint32_t active,
                                               uint32_t t1, uint32_t t2)
{
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    // which lanes sit out: spread over the quads as a traversal's idle lanes are
    const bool on = (mix(lane * 2654435761u + (threadIdx.x >> 6)) & 63u) < active;
    uint32_t idx = mix(gid) & (kCold - 1);
    uint32_t acc = 0;
    if (lds[threadIdx.x] == 12345) acc = 1;
    for (int s = 0; s < steps; s++) {
        float4 r[4];
        if (MODE == 0) {
            if (on) {
                const float4 *p = recs + (size_t)idx * 4;
                r[0] = p[0]; r[1] = p[1]; r[2] = p[2]; r[3] = p[3];
            } else {
                r[0] = r[1] = r[2] = r[3] = make_float4(0, 0, 0, 0);
            }
        } else {
            const uint32_t j = lane & 3u;
            // every lane of a quad takes part in all four loads; a lane without work still names a (valid) record
            const uint32_t want = idx;
            const uint32_t i0 = __builtin_amdgcn_update_dpp(0, (int)want, 0x00, 0xF, 0xF, true);
            const uint32_t i1 = __builtin_amdgcn_update_dpp(0, (int)want, 0x55, 0xF, 0xF, true);
            const uint32_t i2 = __builtin_amdgcn_update_dpp(0, (int)want, 0xAA, 0xF, 0xF, true);
            const uint32_t i3 = __builtin_amdgcn_update_dpp(0, (int)want, 0xFF, 0xF, 0xF, true);
            r[0] = recs[(size_t)i0 * 4 + j];
            r[1] = recs[(size_t)i1 * 4 + j];
            r[2] = recs[(size_t)i2 * 4 + j];
            r[3] = recs[(size_t)i3 * 4 + j];
            if (MODE == 1) {
            float x[4] = {r[0].x, r[1].x, r[2].x, r[3].x}; quad_transpose(x, lane);
            float y[4] = {r[0].y, r[1].y, r[2].y, r[3].y}; quad_transpose(y, lane);
            float zz[4] = {r[0].z, r[1].z, r[2].z, r[3].z}; quad_transpose(zz, lane);
            float w[4] = {r[0].w, r[1].w, r[2].w, r[3].w}; quad_transpose(w, lane);
#pragma unroll
            for (int c = 0; c < 4; c++) r[c] = make_float4(x[c], y[c], zz[c], w[c]);
            }
        }
        uint32_t h = 0;
#pragma unroll
        for (int c = 0; c < 4; c++)
            h += __float_as_uint(r[c].x) + 3 * __float_as_uint(r[c].y) + 5 * __float_as_uint(r[c].z) + 7 * __float_as_uint(r[c].w);
        acc += h;
        const uint32_t m = mix(h + s + gid);
        const uint32_t sel = m >> 22;                      // 10 bits choose the set
        const uint32_t mask = sel < t1 ? kHot - 1 : (sel < t2 ? kMid - 1 : kCold - 1);
        idx = (m * 2654435761u >> 8) & mask;
    }
    out[gid] = acc;
}

int main(int argc, char **argv)
{
    const uint32_t active = argc > 1 ? atoi(argv[1]) : 41;
    const uint32_t p1 = argc > 2 ? atoi(argv[2]) : 38, p2 = argc > 3 ? atoi(argv[3]) : 41;
    const int steps = argc > 4 ? atoi(argv[4]) : 2000;
    const uint32_t t1 = p1 * 1024 / 100, t2 = (p1 + p2) * 1024 / 100;
    std::vector<uint32_t> h((size_t)kCold * 16);
    uint32_t s = 1;
    for (auto &v : h) { s = s * 1664525u + 1013904223u; v = s; }
    float4 *recs; uint32_t *out;
    CHECK(hipMalloc(&recs, h.size() * 4));
    CHECK(hipMemcpy(recs, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    const int per_cu = argc > 5 ? atoi(argv[5]) : 6;
    const int blocks = 256 * per_cu;
    const int lds_bytes = (160 * 1024 / per_cu) & ~1023;
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    std::vector<uint32_t> res0((size_t)blocks * 256), res1(res0.size());
    printf("active lanes %u of 64, %d blocks per CU, L1 set %u %%, L2 set %u %%, rest from a 64 MB set\n", active, per_cu, p1, p2);
    // the quad modes need every lane of a quad: they are run with all 64 lanes only
    const int modes = active == 64 ? 3 : 1;
    for (int mode = 0; mode < modes; mode++) {
        for (int rep = 0; rep < 2; rep++) {
            CHECK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(k_probe<0>, dim3(blocks), dim3(256), lds_bytes, 0, recs, steps, out, active, t1, t2);
            else if (mode == 1) hipLaunchKernelGGL(k_probe<1>, dim3(blocks), dim3(256), lds_bytes, 0, recs, steps, out, active, t1, t2);
            else hipLaunchKernelGGL(k_probe<2>, dim3(blocks), dim3(256), lds_bytes, 0, recs, steps, out, active, t1, t2);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double fetches = (double)blocks * 256 * steps * active / 64.0;
            if (rep) printf("mode %d: %8.3f ms  %7.2f G records/s\n", mode, ms, fetches / ms * 1e-6);
        }
        if (mode < 2) CHECK(hipMemcpy(mode == 0 ? res0.data() : res1.data(), out, res0.size() * 4, hipMemcpyDeviceToHost));
    }
    if (modes == 1) return 0;
    size_t bad = 0;
    for (size_t i = 0; i < res0.size(); i++) bad += res0[i] != res1[i];
    printf("mode 1 vs mode 0: %zu of %zu lanes differ\n", bad, res0.size());
    return 0;
}
