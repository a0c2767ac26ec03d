// Probe: throughput of dependent random 64 B record fetches on one GPU, three ways of issuing them.
//   mode 0  every lane loads its own record with four 16 B loads (64 distinct lines per load instruction)
//   mode 1  the four lanes of a quad load one record together, 16 B each (one 64 B segment per quad and load), four loads
//           cover the quad's four records, then a 4x4 transpose inside the quad gives every lane its record
//   mode 2  as mode 1 without the transpose (upper bound of what the load side can gain; results meaningless)
//   mode 3  structure of arrays: the record's four 16 B parts live in four arrays (part c of record i at c * records + i),
//           so a lane's four loads go to four different cache lines
// Every lane walks a pseudo-random chain: next index = hash(sum of the record's words) so loads depend on each other.
// usage: gather_probe [records] [steps] [lds_bytes_per_block] [active_lanes] [placement]
// active_lanes < 64 (mode 0 only): the other lanes of every wave sit out (exec-masked), as lanes without a ray do in the
// traversal kernel; the rate then tells whether the fetch path is priced per instruction or per active lane.
// placement: 0 (default) the first active_lanes lanes of the wave fetch; 1 a pseudo-random subset of about that size, spread
// over the whole wave; 2 the first active_lanes / 2 lanes of each half-wave; 3 every lane whose index modulo 64 / active_lanes
// is 0 (a regular stride).
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

__device__ __forceinline__ float qperm(float v, int src_lane_in_quad)
{
    // value of v in lane (lane & ~3) + src_lane_in_quad
    return __shfl(v, (threadIdx.x & ~3) | src_lane_in_quad, 64);
}

template <int CTRL> __device__ __forceinline__ float dpp(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}

// 4x4 transpose inside every quad: before, lane j holds a[q] = word of record q (chunk j); after, lane j holds chunk c of
// record j in a[c].  Two butterfly stages, each one DPP move + select per output word.
__device__ __forceinline__ void quad_transpose(float (&a)[4], uint32_t lane)
{
    const bool odd = lane & 1u, hi = lane & 2u;
    // stage 1: partner lane ^ 1, rows (0,1) and (2,3); quad_perm [1,0,3,2] = 0xB1
    {
        const float s01 = odd ? a[0] : a[1], s23 = odd ? a[2] : a[3];
        const float r01 = dpp<0xB1>(s01), r23 = dpp<0xB1>(s23);
        if (odd) { a[0] = r01; a[2] = r23; } else { a[1] = r01; a[3] = r23; }
    }
    // stage 2: partner lane ^ 2, rows (0,2) and (1,3); quad_perm [2,3,0,1] = 0x4E
    {
        const float s02 = hi ? a[0] : a[2], s13 = hi ? a[1] : a[3];
        const float r02 = dpp<0x4E>(s02), r13 = dpp<0x4E>(s13);
        if (hi) { a[0] = r02; a[1] = r13; } else { a[2] = r02; a[3] = r13; }
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void k_probe(const float4 *__restrict__ recs, uint32_t nrec, int steps, uint32_t *out,
                                               uint32_t active, int placement = 0)
{
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t idx = mix(gid) % nrec;
    uint32_t acc = 0;
    if (lds[threadIdx.x] == 12345) acc = 1;      // keep the LDS allocation
    bool on = lane < active;
    if (placement == 1) on = (mix(lane * 2654435761u + (threadIdx.x >> 6)) & 63u) < active;
    if (placement == 2) on = (lane & 31u) < active / 2;
    if (placement == 3) on = active >= 64 || (lane % (64u / active)) == 0;
    if (MODE == 0 && !on) { out[gid] = 0; return; }
    for (int s = 0; s < steps; s++) {
        float4 r[4];
        if (MODE == 0) {
            const float4 *p = recs + (size_t)idx * 4;
            r[0] = p[0]; r[1] = p[1]; r[2] = p[2]; r[3] = p[3];
        } else if (MODE == 3) {
            r[0] = recs[idx]; r[1] = recs[(size_t)nrec + idx]; r[2] = recs[2 * (size_t)nrec + idx]; r[3] = recs[3 * (size_t)nrec + idx];
        } else {
            const uint32_t j = lane & 3u;
            // quad_perm [q,q,q,q]: the index held by lane q of the quad
            const uint32_t i0 = __builtin_amdgcn_update_dpp(0, (int)idx, 0x00, 0xF, 0xF, true);
            const uint32_t i1 = __builtin_amdgcn_update_dpp(0, (int)idx, 0x55, 0xF, 0xF, true);
            const uint32_t i2 = __builtin_amdgcn_update_dpp(0, (int)idx, 0xAA, 0xF, 0xF, true);
            const uint32_t i3 = __builtin_amdgcn_update_dpp(0, (int)idx, 0xFF, 0xF, 0xF, true);
            r[0] = recs[(size_t)i0 * 4 + j];
            r[1] = recs[(size_t)i1 * 4 + j];
            r[2] = recs[(size_t)i2 * 4 + j];
            r[3] = recs[(size_t)i3 * 4 + j];
            if (MODE == 1) {
                float x[4] = {r[0].x, r[1].x, r[2].x, r[3].x}; quad_transpose(x, lane);
                float y[4] = {r[0].y, r[1].y, r[2].y, r[3].y}; quad_transpose(y, lane);
                float z[4] = {r[0].z, r[1].z, r[2].z, r[3].z}; quad_transpose(z, lane);
                float w[4] = {r[0].w, r[1].w, r[2].w, r[3].w}; quad_transpose(w, lane);
#pragma unroll
                for (int c = 0; c < 4; c++) r[c] = make_float4(x[c], y[c], z[c], w[c]);
            }
        }
        uint32_t h = 0;
#pragma unroll
        for (int c = 0; c < 4; c++)
            h += __float_as_uint(r[c].x) + 3 * __float_as_uint(r[c].y) + 5 * __float_as_uint(r[c].z) + 7 * __float_as_uint(r[c].w);
        acc += h;
        idx = mix(h + s) % nrec;
    }
    out[gid] = acc;
}

int main(int argc, char **argv)
{
    const uint32_t nrec = argc > 1 ? atoi(argv[1]) : 1300000;
    const int steps = argc > 2 ? atoi(argv[2]) : 2000;
    const int lds_bytes = argc > 3 ? atoi(argv[3]) : 26 * 1024;
    const uint32_t active = argc > 4 ? atoi(argv[4]) : 64;
    const int placement = argc > 5 ? atoi(argv[5]) : 0;
    std::vector<uint32_t> h((size_t)nrec * 16);
    uint32_t s = 1;
    for (auto &v : h) { s = s * 1664525u + 1013904223u; v = s; }
    float4 *recs; uint32_t *out;
    CHECK(hipMalloc(&recs, h.size() * 4));
    CHECK(hipMemcpy(recs, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    const int blocks = 256 * 6;
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    std::vector<uint32_t> res0((size_t)blocks * 256), res1(res0.size());
    for (int mode = 0; mode < (active < 64 ? 1 : 4); mode++) {
        for (int rep = 0; rep < 2; rep++) {
            CHECK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(k_probe<0>, dim3(blocks), dim3(256), lds_bytes, 0, recs, nrec, steps, out, active, placement);
            if (mode == 1) hipLaunchKernelGGL(k_probe<1>, dim3(blocks), dim3(256), lds_bytes, 0, recs, nrec, steps, out, active);
            if (mode == 2) hipLaunchKernelGGL(k_probe<2>, dim3(blocks), dim3(256), lds_bytes, 0, recs, nrec, steps, out, active);
            if (mode == 3) hipLaunchKernelGGL(k_probe<3>, dim3(blocks), dim3(256), lds_bytes, 0, recs, nrec, steps, out, active);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double fetches = (double)blocks * 256 * steps * (mode == 0 ? active / 64.0 : 1.0);
            if (rep) printf("mode %d: %8.3f ms  %7.2f G records/s  %7.1f GB/s\n", mode, ms, fetches / ms * 1e-6, fetches * 64 / ms * 1e-6);
        }
        if (mode == 0) CHECK(hipMemcpy(res0.data(), out, res0.size() * 4, hipMemcpyDeviceToHost));
        if (mode == 1) {
            CHECK(hipMemcpy(res1.data(), out, res1.size() * 4, hipMemcpyDeviceToHost));
            size_t bad = 0;
            for (size_t i = 0; i < res0.size(); i++) bad += res0[i] != res1[i];
            printf("mode 1 vs mode 0: %zu of %zu lanes differ\n", bad, res0.size());
        }
    }
    return 0;
}
