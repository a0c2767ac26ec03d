// Host BVH builder: binned SAH over centroid buckets, two-triangle leaves, right-first flattening into the
// 64 B two-box node + Woop float4 stream the device upload expects.  The output is byte-identical to what the
// reference's builder produces (src/bvh.cu:100-294); the work is organised differently:
//   * triangle boxes and centroids are computed once, and the build permutes an index array instead of the triangles;
//   * subtrees above a size threshold are built as independent tasks on other threads and spliced (the preorder
//     layout stores subtree sizes, so splicing is concatenation).
// No fused multiply-adds anywhere in this file (the Makefile passes -ffp-contract=off): bucket choice depends on
// float rounding.
#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <future>
#include <vector>

#include "host_internal.h"

namespace pth {

namespace {

constexpr float kEps = 0.001f;      // reference src/common.h:17
constexpr int kBuckets = 12;        // reference src/bvh.cu:13

struct Bounds {
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX};       // reference src/aabb.cpp:8
    float hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    void grow(const Bounds &b)
    {
        for (int a = 0; a < 3; a++) {
            lo[a] = lo[a] < b.lo[a] ? lo[a] : b.lo[a];
            hi[a] = hi[a] > b.hi[a] ? hi[a] : b.hi[a];
        }
    }
    void grow(const float p[3])
    {
        for (int a = 0; a < 3; a++) {
            lo[a] = lo[a] < p[a] ? lo[a] : p[a];
            hi[a] = hi[a] > p[a] ? hi[a] : p[a];
        }
    }
    float area() const                              // reference src/aabb.cpp:21-25
    {
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return 2.0f * (dx * dy + dx * dz + dy * dz);
    }
    int longest_axis() const                        // reference src/aabb.cpp:32-42
    {
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (dx >= dy && dx >= dz) return 0;
        if (dy >= dx && dy >= dz) return 1;
        return 2;
    }
};

struct TreeNode {
    Bounds box;
    uint32_t first = 0;        // leaf: position in the order array
    uint32_t count = 0;        // leaf: 1 or 2 triangles; interior: 0
    uint32_t subtree = 1;      // nodes in this subtree including this one; right child = self + 1 + left.subtree
};

struct Builder {
    const float *verts;
    std::vector<Bounds> tri_box;
    std::vector<float> centroid;        // 3 per triangle
    std::vector<uint32_t> order;
    int spawn_budget;
    std::atomic<int> live_tasks{0};

    explicit Builder(const float *v, uint64_t n, int threads) : verts(v), tri_box(n), centroid(3 * n), order(n)
    {
        spawn_budget = threads > 1 ? threads - 1 : 0;
        for (uint64_t i = 0; i < n; i++) {
            const float *p = v + 9 * i;
            Bounds b;
            for (int a = 0; a < 3; a++) {
                // reference src/bvh.cuh:39-49: min/max of the three vertices, flat boxes padded by EPSILON
                float lo = p[a] < p[3 + a] ? p[a] : p[3 + a];
                lo = lo < p[6 + a] ? lo : p[6 + a];
                float hi = p[a] > p[3 + a] ? p[a] : p[3 + a];
                hi = hi > p[6 + a] ? hi : p[6 + a];
                if (hi - lo == 0) hi += kEps;
                b.lo[a] = lo;
                b.hi[a] = hi;
                centroid[3 * i + a] = (lo + hi) / 2;        // reference src/aabb.cpp:27-30
            }
            tri_box[i] = b;
            order[i] = (uint32_t)i;
        }
    }

    int bucket(uint32_t tri, const Bounds &cb, int axis) const      // reference src/bvh.cu:87-90,142-143
    {
        const float f = kBuckets * (centroid[3 * (size_t)tri + axis] - cb.lo[axis]) / (cb.hi[axis] - cb.lo[axis]);
        int b = (int)(size_t)f;
        if (b == kBuckets) b--;
        return b;
    }

    void build(uint32_t start, uint32_t end, std::vector<TreeNode> &out, uint32_t depth, uint32_t &max_depth)
    {
        const size_t self = out.size();
        out.emplace_back();
        if (depth > max_depth) max_depth = depth;
        Bounds all, cents;
        for (uint32_t i = start; i < end; i++) {
            all.grow(tri_box[order[i]]);
            cents.grow(&centroid[3 * (size_t)order[i]]);
        }
        const int axis = cents.longest_axis();
        const uint32_t n = end - start;
        if (n == 1) {
            out[self].first = start;
            out[self].count = 1;
            out[self].box = tri_box[order[start]];
            return;
        }
        if (n == 2) {
            // reference src/bvh.cu:123-131: the pair is ordered along the axis
            const float ca = centroid[3 * (size_t)order[start] + axis], cb = centroid[3 * (size_t)order[start + 1] + axis];
            if (!(ca < cb)) std::swap(order[start], order[start + 1]);
            out[self].first = start;
            out[self].count = 2;
            Bounds b = tri_box[order[start]];
            b.grow(tri_box[order[start + 1]]);
            out[self].box = b;
            return;
        }

        uint32_t mid;
        if (!(cents.hi[axis] - cents.lo[axis] > 0.0f)) {
            mid = start + n / 2;        // coincident centroids: the reference divides 0/0 and never returns
        } else {
            Bounds bbox[kBuckets];
            size_t bcount[kBuckets] = {};
            for (uint32_t i = start; i < end; i++) {
                const int b = bucket(order[i], cents, axis);
                bcount[b]++;
                bbox[b].grow(tri_box[order[i]]);
            }
            // reference src/bvh.cu:150-173: cost of each of the 11 splits, first minimum wins
            float best = 0;
            size_t best_split = 0;
            for (int i = 0; i < kBuckets - 1; i++) {
                Bounds b0, b1;
                size_t n0 = 0, n1 = 0;
                for (int j = 0; j <= i; j++) { b0.grow(bbox[j]); n0 += bcount[j]; }
                for (int j = i + 1; j < kBuckets; j++) { b1.grow(bbox[j]); n1 += bcount[j]; }
                const float cost = (n0 * b0.area() + n1 * b1.area()) / all.area() + 1;
                if (i == 0 || cost < best) { best = cost; best_split = (size_t)i; }
            }
            // in-place two-pointer partition, element order as std::partition leaves it (reference src/bvh.cu:176-184)
            uint32_t lo = start, hi = end;
            auto left_side = [&](uint32_t t) { return (size_t)bucket(t, cents, axis) <= best_split; };
            for (;;) {
                while (lo != hi && left_side(order[lo])) lo++;
                if (lo == hi) break;
                hi--;
                while (lo != hi && !left_side(order[hi])) hi--;
                if (lo == hi) break;
                std::swap(order[lo], order[hi]);
                lo++;
            }
            mid = lo;
            if (mid == start || mid == end) mid = start + n / 2;
        }

        // children: the left subtree may run on another thread, results are concatenated in preorder
        constexpr uint32_t kTaskMin = 1u << 15;
        bool spawned = false;
        std::future<std::pair<std::vector<TreeNode>, uint32_t>> fut;
        if (spawn_budget > 0 && (mid - start) >= kTaskMin && (end - mid) >= kTaskMin) {
            if (live_tasks.fetch_add(1) < spawn_budget) {
                spawned = true;
                fut = std::async(std::launch::async, [this, start, mid, depth]() {
                    std::vector<TreeNode> sub;
                    uint32_t md = 0;
                    build(start, mid, sub, depth + 1, md);
                    live_tasks.fetch_sub(1);
                    return std::make_pair(std::move(sub), md);
                });
            } else {
                live_tasks.fetch_sub(1);
            }
        }
        if (spawned) {
            std::vector<TreeNode> right;
            uint32_t mdr = 0;
            build(mid, end, right, depth + 1, mdr);
            auto left = fut.get();
            max_depth = std::max(max_depth, std::max(mdr, left.second));
            out.insert(out.end(), left.first.begin(), left.first.end());
            out.insert(out.end(), right.begin(), right.end());
        } else {
            build(start, mid, out, depth + 1, max_depth);
            build(mid, end, out, depth + 1, max_depth);
        }
        const size_t l = self + 1, r = l + out[l].subtree;
        out[self].subtree = 1 + out[l].subtree + out[r].subtree;
        Bounds b = out[l].box;
        b.grow(out[r].box);
        out[self].box = b;
    }
};

// Rows 0..2 of the inverse of the affine matrix whose columns are (e0,0), (e1,0), (e2,0), (p,1), by the general 4x4
// cofactor expansion evaluated term by term as in reference include/linear_math.cuh:85-121, so that the rounding
// (and the signs of zeros) match.  m is column major.
bool affine_inverse_rows(const float m[16], float rows[12])
{
    auto P = [&](int a, int b, int c) { return m[a] * m[b] * m[c]; };
    float c[16];
    c[0]  =  P(5,10,15) - P(5,11,14) - P(9,6,15) + P(9,7,14) + P(13,6,11) - P(13,7,10);
    c[4]  = -m[4] * m[10] * m[15] + P(4,11,14) + P(8,6,15) - P(8,7,14) - P(12,6,11) + P(12,7,10);
    c[8]  =  P(4,9,15) - P(4,11,13) - P(8,5,15) + P(8,7,13) + P(12,5,11) - P(12,7,9);
    c[12] = -m[4] * m[9] * m[14] + P(4,10,13) + P(8,5,14) - P(8,6,13) - P(12,5,10) + P(12,6,9);
    c[1]  = -m[1] * m[10] * m[15] + P(1,11,14) + P(9,2,15) - P(9,3,14) - P(13,2,11) + P(13,3,10);
    c[5]  =  P(0,10,15) - P(0,11,14) - P(8,2,15) + P(8,3,14) + P(12,2,11) - P(12,3,10);
    c[9]  = -m[0] * m[9] * m[15] + P(0,11,13) + P(8,1,15) - P(8,3,13) - P(12,1,11) + P(12,3,9);
    c[13] =  P(0,9,14) - P(0,10,13) - P(8,1,14) + P(8,2,13) + P(12,1,10) - P(12,2,9);
    c[2]  =  P(1,6,15) - P(1,7,14) - P(5,2,15) + P(5,3,14) + P(13,2,7) - P(13,3,6);
    c[6]  = -m[0] * m[6] * m[15] + P(0,7,14) + P(4,2,15) - P(4,3,14) - P(12,2,7) + P(12,3,6);
    c[10] =  P(0,5,15) - P(0,7,13) - P(4,1,15) + P(4,3,13) + P(12,1,7) - P(12,3,5);
    c[14] = -m[0] * m[5] * m[14] + P(0,6,13) + P(4,1,14) - P(4,2,13) - P(12,1,6) + P(12,2,5);
    float det = m[0] * c[0] + m[1] * c[4] + m[2] * c[8] + m[3] * c[12];
    if (det == 0) return false;
    det = 1.f / det;
    for (int r = 0; r < 3; r++)
        for (int k = 0; k < 4; k++) rows[r * 4 + k] = c[r + 4 * k] * det;
    return true;
}

// reference src/bvh.cu:214-227
void woop_rows(const float *v, float out[12])
{
    const float e0[3] = {v[0] - v[6], v[1] - v[7], v[2] - v[8]};
    const float e1[3] = {v[3] - v[6], v[4] - v[7], v[5] - v[8]};
    const float nx = e0[1] * e1[2] - e0[2] * e1[1], ny = e0[2] * e1[0] - e0[0] * e1[2], nz = e0[0] * e1[1] - e0[1] * e1[0];
    const float m[16] = {e0[0], e0[1], e0[2], 0.0f, e1[0], e1[1], e1[2], 0.0f, nx, ny, nz, 0.0f, v[6], v[7], v[8], 1.0f};
    float r[12];
    if (!affine_inverse_rows(m, r)) {
        const float id[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};       // singular: linear_math.cuh:112-113
        std::memcpy(r, id, sizeof r);
    }
    out[0] = r[8]; out[1] = r[9]; out[2] = r[10]; out[3] = -r[11];
    std::memcpy(out + 4, r, 8 * sizeof(float));
}

inline float bits_to_float(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

}   // namespace

int bvh_build(const float *verts, uint64_t ntris, int threads, pth_bvh *out)
{
    std::memset(out, 0, sizeof *out);
    if (!verts) return set_error("pth_bvh_build: verts is NULL");
    if (ntris < 3) return set_error("pth_bvh_build: at least 3 triangles are needed (the root must be an interior node)");
    if (ntris >= (1ull << 29)) return set_error("pth_bvh_build: too many triangles");

    Builder b(verts, ntris, threads);
    std::vector<TreeNode> tree;
    tree.reserve(2 * ntris);
    uint32_t max_depth = 0;
    b.build(0, (uint32_t)ntris, tree, 0, max_depth);

    // flatten: depth-first with the right child visited first, parents patched when a child is emitted
    // (reference src/bvh.cu:234-294)
    std::vector<float> gnodes;
    gnodes.reserve(16 * ntris);
    std::vector<float> gtris;
    gtris.reserve(16 * ntris + 16);
    std::vector<uint32_t> gids;
    gids.reserve(4 * ntris + 4);
    gtris.assign(4, 0.0f);
    gids.push_back(PT_PRIM_MISS);
    uint64_t leaves = 0;

    auto emit_leaf = [&](const TreeNode &leaf) -> uint32_t {
        const uint32_t ref = ~(uint32_t)(gtris.size() / 4);
        for (uint32_t k = 0; k < leaf.count; k++) {
            const uint32_t tri = b.order[leaf.first + k];
            float w[12];
            woop_rows(verts + 9 * (size_t)tri, w);
            gtris.insert(gtris.end(), w, w + 12);
            gids.push_back(tri); gids.push_back(PT_PRIM_MISS); gids.push_back(PT_PRIM_MISS);
        }
        const float term = bits_to_float(0x80000000u);
        gtris.insert(gtris.end(), {term, term, term, term});
        gids.push_back(PT_PRIM_MISS);
        leaves++;
        return ref;
    };

    struct Todo { uint32_t node; uint32_t parent; bool left; };
    std::vector<Todo> todo;
    todo.push_back({0, 0, false});
    while (!todo.empty()) {
        const Todo e = todo.back();
        todo.pop_back();
        const uint32_t li = e.node + 1, ri = li + tree[li].subtree;
        const TreeNode &l = tree[li], &r = tree[ri];
        const uint32_t me = (uint32_t)(gnodes.size() / 16);
        float g[16] = {l.box.lo[0], l.box.hi[0], l.box.lo[1], l.box.hi[1], r.box.lo[0], r.box.hi[0], r.box.lo[1],
                       r.box.hi[1], l.box.lo[2], l.box.hi[2], r.box.lo[2], r.box.hi[2], 0, 0, 0, 0};
        if (e.node != 0) gnodes[(size_t)e.parent * 16 + (e.left ? 12 : 13)] = bits_to_float(me);
        if (l.count == 0) todo.push_back({li, me, true});
        else g[12] = bits_to_float(emit_leaf(l));
        if (r.count == 0) todo.push_back({ri, me, false});
        else g[13] = bits_to_float(emit_leaf(r));
        gnodes.insert(gnodes.end(), g, g + 16);
    }

    out->num_nodes = gnodes.size() / 16;
    out->num_tri_float4 = gtris.size() / 4;
    out->num_tris = ntris;
    out->num_leaves = leaves;
    out->max_depth = max_depth;
    out->nodes = (float *)std::malloc(gnodes.size() * sizeof(float));
    // two float4 of zero padding: a reader in the reference's style fetches three float4 before it looks for the
    // terminator (src/bvh.cuh:174-181)
    out->tris = (float *)std::calloc(gtris.size() + 8, sizeof(float));
    out->prim_ids = (uint32_t *)std::malloc(gids.size() * sizeof(uint32_t));
    if (!out->nodes || !out->tris || !out->prim_ids) {
        std::free(out->nodes); std::free(out->tris); std::free(out->prim_ids);
        std::memset(out, 0, sizeof *out);
        return set_error("pth_bvh_build: out of memory");
    }
    std::memcpy(out->nodes, gnodes.data(), gnodes.size() * sizeof(float));
    std::memcpy(out->tris, gtris.data(), gtris.size() * sizeof(float));
    std::memcpy(out->prim_ids, gids.data(), gids.size() * sizeof(uint32_t));
    return 0;
}

}   // namespace pth
