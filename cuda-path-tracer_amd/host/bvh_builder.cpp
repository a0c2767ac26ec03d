// Host BVH builder: binned SAH over centroid buckets, two-triangle leaves, right-first flattening into the
// 64 B two-box node + Woop float4 stream the device upload expects.  The output is byte-identical to what the
// reference's builder produces (src/bvh.cu:100-294) and does not depend on the thread count; the work is organised
// differently:
//   * triangle boxes are computed once and permuted together with an index array, so that every pass over a range
//     streams through memory; centroids are recomputed from the boxes (the same (lo + hi) / 2 every time);
//   * the top of the tree (ranges of kParMin triangles and more) is built node by node with every pass over the range
//     -- bounds, bucket histogram, partition -- split over all threads.  The partition reproduces the element order of
//     the sequential two-pointer partition: that algorithm swaps the k-th misplaced element from the left with the
//     k-th misplaced element from the right, which can be done for all k at once;
//   * the ranges below are independent subtrees, built one per task;
//   * the position of every node and leaf in the flattened buffers follows from subtree sizes alone, so the same
//     subtrees are flattened (and their triangles Woop-transformed) in parallel, straight into the output buffers.
// Reductions keep the sequential result bit for bit: min/max folds pick the later of two equal values (+0/-0), and
// chunk results are merged in chunk order.
// No fused multiply-adds anywhere in this file (the Makefile passes -ffp-contract=off): bucket choice depends on
// float rounding.
#include <algorithm>
#include <atomic>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "host_internal.h"

namespace pth {

namespace {

constexpr float kEps = 0.001f;      // reference src/common.h:17
constexpr int kBuckets = 12;        // reference src/bvh.cu:13
constexpr uint32_t kParMin = 1u << 16;      // ranges at least this long are split with all threads
constexpr uint32_t kChunkMin = 1u << 13;    // smallest piece of a range handed to one thread

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

// Uninitialised storage for plain data (a std::vector would fill gigabytes on one thread first).
template <class T>
struct Array {
    T *p;
    explicit Array(uint64_t n) : p((T *)std::malloc(std::max<uint64_t>(n, 1) * sizeof(T))) {}
    ~Array() { std::free(p); }
    Array(const Array &) = delete;
    Array &operator=(const Array &) = delete;
    T &operator[](uint64_t i) { return p[i]; }
    const T &operator[](uint64_t i) const { return p[i]; }
};

// Workers that run fn(0..n-1) together with the calling thread.  One job at a time.
class Pool {
public:
    explicit Pool(int threads)
    {
        for (int i = 1; i < threads; i++) workers_.emplace_back([this] { work(); });
    }
    ~Pool()
    {
        {
            std::lock_guard<std::mutex> g(m_);
            stop_ = true;
        }
        cv_job_.notify_all();
        for (auto &t : workers_) t.join();
    }
    int threads() const { return (int)workers_.size() + 1; }
    void run(int n, const std::function<void(int)> &fn)
    {
        if (n <= 0) return;
        if (workers_.empty() || n == 1) {
            for (int i = 0; i < n; i++) fn(i);
            return;
        }
        {
            std::lock_guard<std::mutex> g(m_);
            fn_ = &fn;
            n_ = n;
            next_.store(0);
            busy_ = (int)workers_.size();
            gen_++;
        }
        cv_job_.notify_all();
        drain();
        std::unique_lock<std::mutex> g(m_);
        cv_done_.wait(g, [this] { return busy_ == 0; });
    }

private:
    void drain()
    {
        for (int i; (i = next_.fetch_add(1)) < n_;) (*fn_)(i);
    }
    void work()
    {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> g(m_);
                cv_job_.wait(g, [&] { return stop_ || gen_ != seen; });
                if (stop_) return;
                seen = gen_;
            }
            drain();
            {
                std::lock_guard<std::mutex> g(m_);
                busy_--;
            }
            cv_done_.notify_one();
        }
    }
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_job_, cv_done_;
    const std::function<void(int)> *fn_ = nullptr;
    int n_ = 0, busy_ = 0;
    std::atomic<int> next_{0};
    uint64_t gen_ = 0;
    bool stop_ = false;
};

// A subtree in preorder: the left child of node i is i + 1, the right child i + 1 + [i + 1].subtree.
struct TreeNode {
    Bounds box;
    uint32_t first = 0;        // leaf: position in the order array
    uint32_t count = 0;        // leaf: 1 or 2 triangles; interior: 0
    uint32_t subtree = 1;      // nodes in this subtree including this one
    uint32_t ntris = 0;        // triangles below this node
};

// interior nodes and float4 entries (3 per triangle + 1 terminator per leaf) a subtree takes in the flattened buffers
inline uint64_t interior_of(uint32_t subtree) { return (subtree - 1) / 2; }
inline uint64_t float4s_of(uint32_t subtree, uint64_t ntris) { return 3 * ntris + (subtree + 1) / 2; }

// A node of the tree top: either an interior node with two TopNode children or a whole subtree built by one task.
struct TopNode {
    int left = -1, right = -1;
    int task = -1;
    Bounds box;
    uint32_t subtree = 0;
    uint32_t ntris = 0;
    uint32_t leaf_count = 0;   // > 0: the subtree is a single leaf
};

struct Task {
    uint32_t start, end, depth;
    std::vector<TreeNode> nodes;
    uint32_t max_depth = 0;
    uint64_t node_off = 0, tri_off = 0;     // where its root lands in the flattened buffers
};

struct Builder {
    const float *verts;
    Pool &pool;
    Array<Bounds> pbox;                 // box of the triangle at each position of the order array
    Array<uint32_t> order;
    std::vector<uint8_t> side;          // per position of the range being split: its bucket
    std::vector<uint32_t> bad_l, bad_r; // positions of misplaced elements of the range being split
    std::vector<TopNode> top;
    std::vector<Task> tasks;
    uint32_t top_depth = 0;

    Builder(const float *v, uint64_t n, Pool &p) : verts(v), pool(p), pbox(n), order(n)
    {
        if (!pbox.p || !order.p) return;
        const int chunks = pieces(n);
        pool.run(chunks, [&](int c) {
            const uint64_t b = n * c / chunks, e = n * (c + 1) / chunks;
            for (uint64_t i = b; i < e; i++) {
                const float *q = v + 9 * i;
                Bounds bx;
                for (int a = 0; a < 3; a++) {
                    // reference src/bvh.cuh:39-49: min/max of the three vertices, flat boxes padded by EPSILON
                    float lo = q[a] < q[3 + a] ? q[a] : q[3 + a];
                    lo = lo < q[6 + a] ? lo : q[6 + a];
                    float hi = q[a] > q[3 + a] ? q[a] : q[3 + a];
                    hi = hi > q[6 + a] ? hi : q[6 + a];
                    if (hi - lo == 0) hi += kEps;
                    bx.lo[a] = lo;
                    bx.hi[a] = hi;
                }
                pbox[i] = bx;
                order[i] = (uint32_t)i;
            }
        });
    }

    int pieces(uint64_t n) const
    {
        const uint64_t most = (uint64_t)pool.threads() * 4;
        return (int)std::max<uint64_t>(1, std::min(most, n / kChunkMin));
    }

    static float centre(const Bounds &b, int axis) { return (b.lo[axis] + b.hi[axis]) / 2; }     // reference src/aabb.cpp:27-30
    static void grow_centre(Bounds &c, const Bounds &b)
    {
        const float p[3] = {centre(b, 0), centre(b, 1), centre(b, 2)};
        c.grow(p);
    }
    void swap_positions(uint32_t i, uint32_t j)
    {
        std::swap(order[i], order[j]);
        std::swap(pbox[i], pbox[j]);
    }

    int bucket(uint32_t pos, const Bounds &cb, int axis) const      // reference src/bvh.cu:87-90,142-143
    {
        const float f = kBuckets * (centre(pbox[pos], axis) - cb.lo[axis]) / (cb.hi[axis] - cb.lo[axis]);
        int b = (int)(size_t)f;
        if (b == kBuckets) b--;
        return b;
    }

    // reference src/bvh.cu:150-173: cost of each of the 11 splits, first minimum wins.  The reference folds the boxes
    // of both sides anew for every split; here the right sides come from one sweep from the top.  That fold runs in
    // the opposite order, which can only change the sign of a zero bound, and the areas cannot see that.
    static size_t best_split_of(const Bounds *bbox, const size_t *bcount, const Bounds &all)
    {
        float right_area[kBuckets];
        size_t right_n[kBuckets];
        Bounds b1;
        size_t n1 = 0;
        for (int i = kBuckets - 1; i > 0; i--) {
            b1.grow(bbox[i]);
            n1 += bcount[i];
            right_area[i - 1] = b1.area();
            right_n[i - 1] = n1;
        }
        const float whole = all.area();
        float best = 0;
        size_t best_split = 0;
        Bounds b0;
        size_t n0 = 0;
        for (int i = 0; i < kBuckets - 1; i++) {
            b0.grow(bbox[i]);
            n0 += bcount[i];
            const float cost = (n0 * b0.area() + right_n[i] * right_area[i]) / whole + 1;
            if (i == 0 || cost < best) { best = cost; best_split = (size_t)i; }
        }
        return best_split;
    }

    // ---- one subtree on one thread ---------------------------------------------------------------------------------
    void build(uint32_t start, uint32_t end, std::vector<TreeNode> &out, uint32_t depth, uint32_t &max_depth)
    {
        const size_t self = out.size();
        out.emplace_back();
        if (depth > max_depth) max_depth = depth;
        Bounds all, cents;
        for (uint32_t i = start; i < end; i++) {
            all.grow(pbox[i]);
            grow_centre(cents, pbox[i]);
        }
        const int axis = cents.longest_axis();
        const uint32_t n = end - start;
        out[self].ntris = n;
        if (n == 1) {
            out[self].first = start;
            out[self].count = 1;
            out[self].box = pbox[start];
            return;
        }
        if (n == 2) {
            // reference src/bvh.cu:123-131: the pair is ordered along the axis
            const float ca = centre(pbox[start], axis), cb = centre(pbox[start + 1], axis);
            if (!(ca < cb)) swap_positions(start, start + 1);
            out[self].first = start;
            out[self].count = 2;
            Bounds b = pbox[start];
            b.grow(pbox[start + 1]);
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
                const int b = bucket(i, cents, axis);
                bcount[b]++;
                bbox[b].grow(pbox[i]);
            }
            const size_t best_split = best_split_of(bbox, bcount, all);
            // in-place two-pointer partition, element order as std::partition leaves it (reference src/bvh.cu:176-184)
            uint32_t lo = start, hi = end;
            auto left_side = [&](uint32_t pos) { return (size_t)bucket(pos, cents, axis) <= best_split; };
            for (;;) {
                while (lo != hi && left_side(lo)) lo++;
                if (lo == hi) break;
                hi--;
                while (lo != hi && !left_side(hi)) hi--;
                if (lo == hi) break;
                swap_positions(lo, hi);
                lo++;
            }
            mid = lo;
            if (mid == start || mid == end) mid = start + n / 2;
        }

        build(start, mid, out, depth + 1, max_depth);
        build(mid, end, out, depth + 1, max_depth);
        const size_t l = self + 1, r = l + out[l].subtree;
        out[self].subtree = 1 + out[l].subtree + out[r].subtree;
        Bounds b = out[l].box;
        b.grow(out[r].box);
        out[self].box = b;
    }

    // ---- the tree top: one node at a time, every pass over the range on all threads -------------------------------
    // Splits [start, end) exactly as build() does and returns the split position.
    uint32_t split_parallel(uint32_t start, uint32_t end)
    {
        const uint32_t n = end - start;
        const int chunks = pieces(n);
        auto lo_of = [&](int c) { return start + (uint32_t)((uint64_t)n * c / chunks); };

        std::vector<Bounds> c_all(chunks), c_cent(chunks);
        pool.run(chunks, [&](int c) {
            Bounds a, ce;
            for (uint32_t i = lo_of(c), e = lo_of(c + 1); i < e; i++) {
                a.grow(pbox[i]);
                grow_centre(ce, pbox[i]);
            }
            c_all[c] = a;
            c_cent[c] = ce;
        });
        Bounds all, cents;
        for (int c = 0; c < chunks; c++) { all.grow(c_all[c]); cents.grow(c_cent[c]); }
        const int axis = cents.longest_axis();
        if (!(cents.hi[axis] - cents.lo[axis] > 0.0f)) return start + n / 2;

        struct Hist { Bounds box[kBuckets]; size_t count[kBuckets] = {}; };
        std::vector<Hist> hist(chunks);
        pool.run(chunks, [&](int c) {
            Hist &h = hist[c];
            for (uint32_t i = lo_of(c), e = lo_of(c + 1); i < e; i++) {
                const int b = bucket(i, cents, axis);
                side[i] = (uint8_t)b;
                h.count[b]++;
                h.box[b].grow(pbox[i]);
            }
        });
        Bounds bbox[kBuckets];
        size_t bcount[kBuckets] = {};
        for (int c = 0; c < chunks; c++)
            for (int b = 0; b < kBuckets; b++) { bbox[b].grow(hist[c].box[b]); bcount[b] += hist[c].count[b]; }
        const size_t best_split = best_split_of(bbox, bcount, all);

        size_t nleft = 0;
        for (size_t b = 0; b <= best_split; b++) nleft += bcount[b];
        const uint32_t mid = start + (uint32_t)nleft;
        if (mid == start || mid == end) return start + n / 2;      // nothing is out of place: no swaps either way

        // The two-pointer partition swaps the k-th element from the left that belongs right with the k-th element
        // from the right that belongs left, until the pointers meet at mid.
        std::vector<uint32_t> n_l(chunks + 1, 0), n_r(chunks + 1, 0);
        pool.run(chunks, [&](int c) {
            uint32_t l = 0, r = 0;
            for (uint32_t i = lo_of(c), e = lo_of(c + 1); i < e; i++) {
                const bool left = side[i] <= best_split;
                if (i < mid) l += !left;
                else r += left;
            }
            n_l[c + 1] = l;
            n_r[c + 1] = r;
        });
        for (int c = 0; c < chunks; c++) { n_l[c + 1] += n_l[c]; n_r[c + 1] += n_r[c]; }
        const uint32_t K = n_l[chunks];
        pool.run(chunks, [&](int c) {
            uint32_t l = n_l[c], r = n_r[c];
            for (uint32_t i = lo_of(c), e = lo_of(c + 1); i < e; i++) {
                const bool left = side[i] <= best_split;
                if (i < mid) { if (!left) bad_l[l++] = i; }
                else if (left) bad_r[r++] = i;
            }
        });
        const int sw = pieces(K);
        pool.run(sw, [&](int c) {
            for (uint32_t k = (uint32_t)((uint64_t)K * c / sw), e = (uint32_t)((uint64_t)K * (c + 1) / sw); k < e; k++)
                swap_positions(bad_l[k], bad_r[K - 1 - k]);
        });
        return mid;
    }

    int build_top(uint32_t start, uint32_t end, uint32_t depth)
    {
        const int self = (int)top.size();
        top.emplace_back();
        const uint32_t n = end - start;
        if (n < kParMin || pool.threads() == 1) {
            top[self].task = (int)tasks.size();
            tasks.push_back(Task{start, end, depth, {}, 0, 0, 0});
            return self;
        }
        if (depth > top_depth) top_depth = depth;
        const uint32_t mid = split_parallel(start, end);
        const int l = build_top(start, mid, depth + 1);
        const int r = build_top(mid, end, depth + 1);
        top[self].left = l;
        top[self].right = r;
        return self;
    }

    // sizes and boxes of the top nodes, children first (after the tasks have run)
    void finish_top(int i)
    {
        TopNode &t = top[i];
        if (t.task >= 0) {
            const TreeNode &root = tasks[t.task].nodes[0];
            t.box = root.box;
            t.subtree = root.subtree;
            t.ntris = root.ntris;
            t.leaf_count = root.count;
            return;
        }
        finish_top(t.left);
        finish_top(t.right);
        const TopNode &l = top[t.left], &r = top[t.right];
        t.subtree = 1 + l.subtree + r.subtree;
        t.ntris = l.ntris + r.ntris;
        Bounds b = l.box;
        b.grow(r.box);
        t.box = b;
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



// Writes the flattened form (reference src/bvh.cu:234-294) straight into the output buffers.  The reference emits
// nodes depth first with the right child visited first, and a node's leaf children when the node itself is emitted
// (left leaf, then right leaf); a subtree therefore occupies one contiguous run of nodes and one of float4s, and the
// position of everything follows from subtree sizes.
struct Flattener {
    const Builder &b;
    float *gnodes;
    float *gtris;
    uint32_t *gids;

    uint32_t emit_leaf(uint32_t first, uint32_t count, uint64_t at) const
    {
        for (uint32_t k = 0; k < count; k++) {
            const uint32_t tri = b.order[first + k];
            woop_rows(b.verts + 9 * (size_t)tri, gtris + 4 * (at + 3 * k));
            uint32_t *id = gids + at + 3 * k;
            id[0] = tri; id[1] = PT_PRIM_MISS; id[2] = PT_PRIM_MISS;
        }
        const float term = bits_to_float(0x80000000u);
        float *t = gtris + 4 * (at + 3 * (uint64_t)count);
        t[0] = t[1] = t[2] = t[3] = term;
        gids[at + 3 * (uint64_t)count] = PT_PRIM_MISS;
        return ~(uint32_t)at;
    }

    static void boxes(float g[16], const Bounds &l, const Bounds &r)
    {
        const float v[12] = {l.lo[0], l.hi[0], l.lo[1], l.hi[1], r.lo[0], r.hi[0], r.lo[1], r.hi[1],
                             l.lo[2], l.hi[2], r.lo[2], r.hi[2]};
        std::memcpy(g, v, sizeof v);
        g[12] = g[13] = g[14] = g[15] = 0.0f;
    }

    // one task's subtree, its root an interior node that lands at (node_off, tri_off)
    void subtree(const std::vector<TreeNode> &tree, uint64_t node_off, uint64_t tri_off) const
    {
        struct Todo { uint32_t node; uint64_t node_off, tri_off; };
        std::vector<Todo> todo;
        todo.push_back({0, node_off, tri_off});
        while (!todo.empty()) {
            const Todo e = todo.back();
            todo.pop_back();
            const uint32_t li = e.node + 1, ri = li + tree[li].subtree;
            const TreeNode &l = tree[li], &r = tree[ri];
            float g[16];
            boxes(g, l.box, r.box);
            uint64_t toff = e.tri_off, noff = e.node_off + 1;
            if (l.count) { g[12] = bits_to_float(emit_leaf(l.first, l.count, toff)); toff += 3 * l.count + 1; }
            if (r.count) { g[13] = bits_to_float(emit_leaf(r.first, r.count, toff)); toff += 3 * r.count + 1; }
            if (!r.count) {
                g[13] = bits_to_float((uint32_t)noff);
                todo.push_back({ri, noff, toff});
                noff += interior_of(r.subtree);
                toff += float4s_of(r.subtree, r.ntris);
            }
            if (!l.count) {
                g[12] = bits_to_float((uint32_t)noff);
                todo.push_back({li, noff, toff});
            }
            std::memcpy(gnodes + 16 * e.node_off, g, sizeof g);
        }
    }

    // the tree top; subtrees built by tasks only get their position here
    void top(int i, uint64_t node_off, uint64_t tri_off, std::vector<Task> &tasks) const
    {
        const TopNode &t = b.top[i];
        if (t.task >= 0) {
            tasks[t.task].node_off = node_off;
            tasks[t.task].tri_off = tri_off;
            return;
        }
        const TopNode &l = b.top[t.left], &r = b.top[t.right];
        float g[16];
        boxes(g, l.box, r.box);
        uint64_t toff = tri_off, noff = node_off + 1;
        if (l.leaf_count) {
            g[12] = bits_to_float(emit_leaf(tasks[l.task].nodes[0].first, l.leaf_count, toff));
            toff += 3 * l.leaf_count + 1;
        }
        if (r.leaf_count) {
            g[13] = bits_to_float(emit_leaf(tasks[r.task].nodes[0].first, r.leaf_count, toff));
            toff += 3 * r.leaf_count + 1;
        }
        if (!r.leaf_count) {
            g[13] = bits_to_float((uint32_t)noff);
            top(t.right, noff, toff, tasks);
            noff += interior_of(r.subtree);
            toff += float4s_of(r.subtree, r.ntris);
        }
        if (!l.leaf_count) {
            g[12] = bits_to_float((uint32_t)noff);
            top(t.left, noff, toff, tasks);
        }
        std::memcpy(gnodes + 16 * node_off, g, sizeof g);
    }
};

double seconds_since(std::chrono::steady_clock::time_point t0)
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

}   // namespace

int bvh_build(const float *verts, uint64_t ntris, int threads, pth_bvh *out)
{
    std::memset(out, 0, sizeof *out);
    if (!verts) return set_error("pth_bvh_build: verts is NULL");
    if (ntris < 3) return set_error("pth_bvh_build: at least 3 triangles are needed (the root must be an interior node)");
    if (ntris >= (1ull << 29)) return set_error("pth_bvh_build: too many triangles");
    const bool timing = std::getenv("PTH_BUILD_TIMING") != nullptr;      // phase times on stderr
    const auto t0 = std::chrono::steady_clock::now();

    Pool pool(threads < 1 ? 1 : threads);
    Builder b(verts, ntris, pool);
    if (!b.pbox.p || !b.order.p) return set_error("pth_bvh_build: out of memory");
    if (pool.threads() > 1 && ntris >= kParMin) {
        b.side.resize(ntris);
        b.bad_l.resize(ntris / 2 + 1);
        b.bad_r.resize(ntris / 2 + 1);
    }
    const double t_prep = seconds_since(t0);
    const int root = b.build_top(0, (uint32_t)ntris, 0);
    const double t_top = seconds_since(t0);

    // the subtrees, largest first
    std::vector<int> by_size(b.tasks.size());
    for (size_t i = 0; i < by_size.size(); i++) by_size[i] = (int)i;
    std::sort(by_size.begin(), by_size.end(), [&](int x, int y) {
        const uint32_t nx = b.tasks[x].end - b.tasks[x].start, ny = b.tasks[y].end - b.tasks[y].start;
        return nx != ny ? nx > ny : x < y;
    });
    pool.run((int)by_size.size(), [&](int k) {
        Task &t = b.tasks[by_size[k]];
        t.nodes.reserve(2 * (size_t)(t.end - t.start));
        b.build(t.start, t.end, t.nodes, t.depth, t.max_depth);
    });
    b.finish_top(root);
    uint32_t max_depth = b.top_depth;
    for (const Task &t : b.tasks) max_depth = std::max(max_depth, t.max_depth);
    const double t_sub = seconds_since(t0);

    const TopNode &rt = b.top[root];
    const uint64_t num_nodes = interior_of(rt.subtree);
    const uint64_t num_f4 = 1 + float4s_of(rt.subtree, rt.ntris);       // float4 #0 is a zero pad (src/bvh.cu:240)
    out->num_nodes = num_nodes;
    out->num_tri_float4 = num_f4;
    out->num_tris = ntris;
    out->num_leaves = (rt.subtree + 1) / 2;
    out->max_depth = max_depth;
    out->nodes = (float *)std::malloc(num_nodes * 16 * sizeof(float));
    // two float4 of zero padding: a reader in the reference's style fetches three float4 before it looks for the
    // terminator (src/bvh.cuh:174-181)
    out->tris = (float *)std::calloc((num_f4 + 2) * 4, sizeof(float));
    out->prim_ids = (uint32_t *)std::malloc(num_f4 * sizeof(uint32_t));
    if (!out->nodes || !out->tris || !out->prim_ids) {
        std::free(out->nodes); std::free(out->tris); std::free(out->prim_ids);
        std::memset(out, 0, sizeof *out);
        return set_error("pth_bvh_build: out of memory");
    }
    out->prim_ids[0] = PT_PRIM_MISS;

    const Flattener f{b, out->nodes, out->tris, out->prim_ids};
    f.top(root, 0, 1, b.tasks);
    pool.run((int)by_size.size(), [&](int k) {
        const Task &t = b.tasks[by_size[k]];
        if (t.nodes[0].count == 0) f.subtree(t.nodes, t.node_off, t.tri_off);   // a single leaf was emitted by its parent
    });
    if (timing)
        std::fprintf(stderr, "pth_bvh_build: %llu triangles, %d threads: prepare %.2f s, top %.2f s (%zu subtrees), "
                     "subtrees %.2f s, flatten %.2f s\n", (unsigned long long)ntris, pool.threads(), t_prep,
                     t_top - t_prep, b.tasks.size(), t_sub - t_top, seconds_since(t0) - t_sub);
    return 0;
}

}   // namespace pth
