// libpt_hip.so: wavefront path tracing on MI355X (gfx950) behind the C ABI of include/pt_api.h.
// This file is the host side: contexts, scene upload (BVH re-layout into 64 B records), the render loop that queues
// the wavefront iterations on the lanes' streams, the closest-hit entry points, film handling and the film reduce.
// The kernels are in pt_wavefront.h (bounce, generate, film) and pt_extend.h (traversal).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "pt_api.h"
#include "pt_device.h"
#include "pt_extend.h"
#include "pt_wavefront.h"
#include "pt_path.h"

using namespace ptd;

namespace {

constexpr uint64_t kMaxRaysPerCall = 0xFFFFFFFFull - (1ull << 20);
constexpr uint32_t kDefaultPool = 1u << 27;   // upper bound; a render uses no more slots than it has paths

}   // namespace

// ---- device memory of the closest-hit entry point that allocates nothing (pt_intersect_device_noalloc) ---------------
// One array per GPU, held from the moment the code object is loaded until the process ends: the call's counters in the
// first 256 bytes, then one hit id per ray.  `scratch` is the host-side handle of the symbol, not a device address; host
// code uses the per-device address hipGetSymbolAddress returns (scratch_on_device), never the name.
constexpr uint64_t kNoAllocMaxRays = 1ull << 22;            // 4,194,304 rays per call: 16 MiB of hit ids
constexpr size_t kScratchHeader = 256;
constexpr int kNoAllocMaxStackSlots = 48;                   // the whole traversal stack in LDS: 48 KB per block
static_assert(sizeof(Counters) <= kScratchHeader, "the counters must fit the scratch header");
static __device__ __align__(256) unsigned char scratch[kScratchHeader + kNoAllocMaxRays * sizeof(uint32_t)];

// address of `scratch` on each device, resolved once per device (pt_ctx_create) with that device current
static std::mutex g_scratch_mu;
static std::vector<unsigned char *> g_scratch_ptr;

static hipError_t scratch_on_device(int device, unsigned char **out)
{
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (g_scratch_ptr.size() <= (size_t)device) g_scratch_ptr.resize((size_t)device + 1, nullptr);
    if (!g_scratch_ptr[device]) {
        void *p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        g_scratch_ptr[device] = (unsigned char *)p;
    }
    *out = g_scratch_ptr[device];
    return hipSuccess;
}

// One independent wavefront loop (own stream, path buffers and counters) rendering a share of a slab's samples.  The
// lanes of a context run side by side: the traversal kernel is bound by record fetches through the caches, bounce and
// generate by streaming state through HBM, so one lane's extend runs beside another lane's bounce.  Measured on the
// headline frame, 1 / 2 / 3 / 4 lanes: 324 / 280 / 269 / 288 ms.
// blocks per CU the grids of bounce and generate are capped at (their blocks loop over the rays)
constexpr int kBouncePerCu = 8, kGenPerCu = 8;
constexpr int kMaxLanes = 4;
// Render mode chosen per scene (PT_OPT_RENDER_MODE 0): whole paths in one kernel when less than this share of the rays is
// settled where they are made, else the wavefront loop.  The wavefront loop keeps settled rays out of the traversal kernel
// and handles them in full waves; the whole-path kernel saves every ray's trip through the path buffers.  Measured on
// MI355X, ms per 1920x1080 frame, wavefront / whole paths, with the settled share: spheres only (100 %) 50.9 / 60.7; bunny
// (65 %) 139.5 / 156.3; dragon stand-in (53 %) 230 / 237; city (31 %) 100.9 / 97.6; frame-filling stand-in (4 %) 791 / 723.
constexpr double kAutoSettledShare = 0.42;
struct Lane {
    hipStream_t stream = nullptr;
    uint32_t pool_alloc = 0;
    PathBuf buf[2]{};
    Counters *cnt = nullptr;
    Counters *h_cnt = nullptr;     // pinned: final counters [0], initial value [4]
    uint32_t *h_done = nullptr;    // pinned, written by k_advance: 1 when the lane has run out of work
    uint32_t *d_done = nullptr;    // the same word as the device addresses it
    int *gstk = nullptr;           // traversal stack overflow of this lane's extend launches
    size_t gstk_threads = 0;
    std::vector<hipEvent_t> events;
    hipEvent_t poll_ev[4]{};
    hipEvent_t ev_done = nullptr;
    // state of the slab share being rendered
    uint64_t first = 0, count = 0, total = 0, min_iters = 0, it = 0;
    uint32_t P = 0;
    uint32_t first_take = 0;       // cap on the paths started in the lane's first iteration
    float4 *slab = nullptr;
    size_t ev_used = 0;
    int cur = 0;
    bool done = true;
};

struct pt_ctx {
    int device = 0;
    bool counted = false;          // included in g_live_contexts
    hipStream_t stream = nullptr;
    std::string err;
    int num_cus = 256;

    // scene
    bool have_scene = false;
    DScene sc{};
    std::vector<void *> scene_allocs;

    // options
    uint32_t pool = kDefaultPool;
    bool count_traversal = false;
    bool time_extend = true;
    int extend_mode = 2;       // 2: persistent single-step waves (default), 1: one 64-ray batch per wave step
    int refill_idle = kRefillIdle;
    int lds_slots = 26;
    int extend_blocks = 8;     // cap on extend blocks per CU (the LDS stacks usually allow fewer)
    bool env_pack = true;      // use the packed environment texels when they are exact
    int xcd_group = 0;         // extend: each XCD's blocks work on one contiguous eighth of the ray buffer
    int node_layout = 1;       // 0: inner nodes in the order given, 1: parent + favoured child share a 128 B line
    int render_mode = 0;       // 1: wavefront iterations over path buffers, 2: whole paths in one persistent kernel,
                               // 0: chosen per scene from the first sample rendered (see kAutoSettledShare)
    int auto_mode = 0;         // the choice made for the scene and film in place (0: not made yet)
    double auto_settled = 0;   // the share of settled rays it was made on
    int last_mode = 0;         // mode the last pt_render call ran in
    int path_waves = 0;        // k_path_stream: cap on resident waves per CU (0: as many as fit)
    // k_path_stream: service once this many lanes wait for it.  Measured on MI355X, ms per frame at 44 / 48 / 52 / 56: dense
    // stand-in (128 spp) 182.1 / 181.1 / 181.3 / 184.5, city (256 spp) 95.4 / 94.2 / 93.4 / 93.7, 100.7 M triangles
    // (3840x2160x32) 346 / 339.8 / 336.3 / 335.2.
    int path_refill = 52;
    // k_path_stream: traversal stack slots per lane kept in LDS (256 B per slot and wave).  28 and not the 32 that would
    // fill the 160 KB of a CU with the 20 waves the registers allow: with all of the LDS taken the 100.7 M-triangle frame
    // (a tree that needs 35 slots) is 4 % slower (336.8 ms against 323.4 - 323.8 with 16 to 28 slots).
    int path_lds_slots = 28;
    PathCounters *pcnt = nullptr, *h_pcnt = nullptr;   // k_path_stream's counters, device and pinned host
    int *gstk = nullptr;           // overflow stack of the closest-hit entry points
    size_t gstk_threads = 0;
    uint32_t *tmp_hid = nullptr;   // hit ids of the closest-hit entry points
    size_t tmp_hid_n = 0;
    float4 *stage_o = nullptr, *stage_d = nullptr, *stage_h = nullptr;   // device staging of pt_intersect (host rays)
    float *stage_t = nullptr;
    uint32_t *stage_id = nullptr;
    uint64_t stage_cap = 0;
    unsigned char *scratch = nullptr;   // this device's copy of the static scratch array (pt_intersect_device_noalloc)

    // film
    float *film = nullptr;
    bool film_owned = false;
    uint32_t W = 0, H = 0;

    // sample slab: float4 [samples][pixels]
    float4 *slab = nullptr;
    size_t slab_bytes = 0;
    uint64_t slab_budget = 32ull << 30;

    // wavefront state: one lane per stream (see pt_render)
    int nlanes = 3;
    Lane lanes[kMaxLanes];
    Counters *cnt = nullptr;       // counters of the closest-hit entry points
    Counters *h_cnt = nullptr;     // pinned
    hipEvent_t ev_begin = nullptr, ev_end = nullptr, ev_slab_free = nullptr;
    hipEvent_t ev_k0 = nullptr, ev_k1 = nullptr;     // around a k_path_stream launch
};

static thread_local std::string g_create_err;

static int fail(pt_ctx *ctx, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    else g_create_err = buf;
    return code;
}

#define HIPCHK(ctx, call)                                                                                         \
    do {                                                                                                          \
        hipError_t e_ = (call);                                                                                   \
        if (e_ != hipSuccess)                                                                                     \
            return fail(ctx, PT_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

static int stream_grid(const pt_ctx *ctx, uint64_t n)
{
    const uint64_t cap = (uint64_t)ctx->num_cus * 8;
    const uint64_t need = (n + kBlock - 1) / kBlock;
    return (int)std::max<uint64_t>(1, std::min(cap, need));
}

static void free_pool(Lane &ln)
{
    for (auto &b : ln.buf) {
        (void)hipFree(b.o); (void)hipFree(b.d); (void)hipFree(b.thr); (void)hipFree(b.hid); (void)hipFree(b.t);
        b = PathBuf{};
    }
    ln.pool_alloc = 0;
}

// Grow-only: the lane's buffers hold at least `slots` paths afterwards.
static int ensure_pool(pt_ctx *ctx, Lane &ln, uint32_t slots)
{
    if (ln.pool_alloc >= slots) return PT_OK;
    free_pool(ln);
    const size_t P = slots;
    for (auto &b : ln.buf) {
        HIPCHK(ctx, hipMalloc(&b.o, P * sizeof(float4)));
        HIPCHK(ctx, hipMalloc(&b.d, P * sizeof(float4)));
        HIPCHK(ctx, hipMalloc(&b.thr, P * sizeof(float4)));
        HIPCHK(ctx, hipMalloc(&b.hid, P * sizeof(uint32_t)));
        HIPCHK(ctx, hipMalloc(&b.t, P * sizeof(float)));
    }
    ln.pool_alloc = slots;
    return PT_OK;
}

static void free_scene(pt_ctx *ctx)
{
    for (void *p : ctx->scene_allocs) (void)hipFree(p);
    ctx->scene_allocs.clear();
    ctx->have_scene = false;
    ctx->sc = DScene{};
}

// rgb -> one word (three 8-bit mantissas, one exponent E, value = mantissa * 2^(E - 136)) if that is exact.
static bool pack_shared_exponent(float4 c, uint32_t &word)
{
    const float ch[3] = {c.x, c.y, c.z};
    uint32_t bits[3];
    std::memcpy(bits, ch, 12);
    if (bits[0] == 0 && bits[1] == 0 && bits[2] == 0) { word = 0; return true; }
    for (int i = 0; i < 3; i++)
        if (!(ch[i] >= 0.0f) || std::isinf(ch[i]) || bits[i] == 0x80000000u) return false;
    int e = 0;
    (void)std::frexp(std::max(ch[0], std::max(ch[1], ch[2])), &e);
    const int E = e + 128;
    if (E < 10 || E > 255) return false;
    uint32_t m[3];
    for (int i = 0; i < 3; i++) {
        const float v = std::ldexp(ch[i], 8 - e);
        if (!(v >= 0.0f && v < 256.0f) || v != std::floor(v)) return false;
        m[i] = (uint32_t)v;
        const uint32_t sb = (uint32_t)(E - 9) << 23;
        float scale, back;
        std::memcpy(&scale, &sb, 4);
        back = (float)m[i] * scale;
        if (std::memcmp(&back, &ch[i], 4) != 0) return false;
    }
    word = m[0] | (m[1] << 8) | (m[2] << 16) | ((uint32_t)E << 24);
    return word != 0;
}

// Uninitialised host storage (a std::vector would first fill gigabytes on one thread).
template <class T>
struct HostArray {
    T *p;
    explicit HostArray(size_t n) : p((T *)std::malloc(std::max<size_t>(n, 1) * sizeof(T))) {}
    ~HostArray() { std::free(p); }
    HostArray(const HostArray &) = delete;
    HostArray &operator=(const HostArray &) = delete;
    T &operator[](size_t i) { return p[i]; }
};

// host threads worth using for n independent items
static size_t host_threads(uint64_t n)
{
    const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    return (size_t)std::max<uint64_t>(1, std::min<uint64_t>(hw, n / 16384));
}

// fn(part, lo, hi) over [0, n) cut into `parts` ranges, one host thread each
template <class F> static void parallel_chunks(uint64_t n, size_t parts, F fn)
{
    std::vector<std::thread> th;
    for (size_t p = 1; p < parts; p++) th.emplace_back([=] { fn(p, n * p / parts, n * (p + 1) / parts); });
    fn(0, 0, n / parts);
    for (auto &t : th) t.join();
}

template <class T> static int upload_raw(pt_ctx *ctx, const T *src, size_t n, const T **dptr)
{
    T *d = nullptr;
    HIPCHK(ctx, hipMalloc(&d, std::max<size_t>(1, n) * sizeof(T)));
    ctx->scene_allocs.push_back(d);
    if (n) HIPCHK(ctx, hipMemcpy(d, src, n * sizeof(T), hipMemcpyHostToDevice));
    *dptr = d;
    return PT_OK;
}

template <class T> static int upload(pt_ctx *ctx, const std::vector<T> &v, const T **dptr, size_t pad_elems = 0)
{
    T *d = nullptr;
    const size_t n = std::max<size_t>(1, v.size() + pad_elems);
    HIPCHK(ctx, hipMalloc(&d, n * sizeof(T)));
    ctx->scene_allocs.push_back(d);
    HIPCHK(ctx, hipMemset(d, 0, n * sizeof(T)));
    if (!v.empty()) HIPCHK(ctx, hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *dptr = d;
    return PT_OK;
}

// A lane's traversal stack needs (tree depth + 1) slots; each slot is 1 KB of LDS per 256-thread block, and the blocks a
// CU holds are 160 KB / that (at most 8).  Trees needing more than ctx->lds_slots slots keep the rest in a global
// array.  Measured on the 868 k-triangle stand-in (26 slots): 26 slots in LDS, 6 blocks per CU, beat 20 slots + overflow
// at 8 blocks per CU by 3 %.
// gstk == NULL: the whole stack in LDS, whatever its depth (the caller has checked that it fits); nothing is allocated.
static int launch_extend(pt_ctx *ctx, hipStream_t stream, Counters *cnt, bool n_on_device, int **gstk,
                         size_t *gstk_threads, const float4 *ro, const float4 *rd, uint32_t n_fixed, uint32_t n_upper,
                         uint32_t *hid, float *t)
{
    // slots the LDS column has to hold: tree depth + 1
    const uint32_t need = ctx->sc.has_bvh ? std::max<uint32_t>(1, ctx->sc.stack_slots) : 1;
    const int lds_slots = gstk ? (int)std::min<uint32_t>(need, (uint32_t)ctx->lds_slots) : (int)need;
    const bool overflow = need > (uint32_t)lds_slots;
    const int per_cu = std::max(1, std::min(ctx->extend_blocks, 160 / lds_slots));
    const uint64_t cap = (uint64_t)ctx->num_cus * per_cu;
    const uint64_t want = ((uint64_t)n_upper + kBlock - 1) / kBlock;
    const int grid = (int)std::max<uint64_t>(1, std::min(cap, want));
    if (overflow) {
        const size_t threads = (size_t)ctx->num_cus * 8 * kBlock;
        if (!*gstk || *gstk_threads != threads) {
            if (*gstk) { (void)hipFree(*gstk); *gstk = nullptr; }
            HIPCHK(ctx, hipMalloc(gstk, threads * (size_t)(kMaxTreeDepth + 1) * sizeof(int)));
            *gstk_threads = threads;
        }
    }
    ExtendArgs a{};
    a.ro = ro; a.rd = rd;
    a.n_ptr = n_on_device ? &cnt->n_in : nullptr;
    a.n_fixed = n_fixed;
    a.hid_out = hid; a.t_out = t;
    a.cnt3 = &cnt->trav[0];
    a.refill_idle = ctx->refill_idle;
    a.lds_slots = lds_slots;
    a.gstk = gstk ? *gstk : nullptr;
    a.xcd_group = ctx->xcd_group;
    const size_t shmem = (size_t)lds_slots * kBlock * sizeof(int);
    const bool count = ctx->count_traversal;
    const dim3 g(grid), b(kBlock);
#define PT_LAUNCH(K, OV, CT) hipLaunchKernelGGL((K<OV, CT>), g, b, shmem, stream, ctx->sc, a)
    if (ctx->extend_mode == 1) {
        if (overflow) { if (count) PT_LAUNCH(k_extend_batch, true, true); else PT_LAUNCH(k_extend_batch, true, false); }
        else { if (count) PT_LAUNCH(k_extend_batch, false, true); else PT_LAUNCH(k_extend_batch, false, false); }
    } else {
        if (overflow) { if (count) PT_LAUNCH(k_extend_stream, true, true); else PT_LAUNCH(k_extend_stream, true, false); }
        else { if (count) PT_LAUNCH(k_extend_stream, false, true); else PT_LAUNCH(k_extend_stream, false, false); }
    }
#undef PT_LAUNCH
    HIPCHK(ctx, hipGetLastError());
    return PT_OK;
}

// Points a lane at its share of a slab: `count` samples from `first`, slab entries at `slab`.
static int lane_begin(pt_ctx *ctx, Lane &ln, uint64_t first, uint64_t count, float4 *slab, int nlanes)
{
    const uint64_t npix = (uint64_t)ctx->W * ctx->H;
    ln.first = first;
    ln.count = count;
    ln.total = npix * count;
    ln.slab = slab;
    // path pool: the lane's share of the configured size, but no more slots than it has paths
    const uint64_t want = (ln.total + 65535) & ~65535ull;
    ln.P = (uint32_t)std::min<uint64_t>(std::max<uint32_t>(1024u, ctx->pool / (uint32_t)nlanes), want);
    int rc = ensure_pool(ctx, ln, ln.P);
    if (rc != PT_OK) return rc;
    ln.min_iters = (ln.total + ln.P - 1) / ln.P;
    // Lanes that all start by filling their whole pool stay in step: three generate launches side by side, then three
    // extend launches, and kernels of one kind compete for the same unit.  The k-th of n lanes therefore starts only
    // (k+1)/n of its slots in its first iteration (the rest follow in the second, which refills the free slots anyway),
    // so that from the start one lane traverses while another generates.  Measured on the headline scene: a 64-spp share
    // 32.0 -> 30.8 ms, 128 spp 61.4 -> 59.2, the 512-spp frame 233 -> 230.6.  Which iteration starts a sample does not
    // change its slab entry.  A scene without a tree has no extend launches to interleave with: all slots at once.
    ln.first_take = ln.P;
    if (ctx->sc.has_bvh) {
        const int k = (int)(&ln - &ctx->lanes[0]);
        ln.first_take = std::max<uint32_t>(1024u, (uint32_t)((uint64_t)ln.P * (k + 1) / nlanes));
    }
    ln.it = 0;
    ln.cur = 0;
    ln.ev_used = 0;
    ln.done = false;
    __atomic_store_n(ln.h_done, 0u, __ATOMIC_RELEASE);
    Counters &init = ln.h_cnt[4];
    init = Counters{};
    init.total_samples = ln.total;
    HIPCHK(ctx, hipStreamWaitEvent(ln.stream, ctx->ev_slab_free, 0));     // the previous slab has been reduced
    HIPCHK(ctx, hipMemcpyAsync(ln.cnt, &init, sizeof init, hipMemcpyHostToDevice, ln.stream));
    return PT_OK;
}

// Queues one wavefront iteration of a lane (bounce, take, generate, advance, extend) and, once the lane can have run
// out of work, polls its device counters two iterations behind the launch front; sets ln.done when they say so.
static int lane_iterate(pt_ctx *ctx, Lane &ln, uint32_t s0, uint32_t s1)
{
    const uint32_t W = ctx->W, H = ctx->H, npix = W * H, P = ln.P;
    hipStream_t ls = ln.stream;
    PathBuf A = ln.buf[ln.cur], B = ln.buf[ln.cur ^ 1];
    const int gb = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)ctx->num_cus * kBouncePerCu,
                                                                 ((uint64_t)P + kBounceBlock - 1) / kBounceBlock));
    if (ln.it > 0)
        hipLaunchKernelGGL(k_bounce, dim3(gb), dim3(kBounceBlock), 0, ls, ctx->sc, A, B, ln.cnt, P, ln.slab,
                           (uint32_t)ln.first, npix, s0, s1);
    hipLaunchKernelGGL(k_take, dim3(1), dim3(64), 0, ls, ln.cnt, P, ln.it == 0 ? ln.first_take : P);
    const int gg = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)ctx->num_cus * kGenPerCu,
                                                                 ((uint64_t)P + kGenBlock - 1) / kGenBlock));
    hipLaunchKernelGGL(k_generate, dim3(gg), dim3(kGenBlock), 0, ls, ctx->sc, B, ln.cnt, P, W, H,
                       make_fastdiv(pixel_divisor(W, H)), (uint32_t)ln.first, s0, s1, ln.slab);
    hipLaunchKernelGGL(k_advance, dim3(1), dim3(64), 0, ls, ln.cnt, (ctx->count_traversal && ctx->sc.has_bvh) ? 1u : 0u,
                       ln.d_done);
    HIPCHK(ctx, hipGetLastError());         // of the four launches above
    if (ctx->time_extend) {
        while (ln.events.size() < ln.ev_used + 2) {
            hipEvent_t ev;
            HIPCHK(ctx, hipEventCreate(&ev));
            ln.events.push_back(ev);
        }
        HIPCHK(ctx, hipEventRecord(ln.events[ln.ev_used], ls));
    }
    int rc = launch_extend(ctx, ls, ln.cnt, true, &ln.gstk, &ln.gstk_threads, B.o, B.d, 0, P, B.hid, B.t);
    if (rc != PT_OK) return rc;
    if (ctx->time_extend) {
        HIPCHK(ctx, hipEventRecord(ln.events[ln.ev_used + 1], ls));
        ln.ev_used += 2;
    }
    ln.cur ^= 1;
    ln.it++;
    if (ln.it >= ln.min_iters) {
        const int slot = (int)(ln.it & 3);
        HIPCHK(ctx, hipEventRecord(ln.poll_ev[slot], ls));
        if (ln.it >= ln.min_iters + 2) {
            // wait for the iteration two behind the launch front, then look at the word k_advance writes
            const int old = (int)((ln.it - 2) & 3);
            HIPCHK(ctx, hipEventSynchronize(ln.poll_ev[old]));
            if (__atomic_load_n(ln.h_done, __ATOMIC_ACQUIRE) != 0) ln.done = true;
        }
    }
    return PT_OK;
}

// Error exit of pt_render: waits for whatever the lanes have queued, so that the context's buffers are quiet when the
// caller sees the error and the next call starts from a defined state.
static int drain_lanes(pt_ctx *ctx, int rc)
{
    for (auto &ln : ctx->lanes) {
        if (ln.stream) (void)hipStreamSynchronize(ln.stream);
        ln.done = true;
    }
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipGetLastError();
    return rc;
}

// ---- RCCL, bound at first use ------------------------------------------------------------------------------------
// pt_film_reduce is the library's only collective and only a process that drives several GPUs calls it, so the library
// does not link against RCCL: the few entry points are looked up when the first reduce over distinct devices runs.  A
// communicator set is kept per device list (creating one takes far longer than a reduce) until the last context of
// the process is destroyed.
namespace {

struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Reduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;      // why it is unavailable
    bool tried = false;
};

struct CommSet {
    std::vector<int> devices;
    std::vector<ncclComm_t> comms;
};

std::mutex g_rccl_mu;
RcclApi g_rccl;
std::vector<CommSet> g_comm_sets;
int g_live_contexts = 0;

bool rccl_load()
{
    if (g_rccl.tried) return g_rccl.lib != nullptr;
    g_rccl.tried = true;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        g_rccl.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (g_rccl.lib) break;
    }
    if (!g_rccl.lib) { g_rccl.why = "librccl.so.1 not found"; return false; }
#define PT_SYM(field, sym)                                                       \
    g_rccl.field = (decltype(g_rccl.field))dlsym(g_rccl.lib, sym);               \
    if (!g_rccl.field) { g_rccl.why = std::string("RCCL lacks ") + sym; g_rccl.lib = nullptr; return false; }
    PT_SYM(CommInitAll, "ncclCommInitAll")
    PT_SYM(CommDestroy, "ncclCommDestroy")
    PT_SYM(Reduce, "ncclReduce")
    PT_SYM(GroupStart, "ncclGroupStart")
    PT_SYM(GroupEnd, "ncclGroupEnd")
    PT_SYM(GetErrorString, "ncclGetErrorString")
#undef PT_SYM
    return true;
}

void rccl_drop_all()
{
    for (auto &cs : g_comm_sets)
        for (ncclComm_t c : cs.comms) (void)g_rccl.CommDestroy(c);
    g_comm_sets.clear();
}

// One ncclReduce(sum, f32) of the films onto ctxs[root], every rank on its own context's stream.
int film_reduce_rccl(pt_ctx *const *ctxs, int n, int root, uint64_t count)
{
    pt_ctx *dst = ctxs[root];
    std::lock_guard<std::mutex> lock(g_rccl_mu);
    if (!rccl_load()) return fail(dst, PT_ERR_STATE, "pt_film_reduce: %s", g_rccl.why.c_str());
    std::vector<int> devs(n);
    for (int i = 0; i < n; i++) devs[i] = ctxs[i]->device;
    CommSet *cs = nullptr;
    for (auto &c : g_comm_sets)
        if (c.devices == devs) cs = &c;
    if (!cs) {
        CommSet fresh;
        fresh.devices = devs;
        fresh.comms.resize(n);
        const ncclResult_t r = g_rccl.CommInitAll(fresh.comms.data(), n, devs.data());
        if (r != ncclSuccess) return fail(dst, PT_ERR_HIP, "ncclCommInitAll: %s", g_rccl.GetErrorString(r));
        g_comm_sets.push_back(fresh);
        cs = &g_comm_sets.back();
    }
    ncclResult_t r = g_rccl.GroupStart();
    for (int i = 0; i < n && r == ncclSuccess; i++) {
        if (hipSetDevice(ctxs[i]->device) != hipSuccess) { r = ncclUnhandledCudaError; break; }
        // in place on the root; the other ranks' receive buffers are not written
        r = g_rccl.Reduce(ctxs[i]->film, ctxs[i]->film, count, ncclFloat32, ncclSum, root, cs->comms[i], ctxs[i]->stream);
    }
    const ncclResult_t r2 = g_rccl.GroupEnd();
    if (r == ncclSuccess) r = r2;
    if (r != ncclSuccess) return fail(dst, PT_ERR_HIP, "ncclReduce: %s", g_rccl.GetErrorString(r));
    for (int i = 0; i < n; i++) {
        HIPCHK(dst, hipSetDevice(ctxs[i]->device));
        HIPCHK(dst, hipStreamSynchronize(ctxs[i]->stream));
    }
    HIPCHK(dst, hipSetDevice(dst->device));
    return PT_OK;
}

}   // namespace

// Record index of every inner node.  Layout 0 keeps the order given (the reference's flatten order, src/bvh.cu:234-294).
// Layout 1 pairs nodes so that a node and the child a ray is most likely to visit next share one 128 B cache line: the
// traversal fetches 64 B records, the caches fill 128 B lines, and with the order given the other half of a line is
// rarely the record the ray needs next.  Top-down greedy matching: a node at the head of a line takes its inner child
// with the larger box area (the chance that a ray through the parent's box hits the child's is proportional to it) as
// the tail of the line; the other child and the tail's children head lines of their own.  Pairs come first, in
// depth-first order (the root at record 0), then the nodes left single, two to a line, then anything unreachable.
// Only indices change: visit order and arithmetic of the traversal are the same, so are its answers.
// Returns false when the child references do not form a tree.
static bool node_layout_pairs(const float *nodes, uint64_t N, std::vector<uint32_t> &newid)
{
    const uint32_t none = 0xFFFFFFFFu;
    newid.assign(N, none);
    auto child = [&](uint64_t i, int c) { int32_t v; std::memcpy(&v, nodes + i * 16 + 12 + c, 4); return v; };
    auto half_area = [&](uint64_t i, int c) {
        const float *n = nodes + i * 16;
        const float dx = n[4 * c + 1] - n[4 * c], dy = n[4 * c + 3] - n[4 * c + 2], dz = n[8 + 2 * c + 1] - n[8 + 2 * c];
        return dx * dy + dy * dz + dz * dx;
    };
    // pass 1: the matching, depth first.  state: 1 head of a pair, 2 tail, 3 single
    std::vector<uint8_t> state(N, 0);
    std::vector<uint32_t> order;            // heads and singles in depth-first order
    order.reserve(N);
    std::vector<uint32_t> tail_of(N, none);
    std::vector<uint32_t> st;
    st.push_back(0);
    uint64_t seen = 0;
    while (!st.empty()) {
        const uint32_t p = st.back();
        st.pop_back();
        if (state[p]) return false;
        if (++seen > N) return false;
        const int32_t c0 = child(p, 0), c1 = child(p, 1);
        const bool in0 = c0 > 0, in1 = c1 > 0;
        order.push_back(p);
        if (!in0 && !in1) { state[p] = 3; continue; }
        int fav = in0 ? 0 : 1;
        if (in0 && in1 && half_area(p, 1) > half_area(p, 0)) fav = 1;
        const uint32_t f = (uint32_t)(fav ? c1 : c0);
        if (state[f]) return false;
        if (++seen > N) return false;
        state[p] = 1;
        state[f] = 2;
        tail_of[p] = f;
        // pushed last = visited first: the tail's children, then the head's other child
        const int32_t o = fav ? c0 : c1;
        if (o > 0) st.push_back((uint32_t)o);
        const int32_t f0 = child(f, 0), f1 = child(f, 1);
        if (f1 > 0) st.push_back((uint32_t)f1);
        if (f0 > 0) st.push_back((uint32_t)f0);
    }
    // pass 2: positions
    uint32_t pos = 0;
    for (uint32_t p : order)
        if (state[p] == 1) { newid[p] = pos++; newid[tail_of[p]] = pos++; }
    for (uint32_t p : order)
        if (state[p] == 3) newid[p] = pos++;
    for (uint64_t i = 0; i < N; i++)
        if (newid[i] == none) newid[i] = pos++;
    return pos == N && newid[0] == 0;
}

// What a pt_render call adds up over its slabs and kernels.
struct RenderAcc {
    pt_stats st{};
    double ext_ms = 0;                                     // summed durations of the traversal launches
    std::vector<std::pair<double, double>> ext_spans;     // begin, end of every traversal launch, ms after ev_begin
};

// Render mode 2 for the samples [first, first + count_s) of a slab whose entries for them start at `rows`: ONE k_path_stream
// launch over all their camera paths on the context's stream, waited for; its counters are added to acc.
// settled_share (may be NULL) receives the share of the launch's rays that were settled where they were made.
static int paths_run(pt_ctx *ctx, uint64_t first, uint64_t count_s, float4 *rows, uint32_t s0, uint32_t s1, RenderAcc &acc,
                     double *settled_share)
{
    const uint32_t W = ctx->W, H = ctx->H, npix = W * H;
    hipStream_t st = ctx->stream;
    // traversal stack: tree depth + 1 slots per lane, 256 B of LDS per slot and wave; beyond path_lds_slots the rest goes
    // to a global array
    const uint32_t need = ctx->sc.has_bvh ? std::max<uint32_t>(1, ctx->sc.stack_slots) : 1;
    const int lds_slots = (int)std::min<uint32_t>(need, (uint32_t)ctx->path_lds_slots);
    const bool overflow = need > (uint32_t)lds_slots;
    const size_t shmem = (size_t)lds_slots * kPathBlock * sizeof(int);
    const bool count = ctx->count_traversal;
    const void *kern = overflow ? (count ? (const void *)k_path_stream<true, true> : (const void *)k_path_stream<true, false>)
                                : (count ? (const void *)k_path_stream<false, true> : (const void *)k_path_stream<false, false>);
    int per_cu = 0;
    HIPCHK(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, kPathBlock, shmem));
    if (per_cu < 1) return fail(ctx, PT_ERR_HIP, "pt_render: the path kernel does not fit a CU (%zu B of LDS)", shmem);
    if (ctx->path_waves > 0) per_cu = std::min(per_cu, ctx->path_waves);
    if (std::getenv("PT_DEBUG"))
        std::fprintf(stderr, "[pt] k_path_stream: %d waves per CU, %d stack slots in LDS%s, tree needs %u\n", per_cu, lds_slots,
                     overflow ? ", overflow to global" : "", need);
    const uint32_t cps = (npix + kPathChunk - 1) / kPathChunk;
    if ((uint64_t)cps * count_s >= 0xFFFF0000ull) return fail(ctx, PT_ERR_ARG, "pt_render: sample slab too large");
    const uint32_t total_chunks = (uint32_t)(cps * count_s);
    const int grid = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)ctx->num_cus * per_cu,
                                                                   ((uint64_t)npix * count_s + kPathBlock - 1) / kPathBlock));
    if (overflow) {
        const size_t threads = (size_t)grid * kPathBlock;
        if (!ctx->gstk || ctx->gstk_threads < threads) {
            if (ctx->gstk) { (void)hipFree(ctx->gstk); ctx->gstk = nullptr; ctx->gstk_threads = 0; }
            const size_t want = std::max(threads, (size_t)ctx->num_cus * 8 * kBlock);
            HIPCHK(ctx, hipMalloc(&ctx->gstk, want * (size_t)(kMaxTreeDepth + 1) * sizeof(int)));
            ctx->gstk_threads = want;
        }
    }
    PathArgs a{};
    a.slab = rows;
    a.slab_first_sample = (uint32_t)first;
    a.W = W; a.H = H; a.npix = npix;
    a.row_div = make_fastdiv(pixel_divisor(W, H));
    a.chunks_per_sample = cps;
    a.cps_div = make_fastdiv(cps);
    a.total_chunks = total_chunks;
    a.seed0 = s0; a.seed1 = s1;
    a.cnt = ctx->pcnt;
    a.refill_idle = ctx->path_refill;
    a.lds_slots = lds_slots;
    a.gstk = ctx->gstk;
    HIPCHK(ctx, hipMemsetAsync(ctx->pcnt, 0, sizeof(PathCounters), st));
    HIPCHK(ctx, hipEventRecord(ctx->ev_k0, st));
    const dim3 g(grid), b(kPathBlock);
    if (overflow) {
        if (count) hipLaunchKernelGGL((k_path_stream<true, true>), g, b, shmem, st, ctx->sc, a);
        else hipLaunchKernelGGL((k_path_stream<true, false>), g, b, shmem, st, ctx->sc, a);
    } else {
        if (count) hipLaunchKernelGGL((k_path_stream<false, true>), g, b, shmem, st, ctx->sc, a);
        else hipLaunchKernelGGL((k_path_stream<false, false>), g, b, shmem, st, ctx->sc, a);
    }
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipEventRecord(ctx->ev_k1, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_pcnt, ctx->pcnt, sizeof(PathCounters), hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    float t0 = 0, ms = 0;
    HIPCHK(ctx, hipEventElapsedTime(&t0, ctx->ev_begin, ctx->ev_k0));
    HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev_k0, ctx->ev_k1));
    acc.ext_ms += ms;
    acc.ext_spans.push_back({(double)t0, (double)t0 + ms});
    const PathCounters &c = *ctx->h_pcnt;
    acc.st.rays += c.rays;
    acc.st.paths += c.paths;
    acc.st.rays_escaped += c.settled;
    acc.st.iterations += 1;
    acc.st.extend_launches += (ctx->sc.has_bvh && c.rays > c.settled) ? 1 : 0;
    acc.st.nodes += c.trav[0]; acc.st.tris += c.trav[1]; acc.st.leaves += c.trav[2];
    if (settled_share) *settled_share = c.rays ? (double)c.settled / (double)c.rays : 1.0;
    return PT_OK;
}

// Render mode 1 for the samples [first, first + count) of a slab whose entries for them start at `rows`: the samples are
// split over the lanes, the lanes' iterations are queued round robin until every lane has run out of work, and the
// context's stream is made to wait for them (on the device), so that what the caller queues there next -- the pass that
// folds the slab into the film -- starts when the last lane ends.  Returns the number of lanes used; their counters are
// on their way to pinned memory and are read by wavefront_collect once the caller has waited for the context's stream.
static int wavefront_run(pt_ctx *ctx, uint64_t first, uint64_t count, float4 *rows, uint32_t s0, uint32_t s1, int *lanes_used)
{
    const uint32_t npix = ctx->W * ctx->H;
    hipStream_t st = ctx->stream;
    // Which lane renders a sample does not change its slab entry, so the film does not depend on the lane count.
    const int nl = (int)std::min<uint64_t>((uint64_t)ctx->nlanes, count);
    *lanes_used = nl;
    for (int k = 0; k < nl; k++) {
        const uint64_t b = count * k / nl, e = count * (k + 1) / nl;
        int rc = lane_begin(ctx, ctx->lanes[k], first + b, e - b, rows + (size_t)b * npix, nl);
        if (rc != PT_OK) return rc;
    }
    // round robin: one iteration per live lane, so that the lanes' kernels are queued side by side
    for (int live = nl; live > 0;) {
        for (int k = 0; k < nl; k++) {
            Lane &ln = ctx->lanes[k];
            if (ln.done) continue;
            int rc = lane_iterate(ctx, ln, s0, s1);
            if (rc != PT_OK) return rc;
            if (ln.done) live--;
        }
    }
    for (int k = 0; k < nl; k++) {
        Lane &ln = ctx->lanes[k];
        HIPCHK(ctx, hipMemcpyAsync(&ln.h_cnt[0], ln.cnt, sizeof(Counters), hipMemcpyDeviceToHost, ln.stream));
        HIPCHK(ctx, hipEventRecord(ln.ev_done, ln.stream));
        HIPCHK(ctx, hipStreamWaitEvent(st, ln.ev_done, 0));
    }
    return PT_OK;
}

static int wavefront_collect(pt_ctx *ctx, int nl, RenderAcc &acc)
{
    for (int k = 0; k < nl; k++) {
        Lane &ln = ctx->lanes[k];
        const Counters &c = ln.h_cnt[0];
        acc.st.rays += c.rays_total;
        acc.st.paths += c.paths_total;
        acc.st.iterations += c.iter;
        acc.st.extend_launches += c.extend_nonempty;
        acc.st.rays_escaped += c.escaped_total;
        acc.st.nodes += c.trav[0]; acc.st.tris += c.trav[1]; acc.st.leaves += c.trav[2];
        for (size_t i = 0; i + 1 < ln.ev_used; i += 2) {
            float b = 0, e = 0;
            HIPCHK(ctx, hipEventElapsedTime(&b, ctx->ev_begin, ln.events[i]));
            HIPCHK(ctx, hipEventElapsedTime(&e, ln.events[i], ln.events[i + 1]));
            acc.ext_ms += e;
            acc.ext_spans.push_back({(double)b, (double)b + e});
        }
    }
    return PT_OK;
}

extern "C" {

int pt_ctx_create(int device, pt_ctx **out)
{
    if (!out) return fail(nullptr, PT_ERR_ARG, "pt_ctx_create: out is NULL");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, PT_ERR_HIP, "pt_ctx_create: no HIP device (%s)", hipGetErrorString(e));
    if (device < 0 || device >= ndev) return fail(nullptr, PT_ERR_ARG, "pt_ctx_create: device %d of %d", device, ndev);
    pt_ctx *ctx = new pt_ctx();
    ctx->device = device;
#define CREATECHK(call)                                                                           \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            fail(nullptr, PT_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_));            \
            pt_ctx_destroy(ctx);                                                                  \
            return PT_ERR_HIP;                                                                    \
        }                                                                                         \
    } while (0)
    CREATECHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    CREATECHK(hipGetDeviceProperties(&prop, device));
    ctx->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    CREATECHK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    CREATECHK(scratch_on_device(device, &ctx->scratch));
    CREATECHK(hipMalloc(&ctx->cnt, sizeof(Counters)));
    CREATECHK(hipHostMalloc(&ctx->h_cnt, sizeof(Counters)));
    CREATECHK(hipMalloc(&ctx->pcnt, sizeof(PathCounters)));
    CREATECHK(hipHostMalloc(&ctx->h_pcnt, sizeof(PathCounters)));
    for (auto &ln : ctx->lanes) {
        CREATECHK(hipStreamCreateWithFlags(&ln.stream, hipStreamNonBlocking));
        CREATECHK(hipMalloc(&ln.cnt, sizeof(Counters)));
        CREATECHK(hipHostMalloc(&ln.h_cnt, 5 * sizeof(Counters)));
        CREATECHK(hipHostMalloc(&ln.h_done, 64, hipHostMallocMapped | hipHostMallocCoherent));
        CREATECHK(hipHostGetDevicePointer((void **)&ln.d_done, ln.h_done, 0));
        for (auto &ev : ln.poll_ev) CREATECHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        CREATECHK(hipEventCreateWithFlags(&ln.ev_done, hipEventDisableTiming));
    }
    CREATECHK(hipEventCreate(&ctx->ev_begin));
    CREATECHK(hipEventCreate(&ctx->ev_end));
    CREATECHK(hipEventCreate(&ctx->ev_slab_free));
    CREATECHK(hipEventCreate(&ctx->ev_k0));
    CREATECHK(hipEventCreate(&ctx->ev_k1));
#undef CREATECHK
    {
        std::lock_guard<std::mutex> lock(g_rccl_mu);
        g_live_contexts++;
    }
    ctx->counted = true;
    *out = ctx;
    return PT_OK;
}

void pt_ctx_destroy(pt_ctx *ctx)
{
    if (!ctx) return;
    if (ctx->counted) {
        // the communicators go with the last context of the process
        std::lock_guard<std::mutex> lock(g_rccl_mu);
        if (--g_live_contexts == 0) rccl_drop_all();
    }
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &ln : ctx->lanes) {
        if (ln.stream) (void)hipStreamSynchronize(ln.stream);
        free_pool(ln);
        (void)hipFree(ln.cnt);
        (void)hipFree(ln.gstk);
        if (ln.h_cnt) (void)hipHostFree(ln.h_cnt);
        if (ln.h_done) (void)hipHostFree(ln.h_done);
        for (auto ev : ln.events) (void)hipEventDestroy(ev);
        for (auto ev : ln.poll_ev) if (ev) (void)hipEventDestroy(ev);
        if (ln.ev_done) (void)hipEventDestroy(ln.ev_done);
        if (ln.stream) (void)hipStreamDestroy(ln.stream);
    }
    free_scene(ctx);
    if (ctx->film_owned) (void)hipFree(ctx->film);
    (void)hipFree(ctx->cnt);
    (void)hipFree(ctx->pcnt);
    if (ctx->h_pcnt) (void)hipHostFree(ctx->h_pcnt);
    (void)hipFree(ctx->gstk);
    (void)hipFree(ctx->tmp_hid);
    (void)hipFree(ctx->stage_o); (void)hipFree(ctx->stage_d); (void)hipFree(ctx->stage_h);
    (void)hipFree(ctx->stage_t); (void)hipFree(ctx->stage_id);
    (void)hipHostFree(ctx->h_cnt);
    (void)hipFree(ctx->slab);
    if (ctx->ev_begin) (void)hipEventDestroy(ctx->ev_begin);
    if (ctx->ev_end) (void)hipEventDestroy(ctx->ev_end);
    if (ctx->ev_slab_free) (void)hipEventDestroy(ctx->ev_slab_free);
    if (ctx->ev_k0) (void)hipEventDestroy(ctx->ev_k0);
    if (ctx->ev_k1) (void)hipEventDestroy(ctx->ev_k1);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char *pt_last_error(const pt_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

int pt_env_is_packed(const pt_ctx *ctx)
{
    if (!ctx || !ctx->have_scene) return PT_ERR_STATE;
    return ctx->sc.envp ? 1 : 0;
}

void *pt_stream(pt_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

int pt_set_option(pt_ctx *ctx, int option, uint64_t value)
{
    if (!ctx) return PT_ERR_ARG;
    switch (option) {
    case PT_OPT_POOL_SIZE:
        if (value < 1024 || value > (1ull << 28)) return fail(ctx, PT_ERR_ARG, "pool size %llu out of [1024, 2^28]",
                                                              (unsigned long long)value);
        ctx->pool = (uint32_t)value;
        return PT_OK;
    case PT_OPT_SLAB_BYTES:
        if (value < (1ull << 20)) return fail(ctx, PT_ERR_ARG, "sample slab budget below 1 MiB");
        ctx->slab_budget = value;
        return PT_OK;
    case PT_OPT_COUNT_TRAVERSAL: ctx->count_traversal = value != 0; return PT_OK;
    case PT_OPT_TIME_EXTEND: ctx->time_extend = value != 0; return PT_OK;
    case PT_OPT_EXTEND_MODE:
        if (value != 1 && value != 2) return fail(ctx, PT_ERR_ARG, "extend mode %llu unknown", (unsigned long long)value);
        ctx->extend_mode = (int)value;
        return PT_OK;
    case PT_OPT_TUNE_LDS_SLOTS:
        if (value < 2 || value > 64) return fail(ctx, PT_ERR_ARG, "LDS stack slots %llu out of [2, 64]", (unsigned long long)value);
        ctx->lds_slots = (int)value;
        return PT_OK;
    case PT_OPT_TUNE_REFILL_IDLE:
        if (value < 1 || value > 64) return fail(ctx, PT_ERR_ARG, "refill threshold %llu out of [1, 64]", (unsigned long long)value);
        ctx->refill_idle = (int)value;
        return PT_OK;
    case PT_OPT_STREAMS:
        if (value < 1 || value > (uint64_t)kMaxLanes) return fail(ctx, PT_ERR_ARG, "stream count %llu out of [1, %d]",
                                                                   (unsigned long long)value, kMaxLanes);
        ctx->nlanes = (int)value;
        return PT_OK;
    case PT_OPT_TUNE_EXTEND_BLOCKS:
        if (value < 1 || value > 8) return fail(ctx, PT_ERR_ARG, "extend blocks per CU %llu out of [1, 8]", (unsigned long long)value);
        ctx->extend_blocks = (int)value;
        return PT_OK;
    case PT_OPT_ENV_PACK: ctx->env_pack = value != 0; return PT_OK;
    case PT_OPT_TUNE_XCD_GROUP: ctx->xcd_group = value != 0; return PT_OK;
    case PT_OPT_NODE_LAYOUT:
        if (value > 1) return fail(ctx, PT_ERR_ARG, "node layout %llu unknown", (unsigned long long)value);
        ctx->node_layout = (int)value;
        return PT_OK;
    case PT_OPT_RENDER_MODE:
        if (value > 2) return fail(ctx, PT_ERR_ARG, "render mode %llu unknown", (unsigned long long)value);
        ctx->render_mode = (int)value;
        return PT_OK;
    case PT_OPT_TUNE_PATH_WAVES:
        if (value > 32) return fail(ctx, PT_ERR_ARG, "path waves per CU %llu out of [0, 32]", (unsigned long long)value);
        ctx->path_waves = (int)value;
        return PT_OK;
    case PT_OPT_TUNE_PATH_LDS_SLOTS:
        if (value < 2 || value > 64) return fail(ctx, PT_ERR_ARG, "LDS stack slots %llu out of [2, 64]", (unsigned long long)value);
        ctx->path_lds_slots = (int)value;
        return PT_OK;
    case PT_OPT_TUNE_PATH_REFILL:
        if (value < 1 || value > 64) return fail(ctx, PT_ERR_ARG, "service threshold %llu out of [1, 64]", (unsigned long long)value);
        ctx->path_refill = (int)value;
        return PT_OK;
    default: return fail(ctx, PT_ERR_ARG, "unknown option %d", option);
    }
}

int pt_scene_upload(pt_ctx *ctx, const pt_scene_desc *s)
{
    if (!ctx || !s) return PT_ERR_ARG;
    // arguments first: nothing is allocated or freed for a scene that is refused (reference src/scene.cu:132 prints
    // "Invalid material type" for a type it does not know)
    auto known_type = [](uint32_t t) { return t == PT_DIFFUSE || t == PT_MIRROR || t == PT_GLOSSY || t == PT_GLASS; };
    if (s->num_planes >= (1u << 30) || s->num_spheres >= (1u << 30)) return fail(ctx, PT_ERR_ARG, "too many primitives");
    if ((s->num_planes && !s->planes) || (s->num_spheres && !s->spheres))
        return fail(ctx, PT_ERR_ARG, "pt_scene_upload: primitive count without primitive array");
    for (uint64_t i = 0; i < s->num_planes; i++)
        if (!known_type(s->planes[i].material_type))
            return fail(ctx, PT_ERR_ARG, "plane %llu: invalid material type 0x%x", (unsigned long long)i, s->planes[i].material_type);
    for (uint64_t i = 0; i < s->num_spheres; i++)
        if (!known_type(s->spheres[i].material_type))
            return fail(ctx, PT_ERR_ARG, "sphere %llu: invalid material type 0x%x", (unsigned long long)i, s->spheres[i].material_type);
    const bool have_mesh = s->bvh_nodes && s->num_nodes > 0;
    const bool many_meshes = have_mesh && s->meshes && s->num_meshes > 0;
    if (have_mesh && !many_meshes && !known_type(s->mesh_material_type))
        return fail(ctx, PT_ERR_ARG, "mesh: invalid material type 0x%x", s->mesh_material_type);
    if (many_meshes) {
        if (!s->bvh_prim_ids) return fail(ctx, PT_ERR_ARG, "several meshes need bvh_prim_ids");
        if (s->num_meshes > 65536) return fail(ctx, PT_ERR_ARG, "more than 65536 meshes");
        for (uint64_t k = 0; k < s->num_meshes; k++) {
            if (!known_type(s->meshes[k].material_type))
                return fail(ctx, PT_ERR_ARG, "mesh %llu: invalid material type 0x%x", (unsigned long long)k, s->meshes[k].material_type);
            if (k == 0 ? s->meshes[0].first_face != 0 : s->meshes[k].first_face < s->meshes[k - 1].first_face)
                return fail(ctx, PT_ERR_ARG, "mesh %llu: first_face out of order (the first mesh starts at face 0)", (unsigned long long)k);
        }
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    // The new scene is built beside the one in place: a scene refused further down (malformed tree, tables out of range,
    // no memory) leaves the context as it was; only a complete upload replaces and frees the previous scene.
    struct Replace {
        pt_ctx *ctx;
        std::vector<void *> previous;
        bool complete = false;
        explicit Replace(pt_ctx *c) : ctx(c) { previous.swap(c->scene_allocs); }
        ~Replace()
        {
            if (!complete) ctx->scene_allocs.swap(previous);        // drop what this call allocated, keep the old scene
            for (void *p : previous) (void)hipFree(p);
        }
    } replace(ctx);
    DScene sc{};

    // ---- mesh: re-lay the reference's node + float4-run buffers into one array of 64 B records --------------------
    // records [0, N) are the inner nodes with their child references rewritten to record indices, records [N, ...)
    // the triangles; record N is a triangle nothing can hit, standing in for empty leaves.
    if (s->bvh_nodes && s->num_nodes > 0) {
        if (!s->bvh_tris || s->num_tri_float4 < 2) return fail(ctx, PT_ERR_ARG, "mesh without triangle buffer");
        const uint64_t N = s->num_nodes, M = s->num_tri_float4;
        if (N + M / 3 + 2 >= (1ull << 30)) return fail(ctx, PT_ERR_ARG, "mesh too large for 30-bit record references");
        // Pass 1, per leaf reference: length of its triangle run.  The reference's leaf loop stops at the first float4
        // whose x carries the bits 0x80000000 (src/bvh.cuh:173-181), a real triangle with v00.x == -0.0f included.
        const float *tf = s->bvh_tris;
        const int32_t *child_of = (const int32_t *)s->bvh_nodes;       // node i: words 12 and 13 of its 16
        auto bits = [](float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; };
        std::vector<uint32_t> run(2 * N, 0);            // triangles below child c of node i at [2 i + c]
        std::vector<int> bad(host_threads(N), 0);       // 1 child out of range, 2 run leaves the buffer, 3 truncated
        std::vector<uint64_t> bad_node(bad.size(), 0);
        parallel_chunks(N, bad.size(), [&](size_t part, uint64_t lo, uint64_t hi) {
            for (uint64_t i = lo; i < hi && !bad[part]; i++)
                for (int c = 0; c < 2; c++) {
                    const int32_t ch = child_of[i * 16 + 12 + c];
                    int why = 0;
                    if (ch >= 0) {
                        if ((uint64_t)ch >= N || ch == 0) why = 1;
                    } else {
                        uint64_t a = (uint64_t)(uint32_t)~ch, count = 0;
                        for (;; a += 3, count++) {
                            if (a >= M) { why = 2; break; }
                            if (bits(tf[a * 4]) == 0x80000000u) break;
                            if (a + 2 >= M) { why = 3; break; }
                        }
                        run[2 * i + c] = (uint32_t)count;
                    }
                    if (why) { bad[part] = why; bad_node[part] = i; break; }
                }
        });
        for (size_t part = 0; part < bad.size(); part++)
            if (bad[part])
                return fail(ctx, PT_ERR_ARG, "node %llu: %s", (unsigned long long)bad_node[part],
                            bad[part] == 1 ? "child out of range"
                                           : (bad[part] == 2 ? "leaf run leaves the triangle buffer" : "truncated triangle"));
        // Where every record goes: nodes first, then the record nothing can hit, then the runs in node order.  With the
        // line-aware layout the two-triangle runs come first, each on an even record index, so that both records of such
        // a leaf share one 128 B line; the other runs follow (no record is wasted either way); inner nodes are paired,
        // see node_layout_pairs.
        std::vector<uint64_t> run_at(2 * N);
        std::vector<uint32_t> newid;
        std::vector<uint64_t> holes;                      // records no reference points at
        uint64_t total_recs = N + 1;
        const uint64_t nohit_at = N;
        if (ctx->node_layout == 1) {
            if (total_recs & 1) { holes.push_back(total_recs); total_recs++; }      // at most one unused record
            for (uint64_t k = 0; k < 2 * N; k++)
                if (run[k] == 2) { run_at[k] = total_recs; total_recs += 2; }
            for (uint64_t k = 0; k < 2 * N; k++)
                if (run[k] != 2) { run_at[k] = total_recs; total_recs += run[k]; }
            if (!node_layout_pairs(s->bvh_nodes, N, newid)) return fail(ctx, PT_ERR_ARG, "node graph is not a tree");
        } else {
            for (uint64_t k = 0; k < 2 * N; k++) { run_at[k] = total_recs; total_recs += run[k]; }
            newid.resize(N);
            for (uint64_t i = 0; i < N; i++) newid[i] = (uint32_t)i;
        }
        holes.push_back(nohit_at);
        if (total_recs >= (1ull << 30)) return fail(ctx, PT_ERR_ARG, "mesh too large for 30-bit record references");
        HostArray<float4> recs(total_recs * 4);
        if (!recs.p) return fail(ctx, PT_ERR_NOMEM, "pt_scene_upload: out of host memory");
        {
            // the record nothing can hit stands in for empty leaves; the records the layout leaves unused are never
            // referenced, but defined: copies of it
            float4 nohit[4];
            nohit[0] = nohit[1] = nohit[2] = make_float4(0, 0, 0, 0);
            float4 meta = make_float4(0, 0, 0, 0);
            const uint32_t m[2] = {PT_PRIM_MISS, 3u};
            std::memcpy(&meta, m, 8);
            nohit[3] = meta;
            for (uint64_t h : holes) std::memcpy(recs.p + h * 4, nohit, 64);
        }
        std::vector<uint32_t> mesh_first;
        if (many_meshes)
            for (uint64_t k = 0; k < s->num_meshes; k++) mesh_first.push_back(s->meshes[k].first_face);
        // Pass 2: node records with their child references rewritten, triangle records with id and flags
        parallel_chunks(N, bad.size(), [&](size_t, uint64_t lo, uint64_t hi) {
            for (uint64_t i = lo; i < hi; i++) {
                float4 *node = recs.p + (size_t)newid[i] * 4;
                std::memcpy(node, s->bvh_nodes + i * 16, 64);
                int32_t ch[2];
                std::memcpy(ch, &node[3], 8);
                for (int c = 0; c < 2; c++) {
                    if (ch[c] >= 0) { ch[c] = (int32_t)newid[ch[c]]; continue; }
                    const uint64_t count = run[2 * i + c];
                    if (count == 0) { ch[c] = (int32_t)(0x80000000u | (uint32_t)nohit_at); continue; }
                    uint64_t a = (uint64_t)(uint32_t)~ch[c];
                    float4 *out = recs.p + run_at[2 * i + c] * 4;
                    for (uint64_t k = 0; k < count; k++, a += 3, out += 4) {
                        std::memcpy(out, tf + a * 4, 48);
                        float4 meta = make_float4(0, 0, 0, 0);
                        const uint32_t prim = s->bvh_prim_ids ? s->bvh_prim_ids[a] : (uint32_t)a;
                        uint32_t mesh = 0;
                        if (many_meshes)        // last mesh whose first face is not beyond this one
                            mesh = (uint32_t)(std::upper_bound(mesh_first.begin(), mesh_first.end(), prim) - mesh_first.begin()) - 1u;
                        const uint32_t m[3] = {prim, k + 1 == count ? 1u : 0u, mesh};
                        std::memcpy(&meta, m, 12);
                        out[3] = meta;
                    }
                    ch[c] = (int32_t)(0x80000000u | (uint32_t)run_at[2 * i + c]);      // sign bit: a leaf
                }
                std::memcpy(&node[3], ch, 8);
            }
        });
        // depth: a ray's traversal stack needs one slot per level plus the sentinel; the reference's own stack has 64
        // entries (src/bvh.cuh:12)
        uint32_t maxd = 0;
        {
            std::vector<std::pair<uint32_t, uint32_t>> st;
            st.push_back({0u, 1u});
            uint64_t visited = 0;
            while (!st.empty()) {
                auto [ni, dep] = st.back();
                st.pop_back();
                if (++visited > N) return fail(ctx, PT_ERR_ARG, "node graph is not a tree");
                maxd = std::max(maxd, dep);
                int32_t ch[2];
                std::memcpy(ch, &recs[(size_t)ni * 4 + 3], 8);
                for (int c = 0; c < 2; c++)
                    if (ch[c] >= 0) st.push_back({(uint32_t)ch[c], dep + 1});
            }
            if (maxd > (uint32_t)kMaxTreeDepth)
                return fail(ctx, PT_ERR_ARG, "BVH depth %u exceeds the %d-entry traversal stack", maxd, kMaxTreeDepth);
        }
        int rc;
        if ((rc = upload_raw(ctx, recs.p, total_recs * 4, &sc.recs)) != PT_OK) return rc;
        sc.stack_slots = maxd + 1;
        sc.has_bvh = 1;
        if (many_meshes) {
            std::vector<DMesh> dm(s->num_meshes);
            for (uint64_t k = 0; k < s->num_meshes; k++) {
                const pt_mesh &m = s->meshes[k];
                dm[k].mat = make_float4(m.material[0], m.material[1], m.material[2], m.material[3]);
                dm[k].type = m.material_type;
            }
            if ((rc = upload(ctx, dm, &sc.meshes)) != PT_OK) return rc;
            sc.num_meshes = (uint32_t)s->num_meshes;
            // one mesh in a list is the one-mesh case
            sc.mesh_type = dm[0].type;
            sc.mesh_mat = dm[0].mat;
        } else {
            sc.num_meshes = 1;
            sc.mesh_type = s->mesh_material_type;
            sc.mesh_mat = make_float4(s->mesh_material[0], s->mesh_material[1], s->mesh_material[2], s->mesh_material[3]);
        }
    }

    // ---- analytic primitives ------------------------------------------------------------------------------------
    std::vector<DPlane> planes(s->num_planes);
    for (uint64_t i = 0; i < s->num_planes; i++) {
        const pt_plane &p = s->planes[i];
        planes[i].pos = make_float4(p.position[0], p.position[1], p.position[2], 0);
        planes[i].nrm = make_float4(p.normal[0], p.normal[1], p.normal[2], 0);
        planes[i].mat = make_float4(p.material[0], p.material[1], p.material[2], p.material[3]);
        planes[i].type = p.material_type;
    }
    std::vector<DSphere> spheres(s->num_spheres);
    for (uint64_t i = 0; i < s->num_spheres; i++) {
        const pt_sphere &p = s->spheres[i];
        spheres[i].pos_r = make_float4(p.position[0], p.position[1], p.position[2], p.radius);
        spheres[i].mat = make_float4(p.material[0], p.material[1], p.material[2], p.material[3]);
        spheres[i].type = p.material_type;
    }
    int rc;
    if ((rc = upload(ctx, planes, &sc.planes)) != PT_OK) return rc;
    if ((rc = upload(ctx, spheres, &sc.spheres)) != PT_OK) return rc;
    sc.num_planes = (uint32_t)s->num_planes;
    sc.num_spheres = (uint32_t)s->num_spheres;

    // ---- environment ----------------------------------------------------------------------------------------------
    if (!s->env_texels || !s->env_marginal || !s->env_conditional || s->env_width == 0 || s->env_height == 0)
        return fail(ctx, PT_ERR_ARG, "scene has no environment map (the reference requires one, src/scene.cu:219)");
    if (s->env_width > 65535 || s->env_height > 65535) return fail(ctx, PT_ERR_ARG, "environment map larger than 65535");
    {
        const size_t W = s->env_width, H = s->env_height;
        std::vector<float4> tex(W * H);
        std::memcpy(tex.data(), s->env_texels, W * H * 16);
        std::vector<uint16_t> marg(W), cond(W * H);
        for (size_t i = 0; i < W; i++) {
            if (s->env_marginal[i] >= W) return fail(ctx, PT_ERR_ARG, "env_marginal[%zu] out of range", i);
            marg[i] = (uint16_t)s->env_marginal[i];
        }
        for (size_t i = 0; i < W * H; i++) {
            if (s->env_conditional[i] >= H) return fail(ctx, PT_ERR_ARG, "env_conditional[%zu] out of range", i);
            cond[i] = (uint16_t)s->env_conditional[i];
        }
        // shared-exponent packing of the texels, kept only if it reproduces every texel bit for bit
        const size_t TX = (W + 3) / 4, TY = (H + 3) / 4;
        std::vector<uint2> packed(TX * TY * 16, make_uint2(0u, 0u));
        bool exact = true;
        for (size_t y = 0; y < H && exact; y++)
            for (size_t x = 0; x < W; x++) {
                const float4 c = tex[y * W + x];
                uint32_t word;
                if (!pack_shared_exponent(c, word)) { exact = false; break; }
                uint32_t pdf_bits;
                std::memcpy(&pdf_bits, &c.w, 4);
                packed[((y >> 2) * TX + (x >> 2)) * 16 + (y & 3) * 4 + (x & 3)] = make_uint2(word, pdf_bits);
            }
        if (exact && ctx->env_pack) {
            if ((rc = upload(ctx, packed, &sc.envp)) != PT_OK) return rc;
            sc.env_tiles_x = (uint32_t)TX;
        } else {
            if ((rc = upload(ctx, tex, &sc.env)) != PT_OK) return rc;
        }
        if ((rc = upload(ctx, marg, &sc.marg)) != PT_OK) return rc;
        if ((rc = upload(ctx, cond, &sc.cond)) != PT_OK) return rc;
        sc.envW = s->env_width;
        sc.envH = s->env_height;
    }
    sc.cam = s->camera;
    ctx->sc = sc;
    ctx->have_scene = true;
    ctx->auto_mode = 0;
    replace.complete = true;
    return PT_OK;
}

int pt_intersect_device(pt_ctx *ctx, uint64_t n, const void *d_org4, const void *d_dir4, void *d_hit4, void *d_t,
                        pt_stats *stats)
{
    if (!ctx) return PT_ERR_ARG;
    if (!ctx->have_scene) return fail(ctx, PT_ERR_STATE, "pt_intersect_device: no scene uploaded");
    if (n == 0) return PT_OK;
    // the kernels walk the ray buffer with 32-bit cursors that step by up to 2^19 (grid stride) past the last ray
    if (n > kMaxRaysPerCall) return fail(ctx, PT_ERR_ARG, "pt_intersect_device: more than 2^32 - 2^20 rays in one call");
    if (!d_org4 || !d_dir4 || !d_hit4 || !d_t) return fail(ctx, PT_ERR_ARG, "pt_intersect_device: NULL buffer");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (ctx->tmp_hid_n < n) {
        if (ctx->tmp_hid) { (void)hipFree(ctx->tmp_hid); ctx->tmp_hid = nullptr; ctx->tmp_hid_n = 0; }
        HIPCHK(ctx, hipMalloc(&ctx->tmp_hid, n * sizeof(uint32_t)));
        ctx->tmp_hid_n = n;
    }
    HIPCHK(ctx, hipMemsetAsync(ctx->cnt, 0, sizeof(Counters), ctx->stream));
    HIPCHK(ctx, hipEventRecord(ctx->ev_begin, ctx->stream));
    int rc = launch_extend(ctx, ctx->stream, ctx->cnt, false, &ctx->gstk, &ctx->gstk_threads, (const float4 *)d_org4,
                           (const float4 *)d_dir4, (uint32_t)n, (uint32_t)n, ctx->tmp_hid, (float *)d_t);
    if (rc != PT_OK) return rc;
    HIPCHK(ctx, hipEventRecord(ctx->ev_end, ctx->stream));
    hipLaunchKernelGGL(k_hit_expand, dim3(stream_grid(ctx, n)), dim3(kBlock), 0, ctx->stream, ctx->sc,
                       (const float4 *)d_org4, (const float4 *)d_dir4, ctx->tmp_hid, (const float *)d_t, (uint32_t)n,
                       (float4 *)d_hit4);
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_cnt, ctx->cnt, sizeof(Counters), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (stats) {
        std::memset(stats, 0, sizeof *stats);
        float ms = 0;
        HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev_begin, ctx->ev_end));
        stats->rays = n; stats->iterations = 1; stats->extend_launches = 1;
        stats->ms_total = ms; stats->ms_extend = ms; stats->ms_extend_busy = ms;
        stats->nodes = ctx->h_cnt->trav[0]; stats->tris = ctx->h_cnt->trav[1]; stats->leaves = ctx->h_cnt->trav[2];
    }
    return PT_OK;
}

int pt_intersect_device_noalloc(pt_ctx *ctx, uint64_t n, const void *d_org4, const void *d_dir4, void *d_hit4, void *d_t,
                                pt_stats *stats)
{
    if (!ctx) return PT_ERR_ARG;
    if (!ctx->have_scene) return fail(ctx, PT_ERR_STATE, "pt_intersect_device_noalloc: no scene uploaded");
    if (n == 0) return PT_OK;
    if (n > kNoAllocMaxRays)
        return fail(ctx, PT_ERR_ARG, "pt_intersect_device_noalloc: %llu rays, the static scratch holds %llu per call",
                    (unsigned long long)n, (unsigned long long)kNoAllocMaxRays);
    if (!d_org4 || !d_dir4 || !d_hit4 || !d_t) return fail(ctx, PT_ERR_ARG, "pt_intersect_device_noalloc: NULL buffer");
    if (ctx->sc.has_bvh && ctx->sc.stack_slots > (uint32_t)kNoAllocMaxStackSlots)
        return fail(ctx, PT_ERR_ARG, "pt_intersect_device_noalloc: the tree needs %u stack slots per ray, LDS alone holds %d",
                    ctx->sc.stack_slots, kNoAllocMaxStackSlots);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // the device's one scratch array: counters first, then the hit ids.  It keeps the last call's contents; the
    // counters must start at zero, every hit id the call reads is written by the traversal first.
    Counters *cnt = (Counters *)ctx->scratch;
    uint32_t *hid = (uint32_t *)(ctx->scratch + kScratchHeader);
    // stats == NULL: the two kernels are queued on the context's stream and the call returns; nothing is copied back,
    // nothing is waited for (the form a caller under stream capture needs).  The traversal counters then are not reset
    // either: nobody reads them.
    if (stats) {
        HIPCHK(ctx, hipMemsetAsync(cnt, 0, kScratchHeader, ctx->stream));
        HIPCHK(ctx, hipEventRecord(ctx->ev_begin, ctx->stream));
    }
    int rc = launch_extend(ctx, ctx->stream, cnt, false, nullptr, nullptr, (const float4 *)d_org4, (const float4 *)d_dir4,
                           (uint32_t)n, (uint32_t)n, hid, (float *)d_t);
    if (rc != PT_OK) return rc;
    if (stats) HIPCHK(ctx, hipEventRecord(ctx->ev_end, ctx->stream));
    hipLaunchKernelGGL(k_hit_expand, dim3(stream_grid(ctx, n)), dim3(kBlock), 0, ctx->stream, ctx->sc,
                       (const float4 *)d_org4, (const float4 *)d_dir4, hid, (const float *)d_t, (uint32_t)n,
                       (float4 *)d_hit4);
    HIPCHK(ctx, hipGetLastError());
    if (!stats) return PT_OK;
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_cnt, cnt, sizeof(Counters), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    std::memset(stats, 0, sizeof *stats);
    float ms = 0;
    HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev_begin, ctx->ev_end));
    stats->rays = n; stats->iterations = 1; stats->extend_launches = 1;
    stats->ms_total = ms; stats->ms_extend = ms; stats->ms_extend_busy = ms;
    stats->nodes = ctx->h_cnt->trav[0]; stats->tris = ctx->h_cnt->trav[1]; stats->leaves = ctx->h_cnt->trav[2];
    return PT_OK;
}

int pt_intersect(pt_ctx *ctx, uint64_t n, const float *org, const float *dir, float *t_out, uint32_t *prim_out,
                 float *normal_out)
{
    if (!ctx) return PT_ERR_ARG;
    if (!ctx->have_scene) return fail(ctx, PT_ERR_STATE, "pt_intersect: no scene uploaded");
    if (n == 0) return PT_OK;
    if (!org || !dir || !t_out || !prim_out) return fail(ctx, PT_ERR_ARG, "pt_intersect: NULL buffer");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // rays go through the device in chunks of 2^20; the staging buffers belong to the context and are kept between calls
    const uint64_t chunk = 1u << 20;
    const uint64_t cap = std::min(n, chunk);
    if (ctx->stage_cap < cap) {
        for (void *p : {(void *)ctx->stage_o, (void *)ctx->stage_d, (void *)ctx->stage_h, (void *)ctx->stage_t, (void *)ctx->stage_id})
            (void)hipFree(p);
        ctx->stage_o = ctx->stage_d = ctx->stage_h = nullptr;
        ctx->stage_t = nullptr;
        ctx->stage_id = nullptr;
        ctx->stage_cap = 0;
        HIPCHK(ctx, hipMalloc(&ctx->stage_o, cap * sizeof(float4)));
        HIPCHK(ctx, hipMalloc(&ctx->stage_d, cap * sizeof(float4)));
        HIPCHK(ctx, hipMalloc(&ctx->stage_h, cap * sizeof(float4)));
        HIPCHK(ctx, hipMalloc(&ctx->stage_t, cap * sizeof(float)));
        HIPCHK(ctx, hipMalloc(&ctx->stage_id, cap * sizeof(uint32_t)));
        ctx->stage_cap = cap;
    }
    float4 *d_o = ctx->stage_o, *d_d = ctx->stage_d, *d_h = ctx->stage_h;
    float *d_t = ctx->stage_t;
    uint32_t *d_id = ctx->stage_id;
    std::vector<float4> ho(cap), hd(cap), hh(cap);
    std::vector<float> ht(cap);
    for (uint64_t base = 0; base < n; base += chunk) {
        const uint64_t m = std::min(chunk, n - base);
        for (uint64_t i = 0; i < m; i++) {
            const float *po = org + 3 * (base + i), *pd = dir + 3 * (base + i);
            ho[i] = make_float4(po[0], po[1], po[2], 0);
            hd[i] = make_float4(pd[0], pd[1], pd[2], 0);
        }
        HIPCHK(ctx, hipMemcpyAsync(d_o, ho.data(), m * sizeof(float4), hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(d_d, hd.data(), m * sizeof(float4), hipMemcpyHostToDevice, ctx->stream));
        const int rc = launch_extend(ctx, ctx->stream, ctx->cnt, false, &ctx->gstk, &ctx->gstk_threads, d_o, d_d, (uint32_t)m,
                                     (uint32_t)m, d_id, d_t);
        if (rc != PT_OK) return rc;
        hipLaunchKernelGGL(k_hit_expand, dim3(stream_grid(ctx, m)), dim3(kBlock), 0, ctx->stream, ctx->sc, d_o, d_d, d_id, d_t,
                           (uint32_t)m, d_h);
        HIPCHK(ctx, hipGetLastError());
        HIPCHK(ctx, hipMemcpyAsync(hh.data(), d_h, m * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(ht.data(), d_t, m * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        for (uint64_t i = 0; i < m; i++) {
            t_out[base + i] = ht[i];
            uint32_t p;
            std::memcpy(&p, &hh[i].w, 4);
            prim_out[base + i] = p;
            if (normal_out) {
                normal_out[3 * (base + i)] = hh[i].x;
                normal_out[3 * (base + i) + 1] = hh[i].y;
                normal_out[3 * (base + i) + 2] = hh[i].z;
            }
        }
    }
    return PT_OK;
}

int pt_primary_rays(pt_ctx *ctx, uint32_t W, uint32_t H, uint64_t sample, uint64_t seed, float *org_out, float *dir_out)
{
    if (!ctx) return PT_ERR_ARG;
    if (!ctx->have_scene) return fail(ctx, PT_ERR_STATE, "pt_primary_rays: no scene uploaded");
    if (W < 2 || H < 2 || (uint64_t)W * H >= (1ull << 31) || !org_out || !dir_out || sample >= (1ull << 27))
        return fail(ctx, PT_ERR_ARG, "pt_primary_rays: bad argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t n = (size_t)W * H;
    float4 *d_o = nullptr, *d_d = nullptr;
    HIPCHK(ctx, hipMalloc(&d_o, n * sizeof(float4)));
    hipError_t e = hipMalloc(&d_d, n * sizeof(float4));
    if (e != hipSuccess) { (void)hipFree(d_o); return fail(ctx, PT_ERR_HIP, "hipMalloc: %s", hipGetErrorString(e)); }
    hipLaunchKernelGGL(k_primary, dim3(stream_grid(ctx, n)), dim3(kBlock), 0, ctx->stream, ctx->sc, W, H,
                       make_fastdiv(pixel_divisor(W, H)), (uint32_t)sample, (uint32_t)seed, (uint32_t)(seed >> 32), d_o, d_d);
    std::vector<float4> ho(n), hd(n);
    e = hipMemcpyAsync(ho.data(), d_o, n * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(hd.data(), d_d, n * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_o);
    (void)hipFree(d_d);
    if (e != hipSuccess) return fail(ctx, PT_ERR_HIP, "pt_primary_rays: %s", hipGetErrorString(e));
    for (size_t i = 0; i < n; i++) {
        org_out[3 * i] = ho[i].x; org_out[3 * i + 1] = ho[i].y; org_out[3 * i + 2] = ho[i].z;
        dir_out[3 * i] = hd[i].x; dir_out[3 * i + 1] = hd[i].y; dir_out[3 * i + 2] = hd[i].z;
    }
    return PT_OK;
}

int pt_math_eval(pt_ctx *ctx, int which, uint64_t n, const float *a, const float *b, float *out0, float *out1)
{
    if (!ctx) return PT_ERR_ARG;
    if (which < 0 || which > 3 || n > (1ull << 28) || (n && (!a || !out0)) || (which == 1 && n && !b) || (which == 0 && n && !out1))
        return fail(ctx, PT_ERR_ARG, "pt_math_eval: bad argument");
    if (n == 0) return PT_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    float *d = nullptr;                                   // a, b, out0, out1 back to back
    HIPCHK(ctx, hipMalloc(&d, 4 * n * sizeof(float)));
    hipError_t e = hipMemcpyAsync(d, a, n * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d + n, which == 1 ? b : a, n * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_math_eval, dim3(stream_grid(ctx, n)), dim3(kBlock), 0, ctx->stream, which, (uint32_t)n, d, d + n,
                           d + 2 * n, d + 3 * n);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out0, d + 2 * n, n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && which == 0) e = hipMemcpyAsync(out1, d + 3 * n, n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(ctx, PT_ERR_HIP, "pt_math_eval: %s", hipGetErrorString(e));
    return PT_OK;
}

int pt_film_bind(pt_ctx *ctx, uint32_t W, uint32_t H, void *device_film)
{
    if (!ctx) return PT_ERR_ARG;
    if (W < 2 || H < 2 || (uint64_t)W * H >= (1ull << 31)) return fail(ctx, PT_ERR_ARG, "film %ux%u not supported", W, H);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->film_owned) { (void)hipFree(ctx->film); }
    ctx->film = nullptr;
    ctx->film_owned = false;
    ctx->W = W; ctx->H = H;
    ctx->auto_mode = 0;
    if (device_film) {
        ctx->film = (float *)device_film;
    } else {
        HIPCHK(ctx, hipMalloc(&ctx->film, (size_t)W * H * 3 * sizeof(float)));
        ctx->film_owned = true;
        HIPCHK(ctx, hipMemsetAsync(ctx->film, 0, (size_t)W * H * 3 * sizeof(float), ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    return PT_OK;
}

int pt_film_clear(pt_ctx *ctx)
{
    if (!ctx) return PT_ERR_ARG;
    if (!ctx->film) return fail(ctx, PT_ERR_STATE, "pt_film_clear: no film bound");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipMemsetAsync(ctx->film, 0, (size_t)ctx->W * ctx->H * 3 * sizeof(float), ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return PT_OK;
}

int pt_film_download(pt_ctx *ctx, float *rgb_out)
{
    if (!ctx || !rgb_out) return PT_ERR_ARG;
    if (!ctx->film) return fail(ctx, PT_ERR_STATE, "pt_film_download: no film bound");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipMemcpyAsync(rgb_out, ctx->film, (size_t)ctx->W * ctx->H * 3 * sizeof(float), hipMemcpyDeviceToHost,
                               ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return PT_OK;
}

int pt_film_tonemap16(pt_ctx *ctx, float exposure, float gamma, uint16_t *rgb_out)
{
    if (!ctx || !rgb_out) return PT_ERR_ARG;
    if (!ctx->film) return fail(ctx, PT_ERR_STATE, "pt_film_tonemap16: no film bound");
    if (!(gamma > 0.0f)) return fail(ctx, PT_ERR_ARG, "pt_film_tonemap16: gamma must be positive");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const uint64_t n = (uint64_t)ctx->W * ctx->H * 3;
    uint16_t *d16 = nullptr;
    HIPCHK(ctx, hipMalloc(&d16, n * sizeof(uint16_t)));
    hipLaunchKernelGGL(k_tonemap16, dim3(stream_grid(ctx, n)), dim3(kBlock), 0, ctx->stream, ctx->film, n, exposure,
                       1.0f / gamma, d16);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(rgb_out, d16, n * sizeof(uint16_t), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d16);
    if (e != hipSuccess) return fail(ctx, PT_ERR_HIP, "pt_film_tonemap16: %s", hipGetErrorString(e));
    return PT_OK;
}

int pt_film_reduce(pt_ctx *const *ctxs, int n, int root)
{
    if (!ctxs || n < 1 || root < 0 || root >= n || !ctxs[root]) return PT_ERR_ARG;
    pt_ctx *dst = ctxs[root];
    if (!dst->film) return fail(dst, PT_ERR_STATE, "pt_film_reduce: no film bound");
    bool distinct_devices = true;
    for (int i = 0; i < n; i++) {
        if (!ctxs[i] || !ctxs[i]->film) return fail(dst, PT_ERR_STATE, "pt_film_reduce: context %d has no film", i);
        if (ctxs[i]->W != dst->W || ctxs[i]->H != dst->H)
            return fail(dst, PT_ERR_ARG, "pt_film_reduce: context %d has a %ux%u film, the root %ux%u", i, ctxs[i]->W,
                        ctxs[i]->H, dst->W, dst->H);
        for (int j = 0; j < i; j++) {
            if (ctxs[j] == ctxs[i]) return fail(dst, PT_ERR_ARG, "pt_film_reduce: context %d is listed twice", i);
            if (ctxs[j]->device == ctxs[i]->device) distinct_devices = false;
        }
    }
    const uint64_t count = (uint64_t)dst->W * dst->H * 3;
    // every film must be complete before it is read from another stream
    for (int i = 0; i < n; i++) {
        HIPCHK(dst, hipSetDevice(ctxs[i]->device));
        HIPCHK(dst, hipStreamSynchronize(ctxs[i]->stream));
    }
    HIPCHK(dst, hipSetDevice(dst->device));
    // PT_FILM_REDUCE_RCCL=1 sends even a single context through RCCL (a one-rank reduce, in place): a way to exercise
    // the binding on a machine with one GPU
    if (n == 1) {
        const char *force = std::getenv("PT_FILM_REDUCE_RCCL");
        if (!(force && force[0] == '1')) return PT_OK;
    }
    // One context per GPU: one RCCL reduce over xGMI.  (RCCL takes one rank per device, so contexts that share a device
    // -- two contexts on one GPU, as the single-GPU tests use -- are summed by copies and adds on the root instead.)
    if (distinct_devices) return film_reduce_rccl(ctxs, n, root, count);
    float *staging = nullptr;
    HIPCHK(dst, hipMalloc(&staging, count * sizeof(float)));
    hipError_t e = hipSuccess;
    for (int i = 0; i < n && e == hipSuccess; i++) {
        if (i == root) continue;
        // stream order keeps the one staging buffer safe: copy i+1 starts after add i
        e = hipMemcpyPeerAsync(staging, dst->device, ctxs[i]->film, ctxs[i]->device, count * sizeof(float), dst->stream);
        if (e != hipSuccess) break;
        hipLaunchKernelGGL(k_film_add, dim3(stream_grid(dst, count)), dim3(kBlock), 0, dst->stream, dst->film, staging, count);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(dst->stream);
    (void)hipFree(staging);
    if (e != hipSuccess) return fail(dst, PT_ERR_HIP, "pt_film_reduce: %s", hipGetErrorString(e));
    return PT_OK;
}

int pt_render(pt_ctx *ctx, uint64_t spp_begin, uint64_t spp_count, uint64_t spp_total, uint64_t seed, pt_stats *stats)
{
    if (!ctx) return PT_ERR_ARG;
    if (!ctx->have_scene) return fail(ctx, PT_ERR_STATE, "pt_render: no scene uploaded");
    if (!ctx->film) return fail(ctx, PT_ERR_STATE, "pt_render: no film bound");
    if (spp_total == 0 || spp_begin + spp_count > spp_total || spp_total >= (1ull << 27))
        return fail(ctx, PT_ERR_ARG, "pt_render: sample range [%llu,+%llu) of %llu not supported",
                    (unsigned long long)spp_begin, (unsigned long long)spp_count, (unsigned long long)spp_total);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const uint32_t W = ctx->W, H = ctx->H;
    const uint32_t npix = W * H;
    const uint32_t s0 = (uint32_t)seed, s1 = (uint32_t)(seed >> 32);
    hipStream_t st = ctx->stream;

    // sample slab: as many samples per pixel as the budget allows; the samples of a call are rendered slab by slab
    const uint64_t per_sample = (uint64_t)npix * sizeof(float4);
    uint64_t slab_samples = std::max<uint64_t>(1, ctx->slab_budget / per_sample);
    slab_samples = std::min<uint64_t>(slab_samples, std::max<uint64_t>(1, spp_count));
    if (ctx->slab_bytes < slab_samples * per_sample) {
        if (ctx->slab) { (void)hipFree(ctx->slab); ctx->slab = nullptr; ctx->slab_bytes = 0; }
        // the budget is a wish: when the device cannot give that much (other users of its memory), halve the slab until
        // it fits; the call then simply runs more slabs
        for (;;) {
            const hipError_t e = hipMalloc(&ctx->slab, slab_samples * per_sample);
            if (e == hipSuccess) break;
            (void)hipGetLastError();
            ctx->slab = nullptr;
            if (e != hipErrorOutOfMemory || slab_samples == 1)
                return fail(ctx, PT_ERR_HIP, "pt_render: sample slab of %llu bytes: %s",
                            (unsigned long long)(slab_samples * per_sample), hipGetErrorString(e));
            slab_samples = (slab_samples + 1) / 2;
        }
        ctx->slab_bytes = slab_samples * per_sample;
    }

    RenderAcc acc;
    HIPCHK(ctx, hipEventRecord(ctx->ev_begin, st));
    HIPCHK(ctx, hipEventRecord(ctx->ev_slab_free, st));

    for (uint64_t done_samples = 0; done_samples < spp_count; done_samples += slab_samples) {
        const uint64_t first = spp_begin + done_samples;
        const uint64_t count = std::min<uint64_t>(slab_samples, spp_count - done_samples);
        uint64_t done_here = 0;
        int mode = ctx->render_mode ? ctx->render_mode : ctx->auto_mode;
        if (mode == 0) {
            // Not decided yet for this scene and film: the slab's first sample is rendered by the whole-path kernel (work that
            // has to be done anyway; its slab entries are the ones either mode writes), and the share of its rays that were
            // settled where they were made decides the mode for everything after it.
            double settled = 1.0;
            int rc = paths_run(ctx, first, 1, ctx->slab, s0, s1, acc, &settled);
            if (rc != PT_OK) return drain_lanes(ctx, rc);
            ctx->auto_mode = settled < kAutoSettledShare ? 2 : 1;
            ctx->auto_settled = settled;
            mode = ctx->auto_mode;
            done_here = 1;
        }
        int nl = 0;
        if (done_here < count) {
            float4 *rows = ctx->slab + (size_t)done_here * npix;
            int rc = mode == 2 ? paths_run(ctx, first + done_here, count - done_here, rows, s0, s1, acc, nullptr)
                               : wavefront_run(ctx, first + done_here, count - done_here, rows, s0, s1, &nl);
            if (rc != PT_OK) return drain_lanes(ctx, rc);
        }
        // every path of this slab has ended and its entry holds what it contributes: fold the slab into the film
        hipLaunchKernelGGL(k_film_reduce, dim3(stream_grid(ctx, npix)), dim3(kBlock), 0, st, ctx->slab, (uint32_t)count, npix,
                           (float)spp_total, ctx->film);
        HIPCHK(ctx, hipEventRecord(ctx->ev_slab_free, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
        HIPCHK(ctx, hipGetLastError());
        int rc = wavefront_collect(ctx, nl, acc);
        if (rc != PT_OK) return rc;
    }
    HIPCHK(ctx, hipEventRecord(ctx->ev_end, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    ctx->last_mode = ctx->render_mode ? ctx->render_mode : ctx->auto_mode;
    if (stats) {
        float ms = 0;
        HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev_begin, ctx->ev_end));
        acc.st.ms_total = ms;
        acc.st.ms_extend = acc.ext_ms;
        // length of the union of the traversal launches' time spans
        std::sort(acc.ext_spans.begin(), acc.ext_spans.end());
        double busy = 0, cur_b = 0, cur_e = -1;
        for (const auto &sp : acc.ext_spans) {
            if (sp.first > cur_e) {
                if (cur_e > cur_b) busy += cur_e - cur_b;
                cur_b = sp.first;
                cur_e = sp.second;
            } else {
                cur_e = std::max(cur_e, sp.second);
            }
        }
        if (cur_e > cur_b) busy += cur_e - cur_b;
        acc.st.ms_extend_busy = busy;
        *stats = acc.st;
    }
    return PT_OK;
}

int pt_render_mode_used(const pt_ctx *ctx) { return ctx ? ctx->last_mode : PT_ERR_ARG; }

}   // extern "C"
