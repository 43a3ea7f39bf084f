// ptcore.hip — C ABI of include/ptcore.h over the kernels in kernel.h.
// gfx950 only; no CUDA-compat headers, no dual paths.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h> // types and prototypes only: librccl.so is opened on first use (pt_group_create)

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

#include "kernel.h"
#include "launch.h"
#include "layout.h"
#include "ptcore.h"

using namespace ptc;
using namespace ptc::exact; // this file holds the exact-arithmetic kernels; kernels_fast.hip the tolerance ones

namespace {

thread_local char g_error[1024] = "";

int fail(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
    return 1;
}

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) return fail("HIP error: %s (%s)", hipGetErrorString(e_), #expr); \
    } while (0)

template <typename T> struct DeviceArray {
    T *ptr = nullptr;
    size_t count = 0;
    size_t bytes() const { return count * sizeof(T); }
    void release() {
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        count = 0;
    }
    // (count is recorded only once the allocation has succeeded: a failed call leaves an empty array,
    // so the next call with the same size allocates again instead of trusting a null pointer)
    int resize(size_t n) {
        release();
        size_t sz = (n ? n : 1) * sizeof(T); // never a null device pointer
        HIP_TRY(hipMalloc((void **)&ptr, sz));
        count = n;
        return 0;
    }
    int upload(const T *src, size_t n) {
        if (resize(n)) return 1;
        if (n) HIP_TRY(hipMemcpy(ptr, src, n * sizeof(T), hipMemcpyHostToDevice));
        return 0;
    }
};

// ---- XORWOW jump tables (host) ------------------------------------------------------
// A = one xorshift step on the 160-bit state; table k = A^(2^67 * 2^k), stored as
// [bit][8] words (5 used): the image of each input bit. Built by repeated squaring.

void xorwow_step(uint32_t v[5]) {
    uint32_t t = v[0] ^ (v[0] >> 2);
    v[0] = v[1]; v[1] = v[2]; v[2] = v[3]; v[3] = v[4];
    v[4] = (v[4] ^ (v[4] << 4)) ^ (t ^ (t << 1));
}

struct BitMatrix {
    uint32_t col[160][5];
    void apply(const uint32_t in[5], uint32_t out[5]) const {
        uint32_t r[5] = {0, 0, 0, 0, 0};
        for (int b = 0; b < 160; b++)
            if (in[b >> 5] & (1u << (b & 31)))
                for (int k = 0; k < 5; k++) r[k] ^= col[b][k];
        memcpy(out, r, sizeof(r));
    }
    void square() {
        BitMatrix next;
        for (int b = 0; b < 160; b++) apply(col[b], next.col[b]);
        *this = next;
    }
};

constexpr int kJumpTables = 32; // subsequence = y*W + x < 2^32

std::vector<uint32_t> build_jump_tables() {
    BitMatrix m;
    for (int b = 0; b < 160; b++) {
        uint32_t v[5] = {0, 0, 0, 0, 0};
        v[b >> 5] = 1u << (b & 31);
        xorwow_step(v);
        memcpy(m.col[b], v, sizeof(v));
    }
    for (int s = 0; s < 67; s++) m.square();
    std::vector<uint32_t> out((size_t)kJumpTables * 160 * 8, 0u);
    for (int k = 0; k < kJumpTables; k++) {
        for (int b = 0; b < 160; b++) memcpy(&out[((size_t)k * 160 + (size_t)b) * 8], m.col[b], 5 * sizeof(uint32_t));
        m.square();
    }
    return out;
}

// ---- host float helpers with the reference's operation order (camera basis) ------------

struct H3 {
    float x, y, z;
};
H3 hsub(H3 a, H3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
H3 hcross(H3 a, H3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
float hdot(H3 a, H3 b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
H3 hscale(H3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
H3 hunit(H3 a) { return hscale(a, (float)(1.0 / (double)sqrtf(hdot(a, a)))); }
float hlum(H3 c) { return c.x * 0.2126f + c.y * 0.7152f + c.z * 0.0722f; }

// rgb_to_radiometric, src/renderer.cu:146-162 (same operation order as the kernel used to run per sample)
H3 h_rgb_to_radiometric(H3 color, float intensity) {
    float m = sqrtf(hdot(color, color));
    H3 cn{1, 1, 1};
    if ((double)m > 1e-12) cn = hscale(color, (float)(1.0 / (double)m));
    float photometric = intensity * m;
    float alpha = (float)((double)photometric / (683.0 * (double)hlum(cn)));
    return hscale(cn, alpha);
}

// Two faces on the same three points are hit at the same distance (bit for bit when the corners are in
// the same order, within an ulp otherwise), and the reference keeps the one its traversal meets first.
// The kernel repeats such a ray sequentially in the reference's order (kernel.h, kCullSlack); a scene with
// such faces would do that for every ray through them, so it is walked on the binary tree, the cheaper
// one to repeat from, without timing the other. One 64-bit hash per face over its sorted corners, sorted (on up to 16 threads, each
// taking a share of the hash range); equal hashes are compared for real.
bool has_coincident_faces(const std::vector<GShade> &shade) {
    auto corners = [](const GShade &g, float out[9]) {
        int order[3] = {0, 1, 2};
        std::sort(order, order + 3, [&](int a, int b) { return memcmp(g.pos[a], g.pos[b], 12) < 0; });
        for (int k = 0; k < 3; k++) memcpy(out + 3 * k, g.pos[order[k]], 12);
    };
    const size_t n = shade.size();
    const unsigned hw = std::thread::hardware_concurrency();
    const unsigned threads = n < 100000 ? 1u : (hw == 0 ? 4u : (hw > 16u ? 16u : hw));
    std::vector<uint64_t> hash(n);
    auto run = [&](auto &&body) { // body(t) on `threads` threads
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < threads; t++) pool.emplace_back(body, t);
        body(0u);
        for (auto &th : pool) th.join();
    };
    run([&](unsigned t) {
        for (size_t f = n * t / threads; f < n * (t + 1) / threads; f++) {
            float c[9];
            corners(shade[f], c);
            uint64_t h = 1469598103934665603ull; // FNV-1a over the 36 bytes
            const unsigned char *b = (const unsigned char *)c;
            for (int i = 0; i < 36; i++) h = (h ^ b[i]) * 1099511628211ull;
            hash[f] = h;
        }
    });
    // thread t takes the faces whose hash falls into its share of the hash range: equal hashes meet in one thread
    std::vector<char> found(threads, 0);
    run([&](unsigned t) {
        struct Key {
            uint64_t hash;
            uint32_t face;
        };
        std::vector<Key> keys;
        keys.reserve(n / threads + n / (8 * threads) + 16);
        for (size_t f = 0; f < n; f++)
            if ((unsigned)(((hash[f] >> 32) * threads) >> 32) == t) keys.push_back({hash[f], (uint32_t)f});
        std::sort(keys.begin(), keys.end(), [](const Key &a, const Key &b) { return a.hash < b.hash; });
        for (size_t i = 1; i < keys.size() && !found[t]; i++) {
            if (keys[i].hash != keys[i - 1].hash) continue;
            float a[9], b[9];
            corners(shade[keys[i].face], a);
            corners(shade[keys[i - 1].face], b);
            if (memcmp(a, b, sizeof(a)) == 0) found[t] = 1;
        }
    });
    for (char f : found)
        if (f) return true;
    return false;
}

} // namespace

struct pt_context {
    int device = 0;
    hipStream_t stream = nullptr;
    // HIP events around the last kEventRing render launches (pt_kernel_ms)
    static constexpr int kEventRing = 64;
    hipEvent_t ev_start[kEventRing] = {}, ev_stop[kEventRing] = {};
    uint64_t launches = 0;
    hipEvent_t ev_tune[2] = {};

    // host copies needed across uploads
    std::vector<GShade> h_shade;
    std::vector<uint32_t> h_light_mult; // per face: times listed in the LST

    DeviceArray<GNode> nodes;
    DeviceArray<GNode4> nodes4;
    DeviceArray<GTri> tris;
    DeviceArray<GShade> shade;
    DeviceArray<GMaterial> materials;
    DeviceArray<pt_light> lights;
    DeviceArray<pt_lst_node> lst;
    DeviceArray<GLight> glights;
    std::vector<pt_light> h_lights;
    std::vector<GMaterial> h_materials;
    DeviceArray<Texture> textures;
    std::vector<void *> texel_blocks;
    size_t texel_bytes = 0;
    DeviceArray<uint32_t> jump;
    DeviceArray<float> image;
    DeviceArray<unsigned long long> counters;
    DeviceArray<uint32_t> fault; // KernelParams::fault: the kernels' watchdog report (8 words, zero while all is well)
    DeviceArray<uint32_t> tile_cost, tile_key, tile_order; // one entry per tile (sized for 8x8 tiles)
    DeviceArray<uint32_t> tile_rng;                        // 8 words per tile
    bool order_valid = false;                    // tile_order holds a permutation for this resolution
    int tile_rows = 8;                           // render_kernel's tile is 8 x tile_rows pixels (8 or 16)
    int cus = 0;                                 // compute units of the device
    uint64_t serial = 0;                         // unique per context; with `epoch` it names an uploaded scene
    uint64_t epoch = 0;                          // bumped by every upload

    bool have_scene = false, have_bvh = false, have_lst = false, have_cfg = false;
    uint32_t face_count = 0, light_count = 0;
    int bvh_height = 0;
    int stack_need = 0;  // traversal stack entries a walk of the binary tree can need (height - 1)
    int stack_need4 = 0; // ... of the four-wide tree (with its binary fallback): picks the 32/48/64-entry kernel
    int walk_tree = 0;  // 0 = not chosen yet for this tree, 2 = binary, 4 = four-wide
    int tree_at_upload = 0; // what pt_bvh_upload decided without timing (PT_TREE, stack bound): 0 = open
    bool coincident = false; // faces on the same three points: exact arithmetic keeps to the binary tree
    const KernelLaunchers *kern = &launchers_exact(); // arithmetic variant of the kernels (pt_set_arithmetic)
    KernelParams P{};
};

namespace {

int bind(pt_context *ctx) {
    if (!ctx) return fail("null context");
    HIP_TRY(hipSetDevice(ctx->device));
    return 0;
}

// After the primary-hit probe has finished: did a wave's watchdog fire (KernelParams::fault; the render kernel carries
// none, see kernel.h)? Reports it once and clears it.
int check_fault(pt_context *ctx, const char *where) {
    uint32_t f[8];
    HIP_TRY(hipMemcpy(f, ctx->fault.ptr, sizeof(f), hipMemcpyDeviceToHost));
    if (f[0] == 0) return 0;
    uint32_t zero[8] = {0};
    (void)hipMemcpy(ctx->fault.ptr, zero, sizeof(zero), hipMemcpyHostToDevice);
    float t;
    memcpy(&t, &f[5], 4);
    return fail("%s: the watchdog of %u wavefront(s) fired (first: node reference 0x%08x, parked leaf 0x%08x, stack depth %u, best face %d, "
                "t_init %g, ray kind %d, tile %u); the image of this launch is incomplete",
                where, f[0], f[1], f[2], f[3], (int)f[4], t, (int)f[6], f[7]);
}

int ready(pt_context *ctx) {
    if (bind(ctx)) return 1;
    if (!ctx->have_scene) return fail("pt_scene_upload has not been called");
    if (!ctx->have_bvh) return fail("pt_bvh_upload has not been called");
    if (!ctx->have_lst) return fail("pt_lst_upload has not been called");
    if (!ctx->have_cfg) return fail("pt_configure has not been called");
    return 0;
}

// The tree form known without timing: what the upload fixed, and for exact arithmetic the binary tree when faces
// coincide (it keeps the reference's tie order; tolerance arithmetic keeps no tie order). 0 = time both.
int initial_tree(const pt_context *ctx) {
    if (ctx->tree_at_upload) return ctx->tree_at_upload;
    return (ctx->coincident && ctx->kern == &launchers_exact()) ? 2 : 0;
}

void set_launch(KernelParams &P, int seed, int prev, int cur, int jitter) {
    uint64_t s = (uint64_t)(int64_t)seed; // config_t::seed is int, widened to uint64_t (renderer.cu:374)
    P.seed_lo = (uint32_t)s;
    P.seed_hi = (uint32_t)(s >> 32);
    P.previous_samples = prev;
    P.current_samples = cur;
    P.jitter = jitter;
    P.tile_order = nullptr; // launch_render opts in
    P.tile_cost = nullptr;
}

// PT_LPT=0 in the environment keeps the static tile order (A/B measurements)
const bool g_lpt = []() { const char *e = getenv("PT_LPT"); return !(e && e[0] == '0'); }();

// PT_HALF_PROBE=0: time the trees at the full image size whatever it is (A/B measurements)
const bool g_half_probe = []() { const char *e = getenv("PT_HALF_PROBE"); return !(e && e[0] == '0'); }();

// samples per pixel of the two launches that time the trees (render_launches); PT_TUNE_SAMPLES overrides
const int g_tune_samples = []() { const char *e = getenv("PT_TUNE_SAMPLES"); int n = e ? atoi(e) : 2; return n >= 1 && n <= 8 ? n : 2; }();

// 8 or 16 pixel rows per tile (see pt_configure). PT_TILE_ROWS in the environment forces one.
int pick_tile_rows(int width, int height, int cus) {
    size_t tiles16 = (size_t)((width + 7) / 8) * (size_t)((height + 15) / 16);
    size_t slots = (size_t)cus * 4 * PT_MIN_WAVES;
    int rows = tiles16 >= 8 * slots ? 16 : 8;
    if (const char *e = getenv("PT_TILE_ROWS")) rows = atoi(e) == 16 ? 16 : 8;
    return rows;
}

// The per-tile scratch of a launch: device memory beyond its inputs and its image. pt_configure allocates
// one set per context; pt_render_batch_static points it into the file-scope static array instead.
struct TileScratch {
    uint32_t *rng;   // [tiles][8]: RNG state of each tile's first pixel for this launch's seed
    uint32_t *cost;  // [tiles]: ticks the tile kept its wave slot (written by the render launch)
    uint32_t *key;   // [tiles]: sort keys of launch_rank
    uint32_t *order; // [tiles]: launch order of the next launch
    int rows;        // the tile is 8 x rows pixels
    bool order_valid; // `order` holds a permutation for this scene, resolution and tile shape
};

// Tiles of render_kernel: 8 x rows pixels, one per wave.
void render_tiles(const KernelParams &P, int rows, uint32_t *ntx, uint32_t *nty) {
    *ntx = ((uint32_t)P.width + 7u) / 8u;
    *nty = ((uint32_t)P.height + (uint32_t)rows - 1u) / (uint32_t)rows;
}

// Ranks the tiles by the cost the launch before it recorded (same stream, so ordered after it).
void launch_rank(const KernelParams &P, TileScratch &ts, hipStream_t st) {
    if (!g_lpt) return;
    uint32_t ntx, nty;
    render_tiles(P, ts.rows, &ntx, &nty);
    uint32_t n = ntx * nty;
    if (n >= (1u << 24)) return; // keys hold the static position in 24 bits; larger images keep the static order
    dim3 grid((n + 255u) / 256u), block(256);
    hipLaunchKernelGGL(tile_key_kernel, grid, block, 0, st, ts.cost, ts.key, n, ntx, nty);
    // rank of every key among all keys (O(n^2) compares spread over n/64 x 16 blocks), then the scatter;
    // the cost array is free once the keys exist and holds the ranks
    (void)hipMemsetAsync(ts.cost, 0, (size_t)n * sizeof(uint32_t), st);
    hipLaunchKernelGGL(tile_rank_kernel, dim3((n + 63u) / 64u, 16), dim3(64), 0, st, ts.key, ts.cost, n);
    hipLaunchKernelGGL(tile_order_kernel, dim3((n + 63u) / 64u), dim3(64), 0, st, ts.cost, ts.order, n);
    ts.order_valid = true;
}

dim3 grid_of(const pt_context *ctx) {
    uint32_t ntx = ((uint32_t)ctx->P.width + 7u) / 8u, nty = ((uint32_t)ctx->P.height + 7u) / 8u;
    uint32_t waves = ntx * nty, per_block = kBlock / 64;
    return dim3((waves + per_block - 1) / per_block);
}

// Per-tile RNG base states for this launch's seed (ordered before the render launch on its stream).
void launch_tile_rng(KernelParams &P, const TileScratch &ts, uint32_t ntx, uint32_t nty, hipStream_t st) {
    P.tile_rng = ts.rng;
    hipLaunchKernelGGL(tile_rng_kernel, dim3((ntx * nty + kTileRngLanes - 1u) / kTileRngLanes), dim3(64), 0, st, P, ntx, nty,
                       (uint32_t)ts.rows);
}

template <bool COUNT>
void launch_render(const pt_context *ctx, KernelParams &P, const TileScratch &ts, bool wide, float *img, unsigned long long *counters,
                   hipStream_t st) {
    uint32_t ntx, nty;
    render_tiles(P, ts.rows, &ntx, &nty);
    dim3 grid(ntx * nty);
    P.tile_order = (ts.order_valid && g_lpt) ? ts.order : nullptr;
    P.tile_cost = g_lpt ? ts.cost : nullptr;
    launch_tile_rng(P, ts, ntx, nty, st);
    ctx->kern->render(COUNT, wide ? ctx->stack_need4 : ctx->stack_need, ts.rows, wide, grid, P, img, counters, st);
}

// Short launch without an image: records the tile costs (see pt_render_batch_into).
void launch_probe(const pt_context *ctx, KernelParams &P, const TileScratch &ts, hipStream_t st, bool wide) {
    uint32_t ntx, nty;
    render_tiles(P, ts.rows, &ntx, &nty);
    dim3 grid(ntx * nty);
    P.tile_order = nullptr;
    P.tile_cost = ts.cost;
    launch_tile_rng(P, ts, ntx, nty, st);
    ctx->kern->probe(wide ? ctx->stack_need4 : ctx->stack_need, ts.rows, wide, grid, P, st);
}

// The context's own tile scratch (allocated by pt_configure), and what a launch left in it.
TileScratch own_scratch(const pt_context *ctx) {
    return TileScratch{ctx->tile_rng.ptr, ctx->tile_cost.ptr, ctx->tile_key.ptr, ctx->tile_order.ptr, ctx->tile_rows, ctx->order_valid};
}

// ---- static scratch (pt_render_batch_static) ---------------------------------------------------
// One array per GPU, held until the process ends: per-tile RNG base states (8 words), cost, key and order
// (one word each) for up to PT_STATIC_MAX_TILES tiles of 8x8 pixels. In HOST code the array's name is only
// a handle for hipGetSymbolAddress: pt_create asks for its device address once per device, with that device
// current, and every use below goes through that pointer.
constexpr size_t kStaticWordsPerTile = 11;
static __device__ __align__(256) uint32_t scratch[(size_t)PT_STATIC_MAX_TILES * kStaticWordsPerTile];

constexpr int kMaxDevices = 64;
struct StaticScratch {
    uint32_t *ptr = nullptr; // device address of `scratch` on this device
    // what the last pt_render_batch_static call on this device left in the order array
    bool order_valid = false;
    uint64_t serial = 0, epoch = 0;
    int width = 0, height = 0, rows = 0;
};
StaticScratch g_static[kMaxDevices];

int static_scratch_init(int device) { // device is current
    if (device < 0 || device >= kMaxDevices) return fail("pt_create: device %d beyond the %d the static scratch table holds", device, kMaxDevices);
    if (g_static[device].ptr) return 0;
    uint32_t *ptr = nullptr;
    HIP_TRY(hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch)));
    if (!ptr) return fail("pt_create: hipGetSymbolAddress returned a null address for the static scratch");
    g_static[device].ptr = ptr;
    return 0;
}

} // namespace

extern "C" const char *pt_last_error(void) { return g_error; }

namespace {
int create_resources(pt_context *ctx) {
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipDeviceGetAttribute(&ctx->cus, hipDeviceAttributeMultiprocessorCount, ctx->device));
    {
        static std::mutex mu;
        static uint64_t next_serial = 0;
        std::lock_guard<std::mutex> lock(mu);
        ctx->serial = ++next_serial;
        if (static_scratch_init(ctx->device)) return 1; // once per device, with that device current
    }
    HIP_TRY(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreate(&ctx->ev_tune[0]));
    HIP_TRY(hipEventCreate(&ctx->ev_tune[1]));
    for (int i = 0; i < pt_context::kEventRing; i++) {
        HIP_TRY(hipEventCreate(&ctx->ev_start[i]));
        HIP_TRY(hipEventCreate(&ctx->ev_stop[i]));
    }
    std::vector<uint32_t> tables = build_jump_tables();
    if (ctx->jump.upload(tables.data(), tables.size())) return 1;
    unsigned long long zeros[32] = {0};
    if (ctx->counters.upload(zeros, 32)) return 1;
    ctx->P.jump = ctx->jump.ptr;
    uint32_t no_fault[8] = {0};
    if (ctx->fault.upload(no_fault, 8)) return 1;
    ctx->P.fault = ctx->fault.ptr;
    if (const char *e = getenv("PT_ARITHMETIC")) // default of new contexts (the CLI has no flag for it: the reference's flag surface is kept)
        if (strcmp(e, "fast") == 0) ctx->kern = &launchers_fast();
    return 0;
}
} // namespace

extern "C" int pt_create(int device, pt_context **out) {
    if (!out) return fail("pt_create: null output pointer");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) return fail("pt_create: no HIP device available (%s)", hipGetErrorString(e));
    if (device < 0 || device >= ndev) return fail("pt_create: device %d out of range (0..%d)", device, ndev - 1);
    pt_context *ctx = new pt_context();
    ctx->device = device;
    if (create_resources(ctx)) {
        pt_destroy(ctx); // keeps the message of the call that failed
        return 1;
    }
    *out = ctx;
    return 0;
}

extern "C" int pt_device_count(void) {
    int ndev = 0;
    return hipGetDeviceCount(&ndev) == hipSuccess ? ndev : 0;
}

extern "C" void pt_destroy(pt_context *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    ctx->nodes.release();
    ctx->nodes4.release();
    ctx->tris.release();
    ctx->shade.release();
    ctx->materials.release();
    ctx->lights.release();
    ctx->lst.release();
    ctx->glights.release();
    ctx->textures.release();
    for (void *p : ctx->texel_blocks) (void)hipFree(p);
    ctx->jump.release();
    ctx->image.release();
    ctx->counters.release();
    ctx->fault.release();
    ctx->tile_cost.release();
    ctx->tile_order.release();
    ctx->tile_key.release();
    ctx->tile_rng.release();
    for (int i = 0; i < 2; i++)
        if (ctx->ev_tune[i]) (void)hipEventDestroy(ctx->ev_tune[i]);
    for (int i = 0; i < pt_context::kEventRing; i++) {
        if (ctx->ev_start[i]) (void)hipEventDestroy(ctx->ev_start[i]);
        if (ctx->ev_stop[i]) (void)hipEventDestroy(ctx->ev_stop[i]);
    }
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" int pt_scene_upload(pt_context *ctx, const pt_scene_desc *scene) {
    if (bind(ctx)) return 1;
    if (!scene) return fail("pt_scene_upload: null scene");
    if (scene->face_count == 0 || !scene->faces || !scene->vertices) return fail("pt_scene_upload: scene has no faces");
    for (uint32_t f = 0; f < scene->face_count; f++) {
        const pt_face &face = scene->faces[f];
        if (face.vertexCount != 3) return fail("pt_scene_upload: face %u has %u vertices (triangles only)", f, face.vertexCount);
        for (int k = 0; k < 3; k++)
            if (face.vertices[k] >= scene->vertex_count)
                return fail("pt_scene_upload: face %u references vertex %u of %u", f, face.vertices[k], scene->vertex_count);
        if (face.material >= scene->material_count)
            return fail("pt_scene_upload: face %u references material %u of %u", f, face.material, scene->material_count);
    }
    for (uint32_t m = 0; m < scene->material_count; m++) {
        const pt_material &mat = scene->materials[m];
        int32_t t[3] = {mat.baseColorTexture, mat.normalTexture, mat.metallicRoughnessTexture};
        for (int k = 0; k < 3; k++)
            if (t[k] >= (int32_t)scene->texture_count)
                return fail("pt_scene_upload: material %u references texture %d of %u", m, t[k], scene->texture_count);
    }
    if (scene->clearTexture >= (int32_t)scene->texture_count)
        return fail("pt_scene_upload: clearTexture %d of %u", scene->clearTexture, scene->texture_count);

    // from here on the context holds no complete scene until this call has succeeded
    ctx->have_scene = ctx->have_bvh = ctx->have_lst = false;
    ctx->epoch++;
    ctx->order_valid = false;
    std::vector<GMaterial> mats;
    bake_shading(*scene, ctx->h_shade, mats);
    ctx->h_light_mult.assign(scene->face_count, 0);
    if (ctx->shade.upload(ctx->h_shade.data(), ctx->h_shade.size())) return 1;
    if (ctx->materials.upload(mats.data(), mats.size())) return 1;
    if (ctx->lights.upload(scene->lights, scene->light_count)) return 1;
    ctx->h_materials = mats;
    ctx->h_lights.assign(scene->lights, scene->lights + scene->light_count);

    for (void *p : ctx->texel_blocks) (void)hipFree(p);
    ctx->texel_blocks.clear();
    ctx->texel_bytes = 0;
    std::vector<Texture> tex(scene->texture_count);
    for (uint32_t i = 0; i < scene->texture_count; i++) {
        const pt_texture &src = scene->textures[i];
        if (src.width <= 0 || src.height <= 0 || !src.pixels || src.channels != 4)
            return fail("pt_scene_upload: texture %u is empty or not 4-channel", i);
        size_t bytes = (size_t)src.width * (size_t)src.height * 4 * (src.isFloat ? sizeof(float) : 1);
        void *d = nullptr;
        HIP_TRY(hipMalloc(&d, bytes));
        ctx->texel_blocks.push_back(d);
        ctx->texel_bytes += bytes;
        HIP_TRY(hipMemcpy(d, src.pixels, bytes, hipMemcpyHostToDevice));
        tex[i].texels = d;
        tex[i].width = src.width;
        tex[i].height = src.height;
        tex[i].isFloat = src.isFloat;
        tex[i].wrapS = src.wrapS;
        tex[i].wrapT = src.wrapT;
        tex[i].filter = src.filter;
    }
    if (ctx->textures.upload(tex.data(), tex.size())) return 1;

    // camera basis, renderer.cu:32-46 (depends only on the camera, so once per scene)
    const pt_camera &cam = scene->camera;
    H3 Pp{cam.position.x, cam.position.y, cam.position.z}, T{cam.target.x, cam.target.y, cam.target.z};
    H3 Up{cam.updir.x, cam.updir.y, cam.updir.z};
    H3 W = hsub(T, Pp);
    H3 U = hcross(W, Up);
    H3 V = hcross(U, W);
    float focal = 0.1f;
    float yheight = atanf((float)(0.5 * (double)cam.yfov)) * focal;
    U = hscale(hunit(U), yheight);
    V = hscale(hunit(V), yheight);
    W = hscale(hunit(W), focal);
    KernelParams &P = ctx->P;
    P.cam_u[0] = U.x; P.cam_u[1] = U.y; P.cam_u[2] = U.z;
    P.cam_v[0] = V.x; P.cam_v[1] = V.y; P.cam_v[2] = V.z;
    P.cam_w[0] = W.x; P.cam_w[1] = W.y; P.cam_w[2] = W.z;
    P.cam_p[0] = Pp.x; P.cam_p[1] = Pp.y; P.cam_p[2] = Pp.z;
    P.shade = ctx->shade.ptr;
    P.materials = ctx->materials.ptr;
    P.lights = ctx->lights.ptr;
    P.textures = ctx->textures.ptr;
    P.clear_texture = scene->clearTexture;
    ctx->face_count = scene->face_count;
    ctx->light_count = scene->light_count;
    ctx->have_scene = true;
    ctx->order_valid = false; // tile costs belong to one scene
    ctx->have_bvh = false; // a BVH belongs to one scene
    ctx->have_lst = false;
    return 0;
}

extern "C" int pt_bvh_upload(pt_context *ctx, const pt_bvh_node *nodes, uint32_t node_count, const uint32_t *indices,
                             uint32_t index_count) {
    if (bind(ctx)) return 1;
    if (!ctx->have_scene) return fail("pt_bvh_upload: call pt_scene_upload first");
    if (!nodes || !indices) return fail("pt_bvh_upload: null argument");
    ctx->have_bvh = false;
    ctx->epoch++;
    BakedBvh baked;
    std::string err = bake_bvh(ctx->h_shade, nodes, node_count, indices, index_count, baked);
    if (!err.empty()) return fail("pt_bvh_upload: %s", err.c_str());
    // the reference refuses trees taller than its 64-entry stack (bvh.cu:338-342)
    if (baked.height > 65) return fail("pt_bvh_upload: tree height %d exceeds the 64-entry traversal stack", baked.height);
    bake_bvh4(baked);
    // stack entries a walk may hold: height - 1 in the binary tree; in the four-wide one the folded bound
    // (every sibling hit on the whole path) plus the 2 scratch slots the wide step writes above the top
    int stack_need = baked.height - 1;
    int stack_need4 = stack_need > baked.stack4 + 2 ? stack_need : baked.stack4 + 2;
    if (stack_need > 64) return fail("pt_bvh_upload: a walk of this tree can need %d stack entries (64 available)", stack_need);
    if (ctx->nodes.upload(baked.nodes.data(), baked.nodes.size())) return 1;
    if (ctx->nodes4.upload(baked.nodes4.data(), baked.nodes4.size())) return 1;
    ctx->P.nodes4 = ctx->nodes4.ptr;
    ctx->P.root4 = baked.root4;

    ctx->stack_need = stack_need;
    ctx->stack_need4 = stack_need4;
    ctx->tree_at_upload = 0;
    if (const char *e = getenv("PT_TREE")) ctx->tree_at_upload = strcmp(e, "wide") == 0 ? 4 : (strcmp(e, "binary") == 0 ? 2 : 0);
    if (stack_need4 > 64) ctx->tree_at_upload = 2; // the folded tree could overflow the largest stack: binary only
    ctx->coincident = ctx->tree_at_upload == 0 && has_coincident_faces(ctx->h_shade);
    ctx->walk_tree = initial_tree(ctx);
    ctx->order_valid = false;
    if (ctx->tris.upload(baked.tris.data(), baked.tris.size())) return 1;
    ctx->P.nodes = ctx->nodes.ptr;
    ctx->P.tris = ctx->tris.ptr;
    ctx->P.root = baked.root;
    ctx->bvh_height = baked.height;
    ctx->have_bvh = true;
    return 0;
}

extern "C" int pt_lst_upload(pt_context *ctx, const pt_lst_node *nodes, uint32_t count) {
    if (bind(ctx)) return 1;
    if (!ctx->have_scene) return fail("pt_lst_upload: call pt_scene_upload first");
    if (count && !nodes) return fail("pt_lst_upload: null argument");
    ctx->have_lst = false;
    ctx->epoch++;
    std::vector<uint32_t> mult(ctx->face_count, 0);
    for (uint32_t i = 0; i < count; i++) {
        if (nodes[i].type == PT_LST_SOURCE_FACE) {
            if (nodes[i].index >= ctx->face_count) return fail("pt_lst_upload: node %u references face %u", i, nodes[i].index);
            mult[nodes[i].index]++;
        } else if (nodes[i].type == PT_LST_SOURCE_LIGHT) {
            if (nodes[i].index >= ctx->light_count) return fail("pt_lst_upload: node %u references light %u", i, nodes[i].index);
        } else {
            return fail("pt_lst_upload: node %u has unknown type %u", i, nodes[i].type);
        }
    }
    for (uint32_t f = 0; f < ctx->face_count; f++) {
        uint32_t flags = ctx->h_shade[f].flags & kShadeFlat;
        if (mult[f]) flags |= kShadeIsLight | (mult[f] << 8);
        ctx->h_shade[f].flags = flags;
    }
    if (ctx->shade.upload(ctx->h_shade.data(), ctx->h_shade.size())) return 1;
    ctx->P.shade = ctx->shade.ptr;
    if (ctx->lst.upload(nodes, count)) return 1;
    std::vector<GLight> gl(count);
    uint32_t face_nodes = 0;
    for (uint32_t i = 0; i < count; i++) {
        GLight g;
        memset(&g, 0, sizeof(g));
        if (nodes[i].type == PT_LST_SOURCE_LIGHT) {
            const pt_light &L = ctx->h_lights[nodes[i].index];
            H3 rad = h_rgb_to_radiometric(H3{L.color.x, L.color.y, L.color.z}, L.intensity);
            g.radiance[0] = rad.x; g.radiance[1] = rad.y; g.radiance[2] = rad.z;
            if (L.type == PT_LIGHT_POINT) {
                g.kind = 0;
                g.a[0] = L.position.x; g.a[1] = L.position.y; g.a[2] = L.position.z;
            } else {
                g.kind = 1;
                H3 d = hunit(H3{-L.direction.x, -L.direction.y, -L.direction.z}); // renderer.cu:211-212
                g.a[0] = d.x; g.a[1] = d.y; g.a[2] = d.z;
            }
        } else {
            face_nodes++;
            const GShade &sh = ctx->h_shade[nodes[i].index];
            g.kind = 2;
            g.face = nodes[i].index;
            H3 A{sh.pos[0][0], sh.pos[0][1], sh.pos[0][2]}, B{sh.pos[1][0], sh.pos[1][1], sh.pos[1][2]};
            H3 C{sh.pos[2][0], sh.pos[2][1], sh.pos[2][2]};
            memcpy(g.a, sh.pos[0], 12);
            memcpy(g.b, sh.pos[1], 12);
            memcpy(g.c, sh.pos[2], 12);
            H3 tri_cross = hcross(hsub(C, B), hsub(A, B)); // renderer.cu:102-105
            float len = sqrtf(hdot(tri_cross, tri_cross));
            H3 nrm = hscale(tri_cross, (float)(1.0 / (double)len));
            g.normal[0] = nrm.x; g.normal[1] = nrm.y; g.normal[2] = nrm.z;
            g.area = (float)(0.5 * (double)len);
            const GMaterial &m = ctx->h_materials[sh.material];
            H3 rad = h_rgb_to_radiometric(H3{m.emissive[0], m.emissive[1], m.emissive[2]}, 1.0f); // renderer.cu:242
            g.radiance[0] = rad.x; g.radiance[1] = rad.y; g.radiance[2] = rad.z;
        }
        gl[i] = g;
    }
    if (ctx->glights.upload(gl.data(), gl.size())) return 1;
    ctx->P.glights = ctx->glights.ptr;
    ctx->P.lst_face_nodes = face_nodes;
    ctx->P.lst = ctx->lst.ptr;
    ctx->P.lst_count = count;
    ctx->have_lst = true;
    return 0;
}

extern "C" int pt_configure(pt_context *ctx, const pt_render_config *cfg) {
    if (bind(ctx)) return 1;
    if (!cfg) return fail("pt_configure: null config");
    if (cfg->resolution_x <= 0 || cfg->resolution_y <= 0) return fail("pt_configure: empty image");
    if ((uint64_t)cfg->resolution_x * (uint64_t)cfg->resolution_y > 0xFFFFFFFFull)
        return fail("pt_configure: more than 2^32 pixels");
    KernelParams &P = ctx->P;
    ctx->have_cfg = false; // until every buffer below exists
    // a tile order belongs to one resolution (two sizes can share a tile count and still map tiles differently)
    if (P.width != cfg->resolution_x || P.height != cfg->resolution_y) ctx->order_valid = false;
    P.width = cfg->resolution_x;
    P.height = cfg->resolution_y;
    P.clear_color[0] = cfg->world_clear_color.x;
    P.clear_color[1] = cfg->world_clear_color.y;
    P.clear_color[2] = cfg->world_clear_color.z;
    P.exposure_scale = powf(2.0f, cfg->output_exposure); // renderer.cu:393
    size_t n = (size_t)P.width * (size_t)P.height * 3;
    if (ctx->image.count != n && ctx->image.resize(n)) return 1;
    HIP_TRY(hipMemset(ctx->image.ptr, 0, n * sizeof(float)));
    // Two pixels per lane (8x16 tiles, pixels swapped between lanes every few samples) keep the lanes of
    // a wave busy to the end of its tile, but halve the number of tiles; that pays only while the
    // launch still has many tiles per wave slot (measured: a loss at 1024^2).
    {
        int rows = pick_tile_rows(P.width, P.height, ctx->cus);
        if (rows != ctx->tile_rows) ctx->order_valid = false;
        ctx->tile_rows = rows;
    }
    size_t tiles = (size_t)((P.width + 7) / 8) * (size_t)((P.height + 7) / 8);
    if (ctx->tile_rng.count != tiles * 8 || ctx->tile_cost.count != tiles || ctx->tile_key.count != tiles || ctx->tile_order.count != tiles) {
        ctx->order_valid = false;
        if (ctx->tile_rng.resize(tiles * 8) || ctx->tile_cost.resize(tiles) || ctx->tile_key.resize(tiles) || ctx->tile_order.resize(tiles)) {
            // all four or none: a later call with the same size must allocate again, not trust a null pointer
            ctx->tile_rng.release();
            ctx->tile_cost.release();
            ctx->tile_key.release();
            ctx->tile_order.release();
            return 1;
        }
    }
    ctx->have_cfg = true;
    return 0;
}

extern "C" int pt_clear_image(pt_context *ctx) {
    if (bind(ctx)) return 1;
    if (!ctx->have_cfg) return fail("pt_clear_image: pt_configure has not been called");
    HIP_TRY(hipMemsetAsync(ctx->image.ptr, 0, ctx->image.bytes(), ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

namespace {
// One render launch of `P` with its tile scratch `ts` on `st`: the launches around it that record tile
// costs and rank them, and the HIP events of pt_kernel_ms. may_block: the first launch after an upload may
// time both trees (blocks the host for about 5 ms); otherwise an untimed choice is made.
int render_launches(pt_context *ctx, KernelParams &P, TileScratch &ts, float *img, hipStream_t st, bool may_block) {
    const int current_samples = P.current_samples;
    const int slot = (int)(ctx->launches % pt_context::kEventRing);
    HIP_TRY(hipEventRecord(ctx->ev_start[slot], st));
    if (ctx->walk_tree == 0 && current_samples >= 8 && may_block) {
        // Which tree walks this scene faster depends on the scene (four-wide: half the node visits, but
        // fewer lanes stepping together; measured both ways, -12 % .. +7 %). Both give the same hits, so
        // time a 2-sample launch without an image on each and keep the faster. Blocks the host once per
        // uploaded tree (about 5 ms).
        // A large image is timed at half its resolution: the same view through a quarter of the pixels (the camera
        // maps pixel to direction by the image size, renderer.cu:380-386), a quarter of the time, and still more
        // tiles than the GPU holds waves. Its tile costs stand in for the 2 x 2 full-size tiles each one covers.
        float ms[2] = {0.f, 0.f};
        uint32_t ntx, nty;
        render_tiles(P, ts.rows, &ntx, &nty);
        const bool half = g_half_probe && (size_t)ntx * nty >= 4 * (size_t)ctx->cus * 4 * PT_MIN_WAVES;
        KernelParams Q = P;
        TileScratch tq = ts;
        if (half) {
            Q.width = (P.width + 1) / 2;
            Q.height = (P.height + 1) / 2;
            tq.cost = ts.key; // (free until launch_rank builds the keys)
        }
        Q.current_samples = g_tune_samples;
        for (int k = 0; k < 2; k++) {
            HIP_TRY(hipEventRecord(ctx->ev_tune[0], st));
            launch_probe(ctx, Q, tq, st, k == 1);
            HIP_TRY(hipEventRecord(ctx->ev_tune[1], st));
            HIP_TRY(hipEventSynchronize(ctx->ev_tune[1]));
            HIP_TRY(hipEventElapsedTime(&ms[k], ctx->ev_tune[0], ctx->ev_tune[1]));
        }
        if (half) {
            uint32_t htx, hty;
            render_tiles(Q, ts.rows, &htx, &hty);
            hipLaunchKernelGGL(tile_expand_kernel, dim3((ntx * nty + 255u) / 256u), dim3(256), 0, st, tq.cost, ts.cost, ntx, nty, htx, hty);
        }
        ctx->walk_tree = ms[1] < ms[0] ? 4 : 2;
        if (getenv("PT_VERBOSE"))
            fprintf(stderr, "[ptcore] tree timing at %d x %d, %d sample(s): binary %.3f ms, four-wide %.3f ms\n", Q.width, Q.height, g_tune_samples, ms[0], ms[1]);
        P.current_samples = current_samples;
        if (g_lpt && !ts.order_valid) launch_rank(P, ts, st); // the second probe's tile costs order the first batch
    } else if (ctx->walk_tree == 0 && may_block) {
        ctx->walk_tree = 2;
    }
    // a call that may not block leaves the choice open: untimed, the four-wide tree (it wins on most scenes, and
    // pt_bvh_upload has already ruled it out where it cannot be used)
    const bool wide = (ctx->walk_tree == 0 ? 4 : ctx->walk_tree) == 4;
    if (g_lpt && !ts.order_valid && current_samples >= 8) {
        // No costs recorded yet for this scene/resolution: a 1-sample launch that writes no pixel
        // (null image) records them, so that even the first batch is dispatched heaviest first.
        P.current_samples = 1;
        launch_probe(ctx, P, ts, st, wide);
        launch_rank(P, ts, st);
        P.current_samples = current_samples;
    }
    launch_render<false>(ctx, P, ts, wide, img, nullptr, st);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ctx->ev_stop[slot], st));
    launch_rank(P, ts, st);
    HIP_TRY(hipGetLastError());
    ctx->launches++;
    return 0;
}
} // namespace

extern "C" int pt_render_batch_into(pt_context *ctx, float *device_img, void *stream, int seed, int previous_samples,
                                    int current_samples) {
    if (ready(ctx)) return 1;
    if (!device_img) return fail("pt_render_batch_into: null image");
    if (current_samples <= 0 || previous_samples < 0) return fail("pt_render_batch: bad sample counts (%d, %d)", previous_samples, current_samples);
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    set_launch(ctx->P, seed, previous_samples, current_samples, 1);
    TileScratch ts = own_scratch(ctx);
    int rc = render_launches(ctx, ctx->P, ts, device_img, st, true);
    ctx->order_valid = ts.order_valid;
    return rc;
}

extern "C" int pt_render_batch_static(pt_context *ctx, float *device_img, void *stream, const pt_render_config *cfg, int seed,
                                      int previous_samples, int current_samples) {
    if (bind(ctx)) return 1;
    if (!ctx->have_scene) return fail("pt_scene_upload has not been called");
    if (!ctx->have_bvh) return fail("pt_bvh_upload has not been called");
    if (!ctx->have_lst) return fail("pt_lst_upload has not been called");
    if (!device_img) return fail("pt_render_batch_static: null image");
    if (!cfg) return fail("pt_render_batch_static: null config");
    if (current_samples <= 0 || previous_samples < 0) return fail("pt_render_batch_static: bad sample counts (%d, %d)", previous_samples, current_samples);
    if (cfg->resolution_x <= 0 || cfg->resolution_y <= 0) return fail("pt_render_batch_static: empty image");
    const uint64_t tiles = (uint64_t)(((int64_t)cfg->resolution_x + 7) / 8) * (uint64_t)(((int64_t)cfg->resolution_y + 7) / 8);
    if (tiles > (uint64_t)PT_STATIC_MAX_TILES)
        return fail("pt_render_batch_static: %d x %d pixels are %llu tiles of 8x8, the static scratch holds %d (use pt_configure + "
                    "pt_render_batch_into for larger images)", cfg->resolution_x, cfg->resolution_y, (unsigned long long)tiles, PT_STATIC_MAX_TILES);
    StaticScratch &g = g_static[ctx->device];
    if (!g.ptr) return fail("pt_render_batch_static: no static scratch address for device %d", ctx->device);
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;

    // the launch parameters of this call: the scene's, with the call's own image size, exposure and clear colour
    KernelParams P = ctx->P;
    P.width = cfg->resolution_x;
    P.height = cfg->resolution_y;
    P.clear_color[0] = cfg->world_clear_color.x;
    P.clear_color[1] = cfg->world_clear_color.y;
    P.clear_color[2] = cfg->world_clear_color.z;
    P.exposure_scale = powf(2.0f, cfg->output_exposure); // renderer.cu:393
    set_launch(P, seed, previous_samples, current_samples, 1);

    // the tile arrays inside the static array: RNG states first (8 words per tile), then cost, key, order
    TileScratch ts;
    ts.rng = g.ptr;
    ts.cost = g.ptr + (size_t)PT_STATIC_MAX_TILES * 8;
    ts.key = ts.cost + PT_STATIC_MAX_TILES;
    ts.order = ts.key + PT_STATIC_MAX_TILES;
    ts.rows = pick_tile_rows(P.width, P.height, ctx->cus);
    // Every call on this GPU shares the array, which keeps the last call's contents. The order array is the only
    // part read before it is written: it counts only if the last call left it for this very scene, size and
    // tile shape. (Costs are written by every launch before they are read; keys likewise; the ranking clears
    // the cost array itself before it accumulates ranks into it.)
    ts.order_valid = g.order_valid && g.serial == ctx->serial && g.epoch == ctx->epoch && g.width == P.width && g.height == P.height &&
                     g.rows == ts.rows;
    g.order_valid = false; // until this call's launches are in the stream
    int rc = render_launches(ctx, P, ts, device_img, st, false);
    if (rc == 0) {
        g.order_valid = ts.order_valid;
        g.serial = ctx->serial;
        g.epoch = ctx->epoch;
        g.width = P.width;
        g.height = P.height;
        g.rows = ts.rows;
    }
    return rc;
}

extern "C" int pt_render_batch(pt_context *ctx, int seed, int previous_samples, int current_samples) {
    if (ready(ctx)) return 1;
    if (pt_render_batch_into(ctx, ctx->image.ptr, nullptr, seed, previous_samples, current_samples)) return 1;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

extern "C" int pt_synchronize(pt_context *ctx) {
    if (bind(ctx)) return 1;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

extern "C" float pt_kernel_ms(pt_context *ctx, int launches_back) {
    if (!ctx || launches_back < 0 || launches_back >= pt_context::kEventRing || (uint64_t)launches_back >= ctx->launches) return -1.f;
    if (hipSetDevice(ctx->device) != hipSuccess) return -1.f;
    const int slot = (int)((ctx->launches - 1 - (uint64_t)launches_back) % pt_context::kEventRing);
    if (hipEventSynchronize(ctx->ev_stop[slot]) != hipSuccess) return -1.f;
    float ms = -1.f;
    if (hipEventElapsedTime(&ms, ctx->ev_start[slot], ctx->ev_stop[slot]) != hipSuccess) return -1.f;
    return ms;
}

extern "C" float pt_last_kernel_ms(pt_context *ctx) { return pt_kernel_ms(ctx, 0); }

extern "C" int pt_read_image(pt_context *ctx, float *host_rgb) {
    if (bind(ctx)) return 1;
    if (!ctx->have_cfg) return fail("pt_read_image: pt_configure has not been called");
    if (!host_rgb) return fail("pt_read_image: null buffer");
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipMemcpy(host_rgb, ctx->image.ptr, ctx->image.bytes(), hipMemcpyDeviceToHost));
    return 0;
}

extern "C" void *pt_host_alloc(size_t bytes) {
    void *p = nullptr;
    hipError_t e = hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) {
        fail("pt_host_alloc: %s (%zu bytes)", hipGetErrorString(e), bytes);
        return nullptr;
    }
    return p;
}

extern "C" void pt_host_free(void *ptr) {
    if (ptr) (void)hipHostFree(ptr);
}

extern "C" int pt_rng_stream(pt_context *ctx, uint64_t seed, uint64_t subsequence, uint32_t *host_out, int n) {
    if (bind(ctx)) return 1;
    if (!host_out || n <= 0) return fail("pt_rng_stream: bad arguments");
    if (subsequence > 0xFFFFFFFFull) return fail("pt_rng_stream: subsequence beyond 2^32 (pixel indices are 32-bit)");
    uint32_t *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, (size_t)n * sizeof(uint32_t)));
    hipLaunchKernelGGL(rng_stream_kernel, dim3(1), dim3(64), 0, ctx->stream, ctx->jump.ptr, (uint32_t)seed,
                       (uint32_t)(seed >> 32), (uint32_t)subsequence, d, n);
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) e = hipMemcpy(host_out, d, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail("HIP error: %s (pt_rng_stream)", hipGetErrorString(e));
    return 0;
}

extern "C" int pt_primary_hits(pt_context *ctx, int seed, int jitter, int tree, int32_t *host_face_ids, float *host_distances,
                               uint64_t *redone_walks) {
    if (ready(ctx)) return 1;
    if (!host_face_ids) return fail("pt_primary_hits: null buffer");
    if (tree != 0 && tree != 2 && tree != 4) return fail("pt_primary_hits: tree must be 0, 2 or 4");
    if (tree == 0) tree = ctx->walk_tree ? ctx->walk_tree : 2;
    if (tree == 4 && ctx->stack_need4 > 64) return fail("pt_primary_hits: the four-wide form of this tree needs more than 64 stack entries");
    size_t n = (size_t)ctx->P.width * (size_t)ctx->P.height;
    int32_t *d_faces = nullptr;
    float *d_dist = nullptr;
    HIP_TRY(hipMalloc((void **)&d_faces, n * sizeof(int32_t)));
    hipError_t e = hipMalloc((void **)&d_dist, n * sizeof(float));
    if (e != hipSuccess) {
        (void)hipFree(d_faces);
        return fail("HIP error: %s (pt_primary_hits)", hipGetErrorString(e));
    }
    set_launch(ctx->P, seed, 0, 1, jitter);
    e = hipMemsetAsync(ctx->counters.ptr, 0, sizeof(unsigned long long), ctx->stream);
    if (e == hipSuccess) {
        ctx->kern->primary(tree == 4 ? ctx->stack_need4 : ctx->stack_need, tree == 4, grid_of(ctx), ctx->P, d_faces, d_dist,
                           ctx->counters.ptr, ctx->stream);
        e = hipStreamSynchronize(ctx->stream);
    }
    unsigned long long redone = 0;
    if (e == hipSuccess) e = hipMemcpy(host_face_ids, d_faces, n * sizeof(int32_t), hipMemcpyDeviceToHost);
    if (e == hipSuccess && host_distances) e = hipMemcpy(host_distances, d_dist, n * sizeof(float), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(&redone, ctx->counters.ptr, sizeof(redone), hipMemcpyDeviceToHost);
    (void)hipFree(d_faces);
    (void)hipFree(d_dist);
    if (e != hipSuccess) return fail("HIP error: %s (pt_primary_hits)", hipGetErrorString(e));
    if (redone_walks) *redone_walks = redone;
    return check_fault(ctx, "pt_primary_hits");
}

extern "C" int pt_count_batch(pt_context *ctx, int seed, int current_samples, pt_counters *out) {
    if (ready(ctx)) return 1;
    if (!out || current_samples <= 0) return fail("pt_count_batch: bad arguments");
    HIP_TRY(hipMemsetAsync(ctx->counters.ptr, 0, ctx->counters.bytes(), ctx->stream));
    set_launch(ctx->P, seed, 0, current_samples, 1);
    {
        TileScratch ts = own_scratch(ctx);
        launch_render<true>(ctx, ctx->P, ts, ctx->walk_tree == 4, nullptr, ctx->counters.ptr, ctx->stream);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    unsigned long long h[32];
    HIP_TRY(hipMemcpy(h, ctx->counters.ptr, sizeof(h), hipMemcpyDeviceToHost));
    for (int k = 0; k < 6; k++) {
        out->region_iters[k] = h[8 + 2 * k];
        out->region_lanes[k] = h[9 + 2 * k];
        out->region_ticks[k] = h[20 + k];
    }
    out->wave_ticks = h[26];
    for (int k = 0; k < 3; k++) out->node_step_idle[k] = h[27 + k];
    out->redone_walks = h[30];
    out->samples = h[0];
    out->rays = h[1];
    out->inner_visits = h[2];
    out->tri_tests = h[3];
    out->traversals = h[4];
    out->shaded_hits = h[5];
    out->bounces = h[6];
    return 0;
}

// ---- multi-GPU in one process -------------------------------------------------------------
// Scene, BVH and LST replicated per GPU; batch b goes to rank b mod N with seed + b; every rank keeps the running
// mean of ITS batches in its library-owned accumulator; at the end each rank scales its mean by its share of
// the samples and ONE ncclReduce(sum) over xGMI puts the image on rank 0. No other collective.

namespace {
struct Rccl {
    void *lib = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclReduce) Reduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
Rccl g_rccl;

int rccl_load() {
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    if (g_rccl.lib) return 0;
    void *lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!lib) return fail("pt_group_create: cannot open librccl.so (%s)", dlerror());
    Rccl r;
    r.lib = lib;
    r.CommInitAll = (decltype(r.CommInitAll))dlsym(lib, "ncclCommInitAll");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(lib, "ncclCommDestroy");
    r.Reduce = (decltype(r.Reduce))dlsym(lib, "ncclReduce");
    r.GroupStart = (decltype(r.GroupStart))dlsym(lib, "ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))dlsym(lib, "ncclGroupEnd");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(lib, "ncclGetErrorString");
    if (!r.CommInitAll || !r.CommDestroy || !r.Reduce || !r.GroupStart || !r.GroupEnd || !r.GetErrorString) {
        dlclose(lib);
        return fail("pt_group_create: librccl.so lacks a symbol");
    }
    g_rccl = r;
    return 0;
}

#define RCCL_TRY(expr)                                                                                   \
    do {                                                                                                 \
        ncclResult_t r_ = (expr);                                                                        \
        if (r_ != ncclSuccess) return fail("RCCL error: %s (%s)", g_rccl.GetErrorString(r_), #expr);      \
    } while (0)
} // namespace

struct pt_group {
    std::vector<pt_context *> ctx;
    std::vector<ncclComm_t> comm;
};

extern "C" int pt_group_create(const int *devices, int count, pt_group **out) {
    if (!out) return fail("pt_group_create: null output pointer");
    *out = nullptr;
    if (!devices || count <= 0 || count > 64) return fail("pt_group_create: bad device list");
    for (int i = 0; i < count; i++)
        for (int j = 0; j < i; j++)
            if (devices[i] == devices[j]) return fail("pt_group_create: device %d listed twice", devices[i]);
    if (rccl_load()) return 1;
    pt_group *g = new pt_group();
    for (int i = 0; i < count; i++) {
        pt_context *c = nullptr;
        if (pt_create(devices[i], &c)) {
            pt_group_destroy(g);
            return 1;
        }
        g->ctx.push_back(c);
    }
    g->comm.assign((size_t)count, nullptr);
    ncclResult_t r = g_rccl.CommInitAll(g->comm.data(), count, devices);
    if (r != ncclSuccess) {
        g->comm.clear();
        pt_group_destroy(g);
        return fail("RCCL error: %s (ncclCommInitAll)", g_rccl.GetErrorString(r));
    }
    *out = g;
    return 0;
}

extern "C" void pt_group_destroy(pt_group *g) {
    if (!g) return;
    for (size_t i = 0; i < g->comm.size(); i++)
        if (g->comm[i]) {
            (void)hipSetDevice(g->ctx[i]->device);
            (void)g_rccl.CommDestroy(g->comm[i]);
        }
    for (pt_context *c : g->ctx) pt_destroy(c);
    delete g;
}

extern "C" int pt_group_size(pt_group *g) { return g ? (int)g->ctx.size() : 0; }

extern "C" pt_context *pt_group_context(pt_group *g, int rank) {
    return (g && rank >= 0 && (size_t)rank < g->ctx.size()) ? g->ctx[(size_t)rank] : nullptr;
}

// The batch plan of a sharded render, as a pure function of its arguments (no device, no group): batch b of the
// reference's loop (main.cu:101-102: cur = min(samples_every_update, remaining)) runs on rank b mod ranks with
// seed + b, and continues that rank's running mean (previous_samples = what the rank has rendered so far).
// weights[r] = rank r's share of the samples; they sum to 1 (in double, before the rounding to float), and a rank
// without a batch has weight 0 (its zero-filled accumulator then adds nothing to the reduce).
extern "C" int pt_group_plan(int seed, int samples, int samples_every_update, int ranks, pt_group_batch *batches, int capacity,
                             float *weights) {
    if (samples <= 0 || samples_every_update <= 0 || ranks <= 0) {
        fail("pt_group_plan: bad arguments (samples %d, samples_every_update %d, ranks %d)", samples, samples_every_update, ranks);
        return -1;
    }
    std::vector<int> local((size_t)ranks, 0);
    int done = 0, b = 0;
    while (done < samples) {
        const int cur = samples_every_update < samples - done ? samples_every_update : samples - done;
        const int r = b % ranks;
        if (batches && b < capacity) batches[b] = pt_group_batch{b, r, (int)((int64_t)seed + b), local[(size_t)r], cur};
        local[(size_t)r] += cur;
        done += cur;
        b++;
    }
    if (weights)
        for (int r = 0; r < ranks; r++) weights[r] = (float)((double)local[(size_t)r] / (double)samples);
    return b;
}

extern "C" int pt_group_render(pt_group *g, int seed, int samples, int samples_every_update) {
    if (!g || g->ctx.empty()) return fail("pt_group_render: null group");
    if (samples <= 0 || samples_every_update <= 0) return fail("pt_group_render: bad sample counts (%d, %d)", samples, samples_every_update);
    const int n = (int)g->ctx.size();
    const KernelParams &P0 = g->ctx[0]->P;
    for (pt_context *c : g->ctx) {
        if (ready(c)) return 1;
        if (c->P.width != P0.width || c->P.height != P0.height) return fail("pt_group_render: the contexts are configured for different image sizes");
    }
    const int n_batches = pt_group_plan(seed, samples, samples_every_update, n, nullptr, 0, nullptr);
    if (n_batches < 0) return 1;
    std::vector<pt_group_batch> plan((size_t)n_batches);
    std::vector<float> weight((size_t)n);
    (void)pt_group_plan(seed, samples, samples_every_update, n, plan.data(), n_batches, weight.data());

    // From the first asynchronous call on, a failure must not return while other ranks' streams still carry work
    // into their accumulators, and must not leave an RCCL group open: every exit goes through `unwind`.
    bool in_group = false;
    auto unwind = [&](int rc) {
        char keep[sizeof(g_error)];
        memcpy(keep, g_error, sizeof(keep)); // the first failure's message survives the clean-up calls
        if (in_group) (void)g_rccl.GroupEnd();
        for (pt_context *c : g->ctx)
            if (hipSetDevice(c->device) == hipSuccess) (void)hipStreamSynchronize(c->stream);
        (void)hipSetDevice(g->ctx[0]->device);
        if (rc) memcpy(g_error, keep, sizeof(keep));
        return rc;
    };
#define GROUP_HIP(expr)                                                                                      \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess) return unwind(fail("HIP error: %s (%s)", hipGetErrorString(e_), #expr));        \
    } while (0)
#define GROUP_RCCL(expr)                                                                                     \
    do {                                                                                                     \
        ncclResult_t r_ = (expr);                                                                            \
        if (r_ != ncclSuccess) return unwind(fail("RCCL error: %s (%s)", g_rccl.GetErrorString(r_), #expr));  \
    } while (0)

    for (pt_context *c : g->ctx) {
        GROUP_HIP(hipSetDevice(c->device));
        GROUP_HIP(hipMemsetAsync(c->image.ptr, 0, c->image.bytes(), c->stream));
    }
    // the reference's batch loop (main.cu:101-133), batch b on rank b mod n: all launches are asynchronous on
    // their context's stream, so the GPUs render side by side. The first launch of a context that has not chosen
    // the form of its tree yet blocks the host for about 5 ms while it times both: rank 0 (batch 0 is its) does
    // that once, and a rank that holds the same tree under the same arithmetic takes rank 0's answer instead of
    // timing it again, one device after the other (n x 8 ms before the GPUs would all be rendering).
    for (const pt_group_batch &b : plan) {
        pt_context *c = g->ctx[(size_t)b.rank];
        const pt_context *c0 = g->ctx[0];
        if (c != c0 && c->walk_tree == 0 && c0->walk_tree != 0 && c->kern == c0->kern && c->tree_at_upload == 0 &&
            c->coincident == c0->coincident && c->nodes.count == c0->nodes.count && c->nodes4.count == c0->nodes4.count &&
            c->tris.count == c0->tris.count && c->stack_need == c0->stack_need && c->stack_need4 == c0->stack_need4)
            c->walk_tree = c0->walk_tree;
        if (pt_render_batch_into(c, c->image.ptr, nullptr, b.seed, b.previous_samples, b.current_samples)) return unwind(1);
    }
    // weight every rank's mean by its share of the samples, then ONE sum-reduce to rank 0
    const size_t count = g->ctx[0]->image.count;
    for (int r = 0; r < n; r++) {
        pt_context *c = g->ctx[(size_t)r];
        GROUP_HIP(hipSetDevice(c->device));
        hipLaunchKernelGGL(scale_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, c->stream, c->image.ptr, count, weight[(size_t)r]);
        GROUP_HIP(hipGetLastError());
    }
    GROUP_RCCL(g_rccl.GroupStart());
    in_group = true;
    for (int r = 0; r < n; r++) {
        pt_context *c = g->ctx[(size_t)r];
        GROUP_HIP(hipSetDevice(c->device));
        GROUP_RCCL(g_rccl.Reduce(c->image.ptr, c->image.ptr, count, ncclFloat, ncclSum, 0, g->comm[(size_t)r], c->stream));
    }
    in_group = false;
    GROUP_RCCL(g_rccl.GroupEnd());
    for (pt_context *c : g->ctx) {
        GROUP_HIP(hipSetDevice(c->device));
        GROUP_HIP(hipStreamSynchronize(c->stream));
    }
    (void)hipSetDevice(g->ctx[0]->device);
#undef GROUP_HIP
#undef GROUP_RCCL
    return 0;
}

extern "C" int pt_walk_tree(pt_context *ctx) { return ctx ? ctx->walk_tree : 0; }

extern "C" int pt_set_arithmetic(pt_context *ctx, int arithmetic) {
    if (!ctx) return fail("null context");
    if (arithmetic != PT_ARITHMETIC_EXACT && arithmetic != PT_ARITHMETIC_FAST) return fail("pt_set_arithmetic: unknown arithmetic %d", arithmetic);
    const KernelLaunchers *k = arithmetic == PT_ARITHMETIC_FAST ? &launchers_fast() : &launchers_exact();
    if (k == ctx->kern) return 0;
    if (bind(ctx)) return 1;
    HIP_TRY(hipStreamSynchronize(ctx->stream)); // launches of the other variant in flight keep their tile-cost writes to themselves
    ctx->kern = k;
    ctx->walk_tree = initial_tree(ctx); // which tree is faster is timed per variant
    ctx->order_valid = false;
    ctx->epoch++; // the static scratch's tile order belonged to the other variant's costs
    return 0;
}

extern "C" int pt_arithmetic(pt_context *ctx) { return (ctx && ctx->kern == &launchers_fast()) ? PT_ARITHMETIC_FAST : PT_ARITHMETIC_EXACT; }

extern "C" uint64_t pt_device_footprint(pt_context *ctx) {
    if (!ctx) return 0;
    return ctx->nodes.bytes() + ctx->nodes4.bytes() + ctx->tris.bytes() + ctx->shade.bytes() + ctx->materials.bytes() + ctx->lights.bytes() +
           ctx->lst.bytes() + ctx->glights.bytes() + ctx->textures.bytes() + ctx->texel_bytes + ctx->jump.bytes() + ctx->image.bytes() +
           ctx->tile_cost.bytes() + ctx->tile_key.bytes() + ctx->tile_order.bytes() + ctx->tile_rng.bytes();
}
