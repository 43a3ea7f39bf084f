// kernel.h — the radiance kernel for gfx950 (wave64).
//
// What it computes is render_kernel of the reference (src/renderer.cu:361-405)
// and everything below it; HOW is different:
//
//  * One trace site. The reference nests sample loop > bounce loop > traversal
//    loop, with separate traversals for the bounce ray, the NEE shadow ray and
//    one light-pdf ray per emissive LST node. On a 64-wide wave that nesting
//    leaves most lanes parked at inner reconvergence points. Here every lane
//    runs a flat state machine whose every iteration ends in the SAME closest-hit
//    walk; a lane that finishes a path starts its next sample in the next
//    iteration instead of waiting for the wave. The per-pixel order of RNG
//    draws is unchanged, so results are too.
//  * The light-pdf rays of evaluate_direct_p (renderer.cu:251-284) have the same
//    origin and direction as the next bounce ray (renderer.cu:350-351 ->
//    :112-117), so their closest hit IS the next bounce's closest hit: one walk
//    answers all N+1 queries.
//  * Hit records are built once per path vertex from (face, u, v), not at every
//    closer hit inside the walk (intersect.cu:100-210).
//  * Traversal stack: child references (u32) in LDS, [entry][thread] so a wave's
//    push/pop is one conflict-free ds_write_b32/ds_read_b32.
//  * The camera basis and 2^exposure are per-launch constants computed on the
//    host with the reference's operation order (renderer.cu:32-46, :393).
//  * One handler per path vertex (vertex_handler): the direct term and the next
//    bounce ray are finished before the shadow walk, which only answers "visible?".
//  * The walk (walk_slice) is sliced and speculative, on the reference's tree folded
//    four-wide at upload (layout.h) for rays with finite direction reciprocals; a walk
//    that meets two candidates within rounding of each other starts over as a sequential
//    walk of the binary tree, the reference's loop step for step (kCullSlack).
//  * Independent pairs of float operations are written on 2-vectors (packed fp32).
//  * Tiles are dispatched heaviest first from the costs the previous launch recorded;
//    large images give a wave two pixels per lane (render_body).
// DESIGN.md §3 has the measurements behind each of these.
//
// Every float operation keeps the reference's association; build with
// -ffp-contract=off. Doubles appear where the reference's unsuffixed literals
// promote (SURVEY.md §7 "Hidden double precision").
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "layout.h"
#include "pt_detmath.h"
#include "ptcore.h"

namespace ptc {

#define PT_DEV __device__ __forceinline__

#ifndef PT_FAST
#define PT_FAST 0 /* 1: tolerance arithmetic (contraction on, hardware reciprocal / root, no tie bookkeeping) */
#endif
// tools/isa_flops.py builds this file with -DPT_MARK_REGIONS -S and counts the fp32 operations between the marks
// (comment lines in the ISA; the product build has none)
#ifdef PT_MARK_REGIONS
#define PT_MARK(text) asm volatile("; PT_MARK " text)
#else
#define PT_MARK(text)
#endif
#ifndef PT_BLOCK
#define PT_BLOCK 64 /* threads per workgroup: one wave64 = one 8x8 pixel tile */
#endif
constexpr int kBlock = PT_BLOCK;

// ---- what the host side (ptcore.hip) and both arithmetic variants of the kernels share -------------------

struct Texture {
    const void *texels;
    int32_t width, height, isFloat, wrapS, wrapT, filter;
};

struct KernelParams {
    // geometry
    const GNode *nodes;   // binary tree: rays with an infinite direction reciprocal, sequential re-walks, probes
    const GNode4 *nodes4; // the same tree folded four-wide: every other ray
    const GTri *tris;
    const GShade *shade;
    const GMaterial *materials;
    const pt_light *lights;
    const pt_lst_node *lst;
    const GLight *glights; // one baked record per LST node
    const Texture *textures;
    const uint32_t *jump; // [32][160][8] words: XORWOW A^(2^67 * 2^k), rows padded to 8
    uint32_t root, root4;
    uint32_t lst_count;
    uint32_t lst_face_nodes; // LST nodes that are emissive faces (one light-pdf ray each in the reference)
    int32_t clear_texture;
    // image
    int32_t width, height;
    float clear_color[3];
    float exposure_scale; // powf(2, exposure)
    // camera: columns of the basis matrix and the origin (renderer.cu:32-51)
    float cam_u[3], cam_v[3], cam_w[3], cam_p[3];
    // launch
    uint32_t seed_lo, seed_hi;
    int32_t previous_samples, current_samples;
    int32_t jitter; // primary-hit probe only

    // launch order of the 8x8 tiles (heaviest first, from the previous launch's per-tile cost); both may be null
    const uint32_t *tile_order;
    uint32_t *tile_cost;
    uint32_t *tile_rng; // [tiles][8]: RNG state (v0..v4, d) of each render tile's first pixel for this launch's seed

    // Watchdog of the primary-hit probe (never null in a launch): [0] = waves that gave up on a walk of more than kWatchdog
    // slices, [1..7] = the state of the first such lane. A walk visits every node at most once and the upload rejects
    // trees with cycles, so a correct kernel never gets near the limit; a bug must not hang the GPU (one did, in round 3:
    // a miscompiled short-circuit made a lane walk the same ray for ever). The probe is where a new walk is tried first.
    // The render kernel carries no such check: two forms of it (a ballot-and-count per slice; a look at the clock every
    // 2^14 slices, scalar arithmetic only) cost 17 % and 8 % of the launch through the register allocation of its loop
    // (profiles/r03_kernel_experiments.md).
    uint32_t *fault;
};
constexpr uint32_t kWatchdog = 1u << 20; // slices of one primary-hit walk
#ifndef PT_WATCHDOG
#define PT_WATCHDOG 1 /* 0: A/B builds that measure what the check costs */
#endif

// Everything below exists twice in libptcore.so, compiled from this one source: ptc::exact (ptcore.hip, built with
// -ffp-contract=off: every float operation rounds as the reference's restatement does, the gate) and ptc::fast
// (kernels_fast.hip, built with -DPT_FAST=1 -ffp-contract=fast: tolerance arithmetic). pt_set_arithmetic picks
// one per context through the table in launch.h.
#if PT_FAST
#define PT_ARITH_NS fast
#else
#define PT_ARITH_NS exact
#endif
namespace PT_ARITH_NS {


struct f3 {
    float x, y, z;
};
PT_DEV f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
PT_DEV f3 ld3(const float *p) { return f3{p[0], p[1], p[2]}; }
PT_DEV f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
PT_DEV f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
PT_DEV f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
PT_DEV f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
PT_DEV f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
PT_DEV f3 operator*(float s, f3 a) { return mk3(a.x * s, a.y * s, a.z * s); }
// Division and square root. Exact arithmetic (the default, PT_FAST == 0): IEEE, correctly rounded, as the reference's
// host-side restatement computes them (hipcc expands 1.0f / x into 11 and sqrtf into 16 instructions). Tolerance
// arithmetic (PT_FAST == 1, kernels_fast.hip): the hardware's 1-ulp v_rcp_f32 / v_sqrt_f32 / v_rsq_f32.
#if PT_FAST
PT_DEV float rcp_exact(float s) { return __builtin_amdgcn_rcpf(s); }
PT_DEV float pt_div(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
PT_DEV float pt_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
#else
PT_DEV float rcp_exact(float s) { return 1.0f / s; } // == (float)(1.0 / (double)s): division is safe under double rounding
PT_DEV float pt_div(float a, float b) { return a / b; }
PT_DEV float pt_sqrt(float x) { return __builtin_sqrtf(x); }
#endif
PT_DEV f3 operator/(f3 a, float s) { return a * rcp_exact(s); }
PT_DEV float dot(f3 a, f3 b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
PT_DEV f3 cross(f3 a, f3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
PT_DEV float length(f3 a) { return pt_sqrt(dot(a, a)); }
#if PT_FAST
PT_DEV f3 normalize(f3 a) { return a * __builtin_amdgcn_rsqf(dot(a, a)); }
#else
PT_DEV f3 normalize(f3 a) { return a * rcp_exact(length(a)); }
#endif
PT_DEV float luminance(f3 c) { return c.x * 0.2126f + c.y * 0.7152f + c.z * 0.0722f; }
// glibc fmaxf/fminf selection rules (sign of zero and NaN handling match the host)
PT_DEV float max_gl(float x, float y) { return (x >= y || y != y) ? x : y; }
PT_DEV float min_gl(float x, float y) { return (x <= y || y != y) ? x : y; }

// 1e-12 is not a float; these are the float thresholds equivalent to the
// reference's double compares (intersect.cu:56,65,73; renderer.cu:149).
#define PT_EPS_UP 1.0000001044244144e-12f /* 0x2b8cbccd: smallest float >= 1e-12 */

// ---- XORWOW ------------------------------------------------------------------------

// (curand's state also caches the second value of a Box-Muller pair. The kernel draws normals only as the
// two jitter offsets of a sample, renderer.cu:376-380, i.e. always a whole pair: rng_normal_pair, no cache.)
struct Rng {
    uint32_t v0, v1, v2, v3, v4, d;
};

PT_DEV uint32_t rng_next(Rng &s) {
    uint32_t t = s.v0 ^ (s.v0 >> 2);
    s.v0 = s.v1;
    s.v1 = s.v2;
    s.v2 = s.v3;
    s.v3 = s.v4;
    s.v4 = (s.v4 ^ (s.v4 << 4)) ^ (t ^ (t << 1));
    s.d += 362437u;
    return s.v4 + s.d;
}

// One 160x160 GF(2) matrix-vector product. The matrix address is wave-uniform,
// so rows arrive through scalar loads; lanes only differ in the select mask.
PT_DEV void rng_matvec(const uint32_t *__restrict__ m, uint32_t v[5]) {
    uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0, r4 = 0;
    for (int i = 0; i < 5; i++) {
        uint32_t word = v[i];
#pragma unroll 8
        for (int j = 0; j < 32; j++) {
            uint32_t mask = 0u - ((word >> j) & 1u);
            const uint32_t *row = m + (i * 32 + j) * 8;
            r0 ^= mask & row[0];
            r1 ^= mask & row[1];
            r2 ^= mask & row[2];
            r3 ^= mask & row[3];
            r4 ^= mask & row[4];
        }
    }
    v[0] = r0; v[1] = r1; v[2] = r2; v[3] = r3; v[4] = r4;
}

// curand_init(seed, subsequence, 0): scramble, then jump subsequence * 2^67.
// curand_init's seed scramble: the state of subsequence 0.
PT_DEV void rng_seed(uint32_t v[5], uint32_t *d, uint32_t seed_lo, uint32_t seed_hi) {
    uint32_t s0 = seed_lo ^ 0xaad26b49u;
    uint32_t s1 = seed_hi ^ 0xf7dcefddu;
    uint32_t t0 = 1099087573u * s0;
    uint32_t t1 = 2591861531u * s1;
    v[0] = 123456789u + t0;
    v[1] = 362436069u ^ t0;
    v[2] = 521288629u + t1;
    v[3] = 88675123u ^ t1;
    v[4] = 5783321u + t0;
    *d = 6615241u + t1 + t0;
}

// Jump `count` subsequences (count * 2^67 steps) ahead: one matrix-vector product per set bit.
// Bits that are set in NO lane of the wave are skipped by a uniform branch.
PT_DEV void rng_jump(uint32_t v[5], const uint32_t *__restrict__ jump, uint32_t count) {
    for (int k = 0; k < 32; k++) {
        bool mine = (count >> k) & 1u;
        if (__builtin_amdgcn_ballot_w64(mine) == 0) continue;
        uint32_t w[5] = {v[0], v[1], v[2], v[3], v[4]};
        rng_matvec(jump + (size_t)k * 160 * 8, w);
        if (mine) {
            v[0] = w[0]; v[1] = w[1]; v[2] = w[2]; v[3] = w[3]; v[4] = w[4];
        }
    }
}

// curand_init(seed, subsequence, 0): scramble, then jump subsequence * 2^67.
PT_DEV void rng_init(Rng &s, const uint32_t *__restrict__ jump, uint32_t seed_lo, uint32_t seed_hi, uint32_t subsequence) {
    uint32_t v[5];
    rng_seed(v, &s.d, seed_lo, seed_hi);
    rng_jump(v, jump, subsequence);
    s.v0 = v[0]; s.v1 = v[1]; s.v2 = v[2]; s.v3 = v[3]; s.v4 = v[4];
}

// The same state reached in two jumps: `base` is the state of the tile's first pixel (tile_rng_kernel
// computes it once per tile, where the render kernel's 64 lanes would each repeat the work), `delta`
// the pixel's subsequence minus that pixel's. A^(a+b) = A^a * A^b, so the state is the same.
PT_DEV void rng_init_from(Rng &s, const uint32_t *__restrict__ jump, const uint32_t *__restrict__ base, uint32_t delta) {
    uint32_t v[5] = {base[0], base[1], base[2], base[3], base[4]};
    s.d = base[5];
    rng_jump(v, jump, delta);
    s.v0 = v[0]; s.v1 = v[1]; s.v2 = v[2]; s.v3 = v[3]; s.v4 = v[4];
}

PT_DEV float rng_uniform(Rng &s) { return (float)rng_next(s) * 2.3283064e-10f + (2.3283064e-10f / 2.0f); }

// Two calls of curand_normal: the first computes the pair and returns r * sin, the second returns the cached r * cos.
PT_DEV void rng_normal_pair(Rng &s, float *first, float *second) {
    uint32_t x = rng_next(s), y = rng_next(s);
    const float two_pi_scale = 2.3283064e-10f * 6.2831855f;
    float u = (float)x * 2.3283064e-10f + (2.3283064e-10f / 2.0f);
    float v = (float)y * two_pi_scale + (two_pi_scale / 2.0f);
#if PT_FAST
    // v_log_f32 is log2, v_sin_f32 / v_cos_f32 take revolutions (the fast intrinsics curand_normal itself uses on
    // the device are of this kind): the u32 stream is untouched, the two floats are within 1e-6 of the exact form's
    (void)v;
    float r = pt_sqrt(-2.0f * (__builtin_amdgcn_logf(u) * 0.69314718056f));
    float turn = (float)y * 2.3283064e-10f + (2.3283064e-10f / 2.0f);
    float sn = __builtin_amdgcn_sinf(turn), cs = __builtin_amdgcn_cosf(turn);
#else
    float r = __builtin_sqrtf(-2.0f * pt_logf(u));
    float sn, cs;
    pt_sincosf(v, &sn, &cs);
#endif
    *first = r * sn;
    *second = r * cs;
}

// ---- textures ------------------------------------------------------------------------

PT_DEV float wrap_coord(float u, int mode) {
    if (mode == PT_WRAP_CLAMP) return u < 0.0f ? 0.0f : (u > 1.0f ? 1.0f : u);
    if (mode == PT_WRAP_MIRROR) {
        float t = u - 2.0f * __builtin_floorf(u * 0.5f);
        return t > 1.0f ? 2.0f - t : t;
    }
    return u - __builtin_floorf(u);
}
PT_DEV int wrap_index(int i, int n, int mode) {
    if (mode == PT_WRAP_REPEAT) {
        i %= n;
        return i < 0 ? i + n : i;
    }
    return i < 0 ? 0 : (i >= n ? n - 1 : i);
}
PT_DEV float4 fetch_texel(const Texture &t, int x, int y) {
    size_t at = (size_t)y * (size_t)t.width + (size_t)x;
    if (t.isFloat) return ((const float4 *)t.texels)[at];
    uchar4 p = ((const uchar4 *)t.texels)[at];
    return make_float4(pt_div((float)p.x, 255.0f), pt_div((float)p.y, 255.0f), pt_div((float)p.z, 255.0f), pt_div((float)p.w, 255.0f));
}
__device__ __noinline__ float4 sample_texture(const Texture *tp, float u, float v) {
    Texture t = *tp;
    float fu = wrap_coord(u, t.wrapS) * (float)t.width;
    float fv = wrap_coord(v, t.wrapT) * (float)t.height;
    if (t.filter == PT_FILTER_POINT)
        return fetch_texel(t, wrap_index((int)__builtin_floorf(fu), t.width, t.wrapS),
                           wrap_index((int)__builtin_floorf(fv), t.height, t.wrapT));
    float xb = fu - 0.5f, yb = fv - 0.5f;
    float x0 = __builtin_floorf(xb), y0 = __builtin_floorf(yb);
    float a = __builtin_floorf((xb - x0) * 256.0f + 0.5f) / 256.0f;
    float b = __builtin_floorf((yb - y0) * 256.0f + 0.5f) / 256.0f;
    int ix0 = wrap_index((int)x0, t.width, t.wrapS), ix1 = wrap_index((int)x0 + 1, t.width, t.wrapS);
    int iy0 = wrap_index((int)y0, t.height, t.wrapT), iy1 = wrap_index((int)y0 + 1, t.height, t.wrapT);
    float4 t00 = fetch_texel(t, ix0, iy0), t10 = fetch_texel(t, ix1, iy0);
    float4 t01 = fetch_texel(t, ix0, iy1), t11 = fetch_texel(t, ix1, iy1);
    float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
    float4 r;
    r.x = ((w00 * t00.x + w10 * t10.x) + w01 * t01.x) + w11 * t11.x;
    r.y = ((w00 * t00.y + w10 * t10.y) + w01 * t01.y) + w11 * t11.y;
    r.z = ((w00 * t00.z + w10 * t10.z) + w01 * t01.z) + w11 * t11.z;
    r.w = ((w00 * t00.w + w10 * t10.w) + w01 * t01.w) + w11 * t11.w;
    return r;
}

// ---- closest-hit walk ------------------------------------------------------------------

constexpr float kNoHit = 1e30f;

struct HitLite {
    float t, u, v;
    float t2; // distance of the nearest other candidate tested (see kCullSlack)
    int32_t face; // -1: miss
    bool early;   // walk stopped at an occluder closer than t_stop (shadow rays only)
};

// Shadow rays need a yes/no answer, not the closest hit. A walk may start with t_init below
// 1e30 (nothing beyond it matters) and stop at the first hit with t <= t_stop. To stay EXACTLY
// equal to the reference's closest-hit visibility tests the two limits straddle the decisive
// distance D by a relative band far wider than any rounding difference between the slab and
// triangle arithmetic: t_stop = D(1-kBand) ("certainly in front"), t_init = D(1+kBand)
// ("certainly behind"). A hit inside the band does not stop the walk, which then finishes as an
// ordinary closest-hit walk among t < t_init, with the reference's order and tie rule, and the
// reference's own comparison is applied to its result.
constexpr float kBand = 1e-4f;

// The walks below are speculative: a lane parks a leaf and keeps descending, so it culls boxes against a
// staler best distance than the reference, and a four-wide walk also meets leaves in another order than the
// reference's binary traversal. While the closest hit is closest by a margin none of that matters. It does
// when two candidates lie within rounding of each other: which is tested first decides an exact tie (strict
// "closer" rule), and the box test (other arithmetic than the triangle test) may cull one of them against one
// best distance and not against another. So a speculative walk
//  (a) on the wide tree, culls boxes against the best distance times kCullSlack (on the binary tree its stale
//      distance already culls less than the reference does): every candidate that close to the best is tested;
//  (b) keeps, beside the best distance, the distance of the nearest OTHER candidate it tested (HitLite::t2);
//  (c) if at its end that one lies within kNearHi of the winner, starts over as a SEQUENTIAL walk of the binary
//      tree (render_body): a leaf is tested before the walk goes on, which is the reference's loop
//      (bvh.cu:239-286) step for step. The ray's t_init, negated, marks such a walk until the next ray.
// Box and triangle arithmetic agree to a few units of 2^-24, far inside both margins, so a speculative walk that
// noticed nothing has a winner every other candidate loses to in any order and under any culling.
constexpr float kCullSlack = 1.0f + 0x1p-13f;
constexpr float kNearHi = 1.0f + 0x1p-12f; // twice the cull slack

struct Tally {
    uint32_t samples, rays, inner, tris, walks, shaded, bounces;
    // wave-level occupancy of each code region (instrumented variant only): iterations executed by
    // the wave, and the sum of lanes that were active in them
    uint32_t w_iters[6], w_lanes[6];
    uint32_t redone; // speculative walks repeated as sequential ones (see kCullSlack)
    uint32_t w_idle[3]; // lanes that sat out a node step: blocked on two leaves | walk over but for a parked leaf | waiting for a handler
    unsigned long long *w_time; // LDS: shader-clock ticks per region, written by the wave's leader lane
};
enum : int { kRegLoop = 0, kRegInner = 1, kRegLeaf = 2, kRegMain = 3, kRegNee = 4, kRegBsdf = 5 };

template <bool COUNT> PT_DEV unsigned long long stamp() {
    if (COUNT) return __builtin_amdgcn_s_memtime();
    return 0;
}
template <bool COUNT> PT_DEV void region_time(Tally &n, int region, unsigned long long t0, unsigned long long t1) {
    if (COUNT) {
        unsigned long long m = __builtin_amdgcn_ballot_w64(true);
        unsigned lane = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
        if (lane == 0) n.w_time[region] += t1 - t0;
    }
}

template <bool COUNT> PT_DEV void region_tick(Tally &n, int region) {
    if (COUNT) {
        unsigned long long m = __builtin_amdgcn_ballot_w64(true);
        unsigned lane = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
        if (lane == 0) { // lowest active lane speaks for the wave
            n.w_iters[region]++;
            n.w_lanes[region] += (uint32_t)__builtin_popcountll(m);
        }
    }
}

// intersect_aabb, src/bvh.cu:222-237 (select forms and tie rules kept)
PT_DEV float slab(f3 bmin, f3 bmax, f3 o, f3 rinv, float limit) {
    f3 t1 = (bmin - o) * rinv;
    f3 t2 = (bmax - o) * rinv;
    f3 lo = mk3(t1.x < t2.x ? t1.x : t2.x, t1.y < t2.y ? t1.y : t2.y, t1.z < t2.z ? t1.z : t2.z);
    f3 hi = mk3(t1.x > t2.x ? t1.x : t2.x, t1.y > t2.y ? t1.y : t2.y, t1.z > t2.z ? t1.z : t2.z);
    float tmin = (lo.x > lo.y && lo.x > lo.z) ? lo.x : (lo.y > lo.z ? lo.y : lo.z);
    float tmax = (hi.x < hi.y && hi.x < hi.z) ? hi.x : (hi.y < hi.z ? hi.y : hi.z);
    return (tmax >= tmin && tmin < limit && tmax > 0) ? tmin : kNoHit;
}

// The same test with v_min/v_max/v_min3/v_max3. Equal to slab() in value whenever no
// operand is NaN (signed zeros only differ in sign, which no later compare sees); NaN needs
// 0 * inf, i.e. an infinite reciprocal, and those rays take slab() instead.
PT_DEV float slab_fast(f3 bmin, f3 bmax, f3 o, f3 rinv, float limit) {
#if PT_FAST /* here `o` is origin * rinv: one fused multiply-add per plane */
    f3 t1 = mk3(__builtin_fmaf(bmin.x, rinv.x, -o.x), __builtin_fmaf(bmin.y, rinv.y, -o.y), __builtin_fmaf(bmin.z, rinv.z, -o.z));
    f3 t2 = mk3(__builtin_fmaf(bmax.x, rinv.x, -o.x), __builtin_fmaf(bmax.y, rinv.y, -o.y), __builtin_fmaf(bmax.z, rinv.z, -o.z));
#else
    f3 t1 = (bmin - o) * rinv;
    f3 t2 = (bmax - o) * rinv;
#endif
    float tmin = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(t1.x, t2.x), __builtin_fminf(t1.y, t2.y)),
                                 __builtin_fminf(t1.z, t2.z));
    float tmax = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(t1.x, t2.x), __builtin_fmaxf(t1.y, t2.y)),
                                 __builtin_fmaxf(t1.z, t2.z));
    return (tmax >= tmin && tmin < limit && tmax > 0) ? tmin : kNoHit;
}

// slab_fast after the plane distances have been formed (the wide step forms them two children at a time)
typedef float v2f __attribute__((ext_vector_type(2)));
PT_DEV float slab_planes(float ax, float ay, float az, float bx, float by, float bz, float limit) {
    float tmin = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(ax, bx), __builtin_fminf(ay, by)), __builtin_fminf(az, bz));
    float tmax = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(ax, bx), __builtin_fmaxf(ay, by)), __builtin_fmaxf(az, bz));
    return (tmax >= tmin && tmin < limit && tmax > 0) ? tmin : kNoHit;
}

constexpr uint32_t kDone = 0x7FFFFFFFu; // not a node index (< 2^28 triangles => < 2^28 nodes)

// ---- sliced, speculative walk ------------------------------------------------------------
//
// The same closest-hit walk, restructured for a 64-wide wave:
//  * Speculation: a lane that reaches a leaf parks it (`pleaf`) and keeps descending; leaf
//    triangles are tested later, in the order the leaves were found. The lane therefore
//    culls with a slightly stale (larger) best distance and may visit nodes the reference
//    would have skipped; every face the reference tests is still tested, in the same relative
//    order, with the same strict "closer" rule. The result is the same unless one of the extra
//    faces lies within rounding of the winner: those walks are found out and repeated
//    sequentially (kCullSlack above; a sequential walk never goes on with a leaf untested).
//  * Wave-uniform phases: the node loop runs while no lane is blocked (holding a parked leaf
//    AND standing on another one), so no lane idles while others descend; then one leaf phase
//    serves every parked leaf.
//  * Time slicing: a slice ends after a fixed number of node iterations (PT_SLICE in the binary tree,
//    PT_SLICE_WIDE in the four-wide one). Lanes whose walk finished go
//    on to shading and come back with a new ray; lanes still walking keep (ref, sp, pleaf, h)
//    and resume in the next slice. One very long walk no longer holds 63 lanes hostage.
#ifndef PT_SLICE
#define PT_SLICE 12
#endif
#ifndef PT_SLICE_WIDE
#define PT_SLICE_WIDE (PT_FAST ? 5 : 6) /* measured per arithmetic (profiles/r03_kernel_experiments.md) */
#endif
#ifndef PT_SHADE_NUM
#define PT_SHADE_NUM 1 /* shade when finished lanes / live lanes >= NUM / DEN (0 = as soon as any lane finished) */
#endif
#ifndef PT_SHADE_DEN
#define PT_SHADE_DEN 4
#endif
#ifndef PT_BLOCKED_SHIFT
#define PT_BLOCKED_SHIFT 0 /* leave the node loop when blocked lanes >= stepping lanes >> shift */
#endif

// The lane's traversal stack of child references: every entry in LDS, [entry][lane], so a wave's push or pop is
// one conflict-free ds_write_b32 / ds_read_b32. (walk_slice takes the stack as a type: round 2 also measured a
// 16-entry LDS stack whose deeper entries spill to private memory, for kernels that need the LDS for path
// records; profiles/r02_kernel_experiments.md.)
template <int ENTRIES> struct LdsStack {
    uint32_t *base; // &lds[lane]
    static constexpr bool kUnchecked = true; // the upload checked the bound: slots above sp may be written freely
    static constexpr int kFree = ENTRIES;
    PT_DEV void put(int i, uint32_t v) { base[i * kBlock] = v; }
    PT_DEV uint32_t get(int i) const { return base[i * kBlock]; }
};
struct Walk {
    uint32_t ref;   // next node/leaf reference, kDone when the walk is finished
    uint32_t pleaf; // parked leaf reference, 0 = none
    int sp;
    HitLite h;
};

// HitLite::t2: a candidate at distance t (already known to hit the triangle in front of the origin) is about
// to be compared with the best so far. Whichever of the two does not end up as the best is a candidate
// for "nearest other". The distance a shadow walk starts from is not a hit and does not count.
PT_DEV void note_other(HitLite &h, float t) {
#if !PT_FAST /* tolerance arithmetic keeps no second distance: ties within rounding go to whichever is tested first */
    const float best = h.face >= 0 ? h.t : kNoHit;
    h.t2 = __builtin_fminf(h.t2, __builtin_fmaxf(t, best));
#endif
}

// Moeller-Trumbore on one pre-gathered triangle (src/intersect.cu:12-81, non-culling branch)
// followed by the "closer" test of intersect_face (src/intersect.cu:100).
PT_DEV void test_triangle(float4 a, float4 b, float4 c, f3 o, f3 d, float t_stop, HitLite &h) {
    f3 v0 = mk3(a.x, a.y, a.z), e1 = mk3(a.w, b.x, b.y), e2 = mk3(b.z, b.w, c.x);
    f3 pvec = cross(d, e2);
    float det = dot(e1, pvec);
    if (det > -PT_EPS_UP && det < PT_EPS_UP) return;
    float inv_det = rcp_exact(det);
    f3 tvec = o - v0;
    float u = dot(tvec, pvec) * inv_det;
    if (u <= -PT_EPS_UP || u > 1.0f) return;
    f3 qvec = cross(tvec, e1);
    float v = dot(d, qvec) * inv_det;
    if (v <= -PT_EPS_UP || (u + v) > 1.0f) return;
    float t = dot(e2, qvec) * inv_det;
    if (t >= 0) note_other(h, t);
    if (t >= 0 && t < h.t) {
        h.t = t;
        h.u = u;
        h.v = v;
        h.face = (int32_t)__float_as_uint(c.y);
        if (t <= t_stop) h.early = true;
    }
}

// Two triangles at once: every product and sum of the two Moeller-Trumbore tests as one packed fp32
// operation (same operations, same order and rounding as test_triangle; the reference's early
// returns become the `ok` flags, and triangle 0 is committed before triangle 1 is compared, as in
// the reference's loop over a leaf, bvh.cu:252-256).
PT_DEV void test_triangle_pair(float4 a0, float4 b0, float4 c0, float4 a1, float4 b1, float4 c1, bool second, f3 o, f3 d,
                               float t_stop, HitLite &h) {
    const v2f dx = {d.x, d.x}, dy = {d.y, d.y}, dz = {d.z, d.z};
    const v2f v0x = {a0.x, a1.x}, v0y = {a0.y, a1.y}, v0z = {a0.z, a1.z};
    const v2f e1x = {a0.w, a1.w}, e1y = {b0.x, b1.x}, e1z = {b0.y, b1.y};
    const v2f e2x = {b0.z, b1.z}, e2y = {b0.w, b1.w}, e2z = {c0.x, c1.x};
    // pvec = cross(d, e2)
    v2f px = dy * e2z - dz * e2y, py = dz * e2x - dx * e2z, pz = dx * e2y - dy * e2x;
    v2f det = e1x * px + (e1y * py + e1z * pz);
    v2f inv_det = {rcp_exact(det.x), rcp_exact(det.y)};
    v2f tx = v2f{o.x, o.x} - v0x, ty = v2f{o.y, o.y} - v0y, tz = v2f{o.z, o.z} - v0z;
    v2f u = (tx * px + (ty * py + tz * pz)) * inv_det;
    // qvec = cross(tvec, e1)
    v2f qx = ty * e1z - tz * e1y, qy = tz * e1x - tx * e1z, qz = tx * e1y - ty * e1x;
    v2f v = (dx * qx + (dy * qy + dz * qz)) * inv_det;
    v2f t = (e2x * qx + (e2y * qy + e2z * qz)) * inv_det;
    v2f uv = u + v;
    const bool in0 = !(det.x > -PT_EPS_UP && det.x < PT_EPS_UP) && !(u.x <= -PT_EPS_UP || u.x > 1.0f) &&
                     !(v.x <= -PT_EPS_UP || uv.x > 1.0f) && t.x >= 0;
    if (in0) note_other(h, t.x);
    bool ok0 = in0 && t.x < h.t;
    if (ok0) {
        h.t = t.x;
        h.u = u.x;
        h.v = v.x;
        h.face = (int32_t)__float_as_uint(c0.y);
        if (t.x <= t_stop) h.early = true;
    }
    const bool in1 = second && !(det.y > -PT_EPS_UP && det.y < PT_EPS_UP) && !(u.y <= -PT_EPS_UP || u.y > 1.0f) &&
                     !(v.y <= -PT_EPS_UP || uv.y > 1.0f) && t.y >= 0;
    if (in1) note_other(h, t.y);
    bool ok1 = in1 && t.y < h.t;
    if (ok1) {
        h.t = t.y;
        h.u = u.y;
        h.v = v.y;
        h.face = (int32_t)__float_as_uint(c1.y);
        if (t.y <= t_stop) h.early = true;
    }
}

// Leaves of the reference builder hold one or two faces (src/bvh.cu:143): both records are
// fetched before either is tested so the second fetch does not wait for the first test.
template <bool COUNT>
PT_DEV void test_leaf(const KernelParams &P, uint32_t leaf, f3 o, f3 d, float t_stop, HitLite &h, Tally &n) {
    uint32_t first = leaf & 0x0FFFFFFFu;
    uint32_t count = ((leaf >> 28) & 7u) + 1;
    const float4 *tp = (const float4 *)(P.tris + first);
    uint32_t second = count > 1 ? 3u : 0u; // re-reads triangle 0 when the leaf has one face (result unused)
    float4 a0 = tp[0], b0 = tp[1], c0 = tp[2];
    float4 a1 = tp[second + 0], b1 = tp[second + 1], c1 = tp[second + 2];
    if (COUNT) n.tris += count;
    test_triangle_pair(a0, b0, c0, a1, b1, c1, count > 1, o, d, t_stop, h);
    for (uint32_t k = 2; k < count; k++) { // (a foreign tree's leaves may hold up to 8)
        const float4 *tq = (const float4 *)(P.tris + first + k);
        test_triangle(tq[0], tq[1], tq[2], o, d, t_stop, h);
    }
}

// A ray with an infinite direction reciprocal walks the reference's binary tree with the exact
// select-form box test; every other ray walks the four-wide tree (see walk_slice).
template <bool WIDE>
PT_DEV bool ray_is_exact(f3 d) {
    // 1.0f / x rounds to infinity exactly when |x| <= 2^-128 (zero and the smallest subnormals)
    const float tiny = 0x1p-128f;
    return !WIDE || __builtin_fabsf(d.x) <= tiny || __builtin_fabsf(d.y) <= tiny || __builtin_fabsf(d.z) <= tiny;
}

PT_DEV void report_stuck(const KernelParams &P, const Walk &w, float t_init, int mode, uint32_t where) {
    if (atomicAdd(&P.fault[0], 1u) == 0u) {
        P.fault[1] = w.ref;
        P.fault[2] = w.pleaf;
        P.fault[3] = (uint32_t)w.sp;
        P.fault[4] = (uint32_t)w.h.face;
        P.fault[5] = __float_as_uint(t_init);
        P.fault[6] = (uint32_t)mode;
        P.fault[7] = where;
    }
}

PT_DEV void walk_reset(uint32_t root, float t_init, Walk &w) {
    w.ref = root;
    w.pleaf = 0;
    w.sp = 0;
    w.h.t = t_init;
    w.h.t2 = kNoHit;
    w.h.u = w.h.v = 0.f;
    w.h.face = -1;
    w.h.early = false;
    if (w.ref > kDone) { // the whole tree is one leaf
        w.pleaf = w.ref;
        w.ref = kDone;
    }
}

template <bool WIDE>
PT_DEV void walk_begin(const KernelParams &P, f3 d, float &t_init, Walk &w) {
    t_init = __builtin_fabsf(t_init); // a walk sent back to the binary tree leaves it negated (walk_slice)
    walk_reset((PT_FAST ? !WIDE : ray_is_exact<WIDE>(d)) ? P.root : P.root4, t_init, w);
}

template <bool COUNT, bool WIDE, class STK>
PT_DEV void walk_slice(const KernelParams &P, f3 o, f3 d, float t_init, float t_stop, STK &stack, Walk &w, Tally &n) {
#if PT_FAST
    // Tolerance arithmetic has no use for the reference's 0 * inf = NaN rules: a zero direction component is nudged to
    // +-2^-64 (every plane distance stays finite), every ray takes the same box test on the same tree, and there are no
    // sequential re-walks: in a four-wide kernel the binary step below is not even compiled.
    const float nudge = 0x1p-64f;
    f3 rinv = mk3(rcp_exact(__builtin_fabsf(d.x) < nudge ? __builtin_copysignf(nudge, d.x) : d.x),
                  rcp_exact(__builtin_fabsf(d.y) < nudge ? __builtin_copysignf(nudge, d.y) : d.y),
                  rcp_exact(__builtin_fabsf(d.z) < nudge ? __builtin_copysignf(nudge, d.z) : d.z));
    constexpr bool inf_rinv = false, seq = false, exact = !WIDE;
#else
    f3 rinv = mk3(rcp_exact(d.x), rcp_exact(d.y), rcp_exact(d.z));
    const float inf = __builtin_inff();
    const bool inf_rinv = __builtin_fabsf(rinv.x) == inf || __builtin_fabsf(rinv.y) == inf || __builtin_fabsf(rinv.z) == inf;
    const bool seq = t_init < 0.0f;                  // sequential walk (see kCullSlack)
    const bool exact = !WIDE || inf_rinv || seq;      // walks the binary tree
#endif
#if PT_FAST
    const f3 orr = o * rinv; // (NaN where 0 * inf: only for rays that take the select-form test below, which does not use it)
#endif
    constexpr int slice = WIDE ? PT_SLICE_WIDE : PT_SLICE;
    int steps = 0; // wave-uniform: every exit test below is a ballot or this counter
    for (;;) {
        unsigned long long ts0 = stamp<COUNT>();
        for (;;) {
            bool stepping = w.ref < kDone;
            bool blocked = w.ref > kDone && w.pleaf != 0;
            unsigned long long m_step = __builtin_amdgcn_ballot_w64(stepping);
            unsigned long long m_blocked = __builtin_amdgcn_ballot_w64(blocked);
            int n_step = __builtin_popcountll(m_step), n_blocked = __builtin_popcountll(m_blocked);
            if (n_step == 0 || steps >= slice || (n_blocked != 0 && n_blocked >= (n_step >> PT_BLOCKED_SHIFT))) break;
            steps++;
            if (COUNT) {
                unsigned long long m_parked = __builtin_amdgcn_ballot_w64(w.ref == kDone && w.pleaf != 0);
                unsigned long long m_fin = __builtin_amdgcn_ballot_w64(w.ref == kDone && w.pleaf == 0);
                if (stepping && __builtin_amdgcn_mbcnt_hi((unsigned)(m_step >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m_step, 0)) == 0) {
                    n.w_idle[0] += (uint32_t)n_blocked;
                    n.w_idle[1] += (uint32_t)__builtin_popcountll(m_parked);
                    n.w_idle[2] += (uint32_t)__builtin_popcountll(m_fin);
                }
            }
            if (stepping) {
                if (COUNT) n.inner++;
                region_tick<COUNT>(n, kRegInner);
                uint32_t next;
                if (WIDE && !exact) {
                    PT_MARK("wide_step begin");
                    // four-wide node (layout.h): four slab tests, nearest child next, the other hits
                    // pushed farthest first
                    const float4 *np = (const float4 *)(P.nodes4 + w.ref);
                    float4 lx = np[0], ly = np[1], lz = np[2], hx = np[3], hy = np[4], hz = np[5], cr = np[6];
                    // (b - o) * rinv for the eight planes of two children at a time: packed fp32 math
                    // (v_pk_add_f32 / v_pk_mul_f32 round like the scalar forms)
                    const v2f rx = {rinv.x, rinv.x}, ry = {rinv.y, rinv.y}, rz = {rinv.z, rinv.z};
#if PT_FAST
                    // b * rinv - o * rinv: one v_pk_fma_f32 per pair of planes (o * rinv once per slice)
                    const v2f ox = {orr.x, orr.x}, oy = {orr.y, orr.y}, oz = {orr.z, orr.z};
                    v2f ax01 = v2f{lx.x, lx.y} * rx - ox, ax23 = v2f{lx.z, lx.w} * rx - ox;
                    v2f ay01 = v2f{ly.x, ly.y} * ry - oy, ay23 = v2f{ly.z, ly.w} * ry - oy;
                    v2f az01 = v2f{lz.x, lz.y} * rz - oz, az23 = v2f{lz.z, lz.w} * rz - oz;
                    v2f bx01 = v2f{hx.x, hx.y} * rx - ox, bx23 = v2f{hx.z, hx.w} * rx - ox;
                    v2f by01 = v2f{hy.x, hy.y} * ry - oy, by23 = v2f{hy.z, hy.w} * ry - oy;
                    v2f bz01 = v2f{hz.x, hz.y} * rz - oz, bz23 = v2f{hz.z, hz.w} * rz - oz;
#else
                    const v2f ox = {o.x, o.x}, oy = {o.y, o.y}, oz = {o.z, o.z};
                    v2f ax01 = (v2f{lx.x, lx.y} - ox) * rx, ax23 = (v2f{lx.z, lx.w} - ox) * rx;
                    v2f ay01 = (v2f{ly.x, ly.y} - oy) * ry, ay23 = (v2f{ly.z, ly.w} - oy) * ry;
                    v2f az01 = (v2f{lz.x, lz.y} - oz) * rz, az23 = (v2f{lz.z, lz.w} - oz) * rz;
                    v2f bx01 = (v2f{hx.x, hx.y} - ox) * rx, bx23 = (v2f{hx.z, hx.w} - ox) * rx;
                    v2f by01 = (v2f{hy.x, hy.y} - oy) * ry, by23 = (v2f{hy.z, hy.w} - oy) * ry;
                    v2f bz01 = (v2f{hz.x, hz.y} - oz) * rz, bz23 = (v2f{hz.z, hz.w} - oz) * rz;
#endif
                    const float lim = w.h.t * kCullSlack;
                    float t0 = slab_planes(ax01.x, ay01.x, az01.x, bx01.x, by01.x, bz01.x, lim);
                    float t1 = slab_planes(ax01.y, ay01.y, az01.y, bx01.y, by01.y, bz01.y, lim);
                    float t2 = slab_planes(ax23.x, ay23.x, az23.x, bx23.x, by23.x, bz23.x, lim);
                    float t3 = slab_planes(ax23.y, ay23.y, az23.y, bx23.y, by23.y, bz23.y, lim);
                    uint32_t c0 = __float_as_uint(cr.x), c1 = __float_as_uint(cr.y), c2 = __float_as_uint(cr.z), c3 = __float_as_uint(cr.w);
                    if (c2 == kDone) t2 = kNoHit; // a wide node has two to four children
                    if (c3 == kDone) t3 = kNoHit;
#define PT_CSWAP(ta, ca, tb, cb)            \
    if (tb < ta) {                          \
        float tt_ = ta; ta = tb; tb = tt_;  \
        uint32_t cc_ = ca; ca = cb; cb = cc_; \
    }
                    PT_CSWAP(t0, c0, t1, c1)
                    PT_CSWAP(t2, c2, t3, c3)
                    PT_CSWAP(t0, c0, t2, c2)
                    PT_CSWAP(t1, c1, t3, c3)
                    PT_CSWAP(t1, c1, t2, c2)
#undef PT_CSWAP
                    if (t0 == kNoHit) {
                        next = kDone;
                        if (w.sp > 0) next = stack.get(--w.sp);
                    } else {
                        next = c0;
                        // hits sort before misses, so the pushed entries are a prefix of (c3, c2, c1) reversed
                        int h3 = t3 != kNoHit, h2 = t2 != kNoHit, h1 = t1 != kNoHit;
                        if (STK::kUnchecked || w.sp + 3 <= STK::kFree) {
                            // slots above sp are free (the stack bound is checked at upload): write all three
                            stack.put(w.sp, h3 ? c3 : (h2 ? c2 : c1));
                            stack.put(w.sp + 1, h3 ? c2 : c1);
                            stack.put(w.sp + 2, c1);
                        } else { // a stack type that checks its bounds: only what is pushed
                            int at = w.sp;
                            if (h3) stack.put(at++, c3);
                            if (h2) stack.put(at++, c2);
                            if (h1) stack.put(at++, c1);
                        }
                        w.sp += h3 + h2 + h1;
                    }
                    PT_MARK("wide_step end");
                } else {
                    PT_MARK("binary_step begin");
                    // the reference's binary tree; with an infinite reciprocal (a zero direction component) its
                    // exact select-form test, where 0 * inf = NaN decides as in intersect_aabb
                    const float4 *np = (const float4 *)(P.nodes + w.ref);
                    float4 q0 = np[0], q1 = np[1], q2 = np[2], q3 = np[3];
                    float t1, t2;
                    if (!inf_rinv) {
#if PT_FAST
                        t1 = slab_fast(mk3(q0.x, q0.y, q0.z), mk3(q0.w, q1.x, q1.y), orr, rinv, w.h.t * kCullSlack);
                        t2 = slab_fast(mk3(q1.z, q1.w, q2.x), mk3(q2.y, q2.z, q2.w), orr, rinv, w.h.t * kCullSlack);
#else
                        t1 = slab_fast(mk3(q0.x, q0.y, q0.z), mk3(q0.w, q1.x, q1.y), o, rinv, w.h.t);
                        t2 = slab_fast(mk3(q1.z, q1.w, q2.x), mk3(q2.y, q2.z, q2.w), o, rinv, w.h.t);
#endif
                    } else {
                        t1 = slab(mk3(q0.x, q0.y, q0.z), mk3(q0.w, q1.x, q1.y), o, rinv, w.h.t);
                        t2 = slab(mk3(q1.z, q1.w, q2.x), mk3(q2.y, q2.z, q2.w), o, rinv, w.h.t);
                    }
                    uint32_t c1 = __float_as_uint(q3.x), c2 = __float_as_uint(q3.y);
                    if (t2 < t1) {
                        uint32_t tc = c1; c1 = c2; c2 = tc;
                        float tt = t1; t1 = t2; t2 = tt;
                    }
                    if (t1 == kNoHit) {
                        next = kDone;
                        if (w.sp > 0) next = stack.get(--w.sp);
                    } else {
                        next = c1;
                        if (t2 != kNoHit) {
                            stack.put(w.sp, c2);
                            w.sp++;
                        }
                    }
                    PT_MARK("binary_step end");
                }
                if (next > kDone && w.pleaf == 0) { // a leaf and room to park it: keep descending
                    w.pleaf = next;
                    next = kDone;
                    if (w.sp > 0 && !seq) next = stack.get(--w.sp); // (a sequential walk waits for the leaf test)
                }
                w.ref = next;
            }
        }
        unsigned long long ts1 = stamp<COUNT>();
        region_time<COUNT>(n, kRegInner, ts0, ts1);
        // leaf phase: parked leaves first, then the leaf a lane stands on (order of discovery)
        for (;;) {
            bool has = w.pleaf != 0;
            if (__builtin_amdgcn_ballot_w64(has) == 0) break;
            if (has) {
                region_tick<COUNT>(n, kRegLeaf);
                PT_MARK("leaf_pair begin");
                test_leaf<COUNT>(P, w.pleaf, o, d, t_stop, w.h, n);
                PT_MARK("leaf_pair end");
                w.pleaf = 0;
                if (w.h.early) {
                    w.ref = kDone;
                    w.sp = 0;
                } else if (w.ref > kDone) { // standing on the next leaf: park it, step off
                    w.pleaf = w.ref;
                    w.ref = kDone;
                    if (w.sp > 0 && !seq) w.ref = stack.get(--w.sp);
                } else if (w.ref == kDone && w.sp > 0) { // a sequential walk waited for this leaf: on it goes
                    w.ref = stack.get(--w.sp);
                    if (w.ref > kDone) { // to another leaf: that one is next, and again the walk waits for it
                        w.pleaf = w.ref;
                        w.ref = kDone;
                    }
                }
            }
        }
        region_time<COUNT>(n, kRegLeaf, ts1, stamp<COUNT>());
        bool more = w.ref < kDone;
        if (steps >= slice || __builtin_amdgcn_ballot_w64(more) == 0) break;
    }
}

// ---- BRDF (src/brdf.cu:3-56, 164-178) ------------------------------------------------------

struct Brdf {
    f3 baseColor;
    float roughness, metallic, specular;
};

PT_DEV float g1_schlick_ggx(float n_dot_v, float roughness) {
    float alpha = roughness * roughness;
    float k = 0.5f * alpha; // 0.5 * alpha in double is exact
    return pt_div(n_dot_v, n_dot_v * (1.0f - k) + k);
}

// BRDF::eval for one direction, as the reference writes it. The kernel evaluates the two directions of
// a vertex together (brdf_eval_pair below, element for element the same operations); this form is
// kept as the statement of what each element computes.
PT_DEV f3 brdf_eval(Brdf b, f3 wo, f3 wi) {
    float cos_theta_wi = max_gl(wi.z, 0.0f);
    f3 l = wi, v = wo;
    f3 h = normalize(l + v);
    float f0d = PT_FAST ? 0.16f * b.specular * b.specular : (float)(0.16 * (double)b.specular * (double)b.specular);
    f3 F0_diel = mk3(1, 1, 1) * f0d;
    f3 F0 = (b.baseColor * b.metallic) + (F0_diel * (1 - b.metallic));
    float x = 1 - dot(v, h);
    f3 F = F0 + ((((((mk3(1, 1, 1) - F0) * x) * x) * x) * x) * x);
    float alpha = b.roughness * b.roughness;
    float bb = h.z * h.z * (alpha * alpha - 1.0f) + 1.0f;
    float D = pt_div(alpha * alpha, PT_PI_F * bb * bb);
    float G = g1_schlick_ggx(l.z, b.roughness) * g1_schlick_ggx(v.z, b.roughness);
    float n_dot_v = max_gl(v.z, 0.001f);
    float n_dot_l = max_gl(l.z, 0.001f);
    f3 f_spec = ((F * D) * G) / (4.0f * n_dot_v * n_dot_l);
    f3 f_diff = ((b.baseColor * (mk3(1, 1, 1) - F)) * (float)(1.0 - (double)b.metallic)) * PT_1_PI_F;
    return (f_diff + f_spec) * cos_theta_wi;
}

struct BrdfPair {
    f3 first, second;
};

// brdf_eval for two incident directions of one vertex (the light sample and the BSDF sample) at once:
// what depends on the direction runs as packed fp32 pairs, what does not (F0, G1 of the outgoing
// direction, the alpha terms) once. Every element sees the operations of brdf_eval in its order.
// (kept out of line: inlined into the vertex handler it raises the register pressure there and costs 2 %)
__device__ __noinline__ BrdfPair brdf_eval_pair(Brdf b, f3 wo, f3 wi0, f3 wi1) {
    const v2f lx = {wi0.x, wi1.x}, ly = {wi0.y, wi1.y}, lz = {wi0.z, wi1.z};
    const v2f cos_theta_wi = {max_gl(wi0.z, 0.0f), max_gl(wi1.z, 0.0f)};
    const f3 v = wo;
    // h = normalize(l + v)
    v2f sx = lx + v2f{v.x, v.x}, sy = ly + v2f{v.y, v.y}, sz = lz + v2f{v.z, v.z};
    v2f len2 = sx * sx + (sy * sy + sz * sz);
    #if PT_FAST
    v2f inv_len = {__builtin_amdgcn_rsqf(len2.x), __builtin_amdgcn_rsqf(len2.y)};
#else
    v2f inv_len = {rcp_exact(__builtin_sqrtf(len2.x)), rcp_exact(__builtin_sqrtf(len2.y))};
#endif
    v2f hx = sx * inv_len, hy = sy * inv_len, hz = sz * inv_len;
    float f0d = PT_FAST ? 0.16f * b.specular * b.specular : (float)(0.16 * (double)b.specular * (double)b.specular);
    f3 F0_diel = mk3(1, 1, 1) * f0d;
    f3 F0 = (b.baseColor * b.metallic) + (F0_diel * (1 - b.metallic));
    v2f x = v2f{1.0f, 1.0f} - (v2f{v.x, v.x} * hx + (v2f{v.y, v.y} * hy + v2f{v.z, v.z} * hz));
    f3 one_minus_F0 = mk3(1, 1, 1) - F0;
    v2f Fx = v2f{F0.x, F0.x} + (((((v2f{one_minus_F0.x, one_minus_F0.x} * x) * x) * x) * x) * x);
    v2f Fy = v2f{F0.y, F0.y} + (((((v2f{one_minus_F0.y, one_minus_F0.y} * x) * x) * x) * x) * x);
    v2f Fz = v2f{F0.z, F0.z} + (((((v2f{one_minus_F0.z, one_minus_F0.z} * x) * x) * x) * x) * x);
    float alpha = b.roughness * b.roughness;
    float a2 = alpha * alpha, a2m1 = alpha * alpha - 1.0f;
    v2f bb = hz * hz * v2f{a2m1, a2m1} + v2f{1.0f, 1.0f};
    v2f denom = v2f{PT_PI_F, PT_PI_F} * bb * bb;
    v2f D = {pt_div(a2, denom.x), pt_div(a2, denom.y)};
    float g1v = g1_schlick_ggx(v.z, b.roughness);
    v2f G = v2f{g1_schlick_ggx(wi0.z, b.roughness), g1_schlick_ggx(wi1.z, b.roughness)} * v2f{g1v, g1v};
    float n_dot_v = max_gl(v.z, 0.001f);
    v2f n_dot_l = {max_gl(wi0.z, 0.001f), max_gl(wi1.z, 0.001f)};
    v2f four_nn = v2f{4.0f * n_dot_v, 4.0f * n_dot_v} * n_dot_l;
    v2f inv_four_nn = {rcp_exact(four_nn.x), rcp_exact(four_nn.y)};
    v2f spx = ((Fx * D) * G) * inv_four_nn, spy = ((Fy * D) * G) * inv_four_nn, spz = ((Fz * D) * G) * inv_four_nn;
    float dielectric = PT_FAST ? 1.0f - b.metallic : (float)(1.0 - (double)b.metallic);
    const v2f one = {1.0f, 1.0f}, dd = {dielectric, dielectric}, ipi = {PT_1_PI_F, PT_1_PI_F};
    v2f dfx = ((v2f{b.baseColor.x, b.baseColor.x} * (one - Fx)) * dd) * ipi;
    v2f dfy = ((v2f{b.baseColor.y, b.baseColor.y} * (one - Fy)) * dd) * ipi;
    v2f dfz = ((v2f{b.baseColor.z, b.baseColor.z} * (one - Fz)) * dd) * ipi;
    v2f rx = (dfx + spx) * cos_theta_wi, ry = (dfy + spy) * cos_theta_wi, rz = (dfz + spz) * cos_theta_wi;
    BrdfPair out;
    out.first = mk3(rx.x, ry.x, rz.x);
    out.second = mk3(rx.y, ry.y, rz.y);
    return out;
}

PT_DEV float brdf_pdf() { return 1.0f / (2 * PT_PI_F); }

// rgb_to_radiometric, src/renderer.cu:146-162
PT_DEV f3 rgb_to_radiometric(f3 color, float intensity) {
    float m = length(color);
    f3 cn = mk3(1, 1, 1);
    if (m >= PT_EPS_UP) cn = color / m;
    float photometric = intensity * m;
    float alpha = (float)((double)photometric / (683.0 * (double)luminance(cn))); // (not on the kernel path: lights are baked on the host)
    return cn * alpha;
}

// ---- the per-lane state machine ------------------------------------------------------------

enum : int { kModeMain = 0, kModeNee = 1 };

struct Vertex { // what survives from a shaded hit until the next main-ray result
    f3 position, normal; // incident normal
    f3 tx, ty;           // tangent, bitangent (TBN columns 0 and 1; column 2 = normal)
    Brdf brdf;
    f3 wo_ts;
};

PT_DEV f3 tbn_mul(const Vertex &vx, f3 w) { // hit.tangentBasis * w
    return mk3((vx.tx.x * w.x + vx.ty.x * w.y) + vx.normal.x * w.z,
               (vx.tx.y * w.x + vx.ty.y * w.y) + vx.normal.y * w.z,
               (vx.tx.z * w.x + vx.ty.z * w.y) + vx.normal.z * w.z);
}
PT_DEV f3 tbn_t_mul(const Vertex &vx, f3 w) { // transposed basis * w
    return mk3((vx.tx.x * w.x + vx.tx.y * w.y) + vx.tx.z * w.z,
               (vx.ty.x * w.x + vx.ty.y * w.y) + vx.ty.z * w.z,
               (vx.normal.x * w.x + vx.normal.y * w.y) + vx.normal.z * w.z);
}

// Area-light pdf of a ray that hit `face` at parametric distance t
// (sample_area_light, src/renderer.cu:98-105,131-133).
PT_DEV float area_pdf(const GShade *s, f3 dir, float t, f3 *normal_out) {
    f3 A = ld3(s->pos[0]), B = ld3(s->pos[1]), C = ld3(s->pos[2]);
    f3 tri_cross = cross(C - B, A - B);
    float len = length(tri_cross);
    f3 normal = tri_cross / len;
    float area = 0.5f * len;
    float cos_theta_y = __builtin_fabsf(dot(normal, dir));
    if (normal_out) *normal_out = normal;
    return pt_div(t * t, area * cos_theta_y);
}

PT_DEV uint32_t pixel_of_thread(const KernelParams &P, int *px, int *py) {
    // One wave64 renders one 8x8 pixel tile, so its 64 primary rays stay close. With
    // one-wave workgroups the dispatcher balances at tile granularity (a 16x16 block of four
    // waves left 38 % of the wave slots empty in the tail of a launch). Workgroups that share
    // blockIdx % 8 share an XCD and its L2: XCD x takes tile rows x, x+8, ..., so every XCD
    // sees the whole picture (balance) while horizontal neighbours stay on one L2.
    // Once a launch has recorded per-tile costs, tile_order lists the tiles heaviest first
    // (longest-processing-time-first): the dispatcher hands out workgroups in index order, so
    // the expensive tiles start early and the cheap ones fill the tail.
    uint32_t ntx = ((uint32_t)P.width + 7u) / 8u, nty = ((uint32_t)P.height + 7u) / 8u;
    uint32_t wave = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    uint32_t tx = wave % ntx, ty = wave / ntx;
    if (P.tile_order) {
        uint32_t t = P.tile_order[wave];
        tx = t % ntx;
        ty = t / ntx;
    } else if (nty % 8u == 0) {
        uint32_t xcd = wave % 8u, i = wave / 8u;
        tx = i % ntx;
        ty = (i / ntx) * 8u + xcd;
    }
    *px = (int)(tx * 8u + (lane & 7u));
    *py = (int)(ty * 8u + (lane >> 3));
    return ty * ntx + tx;
}

#if !PT_FAST /* integer-only helper kernels: one copy, in the exact translation unit */
// Launch order for the next launch. key = (cost class, position in the static order): tiles are
// ranked heaviest class first, and inside a class they keep the static XCD-interleaved order, so
// neighbouring tiles still run together on one XCD (a plain sort by cost scatters them, which costs
// more in L2 misses on a 10 M-triangle scene than the balance gains). Classes are quarter octaves
// of the recorded ticks. order[rank(key)] = tile; keys are unique, so order is a permutation.
__global__ void __launch_bounds__(256) tile_key_kernel(const uint32_t *__restrict__ cost, uint32_t *__restrict__ key,
                                                       uint32_t n, uint32_t ntx, uint32_t nty) {
    uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= n) return;
    uint32_t c = cost[t] | 4u;
    uint32_t msb = 31u - (uint32_t)__builtin_clz(c);
    uint32_t cls = msb * 4u + ((c >> (msb - 2u)) & 3u); // <= 127
    uint32_t tx = t % ntx, ty = t / ntx;
    uint32_t w0 = (nty % 8u == 0) ? ((ty / 8u) * ntx + tx) * 8u + (ty % 8u) : t; // inverse of tile_of_wave's static map
    key[t] = ((127u - cls) << 24) | w0;
}

// rank[i] += number of keys in this block's slice of the list that are smaller than key[i]. The slice is
// read through a wave-uniform index (scalar loads), so the inner loop is one compare and one add per key;
// gridDim.y slices run in parallel and meet in `rank` (zeroed by the host beforehand).
__global__ void __launch_bounds__(64) tile_rank_kernel(const uint32_t *__restrict__ key, uint32_t *__restrict__ rank, uint32_t n) {
    uint32_t i = blockIdx.x * 64u + threadIdx.x;
    uint32_t mine = i < n ? key[i] : 0u;
    uint32_t per = (n + gridDim.y - 1u) / gridDim.y;
    uint32_t j0 = blockIdx.y * per, j1 = j0 + per < n ? j0 + per : n;
    uint32_t count = 0;
#pragma unroll 8
    for (uint32_t j = j0; j < j1; j++) count += key[j] < mine ? 1u : 0u;
    if (i < n && count) atomicAdd(&rank[i], count);
}

__global__ void __launch_bounds__(64) tile_order_kernel(const uint32_t *__restrict__ rank, uint32_t *__restrict__ order, uint32_t n) {
    uint32_t i = blockIdx.x * 64u + threadIdx.x;
    if (i < n) order[rank[i]] = i;
}

// Tile costs recorded at half the image size (render_launches), spread over the full-size tiles each one covers.
__global__ void __launch_bounds__(256) tile_expand_kernel(const uint32_t *__restrict__ half, uint32_t *__restrict__ full, uint32_t ntx,
                                                          uint32_t nty, uint32_t htx, uint32_t hty) {
    uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= ntx * nty) return;
    uint32_t x = (t % ntx) / 2u, y = (t / ntx) / 2u;
    full[t] = half[(y < hty ? y : hty - 1u) * htx + (x < htx ? x : htx - 1u)];
}

#endif // !PT_FAST

PT_DEV void camera_ray(const KernelParams &P, float u, float v, f3 *o, f3 *d) {
    f3 U = ld3(P.cam_u), V = ld3(P.cam_v), W = ld3(P.cam_w);
    // A * (u, v, 1), A = columns U,V,W (renderer.cu:46-52)
    f3 a = mk3((U.x * u + V.x * v) + W.x * 1.0f, (U.y * u + V.y * v) + W.y * 1.0f, (U.z * u + V.z * v) + W.z * 1.0f);
    *o = ld3(P.cam_p);
    *d = normalize(a);
}

// ---- one pixel's path state between walks --------------------------------------------------

struct Path {
    Rng rng;
    f3 light, thr;
    f3 ro, rd;         // the ray to walk next
    float t_init, t_stop; // its limits (see kBand); t_stop == -1 for a main ray
    int mode;          // kModeMain | kModeNee: what that ray is
    float rr_prob;
    f3 f_cos_next;     // BSDF sample's f*cos, applied once p_direct is known
    bool pending;      // a BSDF sample is waiting for the next hit to finish its MIS weight
    // Carried across a shadow walk. Every random draw of a path vertex (light choice, point on the
    // emitter, BSDF direction) precedes both of its rays in the reference (renderer.cu:330-351), and
    // the distance at which a visible emitter is met is known beforehand (t_f below), so the vertex
    // handler finishes ALL of the vertex's arithmetic: the MIS-weighted direct term as it would be
    // if the light is visible (direct_c) and the next bounce ray (rd_next). The shadow walk only
    // answers "visible?", and its result needs no second heavy handler.
    f3 rd_next, direct_c;
    float nee_dist;
    int nee_kind;      // 0 point, 1 directional, 2 area
    int32_t nee_face;
};

// First ray of a sample (render_kernel's loop body up to integrate_Li, renderer.cu:376-388).
template <bool COUNT>
PT_DEV void start_sample(const KernelParams &P, int px, int py, Path &s, Tally &n) {
    float nx, ny;
    rng_normal_pair(s.rng, &nx, &ny);
    float sx = (float)px + 0.33f * nx;
    float sy = (float)py + 0.33f * ny;
    float u = pt_div(2 * sx - (float)P.width, (float)P.height);
    float v = pt_div(2 * sy - (float)P.height, (float)P.height);
    camera_ray(P, u, v, &s.ro, &s.rd);
    s.light = mk3(0, 0, 0);
    s.thr = mk3(1, 1, 1);
    s.pending = false;
    s.mode = kModeMain;
    s.t_init = kNoHit;
    s.t_stop = -1.0f;
    if (COUNT) n.samples++;
}

// Shadow walk finished: visibility only (sample_light_source, renderer.cu:193-247); the next
// bounce ray leaves from the same origin (renderer.cu:203, :350).
template <bool COUNT>
PT_DEV void shadow_result(const HitLite &h, Path &s, Tally &n) {
    if (COUNT) n.rays++;
    bool visible;
    if (s.nee_kind == 0) {
        float closest = h.face >= 0 ? h.t : kNoHit;
        visible = !h.early && s.nee_dist < closest;
    } else if (s.nee_kind == 1) {
        visible = h.face < 0;
    } else { // met the emitter itself, in front of the origin, nothing nearer (then h.t == t_f)
        visible = !h.early && h.face == s.nee_face && h.t > 0;
    }
    if (visible) s.light = s.light + s.direct_c;
    s.rd = s.rd_next;
    s.pending = true;
    s.mode = kModeMain;
    s.t_init = kNoHit;
    s.t_stop = -1.0f;
}

// Main-ray result: one handler per path vertex. Returns true when the path ended (the caller adds
// s.light to the pixel and starts the next sample); otherwise s holds the next ray.
template <bool COUNT>
PT_DEV bool vertex_handler(const KernelParams &P, HitLite hit, Path &s, Tally &n) {
    // ---- main-ray result: one handler per path vertex ----
    region_tick<COUNT>(n, kRegMain);
    if (COUNT) n.rays++;
    bool end_path = false;
    const bool sh = hit.face >= 0;
    GShade gs;
    GMaterial m;
    if (sh) {
        // The material index comes with the shading record. (Carrying the copy in GTri through the walk, so
        // that both fetches start together, costs a register for the whole walk: with the walk's second
        // distance t2 that meant spills, 33.0 against 32.0 ms.)
        gs = P.shade[hit.face];
        m = P.materials[gs.material];
    }
    if (s.pending) {
        // evaluate_direct_p: only the LST node whose face this ray hit contributes
        float p_direct = 0.f;
        if (P.lst_count != 0) {
            if (sh && (gs.flags & kShadeIsLight) && hit.t > 0) {
                // a face listed k times in the LST contributes k equal terms
                float p_als = area_pdf(&gs, s.rd, hit.t, nullptr);
                for (uint32_t k = gs.flags >> 8; k > 0; k--) p_direct += p_als;
            }
            p_direct = pt_div(p_direct, (float)P.lst_count);
        }
        float pdf = brdf_pdf();
        float weight = pt_div(pdf, p_direct + pdf);
        s.thr = s.thr * (s.f_cos_next * pt_div(weight, pdf * s.rr_prob));
        s.pending = false;
    }
    if (!sh) {
        f3 clear = ld3(P.clear_color);
        if (P.clear_texture >= 0) {
            float phi = pt_atan2f(s.rd.z, s.rd.x);
            float theta = pt_acosf(min_gl(max_gl(s.rd.y, -1.0f), 1.0f));
            float eu = PT_FAST ? (phi + PT_PI_F) * (0.5f * PT_1_PI_F) : (float)(((double)phi + PT_PI_D) / (2.0f * PT_PI_D));
            float ev = PT_FAST ? theta * PT_1_PI_F : (float)((double)theta / PT_PI_D);
            float4 t = sample_texture(P.textures + P.clear_texture, eu, ev);
            clear = clear * mk3(t.x, t.y, t.z);
        }
        s.light = s.light + s.thr * clear;
        end_path = true;
    } else {
        // ---- hit record (intersect.cu:100-198) ----
        float u = hit.u, v = hit.v;
        float w = PT_FAST ? 1.0f - u - v : (float)(1.0 - (double)u - (double)v);
        f3 position = (w * ld3(gs.pos[0]) + u * ld3(gs.pos[1])) + v * ld3(gs.pos[2]);
        f3 baseColor = mk3(m.baseColorFactor[0], m.baseColorFactor[1], m.baseColorFactor[2]);
        float alpha = m.baseColorFactor[3];
        float roughness = m.roughness, metallic = m.metallic;
        if (m.baseColorTexture >= 0 || m.metallicRoughnessTexture >= 0) {
            float tu = (w * gs.uv[0][0] + u * gs.uv[1][0]) + v * gs.uv[2][0];
            float tv = (w * gs.uv[0][1] + u * gs.uv[1][1]) + v * gs.uv[2][1];
            if (m.baseColorTexture >= 0) {
                float4 t = sample_texture(P.textures + m.baseColorTexture, tu, tv);
                baseColor = baseColor * mk3(t.x, t.y, t.z);
                alpha *= t.w;
            }
            if (m.metallicRoughnessTexture >= 0) {
                float4 t = sample_texture(P.textures + m.metallicRoughnessTexture, tu, tv);
                roughness *= t.y;
                metallic *= t.z;
            }
        }
        if (m.alphaMode == PT_ALPHA_OPAQUE) alpha = 1.0f;
        else if (m.alphaMode == PT_ALPHA_MASK) alpha = alpha > m.alphaCutoff ? 1.0f : 0.0f;

        f3 normal, tangent;
        if (gs.flags & kShadeFlat) {
            normal = ld3(gs.nrm[0]);
            tangent = ld3(gs.tan[0]);
        } else {
            normal = (w * ld3(gs.nrm[0]) + u * ld3(gs.nrm[1])) + v * ld3(gs.nrm[2]);
            tangent = (w * ld3(gs.tan[0]) + u * ld3(gs.tan[1])) + v * ld3(gs.tan[2]);
        }
        float handed = gs.handed < 0 ? -1.0f : 1.0f;
        f3 bitangent = handed * cross(normal, tangent);
        tangent = normalize(tangent);
        bitangent = normalize(bitangent);
        normal = normalize(normal);
        if (dot(normal, s.rd) > 0) {
            tangent = tangent * -1.0f;
            bitangent = bitangent * -1.0f;
            normal = normal * -1.0f;
        }

        // ---- integrate_Li body (renderer.cu:309-355) ----
        if (rng_uniform(s.rng) >= alpha) {
            s.ro = position + 1e-6f * (-normal);
            // direction unchanged; no RR, no throughput change (renderer.cu:310-314)
        } else {
            if (COUNT) n.shaded++;
            s.light = s.light + ld3(m.emissive) * s.thr;
            s.rr_prob = min_gl(luminance(s.thr), 0.99f);
            if (rng_uniform(s.rng) >= s.rr_prob) {
                end_path = true;
            } else {
                Vertex vx;
                vx.position = position;
                vx.normal = normal;
                vx.tx = tangent;
                vx.ty = bitangent;
                vx.brdf.baseColor = baseColor;
                vx.brdf.roughness = roughness;
                vx.brdf.metallic = metallic;
                vx.brdf.specular = m.specular;
                vx.wo_ts = tbn_t_mul(vx, -s.rd);

                // ---- sample_light_source (renderer.cu:164-249), everything but the visibility walk ----
                bool shadow = false;
                f3 nee_dir = mk3(0, 0, 0), l_ts = mk3(0, 0, 1), nee_radiance = mk3(0, 0, 0);
                float nee_p = 1.0f;
                float s_init = kNoHit, s_stop = -1.0f;
                if (P.lst_count != 0) {
                    int node_index = (int)((float)P.lst_count * rng_uniform(s.rng));
                    GLight gl = P.glights[(uint32_t)node_index % P.lst_count];
                    float p_lss = rcp_exact((float)P.lst_count); // times the area pdf for an emitter
                    f3 radiance = ld3(gl.radiance);
                    f3 A = ld3(gl.a), B = ld3(gl.b), C = ld3(gl.c);
                    bool can_see = true; // false: the reference's walk cannot end on this s.light
                    s.nee_kind = (int)gl.kind;
                    if (gl.kind == 0) {
                        f3 dl = A - position;
                        float distance = length(dl);
                        nee_dir = dl / distance;
                        s.nee_dist = distance;
                        radiance = radiance / (distance * distance);
                        s_init = distance + distance * kBand;
                        s_stop = distance - distance * kBand;
                    } else if (gl.kind == 1) {
                        nee_dir = A;
                        s_init = kNoHit; // visible iff nothing at all is hit: stop at the first hit
                        s_stop = kNoHit;
                    } else {
                        float u1, u2;
                        unsigned long long tr0 = stamp<COUNT>();
                        do {
                            u1 = rng_uniform(s.rng);
                            u2 = rng_uniform(s.rng);
                        } while (u1 + u2 > 1);
                        region_time<COUNT>(n, kRegBsdf, tr0, stamp<COUNT>());
                        f3 light_pos = (A + u1 * (B - A)) + u2 * (C - A);
                        nee_dir = normalize(light_pos - position);
                        s.nee_face = (int32_t)gl.face;
                        // distance at which this ray meets the emitter itself (same arithmetic as the
                        // walk's triangle test); without such a hit the emitter cannot be the closest face
                        f3 so = position + 1e-6f * normal;
                        f3 e1 = B - A, e2 = C - A;
                        f3 pvec = cross(nee_dir, e2);
                        float det = dot(e1, pvec);
                        float t_f = -1.0f;
                        if (!(det > -PT_EPS_UP && det < PT_EPS_UP)) {
                            float inv_det = rcp_exact(det);
                            f3 tvec = so - A;
                            float uu = dot(tvec, pvec) * inv_det;
                            if (!(uu <= -PT_EPS_UP || uu > 1.0f)) {
                                f3 qvec = cross(tvec, e1);
                                float vv = dot(nee_dir, qvec) * inv_det;
                                if (!(vv <= -PT_EPS_UP || (uu + vv) > 1.0f)) t_f = dot(e2, qvec) * inv_det;
                            }
                        }
                        if (t_f > 0) {
                            s_init = t_f + t_f * kBand;
                            s_stop = t_f - t_f * kBand;
                            // sample_area_light, renderer.cu:131-133, with the emitter's baked normal and area
                            float cos_theta_y = __builtin_fabsf(dot(ld3(gl.normal), nee_dir));
                            float p_als = pt_div(t_f * t_f, gl.area * cos_theta_y);
                            p_lss = p_lss * p_als;
                        } else {
                            can_see = false;
                        }
                    }
                    if (can_see && p_lss > 0) {
                        l_ts = tbn_t_mul(vx, nee_dir);
                        nee_radiance = radiance;
                        nee_p = p_lss;
                        shadow = true;
                    } else {
                        if (COUNT) n.rays++; // the reference still traced this shadow ray
                    }
                }

                // ---- BRDF::sample (brdf.cu:180-192) + next ray (renderer.cu:345-350) ----
                region_tick<COUNT>(n, kRegBsdf);
                f3 r;
                unsigned long long tr1 = stamp<COUNT>();
                do {
                    r.x = 2 * rng_uniform(s.rng) - 1;
                    r.y = 2 * rng_uniform(s.rng) - 1;
                    r.z = 2 * rng_uniform(s.rng) - 1;
                } while (dot(r, r) > 1);
                region_time<COUNT>(n, kRegBsdf, tr1, stamp<COUNT>());
                f3 wi = normalize(r);
                wi.z = __builtin_fabsf(wi.z);
                // both BRDF evaluations of the vertex together (the light's, renderer.cu:338-341, is only used
                // when there is a shadow ray to trace)
                const BrdfPair f_pair = brdf_eval_pair(vx.brdf, vx.wo_ts, l_ts, wi);
                const f3 f_cos = f_pair.first;
                s.f_cos_next = f_pair.second;
                if (shadow) {
                    float p_brdf = brdf_pdf();
                    f3 direct = f_cos * nee_radiance;
                    float weight = pt_div(nee_p, nee_p + p_brdf);
                    s.direct_c = (pt_div(weight, nee_p) * s.thr) * direct;
                    // A light below the vertex's horizon (cos = 0) or a black throughput makes the term an
                    // exact zero; light + 0 is light whatever the walk would find, so the walk is not taken.
                    // (NaN and infinity are not zero: those vertices keep their walk and the reference's result.)
                    if (s.direct_c.x == 0.0f && s.direct_c.y == 0.0f && s.direct_c.z == 0.0f) {
                        shadow = false;
                        if (COUNT) n.rays++; // the reference traced it
                    }
                }
                if (COUNT) {
                    n.bounces++;
                    // the reference walks the BVH once per emissive LST node for the pdf
                    n.rays += P.lst_face_nodes;
                }
                s.rd_next = tbn_mul(vx, wi);
                s.ro = position + 1e-6f * normal;
                if (shadow) {
                    s.rd = nee_dir;
                    s.mode = kModeNee;
                    s.t_init = s_init;
                    s.t_stop = s_stop;
                } else {
                    s.rd = s.rd_next;
                    s.pending = true;
                }
            }
        }
    }
    return end_path;
}

#ifndef PT_PRIO_WALK
#define PT_PRIO_WALK 1 /* s_setprio of a wave inside walk_slice ... */
#define PT_PRIO_SHADE 0 /* ... and everywhere else (render_body) */
#endif
#ifndef PT_MIN_WAVES
#define PT_MIN_WAVES 3
#endif
#ifndef PT_SWAP_EVERY
#define PT_SWAP_EVERY 5 /* samples a lane renders of one pixel before it hands the pixel back */
#endif
// A wave's tile is 8 x ROWS pixels (ROWS = 8: one pixel per lane; 16: two, see render_kernel).
constexpr int kRing = 64;      // waiting pixels never exceed 8 * ROWS - 64 <= 64
constexpr int kRingWords = 11; // pool index, 6 RNG words, 3 sum, samples done
static_assert(kBlock == 64, "render_kernel is written for one-wave workgroups");

// The wave's tile. Static order: workgroups that share blockIdx % 8 share an XCD and its L2; XCD x
// takes tile rows x, x+8, ... so every XCD sees the whole picture (balance) while horizontal
// neighbours stay on one L2. Once a launch has recorded per-tile costs, tile_order lists the
// tiles heaviest class first: the dispatcher hands out workgroups in index order, so the expensive
// tiles start early and the cheap ones fill the tail.
template <int ROWS>
PT_DEV uint32_t tile_of_wave(const KernelParams &P, uint32_t *tx, uint32_t *ty) {
    uint32_t ntx = ((uint32_t)P.width + 7u) / 8u, nty = ((uint32_t)P.height + (uint32_t)ROWS - 1u) / (uint32_t)ROWS;
    uint32_t wave = blockIdx.x;
    uint32_t x = wave % ntx, y = wave / ntx;
    if (P.tile_order) {
        uint32_t t = P.tile_order[wave];
        x = t % ntx;
        y = t / ntx;
    } else if (nty % 8u == 0) {
        uint32_t xcd = wave % 8u, i = wave / 8u;
        x = i % ntx;
        y = (i / ntx) * 8u + xcd;
    }
    *tx = x;
    *ty = y;
    return y * ntx + x;
}

template <int ROWS>
PT_DEV void pool_pixel(uint32_t tx, uint32_t ty, uint32_t q, int *px, int *py) {
    *px = (int)(tx * 8u + (q & 7u));
    *py = (int)(ty * (uint32_t)ROWS + (q >> 3));
}

// A pixel between two turns, in the wave's LDS ring: word w of position p at ring[w * kRing + p].
PT_DEV void ring_put(uint32_t *ring, uint32_t pos, uint32_t q, const Rng &g, f3 total, int sample) {
    ring[0 * kRing + pos] = q;
    ring[1 * kRing + pos] = g.v0; ring[2 * kRing + pos] = g.v1; ring[3 * kRing + pos] = g.v2;
    ring[4 * kRing + pos] = g.v3; ring[5 * kRing + pos] = g.v4; ring[6 * kRing + pos] = g.d;
    ring[7 * kRing + pos] = __float_as_uint(total.x);
    ring[8 * kRing + pos] = __float_as_uint(total.y);
    ring[9 * kRing + pos] = __float_as_uint(total.z);
    ring[10 * kRing + pos] = (uint32_t)sample;
}
PT_DEV void ring_get(const uint32_t *ring, uint32_t pos, uint32_t &q, Rng &g, f3 &total, int &sample) {
    q = ring[0 * kRing + pos];
    g.v0 = ring[1 * kRing + pos]; g.v1 = ring[2 * kRing + pos]; g.v2 = ring[3 * kRing + pos];
    g.v3 = ring[4 * kRing + pos]; g.v4 = ring[5 * kRing + pos]; g.d = ring[6 * kRing + pos];
    total = mk3(__uint_as_float(ring[7 * kRing + pos]), __uint_as_float(ring[8 * kRing + pos]), __uint_as_float(ring[9 * kRing + pos]));
    sample = (int)ring[10 * kRing + pos];
}

// RNG state of every render tile's first pixel (tx * 8, ty * rows): one lane per tile, kTileRngLanes
// tiles per wave (the work per wave is a chain of scalar row loads; more, emptier waves hide their
// latency better than 256 full ones).
constexpr uint32_t kTileRngLanes = 16;
#if !PT_FAST
__global__ void __launch_bounds__(64) tile_rng_kernel(KernelParams P, uint32_t ntx, uint32_t nty, uint32_t rows) {
    uint32_t t = blockIdx.x * kTileRngLanes + threadIdx.x;
    bool valid = threadIdx.x < kTileRngLanes && t < ntx * nty;
    uint32_t tx = valid ? t % ntx : 0u, ty = valid ? t / ntx : 0u;
    uint32_t v[5], d;
    rng_seed(v, &d, P.seed_lo, P.seed_hi);
    rng_jump(v, P.jump, valid ? ty * rows * (uint32_t)P.width + tx * 8u : 0u);
    if (valid) {
        uint32_t *out = P.tile_rng + (size_t)t * 8;
        out[0] = v[0]; out[1] = v[1]; out[2] = v[2]; out[3] = v[3]; out[4] = v[4]; out[5] = d;
    }
}
#endif // !PT_FAST

// One wave renders a tile of kPool = 8 * ROWS pixels with 64 lanes. With ROWS = 16 a lane does NOT own a pixel for the whole
// launch: after PT_SWAP_EVERY samples it swaps its pixel (RNG state, sum, count: 40 bytes) with the
// one that has waited longest in the wave's LDS ring. A pixel's samples still run one after the
// other in their own order (so every pixel's arithmetic is unchanged), but the 64 lanes stay busy
// until the tile's last few samples instead of until each lane's own pixel is complete: with one
// pixel per lane the slowest of 64 pixels took 1.27x the mean (21 % of all lane time idle).
template <bool COUNT, int STACK, int ROWS, bool WIDE>
PT_DEV void render_body(const KernelParams &P, float *__restrict__ img, unsigned long long *counters) {
    static_assert(ROWS >= 8 && ROWS <= 16, "tile rows");
    constexpr int kPool = 8 * ROWS;
    __shared__ uint32_t lds_stack[STACK * kBlock];
    __shared__ uint32_t lds_ring[ROWS > 8 ? kRingWords * kRing : 1];
    __shared__ unsigned long long lds_time[COUNT ? 8 : 1];
    LdsStack<STACK> stack;
    stack.base = lds_stack + threadIdx.x;
    const uint32_t lane = threadIdx.x;

    uint32_t tx, ty;
    const uint32_t tile = tile_of_wave<ROWS>(P, &tx, &ty);
    const unsigned long long t_tile0 = P.tile_cost ? __builtin_amdgcn_s_memtime() : 0ull;
    Tally n = {};
    if (COUNT) {
        n.w_time = lds_time;
        if (lane < 8) n.w_time[lane] = 0;
    }
    unsigned long long t_kernel0 = stamp<COUNT>();

    Path s = {};
    f3 total = mk3(0, 0, 0);
    int sample = 0, turn = 0;
    int q = -1; // the pool pixel this lane is rendering, -1 = none
    // (the pixel's coordinates follow from q and the tile: recomputed where needed, not kept in registers)
    bool need_camera = false;
    bool need_switch = true; // lane is at a turn boundary (or idle)

    // RNG init of every pixel of the tile: the first 64 start in their lanes, the rest wait in the ring
    uint32_t head = 0, tail = 0; // wave-uniform running counters (ring position = counter & (kRing - 1))
    for (uint32_t r = 0; r * 64u < (uint32_t)kPool; r++) {
        uint32_t qq = r * 64u + lane;
        int x, y;
        pool_pixel<ROWS>(tx, ty, qq, &x, &y);
        bool inside = qq < (uint32_t)kPool && x < P.width && y < P.height && P.current_samples > 0;
        Rng g;
        rng_init_from(g, P.jump, P.tile_rng + (size_t)tile * 8, inside ? (uint32_t)((y - (int)(ty * (uint32_t)ROWS)) * P.width + (x - (int)(tx * 8u))) : 0u);
        if (r == 0) {
            if (inside) {
                s.rng = g;
                q = (int)qq;
                need_camera = true;
                need_switch = false;
            }
        } else {
            unsigned long long m = __builtin_amdgcn_ballot_w64(inside);
            uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
            if (inside && ROWS > 8) ring_put(lds_ring, (tail + rank) & (kRing - 1), qq, g, mk3(0, 0, 0), 0);
            tail += (uint32_t)__builtin_popcountll(m);
        }
    }

    s.thr = mk3(1, 1, 1);
    s.rd = s.rd_next = mk3(0, 0, 1);
    s.rr_prob = 1.f;
    s.nee_face = -1;
    s.t_init = kNoHit;
    s.t_stop = -1.0f;
    s.mode = kModeMain;
    Walk wk = {};
    wk.ref = kDone;
    bool walking = false;

    for (;;) {
        // ---- turn boundary: swap the pixel with the one that has waited longest ----
        if (__builtin_amdgcn_ballot_w64(need_switch && (q >= 0 || head != tail)) != 0) {
            const bool more = q >= 0 && sample < P.current_samples; // this lane's pixel is not complete
            if (need_switch && q >= 0 && !more && img) { // complete: running-mean blend, renderer.cu:393-404
                f3 mean = total * P.exposure_scale;
                mean = mean / (float)P.current_samples;
                int total_samples = P.previous_samples + P.current_samples;
                int px, py;
                pool_pixel<ROWS>(tx, ty, (uint32_t)q, &px, &py);
                float *p = img + 3 * ((size_t)py * (size_t)P.width + (size_t)px);
                f3 last = mk3(p[0], p[1], p[2]);
                f3 next = last * pt_div((float)P.previous_samples, (float)total_samples) +
                          mean * pt_div((float)P.current_samples, (float)total_samples);
                p[0] = next.x;
                p[1] = next.y;
                p[2] = next.z;
            }
            // take (reads come before this step's writes, so a full ring may be refilled in place)
            unsigned long long m_take = __builtin_amdgcn_ballot_w64(need_switch);
            uint32_t r_take = __builtin_amdgcn_mbcnt_hi((uint32_t)(m_take >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m_take, 0));
            uint32_t n_avail = tail - head, n_take = (uint32_t)__builtin_popcountll(m_take);
            const bool got = need_switch && r_take < n_avail;
            uint32_t new_q = 0;
            Rng new_rng = s.rng;
            f3 new_total = total;
            int new_sample = sample;
            if (got && ROWS > 8) ring_get(lds_ring, (head + r_take) & (kRing - 1), new_q, new_rng, new_total, new_sample);
            head += n_take < n_avail ? n_take : n_avail;
            // give
            const bool give = got && more;
            unsigned long long m_give = __builtin_amdgcn_ballot_w64(give);
            uint32_t r_give = __builtin_amdgcn_mbcnt_hi((uint32_t)(m_give >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m_give, 0));
            if (give && ROWS > 8) ring_put(lds_ring, (tail + r_give) & (kRing - 1), (uint32_t)q, s.rng, total, sample);
            tail += (uint32_t)__builtin_popcountll(m_give);
            if (got) {
                q = (int)new_q;
                s.rng = new_rng;
                total = new_total;
                sample = new_sample;
            } else if (need_switch && !more) {
                q = -1; // completed or idle, and nothing waits
            }
            if (need_switch && q >= 0) { // a new pixel, or nobody waited and this one goes on
                turn = 0;
                need_camera = true;
                need_switch = false;
            }
        }
        if (__builtin_amdgcn_ballot_w64(q >= 0) == 0) break; // no pixel in flight and none waiting

        if (need_camera) {
            int px, py;
            pool_pixel<ROWS>(tx, ty, (uint32_t)q, &px, &py);
            PT_MARK("start_sample begin");
            start_sample<COUNT>(P, px, py, s, n);
            PT_MARK("start_sample end");
            need_camera = false;
        }

        const bool active = q >= 0;
        if (active) region_tick<COUNT>(n, kRegLoop);
        unsigned long long tl0 = stamp<COUNT>();
        if (active && !walking) { // start the walk of the ray set up by the previous iteration
            walk_begin<WIDE>(P, s.rd, s.t_init, wk);
            if (COUNT) n.walks++;
            walking = true;
        }
        // The walk issues the wave's dependent fetches (a node, then its nearest child, ...); shading is long stretches
        // of arithmetic. With the walk at the higher issue priority a wave's next fetch leaves as soon as the last one
        // is back, and the SIMD's other waves shade in the gaps: 1.4 - 1.8 % per launch on every scene measured
        // (profiles/r03_kernel_experiments.md; priorities 2 / 3, or node and leaf phases apart, give the same).
        __builtin_amdgcn_s_setprio(PT_PRIO_WALK);
        walk_slice<COUNT, WIDE>(P, s.ro, s.rd, s.t_init, s.t_stop, stack, wk, n);
        __builtin_amdgcn_s_setprio(PT_PRIO_SHADE);
        region_time<COUNT>(n, kRegLoop, tl0, stamp<COUNT>());
        bool finished = active && wk.ref == kDone && wk.pleaf == 0;
        // A speculative walk whose winner has another candidate within rounding of it starts over as a sequential
        // one (see kCullSlack). Not after an early exit: whatever is nearest there, the light is hidden.
        if (!PT_FAST && finished && s.t_init > 0.0f && !wk.h.early && wk.h.face >= 0 && wk.h.t2 <= wk.h.t * kNearHi) {
            s.t_init = -s.t_init;
            walk_reset(P.root, -s.t_init, wk);
            finished = false;
            if (COUNT) n.redone++;
        }

        if (finished && s.mode == kModeNee) { // shadow results are handled on the spot
            unsigned long long tn0 = stamp<COUNT>();
            region_tick<COUNT>(n, kRegNee);
            PT_MARK("shadow_result begin");
            shadow_result<COUNT>(wk.h, s, n);
            PT_MARK("shadow_result end");
            walking = false;
            region_time<COUNT>(n, kRegNee, tn0, stamp<COUNT>());
            continue;
        }
#if PT_SHADE_NUM > 0
        { // run the vertex handler only once enough of the wave's live lanes wait for it (wave-uniform)
            int n_fin = __builtin_popcountll(__builtin_amdgcn_ballot_w64(finished));
            int n_live = __builtin_popcountll(__builtin_amdgcn_ballot_w64(active));
            if (n_fin != n_live && n_fin * PT_SHADE_DEN < n_live * PT_SHADE_NUM) continue;
        }
#endif
        if (!finished) continue; // still walking (or idle): next slice
        unsigned long long tsh0 = stamp<COUNT>();
        walking = false;
        HitLite hit = wk.h;
        if (hit.face < 0) hit.t = kNoHit;

        // ---- main-ray result: one handler per path vertex ----
        PT_MARK("vertex_handler begin");
        const bool end_path = vertex_handler<COUNT>(P, hit, s, n);
        PT_MARK("vertex_handler end");
        region_time<COUNT>(n, kRegMain, tsh0, stamp<COUNT>());
        if (end_path) {
            total = total + s.light;
            sample++;
            turn++;
            if (sample >= P.current_samples || (ROWS > 8 && turn >= PT_SWAP_EVERY)) need_switch = true;
            else need_camera = true;
        }
    }

    if (P.tile_cost && lane == 0) { // ticks this tile kept a wave slot busy, clamped to 32 bits
        unsigned long long dt = __builtin_amdgcn_s_memtime() - t_tile0;
        P.tile_cost[tile] = dt > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)dt;
    }
    if (COUNT && counters) {
        atomicAdd(&counters[0], (unsigned long long)n.samples);
        atomicAdd(&counters[1], (unsigned long long)n.rays);
        atomicAdd(&counters[2], (unsigned long long)n.inner);
        atomicAdd(&counters[3], (unsigned long long)n.tris);
        atomicAdd(&counters[4], (unsigned long long)n.walks);
        atomicAdd(&counters[5], (unsigned long long)n.shaded);
        atomicAdd(&counters[6], (unsigned long long)n.bounces);
        for (int k = 0; k < 6; k++) {
            if (n.w_iters[k]) atomicAdd(&counters[8 + 2 * k], (unsigned long long)n.w_iters[k]);
            if (n.w_lanes[k]) atomicAdd(&counters[9 + 2 * k], (unsigned long long)n.w_lanes[k]);
        }
        for (int k = 0; k < 3; k++)
            if (n.w_idle[k]) atomicAdd(&counters[27 + k], (unsigned long long)n.w_idle[k]);
        if (n.redone) atomicAdd(&counters[30], (unsigned long long)n.redone);
        if (lane == 0) {
            for (int k = 0; k < 6; k++) atomicAdd(&counters[20 + k], n.w_time[k]);
            atomicAdd(&counters[26], stamp<COUNT>() - t_kernel0); // wave lifetime
        }
    }
}

template <bool COUNT, int STACK, int ROWS, bool WIDE>
__global__ void __launch_bounds__(kBlock, PT_MIN_WAVES) render_kernel(KernelParams P, float *__restrict__ img, unsigned long long *counters) {
    render_body<COUNT, STACK, ROWS, WIDE>(P, img, counters);
}

// The same code under its own name, for the 1-sample launch that records tile costs before the first
// batch of a scene (no image): profiles then show the batches and the probe as separate kernels.
template <int STACK, int ROWS, bool WIDE>
__global__ void __launch_bounds__(kBlock, PT_MIN_WAVES) cost_probe_kernel(KernelParams P) {
    render_body<false, STACK, ROWS, WIDE>(P, nullptr, nullptr);
}

// First camera ray of every pixel: face id + distance (parity probe), walked by the PRODUCT walker: walk_begin /
// walk_slice on the tree render launches use (WIDE or not), with the same start-over rule as render_body (a
// speculative walk whose winner has another candidate within kNearHi of it is repeated sequentially on the binary
// tree). redone counts those repeats: the "stated count of epsilon edge ties" of this launch.
template <int STACK, bool WIDE>
__global__ void __launch_bounds__(kBlock) primary_hits_kernel(KernelParams P, int32_t *__restrict__ faces, float *__restrict__ dist,
                                                              unsigned long long *__restrict__ redone) {
    __shared__ uint32_t lds_stack[STACK * kBlock];
    LdsStack<STACK> stack;
    stack.base = lds_stack + threadIdx.x;
    int px, py;
    pixel_of_thread(P, &px, &py);
    const bool inside = px < P.width && py < P.height;
    Rng rng;
    rng_init(rng, P.jump, P.seed_lo, P.seed_hi, inside ? (uint32_t)(py * P.width + px) : 0u);
    float nx, ny;
    rng_normal_pair(rng, &nx, &ny);
    if (!P.jitter) nx = ny = 0;
    float sx = (float)px + 0.33f * nx, sy = (float)py + 0.33f * ny;
    float u = pt_div(2 * sx - (float)P.width, (float)P.height);
    float v = pt_div(2 * sy - (float)P.height, (float)P.height);
    f3 o, d;
    camera_ray(P, u, v, &o, &d);
    Tally n = {};
    Walk wk = {};
    float t_init = kNoHit;
    walk_begin<WIDE>(P, d, t_init, wk);
    if (!inside) { // lanes outside the image take part in the wave's ballots with a finished walk
        wk.ref = kDone;
        wk.pleaf = 0;
    }
    uint32_t again = 0;
    for (uint32_t slices = 0;; slices++) {
        walk_slice<false, WIDE>(P, o, d, t_init, -1.0f, stack, wk, n);
        bool finished = wk.ref == kDone && wk.pleaf == 0;
        if (PT_WATCHDOG && slices > kWatchdog) { // see KernelParams::fault
            if (!finished) report_stuck(P, wk, t_init, -1, blockIdx.x);
            break;
        }
        if (!PT_FAST && finished && inside && t_init > 0.0f && wk.h.face >= 0 && wk.h.t2 <= wk.h.t * kNearHi) {
            t_init = -t_init;
            walk_reset(P.root, -t_init, wk);
            finished = false;
            again++;
        }
        if (__builtin_amdgcn_ballot_w64(!finished) == 0) break;
    }
    if (inside) {
        size_t at = (size_t)py * (size_t)P.width + (size_t)px;
        faces[at] = wk.h.face;
        if (dist) dist[at] = wk.h.face >= 0 ? wk.h.t : kNoHit;
        if (redone && again) atomicAdd(redone, (unsigned long long)again);
    }
}

#if !PT_FAST
// img[i] *= scale: a rank's running mean weighted by its share of the samples, before the sum-reduce (pt_group_render)
__global__ void __launch_bounds__(256) scale_kernel(float *__restrict__ img, size_t n, float scale) {
    size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i < n) img[i] = img[i] * scale;
}

__global__ void rng_stream_kernel(const uint32_t *jump, uint32_t seed_lo, uint32_t seed_hi, uint32_t subsequence,
                                  uint32_t *out, int count) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    Rng rng;
    rng_init(rng, jump, seed_lo, seed_hi, subsequence);
    for (int i = 0; i < count; i++) out[i] = rng_next(rng);
}
#endif // !PT_FAST

} // namespace PT_ARITH_NS
} // namespace ptc
