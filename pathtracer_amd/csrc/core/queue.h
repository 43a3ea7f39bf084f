// queue.h — the same radiance computation as render_kernel (kernel.h), with path state
// outside the lanes.
//
// render_kernel keeps one pixel per lane for a whole launch. Measured on it: of
// the lanes of a wave executing a node step only 48 % step; the others wait for the vertex
// handler, hold a finished pixel (the spread of per-pixel work inside a tile) or wait for a leaf
// phase. A lane that owns its pixel cannot do better than about half of ideal.
//
// Here a pixel's Path lives in HBM as 16-byte chunks, chunk-major ([chunk][slot]) so a wave's
// accesses coalesce, and two kernels alternate over a compacted list of live pixels:
//
//   walk_kernel   persistent waves; a lane whose walk ended writes the hit and pulls the next ray
//                 from the list (one wave-aggregated atomic per refill), so node steps run with
//                 nearly all lanes. A finished SHADOW walk is resolved on the spot (visibility,
//                 light += direct_c) and the lane goes on with the bounce ray of the same vertex.
//   shade_kernel  one lane per live pixel, all in the vertex handler at once; starts the next
//                 sample when a path ends; writes the next ray; appends surviving pixels to the
//                 other list.
//
// One iteration = one path vertex of every live pixel. The arithmetic is the code of kernel.h
// (start_sample / shadow_result / vertex_handler / walk_slice), per-pixel RNG draw order is
// unchanged, and no float sum changes its order, so the image is bit-identical.
//
// MEASURED (many_lights stand-in, 1024^2 x 50 spp): 135 ms per batch against 44 ms for
// render_kernel, so this path is opt-in (PT_KERNEL=queue) and NOT the product path. Why: a batch
// needs ~280 iterations (the pixel with the most path vertices), and every iteration ends in a
// device-wide barrier that cannot come before its longest walk has finished (150-500 us: a walk is
// a chain of ~1 us dependent fetches). The last 100 iterations hold under 1 % of the pixels and
// still cost that much each. Node-step lane utilisation does go up as predicted, but the barriers
// eat it. What would be needed instead is a barrier-free scheduler inside one persistent kernel
// (walk and shade work exchanged through device queues), with path state made visible across
// XCDs; that is round-2 material.
#pragma once

#include "kernel.h"

namespace ptc {

enum : int { qRay0 = 0, qRay1, qNext, qDirect, qLight, qThr, qFcos, qTotal, qRng0, qRng1, qChunks };

struct QueueParams {
    float4 *state;      // [qChunks][slots]
    float4 *hit;        // [slots]: t, u, v, face (bits)
    uint32_t *list[2];  // live slots of the current / next iteration
    uint32_t *cnt;      // [0],[1]: list sizes; [2]: walk cursor
    uint32_t slots;     // tiles * 64
};

PT_DEV float4 &q_at(const QueueParams &Q, int chunk, uint32_t slot) { return Q.state[(size_t)chunk * Q.slots + slot]; }

PT_DEV void slot_pixel(const KernelParams &P, uint32_t slot, int *px, int *py) {
    uint32_t ntx = ((uint32_t)P.width + 7u) / 8u;
    uint32_t tile = slot >> 6, lane = slot & 63u;
    *px = (int)((tile % ntx) * 8u + (lane & 7u));
    *py = (int)((tile / ntx) * 8u + (lane >> 3));
}

PT_DEV float4 pack(f3 a, float w) { return make_float4(a.x, a.y, a.z, w); }
PT_DEV f3 xyz(float4 a) { return mk3(a.x, a.y, a.z); }

// Everything the next walk and the next vertex need, written by init and shade.
PT_DEV void save_path(const QueueParams &Q, uint32_t slot, const Path &s, f3 total, int sample) {
    q_at(Q, qRay0, slot) = pack(s.ro, s.t_init);
    q_at(Q, qRay1, slot) = pack(s.rd, s.t_stop);
    if (s.mode == kModeNee) {
        q_at(Q, qNext, slot) = pack(s.rd_next, s.nee_dist);
        q_at(Q, qDirect, slot) = pack(s.direct_c, __uint_as_float(((uint32_t)s.nee_kind << 30) | ((uint32_t)s.nee_face & 0x3FFFFFFFu)));
    }
    q_at(Q, qLight, slot) = pack(s.light, s.rr_prob);
    q_at(Q, qThr, slot) = pack(s.thr, __uint_as_float((s.pending ? 1u : 0u) | ((uint32_t)sample << 1)));
    q_at(Q, qFcos, slot) = pack(s.f_cos_next, 0.f);
    q_at(Q, qTotal, slot) = pack(total, 0.f);
    q_at(Q, qRng0, slot) = make_float4(__uint_as_float(s.rng.v0), __uint_as_float(s.rng.v1), __uint_as_float(s.rng.v2), __uint_as_float(s.rng.v3));
    q_at(Q, qRng1, slot) = make_float4(__uint_as_float(s.rng.v4), __uint_as_float(s.rng.d), 0.f, 0.f);
}

// Wave-aggregated append of the lanes with `keep` to list[which]; all lanes of the wave call it.
PT_DEV void append_slot(const QueueParams &Q, int which, bool keep, uint32_t slot) {
    unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
    if (m == 0) return;
    uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
    int leader = __builtin_ctzll(m);
    uint32_t base = 0;
    if (keep && rank == 0) base = atomicAdd(&Q.cnt[which], (uint32_t)__builtin_popcountll(m));
    base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
    if (keep) Q.list[which][base + rank] = slot;
}

// One lane per pixel slot: RNG init, first camera ray, list[0] = the pixels inside the image.
__global__ void __launch_bounds__(64) queue_init_kernel(KernelParams P, QueueParams Q) {
    uint32_t slot = blockIdx.x * 64u + threadIdx.x;
    int px, py;
    slot_pixel(P, slot, &px, &py);
    bool inside = px < P.width && py < P.height;
    Tally n = {};
    Path s = {};
    rng_init(s.rng, P.jump, P.seed_lo, P.seed_hi, inside ? (uint32_t)(py * P.width + px) : 0u);
    s.thr = mk3(1, 1, 1);
    s.rd = s.rd_next = mk3(0, 0, 1);
    s.rr_prob = 1.f;
    s.nee_face = -1;
    bool alive = inside && P.current_samples > 0;
    if (alive) {
        start_sample<false>(P, px, py, s, n);
        save_path(Q, slot, s, mk3(0, 0, 0), 0);
    } else {
        q_at(Q, qTotal, slot) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    append_slot(Q, 0, alive, slot);
}

#ifndef PT_REFILL_MIN
#define PT_REFILL_MIN 1 /* idle lanes a wave collects before it pulls new rays */
#endif
#ifndef PT_WALK_BLOCK
#define PT_WALK_BLOCK 256 /* list positions a wave reserves per atomic */
#endif
#ifndef PT_WALK_WAVES
#define PT_WALK_WAVES 5
#endif

template <int STACK>
__global__ void __launch_bounds__(64, PT_WALK_WAVES) queue_walk_kernel(KernelParams P, QueueParams Q, int cur) {
    __shared__ uint32_t lds_stack[STACK * 64];
    uint32_t *stack = lds_stack + threadIdx.x;
    const uint32_t n_rays = Q.cnt[cur];
    if (blockIdx.x == 0 && threadIdx.x == 0) Q.cnt[cur ^ 1] = 0; // the list shade_kernel appends to next
    const uint32_t *list = Q.list[cur];

    Tally n = {};
    int slot = -1;
    f3 ro = mk3(0, 0, 0), rd = mk3(0, 0, 1);
    float t_stop = -1.0f, t_init = kNoHit; // t_init goes negative while the walk is a sequential one (kernel.h, kCullSlack)
    Walk wk = {};
    wk.ref = kDone;
    bool exhausted = false; // wave-uniform
    // The wave draws list positions from the shared cursor in blocks (one atomic per
    // PT_WALK_BLOCK rays: per-ray atomics on one address serialise in L2 and were 4x the walk
    // itself) and hands them to its lanes from this wave-local reserve.
    uint32_t res_next = 0, res_end = 0; // wave-uniform
    for (;;) {
        unsigned long long m_empty = __builtin_amdgcn_ballot_w64(slot < 0);
        int n_empty = __builtin_popcountll(m_empty);
        if (!exhausted && (n_empty >= PT_REFILL_MIN || n_empty == 64) && n_empty != 0) {
            if (res_next == res_end) {
                uint32_t base = 0;
                if (threadIdx.x == 0) base = atomicAdd(&Q.cnt[2], (uint32_t)PT_WALK_BLOCK);
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                res_next = base < n_rays ? base : n_rays;
                res_end = base + PT_WALK_BLOCK < n_rays ? base + PT_WALK_BLOCK : n_rays;
                if (res_next == res_end) exhausted = true;
            }
            uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m_empty >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m_empty, 0));
            if (slot < 0 && res_next + rank < res_end) {
                slot = (int)list[res_next + rank];
                float4 r0 = q_at(Q, qRay0, (uint32_t)slot), r1 = q_at(Q, qRay1, (uint32_t)slot);
                ro = xyz(r0);
                rd = xyz(r1);
                t_stop = r1.w;
                t_init = r0.w;
                walk_begin<false>(P, rd, t_init, wk);
            }
            res_next = res_next + (uint32_t)n_empty < res_end ? res_next + (uint32_t)n_empty : res_end;
        }
        if (__builtin_amdgcn_ballot_w64(slot >= 0) == 0) {
            if (exhausted) break;
            continue;
        }
        walk_slice<false, STACK, false>(P, ro, rd, t_init, t_stop, stack, wk, n);
        if (slot >= 0 && wk.ref == kDone && wk.pleaf == 0 && t_init > 0.0f && !wk.h.early && wk.h.face >= 0 &&
            wk.h.t2 <= wk.h.t * kNearHi) { // see kCullSlack in kernel.h: once more, sequentially
            t_init = -t_init;
            walk_reset(P.root, -t_init, wk);
        }
        if (slot >= 0 && wk.ref == kDone && wk.pleaf == 0) {
            if (t_stop != -1.0f) {
                // shadow ray: visibility, then the bounce ray of the same vertex (same origin)
                float4 nx = q_at(Q, qNext, (uint32_t)slot), dc = q_at(Q, qDirect, (uint32_t)slot);
                uint32_t packed = __float_as_uint(dc.w);
                int kind = (int)(packed >> 30);
                int32_t face = (int32_t)(packed & 0x3FFFFFFFu);
                bool visible;
                if (kind == 0) {
                    float closest = wk.h.face >= 0 ? wk.h.t : kNoHit;
                    visible = !wk.h.early && nx.w < closest;
                } else if (kind == 1) {
                    visible = wk.h.face < 0;
                } else {
                    visible = !wk.h.early && wk.h.face == face && wk.h.t > 0;
                }
                if (visible) {
                    float4 l = q_at(Q, qLight, (uint32_t)slot);
                    f3 sum = xyz(l) + xyz(dc);
                    q_at(Q, qLight, (uint32_t)slot) = pack(sum, l.w);
                }
                rd = xyz(nx);
                t_stop = -1.0f;
                q_at(Q, qRay1, (uint32_t)slot) = pack(rd, -1.0f); // shade_kernel reads the direction it hit along
                t_init = kNoHit;
                walk_begin<false>(P, rd, t_init, wk);
            } else {
                float t = wk.h.face < 0 ? kNoHit : wk.h.t;
                Q.hit[(uint32_t)slot] = make_float4(t, wk.h.u, wk.h.v, __int_as_float(wk.h.face));
                slot = -1;
            }
        }
    }
}

// One lane per live pixel: the vertex handler, the next sample when the path ended, the next ray.
__global__ void __launch_bounds__(64, 4) queue_shade_kernel(KernelParams P, QueueParams Q, int cur) {
    const uint32_t n_live = Q.cnt[cur];
    if (blockIdx.x == 0 && threadIdx.x == 0) Q.cnt[2] = 0; // walk cursor of the next iteration
    uint32_t i = blockIdx.x * 64u + threadIdx.x;
    bool alive = i < n_live;
    uint32_t slot = 0;
    if (alive) {
        slot = Q.list[cur][i];
        Tally n = {};
        Path s;
        float4 hv = Q.hit[slot];
        HitLite hit;
        hit.t = hv.x;
        hit.u = hv.y;
        hit.v = hv.z;
        hit.face = __float_as_int(hv.w);
        hit.early = false;
        float4 r1 = q_at(Q, qRay1, slot), li = q_at(Q, qLight, slot), th = q_at(Q, qThr, slot), fc = q_at(Q, qFcos, slot);
        float4 g0 = q_at(Q, qRng0, slot), g1 = q_at(Q, qRng1, slot);
        s.rng.v0 = __float_as_uint(g0.x); s.rng.v1 = __float_as_uint(g0.y); s.rng.v2 = __float_as_uint(g0.z); s.rng.v3 = __float_as_uint(g0.w);
        s.rng.v4 = __float_as_uint(g1.x); s.rng.d = __float_as_uint(g1.y);
        s.ro = mk3(0, 0, 0);
        s.rd = xyz(r1);
        s.t_init = kNoHit;
        s.t_stop = -1.0f;
        s.mode = kModeMain;
        s.light = xyz(li);
        s.rr_prob = li.w;
        s.thr = xyz(th);
        uint32_t bits = __float_as_uint(th.w);
        s.pending = (bits & 1u) != 0;
        int sample = (int)(bits >> 1);
        s.f_cos_next = xyz(fc);
        s.rd_next = mk3(0, 0, 1);
        s.direct_c = mk3(0, 0, 0);
        s.nee_dist = 0.f;
        s.nee_kind = 0;
        s.nee_face = -1;
        f3 total = xyz(q_at(Q, qTotal, slot));

        bool end_path = vertex_handler<false>(P, hit, s, n);
        if (s.mode == kModeNee) s.pending = true; // shadow_result would set it before the bounce ray's hit
        if (end_path) {
            total = total + s.light;
            sample++;
            if (sample >= P.current_samples) {
                alive = false;
                q_at(Q, qTotal, slot) = pack(total, 0.f);
            } else {
                int px, py;
                slot_pixel(P, slot, &px, &py);
                start_sample<false>(P, px, py, s, n);
            }
        }
        if (alive) save_path(Q, slot, s, total, sample);
    }
    append_slot(Q, cur ^ 1, alive, slot);
}

// Running-mean blend of the finished pixels into the image (renderer.cu:393-404).
__global__ void __launch_bounds__(64) queue_final_kernel(KernelParams P, QueueParams Q, float *__restrict__ img) {
    uint32_t slot = blockIdx.x * 64u + threadIdx.x;
    int px, py;
    slot_pixel(P, slot, &px, &py);
    if (px >= P.width || py >= P.height) return;
    f3 total = xyz(q_at(Q, qTotal, slot));
    total = total * P.exposure_scale;
    total = total / (float)P.current_samples;
    int total_samples = P.previous_samples + P.current_samples;
    float *p = img + 3 * ((size_t)py * (size_t)P.width + (size_t)px);
    f3 last = mk3(p[0], p[1], p[2]);
    f3 next = last * ((float)P.previous_samples / (float)total_samples) + total * ((float)P.current_samples / (float)total_samples);
    p[0] = next.x;
    p[1] = next.y;
    p[2] = next.z;
}

} // namespace ptc
