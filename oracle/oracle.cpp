// oracle.cpp — CPU restatement of the reference's per-sample radiance kernel.
// TEST INFRASTRUCTURE ONLY (see oracle.h). PARITY UNPINNED for the first-party
// logic: no golden vectors exist upstream and the reference cannot be built in
// this image; every function cites the reference lines it follows.
//
// Build: g++ -O2 -ffp-contract=off (no FMA contraction: the HIP kernel is built
// the same way and the two are compared bit for bit).
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "oracle.h"
#include "pt_detmath.h"

namespace {

// ---- vector algebra, operation order of src/mathops.h -----------------------------

struct V3 {
    float x, y, z;
};
inline V3 mk(float x, float y, float z) { return V3{x, y, z}; }
inline V3 from(const pt_vec3 &v) { return V3{v.x, v.y, v.z}; }
inline V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }      // :90
inline V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }      // :98
inline V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }                           // :94
inline V3 operator*(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }      // :102
inline V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }         // :106
inline V3 operator*(float s, V3 a) { return a * s; }                                 // :249
inline float inv(float s) { return (float)(1.0 / (double)s); }                       // DIV(1.0, s), :115
inline V3 operator/(V3 a, float s) { return a * inv(s); }                            // :114
inline float dot(V3 a, V3 b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }         // :119
inline V3 cross(V3 a, V3 b) {                                                        // :123
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
inline float mag(V3 a) { return sqrtf(dot(a, a)); }                                  // :129
inline V3 unit(V3 a) { return a * inv(mag(a)); }                                     // :133
inline V3 vmin(V3 a, V3 b) { return mk(a.x < b.x ? a.x : b.x, a.y < b.y ? a.y : b.y, a.z < b.z ? a.z : b.z); }
inline V3 vmax(V3 a, V3 b) { return mk(a.x > b.x ? a.x : b.x, a.y > b.y ? a.y : b.y, a.z > b.z ? a.z : b.z); }
inline float min_comp(V3 a) { return (a.x < a.y && a.x < a.z) ? a.x : (a.y < a.z ? a.y : a.z); } // :188
inline float max_comp(V3 a) { return (a.x > a.y && a.x > a.z) ? a.x : (a.y > a.z ? a.y : a.z); } // :192
inline float luminance(V3 c) { return c.x * 0.2126f + c.y * 0.7152f + c.z * 0.0722f; } // :654

struct M3 {
    float m[3][3];
};
inline M3 from_columns(V3 a, V3 b, V3 c) {                                            // :211
    M3 r;
    r.m[0][0] = a.x; r.m[0][1] = b.x; r.m[0][2] = c.x;
    r.m[1][0] = a.y; r.m[1][1] = b.y; r.m[1][2] = c.y;
    r.m[2][0] = a.z; r.m[2][1] = b.z; r.m[2][2] = c.z;
    return r;
}
inline M3 transposed(const M3 &a) {                                                   // :46
    M3 r;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) r.m[i][j] = a.m[j][i];
    return r;
}
inline V3 operator*(const M3 &a, V3 v) {                                              // :264
    return mk((a.m[0][0] * v.x + a.m[0][1] * v.y) + a.m[0][2] * v.z,
              (a.m[1][0] * v.x + a.m[1][1] * v.y) + a.m[1][2] * v.z,
              (a.m[2][0] * v.x + a.m[2][1] * v.y) + a.m[2][2] * v.z);
}

struct Ray {
    V3 o, r;
};

// ---- XORWOW (cuRAND's default generator; src/random.cu:3-13) -----------------------

struct Jump {
    // mat[k][bit][word]: image of input bit `bit` under A^(2^67 * 2^k)
    uint32_t mat[64][160][5];
    int built = 0;
};

void step_vec(uint32_t v[5]) {
    uint32_t t = v[0] ^ (v[0] >> 2);
    v[0] = v[1];
    v[1] = v[2];
    v[2] = v[3];
    v[3] = v[4];
    v[4] = (v[4] ^ (v[4] << 4)) ^ (t ^ (t << 1));
}

void mat_apply(const uint32_t m[160][5], const uint32_t in[5], uint32_t out[5]) {
    uint32_t r[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < 5; i++)
        for (int j = 0; j < 32; j++)
            if (in[i] & (1u << j))
                for (int k = 0; k < 5; k++) r[k] ^= m[i * 32 + j][k];
    memcpy(out, r, sizeof(r));
}

void mat_square(const uint32_t a[160][5], uint32_t out[160][5]) {
    for (int b = 0; b < 160; b++) mat_apply(a, a[b], out[b]);
}

Jump *g_jump = nullptr;
std::once_flag g_jump_once;

void build_jump() {
    Jump *J = new Jump();
    static uint32_t cur[160][5], nxt[160][5];
    for (int b = 0; b < 160; b++) {
        uint32_t v[5] = {0, 0, 0, 0, 0};
        v[b / 32] = 1u << (b % 32);
        step_vec(v);
        memcpy(cur[b], v, sizeof(v));
    }
    for (int s = 0; s < 67; s++) { // A -> A^(2^67)
        mat_square(cur, nxt);
        memcpy(cur, nxt, sizeof(cur));
    }
    for (int k = 0; k < 64; k++) {
        memcpy(J->mat[k], cur, sizeof(cur));
        mat_square(cur, nxt);
        memcpy(cur, nxt, sizeof(cur));
    }
    J->built = 64;
    g_jump = J;
}

struct Rng {
    uint32_t v[5];
    uint32_t d;
    bool have_extra;
    float extra;
};

// The four constants of curand_init's seed scramble, as recalled from the public curand_kernel.h (UNVERIFIED:
// cuRAND is not in this image). orc_rng_set_scramble swaps in another set for the cross-check against rocRAND's
// engine, whose structure is the same and whose constants differ (rocrand_xorwow.h:113-116).
uint32_t g_scramble[4] = {0xaad26b49u, 0xf7dcefddu, 1099087573u, 2591861531u};

// curand_init(seed, subsequence, 0, &state) for curandStateXORWOW_t.
void rng_init(Rng &s, uint64_t seed, uint64_t subsequence) {
    std::call_once(g_jump_once, build_jump);
    uint32_t s0 = (uint32_t)seed ^ g_scramble[0];
    uint32_t s1 = (uint32_t)(seed >> 32) ^ g_scramble[1];
    uint32_t t0 = g_scramble[2] * s0;
    uint32_t t1 = g_scramble[3] * s1;
    s.d = 6615241u + t1 + t0;
    s.v[0] = 123456789u + t0;
    s.v[1] = 362436069u ^ t0;
    s.v[2] = 521288629u + t1;
    s.v[3] = 88675123u ^ t1;
    s.v[4] = 5783321u + t0;
    for (int k = 0; k < 64 && subsequence; k++, subsequence >>= 1)
        if (subsequence & 1) mat_apply(g_jump->mat[k], s.v, s.v);
    s.have_extra = false;
    s.extra = 0.f;
}

inline uint32_t rng_next(Rng &s) {
    step_vec(s.v);
    s.d += 362437u;
    return s.v[4] + s.d;
}

// curand_uniform: (0, 1]
inline float rng_uniform(Rng &s) { return (float)rng_next(s) * 2.3283064e-10f + (2.3283064e-10f / 2.0f); }

// curand_normal: Box-Muller on two draws, sine half first, cosine half cached.
inline float rng_normal(Rng &s) {
    if (s.have_extra) {
        s.have_extra = false;
        return s.extra;
    }
    uint32_t x = rng_next(s), y = rng_next(s);
    const float two_pi_scale = 2.3283064e-10f * 6.2831855f;
    float u = (float)x * 2.3283064e-10f + (2.3283064e-10f / 2.0f);
    float v = (float)y * two_pi_scale + (two_pi_scale / 2.0f);
    float r = sqrtf(-2.0f * pt_logf(u));
    float sn, cs;
    pt_sincosf(v, &sn, &cs);
    s.extra = r * cs;
    s.have_extra = true;
    return r * sn;
}

// src/random.cu:15-24
V3 sphere_sample_uniform(Rng &s) {
    V3 r;
    do {
        r.x = 2 * rng_uniform(s) - 1;
        r.y = 2 * rng_uniform(s) - 1;
        r.z = 2 * rng_uniform(s) - 1;
    } while (dot(r, r) > 1);
    return unit(r);
}

// ---- textures (role of tex2D<float4>, src/scene.cu:1024-1031) ----------------------

inline float wrap_coord(float u, int mode) {
    if (mode == PT_WRAP_CLAMP) return u < 0.0f ? 0.0f : (u > 1.0f ? 1.0f : u);
    if (mode == PT_WRAP_MIRROR) {
        float t = u - 2.0f * floorf(u * 0.5f); // [0, 2)
        return t > 1.0f ? 2.0f - t : t;
    }
    return u - floorf(u);
}
inline int wrap_index(int i, int n, int mode) {
    if (mode == PT_WRAP_REPEAT) {
        i %= n;
        return i < 0 ? i + n : i;
    }
    return i < 0 ? 0 : (i >= n ? n - 1 : i);
}
inline void fetch_texel(const pt_texture *t, int x, int y, float out[4]) {
    size_t at = ((size_t)y * (size_t)t->width + (size_t)x) * 4;
    if (t->isFloat) {
        const float *p = (const float *)t->pixels + at;
        out[0] = p[0]; out[1] = p[1]; out[2] = p[2]; out[3] = p[3];
    } else {
        const uint8_t *p = (const uint8_t *)t->pixels + at;
        for (int k = 0; k < 4; k++) out[k] = (float)p[k] / 255.0f;
    }
}
void sample_texture(const pt_texture *t, float u, float v, float out[4]) {
    float fu = wrap_coord(u, t->wrapS) * (float)t->width;
    float fv = wrap_coord(v, t->wrapT) * (float)t->height;
    if (t->filter == PT_FILTER_POINT) {
        fetch_texel(t, wrap_index((int)floorf(fu), t->width, t->wrapS), wrap_index((int)floorf(fv), t->height, t->wrapT), out);
        return;
    }
    float xb = fu - 0.5f, yb = fv - 0.5f;
    float x0 = floorf(xb), y0 = floorf(yb);
    // 8 fractional bits, like the texture unit's interpolation weights
    float a = floorf((xb - x0) * 256.0f + 0.5f) / 256.0f;
    float b = floorf((yb - y0) * 256.0f + 0.5f) / 256.0f;
    int ix0 = wrap_index((int)x0, t->width, t->wrapS), ix1 = wrap_index((int)x0 + 1, t->width, t->wrapS);
    int iy0 = wrap_index((int)y0, t->height, t->wrapT), iy1 = wrap_index((int)y0 + 1, t->height, t->wrapT);
    float t00[4], t10[4], t01[4], t11[4];
    fetch_texel(t, ix0, iy0, t00);
    fetch_texel(t, ix1, iy0, t10);
    fetch_texel(t, ix0, iy1, t01);
    fetch_texel(t, ix1, iy1, t11);
    float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
    for (int k = 0; k < 4; k++) out[k] = ((w00 * t00[k] + w10 * t10[k]) + w01 * t01[k]) + w11 * t11[k];
}

// ---- hit record and intersection (src/intersect.h:14-23, src/intersect.cu) ----------

struct Brdf {
    V3 baseColor;
    float roughness, metallic, specular;
};

struct Hit {
    bool has_hit = false;
    float distance = 1e30f;
    V3 position, shaded_normal, incident_normal, true_normal, texcoord0;
    M3 tangentBasis;
    int faceIndex = -1;
    const pt_material *mat = nullptr;
    Brdf brdf;
    float alpha;
};

struct Ctx {
    const orc_inputs *in;
    const pt_scene_desc *sc;
    Rng rng;
    orc_counters n{};
    orc_nonfinite_diag *diag = nullptr; // orc_find_nonfinite only
    orc_margin_stats *probe = nullptr;  // orc_margin_probe only
    int sample_index = 0;
};

inline bool finite3(V3 a) { return std::isfinite(a.x) && std::isfinite(a.y) && std::isfinite(a.z); }

// intersect.cu:12-81, non-culling branch
int moeller_trumbore(const Ray &ray, V3 v0, V3 v1, V3 v2, float *t, float *u, float *v) {
    V3 edge1 = v1 - v0, edge2 = v2 - v0;
    V3 pvec = cross(ray.r, edge2);
    float det = dot(edge1, pvec);
    if ((double)det > -1e-12 && (double)det < 1e-12) return 0;
    float inv_det = (float)(1.0 / (double)det);
    V3 tvec = ray.o - v0;
    *u = dot(tvec, pvec) * inv_det;
    if ((double)*u < -1e-12 || (double)*u > (1.0 + 1e-12)) return 0;
    V3 qvec = cross(tvec, edge1);
    *v = dot(ray.r, qvec) * inv_det;
    if ((double)*v < -1e-12 || (double)(*u + *v) > (1.0 + 1e-12)) return 0;
    *t = dot(edge2, qvec) * inv_det;
    return 1;
}

inline V3 bary(V3 A, V3 B, V3 C, float t, float u, float v) { return (t * A + u * B) + v * C; } // intersect.cu:5-8

// intersect.cu:83-212
void intersect_face(Ctx &c, const Ray &ray, Hit &hit, uint32_t faceIndex) {
    const pt_scene_desc *sc = c.sc;
    const pt_face &face = sc->faces[faceIndex];
    for (uint32_t i = 2; i < face.vertexCount; i++) {
        const pt_vertex &A = sc->vertices[face.vertices[0]];
        const pt_vertex &B = sc->vertices[face.vertices[i - 1]];
        const pt_vertex &C = sc->vertices[face.vertices[i]];
        float t, u, v;
        c.n.tri_tests++;
        int has = moeller_trumbore(ray, from(A.position), from(B.position), from(C.position), &t, &u, &v);
        if (!(has && t >= 0 && t < hit.distance)) continue;
        c.n.hit_records++;
        hit.faceIndex = (int)faceIndex;
        hit.has_hit = true;
        hit.distance = t;
        hit.mat = &sc->materials[face.material];
        const pt_material &mat = *hit.mat;

        float w = (float)(1.0 - (double)u - (double)v);
        hit.position = bary(from(A.position), from(B.position), from(C.position), w, u, v);
        hit.texcoord0 = bary(from(A.texcoord0), from(B.texcoord0), from(C.texcoord0), w, u, v);

        V3 baseColor = mk(mat.baseColorFactor.x, mat.baseColorFactor.y, mat.baseColorFactor.z);
        hit.alpha = mat.baseColorFactor.w;
        if (mat.baseColorTexture >= 0) {
            float px[4];
            sample_texture(&sc->textures[mat.baseColorTexture], hit.texcoord0.x, hit.texcoord0.y, px);
            baseColor = baseColor * mk(px[0], px[1], px[2]);
            hit.alpha *= px[3];
        }
        hit.brdf.baseColor = baseColor;
        hit.brdf.specular = mat.specular;
        hit.brdf.metallic = mat.metallicFactor;
        hit.brdf.roughness = mat.roughnessFactor;
        if (mat.metallicRoughnessTexture >= 0) {
            float px[4];
            sample_texture(&sc->textures[mat.metallicRoughnessTexture], hit.texcoord0.x, hit.texcoord0.y, px);
            hit.brdf.roughness *= px[1];
            hit.brdf.metallic *= px[2];
        }
        switch (mat.alphaMode) {
        case PT_ALPHA_OPAQUE: hit.alpha = 1.0f; break;
        case PT_ALPHA_MASK: hit.alpha = hit.alpha > mat.alphaCutoff ? 1.0f : 0.0f; break;
        default: break;
        }

        V3 normal, tangent;
        float handed = 1;
        if (face.shading == PT_FLAT_SHADING) {
            normal = from(face.normal);
            tangent = mk(face.tangent.x, face.tangent.y, face.tangent.z);
            handed = face.tangent.w;
        } else {
            normal = bary(from(A.normal), from(B.normal), from(C.normal), w, u, v);
            tangent = bary(mk(A.tangent.x, A.tangent.y, A.tangent.z), mk(B.tangent.x, B.tangent.y, B.tangent.z),
                           mk(C.tangent.x, C.tangent.y, C.tangent.z), w, u, v);
            handed = A.tangent.w;
        }
        handed = handed < 0 ? -1.0f : 1.0f;
        hit.true_normal = normal;
        V3 bitangent = handed * cross(normal, tangent);
        tangent = unit(tangent);
        bitangent = unit(bitangent);
        normal = unit(normal);
        if (dot(normal, ray.r) > 0) {
            tangent = tangent * -1.0f;
            bitangent = bitangent * -1.0f;
            normal = normal * -1.0f;
        }
        hit.incident_normal = normal;
        hit.tangentBasis = from_columns(tangent, bitangent, normal);
        if (mat.normalTexture >= 0) {
            float px[4];
            sample_texture(&sc->textures[mat.normalTexture], hit.texcoord0.x, hit.texcoord0.y, px);
            V3 n = mk(2 * px[0] + -1.0f, 2 * px[1] + -1.0f, 2 * px[2] + -1.0f);
            hit.shaded_normal = unit(hit.tangentBasis * n);
        } else {
            hit.shaded_normal = normal;
        }
    }
}

// bvh.cu:222-237
inline float intersect_aabb(const pt_bvh_node &n, const Ray &ray, V3 r_inv, float min_t) {
    V3 t1 = (from(n.bmin) - ray.o) * r_inv;
    V3 t2 = (from(n.bmax) - ray.o) * r_inv;
    float tmin = max_comp(vmin(t1, t2));
    float tmax = min_comp(vmax(t1, t2));
    if (tmax >= tmin && tmin < min_t && tmax > 0) return tmin;
    return 1e30f;
}

// orc_margin_probe: the traversal is about to apply the box test of leaf `leaf` against the best distance `best`
void probe_leaf(Ctx &c, const Ray &ray, V3 r_inv, const pt_bvh_node &leaf, float best, float *t_first, float *t_second) {
    orc_margin_stats &m = *c.probe;
    m.leaf_looks++;
    V3 t1 = (from(leaf.bmin) - ray.o) * r_inv, t2 = (from(leaf.bmax) - ray.o) * r_inv;
    float tmin = max_comp(vmin(t1, t2)), tmax = min_comp(vmax(t1, t2));
    const bool geometric = tmax >= tmin && tmax > 0; // the box test but for the distance cull
    for (uint32_t i = leaf.start; i < leaf.end; i++) {
        const pt_face &face = c.sc->faces[c.in->indices[i]];
        float t, u, v;
        if (!moeller_trumbore(ray, from(c.sc->vertices[face.vertices[0]].position), from(c.sc->vertices[face.vertices[1]].position),
                              from(c.sc->vertices[face.vertices[2]].position), &t, &u, &v)) continue;
        if (!(t >= 0 && t < best)) continue;
        m.hits++;
        if (t < *t_first) { *t_second = *t_first; *t_first = t; } else if (t < *t_second) *t_second = t;
        const double td = t > 0 ? (double)t : 1e-30;
        if (!geometric) {
            m.missed_box++;
            double rim = std::min(std::min((double)u, (double)v), 1.0 - ((double)u + (double)v));
            if (rim > m.max_rim_of_missed) m.max_rim_of_missed = rim;
            if (std::isinf(r_inv.x) || std::isinf(r_inv.y) || std::isinf(r_inv.z)) {
                m.missed_box_inf_rinv++;
            } else {
                if (rim > m.max_rim_of_missed_finite) m.max_rim_of_missed_finite = rim;
                if (t > m.max_t_of_missed_finite) m.max_t_of_missed_finite = t;
            }
            continue;
        }
        if (tmin > t) {
            m.entry_behind++;
            m.max_entry_behind = std::max(m.max_entry_behind, ((double)tmin - t) / td);
        }
        if (tmax < t) {
            m.exit_in_front++;
            m.max_exit_in_front = std::max(m.max_exit_in_front, ((double)t - tmax) / td);
        }
        if (!(tmin < best)) {
            m.culled_by_distance++;
            m.max_culled_gap = std::max(m.max_culled_gap, ((double)tmin - t) / td);
        }
    }
}

// bvh.cu:239-286
void intersect(Ctx &c, const Ray &ray, Hit &hit) {
    c.n.rays++;
    const pt_bvh_node *nodes = c.in->nodes;
    const uint32_t *indices = c.in->indices;
    V3 r_inv = mk(1.0f / ray.r.x, 1.0f / ray.r.y, 1.0f / ray.r.z); // Vec3::Const(1.0f) / r, mathops.h:253
    const pt_bvh_node *stack[64];
    int depth = 0;
    stack[0] = &nodes[0];
    float probe_first = 1e30f, probe_second = 1e30f;
    struct ProbeEnd { // at the end of the walk: the runner-up's gap
        Ctx &c; float &a, &b;
        ~ProbeEnd() {
            if (!c.probe) return;
            c.probe->rays++;
            if (b >= 1e30f || !(a > 0)) return;
            c.probe->rays_with_two_candidates++;
            double gap = ((double)b - a) / a;
            for (int k = 0; k < 16; k++)
                if (gap < std::ldexp(1.0, -(k + 8))) c.probe->gap_below[k]++;
        }
    } probe_end{c, probe_first, probe_second};
    while (depth >= 0) {
        const pt_bvh_node *node = stack[depth--];
        while (node) {
            if (node->end - node->start > 0) {
                for (uint32_t i = node->start; i < node->end; i++) intersect_face(c, ray, hit, indices[i]);
                node = nullptr;
            } else {
                c.n.inner_visits++;
                const pt_bvh_node *c1 = &nodes[node->leftChild];
                const pt_bvh_node *c2 = &nodes[node->rightChild];
                if (c.probe) {
                    if (c1->end - c1->start > 0) probe_leaf(c, ray, r_inv, *c1, hit.distance, &probe_first, &probe_second);
                    if (c2->end - c2->start > 0) probe_leaf(c, ray, r_inv, *c2, hit.distance, &probe_first, &probe_second);
                }
                float t1 = intersect_aabb(*c1, ray, r_inv, hit.distance);
                float t2 = intersect_aabb(*c2, ray, r_inv, hit.distance);
                if (t2 < t1) {
                    std::swap(c1, c2);
                    std::swap(t1, t2);
                }
                if (t1 == 1e30f) {
                    node = nullptr;
                } else {
                    node = c1;
                    if (t2 != 1e30f) stack[++depth] = c2;
                }
            }
        }
    }
}

// ---- BRDF (src/brdf.cu:3-56, 164-192) ---------------------------------------------------

V3 fresnel_schlick(V3 F0, V3 v, V3 h) {
    float x = 1 - dot(v, h);
    return F0 + ((((((mk(1, 1, 1) - F0) * x) * x) * x) * x) * x);
}
float ggx_d(V3 h, float roughness) {
    float n_dot_h = h.z;
    float alpha = roughness * roughness;
    float b = n_dot_h * n_dot_h * (alpha * alpha - 1.0f) + 1.0f;
    return alpha * alpha / (PT_PI_F * b * b);
}
float g1_schlick_ggx(V3 v, float roughness) {
    float alpha = roughness * roughness;
    float k = (float)(0.5 * (double)alpha);
    float n_dot_v = v.z;
    return n_dot_v / (n_dot_v * (1.0f - k) + k);
}
V3 brdf_microfacet(V3 l, V3 v, const Brdf &b) {
    V3 h = unit(l + v);
    float f0d = (float)(0.16 * (double)b.specular * (double)b.specular);
    V3 F0_diel = mk(1, 1, 1) * f0d;
    V3 F0 = (b.metallic * b.baseColor) + ((1 - b.metallic) * F0_diel);
    V3 F = fresnel_schlick(F0, v, h);
    float D = ggx_d(h, b.roughness);
    float G = g1_schlick_ggx(l, b.roughness) * g1_schlick_ggx(v, b.roughness);
    float n_dot_v = fmaxf(v.z, 0.001f);
    float n_dot_l = fmaxf(l.z, 0.001f);
    V3 f_spec = ((F * D) * G) / (4.0f * n_dot_v * n_dot_l);
    V3 f_diff = ((b.baseColor * (mk(1, 1, 1) - F)) * (float)(1.0 - (double)b.metallic)) * PT_1_PI_F;
    return f_diff + f_spec;
}
V3 brdf_eval(const Brdf &b, V3 wo, V3 wi) {
    float cos_theta_wi = fmaxf(wi.z, 0);
    return brdf_microfacet(wi, wo, b) * cos_theta_wi;
}
inline float brdf_pdf() { return (float)(1.0 / (double)(2 * PT_PI_F)); }

// ---- light sampling (src/renderer.cu:56-284) -----------------------------------------------

inline Ray safe_ray(V3 origin, V3 dir, V3 normal) { return Ray{origin + (float)1e-6 * normal, dir}; } // :58-65

V3 sample_triangle_uniform(Rng &s, V3 a, V3 b, V3 c) { // :76-84
    float u1, u2;
    do {
        u1 = rng_uniform(s);
        u2 = rng_uniform(s);
    } while (u1 + u2 > 1);
    return (a + u1 * (b - a)) + u2 * (c - a);
}

struct AreaSample {
    float p_als;
    V3 normal, dir;
    float distance, area;
};

void sample_area_light(AreaSample &out, Ctx &c, V3 shadow_pos, V3 shadow_normal, uint32_t face_index,
                       bool direction_given) { // :91-134
    const pt_face &face = c.sc->faces[face_index];
    V3 A = from(c.sc->vertices[face.vertices[0]].position);
    V3 B = from(c.sc->vertices[face.vertices[1]].position);
    V3 C = from(c.sc->vertices[face.vertices[2]].position);
    V3 tri_cross = cross(C - B, A - B);
    float len = mag(tri_cross);
    out.normal = tri_cross / len;
    out.area = (float)(0.5 * (double)len);
    if (!direction_given) {
        V3 light_pos = sample_triangle_uniform(c.rng, A, B, C);
        out.dir = unit(light_pos - shadow_pos);
    }
    Ray shadow = safe_ray(shadow_pos, out.dir, shadow_normal);
    Hit light_hit;
    intersect(c, shadow, light_hit);
    bool visible = light_hit.has_hit && light_hit.distance > 0 && light_hit.faceIndex == (int)face_index;
    if (!visible) {
        out.p_als = 0;
        return;
    }
    out.distance = light_hit.distance;
    float cos_theta_y = fabsf(dot(out.normal, out.dir));
    out.p_als = out.distance * out.distance / (out.area * cos_theta_y);
}

V3 rgb_to_radiometric(V3 color, float intensity) { // :146-162
    float m = mag(color);
    V3 cn = mk(1, 1, 1);
    if ((double)m > 1e-12) cn = color / m;
    float photometric = intensity * m;
    float alpha = (float)((double)photometric / (683.0 * (double)luminance(cn)));
    return cn * alpha;
}

struct LightSample {
    V3 radiance, dir;
    float p_lss;
};

void sample_light_source(LightSample &out, Ctx &c, const Hit &hit) { // :164-249
    uint32_t count = c.in->lst_count;
    if (count == 0) {
        out.p_lss = 0;
        return;
    }
    int node_index = (int)((float)count * rng_uniform(c.rng));
    const pt_lst_node &node = c.in->lst[(uint32_t)node_index % count];
    out.p_lss = 1.0f / (float)count;
    if (node.type == PT_LST_SOURCE_LIGHT) {
        const pt_light &light = c.sc->lights[node.index];
        if (light.type == PT_LIGHT_POINT) {
            V3 d = from(light.position) - hit.position;
            float distance = mag(d);
            out.dir = d / distance;
            Ray shadow = safe_ray(hit.position, out.dir, hit.incident_normal);
            Hit light_hit;
            intersect(c, shadow, light_hit);
            if (!(distance < light_hit.distance)) {
                out.p_lss = 0;
                return;
            }
            V3 radiant_intensity = rgb_to_radiometric(from(light.color), light.intensity);
            out.radiance = radiant_intensity / (distance * distance);
            out.p_lss *= 1;
            return;
        }
        out.dir = unit(-from(light.direction));
        Ray shadow = safe_ray(hit.position, out.dir, hit.incident_normal);
        Hit shadow_hit;
        intersect(c, shadow, shadow_hit);
        if (shadow_hit.has_hit) {
            out.p_lss = 0;
            return;
        }
        out.radiance = rgb_to_radiometric(from(light.color), light.intensity);
        out.p_lss *= 1;
        return;
    }
    AreaSample als;
    sample_area_light(als, c, hit.position, hit.incident_normal, node.index, false);
    out.dir = als.dir;
    const pt_material &mat = c.sc->materials[c.sc->faces[node.index].material];
    out.radiance = rgb_to_radiometric(from(mat.emissive), 1);
    out.p_lss *= als.p_als;
}

float evaluate_direct_p(Ctx &c, const Ray &ray, const Hit &hit) { // :251-284
    uint32_t count = c.in->lst_count;
    if (count == 0) return 0;
    float p_total = 0;
    for (uint32_t i = 0; i < count; i++) {
        const pt_lst_node &node = c.in->lst[i];
        float p_node;
        if (node.type == PT_LST_SOURCE_LIGHT) {
            p_node = 0;
        } else {
            AreaSample la;
            la.dir = ray.r;
            sample_area_light(la, c, hit.position, hit.incident_normal, node.index, true);
            p_node = la.p_als;
        }
        p_total += p_node;
    }
    return p_total / (float)count; // plain float division (renderer.cu:282)
}

// mathops.h:712-720
inline void project_equirect(V3 d, float *u, float *v) {
    float phi = pt_atan2f(d.z, d.x);
    float theta = pt_acosf(fminf(fmaxf(d.y, -1.0f), 1.0f));
    *u = (float)(((double)phi + PT_PI_D) / (2.0f * PT_PI_D));
    *v = (float)((double)theta / PT_PI_D);
}

// renderer.cu:288-359
V3 integrate_Li(Ctx &c, Ray ray) {
    V3 light = mk(0, 0, 0);
    V3 throughput = mk(1, 1, 1);
    int bounce = -1;
    // diagnostics (orc_find_nonfinite): the first path vertex at which a term stops being finite
    auto note = [&](int where, const Hit &hit, V3 wo_ts, V3 w_ts, V3 f_cos, float p) {
        orc_nonfinite_diag *d = c.diag;
        if (!d || d->found) return;
        d->found = 1;
        d->where = where;
        d->sample = c.sample_index;
        d->bounce = bounce;
        d->face = hit.faceIndex;
        d->material = (int)(hit.mat - c.sc->materials);
        d->roughness = hit.brdf.roughness;
        d->metallic = hit.brdf.metallic;
        V3 h = unit(w_ts + wo_ts);
        d->h_z = h.z;
        d->pdf = p;
        const V3 cols[3] = {mk(hit.tangentBasis.m[0][0], hit.tangentBasis.m[1][0], hit.tangentBasis.m[2][0]),
                            mk(hit.tangentBasis.m[0][1], hit.tangentBasis.m[1][1], hit.tangentBasis.m[2][1]),
                            mk(hit.tangentBasis.m[0][2], hit.tangentBasis.m[1][2], hit.tangentBasis.m[2][2])};
        const V3 src[6] = {wo_ts, w_ts, f_cos, cols[0], cols[1], cols[2]};
        float *dst[6] = {d->wo_ts, d->w_ts, d->f_cos, d->tangent, d->bitangent, d->normal};
        for (int k = 0; k < 6; k++) { dst[k][0] = src[k].x; dst[k][1] = src[k].y; dst[k][2] = src[k].z; }
        d->ray_dir[0] = ray.r.x; d->ray_dir[1] = ray.r.y; d->ray_dir[2] = ray.r.z;
    };
    for (;;) {
        bounce++;
        Hit hit;
        intersect(c, ray, hit);
        if (!hit.has_hit) {
            V3 clear = from(c.in->cfg.world_clear_color);
            if (c.sc->clearTexture >= 0) {
                float u, v, px[4];
                project_equirect(ray.r, &u, &v);
                sample_texture(&c.sc->textures[c.sc->clearTexture], u, v, px);
                clear = clear * mk(px[0], px[1], px[2]);
            }
            light = light + throughput * clear;
            break;
        }
        if (rng_uniform(c.rng) >= hit.alpha) {
            ray = safe_ray(hit.position, ray.r, -hit.incident_normal);
            continue;
        }
        c.n.shaded_hits++;
        light = light + from(hit.mat->emissive) * throughput;
        float rr_prob = fminf(luminance(throughput), (float)0.99);
        if (rng_uniform(c.rng) >= rr_prob) break;

        M3 tbn_t = transposed(hit.tangentBasis);
        V3 wo_ts = tbn_t * -ray.r;

        LightSample lss = {};
        sample_light_source(lss, c, hit);
        if (lss.p_lss > 0) {
            V3 l_ts = tbn_t * lss.dir;
            float p_brdf = brdf_pdf();
            V3 f_cos = brdf_eval(hit.brdf, wo_ts, l_ts);
            V3 direct = f_cos * lss.radiance;
            float weight = lss.p_lss / (lss.p_lss + p_brdf);
            light = light + ((weight / lss.p_lss) * throughput) * direct;
            if (c.diag && !finite3(light)) note(1, hit, wo_ts, l_ts, f_cos, lss.p_lss);
        }

        V3 wi = sphere_sample_uniform(c.rng);
        wi.z = fabsf(wi.z);
        V3 f_cos = brdf_eval(hit.brdf, wo_ts, wi);
        float pdf = brdf_pdf();
        c.n.bounces++;

        V3 wi_global = hit.tangentBasis * wi;
        ray = safe_ray(hit.position, wi_global, hit.incident_normal);
        float p_direct = evaluate_direct_p(c, ray, hit);
        float weight = pdf / (p_direct + pdf);
        throughput = throughput * (f_cos * (weight / (pdf * rr_prob)));
        if (c.diag && !finite3(throughput)) note(2, hit, wo_ts, wi, f_cos, p_direct);
    }
    return light;
}

// renderer.cu:27-53
Ray camera_ray(const pt_camera &cam, float u, float v) {
    V3 P = from(cam.position), T = from(cam.target), Up = from(cam.updir);
    V3 W = T - P;
    V3 U = cross(W, Up);
    V3 V = cross(U, W);
    float focal = 0.1f;
    float yheight = atanf((float)(0.5 * (double)cam.yfov)) * focal;
    U = unit(U) * yheight;
    V = unit(V) * yheight;
    W = unit(W) * focal;
    M3 A = from_columns(U, V, W);
    return Ray{P, unit(A * mk(u, v, 1))};
}

// renderer.cu:361-405, one pixel
void render_pixel(const orc_inputs *in, int seed, int prev, int cur, int px, int py, float *img, orc_counters &acc) {
    int W = in->cfg.resolution_x, H = in->cfg.resolution_y;
    Ctx c;
    c.in = in;
    c.sc = in->scene;
    rng_init(c.rng, (uint64_t)(int64_t)seed, (uint64_t)(int64_t)(py * W + px));
    V3 total = mk(0, 0, 0);
    for (int i = 0; i < cur; i++) {
        float sx = (float)px + 0.33f * rng_normal(c.rng);
        float sy = (float)py + 0.33f * rng_normal(c.rng);
        float u = (2 * sx - (float)W) / (float)H;
        float v = (2 * sy - (float)H) / (float)H;
        c.n.samples++;
        total = total + integrate_Li(c, camera_ray(in->scene->camera, u, v));
    }
    total = total * powf(2.0f, in->cfg.output_exposure);
    total = total / (float)cur;
    int total_samples = prev + cur;
    float *p = img + 3 * ((size_t)py * (size_t)W + (size_t)px);
    V3 last = mk(p[0], p[1], p[2]);
    V3 next = last * ((float)prev / (float)total_samples) + total * ((float)cur / (float)total_samples);
    p[0] = next.x;
    p[1] = next.y;
    p[2] = next.z;
    acc.samples += c.n.samples;
    acc.rays += c.n.rays;
    acc.inner_visits += c.n.inner_visits;
    acc.tri_tests += c.n.tri_tests;
    acc.hit_records += c.n.hit_records;
    acc.shaded_hits += c.n.shaded_hits;
    acc.bounces += c.n.bounces;
}

} // namespace

extern "C" int orc_render_rows(const orc_inputs *in, int seed, int prev, int cur, float *img, int y0, int y1,
                               int threads, orc_counters *counters) {
    if (!in || !in->scene || !img || cur <= 0) return 1;
    int W = in->cfg.resolution_x, H = in->cfg.resolution_y;
    if (y0 < 0) y0 = 0;
    if (y1 > H) y1 = H;
    std::call_once(g_jump_once, build_jump);
    unsigned n = threads > 0 ? (unsigned)threads : std::thread::hardware_concurrency();
    if (n == 0) n = 1;
    std::atomic<int> next_row{y0};
    std::vector<orc_counters> per_thread(n);
    auto work = [&](unsigned id) {
        orc_counters acc{};
        for (;;) {
            int y = next_row.fetch_add(1);
            if (y >= y1) break;
            for (int x = 0; x < W; x++) render_pixel(in, seed, prev, cur, x, y, img, acc);
        }
        per_thread[id] = acc;
    };
    if (n == 1) {
        work(0);
    } else {
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < n; t++) pool.emplace_back(work, t);
        for (auto &t : pool) t.join();
    }
    if (counters) {
        memset(counters, 0, sizeof(*counters));
        for (const orc_counters &a : per_thread) {
            counters->samples += a.samples;
            counters->rays += a.rays;
            counters->inner_visits += a.inner_visits;
            counters->tri_tests += a.tri_tests;
            counters->hit_records += a.hit_records;
            counters->shaded_hits += a.shaded_hits;
            counters->bounces += a.bounces;
        }
    }
    return 0;
}

extern "C" int orc_render_batch(const orc_inputs *in, int seed, int prev, int cur, float *img, int threads,
                                orc_counters *counters) {
    if (!in) return 1;
    return orc_render_rows(in, seed, prev, cur, img, 0, in->cfg.resolution_y, threads, counters);
}

extern "C" int orc_find_nonfinite(const orc_inputs *in, int seed, int cur, int px, int py, orc_nonfinite_diag *out) {
    if (!in || !in->scene || !out) return 0;
    int W = in->cfg.resolution_x, H = in->cfg.resolution_y;
    memset(out, 0, sizeof(*out));
    Ctx c;
    c.in = in;
    c.sc = in->scene;
    c.diag = out;
    rng_init(c.rng, (uint64_t)(int64_t)seed, (uint64_t)(int64_t)(py * W + px));
    for (int i = 0; i < cur && !out->found; i++) { // render_pixel's loop
        c.sample_index = i;
        float sx = (float)px + 0.33f * rng_normal(c.rng);
        float sy = (float)py + 0.33f * rng_normal(c.rng);
        float u = (2 * sx - (float)W) / (float)H;
        float v = (2 * sy - (float)H) / (float)H;
        (void)integrate_Li(c, camera_ray(in->scene->camera, u, v));
    }
    return out->found;
}

extern "C" int orc_margin_probe(const orc_inputs *in, int seed, int cur, int y0, int y1, int threads, orc_margin_stats *out) {
    if (!in || !in->scene || !out || cur <= 0) return 1;
    int W = in->cfg.resolution_x, H = in->cfg.resolution_y;
    if (y0 < 0) y0 = 0;
    if (y1 > H) y1 = H;
    std::call_once(g_jump_once, build_jump);
    unsigned n = threads > 0 ? (unsigned)threads : std::thread::hardware_concurrency();
    if (n == 0) n = 1;
    std::atomic<int> next_row{y0};
    std::vector<orc_margin_stats> per_thread(n);
    auto work = [&](unsigned id) {
        orc_margin_stats m;
        memset(&m, 0, sizeof(m));
        for (;;) {
            int y = next_row.fetch_add(1);
            if (y >= y1) break;
            for (int x = 0; x < W; x++) {
                Ctx c;
                c.in = in;
                c.sc = in->scene;
                c.probe = &m;
                rng_init(c.rng, (uint64_t)(int64_t)seed, (uint64_t)(int64_t)(y * W + x));
                for (int i = 0; i < cur; i++) { // render_pixel's loop
                    float sx = (float)x + 0.33f * rng_normal(c.rng);
                    float sy = (float)y + 0.33f * rng_normal(c.rng);
                    float u = (2 * sx - (float)W) / (float)H;
                    float v = (2 * sy - (float)H) / (float)H;
                    (void)integrate_Li(c, camera_ray(in->scene->camera, u, v));
                }
            }
        }
        per_thread[id] = m;
    };
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < n; t++) pool.emplace_back(work, t);
    work(0);
    for (auto &t : pool) t.join();
    memset(out, 0, sizeof(*out));
    for (const orc_margin_stats &m : per_thread) {
        out->rays += m.rays; out->leaf_looks += m.leaf_looks; out->hits += m.hits;
        out->entry_behind += m.entry_behind; out->exit_in_front += m.exit_in_front;
        out->culled_by_distance += m.culled_by_distance; out->missed_box += m.missed_box;
        out->rays_with_two_candidates += m.rays_with_two_candidates;
        out->max_entry_behind = std::max(out->max_entry_behind, m.max_entry_behind);
        out->max_exit_in_front = std::max(out->max_exit_in_front, m.max_exit_in_front);
        out->max_culled_gap = std::max(out->max_culled_gap, m.max_culled_gap);
        out->max_rim_of_missed = std::max(out->max_rim_of_missed, m.max_rim_of_missed);
        out->missed_box_inf_rinv += m.missed_box_inf_rinv;
        out->max_rim_of_missed_finite = std::max(out->max_rim_of_missed_finite, m.max_rim_of_missed_finite);
        out->max_t_of_missed_finite = std::max(out->max_t_of_missed_finite, m.max_t_of_missed_finite);
        for (int k = 0; k < 16; k++) out->gap_below[k] += m.gap_below[k];
    }
    return 0;
}

extern "C" void orc_rng_set_scramble(const uint32_t constants[4]) {
    static const uint32_t curand[4] = {0xaad26b49u, 0xf7dcefddu, 1099087573u, 2591861531u};
    memcpy(g_scramble, constants ? constants : curand, sizeof(g_scramble));
}

extern "C" void orc_rng_state(uint64_t seed, uint64_t subsequence, uint32_t state[6]) {
    Rng s;
    rng_init(s, seed, subsequence);
    memcpy(state, s.v, sizeof(s.v));
    state[5] = s.d;
}

extern "C" void orc_rng_stream(uint64_t seed, uint64_t subsequence, uint32_t *out, int n) {
    Rng s;
    rng_init(s, seed, subsequence);
    for (int i = 0; i < n; i++) out[i] = rng_next(s);
}

extern "C" void orc_rng_sequence_jump_matrix(int k, uint32_t out[800]) {
    std::call_once(g_jump_once, build_jump);
    // rocRAND's tables step by powers of 4: table k = A^(2^67 * 4^k) = mat[2k]
    memcpy(out, g_jump->mat[2 * k], 800 * sizeof(uint32_t));
}

extern "C" int orc_primary_hits(const orc_inputs *in, int seed, int jitter, int32_t *face_ids, float *distances) {
    if (!in || !in->scene || !face_ids) return 1;
    int W = in->cfg.resolution_x, H = in->cfg.resolution_y;
    for (int py = 0; py < H; py++)
        for (int px = 0; px < W; px++) {
            Ctx c;
            c.in = in;
            c.sc = in->scene;
            rng_init(c.rng, (uint64_t)(int64_t)seed, (uint64_t)(int64_t)(py * W + px));
            float nx = rng_normal(c.rng), ny = rng_normal(c.rng);
            if (!jitter) nx = ny = 0;
            float sx = (float)px + 0.33f * nx, sy = (float)py + 0.33f * ny;
            float u = (2 * sx - (float)W) / (float)H;
            float v = (2 * sy - (float)H) / (float)H;
            Hit hit;
            intersect(c, camera_ray(in->scene->camera, u, v), hit);
            size_t at = (size_t)py * (size_t)W + (size_t)px;
            face_ids[at] = hit.has_hit ? hit.faceIndex : -1;
            if (distances) distances[at] = hit.distance;
        }
    return 0;
}

extern "C" void orc_sample_texture(const pt_texture *tex, float u, float v, float rgba[4]) {
    sample_texture(tex, u, v, rgba);
}

extern "C" float orc_logf(float x) { return pt_logf(x); }
extern "C" void orc_sincosf(float x, float *s, float *c) { pt_sincosf(x, s, c); }
extern "C" float orc_atan2f(float y, float x) { return pt_atan2f(y, x); }
extern "C" float orc_acosf(float x) { return pt_acosf(x); }

// ---- write_image's 8-bit buffer (src/image.cu:11-99), restated for the PNG / CLI tests -------------------------
// Same hidden doubles: `c <= 0.0031308`, `12.92 * c`, `1.055 * powf(c, 1 / 2.4) - 0.055`, `c < 0.0`, `c >= 1.0`.
// A pure-black pixel makes l_out / l_in = 0 / 0 = NaN in the reference; NaN fails both clamps and (uint8_t)(255 * NaN)
// is undefined, 0 on x86-64 (SURVEY.md section 8c "tonemap trap"): stated here as 0. Returns 1 on a non-finite input
// (check_sanity exits there, image.cu:57-70).
namespace {
inline float tm_luminance(const float *v) { return v[0] * 0.2126f + (v[1] * 0.7152f + v[2] * 0.0722f); } // :20-22, Vec3::dot
inline float tm_gamma(float c) {                                                                              // :40-47
    if ((double)c <= 0.0031308) return (float)(12.92 * (double)c);
    return (float)(1.055 * (double)powf(c, (float)(1 / 2.4)) - 0.055);
}
inline uint8_t tm_clamp_256(float c) {                                                                        // :11-17
    if (c != c) return 0;
    if ((double)c < 0.0) c = 0.0f;
    if ((double)c >= 1.0) c = 1.0f;
    return (uint8_t)(255 * c);
}
} // namespace

extern "C" int orc_tonemap_rgb8(const float *linear, int width, int height, uint8_t *out) {
    const float one[3] = {1, 1, 1};
    float white = tm_luminance(one);
    for (size_t i = 0; i < (size_t)width * (size_t)height; i++) {
        const float *c = linear + 3 * i;
        if (!std::isfinite(c[0]) || !std::isfinite(c[1]) || !std::isfinite(c[2])) return 1;
        white = fmaxf(white, tm_luminance(c));
    }
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++) {
            const float *c = linear + 3 * ((size_t)y * (size_t)width + (size_t)x);
            float l_old = tm_luminance(c);                                            // reinhard_extended_luminance :29-34
            float numerator = l_old * (1.0f + (l_old / (white * white)));
            float l_new = numerator / (1.0f + l_old);
            float scale = l_new / tm_luminance(c);                                    // change_luminance :24-27
            size_t i = ((size_t)(height - 1 - y) * (size_t)width + (size_t)x) * 3;    // vertical flip :94
            for (int k = 0; k < 3; k++) out[i + (size_t)k] = tm_clamp_256(tm_gamma(c[k] * scale));
        }
    return 0;
}
