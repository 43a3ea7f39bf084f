/* oracle.h — C ABI of the CPU oracle (liboracle.so).
 *
 * TEST INFRASTRUCTURE ONLY. Nothing under pathtracer_amd/ may include, link or
 * load this; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg do, and there only as the checker / the timed CPU baseline.
 *
 * The oracle is a plain C++ restatement of the reference's render_kernel
 * (/root/reference src/renderer.cu:361-405 and everything it calls), written to
 * follow the reference's own structure: array-of-struct scene records, 40-byte
 * BVH nodes walked with a 64-entry stack, the full hit record rebuilt at every
 * closer hit, one closest-hit traversal per shadow ray and per light-pdf ray.
 *
 * PARITY UNPINNED for the first-party kernel logic: the reference ships no
 * tests, golden vectors or fixtures for this path, and its
 * sources cannot be built here (they need the CUDA runtime and cuRAND headers,
 * which this image lacks; no stand-ins were written). What IS pinned
 * (DESIGN.md §5 has the table):
 *   - the glTF loader, the .hdr/.png decoders and the tangents against the
 *     reference's OWN vendored tinygltf, stb_image and MikkTSpace, built from
 *     where they lie into oracle/_ref (tests/test_reference_pins.py,
 *     tests/test_oracle_and_host.py::test_tangents_match_mikktspace_*);
 *   - scene/BVH structure counts against the figures SURVEY.md §3.3 recorded
 *     from the reference's builder (test_scene_and_bvh_counts_match_survey,
 *     test_golden_scene_fixtures_rebuild_the_surveyed_trees);
 *   - the XORWOW generator end to end against rocRAND's own engine with its
 *     constants substituted (test_whole_xorwow_engine_equals_rocrand_with_its_constants).
 *     cuRAND's four seed-scramble constants are from the public curand_kernel.h
 *     as recalled in SURVEY.md §8c: unverified here.
 */
#ifndef ORACLE_H_
#define ORACLE_H_

#include <stdint.h>

#include "pt_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    const pt_scene_desc *scene;
    const pt_bvh_node *nodes;   uint32_t node_count;
    const uint32_t *indices;    uint32_t index_count;
    const pt_lst_node *lst;     uint32_t lst_count;
    pt_render_config cfg;
} orc_inputs;

typedef struct {
    uint64_t samples;        /* camera paths started */
    uint64_t rays;           /* calls of the closest-hit routine (all kinds) */
    uint64_t inner_visits;   /* inner nodes popped (two slab tests each) */
    uint64_t tri_tests;      /* Moeller-Trumbore evaluations */
    uint64_t hit_records;    /* full hit records written (every closer hit) */
    uint64_t shaded_hits;    /* path vertices that reached shading */
    uint64_t bounces;        /* BSDF samples taken */
} orc_counters;

/* One launch of render_kernel over the whole image (renderer.cu:361-405):
 * img is W*H*3 floats, read and written (running mean with previous_samples).
 * threads <= 0 picks hardware_concurrency. counters may be NULL. */
int orc_render_batch(const orc_inputs *in, int seed, int previous_samples, int current_samples,
                     float *img, int threads, orc_counters *counters);

/* Same, restricted to rows [y0, y1) — the bounded CPU-baseline sample. */
int orc_render_rows(const orc_inputs *in, int seed, int previous_samples, int current_samples,
                    float *img, int y0, int y1, int threads, orc_counters *counters);

/* First n outputs of curand(&state) after curand_init(seed, subsequence, 0). */
void orc_rng_stream(uint64_t seed, uint64_t subsequence, uint32_t *out, int n);

/* TEST HOOK: replaces the four seed-scramble constants {C0, C1, M0, M1} (s0 = lo32(seed) ^ C0, s1 = hi32 ^ C1,
 * t0 = M0 * s0, t1 = M1 * s1) for every later call, in every thread; NULL restores the cuRAND set. Used only
 * to compare the whole generator with rocRAND's engine, whose constants differ. Not thread-safe. */
void orc_rng_set_scramble(const uint32_t constants[4]);

/* XORWOW state (v[0..4], d) right after curand_init(seed, subsequence, 0). */
void orc_rng_state(uint64_t seed, uint64_t subsequence, uint32_t state[6]);

/* The 160x160 GF(2) matrix A^(2^67 * 4^k) in rocRAND's table layout
 * (800 words: row (i*32+j) holds the image of input bit j of word i). */
void orc_rng_sequence_jump_matrix(int k, uint32_t out[800]);

/* Face id of the first camera ray of every pixel (sample 0 of the batch seeded
 * with `seed`), -1 on a miss. jitter=0 replaces the Box-Muller offsets by 0. */
int orc_primary_hits(const orc_inputs *in, int seed, int jitter, int32_t *face_ids, float *distances);

/* Diagnostics: replays pixel (px, py) of a batch and reports the first path vertex at which the radiance
 * (where = 1: the direct term, renderer.cu:333-342) or the throughput (where = 2: the BSDF sample,
 * renderer.cu:352-355) stops being finite. w_ts is the incident direction of that term in tangent space,
 * h_z the z of the half vector brdf.cu:14 evaluates GGX at. Returns 1 if such a vertex was found. */
typedef struct {
    int32_t found, where, sample, bounce, face, material;
    float roughness, metallic, h_z, pdf;
    float wo_ts[3], w_ts[3], f_cos[3], tangent[3], bitangent[3], normal[3], ray_dir[3];
} orc_nonfinite_diag;
int orc_find_nonfinite(const orc_inputs *in, int seed, int current_samples, int px, int py, orc_nonfinite_diag *out);

/* Measurement behind the margins of the product's speculative walks (kernel.h: kCullSlack, kNearHi, kEdgeBand): how far
 * the box arithmetic (bvh.cu:222-237) and the triangle arithmetic (intersect.cu:12-81) disagree about ONE ray. The
 * render of orc_render_rows is repeated with a probe in the traversal: whenever the walk looks at a child that is a
 * leaf, whatever the box test said, every triangle of that leaf is also intersected, and for each one the ray hits in
 * front of the best distance so far (a hit the reference accepts if it gets to test it) the leaf box's raw entry and
 * exit distances are compared with the triangle's t. All gaps are relative to t. */
typedef struct {
    uint64_t rays, leaf_looks, hits;
    /* the box test passes, but the box's entry distance lies BEHIND the triangle's t (or its exit in front of it) */
    uint64_t entry_behind, exit_in_front;
    double max_entry_behind, max_exit_in_front;
    /* the box test culls the leaf against the best distance although the triangle is nearer than it (what kCullSlack is for) */
    uint64_t culled_by_distance;
    double max_culled_gap;
    /* the box test MISSES the leaf (exit before entry, or behind the origin) although the ray hits a triangle in it
     * (what kEdgeBand is for): how far inside the triangle such a hit can be (smallest barycentric weight) */
    uint64_t missed_box;
    double max_rim_of_missed;
    /* ... of which by rays whose direction has a zero component (infinite reciprocal: 0 * inf = NaN in the box test; the
     * product walks those on the reference's own tree with its select-form test), and the deepest hit of the OTHERS */
    uint64_t missed_box_inf_rinv;
    double max_rim_of_missed_finite;
    double max_t_of_missed_finite;
    /* per ray: relative gap between the accepted closest hit and the runner-up among the accepted candidates,
     * histogram over powers of two: gap_below[k] = rays with gap < 2^-(k+8), k = 0..15 (2^-8 .. 2^-23) */
    uint64_t rays_with_two_candidates;
    uint64_t gap_below[16];
} orc_margin_stats;
int orc_margin_probe(const orc_inputs *in, int seed, int current_samples, int y0, int y1, int threads, orc_margin_stats *out);

/* write_image's tonemapped 8-bit RGB buffer (src/image.cu:72-99: extended Reinhard on luminance with the white
 * point max(1, max luminance), sRGB gamma, (uint8_t)(255 * c), vertical flip), restated independently of
 * pathtracer_amd/csrc/host/image_io.cpp; out = width * height * 3 bytes. Returns 1 on a non-finite input. */
int orc_tonemap_rgb8(const float *linear, int width, int height, uint8_t *out);

/* Software texture fetch, exposed for unit tests. */
void orc_sample_texture(const pt_texture *tex, float u, float v, float rgba[4]);

/* The deterministic math of pt_detmath.h, exposed for accuracy tests. */
float orc_logf(float x);
void orc_sincosf(float x, float *s, float *c);
float orc_atan2f(float y, float x);
float orc_acosf(float x);

#ifdef __cplusplus
}
#endif

#endif /* ORACLE_H_ */
