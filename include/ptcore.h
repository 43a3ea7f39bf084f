/* ptcore.h — C ABI of the MI355X radiance kernel (libptcore.so).
 *
 * The reference has no plugin/FFI layer; the boundary of its hot path is the set
 * of CUDA-runtime calls in src/main.cu around one kernel launch. Each entry
 * point below replaces one of those interaction points. Plain
 * pointers and sizes only; no torch, CUDA or HIP types in any signature.
 *
 * Errors: every call returns 0 on success, non-zero on failure with the text in
 * pt_last_error() (the reference prints "CUDA error: %s" and exits,
 * src/utils.cu:16-22; the CLI keeps that exit-on-error policy).
 * Threading: one host thread per context; one launch in flight per context.
 * Ownership: the library copies what it is handed; host arrays stay the
 * caller's. Device copies are released by pt_destroy (the reference leaks
 * them, src/main.cu:148-151).
 */
#ifndef PTCORE_H_
#define PTCORE_H_

#include <stdint.h>

#include "pt_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pt_context pt_context;

/* Counters of the instrumented kernel variant (pt_count_batch). "rays" counts
 * what the reference would have traced (one closest-hit walk per camera,
 * bounce, shadow and light-pdf ray); "traversals" what this kernel walked. */
typedef struct {
    uint64_t samples;
    uint64_t rays;
    uint64_t inner_visits;
    uint64_t tri_tests;
    uint64_t traversals;
    uint64_t shaded_hits;
    uint64_t bounces;
    /* Wave-level occupancy of six code regions, in order: state-machine loop, inner-node
     * step, leaf step, path-vertex handler (main-ray result), shadow-ray result, BSDF sampling
     * (inside the vertex handler). iters = times a wave executed the region, lanes = sum of
     * lanes active in it (64 * iters = full). */
    uint64_t region_iters[6];
    uint64_t region_lanes[6];
    /* Shader-clock ticks summed over waves: [0] inside the walk slices, [1] node phases,
     * [2] leaf phases, [3] path-vertex handler, [4] shadow-ray results, [5] the handler's two rejection loops
     * (point on the emitter, BSDF direction: a wave repeats them until its unluckiest lane accepts);
     * wave_ticks = sum of wave lifetimes (includes RNG init and the image write). */
    uint64_t region_ticks[6];
    uint64_t wave_ticks;
    /* Lanes that sat out the wave's inner-node steps, summed over steps: [0] standing on a leaf
     * with another one parked, [1] walk finished except for a parked leaf, [2] walk finished,
     * waiting for its result handler. (Lanes whose pixel is complete are 64*iters - lanes - these.) */
    uint64_t node_step_idle[3];
    /* Speculative walks whose winner had another candidate within 2^-12 of it and that were repeated as
     * a sequential walk of the binary tree, the reference's loop step for step (kernel.h, kCullSlack). */
    uint64_t redone_walks;
} pt_counters;

const char *pt_last_error(void);

/* Binds a context to HIP device `device` (replaces the implicit device 0 and
 * default stream of src/main.cu). Builds the XORWOW 2^67 jump tables. */
int pt_create(int device, pt_context **out);
/* HIP devices visible to this process (0 without a GPU or a driver; never fails). */
int pt_device_count(void);
void pt_destroy(pt_context *ctx);

/* d_scene.device_from_host(h_scene) — src/main.cu:74, src/scene.cu:974-990,
 * src/scene.h:30-48. Copies the five arrays, camera and clear-texture index and
 * re-lays them out for the kernel (per-face shading records, compact
 * materials, texel arrays). */
int pt_scene_upload(pt_context *ctx, const pt_scene_desc *scene);

/* d_bvh.device_from_host(h_bvh) — src/main.cu:75, src/bvh.cu:347-359. Takes the
 * reference-format node array (40 B nodes, only the first node_count are read)
 * and face-index list; bakes 64 B two-child nodes and 48 B pre-gathered
 * triangles. Call after pt_scene_upload. */
int pt_bvh_upload(pt_context *ctx, const pt_bvh_node *nodes, uint32_t node_count,
                  const uint32_t *indices, uint32_t index_count);

/* d_lst.device_from_host(h_lst) — src/main.cu:76, src/lst.cu:36-46. */
int pt_lst_upload(pt_context *ctx, const pt_lst_node *nodes, uint32_t count);

/* The fields of config_t the kernel reads (resolution, exposure, clear colour;
 * src/renderer.cu:298,366,383,393) and cudaMalloc(&d_img) — src/main.cu:78-83.
 * (Re)allocates the library-owned accumulator, zero-filled (the reference
 * leaves it uninitialised and relies on garbage * 0). */
int pt_configure(pt_context *ctx, const pt_render_config *cfg);

/* Launch order: the kernel renders one tile per wavefront (8x8 pixels; 8x16 with two pixels per
 * lane on images of 3 M pixels and more). Every launch records how long each tile kept its wave
 * slot, and the next launch of the context dispatches the tiles heaviest first (the reference's
 * fixed 16x16 block order, src/main.cu:85-88, leaves the device draining behind its slowest
 * blocks). The first render launch after pt_bvh_upload first runs two short launches that write
 * no pixel (they time the binary and the four-wide form of the tree, see pt_walk_tree, and record
 * tile costs; at half the image size when the image is large, the same view through a quarter of the
 * pixels) and blocks the host for about 5 ms while it reads the two timings.
 * Pixels depend on none of this. PT_LPT=0 in the environment keeps the static order.
 *
 * render_kernel<<<grid, block>>>(d_img, d_bvh, d_scene, d_lst, cfg, prev, cur)
 * + cudaDeviceSynchronize — src/main.cu:104-106, src/renderer.cu:361-405.
 * `seed` is cfg.seed of that launch (the host loop adds 1 per batch,
 * src/main.cu:121). Blocking. Accumulates into the library-owned image. */
int pt_render_batch(pt_context *ctx, int seed, int previous_samples, int current_samples);

/* Same launch into a caller-owned DEVICE buffer of resolution_x*resolution_y*3
 * floats, on a caller stream (hipStream_t passed as void*, NULL = the
 * context's own stream), without synchronising (except in the first launch after
 * pt_bvh_upload, see "Launch order" above). For callers that keep the
 * accumulator in their own device memory (multi-GPU reduce). Launches of one
 * context must stay on one stream: the tile order of a launch is written by the
 * launch before it. */
int pt_render_batch_into(pt_context *ctx, float *device_img, void *stream, int seed,
                         int previous_samples, int current_samples);

/* NO COUNTERPART IN THE REFERENCE. For callers that cannot allocate device memory: tight serving loops, harnesses
 * that forbid hipMalloc, code that runs between a graph's captures. (Not for use UNDER stream capture: the launch
 * records timing events and decides the tile order on the host at call time, neither of which a replay repeats.) The same launch as pt_render_batch_into,
 * with two differences: (1) the image size, exposure and clear colour come with the call (pt_configure, whose
 * job is to allocate, need not have been called, and is not disturbed if it was); (2) the kernels' per-tile
 * scratch (RNG base states, tile costs, sort keys, launch order: 44 bytes per 8x8 tile) lives in ONE
 * file-scope static device array of PT_STATIC_MAX_TILES tiles (2.75 MiB) that every GPU the library runs on
 * holds until the process ends. Calls no hipMalloc / hipFree or other allocating routine and never blocks
 * the host (no tree timing: a tree not chosen yet is walked four-wide; pixels do not depend on it).
 * Limit: ceil(x/8) * ceil(y/8) <= PT_STATIC_MAX_TILES (e.g. 2048 x 2048); larger sizes are refused with an
 * error. All calls on one GPU, from any context, share that GPU's array: calls that overlap on one GPU
 * (two streams, two contexts, two threads) race. device_img: caller-owned device buffer of
 * resolution_x * resolution_y * 3 floats; stream as in pt_render_batch_into. */
#define PT_STATIC_MAX_TILES 65536
int pt_render_batch_static(pt_context *ctx, float *device_img, void *stream, const pt_render_config *cfg, int seed,
                           int previous_samples, int current_samples);

/* cudaMemcpy(h_img, d_img, ..., DeviceToHost) — src/main.cu:111. Caller-owned
 * host buffer of resolution_x*resolution_y*3 floats, row 0 = bottom row as in
 * the reference (the flip happens at PNG write, src/image.cu:94). */
int pt_read_image(pt_context *ctx, float *host_rgb);

/* Page-locked host memory for pt_read_image's destination (the reference reads into pageable memory,
 * src/main.cu:86; from a pinned buffer the 12.6 MB of a 1024 x 1024 image arrive at the DMA engine's rate, 0.3 ms
 * instead of 0.8). Any host pointer works with pt_read_image; these two only make the copy faster.
 * pt_host_alloc returns NULL (and sets pt_last_error) when no device or no memory is available. */
void *pt_host_alloc(size_t bytes);
void pt_host_free(void *ptr);

int pt_clear_image(pt_context *ctx);
int pt_synchronize(pt_context *ctx);

/* ---- multi-GPU in one process (no counterpart in the reference, which is single-GPU: SURVEY.md §8e) ----
 * One context per listed device plus one RCCL communicator over them (librccl.so is opened on first use).
 * Upload the scene, BVH and LST to, and configure, every pt_group_context(g, r) with the single-GPU calls
 * above (the scene is replicated). pt_group_render is the reference's batch loop (src/main.cu:101-133)
 * sharded by batch: batch b runs on rank b mod N with seed + b into that rank's accumulator (running mean of
 * its own batches); then every rank scales its mean by its share of the samples and ONE ncclReduce(sum) over
 * xGMI leaves the image on rank 0: pt_read_image(pt_group_context(g, 0), ...). Equal to the single-GPU run
 * with the same samples_every_update up to fp32 summation order; with one device, bit for bit. Blocking. */
typedef struct pt_group pt_group;
int pt_group_create(const int *devices, int count, pt_group **out);
void pt_group_destroy(pt_group *group);
int pt_group_size(pt_group *group);
pt_context *pt_group_context(pt_group *group, int rank);
int pt_group_render(pt_group *group, int seed, int samples, int samples_every_update);

/* The plan pt_group_render follows, as a pure function (no device, no group; callable on a machine without a
 * GPU): batch b of the reference's loop (src/main.cu:101-102) goes to rank b mod ranks with seed + b and
 * continues that rank's running mean. Writes at most `capacity` entries to `batches` (may be NULL) and one
 * weight per rank to `weights` (may be NULL): the rank's share of the samples, 0 for a rank without a batch; the
 * shares sum to 1. Returns the number of batches of the whole job, or -1 on bad arguments. */
typedef struct {
    int index;            /* b */
    int rank;             /* b mod ranks */
    int seed;             /* seed + b */
    int previous_samples; /* samples this rank has rendered before this batch */
    int current_samples;  /* min(samples_every_update, samples left) */
} pt_group_batch;
int pt_group_plan(int seed, int samples, int samples_every_update, int ranks, pt_group_batch *batches, int capacity, float *weights);

/* ---- verification / measurement hooks (no counterpart in the reference) ---- */

/* First n outputs of the device XORWOW stream for (seed, subsequence, offset 0):
 * what curand(&state) would return after curand_init (src/random.cu:3-5). */
int pt_rng_stream(pt_context *ctx, uint64_t seed, uint64_t subsequence, uint32_t *host_out, int n);

/* Face id hit by the first camera ray of every pixel (-1 = miss) and its
 * distance (1e30 on a miss); jitter = 0 zeroes the Box-Muller offsets. Walked by the
 * same walker as the render launches (sliced speculative walk + sequential
 * start-over), on the tree `tree` names: 2 = binary, 4 = four-wide, 0 = the one
 * render launches use (binary if not chosen yet). *redone_walks (may be NULL) =
 * rays that found two candidates within rounding of each other and were walked
 * again in the reference's order: the count of epsilon ties of this launch.
 * Host buffers of resolution_x*resolution_y entries; distances may be NULL. */
int pt_primary_hits(pt_context *ctx, int seed, int jitter, int tree, int32_t *host_face_ids, float *host_distances,
                    uint64_t *redone_walks);

/* Runs the instrumented variant of one batch (does not touch the image). */
int pt_count_batch(pt_context *ctx, int seed, int current_samples, pt_counters *out);

/* HIP-event duration (ms) of the most recent render launch of this context,
 * measured on the stream it was launched on; waits for that launch. */
float pt_last_kernel_ms(pt_context *ctx);

/* Same for the render launch `launches_back` launches before the most recent one (0 = most
 * recent, at most 63): lets a caller time a whole loop without waiting inside it. -1 if there is
 * no such launch. */
float pt_kernel_ms(pt_context *ctx, int launches_back);

/* Which form of the uploaded tree the render launches walk: 2 = the reference's binary tree, 4 = the
 * same tree folded four-wide at upload, 0 = not chosen yet. Both give the same hits; the first render
 * launch after pt_bvh_upload times a 2-sample launch on each and keeps the faster (PT_TREE=binary|wide
 * in the environment forces one). A walk of either that finds two candidates within rounding of each
 * other starts over as a sequential walk of the binary tree in the reference's order
 * (pt_counters.redone_walks), so the choice never shows in a pixel. Faces on the same three points would
 * make every ray through them do that: pt_bvh_upload looks for such faces and then answers 2 at once. */
int pt_walk_tree(pt_context *ctx);

/* NO COUNTERPART IN THE REFERENCE. Which arithmetic the context's render launches compute in.
 * PT_ARITHMETIC_EXACT (default, the gate): every float operation rounds as the CPU restatement of the reference's
 * kernel does (no contraction, IEEE division and square root); images are bit-identical to the oracle's.
 * PT_ARITHMETIC_FAST: the same kernels compiled a second time with tolerance arithmetic (fp contraction on, the
 * hardware's 1-ulp reciprocal / root, no tie bookkeeping in the walk): what nvcc's default contraction gives the
 * reference itself (bvh.cu:222-237, intersect.cu:12-81, mathops.h:114-121). Same RNG stream bit for bit; primary
 * hits and image agree with the exact variant within the tolerances stated in tests/test_gpu_tolerance.py.
 * PT_ARITHMETIC=fast in the environment makes it the default of new contexts. Waits for the context's own stream; a
 * caller that launches on its own stream (pt_render_batch_into) switches between launches it has synchronised. The
 * next render launch times the two forms of the tree again for the new variant (blocks the host ~5 ms). */
#define PT_ARITHMETIC_EXACT 0
#define PT_ARITHMETIC_FAST 1
int pt_set_arithmetic(pt_context *ctx, int arithmetic);
int pt_arithmetic(pt_context *ctx);

/* Bytes of device memory held by the context (scene + BVH + LST + image). */
uint64_t pt_device_footprint(pt_context *ctx);

#ifdef __cplusplus
}
#endif

#endif /* PTCORE_H_ */
