/* pthost.h — C ABI of the host pipeline (libpthost.so): everything the
 * reference does on the CPU before and after the radiance kernel.
 *
 * Each entry point names the reference routine it restates. None of this is
 * GPU work; it exists because the kernel's inputs (scene arrays, BVH, light
 * table) and the PNG it feeds are defined by these routines.
 */
#ifndef PTHOST_H_
#define PTHOST_H_

#include <stddef.h>
#include <stdint.h>

#include "pt_types.h"

#ifdef __cplusplus
extern "C" {
#endif

#define PTH_PATH_MAX 256 /* src/config.h:7  typedef char path_t[256] */

/* config_t, src/config.h:9-23 — same fields, same order. */
typedef struct {
    int resolution_x, resolution_y;
    float output_exposure;
    int samples, seed, samples_every_update;
    pt_vec3 world_clear_color;
    char world_clear_color_texture[PTH_PATH_MAX];
    int log_level, log_stdout;
    char path_gltf[PTH_PATH_MAX], dir_output[PTH_PATH_MAX], path_render[PTH_PATH_MAX];
    pt_vec3 default_camera_position, default_camera_target, default_camera_updir;
    float default_camera_yfov;
} pth_config;

/* load_config, src/config.cu:55-86. Same flag names, same "scan argv[1..argc-2]"
 * rule. Unlike the reference, fields start zeroed (the reference leaves missing
 * ones uninitialised, src/config.cu:50-52). Returns the number of REQUIRED flags
 * that were missing (the reference always returns 0). */
int pth_load_config(pth_config *cfg, int argc, char **argv);

/* logger, src/logger.h:5-17: levels NONE0 ERROR1 WARNING2 INFO3 TRACE4. */
int pth_log_init(int level, const char *file, int to_stdout);
void pth_log_close(void);
void pth_log(int level, const char *fmt, ...);

/* Last error text of the calling thread ("" if none). */
const char *pth_last_error(void);

/* ---- scene ---------------------------------------------------------------- */

typedef struct pth_scene pth_scene; /* owns every array reachable from its desc */

/* Scene::read_gltf, src/scene.cu:839-972. cfg supplies the clear texture path
 * and the default camera (may be NULL: no env map, camera required in file).
 * Returns 0 on success. */
int pth_scene_load_gltf(const char *path, const pth_config *cfg, pth_scene **out);

/* Deep-copies a caller-built description (procedural scenes, tests). */
int pth_scene_from_desc(const pt_scene_desc *desc, pth_scene **out);

const pt_scene_desc *pth_scene_get_desc(const pth_scene *scene);
void pth_scene_free(pth_scene *scene); /* Scene::_free, src/scene.cu:992-1004 */

/* calculate_tangents (src/scene.cu:371-465) on a face range of a caller array:
 * writes vertices[*].tangent through the face indices exactly as the
 * reference's setTSpaceBasic callback does. Exposed for tests. */
int pth_calculate_tangents(pt_vertex *vertices, uint32_t vertex_count, const pt_face *faces,
                           uint32_t first_face, uint32_t num_faces);

/* Radiance .hdr decode (what stbi_loadf returns for the reference, 4 floats per
 * pixel, alpha 1). Caller frees *rgba with pth_free. */
int pth_load_hdr(const char *path, int *width, int *height, float **rgba);
/* 8-bit PNG decode to RGBA8 (what stbi_load(..., 4) returns). */
int pth_load_png(const char *path, int *width, int *height, uint8_t **rgba);
void pth_free(void *p);

/* ---- BVH / LST ------------------------------------------------------------- */

typedef struct {
    pt_bvh_node *nodes;    /* node_capacity = 2N-1 records, node_count used */
    uint32_t node_capacity;
    uint32_t node_count;
    uint32_t *indices;     /* N face indices in leaf order */
    uint32_t index_count;
    /* bvh_stats_t, src/bvh.h:20-25 */
    uint32_t total_skipped_faces, number_leaves;
    float average_leaf_size;
    int max_depth;
} pth_bvh;

/* BVH::build, src/bvh.cu:288-345 (binned SAH, 16 bins, leaf <= 2). */
int pth_bvh_build(const pt_scene_desc *scene, pth_bvh *out);
void pth_bvh_free(pth_bvh *bvh);

/* LST::build, src/lst.cu:6-34. Caller frees *nodes with pth_free. */
int pth_lst_build(const pt_scene_desc *scene, pt_lst_node **nodes, uint32_t *count);

/* ---- output ---------------------------------------------------------------- */

/* The tonemap half of write_image (src/image.cu:72-99): extended Reinhard on
 * luminance, sRGB gamma, 8-bit, vertical flip. rgb8 holds 3*w*h bytes — the
 * buffer the reference hands to stbi_write_png. Returns 1 if a pixel is not
 * finite (the reference exits there, src/image.cu:57-70). */
int pth_tonemap_rgb8(const float *linear_rgb, int width, int height, uint8_t *rgb8);

/* write_image, src/image.cu:72-111: tonemap + PNG file. */
int pth_write_image(const float *linear_rgb, int width, int height, const char *filename);

/* PNG encode of an RGB8 buffer (role of stbi_write_png). */
int pth_write_png_rgb8(const uint8_t *rgb8, int width, int height, const char *filename);

#ifdef __cplusplus
}
#endif

#endif /* PTHOST_H_ */
