/* pt_types.h — plain-old-data records that cross the host→device boundary.
 *
 * Every record keeps the byte layout of the reference struct it replaces, so a
 * maintainer of the reference can hand its own host arrays to ptcore.h without
 * repacking (reference sizes measured in SURVEY.md §8):
 *
 *   pt_vertex    52 B   src/scene.h:57-64    vertex_t
 *   pt_face      52 B   src/scene.h:77-84    face_t
 *   pt_material 100 B   src/scene.h:86-94    material_t
 *   pt_light     44 B   src/scene.h:103-107  light_t
 *   pt_camera    40 B   src/scene.h:109-112  camera_t
 *   pt_bvh_node  40 B   src/bvh.h:7-11       bvh_node_t
 *   pt_lst_node   8 B   src/lst.h:13-16      lst_node_t
 *
 * pt_texture replaces texture_t (src/scene.h:119-124), which holds two CUDA
 * texture objects; there is no texture-object type at this boundary, so the
 * host passes the decoded texels plus the sampler state that
 * createCudaTextureDescFromGLTF (src/scene.cu:628-682) would have set.
 */
#ifndef PT_TYPES_H_
#define PT_TYPES_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float x, y, z; } pt_vec3;
typedef struct { float x, y, z, w; } pt_vec4;

typedef struct {
    pt_vec3 position, normal, texcoord0;
    pt_vec4 tangent;
} pt_vertex;

enum { PT_FLAT_SHADING = 0, PT_BARY_SHADING = 1 };          /* src/scene.h:66-69 */
enum { PT_ALPHA_OPAQUE = 0, PT_ALPHA_MASK = 1, PT_ALPHA_BLEND = 2 }; /* :71-75 */
enum { PT_LIGHT_DIRECTIONAL = 0, PT_LIGHT_POINT = 1 };      /* src/scene.h:96-100 */
enum { PT_LST_SOURCE_FACE = 0, PT_LST_SOURCE_LIGHT = 1 };   /* src/lst.h:8-11 */

typedef struct {
    uint32_t vertices[3];
    uint32_t vertexCount;
    uint32_t material;
    pt_vec3 normal;
    pt_vec4 tangent;
    uint32_t shading;
} pt_face;

#define PT_MATERIAL_NAME_SIZE 32

typedef struct {
    char name[PT_MATERIAL_NAME_SIZE];
    pt_vec4 baseColorFactor;
    int32_t baseColorTexture, normalTexture, metallicRoughnessTexture;
    pt_vec3 emissive;
    float specular, metallicFactor, roughnessFactor, ior, transmission;
    uint32_t alphaMode;
    float alphaCutoff;
} pt_material;

typedef struct {
    uint32_t type;
    pt_vec3 color, position, direction;
    float intensity;
} pt_light;

typedef struct {
    pt_vec3 position, target, updir;
    float yfov;
} pt_camera;

typedef struct {
    pt_vec3 bmin, bmax;
    uint32_t leftChild, rightChild;
    uint32_t start, end;
} pt_bvh_node;

typedef struct {
    uint32_t type;
    uint32_t index;
} pt_lst_node;

/* Texture address / filter modes (the three CUDA modes the reference maps
 * glTF samplers to, src/scene.cu:632-662). */
enum { PT_WRAP_REPEAT = 0, PT_WRAP_CLAMP = 1, PT_WRAP_MIRROR = 2 };
enum { PT_FILTER_POINT = 0, PT_FILTER_LINEAR = 1 };

typedef struct {
    int32_t width, height;
    int32_t channels;   /* always 4 after load (src/scene.cu:736-744) */
    int32_t isFloat;    /* 1: float RGBA (.hdr), 0: 8-bit RGBA (.png/.jpg) */
    int32_t wrapS, wrapT;
    int32_t filter;
    int32_t _pad;
    const void *pixels; /* host pointer, row 0 first, no vertical flip */
} pt_texture;

/* Scene as handed to the device side: the five arrays Scene::device_from_host
 * uploads (src/scene.cu:974-990) plus camera and clear-texture index. */
typedef struct {
    const pt_vertex *vertices;   uint32_t vertex_count;
    const pt_face *faces;        uint32_t face_count;
    const pt_light *lights;      uint32_t light_count;
    const pt_material *materials; uint32_t material_count;
    const pt_texture *textures;  uint32_t texture_count;
    pt_camera camera;
    int32_t clearTexture;        /* -1: none (src/scene.h:161) */
} pt_scene_desc;

/* The subset of config_t (src/config.h:9-23) the kernel reads. */
typedef struct {
    int32_t resolution_x, resolution_y;
    float output_exposure;
    pt_vec3 world_clear_color;
} pt_render_config;

#ifdef __cplusplus
}
#endif

#endif /* PT_TYPES_H_ */
