// gltf_oracle.cpp — LOADER ORACLE. TEST INFRASTRUCTURE ONLY (like oracle.cpp; see oracle.h).
//
// A restatement of the reference's Scene::read_gltf and its helpers (src/scene.cu:24-626, 713-972) ON TOP OF THE
// REFERENCE'S OWN vendored third-party libraries, which are plain C/C++ and compile as they are from where they lie
// under /root/reference/src (oracle/Makefile builds this file into oracle/_ref/libgltfref.so with -I$(REF)/src):
//   tiny_gltf/tiny_gltf.h + json.hpp   glTF / GLB parsing, buffers, accessors, defaults, KHR_lights_punctual
//   stb/stb_image.h                    .png and .hdr decoding (stbi_load / stbi_loadf)
//   MikkTSpace/mikktspace.c            tangents (linked in: genTangSpaceDefault)
// So everything those libraries decide (accessor bytes, strides, base64 / GLB buffers, material defaults such as
// metallic = roughness = 1 and alphaCutoff 0.5, node TRS parsing, image texels, tangents) is the reference's, not
// ours. What is restated here, line by line with citations, is the first-party glue between them: the node
// transform arithmetic (src/mathops.h Mat4, its operation order), the vertex / face / material / light / camera
// records. pathtracer_amd/csrc/host/scene_load.cpp (own JSON reader, own codecs, own tangents) must produce the
// same bytes: tests/test_reference_pins.py compares every record of every bundled scene.
//
// Nothing of the reference is copied: its headers are included by path at build time; when /root/reference is
// absent (the GPU box) this file is not built and the tests that need it skip.
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#define STB_IMAGE_IMPLEMENTATION
#define STB_IMAGE_WRITE_IMPLEMENTATION
#include "stb/stb_image.h"
#include "stb/stb_image_write.h"

#define TINYGLTF_IMPLEMENTATION
#define TINYGLTF_NO_INCLUDE_STB_IMAGE
#define TINYGLTF_NO_INCLUDE_STB_IMAGE_WRITE
#include "tiny_gltf/tiny_gltf.h"

extern "C" {
#include "MikkTSpace/mikktspace.h"
}

#include "pt_types.h"

namespace tg = tinygltf;

namespace {

// ---- src/mathops.h: Mat4 (:522-586), Vec4 (:487-494), Vec3 (:114-141), Mat3 * Vec3 (:264-268) --------------------

struct M4 {
    float m[4][4];
};
M4 mat4(float m00, float m01, float m02, float m03, float m10, float m11, float m12, float m13, float m20, float m21, float m22,
        float m23, float m30, float m31, float m32, float m33) {
    return M4{{{m00, m01, m02, m03}, {m10, m11, m12, m13}, {m20, m21, m22, m23}, {m30, m31, m32, m33}}};
}
M4 identity() { return mat4(1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1); }
M4 mul(const M4 &a, const M4 &b) { // :564-577
    M4 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            r.m[i][j] = ((a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j]) + a.m[i][2] * b.m[2][j]) + a.m[i][3] * b.m[3][j];
    return r;
}
pt_vec4 mul(const M4 &a, pt_vec4 v) { // :552-561
    return pt_vec4{((a.m[0][0] * v.x + a.m[0][1] * v.y) + a.m[0][2] * v.z) + a.m[0][3] * v.w,
                   ((a.m[1][0] * v.x + a.m[1][1] * v.y) + a.m[1][2] * v.z) + a.m[1][3] * v.w,
                   ((a.m[2][0] * v.x + a.m[2][1] * v.y) + a.m[2][2] * v.z) + a.m[2][3] * v.w,
                   ((a.m[3][0] * v.x + a.m[3][1] * v.y) + a.m[3][2] * v.z) + a.m[3][3] * v.w};
}
pt_vec3 dehomogenise(pt_vec4 v) { // :491-494 -> Vec3 / float :114-117 (DIV(1.0, s) is a double division)
    float t = (float)(1.0 / (double)v.w);
    return pt_vec3{v.x * t, v.y * t, v.z * t};
}
pt_vec3 linear_mul(const M4 &a, pt_vec3 v) { // getLinearPart :579-585, Mat3 * Vec3 :264-268
    return pt_vec3{(a.m[0][0] * v.x + a.m[0][1] * v.y) + a.m[0][2] * v.z, (a.m[1][0] * v.x + a.m[1][1] * v.y) + a.m[1][2] * v.z,
                   (a.m[2][0] * v.x + a.m[2][1] * v.y) + a.m[2][2] * v.z};
}
pt_vec3 normalized(pt_vec3 v) { // :119-141: dot = x*x + (y*y + z*z), t = DIV(1.0, sqrtf(dot))
    float d = v.x * v.x + (v.y * v.y + v.z * v.z);
    float t = (float)(1.0 / (double)sqrtf(d));
    return pt_vec3{v.x * t, v.y * t, v.z * t};
}

// ---- scene.cu:34-108 ---------------------------------------------------------------------------------------------

M4 get_transform(const tg::Node &node) {
    if (node.matrix.size()) { // column major in the file
        float e[16];
        for (int i = 0; i < 16; i++) e[i] = (float)node.matrix[(size_t)i];
        return mat4(e[0], e[4], e[8], e[12], e[1], e[5], e[9], e[13], e[2], e[6], e[10], e[14], e[3], e[7], e[11], e[15]);
    }
    M4 forward = identity();
    if (node.scale.size()) {
        float sx = (float)node.scale[0], sy = (float)node.scale[1], sz = (float)node.scale[2];
        forward = mul(mat4(sx, 0, 0, 0, 0, sy, 0, 0, 0, 0, sz, 0, 0, 0, 0, 1), forward);
    }
    if (node.rotation.size()) {
        float qi = (float)node.rotation[0], qj = (float)node.rotation[1], qk = (float)node.rotation[2], qr = (float)node.rotation[3];
        float s = 1;
        float m00 = 1 - 2 * s * (qj * qj + qk * qk);
        float m01 = 2 * s * (qi * qj - qk * qr);
        float m02 = 2 * s * (qi * qk + qj * qr);
        float m10 = 2 * s * (qi * qj + qk * qr);
        float m11 = 1 - 2 * s * (qi * qi + qk * qk);
        float m12 = 2 * s * (qj * qk - qi * qr);
        float m20 = 2 * s * (qi * qk - qj * qr);
        float m21 = 2 * s * (qj * qk + qi * qr);
        float m22 = 1 - 2 * s * (qi * qi + qj * qj);
        forward = mul(mat4(m00, m01, m02, 0, m10, m11, m12, 0, m20, m21, m22, 0, 0, 0, 0, 1), forward);
    }
    if (node.translation.size()) {
        float tx = (float)node.translation[0], ty = (float)node.translation[1], tz = (float)node.translation[2];
        forward = mul(mat4(1, 0, 0, tx, 0, 1, 0, ty, 0, 0, 1, tz, 0, 0, 0, 1), forward);
    }
    return forward;
}

struct Texels {
    std::vector<uint8_t> bytes;
};

} // namespace

struct gref_scene {
    std::vector<pt_vertex> vertices;
    std::vector<pt_face> faces;
    std::vector<pt_material> materials;
    std::vector<pt_light> lights;
    std::vector<pt_camera> cameras;
    std::vector<pt_texture> textures;
    std::vector<Texels> texels;
    pt_scene_desc desc;
    std::string error;
};

namespace {

// scene.cu:110-132
void parse_camera(gref_scene &scene, const tg::Model &model, const tg::Node &node, const M4 &M) {
    const tg::Camera &src = model.cameras[(size_t)node.camera];
    if (src.type != "perspective") throw std::string("unsupported camera type: ") + src.type;
    pt_camera cam;
    cam.yfov = (float)src.perspective.yfov;
    cam.position = dehomogenise(mul(M, pt_vec4{0, 0, 0, 1}));
    cam.target = dehomogenise(mul(M, pt_vec4{0, 0, -1, 1}));
    pt_vec3 up = dehomogenise(mul(M, pt_vec4{0, 1, 0, 1}));
    cam.updir = pt_vec3{up.x - cam.position.x, up.y - cam.position.y, up.z - cam.position.z};
    scene.cameras.push_back(cam);
}

// scene.cu:134-173
void parse_light(gref_scene &scene, const tg::Model &model, const tg::Node &node, const M4 &M) {
    const tg::Light &src = model.lights[(size_t)node.light];
    pt_light light;
    light.color = pt_vec3{1, 1, 1};
    if (src.color.size()) light.color = pt_vec3{(float)src.color[0], (float)src.color[1], (float)src.color[2]};
    light.intensity = (float)src.intensity;
    light.position = dehomogenise(mul(M, pt_vec4{0, 0, 0, 1}));
    light.direction = normalized(linear_mul(M, pt_vec3{0, 0, -1}));
    if (src.type == "directional") light.type = PT_LIGHT_DIRECTIONAL;
    else if (src.type == "point") light.type = PT_LIGHT_POINT;
    else throw std::string("Unsupported light '") + src.type + "'";
    scene.lights.push_back(light);
}

// scene.cu:193-255
void parse_material(gref_scene &scene, const tg::Material &src) {
    pt_material mat;
    memset(&mat, 0, sizeof(mat)); // (the reference leaves the bytes after the name's terminator uninitialised)
    strncpy(mat.name, src.name.c_str(), PT_MATERIAL_NAME_SIZE - 1);
    mat.name[PT_MATERIAL_NAME_SIZE - 1] = '\0';
    float emissiveStrength = 0, transmissionFactor = 0, ior = 1.3f, specularFactor = 0;
    for (const auto &extension : src.extensions) {
        if (extension.first == "KHR_materials_emissive_strength")
            emissiveStrength = (float)extension.second.Get("emissiveStrength").GetNumberAsDouble();
        else if (extension.first == "KHR_materials_transmission")
            transmissionFactor = (float)extension.second.Get("transmissionFactor").GetNumberAsDouble();
        else if (extension.first == "KHR_materials_ior")
            ior = (float)extension.second.Get("ior").GetNumberAsDouble();
        else if (extension.first == "KHR_materials_specular")
            specularFactor = (float)extension.second.Get("specularFactor").GetNumberAsDouble();
    }
    mat.alphaCutoff = (float)src.alphaCutoff;
    const std::vector<double> &bc = src.pbrMetallicRoughness.baseColorFactor;
    mat.baseColorFactor = pt_vec4{(float)bc[0], (float)bc[1], (float)bc[2], (float)bc[3]};
    const std::vector<double> &ef = src.emissiveFactor;
    mat.emissive = pt_vec3{(float)ef[0] * emissiveStrength, (float)ef[1] * emissiveStrength, (float)ef[2] * emissiveStrength};
    mat.metallicFactor = (float)src.pbrMetallicRoughness.metallicFactor;
    mat.roughnessFactor = (float)src.pbrMetallicRoughness.roughnessFactor;
    mat.transmission = transmissionFactor;
    mat.specular = specularFactor;
    mat.ior = ior;
    mat.baseColorTexture = src.pbrMetallicRoughness.baseColorTexture.index;
    mat.normalTexture = src.normalTexture.index;
    mat.metallicRoughnessTexture = src.pbrMetallicRoughness.metallicRoughnessTexture.index;
    if (src.alphaMode == "OPAQUE") mat.alphaMode = PT_ALPHA_OPAQUE;
    if (src.alphaMode == "MASK") mat.alphaMode = PT_ALPHA_MASK;
    if (src.alphaMode == "BLEND") mat.alphaMode = PT_ALPHA_BLEND;
    scene.materials.push_back(mat);
}

// scene.cu:257-319: bytes reinterpreted as float whatever the component type
void parse_acc_floats(std::vector<pt_vec4> &out, const tg::Model &model, int accIndex, int arity) {
    const tg::Accessor &acc = model.accessors[(size_t)accIndex];
    const tg::BufferView &bufView = model.bufferViews[(size_t)acc.bufferView];
    const tg::Buffer &buf = model.buffers[(size_t)bufView.buffer];
    if (acc.sparse.isSparse) throw std::string("sparse mesh unsupported");
    int byteStride = acc.ByteStride(bufView);
    for (size_t i = 0; i < acc.count; i++) {
        size_t byteNumber = acc.byteOffset + bufView.byteOffset + (size_t)byteStride * i;
        float item[4] = {0, 0, 0, 0};
        memcpy(item, &buf.data[byteNumber], sizeof(float) * (size_t)arity);
        out.push_back(pt_vec4{item[0], item[1], item[2], item[3]});
    }
}

// scene.cu:321-369
void parse_acc_indices(std::vector<uint32_t> &list, const tg::Model &model, int accIndex) {
    const tg::Accessor &acc = model.accessors[(size_t)accIndex];
    const tg::BufferView &bufView = model.bufferViews[(size_t)acc.bufferView];
    const tg::Buffer &buf = model.buffers[(size_t)bufView.buffer];
    if (acc.sparse.isSparse) throw std::string("sparse mesh unsupported");
    int byteStride = acc.ByteStride(bufView);
    for (size_t i = 0; i < acc.count; i++) {
        size_t byteNumber = acc.byteOffset + bufView.byteOffset + (size_t)byteStride * i;
        int index;
        switch (acc.componentType) {
        case TINYGLTF_COMPONENT_TYPE_UNSIGNED_SHORT: { uint16_t t; memcpy(&t, &buf.data[byteNumber], 2); index = (int)t; break; }
        case TINYGLTF_COMPONENT_TYPE_UNSIGNED_INT: { uint32_t t; memcpy(&t, &buf.data[byteNumber], 4); index = (int)t; break; }
        case TINYGLTF_COMPONENT_TYPE_UNSIGNED_BYTE: index = (int)buf.data[byteNumber]; break;
        default: throw std::string("Component type not implemented");
        }
        list.push_back((uint32_t)index);
    }
}

// scene.cu:371-465, with the reference's MikkTSpace itself
struct TangentJob {
    gref_scene *scene;
    int firstFace, numFaces;
};
int mk_num_faces(const SMikkTSpaceContext *c) { return ((TangentJob *)c->m_pUserData)->numFaces; }
int mk_num_verts(const SMikkTSpaceContext *c, const int iFace) {
    TangentJob *d = (TangentJob *)c->m_pUserData;
    return (int)d->scene->faces[(size_t)(d->firstFace + iFace)].vertexCount;
}
const pt_vertex &mk_vertex(const SMikkTSpaceContext *c, int iFace, int iVert) {
    TangentJob *d = (TangentJob *)c->m_pUserData;
    return d->scene->vertices[d->scene->faces[(size_t)(d->firstFace + iFace)].vertices[iVert]];
}
void mk_position(const SMikkTSpaceContext *c, float out[], const int iFace, const int iVert) {
    const pt_vertex &v = mk_vertex(c, iFace, iVert);
    out[0] = v.position.x; out[1] = v.position.y; out[2] = v.position.z;
}
void mk_normal(const SMikkTSpaceContext *c, float out[], const int iFace, const int iVert) {
    const pt_vertex &v = mk_vertex(c, iFace, iVert);
    out[0] = v.normal.x; out[1] = v.normal.y; out[2] = v.normal.z;
}
void mk_texcoord(const SMikkTSpaceContext *c, float out[], const int iFace, const int iVert) {
    const pt_vertex &v = mk_vertex(c, iFace, iVert);
    out[0] = v.texcoord0.x; out[1] = v.texcoord0.y;
}
void mk_set(const SMikkTSpaceContext *c, const float t[], const float sign, const int iFace, const int iVert) {
    pt_vertex &v = const_cast<pt_vertex &>(mk_vertex(c, iFace, iVert));
    v.tangent = pt_vec4{t[0], t[1], t[2], -sign}; // :443-446
}
void calculate_tangents(gref_scene &scene, int firstFace, int numFaces) {
    TangentJob job{&scene, firstFace, numFaces};
    SMikkTSpaceInterface iface;
    memset(&iface, 0, sizeof(iface));
    iface.m_getNumFaces = mk_num_faces;
    iface.m_getNumVerticesOfFace = mk_num_verts;
    iface.m_getPosition = mk_position;
    iface.m_getNormal = mk_normal;
    iface.m_getTexCoord = mk_texcoord;
    iface.m_setTSpaceBasic = mk_set;
    SMikkTSpaceContext ctx;
    ctx.m_pInterface = &iface;
    ctx.m_pUserData = &job;
    if (!genTangSpaceDefault(&ctx)) throw std::string("Mikktspace tangent calculation failed!");
}

// scene.cu:467-605
void parse_mesh(gref_scene &scene, const tg::Model &model, const tg::Node &node, const M4 &M) {
    const tg::Mesh &mesh = model.meshes[(size_t)node.mesh];
    for (const tg::Primitive &prim : mesh.primitives) {
        std::vector<pt_vec4> positions, tangents, normals, uvs;
        std::vector<uint32_t> indices;
        for (const auto &attr : prim.attributes) {
            if (attr.first == "POSITION") parse_acc_floats(positions, model, attr.second, 3);
            else if (attr.first == "NORMAL") parse_acc_floats(normals, model, attr.second, 3);
            else if (attr.first == "TANGENT") parse_acc_floats(tangents, model, attr.second, 4);
            else if (attr.first == "TEXCOORD_0") parse_acc_floats(uvs, model, attr.second, 2);
        }
        parse_acc_indices(indices, model, prim.indices);
        size_t vertCount = positions.size();
        bool hasN = !normals.empty(), hasT = !tangents.empty(), hasUV = !uvs.empty();
        if (!vertCount || indices.empty() || indices.size() % 3) throw std::string("bad primitive");
        if (!hasN) throw std::string("primitive without NORMAL (the reference asserts, scene.cu:586)");
        uint32_t firstVertex = (uint32_t)scene.vertices.size();
        for (size_t i = 0; i < vertCount; i++) {
            pt_vertex v;
            v.position = dehomogenise(mul(M, pt_vec4{positions[i].x, positions[i].y, positions[i].z, 1.0f}));
            v.normal = normalized(linear_mul(M, pt_vec3{normals[i].x, normals[i].y, normals[i].z}));
            if (hasT) {
                pt_vec3 tw = normalized(linear_mul(M, pt_vec3{tangents[i].x, tangents[i].y, tangents[i].z}));
                v.tangent = pt_vec4{tw.x, tw.y, tw.z, tangents[i].w};
            } else {
                v.tangent = pt_vec4{0, 0, 0, 0};
            }
            v.texcoord0 = hasUV ? pt_vec3{uvs[i].x, uvs[i].y, 0} : pt_vec3{0, 0, 0};
            scene.vertices.push_back(v);
        }
        int firstFace = (int)scene.faces.size();
        for (size_t i = 0; i < indices.size(); i += 3) {
            pt_face face;
            memset(&face, 0, sizeof(face));
            face.material = (uint32_t)prim.material;
            face.vertexCount = 3;
            for (int k = 0; k < 3; k++) face.vertices[k] = firstVertex + indices[i + (size_t)k];
            face.shading = PT_BARY_SHADING;
            scene.faces.push_back(face);
        }
        if (!hasT) calculate_tangents(scene, firstFace, (int)scene.faces.size() - firstFace);
    }
}

// scene.cu:607-626
void parse_node(gref_scene &scene, const tg::Model &model, const tg::Node &node, const M4 &parent) {
    M4 M = mul(parent, get_transform(node));
    if (node.camera >= 0) parse_camera(scene, model, node, M);
    if (node.light >= 0) parse_light(scene, model, node, M);
    if (node.mesh >= 0) parse_mesh(scene, model, node, M);
    for (const int &child : node.children) parse_node(scene, model, model.nodes[(size_t)child], M);
}

int wrap_of(int gltf) { // scene.cu:632-643
    switch (gltf) {
    case TINYGLTF_TEXTURE_WRAP_REPEAT: return PT_WRAP_REPEAT;
    case TINYGLTF_TEXTURE_WRAP_CLAMP_TO_EDGE: return PT_WRAP_CLAMP;
    case TINYGLTF_TEXTURE_WRAP_MIRRORED_REPEAT: return PT_WRAP_MIRROR;
    default: return PT_WRAP_REPEAT;
    }
}
int filter_of(int gltf) { // scene.cu:649-662
    switch (gltf) {
    case TINYGLTF_TEXTURE_FILTER_NEAREST:
    case TINYGLTF_TEXTURE_FILTER_NEAREST_MIPMAP_NEAREST:
    case TINYGLTF_TEXTURE_FILTER_NEAREST_MIPMAP_LINEAR: return PT_FILTER_POINT;
    default: return PT_FILTER_LINEAR;
    }
}

// load_image + load_texture, scene.cu:713-837: the decoded stb buffer is what the CUDA array receives
void add_texture(gref_scene &scene, const std::string &path, int magFilter, int wrapS, int wrapT) {
    size_t dot = path.find_last_of('.');
    std::string ext = dot == std::string::npos ? "" : path.substr(dot);
    bool isFloat;
    if (ext == ".png" || ext == ".jpg") isFloat = false;
    else if (ext == ".hdr") isFloat = true;
    else throw std::string("Unknown image extension: ") + ext;
    int w, h, comp;
    if (!stbi_info(path.c_str(), &w, &h, &comp)) throw std::string("STBI error: ") + stbi_failure_reason();
    if (comp != 3 && comp != 4) throw std::string("image component count unsupported");
    void *data = isFloat ? (void *)stbi_loadf(path.c_str(), &w, &h, &comp, 4) : (void *)stbi_load(path.c_str(), &w, &h, &comp, 4);
    if (!data) throw std::string("STBI error: ") + stbi_failure_reason();
    size_t bytes = (isFloat ? sizeof(float) : 1) * 4 * (size_t)w * (size_t)h;
    Texels t;
    t.bytes.assign((uint8_t *)data, (uint8_t *)data + bytes);
    stbi_image_free(data);
    pt_texture tex;
    memset(&tex, 0, sizeof(tex));
    tex.width = w;
    tex.height = h;
    tex.channels = 4;
    tex.isFloat = isFloat ? 1 : 0;
    tex.wrapS = wrap_of(wrapS);
    tex.wrapT = wrap_of(wrapT);
    tex.filter = filter_of(magFilter);
    scene.textures.push_back(tex);
    scene.texels.push_back(std::move(t));
}

std::string dir_of(const std::string &path) {
    size_t slash = path.find_last_of('/');
    return slash == std::string::npos ? std::string(".") : path.substr(0, slash);
}

} // namespace

// Scene::read_gltf, scene.cu:839-972. clear_texture = config.world_clear_color_texture ("" or NULL: none);
// default_camera is used when the file has no camera (may be NULL if it has one).
extern "C" gref_scene *gref_load(const char *filename, const char *clear_texture, const pt_camera *default_camera) {
    gref_scene *scene = new gref_scene();
    try {
        tg::Model model;
        tg::TinyGLTF loader;
        std::string err, warn;
        bool ret = tg::GetFilePathExtension(filename) == "glb" ? loader.LoadBinaryFromFile(&model, &err, &warn, filename)
                                                                : loader.LoadASCIIFromFile(&model, &err, &warn, filename);
        if (!ret) throw std::string("Failed to load glTF: ") + filename + " " + err;
        for (const auto &tex : model.textures) { // :889-909 (one texture per glTF texture, decoded from its image)
            const tg::Image &img = model.images[(size_t)tex.source];
            int mag = -1, ws = TINYGLTF_TEXTURE_WRAP_REPEAT, wt = TINYGLTF_TEXTURE_WRAP_REPEAT;
            if (tex.sampler >= 0) {
                const tg::Sampler &s = model.samplers[(size_t)tex.sampler];
                mag = s.magFilter;
                ws = s.wrapS;
                wt = s.wrapT;
            }
            add_texture(*scene, dir_of(filename) + "/" + img.uri, mag, ws, wt);
        }
        for (const auto &m : model.materials) parse_material(*scene, m);
        const tg::Scene &ms = model.scenes[(size_t)model.defaultScene];
        for (const int &n : ms.nodes) parse_node(*scene, model, model.nodes[(size_t)n], identity());
        scene->desc.clearTexture = -1;
        if (clear_texture && clear_texture[0]) { // :925-936
            add_texture(*scene, clear_texture, TINYGLTF_TEXTURE_FILTER_LINEAR, TINYGLTF_TEXTURE_WRAP_MIRRORED_REPEAT,
                        TINYGLTF_TEXTURE_WRAP_MIRRORED_REPEAT);
            scene->desc.clearTexture = (int32_t)scene->textures.size() - 1;
        }
        if (scene->cameras.empty()) {
            if (!default_camera) throw std::string("no camera in the scene and no default camera given");
            scene->cameras.push_back(*default_camera);
        }
        scene->desc.camera = scene->cameras[0];
    } catch (const std::string &e) {
        scene->error = e;
    }
    for (size_t i = 0; i < scene->textures.size(); i++) scene->textures[i].pixels = scene->texels[i].bytes.data();
    scene->desc.vertices = scene->vertices.data();
    scene->desc.vertex_count = (uint32_t)scene->vertices.size();
    scene->desc.faces = scene->faces.data();
    scene->desc.face_count = (uint32_t)scene->faces.size();
    scene->desc.lights = scene->lights.data();
    scene->desc.light_count = (uint32_t)scene->lights.size();
    scene->desc.materials = scene->materials.data();
    scene->desc.material_count = (uint32_t)scene->materials.size();
    scene->desc.textures = scene->textures.data();
    scene->desc.texture_count = (uint32_t)scene->textures.size();
    return scene;
}

extern "C" const char *gref_error(const gref_scene *s) { return s ? s->error.c_str() : "null scene"; }
extern "C" const pt_scene_desc *gref_desc(const gref_scene *s) { return &s->desc; }
extern "C" void gref_free(gref_scene *s) { delete s; }

// stb_image as the reference calls it (scene.cu:757-765): 4 channels requested
extern "C" float *gref_stbi_loadf(const char *path, int *w, int *h, int *file_channels) { return stbi_loadf(path, w, h, file_channels, 4); }
extern "C" unsigned char *gref_stbi_load(const char *path, int *w, int *h, int *file_channels) { return stbi_load(path, w, h, file_channels, 4); }
extern "C" void gref_stbi_free(void *p) { stbi_image_free(p); }
