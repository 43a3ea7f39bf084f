// layout.h — device-side records and the host-side baker that produces them.
//
// The reference keeps array-of-struct scene data on the device and chases
// indices -> faces -> vertices per triangle test (5 dependent loads,
// src/bvh.cu:253-255, src/intersect.cu:85-98) and node -> 2 child nodes per inner
// visit (3 dependent 40 B loads, src/bvh.cu:258-262). Here every step of the walk
// is ONE aligned record:
//
//   GNode     64 B  both children's boxes + both child references. One inner
//                   visit = one 64 B fetch (4 x dwordx4), no second hop.
//   GTri      48 B  v0, e1 = v1-v0, e2 = v2-v0, face id; stored in leaf order so a
//                   leaf's triangles are contiguous. e1/e2 are the float
//                   differences the reference recomputes per test
//                   (intersect.cu:19-20), so the arithmetic downstream is
//                   unchanged.
//   GShade   144 B  everything intersect_face reads to build the hit record
//                   (positions, normals, tangents, uvs, handedness, material),
//                   indexed by the ORIGINAL face index; fetched once per shaded
//                   hit instead of once per closer hit.
//   GMaterial 64 B  the material fields the kernel reads.
//
// Child reference: bit 31 set = leaf, bits 30..28 = triangle count - 1,
// bits 27..0 = first triangle; bit 31 clear = index of a GNode.
#pragma once

#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "pt_types.h"

namespace ptc {

struct alignas(16) GNode {
    float lmin[3], lmax[3];
    float rmin[3], rmax[3];
    uint32_t left, right;
    uint32_t pad[2];
};
static_assert(sizeof(GNode) == 64, "GNode must be one 64-byte line");

// Four-wide node for rays whose direction reciprocals are all finite: the reference's binary tree
// with every second level folded away at upload (collapse of the child with the larger box first).
// Skipping an inner node's own box test is exact for such rays: (b - o) * rinv is monotone in b, so a
// ray that passes a child's slab test passes every enclosing box's test too. Children are stored
// axis-major so the four slab tests run as packed float math.
struct alignas(16) GNode4 {
    float lo[3][4];
    float hi[3][4];
    uint32_t child[4]; // node index | leaf reference | kNoChild
    uint32_t pad[4];
};
static_assert(sizeof(GNode4) == 128, "GNode4 is one 128-byte line");
constexpr uint32_t kNoChild = 0x7FFFFFFFu; // == kDone in kernel.h: neither a node nor a leaf

struct alignas(16) GTri {
    float v0[3], e1[3], e2[3];
    uint32_t face;
    uint32_t material; // copy of the face's material index: lets the material fetch start with the hit
    uint32_t pad;
};
static_assert(sizeof(GTri) == 48, "GTri is three 16-byte loads");

enum : uint32_t { kShadeFlat = 1u, kShadeIsLight = 2u };

struct alignas(16) GShade {
    float pos[3][3];
    float nrm[3][3];
    float tan[3][3];
    float uv[3][2];
    float handed;      // A.tangent.w (or face.tangent.w when flat)
    uint32_t material;
    uint32_t flags;    // kShadeFlat | kShadeIsLight | (times listed in the LST) << 8
};
static_assert(sizeof(GShade) == 144, "GShade is nine 16-byte loads");

struct alignas(16) GMaterial {
    float baseColorFactor[4];
    float emissive[3];
    float specular;
    float metallic, roughness, alphaCutoff;
    uint32_t alphaMode;
    int32_t baseColorTexture, normalTexture, metallicRoughnessTexture;
    uint32_t pad;
};
static_assert(sizeof(GMaterial) == 64, "GMaterial is four 16-byte loads");

// One record per LST node, everything sample_light_source needs in one 80-byte fetch
// (the reference chases lst node -> light | face -> 3 vertices -> material). The radiometric
// conversion (renderer.cu:146-162), the sun direction and the emitter's normal and area
// (renderer.cu:102-105) depend only on the scene, so the host computes them once with the
// reference's operation order.
struct alignas(16) GLight {
    uint32_t kind;      // 0 point, 1 directional, 2 emissive face
    uint32_t face;      // kind 2: face index
    float a[3];         // point: position; directional: normalize(-direction); face: vertex A
    float b[3], c[3];   // face: vertices B, C
    float radiance[3];  // rgb_to_radiometric(color, intensity) | (emissive, 1)
    float normal[3];    // face: tri_cross / |tri_cross|
    float area;         // face: 0.5 * |tri_cross|
    float pad[2];
};
static_assert(sizeof(GLight) == 80, "GLight is five 16-byte loads");

struct GTexture {
    const void *texels; // device pointer, RGBA float or RGBA8
    int32_t width, height;
    int32_t isFloat;
    int32_t wrapS, wrapT, filter;
};

constexpr uint32_t kLeafBit = 0x80000000u;
constexpr uint32_t kMaxLeafTris = 8;
constexpr uint32_t kMaxTris = 1u << 28;

inline uint32_t leaf_ref(uint32_t first_tri, uint32_t count) { return kLeafBit | ((count - 1) << 28) | first_tri; }

struct BakedBvh {
    std::vector<GNode> nodes;
    std::vector<GTri> tris;
    uint32_t root = 0; // child reference of the root
    int height = 0;    // nodes on the longest root-to-leaf path
    std::vector<GNode4> nodes4;
    uint32_t root4 = 0;   // child reference of the wide root (a leaf reference when the tree is one leaf)
    int stack4 = 0;       // most stack entries a wide walk can hold at once
};

// Converts the reference-format BVH. Returns "" or an error message.
// Triangle positions come from the baked shading records (the library's own
// copy of the scene), so the caller's scene arrays need not outlive the upload.
inline std::string bake_bvh(const std::vector<GShade> &shade, const pt_bvh_node *nodes, uint32_t node_count,
                            const uint32_t *indices, uint32_t index_count, BakedBvh &out) {
    if (node_count == 0) return "BVH has no nodes";
    if (index_count > kMaxTris) return "BVH references more than 2^28 triangles";
    out.nodes.clear();
    out.tris.clear();
    out.tris.reserve(index_count);

    // Triangles in `indices` order: leaf [start, end) -> GTri [start, end).
    for (uint32_t i = 0; i < index_count; i++) {
        uint32_t f = indices[i];
        if (f >= shade.size()) return "BVH index list references face " + std::to_string(f) + " past the scene";
        const float *a = shade[f].pos[0], *b = shade[f].pos[1], *c = shade[f].pos[2];
        GTri t;
        memset(&t, 0, sizeof(t));
        t.v0[0] = a[0]; t.v0[1] = a[1]; t.v0[2] = a[2];
        t.e1[0] = b[0] - a[0]; t.e1[1] = b[1] - a[1]; t.e1[2] = b[2] - a[2];
        t.e2[0] = c[0] - a[0]; t.e2[1] = c[1] - a[1]; t.e2[2] = c[2] - a[2];
        t.face = f;
        t.material = shade[f].material;
        out.tris.push_back(t);
    }

    auto is_leaf = [&](const pt_bvh_node &n) { return n.end - n.start > 0; };
    auto make_leaf = [&](const pt_bvh_node &n, uint32_t *ref) -> std::string {
        uint32_t count = n.end - n.start;
        if (n.end > index_count || n.start >= n.end) return "leaf range outside the index list";
        if (count > kMaxLeafTris) return "leaf with " + std::to_string(count) + " faces (at most 8 supported)";
        *ref = leaf_ref(n.start, count);
        return "";
    };

    if (is_leaf(nodes[0])) {
        out.height = 1;
        return make_leaf(nodes[0], &out.root);
    }

    // Inner nodes keep the reference's numbering order (dense remap).
    std::vector<uint32_t> remap(node_count, 0xFFFFFFFFu);
    uint32_t inner = 0;
    for (uint32_t i = 0; i < node_count; i++)
        if (!is_leaf(nodes[i])) remap[i] = inner++;
    out.nodes.resize(inner);
    for (uint32_t i = 0; i < node_count; i++) {
        const pt_bvh_node &n = nodes[i];
        if (is_leaf(n)) continue;
        if (n.leftChild >= node_count || n.rightChild >= node_count) return "child index outside the node array";
        if (n.leftChild <= i || n.rightChild <= i) return "child index does not follow its parent (cycle?)";
        GNode g;
        memset(&g, 0, sizeof(g));
        const pt_bvh_node &l = nodes[n.leftChild], &r = nodes[n.rightChild];
        g.lmin[0] = l.bmin.x; g.lmin[1] = l.bmin.y; g.lmin[2] = l.bmin.z;
        g.lmax[0] = l.bmax.x; g.lmax[1] = l.bmax.y; g.lmax[2] = l.bmax.z;
        g.rmin[0] = r.bmin.x; g.rmin[1] = r.bmin.y; g.rmin[2] = r.bmin.z;
        g.rmax[0] = r.bmax.x; g.rmax[1] = r.bmax.y; g.rmax[2] = r.bmax.z;
        std::string err;
        if (is_leaf(l)) err = make_leaf(l, &g.left); else g.left = remap[n.leftChild];
        if (!err.empty()) return err;
        if (is_leaf(r)) err = make_leaf(r, &g.right); else g.right = remap[n.rightChild];
        if (!err.empty()) return err;
        out.nodes[remap[i]] = g;
    }
    out.root = 0;

    // Height (children follow parents, so one reverse sweep suffices).
    std::vector<int> h(inner, 1);
    int height = 1;
    for (uint32_t k = inner; k-- > 0;) {
        const GNode &g = out.nodes[k];
        int hl = (g.left & kLeafBit) ? 1 : h[g.left];
        int hr = (g.right & kLeafBit) ? 1 : h[g.right];
        h[k] = 1 + (hl > hr ? hl : hr);
    }
    height = h[0];
    out.height = height;
    return "";
}

// Folds the baked binary tree into four-wide nodes. Call after bake_bvh.
inline void bake_bvh4(BakedBvh &bvh) {
    struct Source { // the binary tree the wide one is made from
        const std::vector<GNode> &nodes;
        uint32_t root;
    } b{bvh.nodes, bvh.root};
    bvh.nodes4.clear();
    bvh.root4 = b.root;
    bvh.stack4 = 0;
    if (b.nodes.empty() || (b.root & kLeafBit)) return; // single leaf
    struct Box { float lo[3], hi[3]; uint32_t ref; };
    auto area = [](const Box &x) {
        float dx = x.hi[0] - x.lo[0], dy = x.hi[1] - x.lo[1], dz = x.hi[2] - x.lo[2];
        return dx * dy + dy * dz + dz * dx;
    };
    auto children_of = [&](uint32_t node, Box out[2]) {
        const GNode &g = b.nodes[node];
        for (int a = 0; a < 3; a++) {
            out[0].lo[a] = g.lmin[a]; out[0].hi[a] = g.lmax[a];
            out[1].lo[a] = g.rmin[a]; out[1].hi[a] = g.rmax[a];
        }
        out[0].ref = g.left;
        out[1].ref = g.right;
    };
    // wide node index of a binary inner node that becomes a wide node's root, assigned on demand
    std::vector<uint32_t> wide_of(b.nodes.size(), 0xFFFFFFFFu);
    std::vector<uint32_t> todo; // binary nodes whose wide node still has to be filled, in allocation order
    auto wide_index = [&](uint32_t node) {
        if (wide_of[node] == 0xFFFFFFFFu) {
            wide_of[node] = (uint32_t)bvh.nodes4.size();
            bvh.nodes4.emplace_back();
            todo.push_back(node);
        }
        return wide_of[node];
    };
    bvh.root4 = wide_index(0);
    std::vector<int> need; // stack entries a walk below wide node i may need, filled afterwards
    for (size_t k = 0; k < todo.size(); k++) {
        uint32_t node = todo[k];
        Box c[4];
        int count = 2;
        children_of(node, c);
        while (count < 4) { // open the inner child with the largest box
            int best = -1;
            float best_area = -1.f;
            for (int i = 0; i < count; i++)
                if (!(c[i].ref & kLeafBit) && area(c[i]) > best_area) {
                    best = i;
                    best_area = area(c[i]);
                }
            if (best < 0) break;
            Box two[2];
            children_of(c[best].ref, two);
            // keep left-to-right order: the opened child's children take its place
            for (int i = count; i > best + 1; i--) c[i] = c[i - 1];
            c[best] = two[0];
            c[best + 1] = two[1];
            count++;
        }
        GNode4 w;
        memset(&w, 0, sizeof(w));
        for (int i = 0; i < 4; i++) {
            const Box &src = c[i < count ? i : 0];
            for (int a = 0; a < 3; a++) {
                w.lo[a][i] = src.lo[a];
                w.hi[a][i] = src.hi[a];
            }
            if (i >= count) w.child[i] = kNoChild;
            else if (src.ref & kLeafBit) w.child[i] = src.ref;
            else w.child[i] = wide_index(src.ref);
        }
        bvh.nodes4[wide_of[node]] = w;
    }
    // stack bound: children follow parents in allocation order, so one reverse sweep suffices
    need.assign(bvh.nodes4.size(), 0);
    for (size_t k = bvh.nodes4.size(); k-- > 0;) {
        const GNode4 &w = bvh.nodes4[k];
        int count = 0, deepest = 0;
        for (int i = 0; i < 4; i++) {
            if (w.child[i] == kNoChild) continue;
            count++;
            if (!(w.child[i] & kLeafBit) && need[w.child[i]] > deepest) deepest = need[w.child[i]];
        }
        need[k] = (count - 1) + deepest;
    }
    bvh.stack4 = need[bvh.root4];
}

inline void bake_shading(const pt_scene_desc &scene, std::vector<GShade> &shade, std::vector<GMaterial> &mats) {
    shade.resize(scene.face_count);
    for (uint32_t f = 0; f < scene.face_count; f++) {
        const pt_face &face = scene.faces[f];
        GShade s;
        memset(&s, 0, sizeof(s));
        for (int k = 0; k < 3; k++) {
            const pt_vertex &v = scene.vertices[face.vertices[k]];
            s.pos[k][0] = v.position.x; s.pos[k][1] = v.position.y; s.pos[k][2] = v.position.z;
            s.nrm[k][0] = v.normal.x;   s.nrm[k][1] = v.normal.y;   s.nrm[k][2] = v.normal.z;
            s.tan[k][0] = v.tangent.x;  s.tan[k][1] = v.tangent.y;  s.tan[k][2] = v.tangent.z;
            s.uv[k][0] = v.texcoord0.x; s.uv[k][1] = v.texcoord0.y;
        }
        s.handed = scene.vertices[face.vertices[0]].tangent.w;
        if (face.shading == PT_FLAT_SHADING) {
            s.flags |= kShadeFlat;
            s.nrm[0][0] = face.normal.x;  s.nrm[0][1] = face.normal.y;  s.nrm[0][2] = face.normal.z;
            s.tan[0][0] = face.tangent.x; s.tan[0][1] = face.tangent.y; s.tan[0][2] = face.tangent.z;
            s.handed = face.tangent.w;
        }
        s.material = face.material;
        shade[f] = s;
    }
    mats.resize(scene.material_count);
    for (uint32_t m = 0; m < scene.material_count; m++) {
        const pt_material &src = scene.materials[m];
        GMaterial g;
        memset(&g, 0, sizeof(g));
        g.baseColorFactor[0] = src.baseColorFactor.x; g.baseColorFactor[1] = src.baseColorFactor.y;
        g.baseColorFactor[2] = src.baseColorFactor.z; g.baseColorFactor[3] = src.baseColorFactor.w;
        g.emissive[0] = src.emissive.x; g.emissive[1] = src.emissive.y; g.emissive[2] = src.emissive.z;
        g.specular = src.specular;
        g.metallic = src.metallicFactor;
        g.roughness = src.roughnessFactor;
        g.alphaCutoff = src.alphaCutoff;
        g.alphaMode = src.alphaMode;
        g.baseColorTexture = src.baseColorTexture;
        g.normalTexture = src.normalTexture;
        g.metallicRoughnessTexture = src.metallicRoughnessTexture;
        mats[m] = g;
    }
}

} // namespace ptc
