// fold_check.cpp — CPU check of the upload-time tree transformations in pathtracer_amd/csrc/core/layout.h
// (bake_bvh -> bake_bvh4). Test infrastructure: built by tests/test_oracle_and_host.py with g++.
// Returns 0 and fills `out` = {binary inner nodes, leaves, largest leaf, wide nodes, wide stack bound, tree height}
// or a negative code naming the invariant that failed.
#include <cstdint>
#include <vector>

#include "layout.h"

using namespace ptc;

namespace {
struct Cover {
    std::vector<uint32_t> seen;
    uint32_t leaves = 0, largest = 0;
    bool leaf(uint32_t ref, uint32_t limit) {
        uint32_t first = ref & 0x0FFFFFFFu, count = ((ref >> 28) & 7u) + 1u;
        if (count > limit || first + count > seen.size()) return false;
        for (uint32_t i = 0; i < count; i++) seen[first + i]++;
        leaves++;
        if (count > largest) largest = count;
        return true;
    }
    bool all_once() const {
        for (uint32_t s : seen)
            if (s != 1) return false;
        return true;
    }
};
bool inside(const float *lo, const float *hi, const GTri &t) {
    for (int a = 0; a < 3; a++) {
        float v[3] = {t.v0[a], t.v0[a] + t.e1[a], t.v0[a] + t.e2[a]};
        for (float x : v)
            if (x < lo[a] - 1e-4f || x > hi[a] + 1e-4f) return false;
    }
    return true;
}
} // namespace

extern "C" int fold_check(const pt_scene_desc *scene, const pt_bvh_node *nodes, uint32_t node_count, const uint32_t *indices,
                          uint32_t index_count, int64_t *out) {
    std::vector<GShade> shade;
    std::vector<GMaterial> mats;
    bake_shading(*scene, shade, mats);
    BakedBvh b;
    if (!bake_bvh(shade, nodes, node_count, indices, index_count, b).empty()) return -1;
    bake_bvh4(b);
    const uint32_t limit = 8; // the leaf format's limit (the reference's builder stops at 2, bvh.cu:143)
    // binary tree: every triangle in exactly one leaf, leaves within the limit, every triangle inside the box
    // its parent holds for it, children after parents
    Cover c;
    c.seen.assign(index_count, 0);
    if (b.root & kLeafBit) {
        if (!c.leaf(b.root, limit)) return -2;
    } else {
        for (size_t k = 0; k < b.nodes.size(); k++) {
            const GNode &g = b.nodes[k];
            const uint32_t ref[2] = {g.left, g.right};
            const float *lo[2] = {g.lmin, g.rmin}, *hi[2] = {g.lmax, g.rmax};
            for (int s = 0; s < 2; s++) {
                if (ref[s] & kLeafBit) {
                    if (!c.leaf(ref[s], limit)) return -3;
                    uint32_t first = ref[s] & 0x0FFFFFFFu, count = ((ref[s] >> 28) & 7u) + 1u;
                    for (uint32_t i = 0; i < count; i++)
                        if (!inside(lo[s], hi[s], b.tris[first + i])) return -4;
                } else if (ref[s] <= k || ref[s] >= b.nodes.size()) {
                    return -5;
                }
            }
        }
    }
    if (!c.all_once()) return -6;
    // four-wide tree: the same cover, through two to four children per node
    Cover w;
    w.seen.assign(index_count, 0);
    if (b.root4 & kLeafBit) {
        if (!w.leaf(b.root4, limit)) return -8;
    } else {
        for (size_t k = 0; k < b.nodes4.size(); k++) {
            const GNode4 &g = b.nodes4[k];
            int children = 0;
            for (int s = 0; s < 4; s++) {
                if (g.child[s] == kNoChild) continue;
                children++;
                if (g.child[s] & kLeafBit) {
                    if (!w.leaf(g.child[s], limit)) return -9;
                    uint32_t first = g.child[s] & 0x0FFFFFFFu, count = ((g.child[s] >> 28) & 7u) + 1u;
                    float lo[3] = {g.lo[0][s], g.lo[1][s], g.lo[2][s]}, hi[3] = {g.hi[0][s], g.hi[1][s], g.hi[2][s]};
                    for (uint32_t i = 0; i < count; i++)
                        if (!inside(lo, hi, b.tris[first + i])) return -10;
                } else if (g.child[s] <= k || g.child[s] >= b.nodes4.size()) {
                    return -11;
                }
            }
            if (children < 2) return -12;
        }
    }
    if (!w.all_once()) return -13;
    out[0] = (int64_t)b.nodes.size();
    out[1] = c.leaves;
    out[2] = c.largest;
    out[3] = (int64_t)b.nodes4.size();
    out[4] = b.stack4;
    out[5] = b.height;
    return 0;
}
