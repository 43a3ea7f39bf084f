// bvh_build.cpp — CPU binned-SAH BVH builder and flat light table.
//
// Restates BVH::build (reference src/bvh.cu:13-218, 288-345) and LST::build
// (src/lst.cu:6-34). The node array this emits is the kernel's input, and the
// traversal order (and therefore which triangle wins an exact distance tie) is
// a function of the node numbering and the order of `indices`, so the builder
// keeps the reference's decisions exactly:
//   * centroid per face with the |(BxA)x(CxA)|/2 weight        bvh.cu:13-40
//   * 16 bins per axis, bin bounds grown from face VERTICES    bvh.cu:90-102
//   * cost nL*areaL + nR*areaR, half-area metric               bvh.cu:104-129
//   * first strictly smaller cost wins, axes x,y,z             bvh.cu:122-128
//   * in-place partition that swaps with the tail              bvh.cu:152-160
//   * failed split -> median index                             bvh.cu:162-167
//   * children numbered as a consecutive pair before recursing bvh.cu:169-170
//   * leaf = at most 2 faces; inner nodes get start=end=0      bvh.cu:143,187
//
// Large scenes: subtrees below a size threshold are independent, so they are
// built by a small thread pool into private node arrays and spliced in with
// the numbering the sequential recursion would have produced (a subtree's
// descendants occupy one contiguous index range right after the sibling pair
// that roots it).
#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "host_internal.h"
#include "hostmath.h"
#include "pthost.h"

namespace pth {

namespace {

constexpr int kBins = 16;

struct BuildInput {
    const pt_vertex *vertices;
    const pt_face *faces;
    const pt_vec3 *centroids;
};

pt_vec3 face_centroid(const pt_vertex *verts, const pt_face &f) {
    float total_area = 0;
    pt_vec3 acc = v3(0, 0, 0);
    for (uint32_t i = 2; i < f.vertexCount; i++) {
        pt_vec3 A = verts[f.vertices[0]].position;
        pt_vec3 B = verts[f.vertices[i - 1]].position;
        pt_vec3 C = verts[f.vertices[i]].position;
        pt_vec3 w = cross(cross(B, A), cross(C, A));
        float tri_area = (float)(0.5 * (double)magnitude(w));
        pt_vec3 tri_centroid = scale(add(add(A, B), C), 1 / 3.0f);
        acc = add(acc, scale(tri_centroid, tri_area));
        total_area += tri_area;
    }
    if (total_area == 0) return verts[f.vertices[0]].position;
    return div(acc, total_area);
}

inline void grow_face(const BuildInput &in, Box &box, uint32_t face_index) {
    const pt_face &f = in.faces[face_index];
    for (uint32_t j = 0; j < f.vertexCount; j++) box.grow(in.vertices[f.vertices[j]].position);
}

// A subtree under construction: nodes[0] is the subtree root (numbered by the
// parent), nodes[1..] its descendants in the order the sequential build would
// number them. Child links are relative to this vector while building.
struct Subtree {
    std::vector<pt_bvh_node> nodes;
    uint32_t skipped = 0;
    int max_depth = 0;
};

void node_bounds(const BuildInput &in, const uint32_t *indices, pt_bvh_node &n) {
    Box b;
    for (uint32_t i = n.start; i < n.end; i++) grow_face(in, b, indices[i]);
    n.bmin = b.lo;
    n.bmax = b.hi;
}

// bvh.cu:72-134. Returns best axis, or -1 if every axis is degenerate (all centroids of the range equal): the
// reference asserts there (bvh.cu:131) and aborts; this builder goes on to the reference's own fallback for a failed
// split, the median index (bvh.cu:162-167), so that scenes with coincident faces build. Every tree the reference
// can build comes out identical.
int best_split(const BuildInput &in, const uint32_t *indices, uint32_t start, uint32_t end, float *split_pos) {
    Box cb;
    for (uint32_t i = start; i < end; i++) cb.grow(in.centroids[indices[i]]);

    float best_cost = 1e30f;
    int best_axis = -1;
    for (int a = 0; a < 3; a++) {
        float lo = comp(cb.lo, a), hi = comp(cb.hi, a);
        if (std::fabs(lo - hi) < FLT_EPSILON) continue;

        Box bin_box[kBins];
        int bin_count[kBins] = {0};
        float width = (hi - lo) / kBins;
        float inv_width = 1 / width;
        for (uint32_t i = start; i < end; i++) {
            float c = comp(in.centroids[indices[i]], a);
            int b = std::min(kBins - 1, (int)((c - lo) * inv_width));
            grow_face(in, bin_box[b], indices[i]);
            bin_count[b]++;
        }

        float l_area[kBins - 1], r_area[kBins - 1];
        int l_count[kBins - 1], r_count[kBins - 1];
        Box l_box, r_box;
        int l_sum = 0, r_sum = 0;
        for (int i = 0; i < kBins - 1; i++) {
            l_sum += bin_count[i];
            l_count[i] = l_sum;
            l_box.grow(bin_box[i]);
            l_area[i] = l_box.area();
            r_sum += bin_count[kBins - 1 - i];
            r_count[kBins - 2 - i] = r_sum;
            r_box.grow(bin_box[kBins - 1 - i]);
            r_area[kBins - 2 - i] = r_box.area();
        }
        for (int i = 0; i < kBins - 1; i++) {
            float cost = l_count[i] * l_area[i] + r_count[i] * r_area[i];
            if (cost < best_cost) {
                best_axis = a;
                *split_pos = lo + width * (i + 1);
                best_cost = cost;
            }
        }
    }
    return best_axis;
}

// Splits [start,end) in place; returns the first index of the right half.
// Sets *ok=false if the node is a leaf (<= 2 faces).
uint32_t split_range(const BuildInput &in, uint32_t *indices, uint32_t start, uint32_t end, uint32_t *skipped,
                     bool *is_leaf) {
    int count = (int)(end - start);
    if (count <= 2) {
        *is_leaf = true;
        return 0;
    }
    *is_leaf = false;
    float pos = 0;
    int axis = best_split(in, indices, start, end, &pos);

    int i = (int)start;
    if (axis >= 0) {
        int j = (int)end - 1;
        while (i <= j) {
            if (comp(in.centroids[indices[i]], axis) < pos) {
                i++;
            } else {
                std::swap(indices[i], indices[j]);
                j--;
            }
        }
    }
    bool ok = axis >= 0 && (int)start < i && i < (int)end;
    if (!ok) {
        (*skipped)++;
        i = (int)start + count / 2;
    }
    return (uint32_t)i;
}

// Sequential recursion on a private node vector (bvh.cu:136-192).
void build_subtree(const BuildInput &in, uint32_t *indices, Subtree &st, uint32_t node_index, int depth) {
    st.max_depth = std::max(st.max_depth, depth);
    uint32_t start = st.nodes[node_index].start, end = st.nodes[node_index].end;
    bool is_leaf;
    uint32_t mid = split_range(in, indices, start, end, &st.skipped, &is_leaf);
    if (is_leaf) return;

    uint32_t left = (uint32_t)st.nodes.size();
    st.nodes.emplace_back();
    st.nodes.emplace_back();
    uint32_t right = left + 1;
    st.nodes[left].start = start;
    st.nodes[left].end = mid;
    st.nodes[right].start = mid;
    st.nodes[right].end = end;
    st.nodes[left].leftChild = st.nodes[left].rightChild = 0;
    st.nodes[right].leftChild = st.nodes[right].rightChild = 0;

    pt_bvh_node &n = st.nodes[node_index];
    n.leftChild = left;
    n.rightChild = right;
    n.start = n.end = 0;
    node_bounds(in, indices, st.nodes[left]);
    node_bounds(in, indices, st.nodes[right]);
    build_subtree(in, indices, st, left, depth + 1);
    build_subtree(in, indices, st, right, depth + 1);
}

// Top of the tree, built sequentially down to ranges of at most `grain`
// faces; those become jobs for the pool.
struct TopNode {
    uint32_t start, end;
    pt_vec3 bmin, bmax;
    int left = -1, right = -1; // indices into top vector
    int job = -1;              // >=0: subtree job id (this node is the job's root)
    int depth = 0;
};

struct Job {
    uint32_t start, end;
    pt_vec3 bmin, bmax;
    int depth;
    Subtree result;
};

void build_top(const BuildInput &in, uint32_t *indices, std::vector<TopNode> &top, std::vector<Job> &jobs, int idx,
               uint32_t grain, uint32_t *skipped, int *max_depth) {
    *max_depth = std::max(*max_depth, top[idx].depth);
    uint32_t start = top[idx].start, end = top[idx].end;
    if (end - start <= grain) {
        Job j;
        j.start = start;
        j.end = end;
        j.bmin = top[idx].bmin;
        j.bmax = top[idx].bmax;
        j.depth = top[idx].depth;
        top[idx].job = (int)jobs.size();
        jobs.push_back(std::move(j));
        return;
    }
    bool is_leaf;
    uint32_t mid = split_range(in, indices, start, end, skipped, &is_leaf);
    // grain >= 2, so a range above grain is never a leaf
    int l = (int)top.size();
    top.emplace_back();
    top.emplace_back();
    int r = l + 1;
    top[l].start = start;
    top[l].end = mid;
    top[r].start = mid;
    top[r].end = end;
    top[l].depth = top[r].depth = top[idx].depth + 1;
    pt_bvh_node tmp;
    tmp.start = start;
    tmp.end = mid;
    node_bounds(in, indices, tmp);
    top[l].bmin = tmp.bmin;
    top[l].bmax = tmp.bmax;
    tmp.start = mid;
    tmp.end = end;
    node_bounds(in, indices, tmp);
    top[r].bmin = tmp.bmin;
    top[r].bmax = tmp.bmax;
    top[idx].left = l;
    top[idx].right = r;
    build_top(in, indices, top, jobs, l, grain, skipped, max_depth);
    build_top(in, indices, top, jobs, r, grain, skipped, max_depth);
}

// Number of descendants (excluding the node itself) under a top node.
uint32_t count_descendants(const std::vector<TopNode> &top, const std::vector<Job> &jobs, int idx,
                           std::vector<uint32_t> &desc) {
    uint32_t d;
    if (top[idx].job >= 0) {
        d = (uint32_t)jobs[top[idx].job].result.nodes.size() - 1;
    } else {
        d = 2 + count_descendants(top, jobs, top[idx].left, desc) + count_descendants(top, jobs, top[idx].right, desc);
    }
    desc[idx] = d;
    return d;
}

// Emits the node `idx` at position `self`; its descendants go to
// [first_free, first_free + desc[idx]).
void emit(const std::vector<TopNode> &top, const std::vector<Job> &jobs, const std::vector<uint32_t> &desc, int idx,
          uint32_t self, uint32_t first_free, pt_bvh_node *out) {
    const TopNode &t = top[idx];
    if (t.job >= 0) {
        const Subtree &st = jobs[t.job].result;
        // st.nodes[0] -> self; st.nodes[k>=1] -> first_free + k - 1
        auto reloc = [&](uint32_t k) { return k == 0 ? self : first_free + k - 1; };
        for (uint32_t k = 0; k < st.nodes.size(); k++) {
            pt_bvh_node n = st.nodes[k];
            bool leaf = n.end - n.start > 0;
            if (!leaf && (n.leftChild || n.rightChild)) {
                n.leftChild = reloc(n.leftChild);
                n.rightChild = reloc(n.rightChild);
            }
            out[reloc(k)] = n;
        }
        return;
    }
    pt_bvh_node n;
    n.bmin = t.bmin;
    n.bmax = t.bmax;
    n.start = n.end = 0;
    n.leftChild = first_free;
    n.rightChild = first_free + 1;
    out[self] = n;
    uint32_t left_desc_at = first_free + 2;
    uint32_t right_desc_at = left_desc_at + desc[t.left];
    emit(top, jobs, desc, t.left, first_free, left_desc_at, out);
    emit(top, jobs, desc, t.right, first_free + 1, right_desc_at, out);
}

} // namespace

} // namespace pth

using namespace pth;

extern "C" int pth_bvh_build(const pt_scene_desc *scene, pth_bvh *out) {
    if (!scene || !out) return set_error("pth_bvh_build: null argument");
    memset(out, 0, sizeof(*out));
    uint32_t n = scene->face_count;
    if (n == 0) return set_error("pth_bvh_build: scene has no faces");
    for (uint32_t i = 0; i < n; i++) {
        const pt_face &f = scene->faces[i];
        if (f.vertexCount != 3) return set_error("pth_bvh_build: face %u is not a triangle", i);
        for (int k = 0; k < 3; k++)
            if (f.vertices[k] >= scene->vertex_count)
                return set_error("pth_bvh_build: face %u references vertex %u of %u", i, f.vertices[k],
                                 scene->vertex_count);
    }

    uint32_t capacity = 2 * n - 1;
    pt_bvh_node *nodes = (pt_bvh_node *)calloc(capacity, sizeof(pt_bvh_node));
    uint32_t *indices = (uint32_t *)malloc(sizeof(uint32_t) * n);
    std::vector<pt_vec3> centroids(n);
    if (!nodes || !indices) {
        free(nodes);
        free(indices);
        return set_error("pth_bvh_build: out of memory");
    }
    for (uint32_t i = 0; i < n; i++) indices[i] = i;
    for (uint32_t i = 0; i < n; i++) centroids[i] = face_centroid(scene->vertices, scene->faces[i]);

    BuildInput in{scene->vertices, scene->faces, centroids.data()};

    unsigned hw = std::thread::hardware_concurrency();
    if (const char *env = getenv("PTH_BUILD_THREADS")) hw = (unsigned)atoi(env);
    if (hw == 0) hw = 1;
    if (hw > 32) hw = 32;
    uint32_t grain = (hw <= 1 || n < 65536) ? n : std::max<uint32_t>(4096, n / (hw * 16));

    std::vector<TopNode> top(1);
    std::vector<Job> jobs;
    top[0].start = 0;
    top[0].end = n;
    top[0].depth = 1;
    {
        pt_bvh_node tmp;
        tmp.start = 0;
        tmp.end = n;
        node_bounds(in, indices, tmp);
        top[0].bmin = tmp.bmin;
        top[0].bmax = tmp.bmax;
    }
    uint32_t skipped = 0;
    int max_depth = 0;
    build_top(in, indices, top, jobs, 0, grain, &skipped, &max_depth);

    std::atomic<size_t> next{0};
    auto worker = [&]() {
        for (;;) {
            size_t k = next.fetch_add(1);
            if (k >= jobs.size()) break;
            Job &j = jobs[k];
            j.result.nodes.reserve(2 * (size_t)(j.end - j.start));
            j.result.nodes.emplace_back();
            pt_bvh_node &root = j.result.nodes[0];
            root.bmin = j.bmin;
            root.bmax = j.bmax;
            root.start = j.start;
            root.end = j.end;
            root.leftChild = root.rightChild = 0;
            build_subtree(in, indices, j.result, 0, j.depth);
        }
    };
    unsigned nthreads = (unsigned)std::min<size_t>(hw, jobs.size());
    if (nthreads <= 1) {
        worker();
    } else {
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < nthreads; t++) pool.emplace_back(worker);
        for (auto &t : pool) t.join();
    }
    for (const Job &j : jobs) {
        skipped += j.result.skipped;
        max_depth = std::max(max_depth, j.result.max_depth);
    }

    std::vector<uint32_t> desc(top.size());
    uint32_t total = 1 + count_descendants(top, jobs, 0, desc);
    emit(top, jobs, desc, 0, 0, 1, nodes);

    out->nodes = nodes;
    out->node_capacity = capacity;
    out->node_count = total;
    out->indices = indices;
    out->index_count = n;
    out->total_skipped_faces = skipped;
    out->max_depth = max_depth;
    // calculate_stats, bvh.cu:194-207
    uint32_t leaves = 0;
    float leaf_faces = 0;
    for (uint32_t i = 0; i < total; i++) {
        uint32_t faces = nodes[i].end - nodes[i].start;
        if (faces > 0) {
            leaves++;
            leaf_faces += faces;
        }
    }
    out->number_leaves = leaves;
    out->average_leaf_size = leaf_faces / leaves;

    pth_log(4, "\nbvh_t stats:\n");
    pth_log(4, "  node count = %u\n", total);
    pth_log(4, "  optimal node count = %u\n", capacity);
    pth_log(4, "  number leafs = %u\n", leaves);
    pth_log(4, "  skipped faces = %u / %u = %.3f\n", skipped, n, skipped / (float)n);
    pth_log(4, "  average leaf size = %.2f\n", out->average_leaf_size);
    pth_log(4, "  max tree height =  %d\n", max_depth);

    // bvh.cu:338-342 — the traversal stack holds 64 entries
    if (max_depth + 1 > 64) {
        pth_bvh_free(out);
        return set_error("bvh max height (%d) is too large for a 64-entry traversal stack", max_depth);
    }
    return 0;
}

extern "C" void pth_bvh_free(pth_bvh *bvh) {
    if (!bvh) return;
    free(bvh->nodes);
    free(bvh->indices);
    memset(bvh, 0, sizeof(*bvh));
}

extern "C" int pth_lst_build(const pt_scene_desc *scene, pt_lst_node **nodes, uint32_t *count) {
    if (!scene || !nodes || !count) return set_error("pth_lst_build: null argument");
    std::vector<pt_lst_node> list;
    for (uint32_t i = 0; i < scene->face_count; i++) {
        const pt_face &f = scene->faces[i];
        if (f.material >= scene->material_count)
            return set_error("pth_lst_build: face %u references material %u of %u", i, f.material,
                             scene->material_count);
        if (max_component(scene->materials[f.material].emissive) > 0) list.push_back({PT_LST_SOURCE_FACE, i});
    }
    for (uint32_t i = 0; i < scene->light_count; i++) list.push_back({PT_LST_SOURCE_LIGHT, i});
    *count = (uint32_t)list.size();
    *nodes = (pt_lst_node *)malloc(sizeof(pt_lst_node) * std::max<size_t>(1, list.size()));
    if (!*nodes) return set_error("pth_lst_build: out of memory");
    std::copy(list.begin(), list.end(), *nodes);
    if (list.empty()) pth_log(2, "No lights found in scene.\n");
    return 0;
}
