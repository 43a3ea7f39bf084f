// tangents.cpp — per-vertex tangents for primitives that ship normals but no
// TANGENT attribute.
//
// The reference calls the vendored MikkTSpace library with its default 180
// degree threshold (src/scene.cu:371-465 -> src/MikkTSpace/mikktspace.c:224-420)
// and stores each returned face-corner tangent into the INDEXED vertex, with
// w = -sign, last writer wins (scene.cu:437-447). The tangents enter the
// (non-orthonormal) shading basis of every bounce, so they are kernel input and
// this file restates the published algorithm for triangle meshes:
//
//   1. weld corners with identical (position, normal, uv)      mikktspace.c:451-576
//   2. set aside triangles with two coinciding positions        :281-302, 1737-1818
//   3. per triangle: first-order derivatives vOs/vOt, orientation = sign of the
//      uv area, "group with any" when the uv area is ~0         :944-1007
//   4. edge neighbours over welded indices, opposite winding    :1501-1594
//   5. groups = triangles around a welded vertex with equal
//      orientation, grown through neighbours                    :1069-1189
//   6. per group member: sub-group by angular threshold, tangent = normalised
//      angle-weighted sum of the derivatives projected into the plane of the
//      vertex normal                                            :1198-1439
//   7. set-aside triangles copy from a good triangle that shares the welded
//      corner                                                   :1820-1860
//
// Quads do not exist at this boundary (faces are triangles, scene.cu:560-567),
// so the quad split, quad orientation fix-up and quad epilogue are not needed.
// tests/test_oracle_and_host.py (test_tangents_match_mikktspace_*) and tests/test_reference_pins.py check this file against the vendored C library built
// from its own source (oracle/_ref) when the reference tree is present.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>

#include "host_internal.h"
#include "pthost.h"

namespace pth {

namespace {

struct F3 {
    float x, y, z;
};
inline F3 f3(pt_vec3 v) { return F3{v.x, v.y, v.z}; }
inline bool same(F3 a, F3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
inline F3 plus(F3 a, F3 b) { return F3{a.x + b.x, a.y + b.y, a.z + b.z}; }
inline F3 minus(F3 a, F3 b) { return F3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline F3 times(float s, F3 v) { return F3{s * v.x, s * v.y, s * v.z}; }
inline float dot3(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float length(F3 v) { return sqrtf(v.x * v.x + v.y * v.y + v.z * v.z); }
inline F3 unit(F3 v) { return times(1 / length(v), v); }
inline bool not_zero(float x) { return fabsf(x) > FLT_MIN; }
inline bool not_zero(F3 v) { return not_zero(v.x) || not_zero(v.y) || not_zero(v.z); }
inline F3 project_unit(F3 n, F3 v) {
    F3 r = minus(v, times(dot3(n, v), n));
    return not_zero(r) ? unit(r) : r;
}

enum : int { kDegenerate = 1, kGroupWithAny = 4, kOrientPreserving = 8 };

struct Tri {
    int corner[3];      // welded corner ids (corner id = 3*face + k of the representative)
    int org_face;       // index in the caller's face range
    int flag = 0;
    int neighbor[3] = {-1, -1, -1};
    int group[3] = {-1, -1, -1};
    F3 os{0, 0, 0}, ot{0, 0, 0};
    float mag_s = 0, mag_t = 0;
};

struct Group {
    int vertex;  // welded corner id all members share
    bool orient;
    std::vector<int> tris;
};

struct Space {
    F3 os{1, 0, 0};
    bool orient = false;
};

struct Mesh {
    const pt_vertex *verts;
    const pt_face *faces; // already offset to first_face
    F3 pos(int corner) const { return f3(verts[faces[corner / 3].vertices[corner % 3]].position); }
    F3 nrm(int corner) const { return f3(verts[faces[corner / 3].vertices[corner % 3]].normal); }
    void uv(int corner, float *u, float *v) const {
        const pt_vec3 &t = verts[faces[corner / 3].vertices[corner % 3]].texcoord0;
        *u = t.x;
        *v = t.y;
    }
};

struct WeldKey {
    uint32_t w[8];
    int corner;
};

inline uint32_t canon(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return u == 0x80000000u ? 0u : u; // -0 == +0 under the library's float compares
}

void weld(const Mesh &m, int ncorners, std::vector<int> &rep) {
    std::vector<WeldKey> keys((size_t)ncorners);
    for (int c = 0; c < ncorners; c++) {
        F3 p = m.pos(c), n = m.nrm(c);
        float u, v;
        m.uv(c, &u, &v);
        WeldKey &k = keys[(size_t)c];
        k.w[0] = canon(p.x); k.w[1] = canon(p.y); k.w[2] = canon(p.z);
        k.w[3] = canon(n.x); k.w[4] = canon(n.y); k.w[5] = canon(n.z);
        k.w[6] = canon(u);   k.w[7] = canon(v);
        k.corner = c;
    }
    std::sort(keys.begin(), keys.end(), [](const WeldKey &a, const WeldKey &b) {
        int c = memcmp(a.w, b.w, sizeof(a.w));
        return c != 0 ? c < 0 : a.corner < b.corner;
    });
    rep.assign((size_t)ncorners, 0);
    size_t i = 0;
    while (i < keys.size()) {
        size_t j = i;
        while (j < keys.size() && memcmp(keys[j].w, keys[i].w, sizeof(keys[i].w)) == 0) {
            rep[(size_t)keys[j].corner] = keys[i].corner; // lowest corner id of the class
            j++;
        }
        i = j;
    }
}

void derivatives(const Mesh &m, std::vector<Tri> &tris) {
    for (Tri &t : tris) {
        t.flag |= kGroupWithAny; // assumed bad until proven good
        F3 v1 = m.pos(t.corner[0]), v2 = m.pos(t.corner[1]), v3 = m.pos(t.corner[2]);
        float t1x, t1y, t2x, t2y, t3x, t3y;
        m.uv(t.corner[0], &t1x, &t1y);
        m.uv(t.corner[1], &t2x, &t2y);
        m.uv(t.corner[2], &t3x, &t3y);
        float t21x = t2x - t1x, t21y = t2y - t1y, t31x = t3x - t1x, t31y = t3y - t1y;
        F3 d1 = minus(v2, v1), d2 = minus(v3, v1);
        float area2 = t21x * t31y - t21y * t31x;
        F3 os = minus(times(t31y, d1), times(t21y, d2));
        F3 ot = plus(times(-t31x, d1), times(t21x, d2));
        if (area2 > 0) t.flag |= kOrientPreserving;
        if (not_zero(area2)) {
            float abs_area = fabsf(area2);
            float len_s = length(os), len_t = length(ot);
            float sign = (t.flag & kOrientPreserving) ? 1.0f : -1.0f;
            if (not_zero(len_s)) t.os = times(sign / len_s, os);
            if (not_zero(len_t)) t.ot = times(sign / len_t, ot);
            t.mag_s = len_s / abs_area;
            t.mag_t = len_t / abs_area;
            if (not_zero(t.mag_s) && not_zero(t.mag_t)) t.flag &= ~kGroupWithAny;
        }
    }
}

struct Edge {
    int lo, hi, tri;
};

// Which edge of a triangle carries the unordered pair {a,b}, and in which
// direction the triangle walks it (mikktspace.c:1703-1731).
inline void find_edge(const int c[3], int a, int b, int *from, int *to, int *edge) {
    if (c[0] == a || c[0] == b) {
        if (c[1] == a || c[1] == b) {
            *edge = 0; *from = c[0]; *to = c[1];
        } else {
            *edge = 2; *from = c[2]; *to = c[0];
        }
    } else {
        *edge = 1; *from = c[1]; *to = c[2];
    }
}

void neighbors(std::vector<Tri> &tris) {
    std::vector<Edge> edges;
    edges.reserve(tris.size() * 3);
    for (int f = 0; f < (int)tris.size(); f++)
        for (int i = 0; i < 3; i++) {
            int a = tris[(size_t)f].corner[i], b = tris[(size_t)f].corner[i < 2 ? i + 1 : 0];
            edges.push_back(Edge{a < b ? a : b, !(a < b) ? a : b, f});
        }
    std::sort(edges.begin(), edges.end(), [](const Edge &x, const Edge &y) {
        if (x.lo != y.lo) return x.lo < y.lo;
        if (x.hi != y.hi) return x.hi < y.hi;
        return x.tri < y.tri;
    });
    for (size_t i = 0; i < edges.size(); i++) {
        const Edge &e = edges[i];
        Tri &A = tris[(size_t)e.tri];
        int a_from, a_to, a_edge;
        find_edge(A.corner, e.lo, e.hi, &a_from, &a_to, &a_edge);
        if (A.neighbor[a_edge] != -1) continue;
        for (size_t j = i + 1; j < edges.size() && edges[j].lo == e.lo && edges[j].hi == e.hi; j++) {
            Tri &B = tris[(size_t)edges[j].tri];
            int b_from, b_to, b_edge;
            find_edge(B.corner, e.lo, e.hi, &b_from, &b_to, &b_edge);
            // neighbours walk the shared edge in opposite directions
            if (a_from == b_to && a_to == b_from && B.neighbor[b_edge] == -1) {
                A.neighbor[a_edge] = edges[j].tri;
                B.neighbor[b_edge] = e.tri;
                break;
            }
        }
    }
}

bool assign(std::vector<Tri> &tris, std::vector<Group> &groups, int tri_index, int g) {
    Tri &t = tris[(size_t)tri_index];
    Group &grp = groups[(size_t)g];
    int i = -1;
    if (t.corner[0] == grp.vertex) i = 0;
    else if (t.corner[1] == grp.vertex) i = 1;
    else if (t.corner[2] == grp.vertex) i = 2;
    if (i < 0) return false;
    if (t.group[i] == g) return true;
    if (t.group[i] != -1) return false;
    if (t.flag & kGroupWithAny) {
        // the first group to reach such a triangle fixes its orientation
        if (t.group[0] == -1 && t.group[1] == -1 && t.group[2] == -1) {
            t.flag &= ~kOrientPreserving;
            if (grp.orient) t.flag |= kOrientPreserving;
        }
    }
    if (((t.flag & kOrientPreserving) != 0) != grp.orient) return false;
    grp.tris.push_back(tri_index);
    t.group[i] = g;
    int left = t.neighbor[i], right = t.neighbor[i > 0 ? i - 1 : 2];
    if (left >= 0) assign(tris, groups, left, g);
    if (right >= 0) assign(tris, groups, right, g);
    return true;
}

void build_groups(std::vector<Tri> &tris, std::vector<Group> &groups) {
    for (int f = 0; f < (int)tris.size(); f++)
        for (int i = 0; i < 3; i++) {
            if ((tris[(size_t)f].flag & kGroupWithAny) || tris[(size_t)f].group[i] != -1) continue;
            int g = (int)groups.size();
            groups.push_back(Group{tris[(size_t)f].corner[i], (tris[(size_t)f].flag & kOrientPreserving) != 0, {}});
            tris[(size_t)f].group[i] = g;
            groups[(size_t)g].tris.push_back(f);
            int left = tris[(size_t)f].neighbor[i], right = tris[(size_t)f].neighbor[i > 0 ? i - 1 : 2];
            if (left >= 0) assign(tris, groups, left, g);
            if (right >= 0) assign(tris, groups, right, g);
        }
}

// mikktspace.c:1367-1439 — only the tangent (vOs) half is needed downstream.
F3 evaluate(const Mesh &m, const std::vector<Tri> &tris, const std::vector<int> &members, int vertex) {
    F3 sum{0, 0, 0};
    for (int f : members) {
        const Tri &t = tris[(size_t)f];
        if (t.flag & kGroupWithAny) continue;
        int i = t.corner[0] == vertex ? 0 : (t.corner[1] == vertex ? 1 : 2);
        F3 n = m.nrm(t.corner[i]);
        F3 os = project_unit(n, t.os);
        F3 p0 = m.pos(t.corner[i > 0 ? i - 1 : 2]);
        F3 p1 = m.pos(t.corner[i]);
        F3 p2 = m.pos(t.corner[i < 2 ? i + 1 : 0]);
        F3 e1 = project_unit(n, minus(p0, p1));
        F3 e2 = project_unit(n, minus(p2, p1));
        float c = dot3(e1, e2);
        c = c > 1 ? 1 : (c < -1 ? -1 : c);
        float angle = (float)acos((double)c);
        sum = plus(sum, times(angle, os));
    }
    return not_zero(sum) ? unit(sum) : sum;
}

} // namespace

} // namespace pth

using namespace pth;

extern "C" int pth_calculate_tangents(pt_vertex *vertices, uint32_t vertex_count, const pt_face *faces,
                                      uint32_t first_face, uint32_t num_faces) {
    if (!vertices || !faces) return set_error("pth_calculate_tangents: null argument");
    if (num_faces == 0) return set_error("pth_calculate_tangents: no faces"); // genTangSpace returns false
    const pt_face *range = faces + first_face;
    for (uint32_t f = 0; f < num_faces; f++) {
        if (range[f].vertexCount != 3) return set_error("pth_calculate_tangents: face %u is not a triangle", f);
        for (int k = 0; k < 3; k++)
            if (range[f].vertices[k] >= vertex_count)
                return set_error("pth_calculate_tangents: vertex index out of range in face %u", f);
    }
    Mesh m{vertices, range};
    int ncorners = (int)num_faces * 3;
    const float thres_cos = (float)cos((double)((180.0f * (float)M_PI) / 180.0f));

    std::vector<int> rep;
    weld(m, ncorners, rep);

    // good triangles keep their order; set-aside ones are handled at the end
    std::vector<Tri> good;
    std::vector<Tri> aside;
    good.reserve(num_faces);
    for (int f = 0; f < (int)num_faces; f++) {
        Tri t;
        t.org_face = f;
        for (int k = 0; k < 3; k++) t.corner[k] = rep[(size_t)(3 * f + k)];
        F3 p0 = m.pos(t.corner[0]), p1 = m.pos(t.corner[1]), p2 = m.pos(t.corner[2]);
        if (same(p0, p1) || same(p0, p2) || same(p1, p2)) {
            t.flag |= kDegenerate;
            aside.push_back(t);
        } else {
            good.push_back(t);
        }
    }

    derivatives(m, good);
    neighbors(good);
    std::vector<Group> groups;
    build_groups(good, groups);

    std::vector<Space> spaces((size_t)ncorners); // default (1,0,0), orient false

    std::vector<int> members;
    std::vector<std::vector<int>> unique_members;
    std::vector<F3> unique_space;
    for (size_t g = 0; g < groups.size(); g++) {
        const Group &grp = groups[g];
        unique_members.clear();
        unique_space.clear();
        for (int f : grp.tris) {
            const Tri &tf = good[(size_t)f];
            int index = tf.group[0] == (int)g ? 0 : (tf.group[1] == (int)g ? 1 : 2);
            F3 n = m.nrm(tf.corner[index]);
            F3 os = project_unit(n, tf.os), ot = project_unit(n, tf.ot);
            members.clear();
            for (int t : grp.tris) {
                const Tri &tt = good[(size_t)t];
                F3 os2 = project_unit(n, tt.os), ot2 = project_unit(n, tt.ot);
                bool any = ((tf.flag | tt.flag) & kGroupWithAny) != 0;
                bool same_face = tf.org_face == tt.org_face;
                float cos_s = dot3(os, os2), cos_t = dot3(ot, ot2);
                if (any || same_face || (cos_s > thres_cos && cos_t > thres_cos)) members.push_back(t);
            }
            std::sort(members.begin(), members.end());
            size_t l = 0;
            while (l < unique_members.size() && unique_members[l] != members) l++;
            if (l == unique_members.size()) {
                unique_members.push_back(members);
                unique_space.push_back(evaluate(m, good, members, grp.vertex));
            }
            Space &out = spaces[(size_t)(3 * tf.org_face + index)];
            out.os = unique_space[l];
            out.orient = grp.orient;
        }
    }

    // set-aside triangles: copy from the first good corner with the same welded id
    if (!aside.empty()) {
        std::vector<int> first_good((size_t)ncorners, -1); // welded id -> slot of its first good corner
        for (const Tri &t : good)
            for (int k = 0; k < 3; k++)
                if (first_good[(size_t)t.corner[k]] < 0) first_good[(size_t)t.corner[k]] = 3 * t.org_face + k;
        for (const Tri &t : aside)
            for (int i = 0; i < 3; i++) {
                int src = first_good[(size_t)t.corner[i]];
                if (src >= 0) spaces[(size_t)(3 * t.org_face + i)] = spaces[(size_t)src];
            }
    }

    // scene.cu:437-447 — write through the index list, w = -sign
    for (uint32_t f = 0; f < num_faces; f++)
        for (int i = 0; i < 3; i++) {
            const Space &s = spaces[(size_t)(3 * f + (uint32_t)i)];
            float sign = s.orient ? 1.0f : -1.0f;
            pt_vec4 &tan = vertices[range[f].vertices[i]].tangent;
            tan.x = s.os.x;
            tan.y = s.os.y;
            tan.z = s.os.z;
            tan.w = -sign;
        }
    return 0;
}
