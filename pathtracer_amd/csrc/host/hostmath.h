// hostmath.h — float vector helpers for the host pipeline (loader, BVH build,
// tonemap). The operation ORDER of each helper follows the reference's
// src/mathops.h (cited per function) because the arrays these produce are the
// kernel's inputs: a different association would move vertex positions, normals
// and BVH planes by an ulp. Build with -ffp-contract=off.
#pragma once

#include <cmath>

#include "pt_types.h"

namespace pth {

inline pt_vec3 v3(float x, float y, float z) { return pt_vec3{x, y, z}; }
inline pt_vec3 add(pt_vec3 a, pt_vec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline pt_vec3 sub(pt_vec3 a, pt_vec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline pt_vec3 scale(pt_vec3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }

// mathops.h:119-121 — x*x' + (y*y' + z*z')
inline float dot(pt_vec3 a, pt_vec3 b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }

// mathops.h:123-127
inline pt_vec3 cross(pt_vec3 a, pt_vec3 b) {
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}

inline float magnitude(pt_vec3 a) { return sqrtf(dot(a, a)); }

// mathops.h:114-117 — division by a scalar is a multiply by (float)(1.0 / s)
inline float recip(float s) { return (float)(1.0 / (double)s); }
inline pt_vec3 div(pt_vec3 a, float s) { return scale(a, recip(s)); }

// mathops.h:133-136
inline pt_vec3 normalized(pt_vec3 a) { return scale(a, recip(magnitude(a))); }

inline float comp(pt_vec3 a, int k) { return k == 0 ? a.x : (k == 1 ? a.y : a.z); }

// mathops.h:180-186 (strict compares: ties keep b)
inline pt_vec3 vmin(pt_vec3 a, pt_vec3 b) {
    return v3(a.x < b.x ? a.x : b.x, a.y < b.y ? a.y : b.y, a.z < b.z ? a.z : b.z);
}
inline pt_vec3 vmax(pt_vec3 a, pt_vec3 b) {
    return v3(a.x > b.x ? a.x : b.x, a.y > b.y ? a.y : b.y, a.z > b.z ? a.z : b.z);
}
// mathops.h:192-194
inline float max_component(pt_vec3 a) { return (a.x > a.y && a.x > a.z) ? a.x : (a.y > a.z ? a.y : a.z); }

struct Box {
    pt_vec3 lo{1e30f, 1e30f, 1e30f};   // mathops.h:589-590
    pt_vec3 hi{-1e30f, -1e30f, -1e30f};
    void grow(pt_vec3 p) {
        lo = vmin(lo, p);
        hi = vmax(hi, p);
    }
    void grow(const Box &o) {
        lo = vmin(lo, o.lo);
        hi = vmax(hi, o.hi);
    }
    // mathops.h:607-610 — half surface area
    float area() const {
        pt_vec3 e = sub(hi, lo);
        return e.x * e.y + e.y * e.z + e.z * e.x;
    }
};

// Row-major 4x4, product and mat*vec in the reference's left-to-right order
// (mathops.h:552-577).
struct M4 {
    float m[4][4];
    static M4 identity() {
        M4 r{};
        for (int i = 0; i < 4; i++) r.m[i][i] = 1.0f;
        return r;
    }
};

inline M4 mul(const M4 &a, const M4 &b) {
    M4 r{};
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            r.m[i][j] = ((a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j]) + a.m[i][2] * b.m[2][j]) +
                        a.m[i][3] * b.m[3][j];
    return r;
}

inline pt_vec4 mul(const M4 &a, pt_vec4 v) {
    pt_vec4 r;
    r.x = ((a.m[0][0] * v.x + a.m[0][1] * v.y) + a.m[0][2] * v.z) + a.m[0][3] * v.w;
    r.y = ((a.m[1][0] * v.x + a.m[1][1] * v.y) + a.m[1][2] * v.z) + a.m[1][3] * v.w;
    r.z = ((a.m[2][0] * v.x + a.m[2][1] * v.y) + a.m[2][2] * v.z) + a.m[2][3] * v.w;
    r.w = ((a.m[3][0] * v.x + a.m[3][1] * v.y) + a.m[3][2] * v.z) + a.m[3][3] * v.w;
    return r;
}

// mathops.h:491-494
inline pt_vec3 dehomogenise(pt_vec4 v) { return div(v3(v.x, v.y, v.z), v.w); }

// linear part times a vector (mathops.h:264-268, 579-585)
inline pt_vec3 mul_linear(const M4 &a, pt_vec3 v) {
    return v3((a.m[0][0] * v.x + a.m[0][1] * v.y) + a.m[0][2] * v.z,
              (a.m[1][0] * v.x + a.m[1][1] * v.y) + a.m[1][2] * v.z,
              (a.m[2][0] * v.x + a.m[2][1] * v.y) + a.m[2][2] * v.z);
}

} // namespace pth
