/*
 * vd_oracle_math.h — strict-fp32 helpers shared by the oracle translation units.
 * TEST INFRASTRUCTURE ONLY (see vd_oracle.h).  Must be compiled with
 * -ffp-contract=off -fno-fast-math: rustc never contracts a*b+c into an FMA and the oracle
 * fixes the same for WGSL (SURVEY.md §8a C2').
 */
#ifndef VD_ORACLE_MATH_H
#define VD_ORACLE_MATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

typedef struct { float x, y, z; } v3;

static inline v3 v3_make(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 v3_add(v3 a, v3 b) { return v3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3_sub(v3 a, v3 b) { return v3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3_mul(v3 a, v3 b) { return v3_make(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 v3_scale(v3 a, float s) { return v3_make(a.x * s, a.y * s, a.z * s); }
/* WGSL dot(a,b) / glam Vec3::dot: (x*x + y*y) + z*z  — spec decision C2'. */
static inline float v3_dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
/* WGSL cross / glam Vec3::cross: (a.y*b.z - b.y*a.z, a.z*b.x - b.z*a.x, a.x*b.y - b.x*a.y) */
static inline v3 v3_cross(v3 a, v3 b) {
    return v3_make(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
/* WGSL length(v) = sqrt(dot(v,v)), correctly-rounded sqrt — spec decision C2'. */
static inline float v3_length(v3 a) { return sqrtf(v3_dot(a, a)); }

static inline uint32_t f32_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float bits_f32(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* Rust's f32::min / f32::max (glam 0.24 scalar Vec3::min/max call them component-wise:
 * crates/bvh/src/blas.rs:190-198, tlas.rs:43,69-70,96-97): "if one of the arguments is NaN, then
 * the other argument is returned" - a NaN operand is IGNORED, never propagated (both NaN -> NaN).
 * Rust leaves the sign of min(-0,+0) unspecified; spec decision: -0 < +0, so
 * that bounds are order-independent and bit-reproducible (total-order key on the non-NaN values;
 * gfx950's v_min_f32 / v_max_f32 have exactly these semantics). */
static inline int32_t f32_key(float f) {
    int32_t i = (int32_t)f32_bits(f);
    return i ^ ((i >> 31) & 0x7fffffff);
}
static inline float f32_min_to(float a, float b) {
    if (a != a) return b;
    if (b != b) return a;
    return f32_key(b) < f32_key(a) ? b : a;
}
static inline float f32_max_to(float a, float b) {
    if (a != a) return b;
    if (b != b) return a;
    return f32_key(b) > f32_key(a) ? b : a;
}
static inline v3 v3_min_to(v3 a, v3 b) {
    return v3_make(f32_min_to(a.x, b.x), f32_min_to(a.y, b.y), f32_min_to(a.z, b.z));
}
static inline v3 v3_max_to(v3 a, v3 b) {
    return v3_make(f32_max_to(a.x, b.x), f32_max_to(a.y, b.y), f32_max_to(a.z, b.z));
}

/* crates/bvh/src/intersection.rs:16-19 — Aabb::area */
static inline float aabb_area(v3 mn, v3 mx) {
    v3 d = v3_sub(mx, mn);
    return (d.x * d.y + d.x * d.z + d.y * d.z) * 2.0f;
}

static inline v3 v3_load(const float* p) { return v3_make(p[0], p[1], p[2]); }
static inline void v3_store(float* p, v3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }

#define VD_REF_MAX_DIST 1e30f

#endif
