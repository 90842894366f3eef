// pt_device.h — device helpers shared by the render kernels (kernels.hip) and the radiosity pre-pass (radiosity.hip).
// Compile with -ffp-contract=off (see include/ptmi_math.h, pt_vec.h).
#pragma once
#include "device_scene.h"
#include "wide_bvh.h"

#include <float.h>

#include "../../include/ptmi_math.h"

namespace ptmi {

// ---------------------------------------------------------------------------------------------
// RNG: cuRAND XORWOW restated (third-party algorithm; see oracle/ptmi_oracle.c header for status)
// ---------------------------------------------------------------------------------------------
struct Rng { uint32_t v0, v1, v2, v3, v4, d; };

__device__ __forceinline__ uint32_t rng_next(Rng& r) {
    const uint32_t t = r.v0 ^ (r.v0 >> 2);
    r.v0 = r.v1; r.v1 = r.v2; r.v2 = r.v3; r.v3 = r.v4;
    r.v4 = (r.v4 ^ (r.v4 << 4)) ^ (t ^ (t << 1));
    r.d += 362437u;
    return r.v4 + r.d;
}
// curand_uniform: x * 2^-32 + 2^-33, in (0, 1]
__device__ __forceinline__ float rng_uniform(Rng& r) {
    const uint32_t x = rng_next(r);
    return (float)x * 2.3283064e-10f + (2.3283064e-10f / 2.0f);
}

// ---------------------------------------------------------------------------------------------
// primitive tests
// ---------------------------------------------------------------------------------------------
// fminf/fmaxf without the v_max_f32 x, x, x "canonicalize" the compiler puts in front of every min/max whose operand
// it cannot prove free of signalling NaNs (loop-carried closest_t, values loaded from LDS).  v_min/v_max/v_min3 return
// the other operand for a quiet NaN, exactly like fminf/fmaxf; they differ only for signalling NaNs, which no
// arithmetic instruction ever produces and the scene loader never stores.  Each avoided canonicalize is a half-rate op.
__device__ __forceinline__ float min_raw(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float max_raw(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float min3_raw(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float max3_raw(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

// Branch-free Moller-Trumbore accept test (edge1/edge2 arrive precomputed), used by all three walks.  The arithmetic
// is exactly triangle.h:64-96 / quad.h:56-87; the chain of early-outs becomes ONE sign test on a running minimum:
//   |a| <  eps  reject   <=>  |a| - eps      < 0      (IEEE subtraction never flips a sign; denormals are on)
//   u   <  0    reject   <=>  u              < 0
//   u   >  1    reject   <=>  1 - u          < 0
//   v   <  0    reject   <=>  v              < 0
//   u+v >  1    reject   <=>  1 - (u + v)    < 0
//   t > eps && t >= t_min  <=>  t - t_lo >= 0  with t_lo = max(t_min, nextafter(eps, +inf))
// fminf ignores NaN operands exactly where the reference's `x < 0 || x > 1` forms let a NaN through; the final
// `t < closest_t` (scene.h:90) rejects a NaN t.  Quad halves use the inclusive forms `|a| > eps`, `u >= 0 && ...`,
// which differ from the above only for |a| == eps (handled through eps_lo = nextafter(eps)) and for NaN u/v; NaNs
// need overflowing intermediates, which the host rules out (scene extent check) before choosing this code path.
// min/max/cmp/cndmask are half-rate on gfx950, add/sub/mul full-rate: 3 min + 5 sub replace 7 cmp + 7 cndmask.
__device__ __forceinline__ float mt_candidate(f3 v0, f3 edge1, f3 edge2, f3 o, f3 d, float eps_for_a, float t_lo) {
    const f3 h = cross(d, edge2);
    const float a = dot(edge1, h);
    const float f = rcp_exact_normal(a);                  // garbage for |a| < 2^-126, which the eps test rejects anyway
    const f3 s = o - v0;
    const float u = f * dot(s, h);
    // wave-level early-out: if the (a, u) tests already reject every lane that is testing this primitive, the second
    // half of Moller-Trumbore is skipped for the whole wave (coherent camera-ray waves do this for most primitives)
    const float m1 = min3_raw(fabsf(a) - eps_for_a, u, 1.0f - u);
    if (!__any(m1 >= 0.0f)) return __builtin_inff();
    const f3 q = cross(s, edge1);
    const float v = f * dot(d, q);
    const float t = f * dot(edge2, q);
    float m = min3_raw(m1, v, 1.0f - (u + v));
    m = min_raw(m, t - t_lo);
    return (m >= 0.0f) ? t : __builtin_inff();
}
// Triangle form: accept flag and t separately, so that the caller needs no +inf select (one half-rate op less)
__device__ __forceinline__ bool mt_accept(f3 v0, f3 edge1, f3 edge2, f3 o, f3 d, float eps_for_a, float t_lo, float closest_t, float& t) {
    const f3 h = cross(d, edge2);
    const float a = dot(edge1, h);
    const float f = rcp_exact_normal(a);
    const f3 s = o - v0;
    const float u = f * dot(s, h);
    const float m1 = min3_raw(fabsf(a) - eps_for_a, u, 1.0f - u);
    if (!__any(m1 >= 0.0f)) return false;
    const f3 q = cross(s, edge1);
    const float v = f * dot(d, q);
    t = f * dot(edge2, q);
    float m = min3_raw(m1, v, 1.0f - (u + v));
    m = min_raw(m, t - t_lo);
    return (m >= 0.0f) & (t < closest_t);          // t <= t_max && scene.h:90's strict t < closest_t; false for a NaN t
}
// The same test without the comparison against closest_t: the fast tree's walk (csrc/wide_bvh.h) needs to see t == closest_t
__device__ __forceinline__ bool mt_hit(f3 v0, f3 edge1, f3 edge2, f3 o, f3 d, float eps_for_a, float t_lo, float& t) {
    const f3 h = cross(d, edge2);
    const float a = dot(edge1, h);
    const float f = rcp_exact_normal(a);
    const f3 s = o - v0;
    const float u = f * dot(s, h);
    const float m1 = min3_raw(fabsf(a) - eps_for_a, u, 1.0f - u);
    if (!__any(m1 >= 0.0f)) return false;
    const f3 q = cross(s, edge1);
    const float v = f * dot(d, q);
    t = f * dot(edge2, q);
    float m = min3_raw(m1, v, 1.0f - (u + v));
    m = min_raw(m, t - t_lo);
    return m >= 0.0f;
}
// t_lo for a given t_min:  t > 1e-8f && t >= t_min  <=>  t >= t_lo
__device__ __forceinline__ float mt_t_lo(float t_min) {
    const float eps_up = __uint_as_float(__float_as_uint(1e-8f) + 1u);
    return t_min > 1e-8f ? t_min : eps_up;
}

__device__ __forceinline__ f3 xyz(const float4& v) { return mk3(v.x, v.y, v.z); }

// Frisvad frame (grid.h:287-297, form_factors.h:93-103, integrator.h:72-82: the same code three times)
__device__ __forceinline__ void build_frame(f3 n, f3& t, f3& b) {                 // grid.h:287-297
    if (n.z < -0.9999999f) { t = mk3(0.0f, -1.0f, 0.0f); b = mk3(-1.0f, 0.0f, 0.0f); return; }
    const float a = rcp_rn(1.0f + n.z);
    const float c = -n.x * n.y * a;
    t = mk3(1.0f - n.x * n.x * a, c, -n.x);
    b = mk3(c, 1.0f - n.y * n.y * a, -n.y);
}

}  // namespace ptmi
