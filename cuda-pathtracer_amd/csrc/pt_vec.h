// pt_vec.h — 3-float vector with the reference's rounding semantics, host + device.
//
// The reference's Vector<T,N> (include/core/vector.h) fixes the evaluation order
// of every expression on the path; the helpers below spell that order out so the
// same bits come out of the host code and the gfx950 kernels.  Everything here is
// compiled with -ffp-contract=off: a*b+c must stay two roundings.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

#define PT_HD __host__ __device__ __forceinline__

namespace ptmi {

struct f3 { float x, y, z; };

PT_HD f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
PT_HD f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }      // vector.h:148-153
PT_HD f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }      // vector.h:155-160
PT_HD f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }      // vector.h:162-167
PT_HD f3 operator*(float t, f3 v) { return mk3(t * v.x, t * v.y, t * v.z); }         // vector.h:177-187
PT_HD f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }                           // vector.h:56-60

// IEEE-correct 1/t.  The reference writes `T k = 1.0 / t` (vector.h:90-94,189-195):
// a binary64 quotient rounded to binary32, which equals the correctly rounded
// binary32 quotient (53 >= 2*24+2, double rounding is innocuous for division).
// On the device both rely on hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt (the build passes it
// explicitly): `/` and __builtin_sqrtf then lower to the IEEE-exact v_div_scale/fmas/fixup and scaled-sqrt
// sequences.  NOT __fsqrt_rn: without OCML_BASIC_ROUNDED_OPERATIONS that is the 1-ulp v_sqrt_f32.
PT_HD float rcp_rn(float t) { return 1.0f / t; }

// Correctly rounded 1/a for the Moller-Trumbore determinant, in 1 v_rcp_f32 + 4 v_fma_f32 instead of the compiler's
// 10-instruction IEEE division (v_div_scale x2, v_rcp, 4 fma, mul, v_div_fmas, v_div_fixup; the scale/fmas/fixup
// ops are half-rate).  Two Newton steps on the <= 1 ulp hardware seed; without the scaling steps this is exact only
// while neither a nor 1/a leaves the normal range.  tests/test_gpu_parity.py::test_fast_reciprocal_is_exact checks
// ALL 2^32 bit patterns against the IEEE quotient on the device and pins the interval [2^-100, 2^100] used here;
// outside it the caller's result is either rejected anyway (|a| < 1e-8) or unreachable for scenes that pass the
// host's extent check.  Device only: the CPU oracle keeps the true division.
__device__ __forceinline__ float rcp_exact_normal(float a) {
    float r = __builtin_amdgcn_rcpf(a);
    float e = __builtin_fmaf(-a, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    e = __builtin_fmaf(-a, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    return r;
}
PT_HD float sqrt_rn(float x) { return __builtin_sqrtf(x); }
PT_HD f3 div_scalar(f3 v, float t) { float k = rcp_rn(t); return mk3(v.x * k, v.y * k, v.z * k); }
PT_HD float dot(f3 a, f3 b) { float s = a.x * b.x; s += a.y * b.y; s += a.z * b.z; return s; }   // vector.h:198-203 (0 + x is exact)
PT_HD float length_squared(f3 a) { return dot(a, a); }                               // vector.h:97-101
PT_HD float length(f3 a) { return sqrt_rn(length_squared(a)); }                      // vector.h:103-105
PT_HD f3 unit_vector(f3 v) { return div_scalar(v, length(v)); }                      // vector.h:205-208
PT_HD f3 cross(f3 a, f3 b) {                                                          // vector.h:211-218
    return mk3(a.y * b.z - a.z * b.y, -(a.x * b.z - a.z * b.x), a.x * b.y - a.y * b.x);
}

}  // namespace ptmi
