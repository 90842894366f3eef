// wide_bvh.h — record layout and node test of the opt-in FAST tree (ptmi_config.fast_tree), host + device.
//
// SURVEY 7 (last bullet) allows "an SAH/wide tree for speed runs (results identical except exact-tie cases, which must be
// reported)".  The exact walk follows the reference's midpoint-split binary tree (rendering/bvh.h:156-218): 72.5 dependent
// node fetches per ray on the 1 M-triangle scene, and a ray is a chain of dependent fetches.  The fast tree keeps the
// triangles, their Moller-Trumbore arithmetic (pt_device.h: mt_accept, bit for bit) and everything above the hit query;
// only WHICH boxes are tested on the way changes: a binned-SAH tree collapsed 8-wide, so that one 128-byte fetch - one L2
// line - decides eight children at once.  Boxes are stored conservatively (8-bit grid of the node's own box, rounded
// outwards, after a relative pad), so the walk can only test MORE triangles than needed, never fewer; a ray's closest hit is
// the same triangle and the same t as the reference's unless (a) two triangles are hit at exactly the same t (the reference
// keeps the one it visits first = the smaller leaf-order slot, scene.h:89-90; the fast walk keeps the smaller slot among the
// tied triangles it tests) or (b) a slab test of the reference's own walk rejects, by rounding, the box of the triangle that
// is hit.  tests/test_fast_tree.py reports how often.
//
// NODE, 32 dwords = 128 bytes:
//   0..2   p = min corner of the node's box (float)
//   3      ex | ey << 8 | ez << 16 | imask << 24   e*: biased exponent of the grid step 2^(e-127) per axis;
//                                                  imask: bit s set = slot s holds an INNER child
//   4      child_base: node index of the first inner child; inner children are consecutive in slot order
//   5      tri_base:   fast-order index of the node's first triangle; the triangles of its leaf children are consecutive
//   6, 7   unused
//   8..19  quantised planes, one byte per slot, 2 dwords per plane: lo.x lo.y lo.z hi.x hi.y hi.z
//          child box = [p + lo * 2^e, p + hi * 2^e]; an empty slot has lo = 255, hi = 0
//   20..27 per slot: what a hit of that child adds to the step's result word - leaf: its triangles as a run of 1-3 bits at
//          (first triangle - tri_base), inner: bit 24 + slot, empty: 0
//   28..31 unused
// Nodes are numbered breadth first: every level is contiguous, the nodes a workgroup keeps in LDS are "index < n_top".
// Slots are assigned so that visiting the hit children by descending (slot ^ octinv) is roughly front to back for every
// ray octant (octinv: bit a set when the ray moves towards +a): slot s sits towards +a of its parent's centre when bit a of
// s is set (the published idea of Ylitie, Karras, Laine 2017; this is an independent implementation with its own layout).
#pragma once
#include <stdint.h>
#include <string.h>

#include "pt_vec.h"

namespace ptmi {

constexpr int kWideNodeDwords = 32;
constexpr int kWideCertStride = 4;                // float4 per triangle in DeviceScene::wcert (proof record + shading record: one 64-byte line)
constexpr int WN_PLANES0 = 8, WN_WORD0 = 20;      // first dword of the quantised planes / of the per-slot result words
constexpr int kWideMaxLeaf = 3;                   // triangles per leaf child (3 bits of a 24-bit triangle word per node)
constexpr float kWideInvLimit = 1.2089258e24f;    // 2^80: |1 / d| is clamped here, so no plane distance is ever inf or NaN

PT_HD float wb_as_float(uint32_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(u);
#else
    float f; memcpy(&f, &u, 4); return f;
#endif
}
PT_HD float wb_max3(float a, float b, float c) {
#if defined(__HIP_DEVICE_COMPILE__)
    float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r;
#else
    return fmaxf(fmaxf(a, b), c);
#endif
}
PT_HD float wb_min3(float a, float b, float c) {
#if defined(__HIP_DEVICE_COMPILE__)
    float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r;
#else
    return fminf(fminf(a, b), c);
#endif
}
// fmaxf / fminf of two operands that are never NaN here (clamped slopes): on the device without the canonicalising
// v_max_f32 x, x, x the compiler puts in front of fmaxf / fminf
PT_HD float wb_max(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
    float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r;
#else
    return fmaxf(a, b);
#endif
}
PT_HD float wb_min(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
    float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r;
#else
    return fminf(a, b);
#endif
}
PT_HD float wb_byte(uint32_t w, int i) { return (float)((w >> (8 * i)) & 0xffu); }      // v_cvt_f32_ubyte<i>

// 1 / d, finite: a zero (or denormal-small) direction component becomes a huge finite slope of the same sign
PT_HD float wide_inv(float d) {
    const float r = 1.0f / d;
    return fabsf(r) <= kWideInvLimit ? r : copysignf(kWideInvLimit, d);
}
// bit a set when the ray moves towards +a
PT_HD uint32_t wide_octinv(f3 inv) { return (inv.x < 0.0f ? 0u : 1u) | (inv.y < 0.0f ? 0u : 2u) | (inv.z < 0.0f ? 0u : 4u); }

// bit s of h -> bit (s ^ c), for an 8-bit mask
PT_HD uint32_t wide_permute(uint32_t h, uint32_t c) {
    if (c & 1u) h = ((h & 0x55u) << 1) | ((h >> 1) & 0x55u);
    if (c & 2u) h = ((h & 0x33u) << 2) | ((h >> 2) & 0x33u);
    if (c & 4u) h = ((h & 0x0fu) << 4) | ((h >> 4) & 0x0fu);
    return h;
}

struct WideStep {
    uint32_t child_base, imask;
    uint32_t inner;          // hit inner children, bit (slot ^ octinv): the highest bit is the child to enter first
    uint32_t tri_base, tris; // hit leaf children's triangles, bit k = triangle tri_base + k
};

// The eight slab tests of one node against [t_min, closest].  q0..q6 = dwords 0..27 of the node.
PT_HD WideStep wide_node_test(uint4 q0, uint4 q1, uint4 q2, uint4 q3, uint4 q4, uint4 q5, uint4 q6, f3 o, f3 inv, uint32_t octinv,
                              float t_min, float closest) {
    const float sx = wb_as_float((q0.w & 0xffu) << 23) * inv.x, sy = wb_as_float(((q0.w >> 8) & 0xffu) << 23) * inv.y,
                sz = wb_as_float(((q0.w >> 16) & 0xffu) << 23) * inv.z;
    const float bx = (wb_as_float(q0.x) - o.x) * inv.x, by = (wb_as_float(q0.y) - o.y) * inv.y, bz = (wb_as_float(q0.z) - o.z) * inv.z;
    const bool nx = inv.x < 0.0f, ny = inv.y < 0.0f, nz = inv.z < 0.0f;
    // entry / exit plane of every slot per axis: lo / hi swapped for a negative direction
    const uint32_t ex_[2] = {nx ? q3.z : q2.x, nx ? q3.w : q2.y}, fx_[2] = {nx ? q2.x : q3.z, nx ? q2.y : q3.w};
    const uint32_t ey_[2] = {ny ? q4.x : q2.z, ny ? q4.y : q2.w}, fy_[2] = {ny ? q2.z : q4.x, ny ? q2.w : q4.y};
    const uint32_t ez_[2] = {nz ? q4.z : q3.x, nz ? q4.w : q3.y}, fz_[2] = {nz ? q3.x : q4.z, nz ? q3.y : q4.w};
    const uint32_t w[8] = {q5.x, q5.y, q5.z, q5.w, q6.x, q6.y, q6.z, q6.w};
    uint32_t hw = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int j = i >> 2, b = i & 3;
        const float tx0 = __builtin_fmaf(wb_byte(ex_[j], b), sx, bx), tx1 = __builtin_fmaf(wb_byte(fx_[j], b), sx, bx);
        const float ty0 = __builtin_fmaf(wb_byte(ey_[j], b), sy, by), ty1 = __builtin_fmaf(wb_byte(fy_[j], b), sy, by);
        const float tz0 = __builtin_fmaf(wb_byte(ez_[j], b), sz, bz), tz1 = __builtin_fmaf(wb_byte(fz_[j], b), sz, bz);
        const float tn = wb_max3(tx0, ty0, wb_max(tz0, t_min));
        const float tf = wb_min3(tx1, ty1, wb_min(tz1, closest));
        hw |= tn <= tf ? w[i] : 0u;
    }
    WideStep s;
    s.child_base = q1.x; s.tri_base = q1.y; s.imask = q0.w >> 24;
    s.tris = hw & 0xffffffu;
    s.inner = wide_permute(hw >> 24, octinv);
    return s;
}

}  // namespace ptmi
