// kernels.hip — gfx950 kernels of the render path.
//
// Compile with -ffp-contract=off: results must be bit-identical to the strict-IEEE
// evaluation of the reference's expressions (see include/ptmi_math.h, pt_vec.h).
//
// Kernels
//   ptmi_render_init   render_init (integrator.h:274-280): XORWOW seeding + 2^67*pixel skip-ahead
//   ptmi_frame_begin   sample 0's camera ray for every pixel (integrator.h:383-387)
//   ptmi_bounce        THE hot kernel (scenes up to 64 primitives, and deep-tree fallback): intersect (scene.h:50-110,
//                      triangle.h:64-96, quad.h:49-132) + integrator() body (integrator.h:189-268) + regeneration +
//                      queue compaction; segment-synchronous, wave-uniform SWEEP walk (or STACK / LANE)
//   ptmi_bounce_phased the same work for larger scenes: per-lane stackless walk with wave-scheduled NODE/PRIM/SHADE phases
//   ptmi_resolve       integrator.h:393-407
#include "pt_device.h"

namespace ptmi {

// slot -> (x, local row).  With tile8 a wave's 64 consecutive slots are an 8x8 pixel tile instead of a 64x1 strip:
// its camera rays span a smaller solid angle and its bounce rays start closer together, so the wave-synchronous
// sweep visits a smaller union of nodes and primitives.  Pure scheduling: results are keyed by the pixel.
__device__ __forceinline__ void slot_to_local(const TileMap& tm, int slot, int& x, int& lr) {
    if (tm.tile8) {
        const int tile = slot >> 6, in = slot & 63;
        const int tiles_per_row = tm.width >> 3;
        const int ty = tile / tiles_per_row, tx = tile - ty * tiles_per_row;
        lr = (ty << 3) + (in >> 3);
        x = (tx << 3) + (in & 7);
    } else {
        lr = slot / tm.width;
        x = slot - lr * tm.width;
    }
}
__device__ __forceinline__ int global_pixel(const TileMap& tm, int slot, int& x, int& y) {
    int lr;
    slot_to_local(tm, slot, x, lr);
    y = ((lr / tm.row_block) * tm.n_ranks + tm.rank) * tm.row_block + (lr % tm.row_block);
    return y * tm.width + x;
}

// One 160x160 GF(2) matrix at a time is staged in LDS (3200 B); rows are read as wave-wide broadcasts.
__global__ __launch_bounds__(kBlock) void ptmi_render_init(TileMap tm, PathState st, const uint32_t* __restrict__ jump,
                                                           unsigned long long seed_base) {
    __shared__ uint32_t M[160 * 5];
    const int n = tm.local_rows * tm.width;
    const int slot = blockIdx.x * kBlock + threadIdx.x;
    const bool live = slot < n;
    int x = 0, y = 0;
    const unsigned int pix = live ? (unsigned int)global_pixel(tm, slot, x, y) : 0u;
    const unsigned long long seed = seed_base + (unsigned long long)pix;
    const uint32_t s0 = ((uint32_t)seed) ^ 0xaad26b49u;
    const uint32_t s1 = (uint32_t)(seed >> 32) ^ 0xf7dcefddu;
    const uint32_t t0 = 1099087573u * s0;
    const uint32_t t1 = 2591861531u * s1;
    uint32_t v[5] = {123456789u + t0, 362436069u ^ t0, 521288629u + t1, 88675123u ^ t1, 5783321u + t0};
    const uint32_t d = 6615241u + t1 + t0;

    for (int k = 0; k < 32; k++) {
        const bool mine = live && ((pix >> k) & 1u);
        if (!__syncthreads_or(mine ? 1 : 0)) continue;     // block-uniform: nobody needs T^(2^67 * 2^k)
        for (int i = threadIdx.x; i < 160 * 5; i += kBlock) M[i] = jump[k * 160 * 5 + i];
        __syncthreads();
        if (mine) {
            uint32_t r[5] = {0u, 0u, 0u, 0u, 0u};
#pragma unroll
            for (int w = 0; w < 5; w++) {
                const uint32_t word = v[w];
                for (int b = 0; b < 32; b++) {
                    const uint32_t m = 0u - ((word >> b) & 1u);
                    const uint32_t* row = &M[(w * 32 + b) * 5];
                    r[0] ^= row[0] & m; r[1] ^= row[1] & m; r[2] ^= row[2] & m; r[3] ^= row[3] & m; r[4] ^= row[4] & m;
                }
            }
#pragma unroll
            for (int w = 0; w < 5; w++) v[w] = r[w];
        }
        __syncthreads();
    }
    if (!live) return;
    st.E[slot] = make_uint4(v[0], v[1], v[2], v[3]);
    st.F[slot] = make_uint2(v[4], d);
    st.A[slot] = make_float4(0, 0, 0, 1.0f);
    st.B[slot] = make_float4(0, 0, 1.0f, 1.0f);
    st.C[slot] = make_float4(0, 0, 0, 1.0f);
    st.D[slot] = make_float4(0, 0, 0, __uint_as_float(0u));
}

// ---------------------------------------------------------------------------------------------
// camera (sensor.h:31-33 + ray.h:9-12) and the per-sample jitter (integrator.h:384-385)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void camera_ray(const FrameParams& fp, const TileMap& tm, int x, int y, Rng& rng, f3& o, f3& d) {
    const float u = ((float)x + rng_uniform(rng)) / (float)tm.width;
    const float v = ((float)y + rng_uniform(rng)) / (float)tm.height;
    const f3 org = mk3(fp.cam_origin[0], fp.cam_origin[1], fp.cam_origin[2]);
    const f3 llc = mk3(fp.cam_llc[0], fp.cam_llc[1], fp.cam_llc[2]);
    const f3 hor = mk3(fp.cam_hor[0], fp.cam_hor[1], fp.cam_hor[2]);
    const f3 ver = mk3(fp.cam_ver[0], fp.cam_ver[1], fp.cam_ver[2]);
    o = org;
    d = unit_vector(llc + u * hor + v * ver - org);
}

__global__ __launch_bounds__(kBlock) void ptmi_frame_begin(TileMap tm, PathState st, FrameParams fp) {
    const int n = tm.local_rows * tm.width;
    const int slot = blockIdx.x * kBlock + threadIdx.x;
    if (slot >= n) return;
    int x, y;
    global_pixel(tm, slot, x, y);
    const uint4 e = st.E[slot]; const uint2 f = st.F[slot];
    Rng rng = {e.x, e.y, e.z, e.w, f.x, f.y};
    f3 o, d;
    camera_ray(fp, tm, x, y, rng, o, d);
    st.A[slot] = make_float4(o.x, o.y, o.z, 1.0f);
    st.B[slot] = make_float4(d.x, d.y, d.z, 1.0f);
    st.C[slot] = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
    st.D[slot] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(0u));
    st.E[slot] = make_uint4(rng.v0, rng.v1, rng.v2, rng.v3);
    st.F[slot] = make_uint2(rng.v4, rng.d);
}

struct LaneCounters { unsigned int rays, node_visits, prim_tests, hits, top_visits, cert_chain, cert_fallback; };

// Slab test of scene.h:66-81 against [t_min, closest_t]; returns false when the reference would `continue`.
// `t0 > tmin_box ? t0 : tmin_box` is written fmaxf(t0, tmin_box): identical for every input because tmin_box /
// tmax_box are never NaN (a NaN t0/t1 - 0 * inf - is ignored by both forms) and the sign of a zero cannot reach the
// final comparison.  One v_max/v_min instead of v_cmp + v_cndmask (all of them half-rate VALU ops on gfx950).
__device__ __forceinline__ bool box_hit(const float4& n0, const float4& n1, f3 o, f3 inv, float t_min, float closest_t) {
    float t0x = (n0.x - o.x) * inv.x, t1x = (n1.x - o.x) * inv.x;
    if (inv.x < 0.0f) { const float tmp = t0x; t0x = t1x; t1x = tmp; }
    float t0y = (n0.y - o.y) * inv.y, t1y = (n1.y - o.y) * inv.y;
    if (inv.y < 0.0f) { const float tmp = t0y; t0y = t1y; t1y = tmp; }
    float t0z = (n0.z - o.z) * inv.z, t1z = (n1.z - o.z) * inv.z;
    if (inv.z < 0.0f) { const float tmp = t0z; t0z = t1z; t1z = tmp; }
    // max/min are associative and NaN-ignoring, so folding the three axes in one max3/min3 keeps the reference's result
    const float tmin_box = max3_raw(max_raw(t0x, t_min), t0y, t0z);
    const float tmax_box = min3_raw(min_raw(t1x, closest_t), t1y, t1z);
    return !(tmax_box < tmin_box);
}

// Primitive::intersect (primitive.h:83-90) + the closer-hit update of scene.h:89-96 for leaf slot k (per-lane k).
template <bool HAS_QUADS>
__device__ __forceinline__ void leaf_prim(const float4* __restrict__ prims, int prim_stride, int k, f3 o, f3 d, float t_lo,
                                          float& closest_t, int& slot_hit) {
    const float4 p0 = prims[k * prim_stride], p1 = prims[k * prim_stride + 1], p2 = prims[k * prim_stride + 2];
    const float eps = 1e-8f, eps_up = __uint_as_float(__float_as_uint(1e-8f) + 1u);
    float t;
    if (HAS_QUADS && __float_as_int(p0.w) != 0) {
        const float4 p3 = prims[k * prim_stride + 3];
        const float t1 = mt_candidate(xyz(p0), xyz(p1), xyz(p2), o, d, eps_up, t_lo);     // (v00, v10, v11)
        const float c1 = min_raw(t1, closest_t);
        const float t2 = mt_candidate(xyz(p0), xyz(p2), xyz(p3), o, d, eps_up, t_lo);     // (v00, v11, v01)
        t = min_raw(t2, c1);
    } else {
        float tt = 0.0f;
        const bool acc = mt_accept(xyz(p0), xyz(p1), xyz(p2), o, d, eps, t_lo, closest_t, tt);
        closest_t = acc ? tt : closest_t;
        slot_hit = acc ? k : slot_hit;
        return;
    }
    const bool closer = t < closest_t;
    closest_t = min_raw(t, closest_t);
    slot_hit = closer ? k : slot_hit;
}

// The same for the 36-byte triangle records of the packed layout (v0, e1, e2: three 12-byte loads)
struct f3p { float x, y, z; };
__device__ __forceinline__ void leaf_prim_packed(const float* __restrict__ gprims, int k, f3 o, f3 d, float t_lo, float& closest_t, int& slot_hit) {
    const f3p* r = reinterpret_cast<const f3p*>(gprims) + 3 * (size_t)k;
    const f3p v0 = r[0], e1 = r[1], e2 = r[2];
    float tt = 0.0f;
    const bool acc = mt_accept(mk3(v0.x, v0.y, v0.z), mk3(e1.x, e1.y, e1.z), mk3(e2.x, e2.y, e2.z), o, d, 1e-8f, t_lo, closest_t, tt);
    closest_t = acc ? tt : closest_t;
    slot_hit = acc ? k : slot_hit;
}

// ---- TRAVERSAL_STACK: Scene::intersect_bvh_optimized (scene.h:50-110) with its explicit stack ------------------
// `stack` points at this lane's column of the LDS stack (entry e lives at stack[e * kBlock]).  The node about to be
// visited is kept in a register instead of being pushed and popped again; the reference's "drop both children when
// stack_ptr >= 62" rule (scene.h:101-105) is evaluated on the same stack_ptr value the reference would see.
template <bool HAS_QUADS, bool STATS>
__device__ __forceinline__ bool intersect_stack(const float4* __restrict__ nodes, const float4* __restrict__ prims, int prim_stride,
                                                int* stack, bool live, f3 o, f3 d, float t_min, float t_max,
                                                float& t_hit, int& slot_hit, LaneCounters& cn) {
    float closest_t = t_max;
    slot_hit = -1;
    const f3 inv = mk3(rcp_rn(d.x), rcp_rn(d.y), rcp_rn(d.z));
    const float t_lo = mt_t_lo(t_min);
    int sp = 0;
    int cur = live ? 0 : -1;
    while (true) {
        if (cur < 0) {
            if (sp == 0) break;
            cur = stack[(--sp) * kBlock];
        }
        const float4 n0 = nodes[2 * cur], n1 = nodes[2 * cur + 1];
        if (STATS) cn.node_visits++;
        const int here = cur;
        cur = -1;
        if (!box_hit(n0, n1, o, inv, t_min, closest_t)) continue;
        const int a = __float_as_int(n0.w), b = __float_as_int(n1.w);
        if (b < 0) {                                       // leaf: -b primitives from slot a
            for (int i = 0; i < -b; i++) {
                if (STATS) cn.prim_tests++;
                leaf_prim<HAS_QUADS>(prims, prim_stride, a + i, o, d, t_lo, closest_t, slot_hit);
            }
        } else if (sp < 62) {                              // push right, visit left (= here + 1) next
            stack[(sp++) * kBlock] = b;
            cur = here + 1;
        }
    }
    t_hit = closest_t;
    return slot_hit >= 0;
}

// ---- TRAVERSAL_LANE: the same walk without a stack ---------------------------------------------------------------
// Pre-order numbering makes "pop" a table lookup: after a node whose box is missed the next node is its skip index,
// otherwise it is index + 1.  Valid while the reference's stack never overflows (tree depth <= 62).
template <bool HAS_QUADS, bool STATS>
__device__ __forceinline__ bool intersect_lane(const float4* __restrict__ nodes, const float4* __restrict__ prims, int prim_stride,
                                               int n_nodes, bool live, f3 o, f3 d, float t_min, float t_max,
                                               float& t_hit, int& slot_hit, LaneCounters& cn) {
    float closest_t = t_max;
    slot_hit = -1;
    const f3 inv = mk3(rcp_rn(d.x), rcp_rn(d.y), rcp_rn(d.z));
    const float t_lo = mt_t_lo(t_min);
    int cur = live ? 0 : n_nodes;
    while (cur < n_nodes) {
        const float4 n0 = nodes[2 * cur], n1 = nodes[2 * cur + 1];
        if (STATS) cn.node_visits++;
        const int a = __float_as_int(n0.w), b = __float_as_int(n1.w);
        const bool pass = box_hit(n0, n1, o, inv, t_min, closest_t);
        int next = cur + 1;
        if (!pass && b >= 0) next = a;
        if (pass && b < 0) {
            for (int i = 0; i < -b; i++) {
                if (STATS) cn.prim_tests++;
                leaf_prim<HAS_QUADS>(prims, prim_stride, a + i, o, d, t_lo, closest_t, slot_hit);
            }
        }
        cur = next;
    }
    t_hit = closest_t;
    return slot_hit >= 0;
}

// ---- TRAVERSAL_SWEEP: the WAVE walks the node indices once -----------------------------------------------------
// Every lane's cursor only moves forward through the pre-order, so one pass n = 0..N-1 with "lanes whose cursor == n
// take part" visits, per lane, exactly the nodes and primitives of the walks above, in the same order.  n is
// wave-uniform: node and primitive records come in through scalar loads (s_load_dwordx4 -> SGPR operands), there is
// no stack, no per-lane LDS read, and lanes at different depths of the tree never serialise against each other.
// The wave pays for the UNION of its lanes' visits, so this is used only for scenes of a few dozen primitives.
// Must be called from wave-uniform control flow (dead lanes pass live = false).
template <bool HAS_QUADS>
__device__ __forceinline__ void leaf_prim_uniform(const float4* prims, int prim_stride, int k, f3 o, f3 d, float t_lo,
                                                  float& closest_t, int& slot_hit) {
    // k is wave-uniform: these are broadcast LDS reads, the operands land in VGPRs (an SGPR operand would halve
    // the issue rate of every multiply/subtract that uses it)
    const float4 p0 = prims[k * prim_stride], p1 = prims[k * prim_stride + 1], p2 = prims[k * prim_stride + 2];
    const float eps = 1e-8f, eps_up = __uint_as_float(__float_as_uint(1e-8f) + 1u);
    float t;
    if (HAS_QUADS && __builtin_amdgcn_readfirstlane(__float_as_int(p0.w)) != 0) {   // wave-uniform branch
        const float4 p3 = prims[k * prim_stride + 3];
        // Quad::intersect: closest = t_max (= closest_t); each half accepts t < closest, second half sees the first's result
        const float t1 = mt_candidate(xyz(p0), xyz(p1), xyz(p2), o, d, eps_up, t_lo);     // (v00, v10, v11), |a| > eps
        const float c1 = min_raw(t1, closest_t);
        const float t2 = mt_candidate(xyz(p0), xyz(p2), xyz(p3), o, d, eps_up, t_lo);     // (v00, v11, v01)
        t = min_raw(t2, c1);                        // == closest_t when neither half was accepted
    } else {
        float tt = 0.0f;                                                                  // !(|a| < eps)
        const bool acc = mt_accept(xyz(p0), xyz(p1), xyz(p2), o, d, eps, t_lo, closest_t, tt);
        closest_t = acc ? tt : closest_t;
        slot_hit = acc ? k : slot_hit;
        return;
    }
    const bool closer = t < closest_t;            // quad: hit && temp.t < closest_t (scene.h:89-90)
    closest_t = min_raw(t, closest_t);
    slot_hit = closer ? k : slot_hit;
}

template <bool HAS_QUADS, bool STATS>
__device__ __forceinline__ bool intersect_sweep(const float4* nodes, const float4* prims, int prim_stride,
                                                int n_nodes, bool live, f3 o, f3 d, float t_min, float t_max,
                                                float& t_hit, int& slot_hit, LaneCounters& cn) {
    float closest_t = t_max;
    slot_hit = -1;
    const f3 inv = mk3(rcp_rn(d.x), rcp_rn(d.y), rcp_rn(d.z));
    const float t_lo = mt_t_lo(t_min);
    int cur = live ? 0 : n_nodes;
    for (int n = 0; n < n_nodes; n++) {
        if (cur == n) {
            // readfirstlane pins the index to an SGPR: inside this branch the optimiser knows cur == n and would
            // otherwise address the node through the per-lane cursor
            const int nu = __builtin_amdgcn_readfirstlane(n);
            const float4 n0 = nodes[2 * nu], n1 = nodes[2 * nu + 1];
            if (STATS) cn.node_visits++;
            const int a = __builtin_amdgcn_readfirstlane(__float_as_int(n0.w));
            const int b = __builtin_amdgcn_readfirstlane(__float_as_int(n1.w));   // wave-uniform
            const bool pass = box_hit(n0, n1, o, inv, t_min, closest_t);
            cur = n + 1;
            if (b < 0) {
                if (pass) {
                    for (int i = 0; i < -b; i++) {
                        if (STATS) cn.prim_tests++;
                        leaf_prim_uniform<HAS_QUADS>(prims, prim_stride, a + i, o, d, t_lo, closest_t, slot_hit);
                    }
                }
            } else if (!pass) cur = a;
        }
    }
    t_hit = closest_t;
    return slot_hit >= 0;
}

template <int MODE, bool HAS_QUADS, bool STATS>
__device__ __forceinline__ bool scene_intersect(const float4* __restrict__ nodes, const float4* __restrict__ prims, int prim_stride,
                                                int n_nodes, int* stack, bool live, f3 o, f3 d, float t_min, float t_max,
                                                float& t_hit, int& slot_hit, LaneCounters& cn) {
    if (MODE == TRAVERSAL_SWEEP) return intersect_sweep<HAS_QUADS, STATS>(nodes, prims, prim_stride, n_nodes, live, o, d, t_min, t_max, t_hit, slot_hit, cn);
    if (MODE == TRAVERSAL_LANE) return intersect_lane<HAS_QUADS, STATS>(nodes, prims, prim_stride, n_nodes, live, o, d, t_min, t_max, t_hit, slot_hit, cn);
    return intersect_stack<HAS_QUADS, STATS>(nodes, prims, prim_stride, stack, live, o, d, t_min, t_max, t_hit, slot_hit, cn);
}

// sampleCosineHemisphere (integrator.h:62-85) with the two uniforms already drawn
__device__ __forceinline__ f3 cosine_hemisphere(f3 n, float u, float v) {
    const float r = sqrt_rn(u);
    const float phi = (float)((double)2.0f * PTMI_PI_D * (double)v);      // 2.0f * M_PI * v with a double M_PI
    float sphi, cphi;
    ptmi_sincosf(phi, &sphi, &cphi);
    const float x = r * cphi;
    const float y = r * sphi;
    const float z = sqrt_rn(fmaxf(0.0f, 1.0f - u));
    f3 tangent, bitangent;
    if (n.z < -0.9999999f) {
        tangent = mk3(0.0f, -1.0f, 0.0f);
        bitangent = mk3(-1.0f, 0.0f, 0.0f);
    } else {
        const float a = rcp_rn(1.0f + n.z);
        const float b = -n.x * n.y * a;
        tangent = mk3(1.0f - n.x * n.x * a, b, -n.x);
        bitangent = mk3(b, 1.0f - n.y * n.y * a, -n.y);
    }
    return unit_vector(x * tangent + y * bitangent + z * n);
}

// ---------------------------------------------------------------------------------------------
// the hot kernel
// ---------------------------------------------------------------------------------------------
struct BounceArgs {
    DeviceScene sc; TileMap tm; PathState st; FrameParams fp;
    const int* queue_in; int n_in;          // n_in: upper bound known to the host (sizes the grid)
    const int* count_in;                    // device-side exact count of queue_in (nullptr: n_in is exact)
    int* queue_out; int* count_out;
    int segments;
    StatCounters* stats;
    int many_waves;                         // 1: more waves than the device holds at once (picks the 8-wave build of the packed walk)
    // count publishing (nullptr: off): the LAST workgroup of the launch to finish stores the launch's output count to a
    // host-mapped pinned slot and zeroes the next launch's counter, so a chunk's stream carries kernels only - no fill and
    // no 4-byte copy between two launches, each of which waits for a CU slot on a saturated GPU (r02: 11 % of c5frame)
    int* done_count;                        // workgroups of this launch that have finished (device memory, zero between launches)
    int* next_count;                        // the counter the next launch of this chunk will add to
    int* host_count;                        // pinned host memory, device address
    // cursor != nullptr: queue entries beyond the launch's threads are handed out through *cursor (zero at launch) to lanes whose
    // pixel has finished (the 8-wide walks; LaunchSchedule::refill_waves)
    int* cursor;
    // cost != nullptr: segments each pixel has taken in this frame, added to at the end of every visit (the next frame's launch
    // order: heaviest first, RenderState::orderByCost); cost_max: their maximum
    unsigned int* cost; unsigned int* cost_max;
};

// Per-lane path registers (the 88-byte HBM record, unpacked).
struct PathRegs {
    f3 o, d, tp, L, color;
    Rng rng;
    unsigned int sample_idx;
    int depth, px, py;
};

__device__ __forceinline__ void load_path(const PathState& st, const TileMap& tm, int slot, PathRegs& p) {
    const float4 A = st.A[slot], B = st.B[slot], C = st.C[slot], D = st.D[slot];
    const uint4 E = st.E[slot]; const uint2 F = st.F[slot];
    p.o = xyz(A); p.d = xyz(B); p.L = xyz(C); p.color = xyz(D);
    p.tp = mk3(A.w, B.w, C.w);
    const unsigned int meta = __float_as_uint(D.w);
    p.sample_idx = meta >> 8; p.depth = (int)(meta & 0xffu);
    p.rng = Rng{E.x, E.y, E.z, E.w, F.x, F.y};
    global_pixel(tm, slot, p.px, p.py);
}
__device__ __forceinline__ void store_path(const PathState& st, int slot, const PathRegs& p) {
    st.A[slot] = make_float4(p.o.x, p.o.y, p.o.z, p.tp.x);
    st.B[slot] = make_float4(p.d.x, p.d.y, p.d.z, p.tp.y);
    st.C[slot] = make_float4(p.L.x, p.L.y, p.L.z, p.tp.z);
    st.D[slot] = make_float4(p.color.x, p.color.y, p.color.z, __uint_as_float((p.sample_idx << 8) | (unsigned int)p.depth));
    st.E[slot] = make_uint4(p.rng.v0, p.rng.v1, p.rng.v2, p.rng.v3);
    st.F[slot] = make_uint2(p.rng.v4, p.rng.d);
}

// ---------------------------------------------------------------------------------------------
// Guided sampling: Grid over a PrecomputedCDF record (rendering/grid.h), MIS (integrator.h:91-167)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int linear_search_cdf(const float* __restrict__ cdf, int size, float xi) {   // grid.h:233-240
    xi = fminf(fmaxf(xi, 0.0f), 0.999999f);
    int r = size - 1;
    for (int i = size - 1; i >= 0; i--) if (xi < cdf[i]) r = i;      // first i with xi < cdf[i]
    return r;
}
// The cell's solid angle (grid.h:248-252) depends on theta_idx only: the eight values fmaxf(solid_angle, 1e-6f) are
// evaluated once per workgroup (fill_grid_solid_angles, same expressions) instead of one binary64 sincos per bounce.
__shared__ float g_grid_solid_angle[8];
__device__ __forceinline__ void fill_grid_solid_angles() {     // call from block-uniform code, before the first shade_step
    if (threadIdx.x < 8) {
        const int theta_idx = threadIdx.x;
        const float theta_center = (float)((double)(((float)theta_idx + 0.5f) * 0.125f) * (PTMI_PI_D * 0.5f));
        float st, ct;
        ptmi_sincosf(theta_center, &st, &ct);
        const float sin_theta = fmaxf(st, 0.01f);
        const float solid_angle = (float)(((double)sin_theta * ((PTMI_PI_D * 0.5f) / 8)) * (2.0f * PTMI_PI_D / 16));
        g_grid_solid_angle[theta_idx] = fmaxf(solid_angle, 1e-6f);
    }
    __syncthreads();
}
__device__ __forceinline__ float grid_pdf_for_cell(const float* __restrict__ g, int theta_idx, int phi_idx) {   // grid.h:242-253
    const float cell_value = g[kCdfPdf + theta_idx * 16 + phi_idx];
    if (cell_value < 1e-8f) return 1e-6f;
    const float cell_prob = cell_value / fmaxf(g[kCdfTotal], 1e-6f);
    return cell_prob / g_grid_solid_angle[theta_idx];
}
__device__ __forceinline__ f3 grid_sample(const float* __restrict__ g, f3 normal, Rng& rng, float& out_pdf) {   // grid.h:141-188
    const float xi1 = rng_uniform(rng);
    const float xi2 = rng_uniform(rng);
    const int theta_idx = linear_search_cdf(g + kCdfMarginal, 8, xi1);
    const int phi_idx = linear_search_cdf(g + kCdfRowCdfs + theta_idx * 16, 16, xi2);
    const float jitter_theta = rng_uniform(rng);
    const float jitter_phi = rng_uniform(rng);
    float theta = (float)((double)(((float)theta_idx + jitter_theta) * 0.125f) * (PTMI_PI_D * 0.5f));
    theta = fminf(theta, (float)(PTMI_PI_D * 0.5f - (double)0.01f));
    const float phi = (float)((double)((((float)phi_idx + jitter_phi) * 0.0625f) * 2.0f) * PTMI_PI_D);
    float sin_t, cos_t, sin_p, cos_p;
    ptmi_sincosf(theta, &sin_t, &cos_t);
    ptmi_sincosf(phi, &sin_p, &cos_p);
    f3 tangent, bitangent;
    build_frame(normal, tangent, bitangent);
    const f3 world = unit_vector((sin_t * cos_p) * tangent + (sin_t * sin_p) * bitangent + cos_t * normal);
    out_pdf = grid_pdf_for_cell(g, theta_idx, phi_idx);
    return world;
}
__device__ __forceinline__ float grid_compute_pdf(const float* __restrict__ g, f3 dir, f3 normal) {   // grid.h:200-216, 299-310
    f3 tangent, bitangent;
    build_frame(normal, tangent, bitangent);
    const float lx = dot(dir, tangent), ly = dot(dir, bitangent), lz = dot(dir, normal);
    const float theta = ptmi_acosf(fminf(fmaxf(lz, -1.0f), 1.0f));
    float phi = ptmi_atan2f(ly, lx);
    if (phi < 0.0f) phi = (float)((double)phi + (double)2.0f * PTMI_PI_D);
    if ((double)theta > PTMI_PI_D * 0.5f) return 0.0f;
    int theta_idx = (int)(((double)theta * ((double)2.0f / PTMI_PI_D)) * 8);
    int phi_idx = (int)(((double)phi * ((double)0.5f / PTMI_PI_D)) * 16);
    theta_idx = max(0, min(theta_idx, 7));
    phi_idx = max(0, min(phi_idx, 15));
    return grid_pdf_for_cell(g, theta_idx, phi_idx);
}
__device__ __forceinline__ float mis_power_heuristic(float pdf_a, float pdf_b) {   // integrator.h:91-96
    if (pdf_a <= 0.0f) return 0.0f;
    const float a2 = pdf_a * pdf_a, b2 = pdf_b * pdf_b;
    return a2 / (a2 + b2);
}
__device__ __forceinline__ f3 cosine_hemisphere(f3 n, float u, float v);
__device__ __forceinline__ f3 sample_mis(const float* __restrict__ g, f3 normal, Rng& rng, float& weight, float bsdf_prob) {   // integrator.h:112-167
    const float BSDF_PROB = fmaxf(fminf(bsdf_prob, 0.99f), 0.01f);
    const float GRID_PROB = 1.0f - BSDF_PROB;
    const float xi = rng_uniform(rng);
    f3 dir;
    if (xi < BSDF_PROB) {
        const float u = rng_uniform(rng), v = rng_uniform(rng);
        dir = cosine_hemisphere(normal, u, v);
        const float cos_theta = fmaxf(dot(dir, normal), 0.0f);
        const float pdf_bsdf = (float)((double)cos_theta / PTMI_PI_D);
        const float pdf_grid = grid_compute_pdf(g, dir, normal);
        const float mis_w = mis_power_heuristic(pdf_bsdf, pdf_grid);
        weight = (pdf_bsdf > 1e-6f) ? mis_w / BSDF_PROB : 0.0f;
    } else {
        float pdf_grid;
        dir = grid_sample(g, normal, rng, pdf_grid);
        const float cos_theta = fmaxf(dot(dir, normal), 0.0f);
        const float pdf_bsdf = (float)((double)cos_theta / PTMI_PI_D);
        const float mis_w = mis_power_heuristic(pdf_grid, pdf_bsdf);
        if (pdf_grid > 1e-6f && cos_theta > 0.0f) {
            const float w = (float)((double)(mis_w * cos_theta) / ((PTMI_PI_D * (double)pdf_grid) * (double)GRID_PROB));
            weight = fminf(w, 10.0f);
        } else weight = 0.0f;
    }
    return dir;
}

// One iteration of integrator()'s depth loop after the intersection (integrator.h:198-266), plus the end of the
// sample and the head of the next spp iteration (integrator.h:383-390) when the path ends.
// Returns true while the pixel still has a ray to trace; false once all spp samples are done.
// GUIDED: the grid / MIS branches of integrator.h:232-263 are compiled in (sampling_mode != SAMPLING_BSDF with CDF
// records present); the plain BSDF instantiation carries none of that code.
// Material record of leaf-order slot k: plain layout mats[3k..3k+2], packed layout (normal, table row) + (Kd, Ke) table.
struct MatSource { const float4* mats; const float4* mtab; const int* load_index; int stride = 1; };      // PACKED: entry k at mats[k * stride]
template <bool PACKED>
__device__ __forceinline__ void fetch_material(const MatSource& ms, int k, f3& n, f3& bsdf, f3& Le, int& row) {
    if (PACKED) {
        const float4 m = ms.mats[(size_t)k * ms.stride];
        n = xyz(m); row = __float_as_int(m.w);
        bsdf = xyz(ms.mtab[2 * row]); Le = xyz(ms.mtab[2 * row + 1]);
    } else {
        const float4 m = ms.mats[3 * k];
        n = xyz(m); row = __float_as_int(m.w);                                    // here: the load-order primitive index
        bsdf = xyz(ms.mats[3 * k + 1]); Le = xyz(ms.mats[3 * k + 2]);
    }
}
template <bool STATS, bool GUIDED, bool PACKED = false, bool BATCH = false>
__device__ __forceinline__ bool shade_step(const FrameParams& fp, const TileMap& tm, const MatSource& ms, const float* cdfs, PathRegs& p,
                                           bool hit, float t, int k, LaneCounters& cn, int slot) {
    bool end_sample = !hit;                                                       // integrator.h:198-201
    if (hit) {
        if (STATS) cn.hits++;
        f3 n, bsdf, Le; int row;
        fetch_material<PACKED>(ms, k, n, bsdf, Le, row);
        const f3 hp = p.o + t * p.d;                                              // triangle.h:90
        p.L = p.L + p.tp * Le;                                                    // integrator.h:204
        if (p.depth > 2) {                                                        // integrator.h:207-212
            const float max_tp = fmaxf(p.tp.x, fmaxf(p.tp.y, p.tp.z));
            const float rr_prob = fminf(max_tp, 0.95f);
            if (rng_uniform(p.rng) > rr_prob) end_sample = true;
            else p.tp = div_scalar(p.tp, rr_prob);
        }
        if (!end_sample) {
            p.tp = p.tp * bsdf;                                                   // integrator.h:215
            if (length(p.tp) < 1e-5f) end_sample = true;                          // integrator.h:218
            else {
                const f3 sn = dot(p.d, n) < 0 ? n : -n;                           // integrator.h:221-222
                // initGridFromPrimitive (integrator.h:31-57): the primitive's precomputed record, if it is valid
                const float* g = nullptr;
                if (GUIDED) {
                    const float* rec = cdfs + (size_t)(PACKED ? ms.load_index[k] : row) * kCdfDwords;
                    if (__float_as_int(rec[kCdfValid]) != 0) g = rec;
                }
                if (GUIDED && g) {
                    f3 next;
                    float weight = 1.0f;
                    if (fp.sampling_mode == 3) {                                  // SAMPLING_MIS, integrator.h:238-241
                        next = sample_mis(g, sn, p.rng, weight, fp.mis_bsdf_fraction);
                    } else {                                                      // pure grid sampling, integrator.h:242-257
                        float grid_pdf;
                        next = grid_sample(g, sn, p.rng, grid_pdf);
                        const float cos_theta = fmaxf(dot(next, sn), 0.0f);
                        weight = (float)((double)cos_theta / (PTMI_PI_D * (double)fmaxf(grid_pdf, 1e-6f)));
                        weight = fminf(fmaxf(weight, 0.0f), 10.0f);
                    }
                    p.tp = mk3(p.tp.x * weight, p.tp.y * weight, p.tp.z * weight);
                    p.depth++;
                    if (p.depth < fp.max_depth) {
                        p.o = hp + 1e-4f * sn;                                    // integrator.h:266
                        p.d = unit_vector(next);
                    } else end_sample = true;
                } else {                                                          // BSDF mode, or the cosine fallback :258-261
                    const float u = rng_uniform(p.rng);                           // integrator.h:63-64
                    const float v = rng_uniform(p.rng);
                    p.depth++;
                    if (p.depth < fp.max_depth) {
                        const f3 next = cosine_hemisphere(sn, u, v);              // integrator.h:230
                        p.o = hp + 1e-4f * sn;                                    // integrator.h:266
                        p.d = unit_vector(next);                                  // Ray ctor normalises again
                    } else end_sample = true;                                     // loop bound; the draws above are still consumed
                }
            }
        }
    }
    if (end_sample) {
        p.color = p.color + p.L;                                                  // integrator.h:390
        p.sample_idx++;
        if (!BATCH) {
            if (p.sample_idx >= (unsigned int)fp.spp) return false;
        } else if ((p.sample_idx & fp.sample_mask) >= (unsigned int)fp.spp) {     // the spp loop of this frame is through
            const unsigned int frame = p.sample_idx >> 16;
            if (frame + 1u >= (unsigned int)fp.n_frames) return false;
            // frame batch: bank this frame's colour sum and go straight on with the next frame's first sample
            fp.frame_color[frame * (unsigned int)fp.n_local + (unsigned int)slot] = make_float4(p.color.x, p.color.y, p.color.z, 0.0f);   // < 2^31 (host check)
            p.sample_idx = (frame + 1u) << 16;
            p.color = mk3(0.0f, 0.0f, 0.0f);
        }
        camera_ray(fp, tm, p.px, p.py, p.rng, p.o, p.d);                          // next iteration of the spp loop
        p.tp = mk3(1.0f, 1.0f, 1.0f); p.L = mk3(0.0f, 0.0f, 0.0f); p.depth = 0;
    }
    return true;
}

// stage the scene into LDS (when LDS_GEOM) and return the LDS cursor after it
template <bool LDS_GEOM>
__device__ __forceinline__ float4* stage_scene(const DeviceScene& sc, float4* lds, const float4*& nodes, const float4*& prims, const float4*& mats) {
    nodes = sc.nodes; prims = sc.prims; mats = sc.mats;
    if (LDS_GEOM) {
        const int n_node_vec = 2 * sc.n_nodes, n_prim_vec = sc.prim_stride * sc.n_prims, n_mat_vec = 3 * sc.n_prims;
        for (int i = threadIdx.x; i < n_node_vec; i += kBlock) lds[i] = sc.nodes[i];
        for (int i = threadIdx.x; i < n_prim_vec; i += kBlock) lds[n_node_vec + i] = sc.prims[i];
        for (int i = threadIdx.x; i < n_mat_vec; i += kBlock) lds[n_node_vec + n_prim_vec + i] = sc.mats[i];
        nodes = lds; prims = lds + n_node_vec; mats = lds + n_node_vec + n_prim_vec;
        lds += n_node_vec + n_prim_vec + n_mat_vec;
        __syncthreads();
    }
    return lds;
}

// kernel tail shared by both bounce kernels: active-path compaction (one atomic per wave reserves queue space, lanes
// scatter by prefix popcount) and the optional workload counters
template <bool STATS, bool COMPACT = true>
__device__ __forceinline__ void finish_launch(const BounceArgs& a, bool alive, int slot, const LaneCounters& cn) {
    if (COMPACT) {
        const unsigned long long mask = __ballot(alive);
        const int lane = threadIdx.x & 63;
        int base = 0;
        if (lane == 0 && mask) base = atomicAdd(a.count_out, __popcll(mask));
        base = __shfl(base, 0);
        if (alive) a.queue_out[base + __popcll(mask & ((1ull << lane) - 1ull))] = slot;
    }
    if (STATS) {
        unsigned long long r = cn.rays, nv = cn.node_visits, pt = cn.prim_tests, h = cn.hits, tv = cn.top_visits;
        unsigned long long cc = cn.cert_chain, cf = cn.cert_fallback;
        for (int off = 32; off > 0; off >>= 1) {
            r += __shfl_down(r, off); nv += __shfl_down(nv, off); pt += __shfl_down(pt, off); h += __shfl_down(h, off); tv += __shfl_down(tv, off);
            cc += __shfl_down(cc, off); cf += __shfl_down(cf, off);
        }
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&a.stats->rays, r); atomicAdd(&a.stats->node_visits, nv);
            atomicAdd(&a.stats->prim_tests, pt); atomicAdd(&a.stats->hits, h);
            if (tv) atomicAdd(&a.stats->top_node_visits, tv);
            if (cc) atomicAdd(&a.stats->cert_chain, cc);
            if (cf) atomicAdd(&a.stats->cert_fallback, cf);
        }
    }
}

// Tail of every bounce kernel when count publishing is on: every WAVE of the grid passes here exactly once (also the ones
// that found nothing to do), after its own reservation in count_out - no workgroup barrier, a finished wave leaves at once
// (with a barrier in front of one arrival per workgroup the waves that finish early keep their registers until the
// workgroup's slowest is through: whole 1 M-triangle frame -4 %).  count_out is only ever touched by device-scope atomics,
// so the last arrival reads the sum.
__device__ __forceinline__ void publish_count(const BounceArgs& a) {
    if (!a.host_count) return;
    if ((threadIdx.x & 63) == 0) {
        __threadfence();
        if (atomicAdd(a.done_count, 1) == (int)(gridDim.x * (kBlock / 64)) - 1) {
            const int c = atomicAdd(a.count_out, 0);
            atomicExch(a.next_count, 0); atomicExch(a.next_count + 1, 0);      // output count, refill cursor
            atomicExch(a.done_count, 0);
            __hip_atomic_store(a.host_count, c, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// The end of a pixel's VISIT to a launch: all its samples are done (!more), or it has had its a.segments segments of this launch.
// Its state goes back to HBM; a pixel that is not through joins the output queue; and when the launch has fewer lanes than queue
// entries (BounceArgs::cursor) the lane takes the next entry no lane has taken yet - true: `slot` / `p` hold that pixel, with
// segs_left segments to go.  Call from the lanes whose visits end now (divergent code): they share one atomicAdd per counter.
__device__ __forceinline__ bool end_visit(const BounceArgs& a, int n_in, bool more, int& slot, PathRegs& p, int& segs_left) {
    store_path(a.st, slot, p);
    if (a.cost) {                                      // what this visit cost, for the next frame's launch order
        const unsigned int c = a.cost[slot] + (unsigned int)(a.segments - segs_left);
        a.cost[slot] = c;
        if (!more) atomicMax(a.cost_max, c);
    }
    const int lane = threadIdx.x & 63;
    const unsigned long long ending = __ballot(1), surviving = __ballot(more);
    const int first = __ffsll((long long)ending) - 1;
    int out_base = 0, in_base = 0;
    if (lane == first) {
        if (surviving) out_base = atomicAdd(a.count_out, __popcll(surviving));
        if (a.cursor) in_base = atomicAdd(a.cursor, __popcll(ending));
    }
    out_base = __shfl(out_base, first); in_base = __shfl(in_base, first);
    if (more) a.queue_out[out_base + __popcll(surviving & ((1ull << lane) - 1ull))] = slot;
    const int entry = (int)(gridDim.x * kBlock) + in_base + __popcll(ending & ((1ull << lane) - 1ull));
    if (!(a.cursor && entry < n_in)) return false;
    slot = a.queue_in ? a.queue_in[entry] : entry;
    load_path(a.st, a.tm, slot, p);
    segs_left = a.segments;
    return true;
}

// ---- ptmi_bounce: segment-synchronous form (SWEEP and STACK walks) ------------------------------------------------
// Every wave traces one ray segment per lane, then shades, K times.  LDS: [nodes | prims | mats] when LDS_GEOM (always
// for SWEEP), then the traversal stacks (STACK only).
// amdgpu_num_sgpr(80): with <= 80 SGPRs eight 256-thread workgroups fit a CU instead of six
// (MI355X_MICROARCH.md, residency rule); measured +2.4 %, no spills.
// GUIDED instantiations would take ~100 VGPRs (4 waves per SIMD); capped at 80 (6 waves, 68 bytes of spills): grid
// sampling +13 %, MIS +9 % on the benchmark frame (5 waves +8 %, 7 the same as 6, 8 waves +10 % / +3 %).
template <int MODE, bool LDS_GEOM, bool HAS_QUADS, bool STATS, bool GUIDED, bool BATCH>
__device__ __forceinline__ void bounce_body(const BounceArgs& a) {
    extern __shared__ float4 smem[];
    static_assert(MODE != TRAVERSAL_SWEEP || LDS_GEOM, "the sweep reads the scene through LDS broadcasts");
    const int n_in = a.count_in ? *a.count_in : a.n_in;
    if ((int)(blockIdx.x * kBlock) >= n_in) return;      // grid was sized from a stale (larger) count: nothing to do
    const int idx = blockIdx.x * kBlock + threadIdx.x;
    const bool active = idx < n_in;
    const float4 *nodes, *prims, *mats;
    float4* lds = stage_scene<LDS_GEOM>(a.sc, smem, nodes, prims, mats);
    int* stack = reinterpret_cast<int*>(lds) + threadIdx.x;
    if (GUIDED) fill_grid_solid_angles();

    const int slot = active ? (a.queue_in ? a.queue_in[idx] : idx) : 0;
    bool alive = active;
    PathRegs p = {};
    if (active) load_path(a.st, a.tm, slot, p);
    LaneCounters cn = {0, 0, 0, 0, 0, 0, 0};

    for (int seg = 0; seg < a.segments; seg++) {
        if (!__any(alive)) break;
        // the whole wave enters the traversal together (finished lanes ride along masked): required by SWEEP
        float t = 0.0f; int k = -1;
        if (STATS && alive) cn.rays++;
        const bool hit = scene_intersect<MODE, HAS_QUADS, STATS>(nodes, prims, a.sc.prim_stride, a.sc.n_nodes, stack, alive,
                                                               p.o, p.d, 1e-4f, FLT_MAX, t, k, cn);
        if (alive) alive = shade_step<STATS, GUIDED, false, BATCH>(a.fp, a.tm, MatSource{mats, nullptr, nullptr}, a.sc.cdfs, p, hit, t, k, cn, slot);
    }

    if (active) store_path(a.st, slot, p);
    finish_launch<STATS>(a, alive, slot, cn);
}
template <int MODE, bool LDS_GEOM, bool HAS_QUADS, bool STATS, bool GUIDED, bool BATCH>
__global__ __launch_bounds__(kBlock, GUIDED ? 6 : (BATCH && MODE == TRAVERSAL_SWEEP ? 8 : 1)) __attribute__((amdgpu_num_sgpr(80))) void ptmi_bounce(BounceArgs a) {
    bounce_body<MODE, LDS_GEOM, HAS_QUADS, STATS, GUIDED, BATCH>(a);
    publish_count(a);
}

// ---- ptmi_bounce_phased: wave-scheduled phases (LANE walk for large scenes) -----------------------------------------
// A lane is always in one of three phases: NODE (next pre-order node to visit), PRIM (pending primitives of a leaf whose
// box it hit) or SHADE (traversal finished).  Every iteration the WAVE executes the one phase that most of its lanes
// are waiting for.  A lane whose ray ends early shades and starts its next segment while its neighbours are still
// walking the tree, instead of idling until the longest ray of the wave is done (segment-synchronous per-lane walk on
// the 1M-triangle scene: 13.8 % VALU lane utilisation).  Per lane the sequence of node visits, primitive tests and RNG
// draws is exactly the reference's; only the interleaving between lanes changes.
#ifdef PTMI_TRACE_WAVES
// experiment-only build (tools/wave_trace.py, never the shipped library): where a wave's clocks go.  [0] walk clocks
// [1] shade clocks [2] walk decisions [3] shade decisions [4] lanes advanced by walk decisions [5] lanes shaded [6] wave clocks
// [7] waves [8] longest wave [9] clocks outside the loop [10] living lanes summed over decisions [11] node decisions [12] node clocks
__device__ unsigned long long g_trace[16];
#define PTMI_TR(...) __VA_ARGS__
#else
#define PTMI_TR(...)
#endif
#ifndef PTMI_NODE_BURST
#define PTMI_NODE_BURST 3
#endif
template <bool LDS_GEOM, bool HAS_QUADS, bool STATS, bool GUIDED, bool PACKED, bool BATCH>
__device__ __forceinline__ void bounce_phased_body(const BounceArgs& a) {
    extern __shared__ float4 smem[];
    static_assert(!(PACKED && LDS_GEOM), "the packed layout is for scenes that do not fit LDS");
    const int n_in = a.count_in ? *a.count_in : a.n_in;
    if ((int)(blockIdx.x * kBlock) >= n_in) return;      // grid was sized from a stale (larger) count: nothing to do
    const int idx = blockIdx.x * kBlock + threadIdx.x;
    const bool active = idx < n_in;
    const float4 *nodes, *prims, *mats;
    stage_scene<LDS_GEOM>(a.sc, smem, nodes, prims, mats);
    if (PACKED) nodes = a.sc.gnodes;
    const int n_top = PACKED ? a.sc.n_top : 0;           // the top of the packed tree, staged into LDS (device_scene.h)
    if (PACKED && n_top) {
        for (int i = threadIdx.x; i < 2 * n_top; i += kBlock) smem[i] = a.sc.gnodes[i];
        __syncthreads();
    }
    const MatSource ms = PACKED ? MatSource{a.sc.gmats, a.sc.mtab, a.sc.load_index} : MatSource{mats, nullptr, nullptr};
    if (GUIDED) fill_grid_solid_angles();

    const int slot = active ? (a.queue_in ? a.queue_in[idx] : idx) : 0;
    bool alive = active;
    PathRegs p = {};
    if (active) load_path(a.st, a.tm, slot, p);
    LaneCounters cn = {0, 0, 0, 0, 0, 0, 0};

    enum { PH_NODE = 0, PH_PRIM = 1, PH_SHADE = 2, PH_DONE = 3 };
    const int n_nodes = PACKED ? a.sc.n_pos : a.sc.n_nodes, prim_stride = a.sc.prim_stride;     // cursor >= n_nodes: walk finished
    const float t_min = 1e-4f, t_lo = mt_t_lo(t_min);
    int phase = alive ? PH_NODE : PH_DONE;
    int segs_left = a.segments;
    int cur = 0, pk = 0, pend = 0, slot_hit = -1;
    float closest_t = FLT_MAX;
    f3 inv = mk3(rcp_rn(p.d.x), rcp_rn(p.d.y), rcp_rn(p.d.z));
    if (STATS && alive) cn.rays++;

    PTMI_TR(const long long tr_t0 = clock64(); long long tr_walk = 0, tr_shade = 0, tr_node = 0; unsigned tr_nw = 0, tr_ns = 0, tr_lw = 0, tr_ls = 0, tr_alive = 0, tr_nn = 0;)
    while (true) {
        const int c_node = __popcll(__ballot(phase == PH_NODE));
        const int c_prim = __popcll(__ballot(phase == PH_PRIM));
        const int c_shade = __popcll(__ballot(phase == PH_SHADE));
        if (c_node + c_prim + c_shade == 0) break;
        PTMI_TR(const long long tr_a = clock64(); tr_alive += c_node + c_prim + c_shade;
                const int tr_kind = c_node >= c_prim && c_node >= c_shade ? 0 : c_prim >= c_shade ? 1 : 2;
                if (tr_kind == 0) { tr_nw++; tr_nn++; tr_lw += c_node; } else if (tr_kind == 1) { tr_nw++; tr_lw += c_prim; } else { tr_ns++; tr_ls += c_shade; })
        if (c_node >= c_prim && c_node >= c_shade) {
            // a short burst of node steps per scheduling decision: in large scenes a ray visits ~10 nodes between two
            // leaves, and the three ballots + branches of a decision cost about as much as a node test
#pragma unroll
            for (int burst = 0; burst < PTMI_NODE_BURST; burst++) {
                if (phase == PH_NODE) {                                // one node of Scene::intersect_bvh_optimized (scene.h:63-106)
                    float4 n0, n1;
                    if (PACKED && cur < n_top) { n0 = smem[2 * cur]; n1 = smem[2 * cur + 1]; if (STATS) cn.top_visits++; }
                    else { n0 = nodes[2 * cur]; n1 = nodes[2 * cur + 1]; }
                    if (STATS) cn.node_visits++;
                    const int na = __float_as_int(n0.w), nb = __float_as_int(n1.w);
                    const bool pass = box_hit(n0, n1, p.o, inv, t_min, closest_t);
                    int next;
                    if (PACKED) {                                      // explicit links (device_scene.h, PACKED LAYOUT)
                        if (nb < 0) {
                            next = ~nb;
                            if (pass) { pk = na >> 3; pend = pk + (na & 7); phase = PH_PRIM; }
                        } else next = pass ? nb : na;
                    } else {                                           // pre-order: left child = cur + 1, a = skip index
                        next = cur + 1;
                        if (nb < 0) {
                            if (pass) { pk = na; pend = na - nb; phase = PH_PRIM; }
                        } else if (!pass) next = na;
                    }
                    cur = next;
                    if (phase == PH_NODE && cur >= n_nodes) phase = PH_SHADE;
                }
            }
        } else if (c_prim >= c_shade) {
            if (phase == PH_PRIM) {                                    // one primitive of the leaf loop (scene.h:85-99)
                if (STATS) cn.prim_tests++;
                if (PACKED && !HAS_QUADS) leaf_prim_packed(a.sc.gprims, pk, p.o, p.d, t_lo, closest_t, slot_hit);
                else leaf_prim<HAS_QUADS>(prims, prim_stride, pk, p.o, p.d, t_lo, closest_t, slot_hit);
                pk++;
                if (pk == pend) phase = cur >= n_nodes ? PH_SHADE : PH_NODE;
            }
        } else {
            if (phase == PH_SHADE) {
                const bool more = shade_step<STATS, GUIDED, PACKED, BATCH>(a.fp, a.tm, ms, a.sc.cdfs, p, slot_hit >= 0, closest_t, slot_hit, cn, slot);
                segs_left--;
                if (!more) { alive = false; phase = PH_DONE; }
                else if (segs_left == 0) phase = PH_DONE;              // state goes back to HBM with the next ray ready
                else {
                    cur = 0; slot_hit = -1; closest_t = FLT_MAX;
                    inv = mk3(rcp_rn(p.d.x), rcp_rn(p.d.y), rcp_rn(p.d.z));
                    phase = PH_NODE;
                    if (STATS) cn.rays++;
                }
            }
        }
        PTMI_TR(const long long tr_d = clock64() - tr_a; if (tr_kind == 2) tr_shade += tr_d; else tr_walk += tr_d; if (tr_kind == 0) tr_node += tr_d;)
    }
    PTMI_TR(const long long tr_loop = clock64() - tr_t0;)

    if (active) store_path(a.st, slot, p);
    finish_launch<STATS>(a, alive, slot, cn);
#ifdef PTMI_TRACE_WAVES
    if ((threadIdx.x & 63) == 0) {
        const unsigned long long tot = (unsigned long long)(clock64() - tr_t0);
        const unsigned long long v[13] = {(unsigned long long)tr_walk, (unsigned long long)tr_shade, tr_nw, tr_ns, tr_lw, tr_ls, tot, 1ull, 0ull,
                                          tot - (unsigned long long)tr_loop, tr_alive, tr_nn, (unsigned long long)tr_node};
        for (int i = 0; i < 13; i++) if (i != 8) atomicAdd(&g_trace[i], v[i]);
        atomicMax(&g_trace[8], tot);
    }
#endif
}
template <bool LDS_GEOM, bool HAS_QUADS, bool STATS, bool GUIDED, bool PACKED, bool BATCH>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_num_sgpr(80))) void ptmi_bounce_phased(BounceArgs a) {
    bounce_phased_body<LDS_GEOM, HAS_QUADS, STATS, GUIDED, PACKED, BATCH>(a);
    publish_count(a);
}
// The packed walk of a triangle scene, BSDF sampling, bounded to 8 waves per SIMD (64 VGPRs, 11 spilled outside the walk
// loop): for frames with more waves than the device holds, where a wave more per SIMD is worth +5 % (whole 1 M-triangle frame
// 1 037 -> 1 088 Msamples/s, half +4.6 %); the chain-bound case keeps the 7-wave kernel (an eighth of that frame: -5 % with
// this one) - host/application_state.cpp decides per launch
template <bool STATS, bool BATCH>
__global__ __launch_bounds__(kBlock, 8) __attribute__((amdgpu_num_sgpr(80))) void ptmi_bounce_packed_w8(BounceArgs a) {
    bounce_phased_body<false, false, STATS, false, true, BATCH>(a);
    publish_count(a);
}


// ---- ptmi_bounce_wide: the opt-in fast tree (csrc/wide_bvh.h) --------------------------------------------------------------
// The phased scheduling of ptmi_bounce_phased over the 8-wide SAH tree.  A NODE step enters ONE inner child: it takes the
// front-most pending child of the lane's current group (or pops a group), fetches that child's 128-byte node - one L2 line,
// or LDS for the top levels - and decides its eight children at once: hit inner children become the new current group, the
// triangles of hit leaf children a 24-bit mask that the following PRIM steps test one by one.  Stack: one (child_base,
// imask << 8 | pending) pair per tree level in LDS, entry e of lane l at stack[e * kBlock + l] - conflict-free whatever e.
// The triangle test is the exact walk's (mt_hit = mt_accept's arithmetic); equal-t hits keep the smaller reference slot.
// Measured and dropped: a UNIFIED step (a lane's pending triangle and its next node fetched and tested in one step, two phases
// to vote between): 92 registers, 5 waves per SIMD: 1 610 / 2 322 Msamples/s (an eighth / the whole 1 M-triangle frame) against
// 1 719 / 2 405 for this form at 6 waves; bounded to 6 waves it spills inside the loop (867 / 1 100).
// CERT (TRAVERSAL_CERTIFIED): the same walk made EXACT - under one stated premise.
//   PREMISE (P).  Every triangle j whose Moller-Trumbore test (mt_hit = the reference's arithmetic) accepts the ray at a distance
//   t_j inside [t_min, c] is TESTED by the fast walk while its closest_t is >= c: the slab tests of the fast walk pass for j's
//   leaf child and for all its ancestors.  Geometrically that is what "conservative boxes" means; in floating point it needs
//   the computed hit point o + t_j d to lie within the pad of j's box (2^-16 of the scene's scale, host/wide_bvh.cpp).  The
//   computed t_j is off by at most about 2^-20 (|o - v0| + t_j) / kappa, kappa = |a| / (|e1| |e2|) (a = e1 . (d x e2), the
//   determinant the test divides by; kappa = sin of the triangle's corner angle x cos of the incidence angle).  With origin and
//   hit inside the scene (|o - v0| + t_j <= 2 S) that is within the pad for kappa >= 1 / 8: there (P) is PROVEN.  Below that
//   floor - rays within about 7 degrees of a triangle's plane, slivers, needles; down at |a| ~ 1e-8 the reference itself accepts
//   distances that are numerical noise - (P) is TESTED, NOT PROVEN: tools/certified_soak.py (sheets skimmed at 1e-7 .. 1e-3 rad, fences of needles with 1e-8 .. 1e-5 short
//   edges, stacked layers of large triangles 1e-4 apart, a sloppy exporter's degenerate primitives, an outlier a million units
//   away; 0 mismatches against the walk over the reference's tree), VERDICT r3's 750 000 adversarial rays (0 mismatches) and
//   BASELINE configs[4] at its full 8.6 G samples (identical frames).
//   Under (P) the fast walk's hit (t*, k*) is the global minimum over all accepted triangles, and the reference's own walk
//   (scene.h:50-110) returns exactly the same hit if (1) it reaches k*'s leaf and (2) no second triangle is hit at exactly t*.
// (1): the reference enters a node when `!(min(t_exit, closest_t) < t_entry)` holds for it and all its ancestors, closest_t being
// whatever it is at that moment - never below the final t*; the test is monotone in closest_t, so if every ancestor of k*'s leaf
// (leaf included) passes with closest_t = t* it passes in the reference.  One fetch (the leaf's box, see VERIFY below) shows
// that for 99.7 % of the hits; the rest evaluate these slab tests themselves (box_hit, the exact walk's arithmetic) from a
// per-leaf list of ancestor node indices, leaf first, four nodes per step fetched in parallel, until a box holds the hit point
// with the margin - mostly the parent or grandparent.
// (2): every triangle hit at t* is tested by the fast walk too (P), so a tie shows as `t == closest_t` there.  A ray for which
// (1) or (2) cannot be shown - a grazed box, a shared edge, an origin outside the range the boxes were padded for - is walked
// again by the reference's own walk (intersect_lane) inside this kernel: about one ray in 10^5 on the 1 M-triangle scene.  No hit
// at all needs no check: under (P) the reference can only accept triangles the fast walk would have found.
// (What was measured on the way and dropped: EXPERIMENTS.md.)
#ifndef PTMI_PRIM_BATCH
#define PTMI_PRIM_BATCH 2
#endif
#ifndef PTMI_QUAD_BATCH
#define PTMI_QUAD_BATCH 2
#endif
#ifdef PTMI_TRACE_WAVES
// experiment-only build (tools/wide_trace.py): one 12-word record per wave of ptmi_bounce_wide - start (wall_clock64, low word),
// duration in ticks, decisions and lanes advanced per kind (NODE / PRIM / SHADE), shader clocks per kind, lanes the wave started with
// (low byte of the last word; above it: ticks after which fewer than 32 of its lanes still had work, 0 = never)
constexpr unsigned int kWideTraceCap = 1u << 19;
__device__ unsigned int g_wt_n;
__device__ unsigned int g_wt[kWideTraceCap * 12];
#endif
template <bool STATS, bool GUIDED, bool BATCH, bool CERT, bool QUADS>
__device__ __forceinline__ void bounce_wide_body(const BounceArgs& a) {
    extern __shared__ float4 smem[];
    PTMI_TR(const unsigned long long wt_t0 = (unsigned long long)wall_clock64(); unsigned int wt_n[3] = {0, 0, 0}, wt_l[3] = {0, 0, 0}, wt_half = 0; unsigned long long wt_c[3] = {0, 0, 0};)
    const int n_in = a.count_in ? *a.count_in : a.n_in;
    if ((int)(blockIdx.x * kBlock) >= n_in) return;      // grid was sized from a stale (larger) count: nothing to do
    const int idx = blockIdx.x * kBlock + threadIdx.x;
    const bool active = idx < n_in;
    const int n_top = a.sc.w_top;
    uint4* top = reinterpret_cast<uint4*>(smem);
    for (int i = threadIdx.x; i < 8 * n_top; i += kBlock) top[i] = a.sc.wnodes[i];
    uint2* stack = reinterpret_cast<uint2*>(top + 8 * n_top) + threadIdx.x;
    __syncthreads();
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) u32x4 LdsU4;
    typedef __attribute__((address_space(1))) u32x4 GlobalU4;
    auto u4 = [](u32x4 v) { return make_uint4(v.x, v.y, v.z, v.w); };
    const LdsU4* top_lds = (const LdsU4*)top;
    const GlobalU4* wnodes_g = (const GlobalU4*)a.sc.wnodes;
    // (the certified walk shades from the line its proof has just read: DeviceScene::wcert)
    const MatSource ms = CERT ? MatSource{a.sc.wcert + 2, a.sc.wmtab, a.sc.wload_index, kWideCertStride} : MatSource{a.sc.wmats, a.sc.wmtab, a.sc.wload_index, 1};
    if (GUIDED) fill_grid_solid_angles();

    int slot = active ? (a.queue_in ? a.queue_in[idx] : idx) : 0;
    bool alive = active;
    PathRegs p = {};
    if (active) load_path(a.st, a.tm, slot, p);
    LaneCounters cn = {0, 0, 0, 0, 0, 0, 0};

    enum { PH_NODE = 0, PH_PRIM = 1, PH_SHADE = 2, PH_DONE = 3, PH_VERIFY = 4, PH_EXACT = 5 };
    constexpr int kTieFlag = 0x40000000;      // CERT: set in slot_hit while a second triangle is accepted at exactly closest_t
    const float t_min = 1e-4f, t_lo = mt_t_lo(t_min);
    // CERT keeps no state of its own through the walk: a tie is a flag in slot_hit (a closer hit clears it - only a tie at the
    // final t* matters), a ray that has to take the reference's walk is a phase (PH_EXACT), and the proof's cursor (next 4-node
    // chunk of the hit leaf's ancestor list, chunks left) lives in g_base / t_base, which are dead once the walk has ended
    int phase = alive ? PH_NODE : PH_DONE;
    int segs_left = a.segments;
    int slot_hit = -1, sp = 0;
    float closest_t = FLT_MAX;
    f3 inv = mk3(wide_inv(p.d.x), wide_inv(p.d.y), wide_inv(p.d.z));
    uint32_t octinv = wide_octinv(inv);
    uint32_t g_base = 0u, g_bits = (1u << 8) | (1u << octinv);      // the root: slot 0 of a virtual parent
    uint32_t t_base = 0u, t_mask = 0u;
    if (STATS && alive) cn.rays++;
    // where the walk goes when it has run out of nodes and triangles
    auto after_walk = [&]() -> int {
        if (!CERT) return PH_SHADE;
        if (slot_hit < 0) return PH_SHADE;
        if (slot_hit & kTieFlag) return PH_EXACT;                      // a tie at t*: let the reference's walk decide
        t_base = 0xffffffffu;                                          // first the one-fetch certificate, then (rarely) the chain
        return PH_VERIFY;
    };
    auto origin_in_range = [&]() { return fmaxf(fabsf(p.o.x), fmaxf(fabsf(p.o.y), fabsf(p.o.z))) <= a.sc.w_guard; };
    if (CERT && alive && !origin_in_range()) phase = PH_EXACT;

#ifdef PTMI_WALK_CAP
    int walk_cap_n = 0;
#endif
    while (true) {
        const int c_node = __popcll(__ballot(phase == PH_NODE));
        const int c_prim = __popcll(__ballot(phase == PH_PRIM));
        // a lane whose hit has to be proven (VERIFY) votes with the SHADE lanes and takes one proof step right before that
        // phase's work - for almost every hit the only one (the leaf-box fetch).  As a phase of its own (round 3's first form)
        // the proof cost a wave iteration per ray and its stragglers waited for a majority: c5tile 1 407 -> 1 455 Msamples/s
        const int c_shade = __popcll(__ballot(phase == PH_SHADE || (CERT && phase >= PH_VERIFY)));
        if (c_node + c_prim + c_shade == 0) break;
        PTMI_TR(if (!wt_half && c_node + c_prim + c_shade < 32) wt_half = (unsigned int)((unsigned long long)wall_clock64() - wt_t0);)
        PTMI_TR(const int wt_k = c_node >= c_prim && c_node >= c_shade ? 0 : c_prim >= c_shade ? 1 : 2; const long long wt_a = clock64();
                wt_n[wt_k]++; wt_l[wt_k] += wt_k == 0 ? c_node : wt_k == 1 ? c_prim : c_shade;)
        if (c_node >= c_prim && c_node >= c_shade) {
            if (phase == PH_NODE) {
                if ((g_bits & 0xffu) == 0u) { sp--; const uint2 e = stack[sp * kBlock]; g_base = e.x; g_bits = e.y; }
                const int bit = 31 - __clz((int)(g_bits & 0xffu));
                g_bits ^= 1u << bit;
                const uint32_t child = (uint32_t)bit ^ octinv;
                const uint32_t ni = g_base + (uint32_t)__popc((g_bits >> 8) & ((1u << child) - 1u));
                if (g_bits & 0xffu) { stack[sp * kBlock] = make_uint2(g_base, g_bits); sp++; }
                uint4 q0, q1, q2, q3, q4, q5, q6;
                if ((int)ni < n_top) {
                    // explicit address spaces: left generic, the two branches are merged into ONE flat_load through a selected pointer
                    const LdsU4* q = top_lds + 8 * ni;
                    q0 = u4(q[0]); q1 = u4(q[1]); q2 = u4(q[2]); q3 = u4(q[3]); q4 = u4(q[4]); q5 = u4(q[5]); q6 = u4(q[6]);
                    if (STATS) cn.top_visits++;
                } else {
                    const GlobalU4* q = wnodes_g + 8 * (size_t)ni;
                    q0 = u4(q[0]); q1 = u4(q[1]); q2 = u4(q[2]); q3 = u4(q[3]); q4 = u4(q[4]); q5 = u4(q[5]); q6 = u4(q[6]);
                }
                if (STATS) cn.node_visits++;
                const WideStep st = wide_node_test(q0, q1, q2, q3, q4, q5, q6, p.o, inv, octinv, t_min, closest_t);
                g_base = st.child_base; g_bits = (st.imask << 8) | st.inner;
                t_base = st.tri_base; t_mask = st.tris;
                if (t_mask) phase = PH_PRIM;
                else if ((g_bits & 0xffu) == 0u && sp == 0) phase = after_walk();
            }
        } else if (c_prim >= c_shade) {
            if (phase == PH_PRIM) {
                // up to PTMI_PRIM_BATCH pending triangles per step: their records are fetched together (one latency), then tested
                // in ascending order
                // (scenes with quads: 64-byte records - v0 | type, e1, e2, e3, the layout of d_prims -, two per step as well: 4 096 / 16 384
                // planar quads 2 180 -> 2 290 / 1 730 -> 1 875 Msamples/s)
                constexpr int kPB = QUADS ? PTMI_QUAD_BATCH : PTMI_PRIM_BATCH;
                int kk[kPB]; bool has[kPB]; f3p r0[kPB], r1[kPB], r2[kPB]; float4 q0[kPB], q1[kPB], q2[kPB], q3[kPB];
#pragma unroll
                for (int b = 0; b < kPB; b++) {
                    has[b] = t_mask != 0u;
                    kk[b] = has[b] ? (int)t_base + __ffs((int)t_mask) - 1 : (b ? kk[b - 1] : 0);
                    t_mask &= t_mask - 1u;                                 // 0 stays 0
                    if (QUADS) {
                        const float4* r = a.sc.wqprims + 4 * (size_t)kk[b];
                        q0[b] = r[0]; q1[b] = r[1]; q2[b] = r[2]; q3[b] = r[3];
                    } else {
                        const f3p* r = reinterpret_cast<const f3p*>(a.sc.wprims) + 3 * (size_t)kk[b];
                        r0[b] = r[0]; r1[b] = r[1]; r2[b] = r[2];
                    }
                }
#pragma unroll
                for (int b = 0; b < kPB; b++) {
                    if (b > 0 && !__any(has[b])) break;
                    if (has[b]) {
                        const int k = kk[b];
                        if (STATS) cn.prim_tests++;
                        float tt = 0.0f;
                        bool ok;
                        if (QUADS && __float_as_int(q0[b].w) != 0) {
                            // Quad::intersect under an upper bound returns the smaller t of its two halves whenever that is below
                            // the bound (each half accepts t < closest, the second sees the first's result): quad.h:56-121
                            const float eps_up = __uint_as_float(__float_as_uint(1e-8f) + 1u);
                            const float ta = mt_candidate(xyz(q0[b]), xyz(q1[b]), xyz(q2[b]), p.o, p.d, eps_up, t_lo);
                            const float tb = mt_candidate(xyz(q0[b]), xyz(q2[b]), xyz(q3[b]), p.o, p.d, eps_up, t_lo);
                            tt = min_raw(ta, tb);
                            ok = tt < __builtin_inff();
                        } else {
                            const f3 v0 = QUADS ? xyz(q0[b]) : mk3(r0[b].x, r0[b].y, r0[b].z), e1 = QUADS ? xyz(q1[b]) : mk3(r1[b].x, r1[b].y, r1[b].z),
                                     e2 = QUADS ? xyz(q2[b]) : mk3(r2[b].x, r2[b].y, r2[b].z);
                            ok = mt_hit(v0, e1, e2, p.o, p.d, 1e-8f, t_lo, tt);
                        }
                        if (ok) {
                            if (tt < closest_t) { closest_t = tt; slot_hit = k; }
                            else if (tt == closest_t && slot_hit >= 0) {           // the reference keeps the hit it visits first (scene.h:89-90)
                                if (CERT) slot_hit |= kTieFlag;
                                else if (a.sc.wref_slot[k] < a.sc.wref_slot[slot_hit]) slot_hit = k;
                            }
                        }
                    }
                }
                if (t_mask == 0u) phase = ((g_bits & 0xffu) != 0u || sp > 0) ? PH_NODE : after_walk();
            }
        } else {
            if (CERT && phase >= PH_VERIFY) {
                if (phase == PH_VERIFY && t_base == 0xffffffffu) {
                    // ONE fetch: the box of the hit triangle's leaf in the reference's tree.  Boxes are nested, so if the hit point
                    // Q = o + t* d lies inside the LEAF's box by eps on every face, it lies inside every ancestor's by at least
                    // as much - and eps = 2^-20 (|o_a| + big) is more than the reference's slab arithmetic can be off by on any box
                    // of the scene: t0' = fl(fl(lo - o) fl(1 / d)) is within 3 * 2^-24 |lo - o| / |d| of the true plane distance,
                    // Q_a' = fl(o_a + fl(t* d_a)) within 2 * 2^-24 (|o_a| + |t* d_a|) of Q_a, |lo_a - o_a| and |t* d_a| <= |o_a| + big.
                    // Then every entry distance comes out <= t*, every exit distance >= t*, and `!(min(exit, closest_t) < entry)`
                    // holds whatever closest_t >= t* the reference carries there.  A direction component below 2^-60 (1 / d near
                    // overflow) or a point within eps of a face goes to the chain of exact slab tests instead.
                    const float4 lo = a.sc.wcert[kWideCertStride * (size_t)slot_hit], hi = a.sc.wcert[kWideCertStride * (size_t)slot_hit + 1];
                    const f3 q = p.o + closest_t * p.d;
                    // eps from THIS box's own coordinates M_a = max(|lo_a|, |hi_a|) (round 3 took the scene's largest coordinate: one
                    // far-away primitive then sent every hit of the scene to the chain).  A point inside the box has |Q_a| <= M_a and
                    // |t* d_a| <= |o_a| + M_a, so the bounds above sum to <= 11 * 2^-24 (|o_a| + M_a) < eps; and the margin carries to
                    // every ancestor: a face of an ancestor at X lies |X - F| beyond the leaf's face F, its own arithmetic error
                    // 2^-22 (|o_a| + |X|) <= 2^-22 (|o_a| + |F| + |X - F|) stays below eps + |X - F|
                    const float ex = 9.5367431640625e-7f * (fabsf(p.o.x) + fmaxf(fabsf(lo.x), fabsf(hi.x))), ey = 9.5367431640625e-7f * (fabsf(p.o.y) + fmaxf(fabsf(lo.y), fabsf(hi.y))),
                                ez = 9.5367431640625e-7f * (fabsf(p.o.z) + fmaxf(fabsf(lo.z), fabsf(hi.z)));
                    const bool inside = q.x - lo.x >= ex && hi.x - q.x >= ex && q.y - lo.y >= ey && hi.y - q.y >= ey && q.z - lo.z >= ez && hi.z - q.z >= ez;
                    const bool finite_slopes = fabsf(p.d.x) >= 8.673617379884035e-19f && fabsf(p.d.y) >= 8.673617379884035e-19f && fabsf(p.d.z) >= 8.673617379884035e-19f;
                    if (inside && finite_slopes) phase = PH_SHADE;
                    else {
                        const uint32_t ref = __float_as_uint(lo.w);
                        g_base = ref >> 5; t_base = ref & 31u;
                        inv = mk3(rcp_rn(p.d.x), rcp_rn(p.d.y), rcp_rn(p.d.z));       // the reference's 1 / d for its slab tests
                        if (STATS) cn.cert_chain++;
                        if (t_base == 0u) phase = PH_EXACT;                         // (no list: never built that way; the count below must not wrap)
                    }
                } else if (phase == PH_VERIFY) {
                    const uint4 idx = a.sc.wanc[g_base];
                    g_base++; t_base--;
                    const uint32_t ni[4] = {idx.x, idx.y, idx.z, idx.w};
                    float4 n0[4], n1[4];
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        const uint32_t j = ni[c] == 0xffffffffu ? 0u : ni[c];      // padding repeats the root
                        n0[c] = a.sc.nodes[2 * (size_t)j]; n1[c] = a.sc.nodes[2 * (size_t)j + 1];
                    }
                    // from the leaf upwards: a box that holds Q with the margin settles all boxes above it (nested) - usually the
                    // leaf's parent or grandparent; below it every box has to pass the reference's own slab test
                    const f3 q = p.o + closest_t * p.d;
                    const float slopes = min3_raw(fabsf(p.d.x), fabsf(p.d.y), fabsf(p.d.z)) - 8.673617379884035e-19f;
                    bool proven = false, failed = false;
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        const float ex = 9.5367431640625e-7f * (fabsf(p.o.x) + fmaxf(fabsf(n0[c].x), fabsf(n1[c].x))), ey = 9.5367431640625e-7f * (fabsf(p.o.y) + fmaxf(fabsf(n0[c].y), fabsf(n1[c].y))),
                                    ez = 9.5367431640625e-7f * (fabsf(p.o.z) + fmaxf(fabsf(n0[c].z), fabsf(n1[c].z)));      // this box's own eps (see the one-fetch step)
                        const float mx = min3_raw(q.x - n0[c].x - ex, n1[c].x - q.x - ex, slopes);
                        const float my = min3_raw(q.y - n0[c].y - ey, n1[c].y - q.y - ey, q.z - n0[c].z - ez);
                        const bool holds = min3_raw(mx, my, n1[c].z - q.z - ez) >= 0.0f;
                        const bool passes = box_hit(n0[c], n1[c], p.o, inv, t_min, closest_t);
                        failed = failed || (!proven && !holds && !passes);
                        proven = proven || holds;
                    }
                    if (STATS) cn.node_visits += 4;
                    if (failed) phase = PH_EXACT;
                    else if (proven || t_base == 0u) phase = PH_SHADE;
                }
                if (phase == PH_EXACT) {                               // the reference's walk itself, for this ray only
                    float t_ref = 0.0f; int slot_ref = -1;
                    const bool h = intersect_lane<QUADS, STATS>(a.sc.nodes, a.sc.prims, a.sc.prim_stride, a.sc.n_nodes, true, p.o, p.d, t_min, FLT_MAX,
                                                                t_ref, slot_ref, cn);
                    closest_t = h ? t_ref : FLT_MAX;
                    slot_hit = h ? a.sc.wfast_of_ref[slot_ref] : -1;
                    phase = PH_SHADE;
                    if (STATS) cn.cert_fallback++;
                }
            }
            if (phase == PH_SHADE) {
                const bool more = shade_step<STATS, GUIDED, true, BATCH>(a.fp, a.tm, ms, a.sc.cdfs, p, slot_hit >= 0, closest_t, slot_hit, cn, slot);
                segs_left--;
                if (more && segs_left != 0) {                          // the next segment of this pixel
                    slot_hit = -1; closest_t = FLT_MAX; sp = 0; t_mask = 0u;
                    inv = mk3(wide_inv(p.d.x), wide_inv(p.d.y), wide_inv(p.d.z));
                    octinv = wide_octinv(inv);
                    g_base = 0u; g_bits = (1u << 8) | (1u << octinv);
                    phase = PH_NODE;
                    if (STATS) cn.rays++;
                    if (CERT && !origin_in_range()) phase = PH_EXACT;
                } else {
                    phase = PH_DONE;
                    if (end_visit(a, n_in, more, slot, p, segs_left)) {            // the lane goes on with the next queued pixel
                        slot_hit = -1; closest_t = FLT_MAX; sp = 0; t_mask = 0u;
                        inv = mk3(wide_inv(p.d.x), wide_inv(p.d.y), wide_inv(p.d.z));
                        octinv = wide_octinv(inv);
                        g_base = 0u; g_bits = (1u << 8) | (1u << octinv);
                        phase = PH_NODE;
                        if (STATS) cn.rays++;
                        if (CERT && !origin_in_range()) phase = PH_EXACT;
                    }
                }
            }
        }
        PTMI_TR(wt_c[wt_k] += (unsigned long long)(clock64() - wt_a);)
    }

    finish_launch<STATS, false>(a, false, slot, cn);      // every visit has banked its pixel and queued it if it goes on: counters only
#ifdef PTMI_TRACE_WAVES
    {
        const unsigned int lanes0 = (unsigned int)__popcll(__ballot(active));
        if ((threadIdx.x & 63) == 0) {
            const unsigned int i = atomicAdd(&g_wt_n, 1u);
            if (i < kWideTraceCap) {
                unsigned int* w = g_wt + 12 * (size_t)i;
                w[0] = (unsigned int)wt_t0; w[1] = (unsigned int)((unsigned long long)wall_clock64() - wt_t0);
                w[2] = wt_n[0]; w[3] = wt_l[0]; w[4] = wt_n[1]; w[5] = wt_l[1]; w[6] = wt_n[2]; w[7] = wt_l[2];
                w[8] = (unsigned int)(wt_c[0] >> 4); w[9] = (unsigned int)(wt_c[1] >> 4); w[10] = (unsigned int)(wt_c[2] >> 4); w[11] = lanes0 | (wt_half << 8);
            }
        }
    }
#endif
}
#ifndef PTMI_WIDE_WAVES
#define PTMI_WIDE_WAVES 6
#endif
template <bool STATS, bool GUIDED, bool BATCH, bool CERT, bool QUADS>
__global__ __launch_bounds__(kBlock, PTMI_WIDE_WAVES) __attribute__((amdgpu_num_sgpr(80))) void ptmi_bounce_wide(BounceArgs a) {
    bounce_wide_body<STATS, GUIDED, BATCH, CERT, QUADS>(a);
    publish_count(a);
}

#ifdef PTMI_TRACE_WAVES
// copies up to cap records (12 words each) of the wide walk's wave trace to out, returns how many there were, and clears
extern "C" long long ptmi_wide_trace_read(unsigned int* out, long long cap) {
    unsigned int n = 0;
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_wt_n), sizeof n) != hipSuccess) return -1;
    const unsigned int m = n < kWideTraceCap ? n : kWideTraceCap;
    const long long k = (long long)m < cap ? (long long)m : cap;
    if (k > 0 && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wt), (size_t)k * 12 * sizeof(unsigned int)) != hipSuccess) return -1;
    const unsigned int z = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_wt_n), &z, sizeof z) != hipSuccess) return -1;
    return (long long)n;
}
extern "C" int ptmi_trace_read(unsigned long long* out) {       // reads and clears the counters
    unsigned long long z[16] = {};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_trace), sizeof(z)) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_trace), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#endif

size_t bounce_lds_bytes_wide(const DeviceScene& sc) { return (size_t)sc.w_top * kWideNodeDwords * 4 + (size_t)sc.w_depth * kBlock * sizeof(uint2); }
size_t bounce_lds_bytes(const DeviceScene& sc) {
    if (sc.traversal == TRAVERSAL_WIDE || sc.traversal == TRAVERSAL_CERTIFIED) return bounce_lds_bytes_wide(sc);
    size_t b = 0;
    const bool geom = sc.lds_resident || sc.traversal == TRAVERSAL_SWEEP;
    if (geom) b += (size_t)(2 * sc.n_nodes + (sc.prim_stride + 3) * sc.n_prims) * sizeof(float4);
    if (sc.traversal == TRAVERSAL_STACK) b += (size_t)sc.stack_entries * kBlock * sizeof(int);
    return b;
}

// guided = sampling_mode != SAMPLING_BSDF and CDF records present; otherwise the lean BSDF instantiation runs
static bool is_guided(const BounceArgs& a) { return a.fp.sampling_mode != 0 && a.sc.cdfs != nullptr; }

// ---- kernel selection: f(kernel, dynamic LDS bytes) is called for the ONE instantiation that `a` selects -----------------
template <int MODE, bool G_, bool Q_, bool S_, typename F>
static void with_bounce_gb(const BounceArgs& a, size_t lds, F&& f) {
    switch ((is_guided(a) ? 2 : 0) | (a.fp.n_frames > 1 ? 1 : 0)) {
        case 0: f(ptmi_bounce<MODE, G_, Q_, S_, false, false>, lds); break;
        case 1: f(ptmi_bounce<MODE, G_, Q_, S_, false, true>, lds); break;
        case 2: f(ptmi_bounce<MODE, G_, Q_, S_, true, false>, lds); break;
        default: f(ptmi_bounce<MODE, G_, Q_, S_, true, true>, lds); break;
    }
}
template <bool G_, bool Q_, bool S_, bool PACKED, typename F>
static void with_phased_gb(const BounceArgs& a, size_t lds, F&& f) {
    if constexpr (PACKED && !Q_) {
        if (a.many_waves && !is_guided(a)) {
            if (a.fp.n_frames > 1) f(ptmi_bounce_packed_w8<S_, true>, lds); else f(ptmi_bounce_packed_w8<S_, false>, lds);
            return;
        }
    }
    switch ((is_guided(a) ? 2 : 0) | (a.fp.n_frames > 1 ? 1 : 0)) {
        case 0: f(ptmi_bounce_phased<G_, Q_, S_, false, PACKED, false>, lds); break;
        case 1: f(ptmi_bounce_phased<G_, Q_, S_, false, PACKED, true>, lds); break;
        case 2: f(ptmi_bounce_phased<G_, Q_, S_, true, PACKED, false>, lds); break;
        default: f(ptmi_bounce_phased<G_, Q_, S_, true, PACKED, true>, lds); break;
    }
}
// KIND: 0..2 = ptmi_bounce<TraversalMode>, 3 = ptmi_bounce_phased over the scene arrays, 4 = over the packed layout
template <int KIND, bool G_, typename F>
static void with_bounce_qs(const BounceArgs& a, size_t lds, F&& f) {
#define PTMI_QS(Q_, S_)                                                                               \
    do {                                                                                              \
        if constexpr (KIND == 4) with_phased_gb<false, Q_, S_, true>(a, lds, f);                      \
        else if constexpr (KIND == 3) with_phased_gb<G_, Q_, S_, false>(a, lds, f);                   \
        else with_bounce_gb<KIND, G_, Q_, S_>(a, lds, f);                                             \
    } while (0)
    switch ((a.sc.has_quads ? 2 : 0) | (a.stats ? 1 : 0)) {
        case 0: PTMI_QS(false, false); break;
        case 1: PTMI_QS(false, true); break;
        case 2: PTMI_QS(true, false); break;
        default: PTMI_QS(true, true); break;
    }
#undef PTMI_QS
}
template <typename F>
static void with_bounce_kernel(const BounceArgs& a, F&& f) {
    const DeviceScene& sc = a.sc;
    const size_t lds = bounce_lds_bytes(sc);
    switch (sc.traversal) {
        case TRAVERSAL_SWEEP: with_bounce_qs<TRAVERSAL_SWEEP, true>(a, lds, f); break;          // the sweep reads the scene through LDS
        case TRAVERSAL_LANE:
            if (sc.lds_resident) with_bounce_qs<TRAVERSAL_LANE, true>(a, lds, f); else with_bounce_qs<TRAVERSAL_LANE, false>(a, lds, f);
            break;
        case TRAVERSAL_PHASED:
            if (sc.lds_resident) with_bounce_qs<3, true>(a, lds, f); else with_bounce_qs<3, false>(a, lds, f);
            break;
        case TRAVERSAL_PACKED: with_bounce_qs<4, false>(a, (size_t)sc.n_top * 2 * sizeof(float4), f); break;
        case TRAVERSAL_WIDE:
        case TRAVERSAL_CERTIFIED: {
            const bool cert = sc.traversal == TRAVERSAL_CERTIFIED;
#define PTMI_WIDE(S_, G_, B_) do { if (sc.wqprims) { if (cert) f(ptmi_bounce_wide<S_, G_, B_, true, true>, lds); else f(ptmi_bounce_wide<S_, G_, B_, false, true>, lds); } \
                                   else { if (cert) f(ptmi_bounce_wide<S_, G_, B_, true, false>, lds); else f(ptmi_bounce_wide<S_, G_, B_, false, false>, lds); } } while (0)
            switch ((a.stats ? 4 : 0) | (is_guided(a) ? 2 : 0) | (a.fp.n_frames > 1 ? 1 : 0)) {
                case 0: PTMI_WIDE(false, false, false); break;
                case 1: PTMI_WIDE(false, false, true); break;
                case 2: PTMI_WIDE(false, true, false); break;
                case 3: PTMI_WIDE(false, true, true); break;
                case 4: PTMI_WIDE(true, false, false); break;
                case 5: PTMI_WIDE(true, false, true); break;
                case 6: PTMI_WIDE(true, true, false); break;
                default: PTMI_WIDE(true, true, true); break;
            }
#undef PTMI_WIDE
            break;
        }
        default:
            if (sc.lds_resident) with_bounce_qs<TRAVERSAL_STACK, true>(a, lds, f); else with_bounce_qs<TRAVERSAL_STACK, false>(a, lds, f);
            break;
    }
}

void launch_bounce(const DeviceScene& sc, const TileMap& tm, const PathState& st, const FrameParams& fp,
                   const int* queue_in, int n_in, const int* count_in, int* queue_out, int* count_out, int segments,
                   StatCounters* stats, bool many_waves, hipStream_t s, const CountPublish& pub, const LaunchSchedule& sched) {
    if (n_in <= 0) return;
    BounceArgs a{sc, tm, st, fp, queue_in, n_in, count_in, queue_out, count_out, segments, stats, many_waves ? 1 : 0,
                 pub.done_count, pub.next_count, pub.host_count, nullptr, nullptr, nullptr};
    dim3 grid((n_in + kBlock - 1) / kBlock);
    const bool wide = sc.traversal == TRAVERSAL_WIDE || sc.traversal == TRAVERSAL_CERTIFIED;
    if (wide && sched.max_waves > 0 && sched.cursor) {      // fewer lanes than queue entries: the lanes take the rest through the cursor
        a.cursor = sched.cursor;
        grid.x = std::min<unsigned int>(grid.x, (unsigned int)(sched.max_waves + kBlock / 64 - 1) / (kBlock / 64));
    }
    if (wide) { a.cost = sched.cost; a.cost_max = sched.cost_max; }
    with_bounce_kernel(a, [&](auto kernel, size_t lds) { hipLaunchKernelGGL(kernel, grid, dim3(kBlock), lds, s, a); });
}

// waves of the frame's bounce kernel that the device holds at once (0: unknown)
int bounce_resident_waves(const DeviceScene& sc, const FrameParams& fp, bool stats, int n_cus) {
    BounceArgs a{sc, TileMap(), PathState(), fp, nullptr, 0, nullptr, nullptr, nullptr, 0, stats ? reinterpret_cast<StatCounters*>(1) : nullptr, 0,
                 nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int blocks = 0;
    with_bounce_kernel(a, [&](auto kernel, size_t lds) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, kernel, kBlock, lds) != hipSuccess) blocks = 0;
    });
    return blocks * (kBlock / 64) * n_cus;
}

// ---- launch order by cost (device_scene.h: launch_order_by_cost) ---------------------------------------------------------------
// class 0 = the heaviest: (255 - floor(256 cost / (cost_max + 1))) >> shift, so ascending classes are descending costs
__device__ __forceinline__ int cost_class(unsigned int c, unsigned int cmax, int shift) {
    const unsigned long long k = ((unsigned long long)min(c, cmax) << 8) / ((unsigned long long)cmax + 1ull);
    return (255 - (int)k) >> shift;
}
__global__ __launch_bounds__(kBlock) void ptmi_cost_histogram(const int* __restrict__ queue_in, int n, const unsigned int* __restrict__ cost,
                                                              const unsigned int* __restrict__ cost_max, int* __restrict__ hist, int shift) {
    __shared__ int h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const unsigned int cmax = *cost_max;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock)
        atomicAdd(&h[cost_class(cost[queue_in ? queue_in[i] : i], cmax, shift)], 1);
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}
// one workgroup: hist[256 + c] = first position of class c
__global__ __launch_bounds__(kBlock) void ptmi_cost_offsets(int* __restrict__ hist) {
    __shared__ int h[256];
    h[threadIdx.x] = hist[threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 0) { int acc = 0; for (int c = 0; c < 256; c++) { const int v = h[c]; h[c] = acc; acc += v; } }
    __syncthreads();
    hist[256 + threadIdx.x] = h[threadIdx.x];
}
// every workgroup takes 256 consecutive entries, counts its classes, reserves one range per class and places its entries there
// in their order (so neighbours of one class stay neighbours)
__global__ __launch_bounds__(kBlock) void ptmi_cost_scatter(const int* __restrict__ queue_in, int n, const unsigned int* __restrict__ cost,
                                                            const unsigned int* __restrict__ cost_max, int* __restrict__ hist, int* __restrict__ queue, int shift) {
    __shared__ int h[256], base[256];
    __shared__ unsigned char cls[kBlock];
    h[threadIdx.x] = 0;
    const int i = blockIdx.x * kBlock + threadIdx.x;
    const int entry = i < n ? (queue_in ? queue_in[i] : i) : 0;
    const int c = i < n ? cost_class(cost[entry], *cost_max, shift) : 255;
    cls[threadIdx.x] = (unsigned char)c;
    __syncthreads();
    if (i < n) atomicAdd(&h[c], 1);
    int local = 0;                                     // entries of my class before me in this workgroup
    for (int j = 0; j < (int)threadIdx.x; j++) local += cls[j] == c ? 1 : 0;
    __syncthreads();
    if (h[threadIdx.x]) base[threadIdx.x] = atomicAdd(&hist[256 + threadIdx.x], h[threadIdx.x]);
    __syncthreads();
    if (i < n) queue[base[c] + local] = entry;
}
void launch_order_by_cost(const int* queue_in, int n, const unsigned int* cost, const unsigned int* cost_max, int* hist, int* queue, int classes, hipStream_t s) {
    if (n <= 0) return;
    int shift = 0;
    while ((256 >> shift) > classes && shift < 7) shift++;
    (void)hipMemsetAsync(hist, 0, 512 * sizeof(int), s);
    const int blocks = (n + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(ptmi_cost_histogram, dim3(std::min(blocks, 1024)), dim3(kBlock), 0, s, queue_in, n, cost, cost_max, hist, shift);
    hipLaunchKernelGGL(ptmi_cost_offsets, dim3(1), dim3(kBlock), 0, s, hist);
    hipLaunchKernelGGL(ptmi_cost_scatter, dim3(blocks), dim3(kBlock), 0, s, queue_in, n, cost, cost_max, hist, queue, shift);
}

void launch_render_init(const TileMap& tm, const PathState& st, const uint32_t* d_jump, uint64_t seed_base, hipStream_t s) {
    const int n = tm.local_rows * tm.width;
    if (n <= 0) return;
    hipLaunchKernelGGL(ptmi_render_init, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, s, tm, st, d_jump,
                       (unsigned long long)seed_base);
}

void launch_frame_begin(const TileMap& tm, const PathState& st, const FrameParams& fp, hipStream_t s) {
    const int n = tm.local_rows * tm.width;
    if (n <= 0) return;
    hipLaunchKernelGGL(ptmi_frame_begin, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, s, tm, st, fp);
}

// ---------------------------------------------------------------------------------------------
// resolve: color /= spp; Reinhard; gamma 1/2.2; 8-bit (integrator.h:393-407)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void ptmi_resolve(TileMap tm, PathState st, int spp, unsigned char* __restrict__ rgb8,
                                                       float* __restrict__ radiance, const float4* __restrict__ color_src) {
    const int n = tm.local_rows * tm.width;
    const int slot = blockIdx.x * kBlock + threadIdx.x;
    if (slot >= n) return;
    const float4 D = color_src ? color_src[slot] : st.D[slot];
    int ox, olr;
    slot_to_local(tm, slot, ox, olr);
    const size_t out = (size_t)olr * (size_t)tm.width + (size_t)ox;      // images are local-row-major whatever the slot order
    const float k = rcp_rn((float)spp);                    // Vector::operator/=(T): T k = 1.0 / t (vector.h:90-94)
    const float c[3] = {D.x * k, D.y * k, D.z * k};
    const float gamma = 1.0f / 2.2f;
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
        if (radiance) radiance[out * 3 + ch] = c[ch];
        if (rgb8) {
            const float tm_ = c[ch] / (c[ch] + 1.0f);       // color / (color + 1), component-wise true division
            const float g = ptmi_powf(tm_, gamma);
            rgb8[out * 3 + ch] = (unsigned char)(255.99f * fminf(g, 1.0f));
        }
    }
}

void launch_resolve(const TileMap& tm, const PathState& st, int spp, unsigned char* rgb8, float* radiance, hipStream_t s,
                    const float4* color_src) {
    const int n = tm.local_rows * tm.width;
    if (n <= 0) return;
    hipLaunchKernelGGL(ptmi_resolve, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, s, tm, st, spp, rgb8, radiance, color_src);
}

// The proof of the certified walk for ONE hit (bounce_wide_body, VERIFY, in straight-line form): the hit point inside the hit leaf's
// box of the reference's tree by eps, else the exact slab tests of the leaf's ancestors, leaf first, up to the first box that holds
// the point with the margin.  false: a box of the chain failed - the reference's own walk has to decide this ray.
__device__ __forceinline__ bool certified_proof(const DeviceScene& sc, f3 o, f3 d, float t_min, float closest_t, int slot_hit) {
    const float4 lo = sc.wcert[kWideCertStride * (size_t)slot_hit], hi = sc.wcert[kWideCertStride * (size_t)slot_hit + 1];
    const f3 q = o + closest_t * d;
    const float kEps = 9.5367431640625e-7f, kSlope = 8.673617379884035e-19f;
    const bool slopes = fabsf(d.x) >= kSlope && fabsf(d.y) >= kSlope && fabsf(d.z) >= kSlope;
    const float ex = kEps * (fabsf(o.x) + fmaxf(fabsf(lo.x), fabsf(hi.x))), ey = kEps * (fabsf(o.y) + fmaxf(fabsf(lo.y), fabsf(hi.y))),
                ez = kEps * (fabsf(o.z) + fmaxf(fabsf(lo.z), fabsf(hi.z)));
    const bool inside = q.x - lo.x >= ex && hi.x - q.x >= ex && q.y - lo.y >= ey && hi.y - q.y >= ey && q.z - lo.z >= ez && hi.z - q.z >= ez;
    if (inside && slopes) return true;
    const f3 rinv = mk3(rcp_rn(d.x), rcp_rn(d.y), rcp_rn(d.z));            // the reference's 1 / d for its slab tests
    const uint32_t ref = __float_as_uint(lo.w);
    uint32_t off = ref >> 5;
    bool proven = false, failed = false;
    for (int left = (int)(ref & 31u); left > 0 && !proven && !failed; left--, off++) {
        const uint4 idx = sc.wanc[off];
        const uint32_t ni[4] = {idx.x, idx.y, idx.z, idx.w};
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const uint32_t j = ni[c] == 0xffffffffu ? 0u : ni[c];              // padding repeats the root
            const float4 n0 = sc.nodes[2 * (size_t)j], n1 = sc.nodes[2 * (size_t)j + 1];
            const float nx = kEps * (fabsf(o.x) + fmaxf(fabsf(n0.x), fabsf(n1.x))), ny = kEps * (fabsf(o.y) + fmaxf(fabsf(n0.y), fabsf(n1.y))),
                        nz = kEps * (fabsf(o.z) + fmaxf(fabsf(n0.z), fabsf(n1.z)));
            const bool holds = slopes && q.x - n0.x >= nx && n1.x - q.x >= nx && q.y - n0.y >= ny && n1.y - q.y >= ny && q.z - n0.z >= nz && n1.z - q.z >= nz;
            const bool passes = box_hit(n0, n1, o, rinv, t_min, closest_t);
            failed = failed || (!proven && !holds && !passes);
            proven = proven || holds;
        }
    }
    return !failed;
}

// ---------------------------------------------------------------------------------------------
// The certified closest hit of ONE ray, lane by lane (no phases): the walk and the proof of bounce_wide_body<..., CERT> in
// straight-line form, for callers that trace a ray at a time (the Radiosity view).  Returns the REFERENCE's hit: its leaf-order
// slot in ref_slot, so that the caller indexes the reference's per-primitive arrays.  stack: this lane's column of w_depth
// 8-byte entries in LDS (entry e at stack[e * kBlock]).
// ---------------------------------------------------------------------------------------------
template <bool QUADS>
__device__ __forceinline__ bool certified_closest_hit(const DeviceScene& sc, uint2* stack, f3 o, f3 d, float t_min, float& t_hit, int& ref_slot) {
    LaneCounters cn = {0, 0, 0, 0, 0, 0, 0};
    auto reference_walk = [&]() { return intersect_lane<QUADS, false>(sc.nodes, sc.prims, sc.prim_stride, sc.n_nodes, true, o, d, t_min, FLT_MAX, t_hit, ref_slot, cn); };
    if (fmaxf(fabsf(o.x), fmaxf(fabsf(o.y), fabsf(o.z))) > sc.w_guard) return reference_walk();      // the boxes are not padded for this origin
    const f3 inv = mk3(wide_inv(d.x), wide_inv(d.y), wide_inv(d.z));
    const uint32_t octinv = wide_octinv(inv);
    const float t_lo = mt_t_lo(t_min);
    float closest_t = FLT_MAX;
    int slot_hit = -1, sp = 0;
    bool tie = false;
    uint32_t g_base = 0u, g_bits = (1u << 8) | (1u << octinv);
    while (true) {
        if ((g_bits & 0xffu) == 0u) {
            if (sp == 0) break;
            sp--; const uint2 e = stack[sp * kBlock]; g_base = e.x; g_bits = e.y;
        }
        const int bit = 31 - __clz((int)(g_bits & 0xffu));
        g_bits ^= 1u << bit;
        const uint32_t child = (uint32_t)bit ^ octinv;
        const uint32_t ni = g_base + (uint32_t)__popc((g_bits >> 8) & ((1u << child) - 1u));
        if (g_bits & 0xffu) { stack[sp * kBlock] = make_uint2(g_base, g_bits); sp++; }
        const uint4* q = sc.wnodes + 8 * (size_t)ni;
        const WideStep st = wide_node_test(q[0], q[1], q[2], q[3], q[4], q[5], q[6], o, inv, octinv, t_min, closest_t);
        uint32_t tris = st.tris;
        while (tris) {
            const int k = (int)st.tri_base + __ffs((int)tris) - 1;
            tris &= tris - 1u;
            float tt = 0.0f;
            bool ok;
            if (QUADS && __float_as_int(sc.wqprims[4 * (size_t)k].w) != 0) {               // a quad: the smaller t of its two halves
                const float4* r = sc.wqprims + 4 * (size_t)k;
                const float eps_up = __uint_as_float(__float_as_uint(1e-8f) + 1u);
                tt = min_raw(mt_candidate(xyz(r[0]), xyz(r[1]), xyz(r[2]), o, d, eps_up, t_lo), mt_candidate(xyz(r[0]), xyz(r[2]), xyz(r[3]), o, d, eps_up, t_lo));
                ok = tt < __builtin_inff();
            } else if (QUADS) {
                const float4* r = sc.wqprims + 4 * (size_t)k;
                ok = mt_hit(xyz(r[0]), xyz(r[1]), xyz(r[2]), o, d, 1e-8f, t_lo, tt);
            } else {
                const f3p* r = reinterpret_cast<const f3p*>(sc.wprims) + 3 * (size_t)k;
                const f3p v0 = r[0], e1 = r[1], e2 = r[2];
                ok = mt_hit(mk3(v0.x, v0.y, v0.z), mk3(e1.x, e1.y, e1.z), mk3(e2.x, e2.y, e2.z), o, d, 1e-8f, t_lo, tt);
            }
            if (ok) {
                if (tt < closest_t) { closest_t = tt; slot_hit = k; tie = false; }
                else if (tt == closest_t && slot_hit >= 0) tie = true;             // the reference keeps the hit it visits first: let it decide
            }
        }
        g_base = st.child_base; g_bits = (st.imask << 8) | st.inner;
    }
    if (slot_hit < 0) return false;                        // the reference can only accept triangles this walk would have found
    if (tie || !certified_proof(sc, o, d, t_min, closest_t, slot_hit)) return reference_walk();
    t_hit = closest_t;
    ref_slot = sc.wref_slot[slot_hit];
    return true;
}

// ---------------------------------------------------------------------------------------------
// render_radiosity (integrator.h:460-504): a visualisation pass, one thread per pixel, not performance-critical
// MODE TRAVERSAL_CERTIFIED: the first hit through the certified walk (scenes above the sweep's 64 primitives: the 8-wide tree + the
// proof per hit, else the reference's walk - the reference's hit for every ray); TRAVERSAL_LANE / TRAVERSAL_STACK: the reference's walk
// ---------------------------------------------------------------------------------------------
template <int MODE, bool HAS_QUADS>
__global__ __launch_bounds__(kBlock) void ptmi_render_radiosity(DeviceScene sc, TileMap tm, PathState st, FrameParams fp,
                                                                unsigned char* __restrict__ rgb8, float* __restrict__ radiance) {
    extern __shared__ float4 smem[];
    int* stack = reinterpret_cast<int*>(smem) + threadIdx.x;
    const int n = tm.local_rows * tm.width;
    const int slot = blockIdx.x * kBlock + threadIdx.x;
    const bool live = slot < n;
    int x = 0, y = 0;
    Rng rng = {0, 0, 0, 0, 0, 0};
    if (live) {
        global_pixel(tm, slot, x, y);
        const uint4 e = st.E[slot]; const uint2 f = st.F[slot];
        rng = Rng{e.x, e.y, e.z, e.w, f.x, f.y};                                      // curandState local_rng = rand_state[pixel_index]
    }
    f3 color = mk3(0.0f, 0.0f, 0.0f);
    LaneCounters cn = {0, 0, 0, 0, 0, 0, 0};
    for (int s = 0; s < fp.spp; s++) {
        f3 o = mk3(0, 0, 0), d = mk3(0, 0, 1);
        if (live) camera_ray(fp, tm, x, y, rng, o, d);
        float t = 0.0f; int k = -1;
        bool hit;
        if constexpr (MODE == TRAVERSAL_CERTIFIED) hit = live && certified_closest_hit<HAS_QUADS>(sc, reinterpret_cast<uint2*>(smem) + threadIdx.x, o, d, 1e-4f, t, k);
        else hit = scene_intersect<MODE, HAS_QUADS, false>(sc.nodes, sc.prims, sc.prim_stride, sc.n_nodes, stack, live, o, d, 1e-4f, FLT_MAX, t, k, cn);
        if (live && hit) {
            color = color + xyz(sc.mats[3 * k + 2]);                                  // color += si.Le
            color = color + (sc.radiosity ? xyz(sc.radiosity[k]) : mk3(0.0f, 0.0f, 0.0f));   // color += prim->getRadiosity()
        }
    }
    if (!live) return;
    const float kk = rcp_rn((float)fp.spp);
    const float c[3] = {color.x * kk, color.y * kk, color.z * kk};
    int ox, olr;
    slot_to_local(tm, slot, ox, olr);
    const size_t out = (size_t)olr * (size_t)tm.width + (size_t)ox;
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
        if (radiance) radiance[out * 3 + ch] = c[ch];
        if (rgb8) rgb8[out * 3 + ch] = (unsigned char)(255.99f * sqrt_rn(fminf(c[ch], 1.0f)));
    }
    st.E[slot] = make_uint4(rng.v0, rng.v1, rng.v2, rng.v3);                          // rand_state[pixel_index] = local_rng
    st.F[slot] = make_uint2(rng.v4, rng.d);
}

void launch_render_radiosity(const DeviceScene& sc, const TileMap& tm, const PathState& st, const FrameParams& fp,
                             unsigned char* rgb8, float* radiance, hipStream_t s) {
    const int n = tm.local_rows * tm.width;
    if (n <= 0) return;
    const dim3 grid((n + kBlock - 1) / kBlock), block(kBlock);
    const bool deep = sc.traversal == TRAVERSAL_STACK;                                // per-lane walk from global memory; stack only for deep trees
    const bool cert = sc.traversal == TRAVERSAL_CERTIFIED && sc.wnodes && sc.wcert && sc.wanc && sc.wref_slot && (!sc.has_quads || sc.wqprims);
    const size_t lds = cert ? (size_t)sc.w_depth * kBlock * sizeof(uint2) : deep ? (size_t)sc.stack_entries * kBlock * sizeof(int) : 0;
#define PTMI_RAD(M_, Q_) hipLaunchKernelGGL((ptmi_render_radiosity<M_, Q_>), grid, block, lds, s, sc, tm, st, fp, rgb8, radiance)
    if (cert) { if (sc.has_quads) PTMI_RAD(TRAVERSAL_CERTIFIED, true); else PTMI_RAD(TRAVERSAL_CERTIFIED, false); }
    else if (deep) { if (sc.has_quads) PTMI_RAD(TRAVERSAL_STACK, true); else PTMI_RAD(TRAVERSAL_STACK, false); }
    else { if (sc.has_quads) PTMI_RAD(TRAVERSAL_LANE, true); else PTMI_RAD(TRAVERSAL_LANE, false); }
#undef PTMI_RAD
}

// ---------------------------------------------------------------------------------------------
// test hooks
// ---------------------------------------------------------------------------------------------
template <int MODE, bool HAS_QUADS>
__global__ __launch_bounds__(kBlock) void ptmi_debug_intersect_k(DeviceScene sc, int n, const float* o, const float* d, float t_min,
                                                                 float t_max, int* hit, int* prim, float* t_out, float* p_out, float* n_out) {
    extern __shared__ float4 smem[];
    int* stack = reinterpret_cast<int*>(smem) + threadIdx.x;
    const int i = blockIdx.x * kBlock + threadIdx.x;
    const bool live = i < n;
    const int j = live ? i : 0;
    const f3 ro = mk3(o[3 * j], o[3 * j + 1], o[3 * j + 2]), rd = mk3(d[3 * j], d[3 * j + 1], d[3 * j + 2]);
    LaneCounters cn = {0, 0, 0, 0, 0, 0, 0};
    float t = 0.0f; int k = -1;
    const bool h = scene_intersect<MODE, HAS_QUADS, false>(sc.nodes, sc.prims, sc.prim_stride, sc.n_nodes, stack, live, ro, rd, t_min, t_max, t, k, cn);
    if (!live) return;
    hit[i] = h ? 1 : 0;
    prim[i] = h ? __float_as_int(sc.mats[3 * k].w) : -1;
    t_out[i] = h ? t : 0.0f;
    const f3 p = h ? ro + t * rd : mk3(0, 0, 0);
    const f3 nn = h ? xyz(sc.mats[3 * k]) : mk3(0, 0, 0);
    p_out[3 * i] = p.x; p_out[3 * i + 1] = p.y; p_out[3 * i + 2] = p.z;
    n_out[3 * i] = nn.x; n_out[3 * i + 1] = nn.y; n_out[3 * i + 2] = nn.z;
}

void launch_debug_intersect(const DeviceScene& sc, int n, const float* o, const float* d, float t_min, float t_max,
                            int* hit, int* prim, float* t, float* p, float* nrm, hipStream_t s) {
    if (n <= 0) return;
    const size_t lds = (size_t)sc.stack_entries * kBlock * sizeof(int);
    const dim3 grid((n + kBlock - 1) / kBlock), block(kBlock);
#define PTMI_DBG(M_, Q_) hipLaunchKernelGGL((ptmi_debug_intersect_k<M_, Q_>), grid, block, lds, s, sc, n, o, d, t_min, t_max, hit, prim, t, p, nrm)
    const int walk = sc.traversal == TRAVERSAL_PHASED || sc.traversal == TRAVERSAL_PACKED || sc.traversal == TRAVERSAL_CERTIFIED ? TRAVERSAL_LANE : sc.traversal;   // the phased kernels walk like LANE
    switch (walk * 2 + (sc.has_quads ? 1 : 0)) {
        case 0: PTMI_DBG(TRAVERSAL_SWEEP, false); break;
        case 1: PTMI_DBG(TRAVERSAL_SWEEP, true); break;
        case 2: PTMI_DBG(TRAVERSAL_LANE, false); break;
        case 3: PTMI_DBG(TRAVERSAL_LANE, true); break;
        case 4: PTMI_DBG(TRAVERSAL_STACK, false); break;
        default: PTMI_DBG(TRAVERSAL_STACK, true); break;
    }
#undef PTMI_DBG
}

// Closest hit through the fast tree for n rays (test hook): the walk of ptmi_bounce_wide, lane by lane, without the phases
__global__ __launch_bounds__(kBlock) void ptmi_debug_intersect_wide_k(DeviceScene sc, int n, const float* o, const float* d, float t_min,
                                                                      float t_max, int* hit, int* prim, float* t_out,
                                                                      unsigned long long* counts /* [0] node visits [1] triangle tests */) {
    extern __shared__ float4 smem[];
    uint2* stack = reinterpret_cast<uint2*>(smem) + threadIdx.x;
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const f3 ro = mk3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), rd = mk3(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
    const f3 inv = mk3(wide_inv(rd.x), wide_inv(rd.y), wide_inv(rd.z));
    const uint32_t octinv = wide_octinv(inv);
    const float t_lo = mt_t_lo(t_min);
    float closest_t = t_max;
    int slot_hit = -1, sp = 0;
    uint32_t g_base = 0u, g_bits = (1u << 8) | (1u << octinv);
    unsigned int nv = 0, pt = 0;
    while (true) {
        if ((g_bits & 0xffu) == 0u) {
            if (sp == 0) break;
            sp--; const uint2 e = stack[sp * kBlock]; g_base = e.x; g_bits = e.y;
        }
        const int bit = 31 - __clz((int)(g_bits & 0xffu));
        g_bits ^= 1u << bit;
        const uint32_t child = (uint32_t)bit ^ octinv;
        const uint32_t ni = g_base + (uint32_t)__popc((g_bits >> 8) & ((1u << child) - 1u));
        if (g_bits & 0xffu) { stack[sp * kBlock] = make_uint2(g_base, g_bits); sp++; }
        const uint4* q = sc.wnodes + 8 * (size_t)ni;
        nv++;
        const WideStep st = wide_node_test(q[0], q[1], q[2], q[3], q[4], q[5], q[6], ro, inv, octinv, t_min, closest_t);
        uint32_t tris = st.tris;
        while (tris) {
            const int k = (int)st.tri_base + __ffs((int)tris) - 1;
            tris &= tris - 1u;
            pt++;
            const f3p* r = reinterpret_cast<const f3p*>(sc.wprims) + 3 * (size_t)k;
            const f3p v0 = r[0], e1 = r[1], e2 = r[2];
            float tt = 0.0f;
            bool ok;
            if (sc.wqprims && __float_as_int(sc.wqprims[4 * (size_t)k].w) != 0) {           // a quad: the smaller t of its two halves
                const float4* q = sc.wqprims + 4 * (size_t)k;
                const float eps_up = __uint_as_float(__float_as_uint(1e-8f) + 1u);
                tt = min_raw(mt_candidate(xyz(q[0]), xyz(q[1]), xyz(q[2]), ro, rd, eps_up, t_lo), mt_candidate(xyz(q[0]), xyz(q[2]), xyz(q[3]), ro, rd, eps_up, t_lo));
                ok = tt < __builtin_inff();
            } else ok = mt_hit(mk3(v0.x, v0.y, v0.z), mk3(e1.x, e1.y, e1.z), mk3(e2.x, e2.y, e2.z), ro, rd, 1e-8f, t_lo, tt);
            if (ok) {
                if (tt < closest_t) { closest_t = tt; slot_hit = k; }
                else if (tt == closest_t && slot_hit >= 0 && sc.wref_slot[k] < sc.wref_slot[slot_hit]) slot_hit = k;
            }
        }
        g_base = st.child_base; g_bits = (st.imask << 8) | st.inner;
    }
    hit[i] = slot_hit >= 0 ? 1 : 0;
    prim[i] = slot_hit >= 0 ? sc.wload_index[slot_hit] : -1;
    t_out[i] = slot_hit >= 0 ? closest_t : 0.0f;
    if (counts) { atomicAdd(&counts[0], (unsigned long long)nv); atomicAdd(&counts[1], (unsigned long long)pt); }
}
void launch_debug_intersect_wide(const DeviceScene& sc, int n, const float* o, const float* d, float t_min, float t_max,
                                 int* hit, int* prim, float* t, unsigned long long* counts, hipStream_t s) {
    if (n <= 0) return;
    const size_t lds = (size_t)sc.w_depth * kBlock * sizeof(uint2);
    hipLaunchKernelGGL(ptmi_debug_intersect_wide_k, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), lds, s, sc, n, o, d, t_min, t_max, hit, prim, t, counts);
}

__global__ void ptmi_debug_rng_k(const uint32_t* __restrict__ jump, unsigned long long seed_base, int n_pixels,
                                 const int* pixels, int count, float* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pixels) return;
    const unsigned int pix = (unsigned int)pixels[i];
    const unsigned long long seed = seed_base + (unsigned long long)pix;
    const uint32_t s0 = ((uint32_t)seed) ^ 0xaad26b49u, s1 = (uint32_t)(seed >> 32) ^ 0xf7dcefddu;
    const uint32_t t0 = 1099087573u * s0, t1 = 2591861531u * s1;
    uint32_t v[5] = {123456789u + t0, 362436069u ^ t0, 521288629u + t1, 88675123u ^ t1, 5783321u + t0};
    for (int k = 0; k < 32; k++) {
        if (!((pix >> k) & 1u)) continue;
        uint32_t r[5] = {0, 0, 0, 0, 0};
        for (int w = 0; w < 5; w++)
            for (int b = 0; b < 32; b++)
                if ((v[w] >> b) & 1u) for (int c = 0; c < 5; c++) r[c] ^= jump[(k * 160 + w * 32 + b) * 5 + c];
        for (int w = 0; w < 5; w++) v[w] = r[w];
    }
    Rng rng = {v[0], v[1], v[2], v[3], v[4], 6615241u + t1 + t0};
    for (int c = 0; c < count; c++) out[(size_t)i * count + c] = rng_uniform(rng);
}

void launch_debug_rng(const uint32_t* d_jump, uint64_t seed_base, int n_pixels, const int* pixels, int count, float* out, hipStream_t s) {
    if (n_pixels <= 0) return;
    hipLaunchKernelGGL(ptmi_debug_rng_k, dim3((n_pixels + 63) / 64), dim3(64), 0, s, d_jump, (unsigned long long)seed_base,
                       n_pixels, pixels, count, out);
}

// exhaustive check of rcp_exact_normal against the IEEE quotient over a range of bit patterns
__global__ void ptmi_debug_rcp_k(unsigned int first, unsigned long long count, unsigned long long* out /* [0]=mismatches [1]=first bad bits+1 */) {
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    unsigned long long bad = 0, first_bad = ~0ull;
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < count; i += stride) {
        const unsigned int bits = first + (unsigned int)i;
        const float a = __uint_as_float(bits);
        const float want = 1.0f / a, got = rcp_exact_normal(a);
        if (__float_as_uint(want) != __float_as_uint(got) && !(want != want && got != got)) { bad++; if (first_bad == ~0ull) first_bad = bits; }
    }
    if (bad) { atomicAdd(&out[0], bad); atomicMin(&out[1], first_bad); }
}
void launch_debug_rcp(unsigned int first, unsigned long long count, unsigned long long* d_out, hipStream_t s) {
    hipLaunchKernelGGL(ptmi_debug_rcp_k, dim3(4096), dim3(256), 0, s, first, count, d_out);
}

__global__ void ptmi_debug_cosine_k(int n, const float* normals, const float* u, const float* v, float* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const f3 r = cosine_hemisphere(mk3(normals[3 * i], normals[3 * i + 1], normals[3 * i + 2]), u[i], v[i]);
    out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z;
}

void launch_debug_cosine(int n, const float* normals, const float* u, const float* v, float* out, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(ptmi_debug_cosine_k, dim3((n + 255) / 256), dim3(256), 0, s, n, normals, u, v, out);
}

}  // namespace ptmi
