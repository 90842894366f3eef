// radiosity.hip — gfx950 kernels of the radiosity pre-pass: RadiosityState::runSolver
// (application_state.h:688-777) and its kernels (form_factors.h:71-467, grid_filter.h:35-312).
//
// Compile with -ffp-contract=off (see include/ptmi_math.h, pt_vec.h): every output except the num_iterations == 0
// radiosity grid (order-dependent float atomics in the reference itself) is bit-identical to oracle/ptmi_oracle.c.
//
// Kernels
//   ptmi_form_factors       one WORKGROUP per receiver i (the reference: one thread per (i, j) pair).  The workgroup owns
//                           row i of the form-factor matrix and primitive i's two directional grids, so the reference's
//                           global atomics become LDS adds and one plain store per cell.  Pairs are culled first
//                           (form_factors.h:234-256: ~2/3 of all pairs in a closed scene) and the survivors compacted
//                           through an LDS queue, so the Monte-Carlo loop runs on dense waves.  The reference keeps n^2
//                           curandStates (48 B each) that nothing reads after the kernel; here a pair's XORWOW stream
//                           is derived where it is used, for surviving pairs only.
//   ptmi_radiosity_iterate  radiosity_iteration_kernel (form_factors.h:441-465) with the race removed (Jacobi)
//   ptmi_radiosity_grid     update_radiosity_grid (form_factors.h:405-439) + the optional 5x5 filter (grid_filter.h);
//                           one workgroup per primitive, one thread per grid cell, contributions added in ascending j
#include "pt_device.h"

namespace ptmi {

namespace {

constexpr int kQueueCap = 2 * kBlock;

struct Geom { f3 v0, v1, v2, v3; int type; float area, ratio; f3 normal, centroid; };

__device__ __forceinline__ Geom load_geom(const float4* __restrict__ geo, int p) {
    const float4 a = geo[6 * p], b = geo[6 * p + 1], c = geo[6 * p + 2], d = geo[6 * p + 3], e = geo[6 * p + 4], f = geo[6 * p + 5];
    Geom g;
    g.v0 = xyz(a); g.v1 = xyz(b); g.v2 = xyz(c); g.v3 = xyz(d);
    g.type = __float_as_int(a.w); g.area = b.w; g.ratio = c.w;
    g.normal = xyz(e); g.centroid = xyz(f);
    return g;
}

// primitive.h:153-157
__device__ __forceinline__ f3 bary_point(f3 a, f3 b, f3 c, float r1, float r2) {
    const float sqrt_r1 = sqrt_rn(r1);
    const float u = 1.0f - sqrt_r1;
    const float v = sqrt_r1 * (1.0f - r2);
    const float w = sqrt_r1 * r2;
    return u * a + v * b + w * c;
}
// Primitive::sampleUniform (primitive.h:150-191); the quad's area ratio comes precomputed from the host
template <bool HAS_QUADS>
__device__ __forceinline__ f3 sample_uniform(const Geom& g, float r1, float r2) {
    if (!HAS_QUADS || g.type == 0) return bary_point(g.v0, g.v1, g.v2, r1, r2);      // triangle-only scenes: v3 / ratio / type stay out of registers
    if (r1 < g.ratio) return bary_point(g.v0, g.v1, g.v3, r1 / g.ratio, r2);                     // (v00, v10, v01)
    return bary_point(g.v1, g.v2, g.v3, (r1 - g.ratio) / (1.0f - g.ratio), r2);                  // (v10, v11, v01)
}

// direction_to_grid_indices_local (form_factors.h:107-130): theta over [0, pi] -> 16 rows, phi over [0, 2 pi) -> 16 columns
__device__ __forceinline__ int direction_to_grid_index_local(f3 world_dir, f3 normal) {
    f3 tangent, bitangent;
    build_frame(normal, tangent, bitangent);
    const float lx = dot(world_dir, tangent), ly = dot(world_dir, bitangent), lz = dot(world_dir, normal);
    const float r = sqrt_rn(lx * lx + ly * ly + lz * lz);
    const float theta = (r > 0.0f) ? ptmi_acosf(fminf(lz / r, 1.0f)) : 0.0f;
    float phi = ptmi_atan2f(ly, lx);
    if (phi < 0.0f) phi = (float)((double)phi + (double)2.0f * PTMI_PI_D);
    int grid_theta = (int)fminf((float)(((double)theta / PTMI_PI_D) * kGridRes), (float)(kGridRes - 1));
    int grid_phi = (int)fminf((float)(((double)phi / ((double)2.0f * PTMI_PI_D)) * kGridRes), (float)(kGridRes - 1));
    grid_theta = max(0, min(grid_theta, kGridRes - 1));
    grid_phi = max(0, min(grid_phi, kGridRes - 1));
    return grid_theta * kGridRes + grid_phi;
}

// Primitive::intersect(r, 1e-5f, max_dist) as a yes/no question: triangle.h:82 accepts t <= t_max, quad.h:78,110
// only t < t_max (closest_t starts at t_max)
template <bool HAS_QUADS>
__device__ __forceinline__ bool anyhit_prim(const float4* __restrict__ prims, int prim_stride, int k, f3 o, f3 d, float max_dist) {
    const float4 p0 = prims[k * prim_stride], p1 = prims[k * prim_stride + 1], p2 = prims[k * prim_stride + 2];
    const float eps = 1e-8f, eps_up = __uint_as_float(__float_as_uint(1e-8f) + 1u);
    const float t_lo = 1e-5f;                                      // t > 1e-8f && t >= 1e-5f
    if (HAS_QUADS && __float_as_int(p0.w) != 0) {
        const float4 p3 = prims[k * prim_stride + 3];
        const float t1 = mt_candidate(xyz(p0), xyz(p1), xyz(p2), o, d, eps_up, t_lo);
        const float t2 = mt_candidate(xyz(p0), xyz(p2), xyz(p3), o, d, eps_up, t_lo);
        return min_raw(t1, t2) < max_dist;
    }
    const f3 v0 = xyz(p0), edge1 = xyz(p1), edge2 = xyz(p2);
    const f3 h = cross(d, edge2);
    const float a = dot(edge1, h);
    const float f = rcp_exact_normal(a);
    const f3 s = o - v0;
    const float u = f * dot(s, h);
    const float m1 = min3_raw(fabsf(a) - eps, u, 1.0f - u);
    if (!__any(m1 >= 0.0f)) return false;
    const f3 q = cross(s, edge1);
    const float v = f * dot(d, q);
    const float t = f * dot(edge2, q);
    float m = min3_raw(m1, v, 1.0f - (u + v));
    m = min_raw(m, t - t_lo);
    return (m >= 0.0f) & (t <= max_dist);
}

// slab test of visibility_test_anyhit (form_factors.h:162-180); true = the reference does NOT `continue`
__device__ __forceinline__ bool anyhit_box(const float4& n0, const float4& n1, f3 o, f3 inv, float max_dist) {
    const float EPSILON = 1e-5f;
    float t1 = (n0.x - o.x) * inv.x, t2 = (n1.x - o.x) * inv.x;
    float tmin = min_raw(t1, t2), tmax = max_raw(t1, t2);
    t1 = (n0.y - o.y) * inv.y; t2 = (n1.y - o.y) * inv.y;
    tmin = max_raw(tmin, min_raw(t1, t2)); tmax = min_raw(tmax, max_raw(t1, t2));
    t1 = (n0.z - o.z) * inv.z; t2 = (n1.z - o.z) * inv.z;
    tmin = max_raw(tmin, min_raw(t1, t2)); tmax = min_raw(tmax, max_raw(t1, t2));
    return !(tmax < EPSILON || tmin > max_dist || tmin > tmax);
}

// visibility_test_anyhit (form_factors.h:143-208).  The answer - is ANY primitive other than source/target hit within
// max_dist - does not depend on the visiting order unless children get dropped (stack_ptr >= 30), which needs a tree
// deeper than 31 levels.  DEEP = false: stackless pre-order walk (skip pointers).  DEEP = true: the reference's walk
// itself - 32-entry stack, left pushed first (so the right child is visited first), children dropped from 30 on.
template <bool HAS_QUADS, bool DEEP>
__device__ __forceinline__ bool visibility_blocked(const DeviceScene& sc, f3 o, f3 d, float max_dist, int slot_a, int slot_b) {
    const f3 inv = mk3(1.0f / (fabsf(d.x) > 1e-8f ? d.x : 1e-8f), 1.0f / (fabsf(d.y) > 1e-8f ? d.y : 1e-8f),
                       1.0f / (fabsf(d.z) > 1e-8f ? d.z : 1e-8f));
    const float4* __restrict__ nodes = sc.nodes;
    if (!DEEP) {
        // while-while: every lane first walks nodes until it stands on a leaf whose box it hits (or runs out of
        // nodes), then the lanes test their leaves together - node steps and primitive tests do not serialise
        int cur = 0;
        const int n_nodes = sc.n_nodes;
        while (true) {
            int first = 0, count = 0;
            while (cur < n_nodes) {
                const float4 n0 = nodes[2 * cur], n1 = nodes[2 * cur + 1];
                const int a = __float_as_int(n0.w), b = __float_as_int(n1.w);
                const bool pass = anyhit_box(n0, n1, o, inv, max_dist);
                const int here = cur;
                cur = (!pass && b >= 0) ? a : here + 1;
                if (pass && b < 0) { first = a; count = -b; break; }
            }
            if (count == 0) return false;
            for (int i = 0; i < count; i++) {
                const int k = first + i;
                if (k == slot_a || k == slot_b) continue;
                if (anyhit_prim<HAS_QUADS>(sc.prims, sc.prim_stride, k, o, d, max_dist)) return true;
            }
        }
    } else {
        int stack[32];
        int sp = 0;
        stack[sp++] = 0;
        while (sp > 0) {
            const int cur = stack[--sp];
            const float4 n0 = nodes[2 * cur], n1 = nodes[2 * cur + 1];
            if (!anyhit_box(n0, n1, o, inv, max_dist)) continue;
            const int a = __float_as_int(n0.w), b = __float_as_int(n1.w);
            if (b < 0) {
                for (int i = 0; i < -b; i++) {
                    const int k = a + i;
                    if (k == slot_a || k == slot_b) continue;
                    if (anyhit_prim<HAS_QUADS>(sc.prims, sc.prim_stride, k, o, d, max_dist)) return true;
                }
            } else if (sp < 30) {
                stack[sp++] = cur + 1;      // left child (pre-order numbering)
                stack[sp++] = b;            // right child: popped first
            }
        }
        return false;
    }
}

// The same question through the opt-in fast tree (ptmi_config.fast_tree; csrc/wide_bvh.h): is any triangle other than the
// pair's own two hit within max_dist.  Same triangles, same test arithmetic (anyhit_prim's triangle form on the tree's own
// 36-byte records), conservative boxes: the answer is the reference's unless the reference's own slab test drops, by rounding,
// the box of a triangle that the ray does hit.  Stack: one 8-byte entry per tree level, entry e of lane l at stack[e * kBlock].
//
// CERT (the default for triangle scenes from RadiosityState::cert_min_prims = 256 primitives up): the reference's answer for every ray, by proof.
//   "not blocked" needs none: the fast walk reaches every triangle whose hit point lies in range (conservative boxes), the
//     reference's walk tests a subset of them with the same arithmetic.
//   "blocked by triangle k at t": the reference tests k iff every box on the way from its root to k's leaf passes ITS slab
//     test (anyhit_box - no closest-hit distance in it, so the visiting order does not matter).  One fetch decides that for
//     almost every ray: the hit point Q = o + t d inside the LEAF's reference box by eps = 2^-20 (|o_a| + big) on all six
//     faces puts it inside every ancestor's box by as much (boxes are nested), which is more than the slab arithmetic can be
//     off by - t0' = fl(fl(lo - o) fl(1 / d)) lies within 3 * 2^-24 |lo - o| / |d| of the plane's true distance, Q_a' within
//     4 * 2^-24 (|o_a| + big) of Q_a - so every entry distance comes out <= t <= max_dist and every exit distance >= t >= 1e-5.
//     Needs |d_a| > 1e-8 (below that the reference replaces 1 / d by 1e8); otherwise, or within eps of a face: the exact
//     anyhit_box over the leaf's ancestor list (wanc), leaf first, up to the first box that holds Q with the margin; a box of
//     that list failing: the reference's own walk for this ray.
template <bool CERT, bool QUADS = false>
__device__ __forceinline__ bool certified_blocked(const DeviceScene& sc, float4 lo, float4 hi, float t, f3 o, f3 d, float max_dist, int slot_a, int slot_b, unsigned long long& chain) {
    const f3 q = o + t * d;
    // eps from the box's own coordinates (kernels.hip, VERIFY: the same bound and the same argument for the ancestors)
    const float ex = 9.5367431640625e-7f * (fabsf(o.x) + fmaxf(fabsf(lo.x), fabsf(hi.x))), ey = 9.5367431640625e-7f * (fabsf(o.y) + fmaxf(fabsf(lo.y), fabsf(hi.y))),
                ez = 9.5367431640625e-7f * (fabsf(o.z) + fmaxf(fabsf(lo.z), fabsf(hi.z)));
    const bool inside = q.x - lo.x >= ex && hi.x - q.x >= ex && q.y - lo.y >= ey && hi.y - q.y >= ey && q.z - lo.z >= ez && hi.z - q.z >= ez;
    const bool slopes = fabsf(d.x) >= 1.4901161193847656e-8f && fabsf(d.y) >= 1.4901161193847656e-8f && fabsf(d.z) >= 1.4901161193847656e-8f;   // 2^-26 > 1e-8
    if (inside && slopes && sc.w_cert_debug == 0) return true;
    chain++;
    const f3 inv = mk3(1.0f / (fabsf(d.x) > 1e-8f ? d.x : 1e-8f), 1.0f / (fabsf(d.y) > 1e-8f ? d.y : 1e-8f), 1.0f / (fabsf(d.z) > 1e-8f ? d.z : 1e-8f));
    const uint32_t ref = __float_as_uint(lo.w);
    uint32_t off = ref >> 5;
    // leaf first: a box that holds Q with the margin settles every box above it; the ones below it have to pass the slab test
    bool ok = true, proven = false;
    for (int left = (int)(ref & 31u); left > 0 && ok && !proven; left--, off++) {
        const uint4 idx = sc.wanc[off];
        const uint32_t ni[4] = {idx.x, idx.y, idx.z, idx.w};
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const uint32_t j = ni[c] == 0xffffffffu ? 0u : ni[c];          // padding repeats the root
            const float4 n0 = sc.nodes[2 * (size_t)j], n1 = sc.nodes[2 * (size_t)j + 1];
            const float nx = 9.5367431640625e-7f * (fabsf(o.x) + fmaxf(fabsf(n0.x), fabsf(n1.x))), ny = 9.5367431640625e-7f * (fabsf(o.y) + fmaxf(fabsf(n0.y), fabsf(n1.y))),
                        nz = 9.5367431640625e-7f * (fabsf(o.z) + fmaxf(fabsf(n0.z), fabsf(n1.z)));
            const bool holds = slopes && q.x - n0.x >= nx && n1.x - q.x >= nx && q.y - n0.y >= ny && n1.y - q.y >= ny && q.z - n0.z >= nz && n1.z - q.z >= nz;
            ok = ok && (proven || holds || anyhit_box(n0, n1, o, inv, max_dist));
            proven = proven || holds;
        }
    }
    if (ok && sc.w_cert_debug < 2) return true;
    chain += 1ull << 32;
    return visibility_blocked<QUADS, false>(sc, o, d, max_dist, slot_a, slot_b);
}

template <bool CERT, bool QUADS>
__device__ __forceinline__ bool visibility_blocked_wide(const DeviceScene& sc, uint2* stack, f3 o, f3 d, float max_dist, int load_a, int load_b,
                                                        int slot_a, int slot_b, unsigned long long& chain) {
    const f3 inv = mk3(wide_inv(d.x), wide_inv(d.y), wide_inv(d.z));
    const uint32_t octinv = wide_octinv(inv);
    int sp = 0;
    uint32_t g_base = 0u, g_bits = (1u << 8) | (1u << octinv);
    while (true) {
        if ((g_bits & 0xffu) == 0u) {
            if (sp == 0) return false;
            sp--; const uint2 e = stack[sp * kBlock]; g_base = e.x; g_bits = e.y;
        }
        const int bit = 31 - __clz((int)(g_bits & 0xffu));
        g_bits ^= 1u << bit;
        const uint32_t child = (uint32_t)bit ^ octinv;
        const uint32_t ni = g_base + (uint32_t)__popc((g_bits >> 8) & ((1u << child) - 1u));
        if (g_bits & 0xffu) { stack[sp * kBlock] = make_uint2(g_base, g_bits); sp++; }
        const uint4* q = sc.wnodes + 8 * (size_t)ni;
        const WideStep st = wide_node_test(q[0], q[1], q[2], q[3], q[4], q[5], q[6], o, inv, octinv, 1e-5f, max_dist);
        uint32_t tris = st.tris;
        while (tris) {
            const int k = (int)st.tri_base + __ffs((int)tris) - 1;
            tris &= tris - 1u;
            const int li = sc.wload_index[k];
            if (li == load_a || li == load_b) continue;
            if (QUADS && __float_as_int(sc.wqprims[4 * (size_t)k].w) != 0) {       // a quad: quad.h:78,110 accept t < t_max only
                const float4* q = sc.wqprims + 4 * (size_t)k;
                const float eps_up = __uint_as_float(__float_as_uint(1e-8f) + 1u);
                const float tq = min_raw(mt_candidate(xyz(q[0]), xyz(q[1]), xyz(q[2]), o, d, eps_up, 1e-5f), mt_candidate(xyz(q[0]), xyz(q[2]), xyz(q[3]), o, d, eps_up, 1e-5f));
                if (!(tq < max_dist)) continue;
                if (!CERT) return true;
                const float4 c_lo = sc.wcert[kWideCertStride * (size_t)k], c_hi = sc.wcert[kWideCertStride * (size_t)k + 1];
                return certified_blocked<CERT, QUADS>(sc, c_lo, c_hi, tq, o, d, max_dist, slot_a, slot_b, chain);
            }
            const float* r = sc.wprims + 9 * (size_t)k;
            const f3 v0 = mk3(r[0], r[1], r[2]), edge1 = mk3(r[3], r[4], r[5]), edge2 = mk3(r[6], r[7], r[8]);
            const f3 h = cross(d, edge2);                                  // anyhit_prim, triangle form
            const float a = dot(edge1, h);
            const float f = rcp_exact_normal(a);
            const f3 s = o - v0;
            const float u = f * dot(s, h);
            const float m1 = min3_raw(fabsf(a) - 1e-8f, u, 1.0f - u);
            if (!(m1 >= 0.0f)) continue;
            const f3 qq = cross(s, edge1);
            const float v = f * dot(d, qq);
            const float t = f * dot(edge2, qq);
            float m = min3_raw(m1, v, 1.0f - (u + v));
            m = min_raw(m, t - 1e-5f);
            if ((m >= 0.0f) & (t <= max_dist)) {
                if (!CERT) return true;
                // (fetching the leaf box together with the triangle record, before the test: no gain - n = 8192: 84.0 vs 84.4 ms)
                const float4 c_lo = sc.wcert[kWideCertStride * (size_t)k], c_hi = sc.wcert[kWideCertStride * (size_t)k + 1];
                return certified_blocked<CERT, QUADS>(sc, c_lo, c_hi, t, o, d, max_dist, slot_a, slot_b, chain);
            }
        }
        g_base = st.child_base; g_bits = (st.imask << 8) | st.inner;
    }
}

// formfactor_rand_init (form_factors.h:85-89): curand_init(12345 + idx, idx, 0).  Block-synchronous: one 160x160 GF(2)
// matrix T^(2^67 * 2^k) at a time is staged in LDS and applied by the threads whose idx has bit k set.
// ROW (rb.row_jump): idx = i * n + j, and the skip-ahead T^(2^67 idx) = T^(2^67 i n) T^(2^67 j) (powers of one matrix commute, the
// exponents add as integers, carries included): the first factor is the same for every pair of receiver i and comes precomputed
// (ptmi_ff_row_jumps), so a pair applies one matrix per set bit of j (< n) plus that one - 7.5 instead of 13 at n = 8192.
__device__ __forceinline__ void gf2_apply(const uint32_t* M, uint32_t (&v)[5]) {
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
// P_i = T^(2^67 * (i * n)) for every receiver i: the product of the table's matrices T^(2^67 * 2^k) over the set bits k of i * n.
// A matrix is stored as the images of the 160 basis vectors (row r = M e_r, 5 words), so (B A) e_r = B applied to row r of A.
__global__ __launch_bounds__(kBlock) void ptmi_ff_row_jumps(uint32_t* __restrict__ out, int n, const uint32_t* __restrict__ jump) {
    __shared__ uint32_t A[160 * 5], B[160 * 5];
    const int i = blockIdx.x, tid = threadIdx.x;
    const unsigned int hi = (unsigned int)(i * n);
    for (int x = tid; x < 160 * 5; x += kBlock) A[x] = (x / 5) / 32 == x % 5 ? (1u << ((x / 5) % 32)) : 0u;      // the identity: row r = e_r
    __syncthreads();
    for (int k = 0; k < 32; k++) {
        if (!((hi >> k) & 1u)) continue;                                  // block-uniform
        for (int x = tid; x < 160 * 5; x += kBlock) B[x] = jump[k * 160 * 5 + x];
        __syncthreads();
        uint32_t v[5] = {0u, 0u, 0u, 0u, 0u};
        if (tid < 160) { for (int w = 0; w < 5; w++) v[w] = A[tid * 5 + w]; gf2_apply(B, v); }
        __syncthreads();
        if (tid < 160) for (int w = 0; w < 5; w++) A[tid * 5 + w] = v[w];
        __syncthreads();
    }
    for (int x = tid; x < 160 * 5; x += kBlock) out[(size_t)i * 160 * 5 + x] = A[x];
}

__device__ __forceinline__ void pair_rng_init(uint32_t* M, const uint32_t* __restrict__ jump, bool have, unsigned int idx, Rng& out,
                                              const uint32_t* __restrict__ row_jump = nullptr, unsigned int j = 0u, bool row_is_identity = false) {
    const unsigned long long seed = 12345ull + (unsigned long long)idx;
    const uint32_t s0 = ((uint32_t)seed) ^ 0xaad26b49u;
    const uint32_t s1 = (uint32_t)(seed >> 32) ^ 0xf7dcefddu;
    const uint32_t t0 = 1099087573u * s0;
    const uint32_t t1 = 2591861531u * s1;
    uint32_t v[5] = {123456789u + t0, 362436069u ^ t0, 521288629u + t1, 88675123u ^ t1, 5783321u + t0};
    const uint32_t d = 6615241u + t1 + t0;
    const unsigned int bits = row_jump ? j : idx;                        // with the row's shared factor: only the bits of j here
    for (int k = 0; k < 32; k++) {
        const bool mine = have && ((bits >> k) & 1u);
        if (!__syncthreads_or(mine ? 1 : 0)) continue;
        for (int i = threadIdx.x; i < 160 * 5; i += kBlock) M[i] = jump[k * 160 * 5 + i];
        __syncthreads();
        if (mine) gf2_apply(M, v);
        __syncthreads();
    }
    if (row_jump && !row_is_identity) {                                   // block-uniform: T^(2^67 i n), the same for the whole row
        for (int i = threadIdx.x; i < 160 * 5; i += kBlock) M[i] = row_jump[i];
        __syncthreads();
        if (have) gf2_apply(M, v);
        __syncthreads();
    }
    out = Rng{v[0], v[1], v[2], v[3], v[4], d};
}

// the sample loop and F_ij of calculate_form_factors_mc_kernel (form_factors.h:259-365) for one surviving pair
// The emitter's record is read again for every sample (it is L1-resident, and only the sample's first lines use it): its
// 13 - 18 registers do not have to live through the visibility walk, where the kernel is short of them (7 waves per SIMD).
template <bool HAS_QUADS, bool DEEP, bool RAD0, int WIDE>
__device__ __forceinline__ float mc_pair(const DeviceScene& sc, uint2* wstack, int i, const Geom& gi, const float4* __restrict__ geo, int j, int slot_i, int slot_j,
                                         int actual_samples, Rng& rng, f3 radiosity_j, unsigned int* counts, float* radg, unsigned int& rays, unsigned long long& chain) {
    float visibility_sum = 0.0f, cos_i_sum = 0.0f, cos_j_sum = 0.0f, dist_sum = 0.0f;
    int valid_samples = 0;
    for (int s = 0; s < actual_samples; ++s) {
        asm volatile("" ::: "memory");                                          // keeps the loads below inside the loop
        const Geom gj = load_geom(geo, j);
        float r1 = rng_uniform(rng), r2 = rng_uniform(rng);
        const f3 p_i = sample_uniform<HAS_QUADS>(gi, r1, r2);
        r1 = rng_uniform(rng); r2 = rng_uniform(rng);
        const f3 p_j = sample_uniform<HAS_QUADS>(gj, r1, r2);
        f3 sample_dir = p_j - p_i;
        const float r = length(sample_dir);
        if (r < 1e-6f) continue;
        sample_dir = div_scalar(sample_dir, r);
        const float cos_theta_i = dot(gi.normal, sample_dir);
        const float cos_theta_j = -dot(gj.normal, sample_dir);
        if (cos_theta_i <= 0.0f || cos_theta_j <= 0.0f) continue;
        const f3 ro = p_i + 1e-4f * gi.normal;
        const f3 rd = unit_vector(sample_dir);                                  // Ray's constructor normalises again (ray.h:9-12)
        rays++;
        const bool blocked = WIDE ? visibility_blocked_wide<WIDE == 2, HAS_QUADS>(sc, wstack, ro, rd, r - 2e-4f, i, j, slot_i, slot_j, chain)
                                  : visibility_blocked<HAS_QUADS, DEEP>(sc, ro, rd, r - 2e-4f, slot_i, slot_j);
        if (!blocked) {
            visibility_sum += 1.0f; cos_i_sum += cos_theta_i; cos_j_sum += cos_theta_j; dist_sum += r;
            valid_samples++;
            const int grid_idx = direction_to_grid_index_local(sample_dir, gi.normal);
            atomicAdd(&counts[grid_idx], 1u);
            if (RAD0) {
                const float geometric_weight = (cos_theta_i * cos_theta_j) / (r * r);
                const f3 contrib = gj.area * (geometric_weight * radiosity_j);
                atomicAdd(&radg[3 * grid_idx], contrib.x); atomicAdd(&radg[3 * grid_idx + 1], contrib.y); atomicAdd(&radg[3 * grid_idx + 2], contrib.z);
            }
        }
    }
    if (valid_samples > 0) {
        const float avg_cos_i = cos_i_sum / (float)valid_samples;
        const float avg_cos_j = cos_j_sum / (float)valid_samples;
        const float avg_dist = dist_sum / (float)valid_samples;
        const float visibility_fraction = visibility_sum / (float)actual_samples;
        const float area_j = geo[6 * j + 1].w;
        const float F_ij = (float)((double)(visibility_fraction * (avg_cos_i * avg_cos_j * area_j)) /
                                   (PTMI_PI_D * (double)avg_dist * (double)avg_dist));
        return fmaxf(0.0f, fminf(F_ij, 1.0f));
    }
    return 0.0f;
}

// calculate_form_factors_kernel (form_factors.h:368-415) after its culling tests
template <bool HAS_QUADS, bool DEEP, int WIDE>
__device__ __forceinline__ float p2p_pair(const DeviceScene& sc, uint2* wstack, int i, int j, const Geom& gi, const Geom& gj, int slot_i, int slot_j, unsigned int& rays,
                                          unsigned long long& chain) {
    const f3 vec_ij = gj.centroid - gi.centroid;
    const float r = length(vec_ij);
    const f3 dir_ij = div_scalar(vec_ij, r);
    const float cos_theta_i = dot(gi.normal, dir_ij);
    const float cos_theta_j = dot(gj.normal, -dir_ij);
    const f3 ro = gi.centroid + 1e-4f * gi.normal;
    const f3 rd = unit_vector(dir_ij);
    rays++;
    if (WIDE ? visibility_blocked_wide<WIDE == 2, HAS_QUADS>(sc, wstack, ro, rd, r - 2e-4f, i, j, slot_i, slot_j, chain)
             : visibility_blocked<HAS_QUADS, DEEP>(sc, ro, rd, r - 2e-4f, slot_i, slot_j)) return 0.0f;
    const float ff = (float)((double)(cos_theta_i * cos_theta_j * gj.area) / (PTMI_PI_D * (double)r * (double)r));
    return fmaxf(0.0f, ff);
}

// 7 waves per SIMD (72 VGPRs, 23 dwords spilled) instead of the 4 the kernel asks for by itself (112 VGPRs): the any-hit walks
// wait on L2, and more waves in flight are worth more than the spills cost - n = 8192: 4 / 5 / 6 / 7 / 8 waves 168.6 / 151.6 /
// 139.0 / 134.8 / 133.5 ms, and 131.4 ms at 7 waves with the emitter's record re-read per sample (mc_pair); n = 2048: 7 waves
// 14.5 ms, 8 waves 15.0 ms
#ifndef PTMI_FF_WIDE_WAVES
#define PTMI_FF_WIDE_WAVES 6
#endif
// WIDE (1: the opt-in fast tree, 2: the certified walk - the default from 256 triangles up): the visibility walk goes through
// visibility_blocked_wide; its node test wants ~80 registers, so that build is bounded to 6 waves per SIMD (n = 8192, fast tree:
// 4 / 5 / 6 waves 78.8 / 74.9 / 70.4 ms; certified: 5 / 6 / 7 waves 84.1 / 84.4 / 83.5 ms; the reference's walk: 130.4) and keeps
// its per-lane stack in dynamic LDS (depth x 2 KB per workgroup)
template <bool MC, bool HAS_QUADS, bool DEEP, bool RAD0, int WIDE>
__global__ __launch_bounds__(kBlock, WIDE ? PTMI_FF_WIDE_WAVES : 7) void ptmi_form_factors(DeviceScene sc, RadiosityBuffers rb, int n_samples,
                                                            const uint32_t* __restrict__ jump) {
    extern __shared__ uint2 ff_wstack[];
    uint2* wstack = ff_wstack + threadIdx.x;
    __shared__ uint32_t M[160 * 5];
    __shared__ unsigned int counts[kGridSize];
    __shared__ float radg[RAD0 ? 3 * kGridSize : 1];
    __shared__ int2 queue[kQueueCap];
    __shared__ int q_n;
    __shared__ unsigned int rays_wg;
    const int n = rb.n;
    const int i = blockIdx.x;
    const int tid = threadIdx.x;
    const Geom gi = load_geom(rb.geo, i);
    const int slot_i = rb.slot_of[i];
    float* __restrict__ row = rb.form_factors + (size_t)i * (size_t)n;
    counts[tid] = 0u;
    if (RAD0) { radg[3 * tid] = 0.0f; radg[3 * tid + 1] = 0.0f; radg[3 * tid + 2] = 0.0f; }
    if (tid == 0) { q_n = 0; rays_wg = 0u; }
    __syncthreads();
    unsigned int rays = 0u;
    unsigned long long chain = 0ull;              // certified walk: rays that took the ancestor chain (low word) / the reference's walk (high word)

    for (int base = 0; base < n; base += kBlock) {
        const int j = base + tid;
        int samples = 0;                                    // 0: this pair's form factor is already decided (0)
        if (j < n) {
            if (j != i) {
                const float4 cj = rb.geo[6 * j + 5], nj = rb.geo[6 * j + 4];
                if (MC) {                                   // form_factors.h:234-256
                    const f3 dir_ij = xyz(cj) - gi.centroid;
                    const float dist_sq = dir_ij.x * dir_ij.x + dir_ij.y * dir_ij.y + dir_ij.z * dir_ij.z;
                    const float dist = sqrt_rn(dist_sq);
                    if (!(dist < 1e-6f)) {
                        const f3 dir_norm = div_scalar(dir_ij, dist);
                        const float cos_i_approx = dot(gi.normal, dir_norm);
                        const float cos_j_approx = -dot(xyz(nj), dir_norm);
                        if (!(cos_i_approx <= 0.0f || cos_j_approx <= 0.0f)) {
                            const float area_j = rb.geo[6 * j + 1].w;
                            const float approx_ff = (float)((double)(cos_i_approx * cos_j_approx * area_j) / (PTMI_PI_D * (double)dist_sq));
                            samples = n_samples;
                            if (approx_ff < 0.001f) samples = max(1, n_samples / 4);
                            else if (approx_ff < 0.01f) samples = max(2, n_samples / 2);
                        }
                    }
                } else {                                    // form_factors.h:385-401
                    const f3 vec_ij = xyz(cj) - gi.centroid;
                    const float r = length(vec_ij);
                    if (!(r < 1e-6f)) {
                        const f3 dir_ij = div_scalar(vec_ij, r);
                        const float cos_theta_i = dot(gi.normal, dir_ij);
                        const float cos_theta_j = dot(xyz(nj), -dir_ij);
                        if (!(cos_theta_i <= 0.0f || cos_theta_j <= 0.0f)) samples = 1;
                    }
                }
            }
            if (samples == 0) row[j] = 0.0f;
        }
        if (samples) { const int pos = atomicAdd(&q_n, 1); queue[pos] = make_int2(j, samples); }
        __syncthreads();
        const bool last = base + kBlock >= n;
        while (q_n >= kBlock || (last && q_n > 0)) {        // q_n is block-uniform between barriers
            const int total = q_n;
            const int take = min(total, kBlock);
            const bool have = tid < take;
            const int2 e = have ? queue[total - take + tid] : make_int2(0, 0);
            __syncthreads();
            if (tid == 0) q_n = total - take;
            Rng rng = {0u, 0u, 0u, 0u, 0u, 0u};
            if (MC) pair_rng_init(M, jump, have, (unsigned int)(i * n + e.x), rng, rb.row_jump ? rb.row_jump + (size_t)i * 160 * 5 : nullptr,
                                  (unsigned int)e.x, i == 0);
            if (have) {
                const int slot_j = rb.slot_of[e.x];
                float F;
                if (MC) F = mc_pair<HAS_QUADS, DEEP, RAD0, WIDE>(sc, wstack, i, gi, rb.geo, e.x, slot_i, slot_j, e.y, rng, xyz(rb.radiosity[e.x]), counts, radg, rays, chain);
                else F = p2p_pair<HAS_QUADS, DEEP, WIDE>(sc, wstack, i, e.x, gi, load_geom(rb.geo, e.x), slot_i, slot_j, rays, chain);
                row[e.x] = F;
            }
            __syncthreads();
        }
    }
    atomicAdd(&rays_wg, rays);
    __syncthreads();
    rb.grid[(size_t)i * kGridSize + tid] = counts[tid];     // initialize_directional_grids + the kernel's atomics, in one store
    rb.rad_grid[(size_t)i * kGridSize + tid] = RAD0 ? make_float4(radg[3 * tid], radg[3 * tid + 1], radg[3 * tid + 2], 0.0f)
                                                    : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (tid == 0 && rb.rays) atomicAdd(rb.rays, (unsigned long long)rays_wg);
    if (WIDE == 2 && rb.rays) {                    // one pair of atomics per wave (the two 32-bit words cannot carry into each other: a lane's rays stay far below 2^32)
        unsigned long long c = chain;
        for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
        if ((threadIdx.x & 63) == 0 && c) { atomicAdd(rb.rays + 1, c & 0xffffffffull); atomicAdd(rb.rays + 2, c >> 32); }
    }
}

// radiosity_iteration_kernel (form_factors.h:441-465): one thread per receiver, ascending j - the float sum is a
// sequential chain per row, so rows are the only parallelism and the kernel is bound by that chain (~12 VALU ops per
// j), provided the row data arrive in time: a lane streams its own row in 64-float groups (sixteen 16-byte loads),
// the NEXT group's loads are issued before the current group is consumed; unshot[] is staged through LDS in chunks and
// read as a broadcast.  One wave per workgroup, so the 64-row groups spread over as many CUs as possible.
constexpr int kIterChunk = 1024, kIterBlock = 64, kIterGroup = 64;
struct RowGroup { float4 q[kIterGroup / 4]; };
__device__ __forceinline__ void load_group(RowGroup& g, const float* p) {
    const float4* p4 = reinterpret_cast<const float4*>(p);
#pragma unroll
    for (int k = 0; k < kIterGroup / 4; k++) g.q[k] = p4[k];
}
__device__ __forceinline__ void consume_group(const RowGroup& g, const float4* u, int j0, int i, f3& incident_rad) {
#pragma unroll
    for (int k = 0; k < kIterGroup / 4; k++) {
        const float f[4] = {g.q[k].x, g.q[k].y, g.q[k].z, g.q[k].w};
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const float4 uj = u[4 * k + c];
            const bool take = (j0 + 4 * k + c != i) & (f[c] > 0.0f);
            const float tx = incident_rad.x + f[c] * uj.x, ty = incident_rad.y + f[c] * uj.y, tz = incident_rad.z + f[c] * uj.z;
            incident_rad.x = take ? tx : incident_rad.x;            // per component: a struct select goes through scratch
            incident_rad.y = take ? ty : incident_rad.y;
            incident_rad.z = take ? tz : incident_rad.z;
        }
    }
}
__global__ __launch_bounds__(kIterBlock) void ptmi_radiosity_iterate(RadiosityBuffers rb, int src) {
    __shared__ float4 u_lds[kIterChunk];
    const int n = rb.n;
    const int i = blockIdx.x * kIterBlock + threadIdx.x;
    const int row_i = min(i, n - 1);                           // lanes past the end walk the last row and store nothing
    const float* __restrict__ row = rb.form_factors + (size_t)row_i * (size_t)n;
    const float4* __restrict__ unshot = rb.unshot[src];
    const int n_vec = (n & 3) == 0 ? (n / kIterGroup) * kIterGroup : 0;   // rows start on 16-byte boundaries only if n % 4 == 0
    f3 incident_rad = mk3(0.0f, 0.0f, 0.0f);
    RowGroup ga, gb;
    if (n_vec > 0) load_group(ga, row);
    for (int base = 0; base < n; base += kIterChunk) {         // kIterChunk is a multiple of 2 * kIterGroup
        const int m = min(kIterChunk, n - base);
        __syncthreads();
        for (int k = threadIdx.x; k < m; k += kIterBlock) u_lds[k] = unshot[base + k];
        __syncthreads();
        int jj = 0;
        for (; base + jj + 2 * kIterGroup <= n_vec && jj + 2 * kIterGroup <= m; jj += 2 * kIterGroup) {
            load_group(gb, row + base + jj + kIterGroup);
            consume_group(ga, u_lds + jj, base + jj, i, incident_rad);
            if (base + jj + 3 * kIterGroup <= n_vec) load_group(ga, row + base + jj + 2 * kIterGroup);
            consume_group(gb, u_lds + jj + kIterGroup, base + jj + kIterGroup, i, incident_rad);
        }
        if (base + jj + kIterGroup <= n_vec && jj + kIterGroup <= m) {      // an odd group left over at the end of the vector part
            consume_group(ga, u_lds + jj, base + jj, i, incident_rad);
            jj += kIterGroup;
            if (base + jj + kIterGroup <= n_vec) load_group(ga, row + base + jj);
        }
        for (; jj < m; jj++) {
            const float F_ij = row[base + jj];
            if (base + jj != i && F_ij > 0.0f) incident_rad = incident_rad + F_ij * xyz(u_lds[jj]);
        }
    }
    if (i >= n) return;
    const f3 bsdf = xyz(rb.bsdf[i]);
    const f3 reflected = mk3(fminf(bsdf.x * incident_rad.x, incident_rad.x), fminf(bsdf.y * incident_rad.y, incident_rad.y),
                             fminf(bsdf.z * incident_rad.z, incident_rad.z));
    const f3 rad = xyz(rb.radiosity[i]) + reflected;
    rb.radiosity[i] = make_float4(rad.x, rad.y, rad.z, 0.0f);
    rb.unshot[1 - src][i] = make_float4(reflected.x, reflected.y, reflected.z, 0.0f);
}


// ---- ptmi_radiosity_iterate_tiled: the same step as an HBM stream ----------------------------------------------------------
// The Jacobi step reads the n x n form-factor matrix once (4 n^2 bytes: 268 MB at n = 8192) and does 6 flops per entry,
// so its roofline is HBM bandwidth; what holds it back is that every row's sum is ONE sequential float chain (ascending j -
// the reference's order, form_factors.h:452-459, which fixes the bits).  With a lane per row (the kernel above) a wave reads
// 64 rows at a 4n-byte stride - 64 cache lines per load instruction - and its chain costs 11 instructions per j.
// Here a 64-thread workgroup owns R = 8 (or 16) rows:
//   * loading: one global_load_dwordx4 per lane covers 256 consecutive floats of ONE row (1 KiB, fully coalesced); R of
//     them are an R x 256 tile, issued for tile t + 1 before tile t is consumed (R KiB in flight per wave), staged in LDS
//     with a row stride of 260 floats (the lanes of a 16-lane group then read different rows from different banks);
//   * summing: lane = channel * R + row, i.e. the three colour channels of a row are three lanes, each with ONE running
//     sum.  F is read from LDS as a broadcast to the row's three lanes, the unshot radiosity as a per-channel (SoA)
//     broadcast to a channel's sixteen.  Entries the reference skips (`j != i && F_ij > 0.0f` false: the diagonal, F <= 0,
//     columns past n) are stored as +0 in the tile, and adding 0 * u_j changes nothing while u_j is finite - so per j the
//     chain is one multiply and one add (the lane-per-row kernel: 3 + 3 + 3 selects + 2 compares); a tile holding a
//     non-finite unshot value takes the compare / select form.  Same sums in the same order: bit-identical to the kernel
//     above and to the oracle.
constexpr int kTileCols = 256, kTileStride = kTileCols + 4, kUStride = kTileCols / 4 + 1;
template <int kTileRows>
__global__ __launch_bounds__(64) void ptmi_radiosity_iterate_tiled(RadiosityBuffers rb, int src) {
    __shared__ float4 f_lds[kTileRows * kTileStride / 4];
    __shared__ float4 u_lds[4][kUStride];                             // [channel][column], channels on different banks
    const int n = rb.n, lane = threadIdx.x;
    const int row0 = blockIdx.x * kTileRows;
    const int ch = lane / kTileRows, r = lane % kTileRows, i = row0 + r;   // the lane's running sum: channel ch (>= 3: none) of row i
    const float* __restrict__ F = rb.form_factors;
    const float4* __restrict__ unshot = rb.unshot[src];
    const int n_tiles = (n + kTileCols - 1) / kTileCols;
    float4 fr[kTileRows], ur[4];
    auto issue = [&](int t) {                                         // tile t -> registers (n % 4 == 0: a float4 never straddles n)
        const int col = t * kTileCols + 4 * lane;
#pragma unroll
        for (int k = 0; k < kTileRows; k++) {
            const int rk = min(row0 + k, n - 1);
            fr[k] = col < n ? *reinterpret_cast<const float4*>(F + (size_t)rk * (size_t)n + col) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int j = t * kTileCols + lane + 64 * c;
            ur[c] = j < n ? unshot[j] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
    };
    bool u_finite = true;                                             // wave-uniform: every unshot value of the tile is finite
    auto commit = [&](int t) {                                        // registers -> LDS
        // entries the reference skips (F_ij <= 0, or NaN) become +0: with a finite u_j, `sum + 0 * u_j` leaves the running sum
        // as it is (it is never -0: it starts at +0), so the chain below needs no compare / select per entry
#pragma unroll
        for (int k = 0; k < kTileRows; k++)
            f_lds[(k * kTileStride) / 4 + lane] = make_float4(fmaxf(fr[k].x, 0.0f), fmaxf(fr[k].y, 0.0f), fmaxf(fr[k].z, 0.0f), fmaxf(fr[k].w, 0.0f));
        bool fin = true;
#pragma unroll
        for (int c = 0; c < 4; c++) fin = fin && isfinite(ur[c].x) && isfinite(ur[c].y) && isfinite(ur[c].z);
        u_finite = __all(fin);
        float* us = reinterpret_cast<float*>(u_lds);
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int jl = lane + 64 * c;
            us[0 * 4 * kUStride + jl] = ur[c].x; us[1 * 4 * kUStride + jl] = ur[c].y; us[2 * 4 * kUStride + jl] = ur[c].z; us[3 * 4 * kUStride + jl] = 0.0f;
        }
        __syncthreads();
        const int d = i - t * kTileCols;                              // the diagonal entry of row i, if it lies in this tile
        if (ch == 0 && d >= 0 && d < kTileCols) reinterpret_cast<float*>(f_lds)[r * kTileStride + d] = 0.0f;
        __syncthreads();
    };
    float acc = 0.0f;
    issue(0);
    for (int t = 0; t < n_tiles; t++) {
        commit(t);
        if (t + 1 < n_tiles) issue(t + 1);
        const float4* frow = f_lds + (r * kTileStride) / 4;
        const float4* urow = u_lds[min(ch, 3)];
        if (u_finite) {
#pragma unroll 8
            for (int q = 0; q < kTileCols / 4; q++) {
                const float4 f4 = frow[q], u4 = urow[q];
                acc = acc + f4.x * u4.x; acc = acc + f4.y * u4.y; acc = acc + f4.z * u4.z; acc = acc + f4.w * u4.w;
            }
        } else {                                                      // an infinite or NaN unshot value: 0 * u_j would not be 0
#pragma unroll 4
            for (int q = 0; q < kTileCols / 4; q++) {
                const float4 f4 = frow[q], u4 = urow[q];
                float tsum;
                tsum = acc + f4.x * u4.x; acc = f4.x > 0.0f ? tsum : acc;
                tsum = acc + f4.y * u4.y; acc = f4.y > 0.0f ? tsum : acc;
                tsum = acc + f4.z * u4.z; acc = f4.z > 0.0f ? tsum : acc;
                tsum = acc + f4.w * u4.w; acc = f4.w > 0.0f ? tsum : acc;
            }
        }
        __syncthreads();                                              // the tile is consumed before commit(t + 1) overwrites it
    }
    if (i >= n) return;
    float* rad = reinterpret_cast<float*>(rb.radiosity + i);
    float* out = reinterpret_cast<float*>(rb.unshot[1 - src] + i);
    if (ch > 3) return;
    if (ch == 3) { out[3] = 0.0f; return; }
    const float bsdf = reinterpret_cast<const float*>(rb.bsdf + i)[ch];
    const float reflected = fminf(bsdf * acc, acc);
    rad[ch] = rad[ch] + reflected;
    out[ch] = reflected;
}

// grid_filter.h:35-41, 55-101 (bilateral), 221-249 (gaussian)
__device__ __forceinline__ float gaussian_weight(float distance, float sigma) { return ptmi_expf(-(distance * distance) / (2.0f * sigma * sigma)); }
__device__ __forceinline__ float luminance_from_rgb(f3 rgb) { return 0.2126f * rgb.x + 0.7152f * rgb.y + 0.0722f * rgb.z; }
__device__ __forceinline__ f3 filter_cell(const float4* g, int center_i, int center_j, bool bilateral, float sigma_spatial, float sigma_range) {
    const f3 center_val = xyz(g[center_i * kGridRes + center_j]);
    const float center_lum = luminance_from_rgb(center_val);
    f3 weighted_sum = mk3(0.0f, 0.0f, 0.0f);
    float total_weight = 0.0f;
    for (int di = -2; di <= 2; di++)
        for (int dj = -2; dj <= 2; dj++) {
            const int ni = center_i + di;
            const int nj = (center_j + dj + kGridRes) % kGridRes;
            if (ni < 0 || ni >= kGridRes) continue;
            const f3 neighbor_val = xyz(g[ni * kGridRes + nj]);
            const float spatial_dist = sqrt_rn((float)(di * di + dj * dj));
            float weight = gaussian_weight(spatial_dist, sigma_spatial);
            if (bilateral) {
                const float range_dist = fabsf(center_lum - luminance_from_rgb(neighbor_val));
                weight = weight * gaussian_weight(range_dist, sigma_range);
            }
            weighted_sum = weighted_sum + weight * neighbor_val;
            total_weight += weight;
        }
    if (total_weight > 1e-6f) return div_scalar(weighted_sum, total_weight);
    return center_val;
}

// update_radiosity_grid (form_factors.h:405-439): workgroup = primitive i, thread = grid cell.  Phase A computes, in
// parallel, which cell each j falls into (the acos/atan2 part); phase B lets every cell's owner add its contributions
// in ascending j, the order of the reference's single thread per primitive.
constexpr int kGridChunk = 2048;
__global__ __launch_bounds__(kBlock) void ptmi_radiosity_grid(RadiosityBuffers rb, RadiosityParams prm) {
    __shared__ unsigned short cell[kGridChunk];
    __shared__ float4 g[kGridSize];
    const int n = rb.n;
    const int i = blockIdx.x;
    const int tid = threadIdx.x;
    const f3 center_i = xyz(rb.geo[6 * i + 5]), normal_i = xyz(rb.geo[6 * i + 4]);
    const float* __restrict__ row = rb.form_factors + (size_t)i * (size_t)n;
    f3 acc = mk3(0.0f, 0.0f, 0.0f);
    for (int base = 0; base < n; base += kGridChunk) {
        const int m = min(kGridChunk, n - base);
        for (int jj = tid; jj < m; jj += kBlock) {
            const int j = base + jj;
            unsigned short c = 0xffffu;
            if (j != i && row[j] > 0.0f) {
                f3 dir_ij = xyz(rb.geo[6 * j + 5]) - center_i;
                const float r = length(dir_ij);
                if (!(r < 1e-6f)) {
                    dir_ij = div_scalar(dir_ij, r);
                    c = (unsigned short)direction_to_grid_index_local(dir_ij, normal_i);
                }
            }
            cell[jj] = c;
        }
        __syncthreads();
        for (int jj = 0; jj < m; jj++) {
            if (cell[jj] == tid) {
                const int j = base + jj;
                acc = acc + row[j] * xyz(rb.radiosity[j]);
            }
        }
        __syncthreads();
    }
    if (prm.enable_filtering) {                              // application_state.h:759-767
        g[tid] = make_float4(acc.x, acc.y, acc.z, 0.0f);
        __syncthreads();
        acc = filter_cell(g, tid / kGridRes, tid % kGridRes, prm.use_bilateral != 0, prm.filter_sigma_spatial, prm.filter_sigma_range);
    }
    rb.rad_grid[(size_t)i * kGridSize + tid] = make_float4(acc.x, acc.y, acc.z, 0.0f);
}

// filter_pdfs_for_primitives (grid_filter.h:329-507): luminance of the radiosity grid and the count grid, each through
// the 5x5 float filter (bilateral :381-406 / gaussian :352-379) and normalize_pdf_kernel (:409-418, a sequential sum).
__device__ __forceinline__ float filter_cell_float(const float* src, int ci, int cj, bool bilateral, float sigma_spatial, float sigma_range) {
    const float center = src[ci * kGridRes + cj];
    float weighted_sum = 0.0f, total_weight = 0.0f;
    for (int di = -2; di <= 2; di++)
        for (int dj = -2; dj <= 2; dj++) {
            const int ni = ci + di, nj = (cj + dj + kGridRes) % kGridRes;
            if (ni < 0 || ni >= kGridRes) continue;
            float w = gaussian_weight(sqrt_rn((float)(di * di + dj * dj)), sigma_spatial);
            if (bilateral) w = w * gaussian_weight(fabsf(center - src[ni * kGridRes + nj]), sigma_range);
            weighted_sum += src[ni * kGridRes + nj] * w;
            total_weight += w;
        }
    if (total_weight > 1e-6f) return weighted_sum / total_weight;
    return center;
}
__global__ __launch_bounds__(kBlock) void ptmi_filter_pdfs(const float* __restrict__ rgb, const float* __restrict__ counts,
                                                           float* __restrict__ out_formfactor, float* __restrict__ out_radiosity,
                                                           int bilateral, float sigma_spatial, float sigma_range) {
    __shared__ float lum[kGridSize], cnt[kGridSize], f_lum[kGridSize], f_cnt[kGridSize];
    __shared__ float sums[2];
    const size_t base = (size_t)blockIdx.x * kGridSize;
    const int tid = threadIdx.x;
    const float* c = rgb + (base + tid) * 3;
    lum[tid] = 0.2126f * c[0] + 0.7152f * c[1] + 0.0722f * c[2];               // luminanceFromRGB (grid_filter.h:39-41)
    cnt[tid] = counts ? counts[base + tid] : 0.0f;
    __syncthreads();
    f_lum[tid] = filter_cell_float(lum, tid / kGridRes, tid % kGridRes, bilateral != 0, sigma_spatial, sigma_range);
    f_cnt[tid] = filter_cell_float(cnt, tid / kGridRes, tid % kGridRes, bilateral != 0, sigma_spatial, sigma_range);
    __syncthreads();
    if (tid == 0 || tid == 64) {                                               // one lane of two different waves
        const float* src = tid == 0 ? f_lum : f_cnt;
        float sum = 0.0f;
        for (int i = 0; i < kGridSize; i++) sum += src[i];
        sums[tid ? 1 : 0] = sum;
    }
    __syncthreads();
    out_radiosity[base + tid] = (sums[0] <= 1e-12f) ? f_lum[tid] : f_lum[tid] / sums[0];
    out_formfactor[base + tid] = (sums[1] <= 1e-12f) ? f_cnt[tid] : f_cnt[tid] / sums[1];
}

// SceneState::precomputeCDFs / precomputeCDFsFromFiltered (application_state.h:492-585, 587-680): one PrecomputedCDF record
// per primitive, one thread each - every sum runs in the reference's order.  SRC 0: float4 radiosity grids (the solver's),
// 1: packed float3 grids (host-supplied), 2: ready pdf values (filtered luminance).
template <int SRC>
__global__ __launch_bounds__(kBlock) void ptmi_cdf_records(const void* __restrict__ src, float* __restrict__ out, int n) {
    // one workgroup per primitive, one thread per cell; the sequential sums are done by one thread per row (and thread 0
    // for the marginal) from LDS, in the reference's order
    __shared__ float pdf[kGridSize], row_cdfs[kGridSize], row_sums[kGridRes / 2], marginal[kGridRes / 2];
    __shared__ float total;
    constexpr int GRID_HALF_RES = kGridRes / 2;
    const float GRID_INV_RES = 1.0f / kGridRes;
    const int p = blockIdx.x, i = threadIdx.x;
    float v;
    if (SRC == 0) { const float4 c = static_cast<const float4*>(src)[(size_t)p * kGridSize + i]; v = 0.2126f * c.x + 0.7152f * c.y + 0.0722f * c.z; }
    else if (SRC == 1) { const float* c = static_cast<const float*>(src) + ((size_t)p * kGridSize + i) * 3; v = 0.2126f * c[0] + 0.7152f * c[1] + 0.0722f * c[2]; }
    else v = static_cast<const float*>(src)[(size_t)p * kGridSize + i];
    pdf[i] = v;
    __syncthreads();
    if (i < GRID_HALF_RES) {
        float row_sum = 0.0f;
        for (int u = 0; u < kGridRes; u++) row_sum += pdf[i * kGridRes + u];
        row_sums[i] = row_sum;
    }
    __syncthreads();
    if (i == 0) {
        float total_weight = 0.0f;
        for (int r = 0; r < GRID_HALF_RES; r++) total_weight += row_sums[r];
        float running = 0.0f;
        const float inv_total = (total_weight > 1e-6f) ? (1.0f / total_weight) : 0.0f;
        for (int r = 0; r < GRID_HALF_RES; r++) { running += row_sums[r]; marginal[r] = running * inv_total; }
        marginal[GRID_HALF_RES - 1] = 1.0f;
        total = total_weight;
    }
    if (i < kGridRes) {                                   // thread i = row i
        const int ro = i * kGridRes;
        if (i >= GRID_HALF_RES || row_sums[i] < 1e-6f) {
            for (int u = 0; u < kGridRes; u++) row_cdfs[ro + u] = (float)(u + 1) * GRID_INV_RES;
        } else {
            float running_row = 0.0f;
            const float inv_row_sum = 1.0f / row_sums[i];
            for (int u = 0; u < kGridRes; u++) { running_row += pdf[ro + u]; row_cdfs[ro + u] = running_row * inv_row_sum; }
            row_cdfs[ro + kGridRes - 1] = 1.0f;
        }
    }
    __syncthreads();
    float* cdf = out + (size_t)p * kCdfDwords;
    cdf[kCdfPdf + i] = pdf[i];
    cdf[kCdfRowCdfs + i] = row_cdfs[i];
    if (i < GRID_HALF_RES) { cdf[kCdfRowSums + i] = row_sums[i]; cdf[kCdfMarginal + i] = marginal[i]; }
    if (i == 0) { cdf[kCdfTotal] = total; cdf[kCdfValid] = __int_as_float(total > 1e-6f ? 1 : 0); }
}

template <bool MC, bool Q_, bool D_>
void launch_ff3(bool rad0, dim3 grid, hipStream_t s, const DeviceScene& sc, const RadiosityBuffers& rb, int n_samples, const uint32_t* jump) {
    if constexpr (!D_) {
        const size_t lds = (size_t)sc.w_depth * kBlock * sizeof(uint2);
        const bool records = Q_ ? sc.wqprims != nullptr : sc.wprims != nullptr;
        if (rb.fast_tree == 1 && sc.wnodes && records) {   // the opt-in fast tree for the visibility walk, no certificate
            if (MC && rad0) hipLaunchKernelGGL((ptmi_form_factors<MC, Q_, false, true, 1>), grid, dim3(kBlock), lds, s, sc, rb, n_samples, jump);
            else hipLaunchKernelGGL((ptmi_form_factors<MC, Q_, false, false, 1>), grid, dim3(kBlock), lds, s, sc, rb, n_samples, jump);
            return;
        }
        if (rb.fast_tree == 2 && sc.wnodes && records && sc.wcert && sc.wanc) {   // the certified walk: the reference's answers, through the fast tree
            if (MC && rad0) hipLaunchKernelGGL((ptmi_form_factors<MC, Q_, false, true, 2>), grid, dim3(kBlock), lds, s, sc, rb, n_samples, jump);
            else hipLaunchKernelGGL((ptmi_form_factors<MC, Q_, false, false, 2>), grid, dim3(kBlock), lds, s, sc, rb, n_samples, jump);
            return;
        }
    }
    if (MC && rad0) hipLaunchKernelGGL((ptmi_form_factors<MC, Q_, D_, true, 0>), grid, dim3(kBlock), 0, s, sc, rb, n_samples, jump);
    else hipLaunchKernelGGL((ptmi_form_factors<MC, Q_, D_, false, 0>), grid, dim3(kBlock), 0, s, sc, rb, n_samples, jump);
}
template <bool MC>
void launch_ff1(bool quads, bool deep, bool rad0, dim3 grid, hipStream_t s, const DeviceScene& sc, const RadiosityBuffers& rb, int n_samples,
                const uint32_t* jump) {
    if (quads) { if (deep) launch_ff3<MC, true, true>(rad0, grid, s, sc, rb, n_samples, jump); else launch_ff3<MC, true, false>(rad0, grid, s, sc, rb, n_samples, jump); }
    else { if (deep) launch_ff3<MC, false, true>(rad0, grid, s, sc, rb, n_samples, jump); else launch_ff3<MC, false, false>(rad0, grid, s, sc, rb, n_samples, jump); }
}

}  // namespace

void launch_form_factors(const DeviceScene& sc, const RadiosityBuffers& rb, const RadiosityParams& prm, const uint32_t* d_jump, hipStream_t s) {
    if (rb.n <= 0) return;
    const dim3 grid(rb.n);
    const bool deep = rb.bvh_depth > 30, rad0 = prm.num_iterations == 0;
    if (prm.use_monte_carlo && rb.row_jump) hipLaunchKernelGGL(ptmi_ff_row_jumps, grid, dim3(kBlock), 0, s, rb.row_jump, rb.n, d_jump);
    if (prm.use_monte_carlo) launch_ff1<true>(sc.has_quads != 0, deep, rad0, grid, s, sc, rb, prm.mc_samples, d_jump);
    else launch_ff1<false>(sc.has_quads != 0, deep, rad0, grid, s, sc, rb, 0, d_jump);
}

void launch_radiosity_iteration(const RadiosityBuffers& rb, int src, hipStream_t s) {
    if (rb.n <= 0) return;
    static const bool force_rows = getenv("PTMI_RADIOSITY_ROWS") != nullptr;        // A/B knob: the lane-per-row kernel
    // rows per 64-thread workgroup: 8 (24 summing lanes, n / 8 waves: 4 per CU at n = 8192) measured 83 us per step at n = 8192
    // against 91 us with 16 (48 summing lanes, 2 waves per CU): more waves, more row data in flight
    static const int tile_rows = getenv("PTMI_RADIOSITY_TILE_ROWS") ? atoi(getenv("PTMI_RADIOSITY_TILE_ROWS")) : 8;
    if ((rb.n & 3) == 0 && rb.n >= 64 && !force_rows) {                              // rows are 16-byte aligned only if n % 4 == 0
        if (tile_rows == 16) hipLaunchKernelGGL(ptmi_radiosity_iterate_tiled<16>, dim3((rb.n + 15) / 16), dim3(64), 0, s, rb, src);
        else hipLaunchKernelGGL(ptmi_radiosity_iterate_tiled<8>, dim3((rb.n + 7) / 8), dim3(64), 0, s, rb, src);
    }
    else
        hipLaunchKernelGGL(ptmi_radiosity_iterate, dim3((rb.n + kIterBlock - 1) / kIterBlock), dim3(kIterBlock), 0, s, rb, src);
}

void launch_cdf_records(int n, const void* d_src, int src_kind, float* d_out, hipStream_t s) {
    if (n <= 0) return;
    const dim3 grid(n), block(kBlock);
    if (src_kind == 0) hipLaunchKernelGGL(ptmi_cdf_records<0>, grid, block, 0, s, d_src, d_out, n);
    else if (src_kind == 1) hipLaunchKernelGGL(ptmi_cdf_records<1>, grid, block, 0, s, d_src, d_out, n);
    else hipLaunchKernelGGL(ptmi_cdf_records<2>, grid, block, 0, s, d_src, d_out, n);
}

void launch_filter_pdfs(int n, const float* d_rgb, const float* d_counts, float* d_out_formfactor, float* d_out_radiosity,
                        bool bilateral, float sigma_spatial, float sigma_range, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(ptmi_filter_pdfs, dim3(n), dim3(kBlock), 0, s, d_rgb, d_counts, d_out_formfactor, d_out_radiosity,
                       bilateral ? 1 : 0, sigma_spatial, sigma_range);
}

void launch_radiosity_grid(const RadiosityBuffers& rb, const RadiosityParams& prm, hipStream_t s) {
    if (rb.n <= 0) return;
    hipLaunchKernelGGL(ptmi_radiosity_grid, dim3(rb.n), dim3(kBlock), 0, s, rb, prm);
}

}  // namespace ptmi
