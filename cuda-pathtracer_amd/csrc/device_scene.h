// device_scene.h — HBM/LDS data layout of the render path and the kernel launch API.
//
// Everything the kernels touch is an array of 16-byte vectors so that a wave
// reads/writes 1 KiB per instruction (global_load_dwordx4, ds_read_b128).
//
// SCENE (read-only, replicated on every GPU; staged into LDS when it fits)
//   nodes[2*i+0] = (bbox.min.xyz, bits(a))         a: inner -> SKIP index (first node after this subtree in pre-order),
//                                                     leaf  -> first primitive slot
//   nodes[2*i+1] = (bbox.max.xyz, bits(b))         b: inner -> right child (>0), leaf -> -prim_count (<0)
//   Nodes are numbered in pre-order (bvh.h:154-218), so an inner node's left child is always i+1 and the
//   reference's "push right, push left, pop" order (scene.h:101-105) is exactly "i+1 if the box is hit, else skip".
//   prims[S*k+0] = (v0.xyz, bits(type))            k = LEAF-ORDER slot (reference's bvh_indices applied on the host,
//   prims[S*k+1] = (v1-v0 | v10-v00, 0)              so a leaf's primitives are contiguous); S = 3 for triangle-only
//   prims[S*k+2] = (v2-v0 | v11-v00, 0)              scenes, 4 when quads are present.  Edges are precomputed with the
//   prims[S*k+3] = (      - | v01-v00, 0)            same float subtraction the reference performs per test.
//   mats[3*k+0]  = (normal.xyz, bits(load-order primitive index))
//   mats[3*k+1]  = (Kd.xyz, 0)
//   mats[3*k+2]  = (Ke.xyz, 0)
//
// PATH STATE (read+written once per launch per active pixel; slot = local pixel)
//   A[slot] = (ray.o.xyz, throughput.x)
//   B[slot] = (ray.d.xyz, throughput.y)
//   C[slot] = (L_sample.xyz, throughput.z)         L_sample = radiance of the sample in flight (integrator.h:204)
//   D[slot] = (color.xyz, bits(sample_idx << 8 | depth))   color = sum of finished samples (integrator.h:390)
//   E[slot] = xorwow v[0..3]                       RNG stream of the pixel: strictly sequential, which is why a
//   F[slot] = (xorwow v[4], weyl d)                pixel has exactly one path in flight
//   = 88 B per pixel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pt_vec.h"

namespace ptmi {

struct DeviceScene {
    const float4* nodes = nullptr;
    const float4* prims = nullptr;
    const float4* mats = nullptr;
    int n_nodes = 0, n_prims = 0;
    int prim_stride = 3;      // float4 per primitive
    int has_quads = 0;
    int stack_entries = 1;    // min(bvh depth + 1, 64); only the STACK traversal uses it
    int lds_resident = 0;     // 1: nodes+prims+mats are staged into LDS by every workgroup (LANE/STACK traversal)
    int traversal = 1;        // TraversalMode
    // Guided sampling (SamplingMode != BSDF): one PrecomputedCDF record per primitive in LOAD order
    // (render_config.h:24-31: pdf[256], row_sums[8], marginal_cdf[8], row_cdfs[256], total_weight, is_valid = 530
    // dwords = 2120 B), HBM/L2-resident, read per hit through the load-order index kept in mats[3k].w.  nullptr = none.
    const float* cdfs = nullptr;
    // Per-primitive radiosity (Triangle/Quad::radiosity, the radiosity solver's output in the reference): float4 per
    // LEAF-ORDER slot, read by ptmi_render_radiosity only.  nullptr = all zero (as after loading a scene).
    const float4* radiosity = nullptr;
    // TRAVERSAL_PACKED (scenes too large for LDS): the same tree in a cache-line-friendly order, see PACKED LAYOUT below.
    const float4* gnodes = nullptr;    // 2 float4 per record position
    int n_pos = 0;                     // record positions (incl. padding); a cursor >= n_pos ends the walk
    const float* gprims = nullptr;     // triangle-only scenes: 9 floats (v0, e1, e2) per leaf-order slot; nullptr: use prims
    const float4* gmats = nullptr;     // (normal.xyz, bits(row of mtab)) per leaf-order slot
    const float4* mtab = nullptr;      // distinct (Kd, Ke) pairs, 2 float4 per row
    const int* load_index = nullptr;   // leaf-order slot -> load-order primitive index (guided sampling reads cdfs through it)
    int n_top = 0, top_depth = 0;      // positions < n_top (the nodes of depth <= top_depth, level by level) are staged into LDS
    // TRAVERSAL_WIDE (opt-in fast tree, csrc/wide_bvh.h): 8-wide nodes of 128 bytes, breadth-first; per-triangle arrays in the
    // fast tree's own leaf order ("fast order")
    const uint4* wnodes = nullptr;     // 8 uint4 per node
    int w_nodes = 0, w_top = 0;        // nodes < w_top (whole levels) are staged into LDS by every workgroup
    int w_depth = 0;                   // entries of the walk's per-lane LDS stack = levels of the tree - 1
    const float* wprims = nullptr;     // 9 floats (v0, e1, e2) per fast-order triangle
    const float4* wmats = nullptr;     // (normal.xyz, bits(row of mtab)) per fast-order triangle
    const float4* wmtab = nullptr;     // distinct (Kd, Ke) pairs, 2 float4 per row
    const int* wload_index = nullptr;  // fast order -> load-order primitive index
    const int* wref_slot = nullptr;    // fast order -> reference leaf-order slot (equal-t hits keep the smaller one, scene.h:89-90)
    // TRAVERSAL_CERTIFIED: what the VERIFY phase and the fallback read (kernels.hip: bounce_wide_body, CERT)
    const uint4* wanc = nullptr;       // per reference leaf: its ancestors' pre-order node indices (leaf included), 4 per chunk, 0xffffffff pads
    const float4* wqprims = nullptr;   // scenes with quads: 64-byte records (v0 | type, e1, e2, e3) in the fast tree's order instead of wprims
    const float4* wcert = nullptr;     // kWideCertStride float4 per fast-order triangle, one 64-byte line per hit: (leaf box min, bits(first chunk << 5 |
                                       // chunks of the leaf's list)) (leaf box max, 0) (wmats' entry) (0)
    float w_big = 0.0f;                // the scene's scale (host/wide_bvh.h: WideBVH::scale); informational since the certificate takes eps from each box
    int w_cert_debug = 0;              // test hook of the solver's certified walk: 1 every blocked ray takes the ancestor chain, 2 the reference's walk
    const int* wfast_of_ref = nullptr; // reference leaf-order slot -> fast order
    float w_guard = 0.0f;              // the boxes are padded for ray origins with |coordinate| <= w_guard; others take the reference's walk
};
// PACKED LAYOUT.  On the 1 M-triangle scene the phased walk is bound by the rate at which L2 misses are served (it runs at
// the same speed with 2 and with 7 waves per SIMD, with and without half of its node reads moved to LDS): what counts is the
// number of distinct 128-byte lines a ray touches.  In pre-order a node's left child shares its line, its right child never
// does.  Here the two children of a node are ONE 64-byte pair, and pairs are laid out in pre-order of their parents, so a
// 128-byte line holds (L, R, LL, LR): the sibling that the walk comes back to after a subtree - or goes to at once when the
// left box is missed - arrived with the first fetch.  Measured on 1000 rays of that scene: 43 -> 28 distinct node lines per
// ray.  Links are explicit record positions (the tree and the visiting order are the reference's, scene.h:63-106):
//   inner  (min, bits(skip position)) (max, bits(left child position))      right child = left + 1; positions > 0
//   leaf   (min, bits(first slot << 3 | count)) (max, bits(~next position)) count <= 7; next = pre-order successor
//   position n_pos = "past the end".  Record 1 pads the root to a pair.
// Primitives of triangle-only scenes shrink from 48 to 36 bytes (three 12-byte loads) and the material record from 48 to 16
// bytes + a table of the distinct (Kd, Ke) pairs: fewer lines per leaf and per hit, and a smaller footprint in L2/MALL.
constexpr int kCdfDwords = 530, kCdfPdf = 0, kCdfRowSums = 256, kCdfMarginal = 264, kCdfRowCdfs = 272, kCdfTotal = 528, kCdfValid = 529;

// How ptmi_bounce walks the BVH.  All three visit the same nodes and primitives in the same order per ray.
enum TraversalMode {
    TRAVERSAL_SWEEP = 0,   // tiny scenes: the WAVE walks node indices 0..N-1 once, node/primitive data are wave-uniform
                           // scalar loads (SGPR operands), lanes whose own cursor equals the index take part
    TRAVERSAL_LANE = 1,    // general: every lane walks its own pre-order cursor (stackless, skip pointers)
    TRAVERSAL_STACK = 2,   // trees deeper than 62: explicit LDS stack with the reference's drop rule (scene.h:101-105)
    TRAVERSAL_PHASED = 3,  // large scenes: the stackless per-lane walk with wave-scheduled phases (ptmi_bounce_phased):
                           // lanes whose ray ends early shade and start their next segment instead of idling
    TRAVERSAL_PACKED = 4,  // scenes too large for LDS: PHASED over the packed layout (sibling pairs, 36-byte triangles)
    TRAVERSAL_WIDE = 5,    // opt-in (ptmi_config.fast_tree): PHASED over the 8-wide SAH tree of csrc/wide_bvh.h; the one walk
                           // that does NOT visit the reference's nodes - same triangles, same hit arithmetic, other boxes
    TRAVERSAL_CERTIFIED = 6,   // WIDE + a proof per ray that the reference's walk returns the same hit (slab tests of the hit
                           // leaf's ancestors in the reference's tree, tie detection), else the reference's walk for that ray: exact
};

struct PathState {
    float4* A = nullptr; float4* B = nullptr; float4* C = nullptr; float4* D = nullptr;
    uint4* E = nullptr; uint2* F = nullptr;
};

struct TileMap {              // local row -> global row (ptmi.h: ptmi_tiling)
    int width = 0, height = 0;
    int n_ranks = 1, rank = 0, row_block = 8;
    int local_rows = 0;
    int tile8 = 0;            // 1: consecutive slots enumerate 8x8 pixel tiles (needs width % 8 == 0 and local_rows % 8 == 0)
};

struct FrameParams {
    float cam_origin[3], cam_llc[3], cam_hor[3], cam_ver[3];
    int spp, max_depth;
    int sampling_mode;          // SamplingMode (render_config.h:38-44)
    float mis_bsdf_fraction;    // Scene::mis_bsdf_fraction (scene.h:217)
    // Frame batches (renderFrames): n_frames successive frames of unchanged scene / camera / config rendered as ONE pipelined
    // run - a pixel that has finished frame k banks its colour sum in frame_color[k * n_local + slot] and starts frame k + 1 at
    // once (its RNG stream simply goes on, as between two renderFrame() calls), so the stragglers of frame k share the GPU with
    // the head of frame k + 1.  The frame index lives above the sample index: sample_idx = frame << 16 | sample (batches need
    // spp < 65536, n_frames <= 256); a single frame keeps all 24 bits for the sample index (sample_mask 0xffffff).
    int n_frames = 1;
    unsigned int sample_mask = 0xffffffu;
    float4* frame_color = nullptr;     // (n_frames - 1) * n_local colour sums of the completed frames; the last frame's stays in D
    int n_local = 0;
};

struct StatCounters { unsigned long long rays, node_visits, prim_tests, hits, top_node_visits, cert_chain, cert_fallback; };

constexpr int kBlock = 256;
constexpr int kXorwowJumpWords = 32 * 160 * 5;   // 32 matrices T^(2^67 * 2^k), 160 rows of 5 words

// ---- launchers (kernels.hip) ------------------------------------------------
// render_init (integrator.h:274-280): seeds every local pixel's stream and clears its state.
void launch_render_init(const TileMap& tm, const PathState& st, const uint32_t* d_jump, uint64_t seed_base, hipStream_t s);
// First camera ray of the frame for every local pixel (sample 0 of the spp loop, integrator.h:383-387).
void launch_frame_begin(const TileMap& tm, const PathState& st, const FrameParams& fp, hipStream_t s);
// The hot kernel: advances every queued path by up to `segments` ray segments, regenerating camera rays when a
// sample ends; appends still-unfinished pixels to queue_out (wave ballot + prefix popcount, one atomic per wave).
// queue_in == nullptr means "all local pixels" (identity queue); n_in is then the pixel count.  count_in (device
// pointer, may be nullptr) holds the exact length of queue_in when the host only knows the upper bound n_in: the
// host can then enqueue launches ahead of the counts coming back.
int bounce_resident_waves(const DeviceScene& sc, const FrameParams& fp, bool stats, int n_cus);
// Count publishing: with host_count set, the last workgroup of the launch to finish stores the output count to *host_count
// (host-mapped pinned memory; the host polls it), zeroes *next_count (the next launch's counter) and *done_count (its own
// arrival counter): no fill or copy command between two launches of a chunk.  All three nullptr: the host resets and reads.
struct CountPublish { int* done_count = nullptr; int* next_count = nullptr; int* host_count = nullptr; };
// Scheduling of one launch; it does not change a result.  max_waves > 0 (with cursor: a device int the caller zeroed): the launch
// has at most max_waves waves, and a lane whose pixel has had its visit (all samples done, or `segments` segments in this launch)
// takes the next queue entry no lane has taken yet - one atomicAdd on *cursor per scheduling decision with ending lanes.  With
// max_waves = the waves the device holds at once, no wave of a launch waits for a slot and no lane idles while pixels are queued.
// (The 8-wide walks; other kernels ignore it.)
// cost / cost_max: per-pixel segment counts of this frame and their maximum (the walk adds to them at the end of every visit).
struct LaunchSchedule { int max_waves = 0; int* cursor = nullptr; unsigned int* cost = nullptr; unsigned int* cost_max = nullptr; };
// Launch order by cost: queue[0 .. n) = the entries of queue_in (nullptr: 0 .. n - 1) in `classes` (2 .. 256, a power of two)
// classes of descending cost[entry] (class width cost_max / classes); inside a class the entries keep their order up to the order in
// which the 256-entry workgroups of the pass reserve their ranges.  hist: 2 x 256 device ints of scratch.  The heaviest pixels of a
// frame are as long as the frame: started first, they end with it instead of after it.
void launch_order_by_cost(const int* queue_in, int n, const unsigned int* cost, const unsigned int* cost_max, int* hist, int* queue, int classes, hipStream_t s);
void launch_bounce(const DeviceScene& sc, const TileMap& tm, const PathState& st, const FrameParams& fp,
                   const int* queue_in, int n_in, const int* count_in, int* queue_out, int* count_out, int segments,
                   StatCounters* stats /* nullptr: counters compiled out */,
                   bool many_waves /* more waves than bounce_resident_waves(): an 8-wave build where there is one */, hipStream_t s,
                   const CountPublish& pub = CountPublish(), const LaunchSchedule& sched = LaunchSchedule());
// render_radiosity (integrator.h:460-504): the alternative "Radiosity" integrator of renderFrame (application.h:193-197):
// spp camera rays per pixel, first hit only, Le + per-primitive radiosity, sqrt gamma, 8-bit (+ float mean).
void launch_render_radiosity(const DeviceScene& sc, const TileMap& tm, const PathState& st, const FrameParams& fp,
                             unsigned char* rgb8, float* radiance, hipStream_t s);
// mean, Reinhard, gamma, 8-bit (integrator.h:393-407) + float radiance.  color_src: the per-slot colour sums to resolve
// (nullptr = the path state's D array, i.e. the frame just finished; else a slice of FrameParams::frame_color)
void launch_resolve(const TileMap& tm, const PathState& st, int spp, unsigned char* rgb8, float* radiance, hipStream_t s,
                    const float4* color_src = nullptr);

size_t bounce_lds_bytes(const DeviceScene& sc);
size_t bounce_lds_bytes_wide(const DeviceScene& sc);      // dynamic LDS of the 8-wide walks for this scene (top of the tree + stacks)

// ---- radiosity pre-pass (radiosity.hip) ----------------------------------------------------------------
// Load-order geometry the form-factor kernels sample (the traversal keeps using the leaf-order prims above):
//   geo[6*p+0] = (v0 | v00, bits(type))      geo[6*p+3] = (-  | v01, 0)
//   geo[6*p+1] = (v1 | v10, area)            geo[6*p+4] = (normal, 0)
//   geo[6*p+2] = (v2 | v11, area_ratio)      geo[6*p+5] = (centroid, 0)
// area / area_ratio / centroid are computed on the host with the float expressions of triangle.h:28, quad.h:31 and
// primitive.h:92-98, 161-170 (pure functions of the vertices).
constexpr int kGridRes = 16, kGridSize = 256;
struct RadiosityBuffers {
    const float4* geo = nullptr;
    const int* slot_of = nullptr;      // load-order index -> leaf-order slot (to skip source/target in the visibility test)
    const float4* bsdf = nullptr;      // (bsdf.xyz, 0), load order
    float4* radiosity = nullptr;       // (rgb, 0), load order
    float4* unshot[2] = {nullptr, nullptr};   // Jacobi double buffer
    float* form_factors = nullptr;     // n * n, row i = receiver
    unsigned int* grid = nullptr;      // n * 256 visible-sample counts (Triangle/Quad::grid; integers, so exact in any order)
    float4* rad_grid = nullptr;        // n * 256 (rgb, 0)  (Triangle/Quad::radiosity_grid)
    unsigned long long* rays = nullptr;   // shadow rays cast (1 counter)
    uint32_t* row_jump = nullptr;      // Monte-Carlo form factors: per receiver i the GF(2) matrix T^(2^67 * (i * n)) (160 rows x 5 words), the part
                                       // of a pair's XORWOW skip-ahead that all pairs of row i share (ptmi_ff_row_jumps); nullptr: not used
    int n = 0;
    int bvh_depth = 0;                 // > 30: the visibility walk keeps the reference's explicit stack and drop rule
    int fast_tree = 0;                 // the visibility walk: 0 the reference's, 1 DeviceScene's fast tree (opt-in), 2 the certified walk
};
struct RadiosityParams {
    int num_iterations, mc_samples, use_monte_carlo, enable_filtering, use_bilateral;
    float filter_sigma_spatial, filter_sigma_range;
};
// calculate_form_factors_mc_kernel / calculate_form_factors_kernel (form_factors.h:219-415) incl. formfactor_rand_init
// (:85-89, fused: a pair's XORWOW stream is derived where it is used) and initialize_directional_grids (:71-83)
void launch_form_factors(const DeviceScene& sc, const RadiosityBuffers& rb, const RadiosityParams& prm, const uint32_t* d_jump, hipStream_t s);
// radiosity_iteration_kernel (form_factors.h:441-465), Jacobi: reads unshot[src], writes unshot[1 - src]
void launch_radiosity_iteration(const RadiosityBuffers& rb, int src, hipStream_t s);
// update_radiosity_grid (form_factors.h:405-439) + the optional filter (grid_filter.h:103-165, 251-312)
void launch_radiosity_grid(const RadiosityBuffers& rb, const RadiosityParams& prm, hipStream_t s);
// precomputeCDFs / precomputeCDFsFromFiltered (application_state.h:492-585, 587-680) on the device: n records of kCdfDwords
// floats from (src_kind 0) float4 radiosity grids, (1) packed float3 grids, (2) pdf values, n * 256 entries each
void launch_cdf_records(int n, const void* d_src, int src_kind, float* d_out, hipStream_t s);
// filter_pdfs_for_primitives (grid_filter.h:420-507): d_rgb n*256*3 radiosity grids, d_counts n*256 count grids (or nullptr
// = zero); outputs n*256 filtered + per-primitive normalised pdfs
void launch_filter_pdfs(int n, const float* d_rgb, const float* d_counts, float* d_out_formfactor, float* d_out_radiosity,
                        bool bilateral, float sigma_spatial, float sigma_range, hipStream_t s);

// test hooks
void launch_debug_intersect(const DeviceScene& sc, int n, const float* o, const float* d, float t_min, float t_max,
                            int* hit, int* prim, float* t, float* p, float* nrm, hipStream_t s);
void launch_debug_intersect_wide(const DeviceScene& sc, int n, const float* o, const float* d, float t_min, float t_max,
                                 int* hit, int* prim, float* t, unsigned long long* counts /* 2, may be nullptr */, hipStream_t s);
void launch_debug_rng(const uint32_t* d_jump, uint64_t seed_base, int n_pixels, const int* pixels, int count, float* out, hipStream_t s);
void launch_debug_rcp(unsigned int first_bits, unsigned long long count, unsigned long long* d_out, hipStream_t s);
void launch_debug_cosine(int n, const float* normals, const float* u, const float* v, float* out, hipStream_t s);

}  // namespace ptmi
