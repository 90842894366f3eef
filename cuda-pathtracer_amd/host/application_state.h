// application_state.h — host-side mirror of the reference's state objects for the render path
// (include/application_state.h: RenderState :77-130, SceneState :136-256/367-490, AppConfig :262-293,
//  ApplicationState :299-308; renderFrame: include/application.h:157-216).
//
// Differences by design: no global g_state (a ptmi_ctx owns one ApplicationState per GPU), no GL/ImGui,
// no radiosity/grid-guiding members, device memory is SoA (csrc/device_scene.h) instead of the 4364-byte
// Primitive records, and the render loop is a queue-driven sequence of ptmi_bounce launches instead of
// one thread-per-pixel megakernel.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../csrc/device_scene.h"
#include "bvh.h"
#include "file_manager.h"
#include "pbrt_loader.h"
#include "sensor.h"
#include "wide_bvh.h"

namespace ptmi {

struct HipError : std::runtime_error {
    hipError_t code;
    HipError(hipError_t c, const std::string& what) : std::runtime_error(what), code(c) {}
};
struct IoError : std::runtime_error { using std::runtime_error::runtime_error; };
struct ArgError : std::runtime_error { using std::runtime_error::runtime_error; };
struct DistError : std::runtime_error { using std::runtime_error::runtime_error; };   // RCCL missing or an nccl* call failed

// cudaMallocSafe (utils/cuda_utils.h:54-60): throws on failure
void* hipMallocSafe(size_t bytes, const char* name);

enum class IntegratorType { PathTracing = 0, Radiosity = 1 };                       // application_state.h:50-53
enum class SamplingMode { SAMPLING_BSDF = 0, SAMPLING_FORMFACTOR = 1, SAMPLING_RADIOSITY = 2, SAMPLING_MIS = 3, SAMPLING_TOPK = 4 };   // render_config.h:38-44

struct AppConfig {                                   // application_state.h:262-293 (path-relevant members)
    int spp = 1;
    int max_depth = 5;                               // literal 5 in integrator.h:389
    SamplingMode sampling_mode = SamplingMode::SAMPLING_BSDF;
    uint64_t seed_base = 2023;                       // integrator.h:279
    f3 camera_origin = {0.5f, 3.0f, 8.5f}, look_at = {0.0f, 2.5f, 0.0f}, up = {0.0f, 1.0f, 0.0f};
    float fov = 40.0f;
    bool convert_quads_to_triangles = false;
    bool orbit = true;                               // renderFrame() always calls updateCameraOrbit()
    int segments_per_launch = 0;                     // 0 = default
    bool collect_stats = false;
    float mis_bsdf_fraction = 0.5f;                  // application_state.h:292 / scene.h:217
    IntegratorType current_integrator = IntegratorType::PathTracing;   // application_state.h:283
    bool fast_tree = false;                          // new: walk the opt-in 8-wide SAH tree instead of the reference's (csrc/wide_bvh.h)
};

struct SceneState {
    std::vector<Primitive> h_primitives;             // load order
    std::vector<BVHNode> bvh_nodes;                  // pre-order
    std::vector<int> bvh_indices;                    // leaf order -> load order
    int bvh_depth = 0;
    int num_tris = 0, num_quads = 0;
    std::string scene_file;

    float4 *d_nodes = nullptr, *d_prims = nullptr, *d_mats = nullptr;
    // packed layout for TRAVERSAL_PACKED (csrc/device_scene.h), built for scenes of at least packed_min_nodes nodes
    float4 *d_gnodes = nullptr, *d_gmats = nullptr, *d_mtab = nullptr;
    float* d_gprims = nullptr;
    int* d_load_index = nullptr;
    int packed_min_nodes = 8192;
    int packed_top_records = 512;                    // record positions of the packed layout kept in LDS (32 B each: 16 KB,
                                                     // 7 waves per SIMD stay resident); < 2: none
    // opt-in fast tree for TRAVERSAL_WIDE (csrc/wide_bvh.h), built at the first frame that asks for it (AppConfig::fast_tree)
    WideBVH h_wide;
    WideBVHParams wide_params;
    bool certified_default = true;                   // scenes above the sweep's 64 primitives (triangles and quads) walk CERTIFIED by default
    bool fast_declined = false;                      // the last buildFast() failed: the automatic choices do not ask again for this scene
    static constexpr int kWideMaxLevels = 24;        // deepest 8-wide tree the walk's LDS stack takes (buildFast)
    int wide_top_nodes = 80;                         // whole levels of the fast tree kept in LDS while they fit this many nodes (128 B each)
    uint4* d_wnodes = nullptr;
    float* d_wprims = nullptr;
    float4 *d_wmats = nullptr, *d_wmtab = nullptr;
    int *d_wload_index = nullptr, *d_wref_slot = nullptr;
    uint4* d_wanc = nullptr;                         // TRAVERSAL_CERTIFIED: ancestor lists of the reference's leaves, see buildFast
    float4* d_wcert = nullptr;
    float4* d_wqprims = nullptr;
    int* d_wfast_of_ref = nullptr;
    void buildFast();                                // host build + upload (triangles and quads); throws ArgError when the builder or the walk's LDS budget declines the scene
    void freeFast();
    bool fastReady() const { return d_wnodes != nullptr; }
    DeviceScene d_scene;
    // guided sampling: per-primitive PrecomputedCDF records (render_config.h:24-31), load order
    std::vector<float> h_precomputed_cdfs;           // n_prims * kCdfDwords: host copy, fetched on demand (precomputedCdfsHost)
    const std::vector<float>& precomputedCdfsHost();
    // the records are built on the device (csrc/radiosity.hip: ptmi_cdf_records); d_src: see launch_cdf_records
    void precomputeCDFsDevice(const void* d_src, int src_kind, hipStream_t stream);
    float* d_precomputed_cdfs = nullptr;
    // precomputeCDFs — application_state.h:492-585, from per-primitive 16x16 radiosity grids (n_prims*256*3 floats, load
    // order; nullptr drops the records).  The radiosity solver that fills the grids in the reference is out of scope:
    // the grids are an input.
    void precomputeCDFs(const float* radiosity_grids_rgb);
    // what the radiosity grids / count grids currently are (kept for the filter button); load order
    std::vector<float> h_radiosity_grids;            // n_prims * 256 * 3, empty = none
    std::vector<float> h_count_grids;                // n_prims * 256 (Triangle/Quad::grid), empty = zero
    std::vector<float> h_filtered_formfactor, h_filtered_radiosity;   // n_prims * 256 each, after precomputeCDFsFromFiltered
    // "Apply Filter & Rebuild CDFs" (ui_windows.h:154-167): filter_pdfs_for_primitives (grid_filter.h:420-507) on the
    // device, then precomputeCDFsFromFiltered (application_state.h:587-680)
    void precomputeCDFsFromFiltered(bool use_bilateral, float sigma_spatial, float sigma_range, hipStream_t stream);
    // per-primitive radiosity for the Radiosity integrator (render_radiosity); n_prims*3 floats, load order; nullptr = zero
    float4* d_radiosity = nullptr;
    void setRadiosity(const float* rgb);
    int sweep_max_prims = 64;                        // scenes up to this many primitives use the wave-uniform sweep
    int force_traversal = -1;                        // test/benchmark override (TraversalMode), -1 = automatic

    // loadScene — application_state.h:367-464.  Throws IoError where the reference prints and returns.
    void loadScene(const std::string& filename, int subdivision_count, bool convert_quads);
    void loadSceneArrays(std::vector<Primitive> prims);
    // host half only (parse + convert + subdivide + BVH), no device involved
    void loadSceneHost(const std::string& filename, int subdivision_count, bool convert_quads);
    void loadSceneArraysHost(std::vector<Primitive> prims);
    void chooseTraversal();                          // picks d_scene.traversal from size/depth and the two knobs below
    void buildPacked();                              // (re)builds the packed layout if the scene qualifies; chooseTraversal() after it
    void cleanup();                                  // application_state.h:466-490
    ~SceneState() { cleanup(); }

private:
    void buildBVH();                                 // RayTracingManager::buildAccelStructure (ray_tracing_backend.h:81-129)
    void upload();                                   // SoA re-layout + H2D
    void freePacked();
};

// RadiosityState — application_state.h:200-216, 688-787: the radiosity pre-pass that produces what the guided sampling
// modes and the Radiosity integrator read.  Differences by design: the solver works on SoA arrays instead of a device
// copy of the 4364-byte Primitive records; no n^2 curandStates (streams are derived in the kernel); the reference's
// per-iteration update_radiosity_grid (+ filter), whose result every later iteration overwrites, runs once at the end;
// radiosity_history (no reader in the reference outside primitive.h) is not kept.
struct RadiosityStats { double seconds = 0, form_factor_ms = 0, iteration_ms = 0, grid_ms = 0; uint64_t pairs = 0, rays = 0, cert_chain = 0, cert_fallback = 0; int walk = 0; };
struct RadiosityState {
    int num_iterations = 10, mc_samples = 64;        // application_state.h:208
    bool use_monte_carlo = true, is_calculated = false;
    int force_walk = -1;                             // new (debug): -1 automatic, 0 the reference's visibility walk, 2 the certified walk (3 / 4: its chain / its fallback for every blocked ray)
    int cert_min_prims = 256;                        // new: automatic choice takes the certified walk from this many triangles up
    RadiosityBuffers d;                              // device arrays (owned)
    int final_unshot = 0;                            // which d.unshot[] holds the last iteration's unshot radiosity
    // results on the host, load order (filled by runSolver)
    std::vector<float> h_radiosity, h_unshot, h_radiosity_grid;   // n*3, n*3, n*256*3
    std::vector<float> h_grid;                                     // n*256 visible-sample counts
    bool grids_are_scene_grids = false;                            // the scene's radiosity grids are still this solution's
    bool host_grids_current = false;                               // h_radiosity_grid / h_grid are fetched on demand
    void fetchGrids();                                             // D2H of the two n*256 grids (33 + 8 MB at n = 8192)
    void runSolver(SceneState& scene, const uint32_t* d_jump, bool enable_filtering, bool use_bilateral,
                   float filter_sigma_spatial, float filter_sigma_range, hipStream_t stream, RadiosityStats* stats,
                   bool fast_tree = false /* AppConfig::fast_tree: the visibility walk through the opt-in 8-wide tree */);
    void readFormFactors(float* out) const;          // n*n floats, row = receiver
    void cleanup();                                  // application_state.h:779-787
    ~RadiosityState() { cleanup(); }
};

struct RenderState {
    int width = 800, height = 800;                   // DEFAULT_WIDTH/HEIGHT, application_state.h:42-43
    TileMap tile;                                    // rows of the frame this GPU renders
    Sensor h_camera;
    PathState d_state;                               // path state + RNG (replaces d_rand_state)
    unsigned char* d_image = nullptr;                // RGB8, local rows
    float* d_radiance = nullptr;                     // float3 mean radiance, local rows
    unsigned char* h_image = nullptr;                // RGB8, local rows, PINNED: RenderState::h_image (application_state.h:77, 99);
                                                     // renderFrame() ends with the D2H into it (application.h:211) when download_image
    bool download_image = false;
    // frame batches (renderFrames): colour sums of the batch's completed frames, (n_frames - 1) * n_local float4
    float4* d_frame_color = nullptr;
    size_t frame_color_frames = 0;                   // capacity in frames
    int batch_frames = 1, batch_spp = 0;             // what the last render call produced (selectFrame)
    hipEvent_t resolve_gate = nullptr;               // not owned: while set, the next resolve waits for it (a frame gather still
                                                     // reading this rank's tile, csrc/dist.hip)
    // The local pixels are dealt to kMaxChunks independent queues (256-slot blocks, round-robin), each driven through
    // its own stream: while one chunk's launch drains (kernel tail, state write-back burst) the other chunk's
    // workgroups keep the CUs busy.  Chunks share nothing but the read-only scene.
    static constexpr int kMaxChunks = 4, kCountRing = 4;
    struct Chunk {
        hipStream_t stream = nullptr;
        int n = 0;                                   // pixels in this chunk
        int *d_queue_init = nullptr, *d_queue[2] = {nullptr, nullptr}, *d_count = nullptr;
        int* h_count = nullptr;                      // pinned + coherent, kCountRing entries
        int* d_hcount = nullptr;                     // the same memory as the device addresses it (count publishing)
    } chunk[kMaxChunks];
    int n_chunks = 1;
    int want_chunks = 0;                             // 0 = automatic, else forced (scheduling knob)
    StatCounters* d_stats = nullptr;
    // launch order by cost (refill launches): the segments every pixel took in the last frame and in the one being rendered
    // (d_cost[frame & 1], n_local each, + their maxima), the scratch of the ordering pass and the ordered queue
    unsigned int* d_cost[2] = {nullptr, nullptr};
    unsigned int* d_cost_max = nullptr;              // [2]
    int* d_cost_hist = nullptr;                      // 512
    int* d_queue_ordered = nullptr;                  // n_local
    unsigned long long cost_frame = 0;               // frames rendered with cost accounting since allocateBuffers
    bool cost_valid = false;                         // d_cost[(cost_frame - 1) & 1] holds a whole frame's costs of the pixels as they are numbered now
    uint32_t* d_jump = nullptr;                      // XORWOW skip-ahead matrices (owned by the ctx, set before allocateBuffers)
    uint64_t seed_base = 2023;
    hipStream_t stream = nullptr;
    size_t n_local = 0;
    bool allow_tile8 = false;                        // scheduling knob (test/benchmark override)

    void allocateBuffers();                          // application_state.h:91-123 (+ render_init)
    void updateResolution(int w, int h, const TileMap* tiling);   // application_state.h:125-129
    void freeBuffers();
    void setupChunks(int n);                         // deals the local pixels to n queues (allocateBuffers; renderFrames when the walk's launch rule wants another count)
    void freeChunks();
    ~RenderState() { freeBuffers(); }
};

// Multi-GPU frame exchange (csrc/dist.hip): one RCCL communicator per ctx (one process per GPU), a stream of its own, and
// on the destination rank the staging + whole-frame buffers.  New in this implementation (the reference is single-GPU).
struct DistState {
    void* comm = nullptr;                            // ncclComm_t
    int n_ranks = 1, rank = 0;
    hipStream_t stream = nullptr;
    hipEvent_t gather_done = nullptr;
    bool pending = false;                            // a gather has been enqueued and not waited for
    bool failed = false;                             // an nccl* call failed: every later ptmi_dist_* call reports it until re-initialised
    int* d_token = nullptr;                          // barrier payload
    // destination side, (re)allocated per frame geometry
    unsigned char *d_stage_rgb = nullptr, *d_frame_rgb = nullptr;
    float *d_stage_rad = nullptr, *d_frame_rad = nullptr;
    long long* d_tile_offset = nullptr;              // per rank: first element of its tile in the staging buffers
    std::vector<long long> h_tile_offset;
    int frame_w = 0, frame_h = 0, frame_row_block = 0, frame_ranks = 0;
    bool have_rgb = false, have_rad = false;         // what the frame buffers hold

    void init(const void* unique_id128, int n_ranks, int rank);      // ncclCommInitRank
    void gatherFrame(const struct RenderState& r, int dst_rank, int what /* 1 rgb8 | 2 radiance */);   // enqueues, does not wait
    void wait();
    void barrier();                                  // all ranks (1-int all-reduce + stream sync)
    double allreduceMax(double v);
    int commCount() const;                           // ncclCommCount: the ranks RCCL itself reports
    void check() const;                              // throws unless initialised and healthy
    void finalize();
    ~DistState() { finalize(); }
private:
    void ensureFrame(const TileMap& tm, bool want_rgb, bool want_rad);
    void freeFrame();
};
void distUniqueId(void* out128);                     // ncclGetUniqueId
void debugPlaceTiles(int width, int height, int n_ranks, int row_block, const unsigned char* h_tiles_rgb, const float* h_tiles_rad,
                     unsigned char* out_rgb, float* out_rad, hipStream_t s);

struct FrameStats {
    double seconds = 0, bounce_kernel_ms = 0;
    uint64_t bounce_launches = 0, path_visits = 0, samples = 0, rays = 0, node_visits = 0, prim_tests = 0, hits = 0, top_node_visits = 0,
             cert_chain = 0, cert_fallback = 0;
};

struct ApplicationState {
    int device_id = 0;
    int n_cus = 0;                                   // compute units of the device
    RenderState render;
    SceneState scene;
    RadiosityState radiosity;
    DistState dist;
    AppConfig config;
    std::vector<uint32_t> h_jump;                    // 32 x 160 x 5 words
    std::vector<hipEvent_t> event_pool;

    explicit ApplicationState(int device);
    ~ApplicationState();
};

// renderFrame — application.h:157-216: camera update, launches, device sync.  Results stay on the device.
void renderFrame(ApplicationState& g_state, FrameStats* stats);
// n_frames successive renderFrame() calls with nothing changed in between, as one pipelined run (device_scene.h:
// FrameParams::n_frames).  Frame by frame the images are bit-identical to n_frames separate calls; afterwards the image
// buffers hold the LAST frame and selectFrame(j) resolves any frame of the batch into them.
void renderFrames(ApplicationState& g_state, int n_frames, FrameStats* stats);
void selectFrame(ApplicationState& g_state, int frame);

bool packBvhNodes(const std::vector<BVHNode>& bvh_nodes, int top_records, std::vector<float4>& g, int& n_pos, int& n_top, int& top_depth);

// GF(2) matrices T^(2^67 * 2^k), k = 0..31, of the xorwow state transition (cuRAND's subsequence skip-ahead).
std::vector<uint32_t> buildXorwowJumpMatrices();

std::vector<int> localRowMap(const TileMap& tm);
int countLocalRows(int height, int n_ranks, int rank, int row_block);

}  // namespace ptmi
