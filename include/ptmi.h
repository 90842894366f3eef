/* ptmi.h — C ABI of libptmi.so, the MI355X-native drop-in for the per-pixel
 * path-tracing render loop of USharma002/CUDA-PathTracer.
 *
 * The reference has no plugin/FFI layer; its hot path is reached through three
 * host entry points that mutate one global ApplicationState `g_state`
 * (include/application_state.h:299-308, defined src/main.cu:51):
 *
 *   SceneState::loadScene(filename, subdivision_count, convert_quads)
 *                                   include/application_state.h:367-464
 *   RenderState::allocateBuffers() / updateResolution(w, h)
 *                                   include/application_state.h:91-129
 *   renderFrame()                   include/application.h:157-216
 *
 * A `ptmi_ctx` is that ApplicationState bound to one GPU; each function below
 * names the reference interface it replaces.  Plain pointers and sizes only;
 * every function returns 0 on success or a negative PTMI_E_* code and never
 * throws across the boundary; `ptmi_last_error()` gives the message of the
 * last failure on the calling thread.  One host thread per ctx.
 *
 * There is NO CPU fallback behind this ABI: without a usable HIP device
 * `ptmi_ctx_create` fails with PTMI_E_NO_DEVICE.
 */
#ifndef PTMI_H
#define PTMI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PTMI_OK            0
#define PTMI_E_INVALID    -1   /* bad argument / state (message says which) */
#define PTMI_E_IO         -2   /* scene file missing, unsupported extension, no primitives */
#define PTMI_E_NO_DEVICE  -3   /* no HIP device / wrong architecture */
#define PTMI_E_HIP        -4   /* a HIP runtime call failed (message carries hipGetErrorString) */
#define PTMI_E_NOMEM      -5   /* device or host allocation failed (reference: cudaMallocSafe throws, utils/cuda_utils.h:54-60) */
#define PTMI_E_DIST       -6   /* librccl could not be loaded or an nccl* call failed (message carries ncclGetErrorString) */

typedef struct ptmi_ctx ptmi_ctx;

/* Camera = the Sensor constructor arguments plus its public yaw/pitch members
 * (include/rendering/sensor.h:16-29, 84-86).  orbit = 1 reproduces renderFrame(),
 * which calls updateCameraOrbit() before every launch (application.h:161,
 * sensor.h:56-67): the origin is re-derived from yaw/pitch/radius, radius being
 * |origin - lookat| of the constructor call.  orbit = 0 uses `origin` as is
 * (Sensor::setPosition, sensor.h:69-72).  Defaults: AppConfig,
 * application_state.h:285-286, and sensor.h:24-25. */
typedef struct {
    float origin[3];      /* (0.5, 3, 8.5) */
    float lookat[3];      /* (0, 2.5, 0)   */
    float vup[3];         /* (0, 1, 0)     */
    float vfov_deg;       /* 40            */
    float yaw_deg;        /* 90            */
    float pitch_deg;      /* 0             */
    int   orbit;          /* 1             */
} ptmi_camera;

/* AppConfig members the path reads (application_state.h:262-293) plus the two
 * values the reference hard-codes: max_depth = 5 (integrator.h:389) and the RNG
 * seed base 2023 (integrator.h:279). */
typedef struct {
    int      spp;             /* AppConfig::spp, default 1 (UI range 1-1000, ui_windows.h:84) */
    int      max_depth;       /* 5 = reference behaviour */
    int      sampling_mode;   /* SamplingMode (render_config.h:38-44): 0 BSDF, 1 FORMFACTOR, 2 RADIOSITY, 4 TOPK (all three: pure
                               * grid sampling, integrator.h:242-257), 3 MIS (integrator.h:238-241).  Modes 1-4 need the records
                               * of ptmi_set_radiosity_grids; without them they fall back to cosine sampling exactly as the
                               * reference does for primitives with an empty grid (integrator.h:258-261) */
    uint64_t seed_base;       /* 2023 */
    /* scheduling knob, results are independent of it: ray segments each path
     * advances per kernel launch before state returns to HBM and the active
     * queue is compacted (1 = pure wavefront, large = megakernel-like). 0 = default: 32 for the small LDS-resident
     * scenes; scenes above 64 primitives render a frame as ONE launch of one wave per wave slot whose lanes take the
     * next queued pixel when theirs is through, in the order of the last frame's per-pixel cost */
    int      segments_per_launch;
    int      collect_stats;   /* 1: also count rays / node visits / primitive tests (slower build of the kernel) */
    /* scheduling knob, results are independent of it: which 64 pixels share a wave.  0 = default (64x1 row strips),
     * 1 = 8x8 pixel tiles (when width and this rank's row count are multiples of 8; measured -1 % with the sweep walk,
     * +1 % with the per-lane walk).  Takes effect at the next ptmi_update_resolution. */
    int      wave_tiles;
    /* scheduling knob, results are independent of it: number of independent pixel chunks, each driven through its own
     * HIP stream so that one chunk's kernel tail overlaps the other's body.  0 = default (2 for frames of >= 2^18 local
     * pixels, else 1), 1 = single stream.  Takes effect at the next ptmi_update_resolution. */
    int      streams;
    float    mis_bsdf_fraction;   /* AppConfig::mis_bsdf_fraction, 0.5 (application_state.h:292) */
    int      integrator;          /* AppConfig::current_integrator (application_state.h:50-53, 283): 0 = PathTracing,
                                   * 1 = Radiosity: renderFrame launches render_radiosity (integrator.h:460-504) - first hit,
                                   * Le + per-primitive radiosity, sqrt gamma - instead of the path tracer */
    int      download_image;      /* 1: ptmi_render_frame ends like renderFrame() does, with the D2H of the 8-bit image into the
                                   * ctx's pinned host image (RenderState::h_image, application.h:211) - read it through
                                   * ptmi_host_image.  0 (default): results stay on the device until asked for */
    int      fast_tree;           /* 0 (default): every ray's hit is the one the reference's walk over its own tree
                                   * (rendering/bvh.h:156-218) returns - frames bit-identical to the reference's.  (Scenes above
                                   * 64 primitives get there through an 8-wide binned-SAH tree plus a per-ray proof, else the
                                   * reference's walk for that ray: the certified walk)  1: opt-in, the same tree
                                   * WITHOUT the proof (last bullet; about 7 % faster): primitives, hit arithmetic, RNG
                                   * draws and shading are untouched; a ray's hit can differ only where two primitives are hit at
                                   * exactly the same t or where the reference's own slab test drops a grazing box
                                   * (cuda-pathtracer_amd/csrc/wide_bvh.h: 1 pixel of 4 M on a 1 M-triangle frame).
                                   * ptmi_run_radiosity_solver reads the switch too: its
                                   * visibility walk (form_factors.h:143-208) then skips the proof as well (n = 8192: 84 -> 71 ms;
                                   * 2 of 67 M form factors differ) */
} ptmi_config;

/* Framebuffer sharding (new in this implementation; the reference is single-GPU).
 * Rows are dealt to ranks in blocks of `row_block` rows, round-robin:
 * global row y belongs to rank (y / row_block) % n_ranks.  RNG streams are keyed
 * by the GLOBAL pixel index (integrator.h:278-279), so the union of all ranks'
 * rows is bit-identical to a single-GPU frame. */
typedef struct {
    int n_ranks;      /* 1 */
    int rank;         /* 0 */
    int row_block;    /* 8 */
} ptmi_tiling;

typedef struct {
    double   seconds;          /* device time of the frame (HIP events around all launches) */
    double   bounce_kernel_ms; /* summed duration of the dominant kernel (ptmi_bounce) */
    uint64_t bounce_launches;  /* launches issued, incl. the 1-2 queued behind the one that emptied the queue (they exit at once) */
    uint64_t path_visits;      /* sum over launches of queued pixels: each reads + writes its 88-byte state once */
    uint64_t samples;          /* local pixels * spp */
    uint64_t rays, node_visits, prim_tests, hits;   /* only with collect_stats */
    uint64_t top_node_visits;  /* of node_visits, those served from the LDS-staged top of the packed layout (large scenes; else 0) */
    uint64_t cert_chain;       /* certified walk (traversal 6), with collect_stats: hits whose one-fetch certificate did not apply and
                                * whose leaf's ancestors were slab-tested one by one */
    uint64_t cert_fallback;    /* ... and rays that were walked again by the reference's own walk (ties, grazed boxes, far origins) */
} ptmi_stats;

/* RadiosityState + the filter switches of AppConfig (application_state.h:207-209, 290-291), defaults in comments */
typedef struct {
    int   num_iterations;         /* "Radiosity Steps" 10 (UI range 0..50) */
    int   mc_samples;             /* 64 (UI range 4..256) */
    int   use_monte_carlo;        /* 1: calculate_form_factors_mc_kernel, 0: point-to-point calculate_form_factors_kernel */
    int   enable_filtering;       /* AppConfig::enable_grid_filtering 0 */
    int   use_bilateral;          /* 1 (0: gaussian) */
    float filter_sigma_spatial;   /* 1.5 */
    float filter_sigma_range;     /* 0.3 */
} ptmi_radiosity_params;

typedef struct {
    double   seconds;             /* device time of the whole solve */
    double   form_factor_ms, iteration_ms, grid_ms;
    uint64_t pairs;               /* n_prims^2 */
    uint64_t rays;                /* shadow rays cast by the form-factor kernel */
    uint64_t cert_chain;          /* certified walk: blocked rays whose proof needed the exact slab tests over the blocker's ancestors */
    uint64_t cert_fallback;       /* certified walk: rays that went through the reference's own walk */
    int      walk;                /* the visibility walk used: 0 the reference's tree, 1 the opt-in fast tree (ptmi_config.fast_tree),
                                   * 2 certified (fast tree + per-ray proof: the reference's answers; default from 256 triangles up) */
} ptmi_radiosity_stats;

/* ---- lifetime ------------------------------------------------------------ */
/* initializeApplication()'s device setup (application.h:92-148). device_id: HIP ordinal. */
int  ptmi_ctx_create(int device_id, ptmi_ctx** out);
void ptmi_ctx_destroy(ptmi_ctx*);                       /* SceneState::cleanup + buffer frees, application_state.h:466-490 */
const char* ptmi_last_error(void);
void ptmi_default_camera(ptmi_camera*);                 /* AppConfig() defaults */
void ptmi_default_config(ptmi_config*);
void ptmi_default_tiling(ptmi_tiling*);

/* ---- SceneState::loadScene (application_state.h:367-464) ------------------ */
/* Parses .obj/.mtl with the reference loader's rules (utils/file_manager.h:39-273),
 * optionally converts quads to triangles (application_state.h:323-365) and
 * subdivides (rendering/form_factors.h:475-574), builds the reference's BVH
 * (rendering/bvh.h:76-219) and uploads an SoA copy.  Replaces any previous scene. */
int ptmi_load_scene(ptmi_ctx*, const char* filename, int subdivision_count, int convert_quads);
/* Same, from arrays (procedural scenes).  type[i]: 0 triangle, 1 quad; verts: n*4*3
 * floats (4th vertex ignored for triangles); normal/bsdf/Le: n*3 floats. */
int ptmi_load_scene_arrays(ptmi_ctx*, int n, const int* type, const float* verts,
                           const float* normal, const float* bsdf, const float* Le);
int ptmi_scene_info(const ptmi_ctx*, int* n_prims, int* n_tris, int* n_quads, int* n_bvh_nodes, int* bvh_depth);
/* Host copies for inspection/tests; arrays sized from ptmi_scene_info. Any pointer may be NULL. */
int ptmi_scene_get_prims(const ptmi_ctx*, int* type, float* verts, float* normal, float* bsdf, float* Le);
int ptmi_scene_get_bvh(const ptmi_ctx*, float* bmin, float* bmax, int* left, int* right, int* count, int* indices);

/* ---- SceneState::precomputeCDFs (application_state.h:492-585) -----------------------------------------------------
 * Per-primitive 16x16 directional radiosity grids -> the 2120-byte PrecomputedCDF records the guided sampling modes
 * read (render_config.h:24-31).  rgb: n_prims * 256 * 3 floats in load order (n_prims must match the loaded scene);
 * NULL drops the records.  In the reference the grids come out of the radiosity pre-pass (form_factors.h), here out of
 * ptmi_run_radiosity_solver (which makes this call itself) or from the caller.  Loading another scene drops the records.
 * The reference's second path of initGridFromPrimitive (integrator.h:44-54: precomputed_cdfs == nullptr, the grid rebuilt
 * per hit from the primitive's raw radiosity grid) cannot arise here: records are built whenever grids are set. */
int ptmi_set_radiosity_grids(ptmi_ctx*, int n_prims, const float* rgb);
/* host copy of the records, n_prims * 530 dwords (is_valid as an int bit pattern); returns PTMI_E_INVALID if there are none */
int ptmi_get_precomputed_cdfs(const ptmi_ctx*, float* out);
/* Per-primitive radiosity (Triangle/Quad::radiosity; n_prims * 3 floats, load order; NULL = zero) shown by the Radiosity
 * integrator: the radiosity solver's output (ptmi_run_radiosity_solver sets it) or the caller's. */
int ptmi_set_radiosity(ptmi_ctx*, int n_prims, const float* rgb);

/* ---- RadiosityState::runSolver (application_state.h:688-777) + what the UI does right after it (ui_windows.h:185-192):
 * SceneState::precomputeCDFs() and the upload of the solved primitives.  Runs on the GPU: form factors for all
 * n_prims^2 pairs (form_factors.h:219-415), num_iterations Jacobi steps (:441-465), the directional radiosity grids
 * (:405-439) and the optional filter (grid_filter.h).  Afterwards the guided sampling modes and the Radiosity integrator
 * use the solution, exactly as if ptmi_set_radiosity_grids / ptmi_set_radiosity had been called with it.
 * The n_prims^2 form factors stay on the device (4 n^2 bytes) until the next solve, scene load or ctx destruction. */
/* "Apply Filter & Rebuild CDFs" (ui_windows.h:154-167): filter_pdfs_for_primitives (grid_filter.h:420-507: luminance of the
 * radiosity grids and the count grids through the 5x5 bilateral / gaussian float filter, each primitive normalised to
 * sum 1) + SceneState::precomputeCDFsFromFiltered (application_state.h:587-680): the guided sampling modes then use
 * records built from the filtered luminance.  Needs radiosity grids (a solver run or ptmi_set_radiosity_grids).
 * "Use Raw CDFs" (ui_windows.h:173-177) = ptmi_use_raw_cdfs: precomputeCDFs() again from the unfiltered grids. */
int ptmi_apply_grid_filter(ptmi_ctx*, int use_bilateral, float sigma_spatial, float sigma_range);
int ptmi_use_raw_cdfs(ptmi_ctx*);
/* d_filtered_formfactor / d_filtered_radiosity (application_state.h:160-161), n_prims * 256 floats each; either may be NULL */
int ptmi_get_filtered_pdfs(const ptmi_ctx*, float* formfactor, float* radiosity);
/* (The count grids - Triangle/Quad::grid - are only ever filled by ptmi_run_radiosity_solver; after ptmi_set_radiosity_grids
 * alone they are zero, as after loadScene in the reference, and the filtered form-factor pdf is all zero.) */
void ptmi_default_radiosity_params(ptmi_radiosity_params*);
int ptmi_run_radiosity_solver(ptmi_ctx*, const ptmi_radiosity_params*, ptmi_radiosity_stats* stats /* may be NULL */);
/* The solution in load order; any pointer may be NULL.  form_factors n*n (row = receiver), radiosity n*3, unshot n*3,
 * grid n*256 (visible-sample counts per direction cell, Triangle/Quad::grid), radiosity_grid n*256*3. */
int ptmi_get_radiosity_solution(const ptmi_ctx*, float* form_factors, float* radiosity, float* unshot, float* grid, float* radiosity_grid);

/* ---- RenderState::allocateBuffers / updateResolution (application_state.h:91-129)
 * (Re)allocates the image, path-state and RNG buffers for this rank's rows of a
 * width x height frame and runs render_init (integrator.h:274-280), i.e. RNG
 * streams are re-seeded on every call, as in the reference. */
int ptmi_update_resolution(ptmi_ctx*, int width, int height, const ptmi_tiling* tiling /* NULL = single GPU */);
int ptmi_set_camera(ptmi_ctx*, const ptmi_camera*);
int ptmi_set_config(ptmi_ctx*, const ptmi_config*);
/* Derived camera exactly as Sensor holds it after updateCameraOrbit()/updateCamera():
 * origin, lower_left_corner, horizontal, vertical (12 floats). */
int ptmi_get_camera_frame(const ptmi_ctx*, float* out12);
int ptmi_local_rows(const ptmi_ctx*, int* n_rows);      /* rows of the frame this rank renders */
/* global row index of each local row, n_rows ints */
int ptmi_local_row_map(const ptmi_ctx*, int* rows_out);

/* ---- renderFrame (application.h:157-216) ----------------------------------
 * Renders config.spp samples for every local pixel (RNG state carries over
 * from the previous frame, as in the reference) and leaves the results on the
 * device.  Asynchronous work is complete when it returns. */
int ptmi_render_frame(ptmi_ctx*, ptmi_stats* stats /* may be NULL */);

/* n_frames successive ptmi_render_frame calls with nothing changed in between (scene, camera, config), as ONE pipelined run:
 * a pixel that has finished frame k starts frame k + 1 at once - its RNG stream goes on exactly as it does between two
 * renderFrame() calls - so the few long-running pixels at the end of frame k share the GPU with the head of frame k + 1
 * instead of leaving it idle.  Frame by frame the images are bit-identical to n_frames separate calls.  Afterwards the image
 * buffers hold the LAST frame (what n calls would leave); ptmi_select_frame(j) puts frame j of the batch there instead
 * (then ptmi_read_image / ptmi_gather_frame / ptmi_host_image as usual).  n_frames in [1, 256]; a batch of more than one frame
 * needs spp < 65536 and the PathTracing integrator.  stats cover the whole batch. */
int ptmi_render_frames(ptmi_ctx*, int n_frames, ptmi_stats* stats /* may be NULL */);
int ptmi_select_frame(ptmi_ctx*, int frame /* 0 .. n_frames-1 of the last ptmi_render_frame(s) call */);

/* Results.  Both images hold this rank's rows only, local row-major
 * (local_rows x width x 3); local row r is global row ptmi_local_row_map()[r];
 * row 0 of the frame is the BOTTOM row (v = y/H from the lower-left corner,
 * integrator.h:384-385; the reference flips on PNG save, ui_windows.h:205).
 * rgb8   : the reference's output (mean -> Reinhard -> gamma 2.2 -> 8 bit, integrator.h:393-407)
 * radiance: mean linear radiance before tone mapping (float), an addition for parity checks. */
int ptmi_device_image(const ptmi_ctx*, void** d_rgb8, void** d_radiance);           /* device pointers (for RCCL gathers) */
int ptmi_read_image(const ptmi_ctx*, unsigned char* rgb8, float* radiance);          /* D2H of the local rows; either may be NULL */
/* D2D copy of the local rows into caller-owned DEVICE buffers (e.g. the send buffers of an RCCL gather); either may be NULL */
int ptmi_copy_image_device(const ptmi_ctx*, void* d_rgb8_dst, void* d_radiance_dst);

/* RenderState::h_image (application_state.h:77): the pinned host copy of this rank's 8-bit rows that ptmi_render_frame fills
 * when config.download_image is set; valid until the next ptmi_update_resolution / ptmi_ctx_destroy. */
int ptmi_host_image(const ptmi_ctx*, const unsigned char** rgb8, uint64_t* n_bytes);

/* ---- multi-GPU: the frame-end exchange (new in this implementation) ------------------------------------------
 * One process (or host thread) and one ctx per GPU; every rank loads the same scene and renders its rows
 * (ptmi_update_resolution with ptmi_tiling {n_ranks, rank, row_block}); no data-path collective.  The ONE exchange step
 * of a frame is ptmi_gather_frame: every rank's tile goes to dst_rank over RCCL (direct ncclSend per peer /
 * N-1 ncclRecv on dst in one group: xGMI is point-to-point, every peer pushes over its own link), exact tile sizes, and
 * a kernel on dst places the rows into the whole frame.  librccl.so.1 is loaded on the first ptmi_dist_* call
 * (environment PTMI_RCCL_LIB = a file to load in its place).
 *
 *   rank 0:  ptmi_dist_unique_id(id)  -> ship the 128 bytes to every rank (MPI, a file, torch.distributed, ...)
 *   all   :  ptmi_dist_init(ctx, id, n_ranks, rank)        (collective: ncclCommInitRank)
 *   per frame, all ranks: ptmi_render_frame(ctx, ...); ptmi_gather_frame(ctx, 0, PTMI_GATHER_RGB8)
 *   dst   :  ptmi_read_frame(ctx, rgb8, NULL)  or  ptmi_frame_device(...)
 *
 * ptmi_gather_frame only ENQUEUES (a stream of its own): frames are independent, so the next ptmi_render_frame may
 * start at once; its resolve pass waits on the device for the gather that still reads this rank's tile.
 * ptmi_gather_wait / ptmi_read_frame / ptmi_dist_barrier wait for it. */
#define PTMI_UNIQUE_ID_BYTES 128
#define PTMI_GATHER_RGB8     1     /* the reference's output, 3 B/pixel */
#define PTMI_GATHER_RADIANCE 2     /* float mean radiance, 12 B/pixel */
int ptmi_dist_unique_id(void* out_id /* PTMI_UNIQUE_ID_BYTES */);
/* Between processes RCCL needs dmabuf IPC on this driver stack: export HSA_ENABLE_IPC_MODE_LEGACY=0 in the environment of every
 * rank BEFORE anything in the process initialises HIP (the library does not set it for you).
 * After an nccl* call has failed the communicator is marked failed: every later ptmi_dist_* / ptmi_gather_frame call returns
 * PTMI_E_DIST until ptmi_dist_finalize + ptmi_dist_init. */
int ptmi_dist_init(ptmi_ctx*, const void* id /* PTMI_UNIQUE_ID_BYTES */, int n_ranks, int rank);
int ptmi_dist_comm_count(ptmi_ctx*, int* n_ranks);       /* ncclCommCount: the ranks RCCL itself reports for this communicator */
int ptmi_dist_finalize(ptmi_ctx*);                       /* also done by ptmi_ctx_destroy */
int ptmi_gather_frame(ptmi_ctx*, int dst_rank, int what /* PTMI_GATHER_RGB8 | PTMI_GATHER_RADIANCE */);
int ptmi_gather_wait(ptmi_ctx*);
/* dst_rank only: the assembled width x height frame (row 0 = bottom), device pointers (NULL for a part never gathered) */
int ptmi_frame_device(const ptmi_ctx*, void** d_rgb8, void** d_radiance);
/* dst_rank only: waits for the gather, then D2H of the whole frame; either may be NULL */
int ptmi_read_frame(ptmi_ctx*, unsigned char* rgb8, float* radiance);
int ptmi_dist_barrier(ptmi_ctx*);                        /* all ranks; also drains this rank's gather stream */
int ptmi_dist_allreduce_max(ptmi_ctx*, double* value);   /* in/out: max over ranks (timing: max-over-ranks of a step time) */

/* ---- host-only halves (no device touched; usable without a GPU) ----------------
 * The parse/convert/subdivide/BVH half of loadScene and the camera/tiling arithmetic, for
 * inspection and for tests of the host logic. */
typedef struct ptmi_host_scene ptmi_host_scene;
int  ptmi_host_scene_load(const char* filename, int subdivision_count, int convert_quads, ptmi_host_scene** out);
int  ptmi_host_scene_from_arrays(int n, const int* type, const float* verts, const float* normal,
                                 const float* bsdf, const float* Le, ptmi_host_scene** out);
void ptmi_host_scene_free(ptmi_host_scene*);
int  ptmi_host_scene_info(const ptmi_host_scene*, int* n_prims, int* n_tris, int* n_quads, int* n_bvh_nodes, int* bvh_depth);
int  ptmi_host_scene_get_prims(const ptmi_host_scene*, int* type, float* verts, float* normal, float* bsdf, float* Le);
int  ptmi_host_scene_get_bvh(const ptmi_host_scene*, float* bmin, float* bmax, int* left, int* right, int* count, int* indices);
/* Sensor after allocateBuffers() + renderFrame()'s camera update for a width x height frame (12 floats). */
/* "Save PNG" (ui/ui_windows.h:195-210): 8-bit RGB file of a whole frame as ptmi_read_image returns it (row 0 = bottom);
 * rows are flipped on write like stbi_flip_vertically_on_write(1) does. */
int  ptmi_write_png(const char* path, int width, int height, const unsigned char* rgb8_bottom_up);
int  ptmi_host_camera_frame(const ptmi_camera*, int width, int height, float* out12);
/* Layout of one PrecomputedCDF record as ptmi_get_precomputed_cdfs returns it and as the kernels read it
 * (render_config.h:24-31): out[0] = bytes per record, out[1..6] = byte offsets of pdf, row_sums, marginal_cdf, row_cdfs,
 * total_weight, is_valid; out[7..9] = GRID_RES, GRID_SIZE, GRID_HALF_RES (render_config.h:7-9). */
int  ptmi_host_cdf_record_layout(int* out10);
/* rows of a `height`-row frame owned by `tiling->rank`; rows_out may be NULL to query the count only */
int  ptmi_host_local_row_map(int height, const ptmi_tiling* tiling, int* n_rows, int* rows_out);

/* ---- unit-test hooks: single stages of the path on the device ------------- */
/* The destination's row-placement step of ptmi_gather_frame alone: tiles in rank order with their exact sizes
 * (sum = width*height*3 elements) -> whole frame.  Either pair may be NULL. */
int ptmi_debug_place_tiles(ptmi_ctx*, int width, int height, int n_ranks, int row_block, const unsigned char* tiles_rgb8,
                           const float* tiles_radiance, unsigned char* out_rgb8, float* out_radiance);
/* Overrides how ptmi_bounce walks the BVH (results are identical in every mode): force_mode -1 = automatic,
 * 0 = wave-uniform sweep, 1 = per-lane stackless, 2 = explicit stack, 3 = per-lane with wave-scheduled phases,
 * 4 = 3 over the packed layout (sibling-pair node order, 36-byte triangles; only where that layout was built, else 3);
 * 6 = certified: the 8-wide tree of ptmi_config.fast_tree + a per-ray proof that the reference's walk returns the same hit,
 *     else the reference's walk for that ray (results identical, the node / test counters are its own) - the automatic choice
 *     of every scene above sweep_max_prims whose tree is no deeper than 62;
 * sweep_max_prims = largest scene (primitives)
 * the automatic choice still sweeps (default 64).  Trees deeper than 62 always use the stack walk.
 * out_mode (may be NULL) receives the mode now in effect for the loaded scene, or -1 without a scene. */
int ptmi_debug_set_traversal(ptmi_ctx*, int force_mode, int sweep_max_prims, int* out_mode);
/* The mode in effect for the loaded scene (-1 without one), nothing changed. */
int ptmi_debug_get_traversal(const ptmi_ctx*, int* out_mode);
/* The visibility walk of the radiosity pre-pass's form-factor kernel: force_walk -1 = automatic (certified from min_prims
 * triangles up, default 256; else the reference's), 0 = the reference's own tree, 2 = certified (the fast tree + a per-ray
 * proof that the reference's any-hit walk answers the same; triangle scenes no deeper than 30; test hooks: 3 / 4 = certified
 * with every blocked ray sent through the proof's second stage / through the reference's own walk).  Form factors are identical
 * either way; ptmi_config.fast_tree (no proof, tolerance mode) takes precedence.  Applies to the next ptmi_run_radiosity_solver. */
int ptmi_debug_set_solver_walk(ptmi_ctx*, int force_walk, int min_prims);
/* The packed layout of traversal mode 4 is built for scenes that do not fit LDS, have at least min_nodes BVH nodes (default
 * 8192), a tree no deeper than 62 and no leaf of more than 7 primitives.  Applies to the loaded scene at once and to later
 * loads; n_positions (may be NULL) receives the number of record positions built (0 = none).  Results do not depend on it. */
int ptmi_debug_set_packed_min_nodes(ptmi_ctx*, int min_nodes, int* n_positions);
/* How much of the packed tree every workgroup keeps in LDS: the nodes of depth <= D, D the largest depth whose levels fit
 * top_records 32-byte records (default 512 = 16 KB; 0 = none; at most 2048).  Applies to the loaded scene at once and to
 * later loads; n_top / top_depth (may be NULL) receive what was built.  Results do not depend on it. */
int ptmi_debug_set_packed_top(ptmi_ctx*, int top_records, int* n_top, int* top_depth);
/* Scene::intersect (scene.h:39-110) for n rays given as-is (no normalisation). out_*: n each; p/nrm 3n. */
int ptmi_debug_intersect(ptmi_ctx*, int n, const float* o, const float* d, float t_min, float t_max,
                         int* hit, int* prim, float* t, float* p, float* nrm);
/* The opt-in fast tree (ptmi_config.fast_tree; cuda-pathtracer_amd/csrc/wide_bvh.h).  Builder knobs: max_leaf 1..3 triangles per
 * leaf child (default 3), c_trav / c_tri = SAH cost of a box level / a triangle test (1, 1), top_nodes = whole levels kept in
 * LDS while they fit this many 128-byte nodes (80).  Rebuilds the loaded scene's fast tree at once and
 * applies to later loads; any out pointer may be NULL. */
int ptmi_debug_set_fast_tree(ptmi_ctx*, int max_leaf, float c_trav, float c_tri, int top_nodes, int* n_nodes, int* depth, int* n_top);
/* Closest hit through the fast tree for n rays as given (the walk of ptmi_bounce_wide).  prim: load-order index or -1;
 * counts (may be NULL): [0] node visits, [1] triangle tests, summed over the rays. */
int ptmi_debug_intersect_fast(ptmi_ctx*, int n, const float* o, const float* d, float t_min, float t_max,
                              int* hit, int* prim, float* t, uint64_t* counts);
/* Host-only halves of the same: build the fast tree of a host scene, and walk it on the CPU decision for decision as the
 * kernel does (tests of the builder without a GPU). */
int ptmi_host_fast_tree_build(ptmi_host_scene*, int max_leaf, float c_trav, float c_tri, int* n_nodes, int* depth, double* sah);
/* shape of the built tree: out[0..8] = nodes with that many children, out[9..12] = leaf children with 0..3 triangles */
int ptmi_host_fast_tree_stats(const ptmi_host_scene*, int* out13);
int ptmi_host_fast_tree_intersect(const ptmi_host_scene*, int n, const float* o, const float* d, float t_min, float t_max,
                                  int* prim, float* t, uint64_t* counts /* [0] node visits [1] triangle tests [2] deepest stack */);
/* render_init + curand_uniform: first `count` uniforms of pixel stream (seed_base+pixel, subsequence pixel). */
int ptmi_debug_rng(ptmi_ctx*, uint64_t seed_base, int n_pixels, const int* pixels, int count, float* out /* n_pixels*count */);
/* Compares the kernels' short reciprocal (1 v_rcp + 4 fma, used for Moller-Trumbore's 1/a) with the IEEE quotient for
 * the `count` consecutive float bit patterns starting at first_bits; reports how many differ and the first one. */
int ptmi_debug_rcp_check(ptmi_ctx*, uint32_t first_bits, uint64_t count, uint64_t* mismatches, uint32_t* first_bad_bits);
/* sampleCosineHemisphere (integrator.h:62-85) with explicit (u, v). */
int ptmi_debug_cosine_sample(ptmi_ctx*, int n, const float* normals, const float* u, const float* v, float* out_dirs);

#ifdef __cplusplus
}
#endif
#endif /* PTMI_H */
