/* ptmi_oracle.h — C interface of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (libptmi.so) never links, loads or calls it.
 */
#ifndef PTMI_ORACLE_H
#define PTMI_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct po_scene po_scene;

typedef struct {
    float origin[3], lookat[3], vup[3];
    float vfov_deg;
    float yaw_deg, pitch_deg;
    int orbit;            /* 1 = renderFrame() behaviour: updateCameraOrbit() before use */
} po_camera;

/* derived camera exactly as the reference's Sensor holds it */
typedef struct {
    float origin[3], lower_left_corner[3], horizontal[3], vertical[3];
} po_camera_frame;

typedef struct {
    double seconds;
    uint64_t samples, rays, node_visits, prim_tests, hits;
} po_stats;

typedef struct {
    int hit;        /* 0/1 */
    int prim;       /* index into the scene's primitive list (load order) */
    float t, p[3], n[3], bsdf[3], Le[3];
    int node_visits, prim_tests;
} po_hit;

/* scene ----------------------------------------------------------------- */
po_scene* po_scene_load(const char* path, int subdivision_count, int convert_quads);
/* type[i]: 0 triangle (v: 3 verts), 1 quad (v: 4 verts); verts: n*4*3 floats
 * (4th vertex ignored for triangles); normal/bsdf/Le: n*3 floats each. */
po_scene* po_scene_from_arrays(int n, const int* type, const float* verts,
                               const float* normal, const float* bsdf, const float* Le);
void po_scene_free(po_scene*);
int po_scene_num_prims(const po_scene*);
int po_scene_num_nodes(const po_scene*);
/* out arrays sized by the two counts above */
void po_scene_get_prims(const po_scene*, int* type, float* verts, float* normal, float* bsdf, float* Le);
void po_scene_get_bvh(const po_scene*, float* bmin, float* bmax, int* left, int* right, int* count, int* indices);

/* guided sampling inputs: per-primitive 16x16 radiosity grids, rgb = n_prims*256*3 floats in load
 * order (NULL removes them); precomputeCDFs (application_state.h:492-585) turns them into the 2120-byte records */
void po_scene_set_radiosity_grids(po_scene*, const float* rgb);
void po_scene_set_mis_fraction(po_scene*, float f);

/* radiosity pre-pass: RadiosityState::runSolver (application_state.h:688-777).  Runs the form-factor
 * kernel (Monte Carlo or point-to-point), num_iterations Jacobi steps, the directional radiosity grids (+ optional
 * filter), then does what the UI does after the solver (ui_windows.h:185-192): precomputeCDFs and the primitive
 * upload, i.e. the scene's radiosity and CDF records are replaced.  Outputs (any may be NULL), load order:
 * form factors n*n, radiosity n*3, unshot n*3, count grid n*256, radiosity grid n*256*3, shadow rays cast. */
typedef struct {
    int num_iterations;       /* RadiosityState::num_iterations, 10 (application_state.h:208) */
    int mc_samples;           /* 64 */
    int use_monte_carlo;      /* 1 */
    int enable_filtering;     /* AppConfig::enable_grid_filtering, 0 (application_state.h:290) */
    int use_bilateral;        /* 1 */
    float filter_sigma_spatial, filter_sigma_range;   /* 1.5, 0.3 */
} po_radiosity_params;
int po_radiosity_solve(po_scene*, const po_radiosity_params*, int n_threads, float* out_form_factors, float* out_radiosity,
                       float* out_unshot, float* out_grid, float* out_rad_grid, uint64_t* out_rays);
/* "Apply Filter & Rebuild CDFs" (ui_windows.h:154-167) on the scene's current grids; -1 without radiosity grids */
int po_scene_apply_grid_filter(po_scene*, int use_bilateral, float sigma_spatial, float sigma_range,
                               float* out_formfactor, float* out_radiosity);
int po_form_factor_rows(const po_scene*, const po_radiosity_params*, int n_threads, int n_rows, const int* rows,
                        float* out_ff, float* out_grid);
/* stage hooks for the tests */
float po_expf(float x);
void po_prim_geometry(const po_scene*, int i, float* area, float centroid[3]);
void po_prim_sample_uniform(const po_scene*, int i, float r1, float r2, float out[3]);
int po_direction_to_grid_index(const float dir[3], const float normal[3]);
int po_visibility_blocked(const po_scene*, const float o[3], const float d[3], float max_dist, int source_idx, int target_idx);
int po_scene_get_cdfs(const po_scene*, float* out /* n_prims * 530 floats */);
void po_cdf_layout(int out[10]);          /* sizeof + field offsets of the PrecomputedCDF restatement, GRID_RES / _SIZE / _HALF_RES */
void po_grid_constants(double out[6]);    /* GRID_INV_RES, GRID_INV_HALF_RES, GRID_D_THETA, GRID_D_PHI, M_PI, M_PI * 0.5f */

/* camera, rng, numerics -------------------------------------------------- */
void po_camera_frame_setup(const po_camera*, int width, int height, po_camera_frame* out);
void po_camera_ray(const po_camera_frame*, float u, float v, float o[3], float d[3]);
void po_rng_init(uint64_t seed, uint64_t subsequence, uint32_t state[6]);
float po_rng_uniform(uint32_t state[6]);
int po_rng_selftest(int log2n, const uint32_t v_in[5]);
void po_xorwow_next_raw(uint32_t state[6], int n, uint32_t* out);       /* n raw 32-bit draws (state: x[0..4], d) */
void po_xorwow_skip_subsequences(uint32_t v[5], uint64_t n);            /* n < 2^32 subsequences of 2^67 draws */
void po_sincosf(float x, float* s, float* c);
float po_powf(float x, float y);
float po_acosf(float x);
float po_atan2f(float y, float x);
void po_sample_cosine_hemisphere(const float n[3], float u, float v, float out[3]);

/* intersection ------------------------------------------------------------ */
void po_intersect(const po_scene*, const float o[3], const float d[3], float t_min, float t_max,
                  int use_bvh, po_hit* out);

/* render -------------------------------------------------------------------
 * Renders rows [y0, y1) of a width x height frame.  out_rgb8 / out_radiance
 * are FULL-frame buffers (width*height*3), row 0 = bottom, only the requested
 * rows are written.  rng_state: optional width*height*6 uint32 persistent
 * state; if NULL or reset_rng != 0 the state is (re)initialised as render_init
 * does.  n_threads <= 0: all cores. */
int po_render(const po_scene*, const po_camera*, int width, int height, int spp, int max_depth, int sampling_mode,
              uint64_t seed_base, int reset_rng, uint32_t* rng_state,
              int y0, int y1, int n_threads,
              unsigned char* out_rgb8, float* out_radiance, po_stats* stats);

/* render_radiosity (integrator.h:460-504): first-hit Le + per-primitive radiosity, sqrt gamma */
void po_scene_set_radiosity(po_scene*, const float* rgb /* n_prims*3, load order; NULL = zero */);
int po_render_radiosity(const po_scene*, const po_camera*, int width, int height, int spp,
                        uint64_t seed_base, int reset_rng, uint32_t* rng_state, int y0, int y1, int n_threads,
                        unsigned char* out_rgb8, float* out_radiance);

#ifdef __cplusplus
}
#endif
#endif
