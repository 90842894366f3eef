// c_api.cpp — the C ABI of libptmi.so (include/ptmi.h) over the host-side state objects.
#include "../../include/ptmi.h"

#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <string>

#include "../host/application_state.h"

using namespace ptmi;

struct ptmi_ctx { ApplicationState app; explicit ptmi_ctx(int dev) : app(dev) {} };
struct ptmi_host_scene { SceneState scene; };

namespace {
thread_local std::string g_last_error;

template <class Fn>
int guarded(Fn&& fn) {
    try { fn(); return PTMI_OK; }
    catch (const HipError& e) {
        g_last_error = e.what();
        if (e.code == hipErrorNoDevice || e.code == hipErrorInvalidDevice) return PTMI_E_NO_DEVICE;
        if (e.code == hipErrorOutOfMemory) return PTMI_E_NOMEM;
        return PTMI_E_HIP;
    }
    catch (const IoError& e) { g_last_error = e.what(); return PTMI_E_IO; }
    catch (const ArgError& e) { g_last_error = e.what(); return PTMI_E_INVALID; }
    catch (const DistError& e) { g_last_error = e.what(); return PTMI_E_DIST; }
    catch (const std::bad_alloc&) { g_last_error = "host allocation failed"; return PTMI_E_NOMEM; }
    catch (const std::exception& e) { g_last_error = e.what(); return PTMI_E_INVALID; }
    catch (...) { g_last_error = "unknown error"; return PTMI_E_INVALID; }
}
void need(bool ok, const char* what) { if (!ok) throw ArgError(what); }
f3 v3(const float* p) { return mk3(p[0], p[1], p[2]); }

#define PTMI_HIP(call)                                                                                   \
    do {                                                                                                 \
        hipError_t e_ = (call);                                                                          \
        if (e_ != hipSuccess) throw HipError(e_, std::string(#call) + ": " + hipGetErrorString(e_));     \
    } while (0)

template <class T>
struct DevBuf {
    T* p = nullptr;
    explicit DevBuf(size_t n) { p = (T*)hipMallocSafe(n * sizeof(T), "debug buffer"); }
    ~DevBuf() { if (p) (void)hipFree(p); }
    void upload(const T* h, size_t n) { PTMI_HIP(hipMemcpy(p, h, n * sizeof(T), hipMemcpyHostToDevice)); }
    void download(T* h, size_t n) { PTMI_HIP(hipMemcpy(h, p, n * sizeof(T), hipMemcpyDeviceToHost)); }
};
}  // namespace

static std::vector<Primitive> prims_from_arrays(int n, const int* type, const float* verts, const float* normal,
                                                const float* bsdf, const float* Le);

extern "C" {

const char* ptmi_last_error(void) { return g_last_error.c_str(); }

void ptmi_default_camera(ptmi_camera* c) {
    const AppConfig d;
    c->origin[0] = d.camera_origin.x; c->origin[1] = d.camera_origin.y; c->origin[2] = d.camera_origin.z;
    c->lookat[0] = d.look_at.x; c->lookat[1] = d.look_at.y; c->lookat[2] = d.look_at.z;
    c->vup[0] = d.up.x; c->vup[1] = d.up.y; c->vup[2] = d.up.z;
    c->vfov_deg = d.fov; c->yaw_deg = 90.0f; c->pitch_deg = 0.0f; c->orbit = 1;
}
void ptmi_default_config(ptmi_config* c) {
    const AppConfig d;
    c->spp = d.spp; c->max_depth = d.max_depth; c->sampling_mode = (int)d.sampling_mode; c->seed_base = d.seed_base;
    c->segments_per_launch = 0; c->collect_stats = 0; c->wave_tiles = 0; c->streams = 0; c->mis_bsdf_fraction = d.mis_bsdf_fraction; c->integrator = 0;
    c->download_image = 0; c->fast_tree = 0;
}
void ptmi_default_tiling(ptmi_tiling* t) { t->n_ranks = 1; t->rank = 0; t->row_block = 8; }

int ptmi_ctx_create(int device_id, ptmi_ctx** out) {
    // (multi-process GPU work - the RCCL gather - needs HSA_ENABLE_IPC_MODE_LEGACY=0 on this driver stack.  The CALLER exports it
    // before anything initialises HIP: a setenv from here would come too late for a process that already has, and races with
    // getenv in other threads; include/ptmi.h says so at ptmi_dist_init)
    return guarded([&] { need(out != nullptr, "out is NULL"); *out = nullptr; *out = new ptmi_ctx(device_id); });
}
void ptmi_ctx_destroy(ptmi_ctx* c) { delete c; }

int ptmi_load_scene(ptmi_ctx* c, const char* filename, int subdivision_count, int convert_quads) {
    return guarded([&] {
        need(c && filename, "ctx/filename is NULL");
        need(subdivision_count >= 0 && subdivision_count <= 10, "subdivision_count must be in [0, 10]");   // UI range, ui_windows.h:213
        PTMI_HIP(hipSetDevice(c->app.device_id));
        c->app.config.convert_quads_to_triangles = convert_quads != 0;
        c->app.radiosity.cleanup();                       // a solution belongs to the scene it was computed for
        c->app.scene.loadScene(filename, subdivision_count, convert_quads != 0);
    });
}

int ptmi_load_scene_arrays(ptmi_ctx* c, int n, const int* type, const float* verts, const float* normal,
                           const float* bsdf, const float* Le) {
    return guarded([&] {
        need(c != nullptr, "ctx is NULL");
        PTMI_HIP(hipSetDevice(c->app.device_id));
        c->app.radiosity.cleanup();
        c->app.scene.loadSceneArrays(prims_from_arrays(n, type, verts, normal, bsdf, Le));
    });
}

static void scene_info(const SceneState& s, int* n_prims, int* n_tris, int* n_quads, int* n_bvh_nodes, int* bvh_depth) {
    if (n_prims) *n_prims = (int)s.h_primitives.size();
    if (n_tris) *n_tris = s.num_tris;
    if (n_quads) *n_quads = s.num_quads;
    if (n_bvh_nodes) *n_bvh_nodes = (int)s.bvh_nodes.size();
    if (bvh_depth) *bvh_depth = s.bvh_depth;
}
static void scene_get_prims(const SceneState& sc, int* type, float* verts, float* normal, float* bsdf, float* Le) {
    {
        const auto& ps = sc.h_primitives;
        for (size_t i = 0; i < ps.size(); i++) {
            const Primitive& p = ps[i];
            if (type) type[i] = (int)p.type;
            if (verts) for (int k = 0; k < 4; k++) { verts[(i * 4 + k) * 3] = p.v[k].x; verts[(i * 4 + k) * 3 + 1] = p.v[k].y; verts[(i * 4 + k) * 3 + 2] = p.v[k].z; }
            if (normal) { normal[i * 3] = p.normal.x; normal[i * 3 + 1] = p.normal.y; normal[i * 3 + 2] = p.normal.z; }
            if (bsdf) { bsdf[i * 3] = p.bsdf.x; bsdf[i * 3 + 1] = p.bsdf.y; bsdf[i * 3 + 2] = p.bsdf.z; }
            if (Le) { Le[i * 3] = p.Le.x; Le[i * 3 + 1] = p.Le.y; Le[i * 3 + 2] = p.Le.z; }
        }
    }
}
static void scene_get_bvh(const SceneState& s, float* bmin, float* bmax, int* left, int* right, int* count, int* indices) {
    {
        for (size_t i = 0; i < s.bvh_nodes.size(); i++) {
            const BVHNode& n = s.bvh_nodes[i];
            if (bmin) { bmin[i * 3] = n.bbox.min.x; bmin[i * 3 + 1] = n.bbox.min.y; bmin[i * 3 + 2] = n.bbox.min.z; }
            if (bmax) { bmax[i * 3] = n.bbox.max.x; bmax[i * 3 + 1] = n.bbox.max.y; bmax[i * 3 + 2] = n.bbox.max.z; }
            if (left) left[i] = n.left_child;
            if (right) right[i] = n.right_child;
            if (count) count[i] = n.prim_count;
        }
        if (indices) std::memcpy(indices, s.bvh_indices.data(), s.bvh_indices.size() * sizeof(int));
    }
}
static std::vector<Primitive> prims_from_arrays(int n, const int* type, const float* verts, const float* normal,
                                                const float* bsdf, const float* Le) {
    need(type && verts && normal && bsdf && Le, "NULL argument");
    need(n > 0, "n must be positive");
    std::vector<Primitive> prims((size_t)n);
    for (int i = 0; i < n; i++) {
        Primitive& p = prims[i];
        need(type[i] == 0 || type[i] == 1, "type must be 0 (triangle) or 1 (quad)");
        p.type = type[i] ? PRIM_QUAD : PRIM_TRIANGLE;
        for (int k = 0; k < 4; k++) p.v[k] = v3(verts + ((size_t)i * 4 + k) * 3);
        if (p.type == PRIM_TRIANGLE) p.v[3] = mk3(0, 0, 0);
        p.normal = v3(normal + (size_t)i * 3); p.bsdf = v3(bsdf + (size_t)i * 3); p.Le = v3(Le + (size_t)i * 3);
    }
    return prims;
}

int ptmi_scene_info(const ptmi_ctx* c, int* n_prims, int* n_tris, int* n_quads, int* n_bvh_nodes, int* bvh_depth) {
    return guarded([&] { need(c != nullptr, "ctx is NULL"); scene_info(c->app.scene, n_prims, n_tris, n_quads, n_bvh_nodes, bvh_depth); });
}
int ptmi_scene_get_prims(const ptmi_ctx* c, int* type, float* verts, float* normal, float* bsdf, float* Le) {
    return guarded([&] { need(c != nullptr, "ctx is NULL"); scene_get_prims(c->app.scene, type, verts, normal, bsdf, Le); });
}
int ptmi_scene_get_bvh(const ptmi_ctx* c, float* bmin, float* bmax, int* left, int* right, int* count, int* indices) {
    return guarded([&] { need(c != nullptr, "ctx is NULL"); scene_get_bvh(c->app.scene, bmin, bmax, left, right, count, indices); });
}

int ptmi_host_scene_load(const char* filename, int subdivision_count, int convert_quads, ptmi_host_scene** out) {
    return guarded([&] {
        need(filename && out, "NULL argument");
        need(subdivision_count >= 0 && subdivision_count <= 10, "subdivision_count must be in [0, 10]");
        *out = nullptr;
        std::unique_ptr<ptmi_host_scene> hs(new ptmi_host_scene);
        hs->scene.loadSceneHost(filename, subdivision_count, convert_quads != 0);
        *out = hs.release();
    });
}
int ptmi_host_scene_from_arrays(int n, const int* type, const float* verts, const float* normal,
                                const float* bsdf, const float* Le, ptmi_host_scene** out) {
    return guarded([&] {
        need(out != nullptr, "out is NULL");
        *out = nullptr;
        std::unique_ptr<ptmi_host_scene> hs(new ptmi_host_scene);
        hs->scene.loadSceneArraysHost(prims_from_arrays(n, type, verts, normal, bsdf, Le));
        *out = hs.release();
    });
}
void ptmi_host_scene_free(ptmi_host_scene* s) { delete s; }
int ptmi_host_scene_info(const ptmi_host_scene* s, int* n_prims, int* n_tris, int* n_quads, int* n_bvh_nodes, int* bvh_depth) {
    return guarded([&] { need(s != nullptr, "scene is NULL"); scene_info(s->scene, n_prims, n_tris, n_quads, n_bvh_nodes, bvh_depth); });
}
int ptmi_host_scene_get_prims(const ptmi_host_scene* s, int* type, float* verts, float* normal, float* bsdf, float* Le) {
    return guarded([&] { need(s != nullptr, "scene is NULL"); scene_get_prims(s->scene, type, verts, normal, bsdf, Le); });
}
int ptmi_host_scene_get_bvh(const ptmi_host_scene* s, float* bmin, float* bmax, int* left, int* right, int* count, int* indices) {
    return guarded([&] { need(s != nullptr, "scene is NULL"); scene_get_bvh(s->scene, bmin, bmax, left, right, count, indices); });
}
int ptmi_write_png(const char* path, int width, int height, const unsigned char* rgb8) {
    return guarded([&] {
        need(path && rgb8, "NULL argument");
        need(width > 0 && height > 0, "width/height must be positive");
        if (!writePNG(path, width, height, rgb8)) throw IoError(std::string("cannot write ") + path);
    });
}
int ptmi_host_camera_frame(const ptmi_camera* cam, int width, int height, float* out12) {
    return guarded([&] {
        need(cam && out12, "NULL argument");
        need(width > 0 && height > 0, "width and height must be positive");
        Sensor s(v3(cam->origin), v3(cam->lookat), v3(cam->vup), cam->vfov_deg, 1.0f);   // application.h:107-113
        s.yaw = cam->yaw_deg; s.pitch = cam->pitch_deg;
        s.image_width = width; s.image_height = height;
        s.aspect = (float)width / (float)height;                                        // application_state.h:108
        s.updateCamera();
        if (cam->orbit) s.updateCameraOrbit();                                          // application.h:161
        const CameraFrame f = s.frame();
        const f3 v[4] = {f.origin, f.lower_left_corner, f.horizontal, f.vertical};
        for (int i = 0; i < 4; i++) { out12[3 * i] = v[i].x; out12[3 * i + 1] = v[i].y; out12[3 * i + 2] = v[i].z; }
    });
}
int ptmi_host_cdf_record_layout(int* out10) {
    return guarded([&] {
        need(out10 != nullptr, "NULL argument");
        const int v[10] = {kCdfDwords * 4, kCdfPdf * 4, kCdfRowSums * 4, kCdfMarginal * 4, kCdfRowCdfs * 4, kCdfTotal * 4, kCdfValid * 4,
                           kGridRes, kGridSize, kGridRes / 2};
        std::memcpy(out10, v, sizeof v);
    });
}
int ptmi_host_local_row_map(int height, const ptmi_tiling* t, int* n_rows, int* rows_out) {
    return guarded([&] {
        need(t && n_rows, "NULL argument");
        need(height > 0 && t->n_ranks >= 1 && t->rank >= 0 && t->rank < t->n_ranks && t->row_block >= 1, "bad tiling");
        TileMap tm; tm.height = height; tm.n_ranks = t->n_ranks; tm.rank = t->rank; tm.row_block = t->row_block;
        const std::vector<int> rows = localRowMap(tm);
        *n_rows = (int)rows.size();
        if (rows_out) std::memcpy(rows_out, rows.data(), rows.size() * sizeof(int));
    });
}

int ptmi_set_radiosity_grids(ptmi_ctx* c, int n_prims, const float* rgb) {
    return guarded([&] {
        need(c != nullptr, "ctx is NULL");
        need(c->app.scene.d_nodes != nullptr, "no scene loaded");
        need(rgb == nullptr || n_prims == (int)c->app.scene.h_primitives.size(), "n_prims does not match the loaded scene");
        PTMI_HIP(hipSetDevice(c->app.device_id));
        c->app.radiosity.grids_are_scene_grids = false;               // the caller's grids replace the solver's
        c->app.scene.precomputeCDFs(rgb);
    });
}
int ptmi_set_radiosity(ptmi_ctx* c, int n_prims, const float* rgb) {
    return guarded([&] {
        need(c != nullptr, "ctx is NULL");
        need(c->app.scene.d_nodes != nullptr, "no scene loaded");
        need(rgb == nullptr || n_prims == (int)c->app.scene.h_primitives.size(), "n_prims does not match the loaded scene");
        PTMI_HIP(hipSetDevice(c->app.device_id));
        c->app.scene.setRadiosity(rgb);
    });
}
void ptmi_default_radiosity_params(ptmi_radiosity_params* p) {
    if (!p) return;
    p->num_iterations = 10; p->mc_samples = 64; p->use_monte_carlo = 1;
    p->enable_filtering = 0; p->use_bilateral = 1; p->filter_sigma_spatial = 1.5f; p->filter_sigma_range = 0.3f;
}
int ptmi_run_radiosity_solver(ptmi_ctx* c, const ptmi_radiosity_params* p, ptmi_radiosity_stats* stats) {
    return guarded([&] {
        need(c != nullptr, "ctx is NULL");
        need(c->app.scene.d_nodes != nullptr, "no scene loaded");
        ptmi_radiosity_params prm;
        if (p) prm = *p; else ptmi_default_radiosity_params(&prm);
        need(prm.num_iterations >= 0 && prm.num_iterations <= 1000, "num_iterations must be in [0, 1000]");
        need(prm.mc_samples >= 1 && prm.mc_samples <= 65536, "mc_samples must be in [1, 65536]");
        need(prm.filter_sigma_spatial > 0.0f && prm.filter_sigma_range > 0.0f, "filter sigmas must be positive");
        PTMI_HIP(hipSetDevice(c->app.device_id));
        RadiosityState& r = c->app.radiosity;
        r.num_iterations = prm.num_iterations; r.mc_samples = prm.mc_samples; r.use_monte_carlo = prm.use_monte_carlo != 0;
        RadiosityStats st;
        r.runSolver(c->app.scene, c->app.render.d_jump, prm.enable_filtering != 0, prm.use_bilateral != 0,
                    prm.filter_sigma_spatial, prm.filter_sigma_range, c->app.render.stream, &st, c->app.config.fast_tree);
        // the grids (Triangle/Quad::grid, ::radiosity_grid) stay on the device; the host copies the filter button and
        // "Use Raw CDFs" work from are fetched when one of them is pressed (sync_solver_grids)
        c->app.scene.h_count_grids.clear(); c->app.scene.h_radiosity_grids.clear();
        r.grids_are_scene_grids = true;
        c->app.scene.h_filtered_formfactor.clear(); c->app.scene.h_filtered_radiosity.clear();
        c->app.scene.precomputeCDFsDevice(r.d.rad_grid, 0, c->app.render.stream);   // ui_windows.h:189, from the solver's device grids
        c->app.scene.setRadiosity(r.h_radiosity.data());                 // ui_windows.h:190-191 (primitive upload)
        if (stats) {
            stats->seconds = st.seconds; stats->form_factor_ms = st.form_factor_ms; stats->iteration_ms = st.iteration_ms;
            stats->grid_ms = st.grid_ms; stats->pairs = st.pairs; stats->rays = st.rays;
            stats->cert_chain = st.cert_chain; stats->cert_fallback = st.cert_fallback; stats->walk = st.walk;
        }
    });
}
int ptmi_get_radiosity_solution(const ptmi_ctx* c, float* form_factors, float* radiosity, float* unshot, float* grid, float* radiosity_grid) {
    return guarded([&] {
        need(c != nullptr, "ctx is NULL");
        const RadiosityState& r = c->app.radiosity;
        need(r.is_calculated, "no radiosity solution (run ptmi_run_radiosity_solver first)");
        PTMI_HIP(hipSetDevice(c->app.device_id));
        if (grid || radiosity_grid) const_cast<RadiosityState&>(r).fetchGrids();
        if (form_factors) r.readFormFactors(form_factors);
        if (radiosity) std::memcpy(radiosity, r.h_radiosity.data(), r.h_radiosity.size() * sizeof(float));
        if (unshot) std::memcpy(unshot, r.h_unshot.data(), r.h_unshot.size() * sizeof(float));
        if (grid) std::memcpy(grid, r.h_grid.data(), r.h_grid.size() * sizeof(float));
        if (radiosity_grid) std::memcpy(radiosity_grid, r.h_radiosity_grid.data(), r.h_radiosity_grid.size() * sizeof(float));
    });
}
// after a solver run the scene's grids live on the device only: bring them to the host side the first time they are needed
static void sync_solver_grids(ptmi_ctx* c) {
    RadiosityState& r = c->app.radiosity;
    if (!r.is_calculated) return;
    if (c->app.scene.h_count_grids.empty()) { r.fetchGrids(); c->app.scene.h_count_grids = r.h_grid; }
    if (c->app.scene.h_radiosity_grids.empty() && r.grids_are_scene_grids) { r.fetchGrids(); c->app.scene.h_radiosity_grids = r.h_radiosity_grid; }
}
int ptmi_apply_grid_filter(ptmi_ctx* c, int use_bilateral, float sigma_spatial, float sigma_range) {
    return guarded([&] {
        need(c != nullptr, "ctx is NULL");
        need(c->app.scene.d_nodes != nullptr, "no scene loaded");
        need(sigma_spatial > 0.0f && sigma_range > 0.0f, "filter sigmas must be positive");
        PTMI_HIP(hipSetDevice(c->app.device_id));
        sync_solver_grids(c);
        c->app.scene.precomputeCDFsFromFiltered(use_bilateral != 0, sigma_spatial, sigma_range, c->app.render.stream);
    });
}
int ptmi_use_raw_cdfs(ptmi_ctx* c) {
    return guarded([&] {
        need(c != nullptr, "ctx is NULL");
        PTMI_HIP(hipSetDevice(c->app.device_id));
        sync_solver_grids(c);
        need(!c->app.scene.h_radiosity_grids.empty(), "the scene has no radiosity grids");
        c->app.scene.precomputeCDFs(c->app.scene.h_radiosity_grids.data());
    });
}
int ptmi_get_filtered_pdfs(const ptmi_ctx* c, float* formfactor, float* radiosity) {
    return guarded([&] {
        need(c != nullptr, "ctx is NULL");
        need(!c->app.scene.h_filtered_radiosity.empty(), "no filtered pdfs (call ptmi_apply_grid_filter first)");
        if (formfactor) std::memcpy(formfactor, c->app.scene.h_filtered_formfactor.data(), c->app.scene.h_filtered_formfactor.size() * sizeof(float));
        if (radiosity) std::memcpy(radiosity, c->app.scene.h_filtered_radiosity.data(), c->app.scene.h_filtered_radiosity.size() * sizeof(float));
    });
}
int ptmi_get_precomputed_cdfs(const ptmi_ctx* c, float* out) {
    return guarded([&] {
        need(c && out, "NULL argument");
        need(c->app.scene.d_precomputed_cdfs != nullptr, "no precomputed CDFs");
        PTMI_HIP(hipSetDevice(c->app.device_id));
        const std::vector<float>& h = const_cast<SceneState&>(c->app.scene).precomputedCdfsHost();
        std::memcpy(out, h.data(), h.size() * sizeof(float));
    });
}

int ptmi_update_resolution(ptmi_ctx* c, int width, int height, const ptmi_tiling* tiling) {
    return guarded([&] {
        need(c != nullptr, "ctx is NULL");
        PTMI_HIP(hipSetDevice(c->app.device_id));
        TileMap tm;
        if (tiling) { tm.n_ranks = tiling->n_ranks; tm.rank = tiling->rank; tm.row_block = tiling->row_block; }
        c->app.render.seed_base = c->app.config.seed_base;
        c->app.render.updateResolution(width, height, tiling ? &tm : nullptr);
    });
}

int ptmi_set_camera(ptmi_ctx* c, const ptmi_camera* cam) {
    return guarded([&] {
        need(c && cam, "NULL argument");
        AppConfig& cfg = c->app.config;
        cfg.camera_origin = v3(cam->origin); cfg.look_at = v3(cam->lookat); cfg.up = v3(cam->vup);
        cfg.fov = cam->vfov_deg; cfg.orbit = cam->orbit != 0;
        Sensor& s = c->app.render.h_camera;
        const int w = s.image_width, h = s.image_height; const float aspect = s.aspect;
        s = Sensor(cfg.camera_origin, cfg.look_at, cfg.up, cfg.fov, 1.0f);    // application.h:107-113
        s.yaw = cam->yaw_deg; s.pitch = cam->pitch_deg;
        s.image_width = w; s.image_height = h;
        if (w > 0) { s.aspect = aspect; s.updateCamera(); }
    });
}

int ptmi_set_config(ptmi_ctx* c, const ptmi_config* cfg) {
    return guarded([&] {
        need(c && cfg, "NULL argument");
        need(cfg->spp >= 1 && cfg->spp < (1 << 24), "spp must be in [1, 2^24)");
        need(cfg->max_depth >= 1 && cfg->max_depth <= 255, "max_depth must be in [1, 255]");
        need(cfg->sampling_mode >= 0 && cfg->sampling_mode <= 4, "sampling_mode must be 0..4 (render_config.h:38-44)");
        need(cfg->mis_bsdf_fraction >= 0.0f && cfg->mis_bsdf_fraction <= 1.0f, "mis_bsdf_fraction must be in [0, 1]");
        need(cfg->integrator == 0 || cfg->integrator == 1, "integrator must be 0 (PathTracing) or 1 (Radiosity)");
        need(cfg->segments_per_launch >= 0, "segments_per_launch must be >= 0");
        need(cfg->streams >= 0 && cfg->streams <= RenderState::kMaxChunks, "streams must be 0..4");
        AppConfig& a = c->app.config;                    // every check is above this line: a rejected config changes nothing
        a.spp = cfg->spp; a.max_depth = cfg->max_depth; a.sampling_mode = (SamplingMode)cfg->sampling_mode;
        a.mis_bsdf_fraction = cfg->mis_bsdf_fraction;
        a.current_integrator = cfg->integrator ? IntegratorType::Radiosity : IntegratorType::PathTracing;
        a.seed_base = cfg->seed_base; a.segments_per_launch = cfg->segments_per_launch; a.collect_stats = cfg->collect_stats != 0;
        c->app.render.allow_tile8 = cfg->wave_tiles != 0;
        c->app.render.want_chunks = cfg->streams;
        c->app.render.download_image = cfg->download_image != 0;
        a.fast_tree = cfg->fast_tree != 0;
    });
}

int ptmi_get_camera_frame(const ptmi_ctx* c, float* out12) {
    return guarded([&] {
        need(c && out12, "NULL argument");
        Sensor s = c->app.render.h_camera;                 // copy: what renderFrame() would derive
        if (c->app.config.orbit) s.updateCameraOrbit(); else s.updateCamera();
        const CameraFrame f = s.frame();
        const f3 v[4] = {f.origin, f.lower_left_corner, f.horizontal, f.vertical};
        for (int i = 0; i < 4; i++) { out12[3 * i] = v[i].x; out12[3 * i + 1] = v[i].y; out12[3 * i + 2] = v[i].z; }
    });
}

int ptmi_local_rows(const ptmi_ctx* c, int* n_rows) {
    return guarded([&] { need(c && n_rows, "NULL argument"); *n_rows = c->app.render.tile.local_rows; });
}
int ptmi_local_row_map(const ptmi_ctx* c, int* rows_out) {
    return guarded([&] {
        need(c && rows_out, "NULL argument");
        const std::vector<int> rows = localRowMap(c->app.render.tile);
        std::memcpy(rows_out, rows.data(), rows.size() * sizeof(int));
    });
}

int ptmi_render_frame(ptmi_ctx* c, ptmi_stats* stats) {
    return guarded([&] {
        need(c != nullptr, "ctx is NULL");
        FrameStats fs;
        renderFrame(c->app, stats ? &fs : nullptr);
        if (stats) {
            stats->seconds = fs.seconds; stats->bounce_kernel_ms = fs.bounce_kernel_ms; stats->bounce_launches = fs.bounce_launches; stats->path_visits = fs.path_visits;
            stats->samples = fs.samples; stats->rays = fs.rays; stats->node_visits = fs.node_visits;
            stats->prim_tests = fs.prim_tests; stats->hits = fs.hits; stats->top_node_visits = fs.top_node_visits;
            stats->cert_chain = fs.cert_chain; stats->cert_fallback = fs.cert_fallback;
        }
    });
}

int ptmi_render_frames(ptmi_ctx* c, int n_frames, ptmi_stats* stats) {
    return guarded([&] {
        need(c != nullptr, "ctx is NULL");
        FrameStats fs;
        renderFrames(c->app, n_frames, stats ? &fs : nullptr);
        if (stats) {
            stats->seconds = fs.seconds; stats->bounce_kernel_ms = fs.bounce_kernel_ms; stats->bounce_launches = fs.bounce_launches; stats->path_visits = fs.path_visits;
            stats->samples = fs.samples; stats->rays = fs.rays; stats->node_visits = fs.node_visits;
            stats->prim_tests = fs.prim_tests; stats->hits = fs.hits; stats->top_node_visits = fs.top_node_visits;
            stats->cert_chain = fs.cert_chain; stats->cert_fallback = fs.cert_fallback;
        }
    });
}
int ptmi_select_frame(ptmi_ctx* c, int frame) {
    return guarded([&] { need(c != nullptr, "ctx is NULL"); selectFrame(c->app, frame); });
}

int ptmi_device_image(const ptmi_ctx* c, void** d_rgb8, void** d_radiance) {
    return guarded([&] {
        need(c != nullptr, "ctx is NULL");
        need(c->app.render.d_image != nullptr, "buffers not allocated");
        if (d_rgb8) *d_rgb8 = c->app.render.d_image;
        if (d_radiance) *d_radiance = c->app.render.d_radiance;
    });
}

int ptmi_read_image(const ptmi_ctx* c, unsigned char* rgb8, float* radiance) {
    return guarded([&] {
        need(c != nullptr, "ctx is NULL");
        const RenderState& r = c->app.render;
        need(r.d_image != nullptr, "buffers not allocated");
        PTMI_HIP(hipSetDevice(c->app.device_id));
        if (rgb8) PTMI_HIP(hipMemcpy(rgb8, r.d_image, r.n_local * 3, hipMemcpyDeviceToHost));                       // application.h:211
        if (radiance) PTMI_HIP(hipMemcpy(radiance, r.d_radiance, r.n_local * 3 * sizeof(float), hipMemcpyDeviceToHost));
    });
}

int ptmi_copy_image_device(const ptmi_ctx* c, void* d_rgb8_dst, void* d_radiance_dst) {
    return guarded([&] {
        need(c != nullptr, "ctx is NULL");
        const RenderState& r = c->app.render;
        need(r.d_image != nullptr, "buffers not allocated");
        PTMI_HIP(hipSetDevice(c->app.device_id));
        if (d_rgb8_dst) PTMI_HIP(hipMemcpyAsync(d_rgb8_dst, r.d_image, r.n_local * 3, hipMemcpyDeviceToDevice, r.stream));
        if (d_radiance_dst) PTMI_HIP(hipMemcpyAsync(d_radiance_dst, r.d_radiance, r.n_local * 3 * sizeof(float), hipMemcpyDeviceToDevice, r.stream));
        PTMI_HIP(hipStreamSynchronize(r.stream));
    });
}

int ptmi_host_image(const ptmi_ctx* c, const unsigned char** rgb8, uint64_t* n_bytes) {
    return guarded([&] {
        need(c && rgb8, "NULL argument");
        need(c->app.render.h_image != nullptr, "buffers not allocated");
        *rgb8 = c->app.render.h_image;
        if (n_bytes) *n_bytes = (uint64_t)c->app.render.n_local * 3;
    });
}

// ---- multi-GPU frame exchange (csrc/dist.hip) ----
int ptmi_dist_unique_id(void* out_id) {
    return guarded([&] { need(out_id != nullptr, "NULL argument"); distUniqueId(out_id); });
}
int ptmi_dist_init(ptmi_ctx* c, const void* id, int n_ranks, int rank) {
    return guarded([&] {
        need(c && id, "NULL argument");
        PTMI_HIP(hipSetDevice(c->app.device_id));
        c->app.render.resolve_gate = nullptr;
        c->app.dist.init(id, n_ranks, rank);
    });
}
int ptmi_dist_finalize(ptmi_ctx* c) {
    return guarded([&] {
        need(c != nullptr, "ctx is NULL");
        PTMI_HIP(hipSetDevice(c->app.device_id));
        c->app.render.resolve_gate = nullptr;
        c->app.dist.finalize();
    });
}
int ptmi_gather_frame(ptmi_ctx* c, int dst_rank, int what) {
    return guarded([&] {
        need(c != nullptr, "ctx is NULL");
        PTMI_HIP(hipSetDevice(c->app.device_id));
        c->app.dist.gatherFrame(c->app.render, dst_rank, what);
        c->app.render.resolve_gate = c->app.dist.gather_done;
    });
}
int ptmi_gather_wait(ptmi_ctx* c) {
    return guarded([&] { need(c != nullptr, "ctx is NULL"); PTMI_HIP(hipSetDevice(c->app.device_id)); c->app.dist.wait(); });
}
int ptmi_frame_device(const ptmi_ctx* c, void** d_rgb8, void** d_radiance) {
    return guarded([&] {
        need(c != nullptr, "ctx is NULL");
        const DistState& d = c->app.dist;
        need(d.d_frame_rgb || d.d_frame_rad, "no gathered frame on this rank (ptmi_gather_frame with dst_rank = this rank first)");
        if (d_rgb8) *d_rgb8 = d.have_rgb ? d.d_frame_rgb : nullptr;
        if (d_radiance) *d_radiance = d.have_rad ? d.d_frame_rad : nullptr;
    });
}
int ptmi_read_frame(ptmi_ctx* c, unsigned char* rgb8, float* radiance) {
    return guarded([&] {
        need(c != nullptr, "ctx is NULL");
        DistState& d = c->app.dist;
        PTMI_HIP(hipSetDevice(c->app.device_id));
        need(!rgb8 || d.have_rgb, "the 8-bit frame was never gathered to this rank");
        need(!radiance || d.have_rad, "the radiance frame was never gathered to this rank");
        d.wait();
        const size_t n = (size_t)d.frame_w * (size_t)d.frame_h * 3;
        if (rgb8) PTMI_HIP(hipMemcpy(rgb8, d.d_frame_rgb, n, hipMemcpyDeviceToHost));
        if (radiance) PTMI_HIP(hipMemcpy(radiance, d.d_frame_rad, n * sizeof(float), hipMemcpyDeviceToHost));
    });
}
int ptmi_dist_barrier(ptmi_ctx* c) {
    return guarded([&] { need(c != nullptr, "ctx is NULL"); PTMI_HIP(hipSetDevice(c->app.device_id)); c->app.dist.barrier(); });
}
int ptmi_dist_allreduce_max(ptmi_ctx* c, double* value) {
    return guarded([&] {
        need(c && value, "NULL argument");
        PTMI_HIP(hipSetDevice(c->app.device_id));
        *value = c->app.dist.allreduceMax(*value);
    });
}
int ptmi_dist_comm_count(ptmi_ctx* c, int* n_ranks) {
    return guarded([&] { need(c && n_ranks, "NULL argument"); PTMI_HIP(hipSetDevice(c->app.device_id)); *n_ranks = c->app.dist.commCount(); });
}
int ptmi_debug_place_tiles(ptmi_ctx* c, int width, int height, int n_ranks, int row_block, const unsigned char* tiles_rgb8,
                           const float* tiles_radiance, unsigned char* out_rgb8, float* out_radiance) {
    return guarded([&] {
        need(c != nullptr, "ctx is NULL");
        PTMI_HIP(hipSetDevice(c->app.device_id));
        debugPlaceTiles(width, height, n_ranks, row_block, tiles_rgb8, tiles_radiance, out_rgb8, out_radiance, c->app.render.stream);
    });
}

int ptmi_debug_set_traversal(ptmi_ctx* c, int force_mode, int sweep_max_prims, int* out_mode) {
    return guarded([&] {
        need(c != nullptr, "ctx is NULL");
        need(force_mode >= -1 && (force_mode <= 4 || force_mode == TRAVERSAL_CERTIFIED), "force_mode must be -1..4 or 6");
        need(sweep_max_prims >= 0, "sweep_max_prims must be >= 0");
        SceneState& s = c->app.scene;
        s.force_traversal = force_mode; s.sweep_max_prims = sweep_max_prims;
        if (force_mode == TRAVERSAL_CERTIFIED && s.d_nodes && !s.fastReady()) {
            PTMI_HIP(hipSetDevice(c->app.device_id));
            s.buildFast();
        }
        s.chooseTraversal();
        if (out_mode) *out_mode = s.d_nodes ? s.d_scene.traversal : -1;
    });
}

int ptmi_debug_get_traversal(const ptmi_ctx* c, int* out_mode) {
    return guarded([&] {
        need(c != nullptr && out_mode != nullptr, "ctx / out_mode is NULL");
        *out_mode = c->app.scene.d_nodes ? c->app.scene.d_scene.traversal : -1;
    });
}

int ptmi_debug_set_solver_walk(ptmi_ctx* c, int force_walk, int min_prims) {
    return guarded([&] {
        need(c != nullptr, "ctx is NULL");
        need(force_walk == -1 || force_walk == 0 || (force_walk >= 2 && force_walk <= 4), "force_walk must be -1, 0, 2, 3 or 4");
        need(min_prims >= 0, "min_prims must be >= 0");
        c->app.radiosity.force_walk = force_walk; c->app.radiosity.cert_min_prims = min_prims;
    });
}

int ptmi_debug_set_packed_min_nodes(ptmi_ctx* c, int min_nodes, int* n_positions) {
    return guarded([&] {
        need(c != nullptr, "ctx is NULL");
        need(min_nodes >= 0, "min_nodes must be >= 0");
        PTMI_HIP(hipSetDevice(c->app.device_id));
        SceneState& s = c->app.scene;
        s.packed_min_nodes = min_nodes;
        if (s.d_nodes) { s.buildPacked(); s.chooseTraversal(); }
        if (n_positions) *n_positions = s.d_scene.n_pos;
    });
}

int ptmi_debug_set_packed_top(ptmi_ctx* c, int top_records, int* n_top, int* top_depth) {
    return guarded([&] {
        need(c != nullptr, "ctx is NULL");
        need(top_records >= 0 && top_records <= 2048, "top_records must be in [0, 2048] (64 KB of LDS)");
        PTMI_HIP(hipSetDevice(c->app.device_id));
        SceneState& s = c->app.scene;
        s.packed_top_records = top_records;
        if (s.d_nodes) { s.buildPacked(); s.chooseTraversal(); }
        if (n_top) *n_top = s.d_scene.n_top;
        if (top_depth) *top_depth = s.d_scene.top_depth;
    });
}

int ptmi_debug_set_fast_tree(ptmi_ctx* c, int max_leaf, float c_trav, float c_tri, int top_nodes, int* n_nodes, int* depth, int* n_top) {
    return guarded([&] {
        need(c != nullptr, "ctx is NULL");
        need(max_leaf >= 1 && max_leaf <= kWideMaxLeaf, "max_leaf must be 1..3");
        need(c_trav >= 0.0f && c_tri > 0.0f, "c_trav must be >= 0 and c_tri > 0");
        need(top_nodes >= 0 && top_nodes <= 600, "top_nodes must be in [0, 600] (75 KB of LDS)");
        PTMI_HIP(hipSetDevice(c->app.device_id));
        SceneState& s = c->app.scene;
        s.wide_params.max_leaf = max_leaf; s.wide_params.c_trav = c_trav; s.wide_params.c_tri = c_tri; s.wide_top_nodes = top_nodes;
        if (s.d_nodes) s.buildFast();
        if (n_nodes) *n_nodes = s.d_scene.w_nodes;
        if (depth) *depth = s.h_wide.depth;
        if (n_top) *n_top = s.d_scene.w_top;
    });
}

int ptmi_debug_intersect_fast(ptmi_ctx* c, int n, const float* o, const float* d, float t_min, float t_max,
                              int* hit, int* prim, float* t, uint64_t* counts) {
    return guarded([&] {
        need(c && o && d && hit && prim && t, "NULL argument");
        need(c->app.scene.d_nodes != nullptr, "no scene loaded");
        need(n > 0, "n must be positive");
        PTMI_HIP(hipSetDevice(c->app.device_id));
        if (!c->app.scene.fastReady()) c->app.scene.buildFast();
        DevBuf<float> d_o(3 * (size_t)n), d_d(3 * (size_t)n), d_t(n);
        DevBuf<int> d_hit(n), d_prim(n);
        DevBuf<unsigned long long> d_cnt(2);
        PTMI_HIP(hipMemset(d_cnt.p, 0, 2 * sizeof(unsigned long long)));
        d_o.upload(o, 3 * (size_t)n); d_d.upload(d, 3 * (size_t)n);
        launch_debug_intersect_wide(c->app.scene.d_scene, n, d_o.p, d_d.p, t_min, t_max, d_hit.p, d_prim.p, d_t.p, d_cnt.p, c->app.render.stream);
        PTMI_HIP(hipGetLastError());
        PTMI_HIP(hipStreamSynchronize(c->app.render.stream));
        d_hit.download(hit, n); d_prim.download(prim, n); d_t.download(t, n);
        if (counts) { unsigned long long h[2]; d_cnt.download(h, 2); counts[0] = h[0]; counts[1] = h[1]; }
    });
}

int ptmi_host_fast_tree_build(ptmi_host_scene* s, int max_leaf, float c_trav, float c_tri, int* n_nodes, int* depth, double* sah) {
    return guarded([&] {
        need(s != nullptr, "scene is NULL");
        need(max_leaf >= 1 && max_leaf <= kWideMaxLeaf, "max_leaf must be 1..3");
        need(c_trav >= 0.0f && c_tri > 0.0f, "c_trav must be >= 0 and c_tri > 0");
        SceneState& sc = s->scene;
        sc.wide_params.max_leaf = max_leaf; sc.wide_params.c_trav = c_trav; sc.wide_params.c_tri = c_tri;
        try { buildWideBVH(sc.h_primitives, sc.wide_params, sc.h_wide); }
        catch (const std::invalid_argument& e) { throw ArgError(e.what()); }
        if (n_nodes) *n_nodes = sc.h_wide.n_nodes;
        if (depth) *depth = sc.h_wide.depth;
        if (sah) *sah = sc.h_wide.sah;
    });
}
int ptmi_host_fast_tree_stats(const ptmi_host_scene* s, int* out13) {
    return guarded([&] {
        need(s && out13, "NULL argument");
        need(!s->scene.h_wide.empty(), "no fast tree built (ptmi_host_fast_tree_build)");
        for (int i = 0; i < 9; i++) out13[i] = s->scene.h_wide.fill_hist[i];
        for (int i = 0; i < 4; i++) out13[9 + i] = s->scene.h_wide.leaf_hist[i];
    });
}
int ptmi_host_fast_tree_intersect(const ptmi_host_scene* s, int n, const float* o, const float* d, float t_min, float t_max,
                                  int* prim, float* t, uint64_t* counts) {
    return guarded([&] {
        need(s && o && d && prim && t, "NULL argument");
        const SceneState& sc = s->scene;
        need(!sc.h_wide.empty(), "no fast tree built (ptmi_host_fast_tree_build)");
        std::vector<int> ref_slot(sc.h_primitives.size());
        for (size_t k = 0; k < sc.bvh_indices.size(); k++) ref_slot[(size_t)sc.bvh_indices[k]] = (int)k;
        WideWalkCounters cn;
        for (int i = 0; i < n; i++) {
            float th = 0.0f;
            prim[i] = wideIntersectHost(sc.h_wide, sc.h_primitives, ref_slot, v3(o + 3 * i), v3(d + 3 * i), t_min, t_max, th, &cn);
            t[i] = prim[i] >= 0 ? th : 0.0f;
        }
        if (counts) { counts[0] = cn.node_visits; counts[1] = cn.prim_tests; counts[2] = cn.max_stack; }
    });
}

int ptmi_debug_intersect(ptmi_ctx* c, int n, const float* o, const float* d, float t_min, float t_max,
                         int* hit, int* prim, float* t, float* p, float* nrm) {
    return guarded([&] {
        need(c && o && d && hit && prim && t && p && nrm, "NULL argument");
        need(c->app.scene.d_nodes != nullptr, "no scene loaded");
        need(n > 0, "n must be positive");
        PTMI_HIP(hipSetDevice(c->app.device_id));
        DevBuf<float> d_o(3 * (size_t)n), d_d(3 * (size_t)n), d_t(n), d_p(3 * (size_t)n), d_n(3 * (size_t)n);
        DevBuf<int> d_hit(n), d_prim(n);
        d_o.upload(o, 3 * (size_t)n); d_d.upload(d, 3 * (size_t)n);
        launch_debug_intersect(c->app.scene.d_scene, n, d_o.p, d_d.p, t_min, t_max, d_hit.p, d_prim.p, d_t.p, d_p.p, d_n.p, c->app.render.stream);
        PTMI_HIP(hipGetLastError());
        PTMI_HIP(hipStreamSynchronize(c->app.render.stream));
        d_hit.download(hit, n); d_prim.download(prim, n); d_t.download(t, n); d_p.download(p, 3 * (size_t)n); d_n.download(nrm, 3 * (size_t)n);
    });
}

int ptmi_debug_rng(ptmi_ctx* c, uint64_t seed_base, int n_pixels, const int* pixels, int count, float* out) {
    return guarded([&] {
        need(c && pixels && out, "NULL argument");
        need(n_pixels > 0 && count > 0, "n_pixels and count must be positive");
        PTMI_HIP(hipSetDevice(c->app.device_id));
        DevBuf<int> d_pix(n_pixels); DevBuf<float> d_out((size_t)n_pixels * count);
        d_pix.upload(pixels, n_pixels);
        launch_debug_rng(c->app.render.d_jump, seed_base, n_pixels, d_pix.p, count, d_out.p, c->app.render.stream);
        PTMI_HIP(hipGetLastError());
        PTMI_HIP(hipStreamSynchronize(c->app.render.stream));
        d_out.download(out, (size_t)n_pixels * count);
    });
}

int ptmi_debug_rcp_check(ptmi_ctx* c, uint32_t first_bits, uint64_t count, uint64_t* mismatches, uint32_t* first_bad_bits) {
    return guarded([&] {
        need(c && mismatches && first_bad_bits, "NULL argument");
        need(count > 0 && count <= (1ull << 32), "count must be in [1, 2^32]");
        PTMI_HIP(hipSetDevice(c->app.device_id));
        DevBuf<unsigned long long> d(2);
        const unsigned long long init[2] = {0ull, ~0ull};
        d.upload(init, 2);
        launch_debug_rcp(first_bits, count, d.p, c->app.render.stream);
        PTMI_HIP(hipGetLastError());
        PTMI_HIP(hipStreamSynchronize(c->app.render.stream));
        unsigned long long h[2];
        d.download(h, 2);
        *mismatches = h[0]; *first_bad_bits = h[0] ? (uint32_t)h[1] : 0u;
    });
}

int ptmi_debug_cosine_sample(ptmi_ctx* c, int n, const float* normals, const float* u, const float* v, float* out_dirs) {
    return guarded([&] {
        need(c && normals && u && v && out_dirs, "NULL argument");
        need(n > 0, "n must be positive");
        PTMI_HIP(hipSetDevice(c->app.device_id));
        DevBuf<float> d_n(3 * (size_t)n), d_u(n), d_v(n), d_o(3 * (size_t)n);
        d_n.upload(normals, 3 * (size_t)n); d_u.upload(u, n); d_v.upload(v, n);
        launch_debug_cosine(n, d_n.p, d_u.p, d_v.p, d_o.p, c->app.render.stream);
        PTMI_HIP(hipGetLastError());
        PTMI_HIP(hipStreamSynchronize(c->app.render.stream));
        d_o.download(out_dirs, 3 * (size_t)n);
    });
}

}  // extern "C"
