// ref_harness.cpp — thin C-ABI driver around the REFERENCE's own geometry headers.
//
// TEST INFRASTRUCTURE ONLY.  Built by oracle/Makefile into oracle/_ref/ when
// /root/reference is present (development container); the reference sources are
// included BY PATH from where they lie and are never copied into this repo.
//
// What is compiled from the reference, unmodified: core/vector.h, core/ray.h,
// rendering/{triangle,quad,primitive,surface_interaction_record,bvh,scene,
// sensor}.h.  None of these includes a header this image lacks; the only
// CUDA-isms they use are the __host__/__device__ function qualifiers, which
// ROCm's own <hip/amd_detail/host_defines.h> (shipped in this image) defines
// as empty for a plain g++ host compile.  No stand-in headers are written.
//
// utils/file_manager.h (loadOBJ / loadMTL) is compiled too, in ref_obj_harness.cpp: it needs <cuda_runtime.h>, and
// NVIDIA's genuine header ships in this image's Triton wheel (oracle/Makefile finds it).
// What is NOT compilable here: rendering/grid.h (hence integrator.h) and rendering/form_factors.h include
// <curand_kernel.h> (closed NVIDIA library, nowhere in the image); rendering/grid_filter.h launches kernels with
// <<< >>> from host wrappers (no g++ can parse it); application_state.h includes GL/glew.h and GLFW.
// Those are restated in ptmi_oracle.c only.
#include <hip/amd_detail/host_defines.h>

#include <cfloat>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <iostream>
#include <sstream>
#include <vector>

#include "core/vector.h"
#include "core/ray.h"
#include "rendering/primitive.h"
#include "rendering/bvh.h"
#include "rendering/scene.h"
#include "rendering/sensor.h"

namespace {
struct CoutSilencer {   // BVHBuilder logs every step to std::cout (bvh.h:84-99)
    std::streambuf* old; std::ostringstream sink;
    CoutSilencer() : old(std::cout.rdbuf(sink.rdbuf())) {}
    ~CoutSilencer() { std::cout.rdbuf(old); }
};
struct RefScene {
    Primitive* prims = nullptr; int n = 0;
    std::vector<BVHNode> nodes; std::vector<int> indices;
    Scene scene_bvh, scene_linear;
};
Vector3f v3(const float* p) { return Vector3f(p[0], p[1], p[2]); }
}

extern "C" {

struct ref_hit { int hit; int prim; float t, p[3], n[3], bsdf[3], Le[3]; };

// Host flow of the reference: RenderState() constructs the Sensor with aspect 1
// (application_state.h:85, application.h:107-113), allocateBuffers() sets
// image size + aspect and calls updateCamera() (application_state.h:106-109),
// renderFrame() calls updateCameraOrbit() (application.h:161).
void ref_camera(const float* lookfrom, const float* lookat, const float* vup, float vfov,
                float yaw, float pitch, int orbit, int width, int height, float* out12) {
    Sensor s(v3(lookfrom), v3(lookat), v3(vup), vfov, 1.0f);
    s.image_width = width; s.image_height = height;
    s.aspect = (float)width / (float)height;
    s.updateCamera();
    if (orbit) { s.yaw = yaw; s.pitch = pitch; s.updateCameraOrbit(); }
    for (int k = 0; k < 3; k++) {
        out12[k] = s.origin[k]; out12[3 + k] = s.lower_left_corner[k];
        out12[6 + k] = s.horizontal[k]; out12[9 + k] = s.vertical[k];
    }
}

void ref_camera_ray(const float* frame12, float u, float v, float* o, float* d) {
    Sensor s(Vector3f(0, 0, 1), Vector3f(0, 0, 0), Vector3f(0, 1, 0), 40.0f, 1.0f);
    s.origin = v3(frame12); s.lower_left_corner = v3(frame12 + 3);
    s.horizontal = v3(frame12 + 6); s.vertical = v3(frame12 + 9);
    Ray r = s.get_ray(u, v);
    for (int k = 0; k < 3; k++) { o[k] = r.o[k]; d[k] = r.d[k]; }
}

void* ref_scene_create(int n, const int* type, const float* verts, const float* normal,
                       const float* bsdf, const float* Le) {
    CoutSilencer quiet;
    RefScene* rs = new RefScene;
    rs->n = n; rs->prims = new Primitive[n];
    for (int i = 0; i < n; i++) {
        const float* v = verts + (size_t)i * 12;
        if (type[i] == PRIM_TRIANGLE) {
            Triangle t(v3(v), v3(v + 3), v3(v + 6), v3(bsdf + 3 * i), v3(normal + 3 * i));   // as file_manager.h:212
            t.Le = v3(Le + 3 * i);
            rs->prims[i] = Primitive(t);
        } else {
            Quad q(v3(v), v3(v + 3), v3(v + 6), v3(v + 9), v3(bsdf + 3 * i));               // as file_manager.h:233
            q.normal = v3(normal + 3 * i);
            q.Le = v3(Le + 3 * i);
            rs->prims[i] = Primitive(q);
        }
    }
    BVHBuilder builder(rs->prims, n);
    rs->nodes = builder.nodes; rs->indices = builder.primitive_indices;
    rs->scene_bvh = Scene(rs->prims, n, rs->nodes.data(), rs->indices.data());
    rs->scene_linear = Scene(rs->prims, n);
    return rs;
}
void ref_scene_free(void* h) { RefScene* rs = (RefScene*)h; if (!rs) return; delete[] rs->prims; delete rs; }
int ref_scene_num_nodes(void* h) { return (int)((RefScene*)h)->nodes.size(); }

void ref_scene_get_bvh(void* h, float* bmin, float* bmax, int* left, int* right, int* count, int* indices) {
    RefScene* rs = (RefScene*)h;
    for (size_t i = 0; i < rs->nodes.size(); i++) {
        for (int c = 0; c < 3; c++) { bmin[i * 3 + c] = rs->nodes[i].bbox.min[c]; bmax[i * 3 + c] = rs->nodes[i].bbox.max[c]; }
        left[i] = rs->nodes[i].left_child; right[i] = rs->nodes[i].right_child; count[i] = rs->nodes[i].prim_count;
    }
    for (int i = 0; i < rs->n; i++) indices[i] = rs->indices[i];
}

// geometric normal / centroid as the reference constructors compute them
void ref_tri_geometric_normal(const float* v0, const float* v1, const float* v2, float* out) {
    Triangle t(v3(v0), v3(v1), v3(v2), Vector3f(0.5f, 0.5f, 0.5f));
    for (int k = 0; k < 3; k++) out[k] = t.normal[k];
}
void ref_quad_geometric_normal(const float* v00, const float* v10, const float* v11, const float* v01, float* out) {
    Quad q(v3(v00), v3(v10), v3(v11), v3(v01));
    for (int k = 0; k < 3; k++) out[k] = q.normal[k];
}
void ref_centroid(void* h, int i, float* out) {
    Vector3f c = ((RefScene*)h)->prims[i].centroid();
    for (int k = 0; k < 3; k++) out[k] = c[k];
}
// radiosity pre-pass helpers that live in primitive.h (:100-102 getArea, :150-191 sampleUniform)
float ref_area(void* h, int i) { return ((RefScene*)h)->prims[i].getArea(); }
void ref_sample_uniform(void* h, int i, float r1, float r2, float* out) {
    Vector3f p = ((RefScene*)h)->prims[i].sampleUniform(r1, r2);
    for (int k = 0; k < 3; k++) out[k] = p[k];
}
void ref_unit_vector(const float* v, float* out) {
    Vector3f u = unit_vector(v3(v));
    for (int k = 0; k < 3; k++) out[k] = u[k];
}

// Scene::intersect (scene.h:39-47) on n rays; directions are taken as given
// (Ray members set directly, bypassing the normalising constructor).
void ref_intersect(void* h, int n_rays, const float* o, const float* d, float t_min, float t_max,
                   int use_bvh, ref_hit* out) {
    RefScene* rs = (RefScene*)h;
    const Scene& sc = use_bvh ? rs->scene_bvh : rs->scene_linear;
    for (int i = 0; i < n_rays; i++) {
        Ray r; r.o = v3(o + 3 * i); r.d = v3(d + 3 * i);
        SurfaceInteractionRecord si;
        bool hit = sc.intersect(r, t_min, t_max, si);
        ref_hit& o_ = out[i];
        std::memset(&o_, 0, sizeof o_);
        o_.hit = hit ? 1 : 0; o_.prim = hit ? (int)(si.prim_ptr - rs->prims) : -1;
        if (hit) {
            o_.t = si.t;
            for (int k = 0; k < 3; k++) { o_.p[k] = si.p[k]; o_.n[k] = si.n[k]; o_.bsdf[k] = si.bsdf[k]; o_.Le[k] = si.Le[k]; }
        }
    }
}

// Layout and constants of rendering/render_config.h as the reference's own compiler pass sees them (the header comes in
// through triangle.h:7).  out: sizeof(PrecomputedCDF), offsetof pdf / row_sums / marginal_cdf / row_cdfs / total_weight /
// is_valid, GRID_RES, GRID_SIZE, GRID_HALF_RES, the five SamplingMode values, BLOCK_X, BLOCK_Y, and the sizes of the
// records SURVEY 2 quotes (Primitive, BVHNode, SurfaceInteractionRecord, Scene).  22 ints.
void ref_layout(int* out) {
    int k = 0;
    out[k++] = (int)sizeof(PrecomputedCDF);
    out[k++] = (int)offsetof(PrecomputedCDF, pdf);
    out[k++] = (int)offsetof(PrecomputedCDF, row_sums);
    out[k++] = (int)offsetof(PrecomputedCDF, marginal_cdf);
    out[k++] = (int)offsetof(PrecomputedCDF, row_cdfs);
    out[k++] = (int)offsetof(PrecomputedCDF, total_weight);
    out[k++] = (int)offsetof(PrecomputedCDF, is_valid);
    out[k++] = GRID_RES; out[k++] = GRID_SIZE; out[k++] = GRID_HALF_RES;
    out[k++] = (int)SamplingMode::SAMPLING_BSDF; out[k++] = (int)SamplingMode::SAMPLING_FORMFACTOR;
    out[k++] = (int)SamplingMode::SAMPLING_RADIOSITY; out[k++] = (int)SamplingMode::SAMPLING_MIS;
    out[k++] = (int)SamplingMode::SAMPLING_TOPK;
    out[k++] = BLOCK_X; out[k++] = BLOCK_Y;
    out[k++] = (int)sizeof(Primitive); out[k++] = (int)sizeof(BVHNode);
    out[k++] = (int)sizeof(SurfaceInteractionRecord); out[k++] = (int)sizeof(Scene);
    out[k++] = (int)sizeof(GRID_D_THETA);          // 8: M_PI is the double one, so the macro is a double expression
}
// The grid-step macros (render_config.h:14-17) evaluated by the reference's compiler pass, as doubles:
// GRID_INV_RES, GRID_INV_HALF_RES, GRID_D_THETA, GRID_D_PHI, M_PI, M_PI * 0.5f (the factor of grid.h:166, 268).
void ref_grid_constants(double* out) {
    out[0] = (double)GRID_INV_RES; out[1] = (double)GRID_INV_HALF_RES;
    out[2] = (double)GRID_D_THETA; out[3] = (double)GRID_D_PHI;
    out[4] = (double)M_PI; out[5] = (double)(M_PI * 0.5f);
}

}  // extern "C"
