/* ptmi_oracle.c — CPU restatement of the reference's per-pixel render loop.
 *
 * TEST INFRASTRUCTURE ONLY: this file is the checker, never the product.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it.  libptmi.so (the product) has no CPU render path at all.
 *
 * Parity status (see DESIGN.md "Oracle"):
 *   - geometry layer (vector math, Ray, Triangle/Quad::intersect, centroid,
 *     BVHBuilder, Scene::intersect_bvh_optimized / intersect_linear, Sensor)
 *     is PINNED: tests/test_oracle_vs_ref.py compares this file bit-for-bit
 *     with the reference's own headers compiled from /root/reference by
 *     oracle/Makefile into oracle/_ref/ (g++, no stand-in headers).
 *   - OBJ/MTL loader, integrator(), sampleCosineHemisphere, the render kernel's
 *     tone-map and the cuRAND XORWOW generator cannot be compiled here
 *     (file_manager.h / grid.h include <cuda_runtime.h>, <curand_kernel.h>,
 *     absent from this image; no stand-ins are written).  They are restated
 *     below from the source text and pinned only by the known answers the
 *     survey recorded from the reference (primitive counts, BVH
 *     dump, camera-ray hits).  RNG parity with real cuRAND: UNPINNED.
 *
 * All file:line citations are relative to /root/reference/include/.
 * Compile with -ffp-contract=off: every float expression below is written in
 * the reference's evaluation order and must not be fused or re-associated.
 */
#define _GNU_SOURCE
#include "ptmi_oracle.h"
#include "../include/ptmi_math.h"

#include <ctype.h>
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------ */
/* core/vector.h                                                             */
/* ------------------------------------------------------------------------ */
typedef struct { float e[3]; } v3;

static inline v3 V(float x, float y, float z) { v3 r = {{x, y, z}}; return r; }
static inline v3 vadd(v3 a, v3 b) { return V(a.e[0] + b.e[0], a.e[1] + b.e[1], a.e[2] + b.e[2]); }   /* vector.h:148-153 */
static inline v3 vsub(v3 a, v3 b) { return V(a.e[0] - b.e[0], a.e[1] - b.e[1], a.e[2] - b.e[2]); }   /* vector.h:155-160 */
static inline v3 vmul(v3 a, v3 b) { return V(a.e[0] * b.e[0], a.e[1] * b.e[1], a.e[2] * b.e[2]); }   /* vector.h:162-167 */
static inline v3 vdivv(v3 a, v3 b) { return V(a.e[0] / b.e[0], a.e[1] / b.e[1], a.e[2] / b.e[2]); }  /* vector.h:169-174 */
static inline v3 vscale(float t, v3 v) { return V(t * v.e[0], t * v.e[1], t * v.e[2]); }              /* vector.h:177-187 */
static inline v3 vneg(v3 a) { return V(-a.e[0], -a.e[1], -a.e[2]); }                                  /* vector.h:56-60 */
/* vector.h:189-195 and :90-94: division by a scalar multiplies by T k = 1.0 / t
 * (binary64 quotient rounded to binary32). */
static inline float recip_via_double(float t) { return (float)(1.0 / (double)t); }
static inline v3 vdivs(v3 v, float t) { float k = recip_via_double(t); return V(v.e[0] * k, v.e[1] * k, v.e[2] * k); }
static inline float vdot(v3 a, v3 b) {                                                                /* vector.h:198-203 */
    float sum = 0; sum += a.e[0] * b.e[0]; sum += a.e[1] * b.e[1]; sum += a.e[2] * b.e[2]; return sum;
}
static inline float vlen2(v3 a) { float sum = 0; sum += a.e[0] * a.e[0]; sum += a.e[1] * a.e[1]; sum += a.e[2] * a.e[2]; return sum; } /* vector.h:97-101 */
/* vector.h:103-105: sqrt() of a float; whether it resolves to the float or the
 * double overload the rounded result is the same (sqrt double rounding is innocuous). */
static inline float vlen(v3 a) { return sqrtf(vlen2(a)); }
static inline v3 vunit(v3 v) { return vdivs(v, vlen(v)); }                                            /* vector.h:205-208 */
static inline v3 vcross(v3 a, v3 b) {                                                                 /* vector.h:211-218 */
    return V(a.e[1] * b.e[2] - a.e[2] * b.e[1],
             (-(a.e[0] * b.e[2] - a.e[2] * b.e[0])),
             a.e[0] * b.e[1] - a.e[1] * b.e[0]);
}

/* ------------------------------------------------------------------------ */
/* scene data                                                                */
/* ------------------------------------------------------------------------ */
enum { PRIM_TRIANGLE = 0, PRIM_QUAD = 1 };  /* primitive.h:15-18 */

typedef struct {
    int type;
    v3 v[4];            /* tri: v0 v1 v2 ; quad: v00 v10 v11 v01 */
    v3 bsdf, normal, Le;
} oprim;

typedef struct {        /* bvh.h:63-72 */
    v3 bmin, bmax;
    int left_child, right_child, prim_count;
} onode;

/* render_config.h:8-31: 16 x 16 directional grid, upper hemisphere = rows 0..7; PrecomputedCDF is 2120 bytes */
#define GRID_RES 16
#define GRID_SIZE (GRID_RES * GRID_RES)
#define GRID_HALF_RES (GRID_RES / 2)
#define GRID_INV_RES (1.0f / GRID_RES)
#define GRID_INV_HALF_RES (1.0f / GRID_HALF_RES)
#define GRID_D_THETA ((PTMI_PI_D * 0.5f) / GRID_HALF_RES)   /* double: M_PI is a double */
#define GRID_D_PHI (2.0f * PTMI_PI_D / GRID_RES)
typedef struct {
    float pdf[GRID_SIZE];
    float row_sums[GRID_HALF_RES];
    float marginal_cdf[GRID_HALF_RES];
    float row_cdfs[GRID_SIZE];
    float total_weight;
    int is_valid;
} ocdf;

struct po_scene {
    oprim* prims; int n_prims;
    onode* nodes; int n_nodes, cap_nodes;
    int* indices;
    ocdf* cdfs;               /* Scene::precomputed_cdfs (scene.h:205), NULL until radiosity grids are supplied */
    v3* rad_grid;             /* Triangle/Quad::radiosity_grid (n * 256), kept for the filter button (ui_windows.h:154-167) */
    float* count_grid;        /* Triangle/Quad::grid (n * 256), filled by the radiosity solver, NULL = zero */
    v3* radiosity;            /* Triangle/Quad::radiosity per primitive (triangle.h:103), NULL = all zero (as after loading) */
    float mis_bsdf_fraction;  /* Scene::mis_bsdf_fraction, 0.5 (scene.h:217) */
};

/* ------------------------------------------------------------------------ */
/* triangle.h / quad.h constructors used by the loader                       */
/* ------------------------------------------------------------------------ */
static oprim make_tri5(v3 v0, v3 v1, v3 v2, v3 bsdf, v3 normal) {   /* triangle.h:49-62 */
    oprim p; memset(&p, 0, sizeof p);
    p.type = PRIM_TRIANGLE; p.v[0] = v0; p.v[1] = v1; p.v[2] = v2; p.v[3] = V(0, 0, 0);
    p.bsdf = bsdf; p.normal = normal; p.Le = V(0, 0, 0);
    return p;
}
static oprim make_tri4(v3 v0, v3 v1, v3 v2, v3 bsdf) {              /* triangle.h:23-47: geometric normal */
    v3 e1 = vsub(v1, v0), e2 = vsub(v2, v0);
    return make_tri5(v0, v1, v2, bsdf, vunit(vcross(e1, e2)));
}
static oprim make_quad(v3 v00, v3 v10, v3 v11, v3 v01, v3 bsdf) {   /* quad.h:23-47 */
    oprim p; memset(&p, 0, sizeof p);
    p.type = PRIM_QUAD; p.v[0] = v00; p.v[1] = v10; p.v[2] = v11; p.v[3] = v01;
    v3 e1 = vsub(v10, v00), e2 = vsub(v01, v00);
    p.bsdf = bsdf; p.normal = vunit(vcross(e1, e2)); p.Le = V(0, 0, 0);
    return p;
}

/* ------------------------------------------------------------------------ */
/* utils/file_manager.h: loadMTL (39-79), loadOBJ (93-273)                    */
/* ------------------------------------------------------------------------ */
typedef struct { char name[256]; v3 bsdf, Le; } omat;
typedef struct { omat* m; int n, cap; } omatmap;

static void matmap_set(omatmap* mm, const char* name, const omat* val) {   /* std::map operator[] = */
    for (int i = 0; i < mm->n; i++)
        if (strcmp(mm->m[i].name, name) == 0) { mm->m[i].bsdf = val->bsdf; mm->m[i].Le = val->Le; return; }
    if (mm->n == mm->cap) { mm->cap = mm->cap ? mm->cap * 2 : 8; mm->m = (omat*)realloc(mm->m, sizeof(omat) * mm->cap); }
    mm->m[mm->n] = *val;
    snprintf(mm->m[mm->n].name, sizeof mm->m[mm->n].name, "%s", name);
    mm->n++;
}
static const omat* matmap_find(const omatmap* mm, const char* name) {
    for (int i = 0; i < mm->n; i++) if (strcmp(mm->m[i].name, name) == 0) return &mm->m[i];
    return NULL;
}

/* istream >> std::string: skip whitespace, read until whitespace */
static const char* next_token(const char* p, char* tok, size_t cap) {
    while (*p && isspace((unsigned char)*p)) p++;
    size_t n = 0;
    while (*p && !isspace((unsigned char)*p)) { if (n + 1 < cap) tok[n++] = *p; p++; }
    tok[n] = 0;
    return p;
}
/* istream >> float (three of them); returns 0 on failure like the stream's fail state */
static int read_floats(const char** pp, float* out, int n) {
    const char* p = *pp;
    for (int i = 0; i < n; i++) {
        while (*p && isspace((unsigned char)*p)) p++;
        char* end;
        float f = strtof(p, &end);
        if (end == p) { for (int j = i; j < n; j++) out[j] = 0.0f; *pp = p; return 0; }
        out[i] = f; p = end;
    }
    *pp = p; return 1;
}

static char* read_line(FILE* f, char** buf, size_t* cap) {   /* std::getline: strips '\n' only */
    ssize_t n = getline(buf, cap, f);
    if (n < 0) return NULL;
    if (n > 0 && (*buf)[n - 1] == '\n') (*buf)[n - 1] = 0;
    return *buf;
}

static void load_mtl(const char* filename, omatmap* out) {   /* file_manager.h:39-79 */
    out->n = 0;
    FILE* f = fopen(filename, "r");
    if (!f) { fprintf(stderr, "[oracle] Warning: Could not open MTL file: %s\n", filename); return; }
    char cur_name[256] = ""; omat cur; cur.bsdf = V(0.8f, 0.8f, 0.8f); cur.Le = V(0, 0, 0);   /* :27-30 */
    char* line = NULL; size_t cap = 0; char tok[256];
    while (read_line(f, &line, &cap)) {
        const char* p = next_token(line, tok, sizeof tok);
        if (strcmp(tok, "newmtl") == 0) {
            if (cur_name[0]) matmap_set(out, cur_name, &cur);
            next_token(p, cur_name, sizeof cur_name);
            cur.bsdf = V(0.8f, 0.8f, 0.8f); cur.Le = V(0, 0, 0);
        } else if (strcmp(tok, "Kd") == 0) {
            float c[3]; read_floats(&p, c, 3); cur.bsdf = V(c[0], c[1], c[2]);
        } else if (strcmp(tok, "Ke") == 0) {
            float c[3]; read_floats(&p, c, 3); cur.Le = V(c[0], c[1], c[2]);
        }
    }
    if (cur_name[0]) matmap_set(out, cur_name, &cur);
    free(line); fclose(f);
}

/* one face-vertex token: "v", "v/vt", "v//vn", "v/vt/vn" (file_manager.h:162-190).
 * size_t extraction accepts a leading sign and wraps, as num_get does. */
static int parse_face_token(const char* tok, size_t* v_out, size_t* vn_out) {
    const char* p = tok; char* end;
    size_t v = 0, vn = 0;
    if (!(isdigit((unsigned char)*p) || ((*p == '-' || *p == '+') && isdigit((unsigned char)p[1])))) return 0;
    v = (size_t)strtoull(p, &end, 10); p = end;
    if (*p == '/') {
        p++;
        if (*p == '/') {
            p++;
            if (isdigit((unsigned char)*p) || ((*p == '-' || *p == '+') && isdigit((unsigned char)p[1]))) vn = (size_t)strtoull(p, &end, 10);
        } else if (isdigit((unsigned char)*p) || ((*p == '-' || *p == '+') && isdigit((unsigned char)p[1]))) {
            (void)strtoull(p, &end, 10); p = end;   /* vt */
            if (*p == '/') {
                p++;
                if (isdigit((unsigned char)*p) || ((*p == '-' || *p == '+') && isdigit((unsigned char)p[1]))) vn = (size_t)strtoull(p, &end, 10);
            }
        }
    }
    *v_out = v; *vn_out = vn; return 1;
}

typedef struct { oprim* p; int n, cap; } primvec;
static void pv_push(primvec* pv, const oprim* p) {
    if (pv->n == pv->cap) { pv->cap = pv->cap ? pv->cap * 2 : 64; pv->p = (oprim*)realloc(pv->p, sizeof(oprim) * pv->cap); }
    pv->p[pv->n++] = *p;
}

static int load_obj(const char* obj_filename, primvec* out) {   /* file_manager.h:93-273 */
    FILE* f = fopen(obj_filename, "r");
    if (!f) { fprintf(stderr, "[oracle] Error: Cannot open OBJ file: %s\n", obj_filename); return 0; }
    char path_base[4096] = "";
    {   /* :100-104 */
        const char* s1 = strrchr(obj_filename, '/'); const char* s2 = strrchr(obj_filename, '\\');
        const char* s = s1 > s2 ? s1 : s2;
        if (s) { size_t n = (size_t)(s - obj_filename) + 1; if (n >= sizeof path_base) n = sizeof path_base - 1; memcpy(path_base, obj_filename, n); path_base[n] = 0; }
    }
    v3* verts = NULL; size_t nv = 0, capv = 0;
    v3* norms = NULL; size_t nn = 0, capn = 0;
    omatmap mats = {0, 0, 0};
    omat cur; cur.bsdf = V(0.8f, 0.8f, 0.8f); cur.Le = V(0, 0, 0); cur.name[0] = 0;
    char* line = NULL; size_t cap = 0; char tok[1024];
    while (read_line(f, &line, &cap)) {
        if (line[0] == 0 || line[0] == '#' || line[0] == 'o' || line[0] == 's') continue;   /* :120 */
        const char* p = next_token(line, tok, sizeof tok);
        if (strcmp(tok, "v") == 0) {
            float c[3]; if (!read_floats(&p, c, 3)) continue;
            if (nv == capv) { capv = capv ? capv * 2 : 256; verts = (v3*)realloc(verts, sizeof(v3) * capv); }
            verts[nv++] = V(c[0], c[1], c[2]);
        } else if (strcmp(tok, "vn") == 0) {
            float c[3]; if (!read_floats(&p, c, 3)) continue;
            if (nn == capn) { capn = capn ? capn * 2 : 256; norms = (v3*)realloc(norms, sizeof(v3) * capn); }
            norms[nn++] = vunit(V(c[0], c[1], c[2]));   /* :141 */
        } else if (strcmp(tok, "mtllib") == 0) {
            char name[1024], full[8192];
            next_token(p, name, sizeof name);
            snprintf(full, sizeof full, "%s%s", path_base, name);
            load_mtl(full, &mats);
        } else if (strcmp(tok, "usemtl") == 0) {
            char name[256]; next_token(p, name, sizeof name);
            const omat* m = matmap_find(&mats, name);
            if (m) cur = *m; else { cur.bsdf = V(0.8f, 0.8f, 0.8f); cur.Le = V(0, 0, 0); }   /* :151-156 */
        } else if (strcmp(tok, "f") == 0) {
            size_t vi[64], ni[64]; int cnt = 0;
            for (;;) {
                p = next_token(p, tok, sizeof tok);
                if (!tok[0]) break;
                size_t v, vn;
                if (!parse_face_token(tok, &v, &vn)) continue;   /* :167-170: warn, skip token */
                if (cnt < 64) { vi[cnt] = v; ni[cnt] = vn; }
                cnt++;
            }
            if (cnt == 3) {   /* :193-219 */
                if (vi[0] == 0 || vi[1] == 0 || vi[2] == 0 || vi[0] > nv || vi[1] > nv || vi[2] > nv) continue;
                v3 v0 = verts[vi[0] - 1], v1 = verts[vi[1] - 1], v2 = verts[vi[2] - 1];
                v3 nrm;
                if (ni[0] != 0 && ni[0] <= nn) nrm = norms[ni[0] - 1];
                else nrm = vunit(vcross(vsub(v1, v0), vsub(v2, v0)));
                oprim t = make_tri5(v0, v1, v2, cur.bsdf, nrm);
                t.Le = cur.Le;
                pv_push(out, &t);
            } else if (cnt == 4) {   /* :221-246 */
                if (vi[0] == 0 || vi[1] == 0 || vi[2] == 0 || vi[3] == 0 ||
                    vi[0] > nv || vi[1] > nv || vi[2] > nv || vi[3] > nv) continue;
                oprim q = make_quad(verts[vi[0] - 1], verts[vi[1] - 1], verts[vi[2] - 1], verts[vi[3] - 1], cur.bsdf);
                if (ni[0] != 0 && ni[0] <= nn) q.normal = norms[ni[0] - 1];
                q.Le = cur.Le;
                pv_push(out, &q);
            }
            /* other arities: warned about and dropped (:247-250) */
        }
    }
    free(line); free(verts); free(norms); free(mats.m); fclose(f);
    return out->n > 0;   /* :254-257 */
}

/* application_state.h:323-365 */
static void convert_quads_to_triangles(primvec* pv) {
    primvec out = {0, 0, 0};
    for (int i = 0; i < pv->n; i++) {
        const oprim* q = &pv->p[i];
        if (q->type == PRIM_QUAD) {
            oprim t1 = make_tri4(q->v[0], q->v[1], q->v[2], q->bsdf); t1.Le = q->Le; pv_push(&out, &t1);
            oprim t2 = make_tri4(q->v[0], q->v[2], q->v[3], q->bsdf); t2.Le = q->Le; pv_push(&out, &t2);
        } else pv_push(&out, q);
    }
    free(pv->p); *pv = out;
}

/* rendering/form_factors.h:475-574 */
static v3 vhalf(v3 a, v3 b) { v3 s = vadd(a, b); return V(s.e[0] * 0.5f, s.e[1] * 0.5f, s.e[2] * 0.5f); }
static void subdivide_primitives(primvec* pv, int levels) {
    for (int it = 0; it < levels; it++) {
        primvec out = {0, 0, 0};
        for (int i = 0; i < pv->n; i++) {
            const oprim* p = &pv->p[i];
            if (p->type == PRIM_TRIANGLE) {   /* :479-499 */
                v3 m0 = vhalf(p->v[0], p->v[1]), m1 = vhalf(p->v[1], p->v[2]), m2 = vhalf(p->v[2], p->v[0]);
                oprim s[4] = { make_tri4(p->v[0], m0, m2, p->bsdf), make_tri4(m0, p->v[1], m1, p->bsdf),
                               make_tri4(m1, p->v[2], m2, p->bsdf), make_tri4(m0, m1, m2, p->bsdf) };
                for (int k = 0; k < 4; k++) { s[k].Le = p->Le; pv_push(&out, &s[k]); }
            } else {                          /* :501-522 */
                v3 m01 = vhalf(p->v[0], p->v[1]), m12 = vhalf(p->v[1], p->v[2]);
                v3 m23 = vhalf(p->v[2], p->v[3]), m30 = vhalf(p->v[3], p->v[0]);
                v3 c = vadd(vadd(vadd(p->v[0], p->v[1]), p->v[2]), p->v[3]);
                c = V(c.e[0] * 0.25f, c.e[1] * 0.25f, c.e[2] * 0.25f);
                oprim s[4] = { make_quad(p->v[0], m01, c, m30, p->bsdf), make_quad(m01, p->v[1], m12, c, p->bsdf),
                               make_quad(c, m12, p->v[2], m23, p->bsdf), make_quad(m30, c, m23, p->v[3], p->bsdf) };
                for (int k = 0; k < 4; k++) { s[k].Le = p->Le; pv_push(&out, &s[k]); }
            }
        }
        free(pv->p); *pv = out;
    }
}

/* ------------------------------------------------------------------------ */
/* rendering/bvh.h: BVHBuilder (76-219)                                       */
/* ------------------------------------------------------------------------ */
static v3 prim_centroid(const oprim* p) {   /* primitive.h:92-98 */
    if (p->type == PRIM_TRIANGLE) return vdivs(vadd(vadd(p->v[0], p->v[1]), p->v[2]), 3.0f);
    v3 s = vadd(vadd(vadd(p->v[0], p->v[1]), p->v[2]), p->v[3]);
    return V(s.e[0] * 0.25f, s.e[1] * 0.25f, s.e[2] * 0.25f);
}
typedef struct { v3 mn, mx; } aabb;
static aabb aabb_empty(void) { aabb b; b.mn = V(1e30f, 1e30f, 1e30f); b.mx = V(-1e30f, -1e30f, -1e30f); return b; }   /* bvh.h:17 */
static aabb aabb_merge(aabb a, aabb b) {   /* bvh.h:30-35 */
    aabb r;
    for (int k = 0; k < 3; k++) { r.mn.e[k] = fminf(a.mn.e[k], b.mn.e[k]); r.mx.e[k] = fmaxf(a.mx.e[k], b.mx.e[k]); }
    return r;
}
static aabb compute_bounds(const po_scene* s, int start, int end) {   /* bvh.h:108-148 */
    aabb bounds = aabb_empty();
    const float eps = 1e-6f;
    for (int i = start; i < end; i++) {
        const oprim* p = &s->prims[s->indices[i]];
        aabb b;
        for (int k = 0; k < 3; k++) {
            float mn, mx;
            if (p->type == PRIM_TRIANGLE) {
                mn = fminf(fminf(p->v[0].e[k], p->v[1].e[k]), p->v[2].e[k]);
                mx = fmaxf(fmaxf(p->v[0].e[k], p->v[1].e[k]), p->v[2].e[k]);
            } else {
                mn = fminf(fminf(p->v[0].e[k], p->v[1].e[k]), fminf(p->v[2].e[k], p->v[3].e[k]));
                mx = fmaxf(fmaxf(p->v[0].e[k], p->v[1].e[k]), fmaxf(p->v[2].e[k], p->v[3].e[k]));
            }
            b.mn.e[k] = mn - eps; b.mx.e[k] = mx + eps;
        }
        bounds = aabb_merge(bounds, b);
    }
    return bounds;
}
static int push_node(po_scene* s) {
    if (s->n_nodes == s->cap_nodes) { s->cap_nodes = s->cap_nodes ? s->cap_nodes * 2 : 64; s->nodes = (onode*)realloc(s->nodes, sizeof(onode) * s->cap_nodes); }
    onode* n = &s->nodes[s->n_nodes];
    n->bmin = V(1e30f, 1e30f, 1e30f); n->bmax = V(-1e30f, -1e30f, -1e30f);
    n->left_child = -1; n->right_child = -1; n->prim_count = 0;   /* bvh.h:69 */
    return s->n_nodes++;
}
static int build_recursive(po_scene* s, int start, int end) {   /* bvh.h:154-218 */
    int node_idx = push_node(s);
    aabb bbox = compute_bounds(s, start, end);
    int count = end - start;
    if (count <= 4) {
        onode* n = &s->nodes[node_idx];
        n->bmin = bbox.mn; n->bmax = bbox.mx; n->left_child = start; n->prim_count = count;
        return node_idx;
    }
    aabb cb = aabb_empty();
    for (int i = start; i < end; i++) {
        v3 c = prim_centroid(&s->prims[s->indices[i]]);
        aabb one; one.mn = c; one.mx = c;
        cb = aabb_merge(cb, one);
    }
    int best_axis = 0;
    v3 extent = vsub(cb.mx, cb.mn);
    if (extent.e[1] > extent.e[0]) best_axis = 1;
    if (extent.e[2] > extent.e[best_axis]) best_axis = 2;
    if (extent.e[best_axis] < 1e-6f) {
        onode* n = &s->nodes[node_idx];
        n->bmin = bbox.mn; n->bmax = bbox.mx; n->left_child = start; n->prim_count = count;
        return node_idx;
    }
    float split_pos = (cb.mn.e[best_axis] + cb.mx.e[best_axis]) * 0.5f;   /* bvh.h:21-23 center() */
    int mid = start;
    for (int i = start; i < end; i++) {
        if (prim_centroid(&s->prims[s->indices[i]]).e[best_axis] < split_pos) {
            int t = s->indices[i]; s->indices[i] = s->indices[mid]; s->indices[mid] = t;
            mid++;
        }
    }
    if (mid == start || mid == end) mid = start + count / 2;
    int left_idx = build_recursive(s, start, mid);
    int right_idx = build_recursive(s, mid, end);
    onode* n = &s->nodes[node_idx];
    n->bmin = bbox.mn; n->bmax = bbox.mx;
    n->left_child = left_idx; n->right_child = right_idx; n->prim_count = 0;
    return node_idx;
}
static void build_bvh(po_scene* s) {   /* bvh.h:83-104 */
    s->indices = (int*)malloc(sizeof(int) * (size_t)(s->n_prims > 0 ? s->n_prims : 1));
    for (int i = 0; i < s->n_prims; i++) s->indices[i] = i;
    s->nodes = NULL; s->n_nodes = 0; s->cap_nodes = 0;
    build_recursive(s, 0, s->n_prims);
}

/* ------------------------------------------------------------------------ */
/* scene API                                                                 */
/* ------------------------------------------------------------------------ */
po_scene* po_scene_load(const char* path, int subdivision_count, int convert_quads) {   /* application_state.h:367-464 */
    const char* dot = strrchr(path, '.');
    if (!dot) return NULL;
    char ext[16]; size_t n = 0;
    for (const char* p = dot; *p && n + 1 < sizeof ext; p++) ext[n++] = (char)tolower((unsigned char)*p);
    ext[n] = 0;
    if (strcmp(ext, ".obj") != 0) { fprintf(stderr, "[oracle] ERROR: Unsupported file format: %s\n", ext); return NULL; }
    primvec pv = {0, 0, 0};
    if (!load_obj(path, &pv)) { free(pv.p); return NULL; }
    if (convert_quads) convert_quads_to_triangles(&pv);
    if (subdivision_count > 0) subdivide_primitives(&pv, subdivision_count);
    po_scene* s = (po_scene*)calloc(1, sizeof *s);
    s->prims = pv.p; s->n_prims = pv.n; s->mis_bsdf_fraction = 0.5f;
    build_bvh(s);
    return s;
}

po_scene* po_scene_from_arrays(int n, const int* type, const float* verts,
                               const float* normal, const float* bsdf, const float* Le) {
    po_scene* s = (po_scene*)calloc(1, sizeof *s);
    s->mis_bsdf_fraction = 0.5f;
    s->prims = (oprim*)calloc((size_t)(n > 0 ? n : 1), sizeof(oprim)); s->n_prims = n;
    for (int i = 0; i < n; i++) {
        oprim* p = &s->prims[i];
        p->type = type[i];
        for (int k = 0; k < 4; k++) p->v[k] = V(verts[(i * 4 + k) * 3], verts[(i * 4 + k) * 3 + 1], verts[(i * 4 + k) * 3 + 2]);
        p->normal = V(normal[i * 3], normal[i * 3 + 1], normal[i * 3 + 2]);
        p->bsdf = V(bsdf[i * 3], bsdf[i * 3 + 1], bsdf[i * 3 + 2]);
        p->Le = V(Le[i * 3], Le[i * 3 + 1], Le[i * 3 + 2]);
    }
    build_bvh(s);
    return s;
}
void po_scene_free(po_scene* s) { if (!s) return; free(s->prims); free(s->nodes); free(s->indices); free(s->cdfs); free(s->radiosity); free(s->rad_grid); free(s->count_grid); free(s); }
/* per-primitive radiosity (n_prims * 3 floats, load order; NULL = zero): in the reference the radiosity solver's output */
void po_scene_set_radiosity(po_scene* s, const float* rgb) {
    free(s->radiosity); s->radiosity = NULL;
    if (!rgb) return;
    s->radiosity = (v3*)malloc(sizeof(v3) * (size_t)s->n_prims);
    for (int i = 0; i < s->n_prims; i++) s->radiosity[i] = V(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]);
}

/* SceneState::precomputeCDFs (application_state.h:492-585) from per-primitive radiosity grids
 * (rgb: n_prims * 256 * 3 floats, load order).  The grids are an INPUT here: the O(N^2) radiosity
 * solver that fills them in the reference (form_factors.h) is out of scope. */
static void build_cdf_records(po_scene* s, const float* pdfs /* n * 256 */) {   /* application_state.h:509-571 == :609-664 */
    free(s->cdfs);
    s->cdfs = (ocdf*)calloc((size_t)s->n_prims, sizeof(ocdf));
    for (int p = 0; p < s->n_prims; p++) {
        ocdf* cdf = &s->cdfs[p];
        for (int i = 0; i < GRID_SIZE; i++) cdf->pdf[i] = pdfs[(size_t)p * GRID_SIZE + i];
        cdf->total_weight = 0.0f;
        for (int v = 0; v < GRID_HALF_RES; v++) {
            float row_sum = 0.0f;
            for (int u = 0; u < GRID_RES; u++) row_sum += cdf->pdf[v * GRID_RES + u];
            cdf->row_sums[v] = row_sum;
            cdf->total_weight += row_sum;
        }
        float running = 0.0f;
        float inv_total = (cdf->total_weight > 1e-6f) ? (1.0f / cdf->total_weight) : 0.0f;
        for (int v = 0; v < GRID_HALF_RES; v++) { running += cdf->row_sums[v]; cdf->marginal_cdf[v] = running * inv_total; }
        cdf->marginal_cdf[GRID_HALF_RES - 1] = 1.0f;
        for (int v = 0; v < GRID_HALF_RES; v++) {
            const int ro = v * GRID_RES;
            const float row_sum = cdf->row_sums[v];
            if (row_sum < 1e-6f) {
                for (int u = 0; u < GRID_RES; u++) cdf->row_cdfs[ro + u] = (u + 1) * GRID_INV_RES;
            } else {
                float running_row = 0.0f;
                const float inv_row_sum = 1.0f / row_sum;
                for (int u = 0; u < GRID_RES; u++) { running_row += cdf->pdf[ro + u]; cdf->row_cdfs[ro + u] = running_row * inv_row_sum; }
                cdf->row_cdfs[ro + GRID_RES - 1] = 1.0f;
            }
        }
        for (int v = GRID_HALF_RES; v < GRID_RES; v++)
            for (int u = 0; u < GRID_RES; u++) cdf->row_cdfs[v * GRID_RES + u] = (u + 1) * GRID_INV_RES;
        cdf->is_valid = (cdf->total_weight > 1e-6f) ? 1 : 0;
    }
}
void po_scene_set_radiosity_grids(po_scene* s, const float* rgb) {
    free(s->cdfs); s->cdfs = NULL;
    free(s->rad_grid); s->rad_grid = NULL;
    if (!rgb) return;
    const size_t cells = (size_t)s->n_prims * GRID_SIZE;
    s->rad_grid = (v3*)malloc(sizeof(v3) * cells);
    memcpy(s->rad_grid, rgb, sizeof(v3) * cells);
    float* pdfs = (float*)malloc(sizeof(float) * cells);
    for (size_t i = 0; i < cells; i++) pdfs[i] = 0.2126f * rgb[3 * i] + 0.7152f * rgb[3 * i + 1] + 0.0722f * rgb[3 * i + 2];   /* :516 */
    build_cdf_records(s, pdfs);
    free(pdfs);
}
void po_scene_set_mis_fraction(po_scene* s, float f) { s->mis_bsdf_fraction = f; }
/* out: n_prims * 530 floats (the PrecomputedCDF records, is_valid as an int bit pattern) */
int po_scene_get_cdfs(const po_scene* s, float* out) {
    if (!s->cdfs) return 0;
    memcpy(out, s->cdfs, (size_t)s->n_prims * sizeof(ocdf));
    return 1;
}
/* the restatement's view of render_config.h:8-31, 14-17 - compared with the reference's own compiler pass
 * (oracle/ref_harness.cpp: ref_layout, ref_grid_constants) in tests/test_oracle_vs_ref.py */
void po_cdf_layout(int out[10]) {
    out[0] = (int)sizeof(ocdf); out[1] = (int)offsetof(ocdf, pdf); out[2] = (int)offsetof(ocdf, row_sums);
    out[3] = (int)offsetof(ocdf, marginal_cdf); out[4] = (int)offsetof(ocdf, row_cdfs);
    out[5] = (int)offsetof(ocdf, total_weight); out[6] = (int)offsetof(ocdf, is_valid);
    out[7] = GRID_RES; out[8] = GRID_SIZE; out[9] = GRID_HALF_RES;
}
void po_grid_constants(double out[6]) {
    out[0] = (double)GRID_INV_RES; out[1] = (double)GRID_INV_HALF_RES; out[2] = (double)GRID_D_THETA; out[3] = (double)GRID_D_PHI;
    out[4] = (double)PTMI_PI_D; out[5] = (double)(PTMI_PI_D * 0.5f);
}
int po_scene_num_prims(const po_scene* s) { return s->n_prims; }
int po_scene_num_nodes(const po_scene* s) { return s->n_nodes; }
void po_scene_get_prims(const po_scene* s, int* type, float* verts, float* normal, float* bsdf, float* Le) {
    for (int i = 0; i < s->n_prims; i++) {
        const oprim* p = &s->prims[i];
        type[i] = p->type;
        for (int k = 0; k < 4; k++) for (int c = 0; c < 3; c++) verts[(i * 4 + k) * 3 + c] = p->v[k].e[c];
        for (int c = 0; c < 3; c++) { normal[i * 3 + c] = p->normal.e[c]; bsdf[i * 3 + c] = p->bsdf.e[c]; Le[i * 3 + c] = p->Le.e[c]; }
    }
}
void po_scene_get_bvh(const po_scene* s, float* bmin, float* bmax, int* left, int* right, int* count, int* indices) {
    for (int i = 0; i < s->n_nodes; i++) {
        for (int c = 0; c < 3; c++) { bmin[i * 3 + c] = s->nodes[i].bmin.e[c]; bmax[i * 3 + c] = s->nodes[i].bmax.e[c]; }
        left[i] = s->nodes[i].left_child; right[i] = s->nodes[i].right_child; count[i] = s->nodes[i].prim_count;
    }
    for (int i = 0; i < s->n_prims; i++) indices[i] = s->indices[i];
}

/* ------------------------------------------------------------------------ */
/* rendering/sensor.h                                                        */
/* ------------------------------------------------------------------------ */
static void update_camera(v3 origin, v3 lookat, v3 vup, float vfov, float aspect, po_camera_frame* out) {   /* sensor.h:38-51 */
    float theta = (float)((double)vfov * PTMI_PI_D / (double)180.0f);   /* vfov * M_PI / 180.0f with double M_PI */
    float half_height = (float)ptmi_tan_d((double)(theta / 2.0f));
    float half_width = aspect * half_height;
    v3 w = vunit(vsub(origin, lookat));
    v3 u = vunit(vcross(vup, w));
    v3 v = vcross(w, u);
    v3 llc = vsub(vsub(vsub(origin, vscale(half_width, u)), vscale(half_height, v)), w);
    v3 hor = vscale(2 * half_width, u);
    v3 ver = vscale(2 * half_height, v);
    for (int k = 0; k < 3; k++) {
        out->origin[k] = origin.e[k]; out->lower_left_corner[k] = llc.e[k];
        out->horizontal[k] = hor.e[k]; out->vertical[k] = ver.e[k];
    }
}
void po_camera_frame_setup(const po_camera* cam, int width, int height, po_camera_frame* out) {
    v3 origin = V(cam->origin[0], cam->origin[1], cam->origin[2]);
    v3 lookat = V(cam->lookat[0], cam->lookat[1], cam->lookat[2]);
    v3 vup = V(cam->vup[0], cam->vup[1], cam->vup[2]);
    float aspect = (float)width / (float)height;   /* application_state.h:108 */
    if (cam->orbit) {   /* sensor.h:16-29 radius from the constructor's lookfrom; :56-67 orbit */
        float radius = vlen(vsub(origin, lookat));
        float yawRad = (float)((double)cam->yaw_deg * PTMI_PI_D / (double)180.0f);     /* ToRadian, sensor.h:11 */
        float pitchRad = (float)((double)cam->pitch_deg * PTMI_PI_D / (double)180.0f);
        float sy, cy, sp, cp;
        ptmi_sincosf(yawRad, &sy, &cy); ptmi_sincosf(pitchRad, &sp, &cp);
        origin.e[0] = lookat.e[0] + radius * cp * cy;
        origin.e[1] = lookat.e[1] + radius * sp;
        origin.e[2] = lookat.e[2] + radius * cp * sy;
    }
    update_camera(origin, lookat, vup, cam->vfov_deg, aspect, out);
}
typedef struct { v3 o, d; } ray_t;
static ray_t make_ray(v3 origin, v3 direction) { ray_t r; r.o = origin; r.d = vunit(direction); return r; }   /* ray.h:9-12 */
static ray_t camera_get_ray(const po_camera_frame* c, float u, float v) {   /* sensor.h:31-33 */
    v3 o = V(c->origin[0], c->origin[1], c->origin[2]);
    v3 llc = V(c->lower_left_corner[0], c->lower_left_corner[1], c->lower_left_corner[2]);
    v3 hor = V(c->horizontal[0], c->horizontal[1], c->horizontal[2]);
    v3 ver = V(c->vertical[0], c->vertical[1], c->vertical[2]);
    return make_ray(o, vsub(vadd(vadd(llc, vscale(u, hor)), vscale(v, ver)), o));
}
void po_camera_ray(const po_camera_frame* c, float u, float v, float o[3], float d[3]) {
    ray_t r = camera_get_ray(c, u, v);
    for (int k = 0; k < 3; k++) { o[k] = r.o.e[k]; d[k] = r.d.e[k]; }
}

/* ------------------------------------------------------------------------ */
/* cuRAND XORWOW (third-party, NVIDIA CUDA Toolkit curand_kernel.h; not in    */
/* /root/reference, version unpinned by the reference's CMakeLists.txt:2).    */
/* Published algorithm restated: Marsaglia xorwow + Weyl sequence; curand_init */
/* seed scrambling; subsequence = 2^67 steps, skipped with GF(2) matrix powers;*/
/* curand_uniform = x * 2^-32 + 2^-33 in (0,1].                                */
/* Call sites: integrator.h:63-64, 210, 279, 384-385.  UNPINNED vs real cuRAND; */
/* the step and the 2^67 skip-ahead are pinned against rocRAND's implementation */
/* of the same generator (rocrand_harness.cpp, tests/test_rng_vs_rocrand.py),   */
/* the seed scramble and the float conversion are not (rocRAND's differ).       */
/* ------------------------------------------------------------------------ */
typedef struct { uint32_t row[160][5]; } xmat;
static xmat g_jump[32];          /* g_jump[k] = T^(2^67 * 2^k) */
static int g_jump_ready = 0;

static void xorwow_step_v(uint32_t v[5]) {
    uint32_t t = v[0] ^ (v[0] >> 2);
    v[0] = v[1]; v[1] = v[2]; v[2] = v[3]; v[3] = v[4];
    v[4] = (v[4] ^ (v[4] << 4)) ^ (t ^ (t << 1));
}
static void xmat_apply(const xmat* m, const uint32_t in[5], uint32_t out[5]) {
    uint32_t r[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < 5; i++)
        for (int j = 0; j < 32; j++)
            if (in[i] & (1u << j)) { const uint32_t* row = m->row[i * 32 + j]; for (int k = 0; k < 5; k++) r[k] ^= row[k]; }
    for (int k = 0; k < 5; k++) out[k] = r[k];
}
static void xmat_square(const xmat* m, xmat* out) {
    for (int b = 0; b < 160; b++) xmat_apply(m, m->row[b], out->row[b]);
}
static void init_jump_tables(void) {
    if (g_jump_ready) return;
    xmat* a = (xmat*)malloc(sizeof(xmat)); xmat* b = (xmat*)malloc(sizeof(xmat));
    for (int i = 0; i < 5; i++) for (int j = 0; j < 32; j++) {
        uint32_t v[5] = {0, 0, 0, 0, 0}; v[i] = 1u << j; xorwow_step_v(v);
        memcpy(a->row[i * 32 + j], v, sizeof v);
    }
    for (int s = 0; s < 67; s++) { xmat_square(a, b); xmat* t = a; a = b; b = t; }
    g_jump[0] = *a;
    for (int k = 1; k < 32; k++) xmat_square(&g_jump[k - 1], &g_jump[k]);
    free(a); free(b);
    g_jump_ready = 1;
}
void po_rng_init(uint64_t seed, uint64_t subsequence, uint32_t st[6]) {
    init_jump_tables();
    uint32_t s0 = ((uint32_t)seed) ^ 0xaad26b49u;
    uint32_t s1 = (uint32_t)(seed >> 32) ^ 0xf7dcefddu;
    uint32_t t0 = 1099087573u * s0;
    uint32_t t1 = 2591861531u * s1;
    st[5] = 6615241u + t1 + t0;          /* d */
    st[0] = 123456789u + t0;
    st[1] = 362436069u ^ t0;
    st[2] = 521288629u + t1;
    st[3] = 88675123u ^ t1;
    st[4] = 5783321u + t0;
    for (int k = 0; k < 32; k++)
        if (subsequence & (1ull << k)) { uint32_t o[5]; xmat_apply(&g_jump[k], st, o); memcpy(st, o, sizeof o); }
    /* offset = 0 at every call site */
}
static inline uint32_t xorwow_next(uint32_t st[6]) {
    xorwow_step_v(st);
    st[5] += 362437u;
    return st[4] + st[5];
}
static inline float rng_uniform(uint32_t st[6]) {
    uint32_t x = xorwow_next(st);
    return (float)x * 2.3283064e-10f + (2.3283064e-10f / 2.0f);
}
float po_rng_uniform(uint32_t st[6]) { return rng_uniform(st); }

/* test hook: T^(2^log2n) built by repeated squaring must equal 2^log2n direct steps */
int po_rng_selftest(int log2n, const uint32_t v_in[5]) {
    xmat* a = (xmat*)malloc(sizeof(xmat)); xmat* b = (xmat*)malloc(sizeof(xmat));
    for (int i = 0; i < 5; i++) for (int j = 0; j < 32; j++) {
        uint32_t v[5] = {0, 0, 0, 0, 0}; v[i] = 1u << j; xorwow_step_v(v);
        memcpy(a->row[i * 32 + j], v, sizeof v);
    }
    for (int s = 0; s < log2n; s++) { xmat_square(a, b); xmat* t = a; a = b; b = t; }
    uint32_t viaM[5], direct[5];
    xmat_apply(a, v_in, viaM);
    memcpy(direct, v_in, sizeof direct);
    for (uint64_t i = 0; i < (1ull << log2n); i++) xorwow_step_v(direct);
    free(a); free(b);
    return memcmp(viaM, direct, sizeof direct) == 0 ? 0 : 1;
}

/* test hooks for the pin against rocRAND's implementation of the same generator (oracle/rocrand_harness.cpp): raw draws from
 * a given state, and the skip over n subsequences of 2^67 draws as po_rng_init applies it */
void po_xorwow_next_raw(uint32_t st[6], int n, uint32_t* out) { for (int i = 0; i < n; i++) out[i] = xorwow_next(st); }
void po_xorwow_skip_subsequences(uint32_t v[5], uint64_t n) {
    init_jump_tables();
    for (int k = 0; k < 32; k++)
        if (n & (1ull << k)) { uint32_t o[5]; xmat_apply(&g_jump[k], v, o); memcpy(v, o, sizeof o); }
}

void po_sincosf(float x, float* s, float* c) { ptmi_sincosf(x, s, c); }
float po_powf(float x, float y) { return ptmi_powf(x, y); }
float po_acosf(float x) { return ptmi_acosf(x); }
float po_expf(float x) { return ptmi_expf(x); }
float po_atan2f(float y, float x) { return ptmi_atan2f(y, x); }

/* ------------------------------------------------------------------------ */
/* triangle.h:64-96, quad.h:49-132, primitive.h:83-90                        */
/* ------------------------------------------------------------------------ */
static inline int tri_intersect(const oprim* p, const ray_t* r, float t_min, float t_max, float* t_out) {
    const float EPSILON = 1e-8f;
    v3 edge1 = vsub(p->v[1], p->v[0]);
    v3 edge2 = vsub(p->v[2], p->v[0]);
    v3 h = vcross(r->d, edge2);
    float a = vdot(edge1, h);
    if (fabsf(a) < EPSILON) return 0;
    float f = 1.0f / a;
    v3 s = vsub(r->o, p->v[0]);
    float u = f * vdot(s, h);
    if (u < 0.0f || u > 1.0f) return 0;
    v3 q = vcross(s, edge1);
    float v = f * vdot(r->d, q);
    if (v < 0.0f || u + v > 1.0f) return 0;
    float t = f * vdot(edge2, q);
    if (t > EPSILON && t >= t_min && t <= t_max) { *t_out = t; return 1; }
    return 0;
}
static inline int quad_half(v3 v00, v3 va, v3 vb, const ray_t* r, float t_min, float* closest_t) {
    const float EPSILON = 1e-8f;
    v3 edge1 = vsub(va, v00);
    v3 edge2 = vsub(vb, v00);
    v3 h = vcross(r->d, edge2);
    float a = vdot(edge1, h);
    if (fabsf(a) > EPSILON) {
        float f = 1.0f / a;
        v3 s = vsub(r->o, v00);
        float u = f * vdot(s, h);
        if (u >= 0.0f && u <= 1.0f) {
            v3 q = vcross(s, edge1);
            float v = f * vdot(r->d, q);
            if (v >= 0.0f && u + v <= 1.0f) {
                float t = f * vdot(edge2, q);
                if (t > EPSILON && t >= t_min && t < *closest_t) { *closest_t = t; return 1; }
            }
        }
    }
    return 0;
}
static inline int quad_intersect(const oprim* p, const ray_t* r, float t_min, float t_max, float* t_out) {
    float closest_t = t_max; int hit = 0;
    hit |= quad_half(p->v[0], p->v[1], p->v[2], r, t_min, &closest_t);   /* (v00, v10, v11) */
    hit |= quad_half(p->v[0], p->v[2], p->v[3], r, t_min, &closest_t);   /* (v00, v11, v01) */
    if (hit) { *t_out = closest_t; return 1; }
    return 0;
}
static inline int prim_intersect(const oprim* p, const ray_t* r, float t_min, float t_max, float* t_out) {
    return p->type == PRIM_TRIANGLE ? tri_intersect(p, r, t_min, t_max, t_out) : quad_intersect(p, r, t_min, t_max, t_out);
}

/* ------------------------------------------------------------------------ */
/* scene.h:39-129                                                            */
/* ------------------------------------------------------------------------ */
typedef struct { uint64_t rays, node_visits, prim_tests, hits; } counters;

static int scene_intersect_bvh(const po_scene* sc, const ray_t* r, float t_min, float t_max,
                               float* t_hit, int* prim_hit, counters* cn) {   /* scene.h:50-110 */
    int hit_anything = 0;
    float closest_t = t_max;
    int stack[64]; int stack_ptr = 0;
    stack[stack_ptr++] = 0;
    float inv_dir[3] = { 1.0f / r->d.e[0], 1.0f / r->d.e[1], 1.0f / r->d.e[2] };
    while (stack_ptr > 0) {
        int node_idx = stack[--stack_ptr];
        const onode* node = &sc->nodes[node_idx];
        cn->node_visits++;
        float tmin_box = t_min, tmax_box = closest_t;
        for (int a = 0; a < 3; a++) {
            float t0 = (node->bmin.e[a] - r->o.e[a]) * inv_dir[a];
            float t1 = (node->bmax.e[a] - r->o.e[a]) * inv_dir[a];
            if (inv_dir[a] < 0.0f) { float tmp = t0; t0 = t1; t1 = tmp; }
            tmin_box = t0 > tmin_box ? t0 : tmin_box;
            tmax_box = t1 < tmax_box ? t1 : tmax_box;
        }
        if (tmax_box < tmin_box) continue;
        if (node->prim_count > 0) {
            for (int i = 0; i < node->prim_count; i++) {
                int prim_idx = sc->indices[node->left_child + i];
                float t;
                cn->prim_tests++;
                if (prim_intersect(&sc->prims[prim_idx], r, t_min, closest_t, &t)) {
                    if (t < closest_t) { closest_t = t; *prim_hit = prim_idx; hit_anything = 1; }
                }
            }
        } else {
            if (stack_ptr < 62) { stack[stack_ptr++] = node->right_child; stack[stack_ptr++] = node->left_child; }
        }
    }
    *t_hit = closest_t;
    return hit_anything;
}
static int scene_intersect_linear(const po_scene* sc, const ray_t* r, float t_min, float t_max,
                                  float* t_hit, int* prim_hit, counters* cn) {   /* scene.h:113-129 */
    int hit_anything = 0; float si_t = t_max;
    for (int i = 0; i < sc->n_prims; i++) {
        float t;
        cn->prim_tests++;
        if (prim_intersect(&sc->prims[i], r, t_min, si_t, &t)) {
            if (t < si_t) { si_t = t; *prim_hit = i; hit_anything = 1; }
        }
    }
    *t_hit = si_t;
    return hit_anything;
}

void po_intersect(const po_scene* sc, const float o[3], const float d[3], float t_min, float t_max,
                  int use_bvh, po_hit* out) {
    ray_t r; r.o = V(o[0], o[1], o[2]); r.d = V(d[0], d[1], d[2]);   /* direction taken as given */
    counters cn = {0, 0, 0, 0};
    float t = 0; int prim = -1;
    int hit = use_bvh ? scene_intersect_bvh(sc, &r, t_min, t_max, &t, &prim, &cn)
                      : scene_intersect_linear(sc, &r, t_min, t_max, &t, &prim, &cn);
    memset(out, 0, sizeof *out);
    out->hit = hit; out->prim = hit ? prim : -1;
    out->node_visits = (int)cn.node_visits; out->prim_tests = (int)cn.prim_tests;
    if (hit) {
        const oprim* p = &sc->prims[prim];
        v3 pt = vadd(r.o, vscale(t, r.d));   /* triangle.h:90, quad.h:126 */
        out->t = t;
        for (int k = 0; k < 3; k++) { out->p[k] = pt.e[k]; out->n[k] = p->normal.e[k]; out->bsdf[k] = p->bsdf.e[k]; out->Le[k] = p->Le.e[k]; }
    }
}

/* ------------------------------------------------------------------------ */
/* integrator.h:62-85 sampleCosineHemisphere                                  */
/* ------------------------------------------------------------------------ */
static v3 sample_cosine_hemisphere_uv(v3 normal, float u, float v) {
    float r = sqrtf(u);
    float phi = (float)((double)2.0f * PTMI_PI_D * (double)v);   /* 2.0f * M_PI * v, M_PI double */
    float sphi, cphi;
    ptmi_sincosf(phi, &sphi, &cphi);
    float x = r * cphi;
    float y = r * sphi;
    float z = sqrtf(fmaxf(0.0f, 1.0f - u));
    v3 tangent, bitangent;
    if (normal.e[2] < -0.9999999f) {
        tangent = V(0.0f, -1.0f, 0.0f);
        bitangent = V(-1.0f, 0.0f, 0.0f);
    } else {
        float a = 1.0f / (1.0f + normal.e[2]);
        float b = -normal.e[0] * normal.e[1] * a;
        tangent = V(1.0f - normal.e[0] * normal.e[0] * a, b, -normal.e[0]);
        bitangent = V(b, 1.0f - normal.e[1] * normal.e[1] * a, -normal.e[1]);
    }
    /* tangent * x + bitangent * y + normal * z */
    return vunit(vadd(vadd(vscale(x, tangent), vscale(y, bitangent)), vscale(z, normal)));
}
void po_sample_cosine_hemisphere(const float n[3], float u, float v, float out[3]) {
    v3 d = sample_cosine_hemisphere_uv(V(n[0], n[1], n[2]), u, v);
    for (int k = 0; k < 3; k++) out[k] = d.e[k];
}

/* ------------------------------------------------------------------------ */
/* rendering/grid.h: Grid over a PrecomputedCDF (loadPrecomputed path)         */
/* ------------------------------------------------------------------------ */
static void build_frame(v3 n, v3* t, v3* b) {   /* grid.h:287-297, same Frisvad frame as integrator.h:72-82 */
    if (n.e[2] < -0.9999999f) { *t = V(0.0f, -1.0f, 0.0f); *b = V(-1.0f, 0.0f, 0.0f); return; }
    const float a = 1.0f / (1.0f + n.e[2]);
    const float c = -n.e[0] * n.e[1] * a;
    *t = V(1.0f - n.e[0] * n.e[0] * a, c, -n.e[0]);
    *b = V(c, 1.0f - n.e[1] * n.e[1] * a, -n.e[1]);
}
static int linear_search_cdf(const float* cdf, int size, float xi) {   /* grid.h:233-240 */
    xi = fminf(fmaxf(xi, 0.0f), 0.999999f);
    for (int i = 0; i < size; i++) if (xi < cdf[i]) return i;
    return size - 1;
}
static float grid_pdf_for_cell(const ocdf* g, int theta_idx, int phi_idx) {   /* grid.h:242-253 */
    const int idx = theta_idx * GRID_RES + phi_idx;
    const float cell_value = g->pdf[idx];
    if (cell_value < 1e-8f) return 1e-6f;
    const float cell_prob = cell_value / fmaxf(g->total_weight, 1e-6f);
    const float theta_center = (float)((double)((theta_idx + 0.5f) * GRID_INV_HALF_RES) * (PTMI_PI_D * 0.5f));
    float st, ct; ptmi_sincosf(theta_center, &st, &ct);
    const float sin_theta = fmaxf(st, 0.01f);
    const float solid_angle = (float)(((double)sin_theta * GRID_D_THETA) * GRID_D_PHI);
    return cell_prob / fmaxf(solid_angle, 1e-6f);
}
static v3 grid_sample(const ocdf* g, v3 normal, uint32_t rng[6], float* out_pdf) {   /* grid.h:141-188 (is_valid checked by the caller) */
    const float xi1 = rng_uniform(rng);
    const float xi2 = rng_uniform(rng);
    const int theta_idx = linear_search_cdf(g->marginal_cdf, GRID_HALF_RES, xi1);
    const int phi_idx = linear_search_cdf(&g->row_cdfs[theta_idx * GRID_RES], GRID_RES, xi2);
    const float jitter_theta = rng_uniform(rng);
    const float jitter_phi = rng_uniform(rng);
    float theta = (float)((double)(((float)theta_idx + jitter_theta) * GRID_INV_HALF_RES) * (PTMI_PI_D * 0.5f));
    theta = fminf(theta, (float)(PTMI_PI_D * 0.5f - (double)0.01f));
    const float phi = (float)((double)((((float)phi_idx + jitter_phi) * GRID_INV_RES) * 2.0f) * PTMI_PI_D);
    float sin_t, cos_t, sin_p, cos_p;
    ptmi_sincosf(theta, &sin_t, &cos_t);
    ptmi_sincosf(phi, &sin_p, &cos_p);
    const v3 local = V(sin_t * cos_p, sin_t * sin_p, cos_t);
    v3 tangent, bitangent; build_frame(normal, &tangent, &bitangent);
    const v3 world = vunit(vadd(vadd(vscale(local.e[0], tangent), vscale(local.e[1], bitangent)), vscale(local.e[2], normal)));
    *out_pdf = grid_pdf_for_cell(g, theta_idx, phi_idx);
    return world;
}
static float grid_compute_pdf(const ocdf* g, v3 dir, v3 normal) {   /* grid.h:200-216 + worldToSpherical :299-310 */
    v3 tangent, bitangent; build_frame(normal, &tangent, &bitangent);
    const float lx = vdot(dir, tangent), ly = vdot(dir, bitangent), lz = vdot(dir, normal);
    const float theta = ptmi_acosf(fminf(fmaxf(lz, -1.0f), 1.0f));
    float phi = ptmi_atan2f(ly, lx);
    if (phi < 0.0f) phi = (float)((double)phi + (double)2.0f * PTMI_PI_D);
    if ((double)theta > PTMI_PI_D * 0.5f) return 0.0f;
    int theta_idx = (int)(((double)theta * ((double)2.0f / PTMI_PI_D)) * GRID_HALF_RES);
    int phi_idx = (int)(((double)phi * ((double)0.5f / PTMI_PI_D)) * GRID_RES);
    theta_idx = theta_idx < 0 ? 0 : (theta_idx > GRID_HALF_RES - 1 ? GRID_HALF_RES - 1 : theta_idx);
    phi_idx = phi_idx < 0 ? 0 : (phi_idx > GRID_RES - 1 ? GRID_RES - 1 : phi_idx);
    return grid_pdf_for_cell(g, theta_idx, phi_idx);
}
static float mis_power_heuristic(float pdf_a, float pdf_b) {   /* integrator.h:91-96 */
    if (pdf_a <= 0.0f) return 0.0f;
    const float a2 = pdf_a * pdf_a, b2 = pdf_b * pdf_b;
    return a2 / (a2 + b2);
}
static v3 sample_mis(const ocdf* g, v3 normal, uint32_t rng[6], float* weight, float bsdf_prob) {   /* integrator.h:112-167 */
    const float BSDF_PROB = fmaxf(fminf(bsdf_prob, 0.99f), 0.01f);
    const float GRID_PROB = 1.0f - BSDF_PROB;
    const float xi = rng_uniform(rng);
    v3 dir; float pdf_grid, pdf_bsdf, mis_w;
    if (xi < BSDF_PROB) {
        const float u = rng_uniform(rng), v = rng_uniform(rng);
        dir = sample_cosine_hemisphere_uv(normal, u, v);
        const float cos_theta = fmaxf(vdot(dir, normal), 0.0f);
        pdf_bsdf = (float)((double)cos_theta / PTMI_PI_D);
        pdf_grid = grid_compute_pdf(g, dir, normal);
        mis_w = mis_power_heuristic(pdf_bsdf, pdf_grid);
        *weight = (pdf_bsdf > 1e-6f) ? mis_w / BSDF_PROB : 0.0f;
    } else {
        dir = grid_sample(g, normal, rng, &pdf_grid);
        const float cos_theta = fmaxf(vdot(dir, normal), 0.0f);
        pdf_bsdf = (float)((double)cos_theta / PTMI_PI_D);
        mis_w = mis_power_heuristic(pdf_grid, pdf_bsdf);
        if (pdf_grid > 1e-6f && cos_theta > 0.0f) {
            float w = (float)((double)(mis_w * cos_theta) / ((PTMI_PI_D * (double)pdf_grid) * (double)GRID_PROB));
            *weight = fminf(w, 10.0f);
        } else *weight = 0.0f;
    }
    return dir;
}

/* ------------------------------------------------------------------------ */
/* integrator.h:189-268 integrator(), all sampling modes                      */
/* ------------------------------------------------------------------------ */
enum { SAMPLING_BSDF = 0, SAMPLING_FORMFACTOR = 1, SAMPLING_RADIOSITY = 2, SAMPLING_MIS = 3, SAMPLING_TOPK = 4 };   /* render_config.h:38-44 */

static void integrator(const po_scene* sc, ray_t ray, v3* L, int max_depth, uint32_t rng[6], int mode, counters* cn) {
    v3 throughput = V(1.0f, 1.0f, 1.0f);
    ray_t r = ray;
    for (int depth = 0; depth < max_depth; depth++) {
        float t; int prim = -1;
        cn->rays++;
        if (!scene_intersect_bvh(sc, &r, 1e-4f, FLT_MAX, &t, &prim, cn)) break;   /* :198-201 */
        cn->hits++;
        const oprim* p = &sc->prims[prim];
        v3 si_p = vadd(r.o, vscale(t, r.d));
        *L = vadd(*L, vmul(throughput, p->Le));                                  /* :204 */
        if (depth > 2) {                                                         /* :207-212 */
            float max_throughput = fmaxf(throughput.e[0], fmaxf(throughput.e[1], throughput.e[2]));
            float rr_prob = fminf(max_throughput, 0.95f);
            if (rng_uniform(rng) > rr_prob) break;
            float k = recip_via_double(rr_prob);                                 /* throughput /= rr_prob */
            throughput = V(throughput.e[0] * k, throughput.e[1] * k, throughput.e[2] * k);
        }
        throughput = vmul(throughput, p->bsdf);                                  /* :215 */
        if (vlen(throughput) < 1e-5f) break;                                     /* :218 */
        v3 normal = p->normal;
        v3 shading_normal = vdot(r.d, normal) < 0 ? normal : vneg(normal);       /* :221-222 */
        v3 next_dir;
        /* initGridFromPrimitive (:31-57): with precomputed CDFs the grid is the primitive's record; without them the
         * raw radiosity grid of a freshly loaded primitive is all zero, so the grid is invalid either way and the
         * branch below falls back to cosine sampling (:258-261) with the same two draws as the BSDF mode */
        const ocdf* g = (mode != SAMPLING_BSDF && sc->cdfs && sc->cdfs[prim].is_valid) ? &sc->cdfs[prim] : NULL;
        if (g && mode == SAMPLING_MIS) {                                         /* :238-241 */
            float weight = 1.0f;
            next_dir = sample_mis(g, shading_normal, rng, &weight, sc->mis_bsdf_fraction);
            throughput = V(throughput.e[0] * weight, throughput.e[1] * weight, throughput.e[2] * weight);
        } else if (g) {                                                          /* :242-257 pure grid sampling */
            float grid_pdf;
            next_dir = grid_sample(g, shading_normal, rng, &grid_pdf);
            const float cos_theta = fmaxf(vdot(next_dir, shading_normal), 0.0f);
            float weight = (float)((double)cos_theta / (PTMI_PI_D * (double)fmaxf(grid_pdf, 1e-6f)));
            weight = fminf(fmaxf(weight, 0.0f), 10.0f);
            throughput = V(throughput.e[0] * weight, throughput.e[1] * weight, throughput.e[2] * weight);
        } else {
            float u = rng_uniform(rng);                                          /* :63-64: u then v */
            float v = rng_uniform(rng);
            next_dir = sample_cosine_hemisphere_uv(shading_normal, u, v);        /* :230 / :260 */
        }
        r = make_ray(vadd(si_p, vscale(1e-4f, shading_normal)), next_dir);       /* :266 */
    }
}

/* ------------------------------------------------------------------------ */
/* integrator.h:371-408 render kernel (po_render) and :460-504 render_radiosity */
/* ------------------------------------------------------------------------ */
/* ------------------------------------------------------------------------ */
/* the radiosity pre-pass.  RadiosityState::runSolver            */
/* (application_state.h:688-777) and its kernels (form_factors.h:71-467,       */
/* grid_filter.h:35-312).  Restated; form_factors.h/grid_filter.h include      */
/* <cuda_runtime.h>/<curand_kernel.h> and cannot be compiled here, so this part */
/* is pinned only through the pieces that live in primitive.h (area, centroid, */
/* sampleUniform: checked against the compiled reference in                    */
/* tests/test_oracle_vs_ref.py) and by properties (tests/test_radiosity_solver).*/
/*                                                                            */
/* Two places where the reference is not a function of its inputs:            */
/*  - radiosity_iteration_kernel (form_factors.h:441-465) reads unshot_rad of  */
/*    every j while other threads of the same launch overwrite theirs: a race. */
/*    Restated with the evident intent (all reads see the previous iteration). */
/*  - calculate_form_factors_mc_kernel adds per-pair partial sums to            */
/*    radiosity_grid with float atomics in arbitrary order (:344-351).  Every   */
/*    iteration's update_radiosity_grid overwrites that grid, so it only       */
/*    survives with num_iterations == 0; canonical order here: ascending j.    */
/*    (The count grid `grid` sums integers and is exact in any order.)         */
/* ------------------------------------------------------------------------ */
static float prim_area(const oprim* p) {   /* triangle.h:28,54 / quad.h:31 */
    if (p->type == PRIM_TRIANGLE)
        return 0.5f * vlen(vcross(vsub(p->v[1], p->v[0]), vsub(p->v[2], p->v[0])));
    const v3 edge1 = vsub(p->v[1], p->v[0]), edge2 = vsub(p->v[3], p->v[0]);
    return 0.5f * (vlen(vcross(edge1, edge2)) + vlen(vcross(vsub(p->v[2], p->v[1]), vsub(p->v[2], p->v[3]))));
}
static v3 bary_point(v3 a, v3 b, v3 c, float r1, float r2) {   /* primitive.h:153-157 */
    const float sqrt_r1 = sqrtf(r1);
    const float u = 1.0f - sqrt_r1;
    const float v = sqrt_r1 * (1.0f - r2);
    const float w = sqrt_r1 * r2;
    return vadd(vadd(vscale(u, a), vscale(v, b)), vscale(w, c));
}
static v3 prim_sample_uniform(const oprim* p, float r1, float r2) {   /* primitive.h:150-191 */
    if (p->type == PRIM_TRIANGLE) return bary_point(p->v[0], p->v[1], p->v[2], r1, r2);
    const v3 v00 = p->v[0], v10 = p->v[1], v11 = p->v[2], v01 = p->v[3];
    const float area1 = 0.5f * vlen(vcross(vsub(v10, v00), vsub(v01, v00)));
    const float area2 = 0.5f * vlen(vcross(vsub(v11, v10), vsub(v11, v01)));
    const float total_area = area1 + area2;
    const float area_ratio = area1 / total_area;
    if (r1 < area_ratio) return bary_point(v00, v10, v01, r1 / area_ratio, r2);
    return bary_point(v10, v11, v01, (r1 - area_ratio) / (1.0f - area_ratio), r2);
}
void po_prim_geometry(const po_scene* s, int i, float* area, float centroid[3]) {
    *area = prim_area(&s->prims[i]);
    const v3 c = prim_centroid(&s->prims[i]);
    memcpy(centroid, c.e, sizeof c.e);
}
void po_prim_sample_uniform(const po_scene* s, int i, float r1, float r2, float out[3]) {
    const v3 p = prim_sample_uniform(&s->prims[i], r1, r2);
    memcpy(out, p.e, sizeof p.e);
}

/* form_factors.h:107-130 direction_to_grid_indices_local: theta over [0, pi] -> 16 rows, phi over [0, 2 pi) -> 16 columns */
static int direction_to_grid_index_local(v3 world_dir, v3 normal) {
    v3 tangent, bitangent; build_frame(normal, &tangent, &bitangent);   /* form_factors.h:93-103 == grid.h:287-297 */
    const float lx = vdot(world_dir, tangent), ly = vdot(world_dir, bitangent), lz = vdot(world_dir, normal);
    const float r = sqrtf(lx * lx + ly * ly + lz * lz);
    const float theta = (r > 0.0f) ? ptmi_acosf(fminf(lz / r, 1.0f)) : 0.0f;
    float phi = ptmi_atan2f(ly, lx);
    if (phi < 0.0f) phi = (float)((double)phi + (double)2.0f * PTMI_PI_D);
    int grid_theta = (int)fminf((float)(((double)theta / PTMI_PI_D) * GRID_RES), (float)(GRID_RES - 1));
    int grid_phi = (int)fminf((float)(((double)phi / ((double)2.0f * PTMI_PI_D)) * GRID_RES), (float)(GRID_RES - 1));
    grid_theta = grid_theta < 0 ? 0 : (grid_theta > GRID_RES - 1 ? GRID_RES - 1 : grid_theta);
    grid_phi = grid_phi < 0 ? 0 : (grid_phi > GRID_RES - 1 ? GRID_RES - 1 : grid_phi);
    return grid_theta * GRID_RES + grid_phi;
}
int po_direction_to_grid_index(const float dir[3], const float normal[3]) {
    return direction_to_grid_index_local(V(dir[0], dir[1], dir[2]), V(normal[0], normal[1], normal[2]));
}

/* form_factors.h:143-208 visibility_test_anyhit: its own traversal (32-entry stack, children dropped from 30 on,
 * left pushed first so the right child is visited first, guarded reciprocals, EPSILON 1e-5) */
static int visibility_test_anyhit(const po_scene* sc, const ray_t* r, float max_dist, int source_idx, int target_idx) {
    const float EPSILON = 1e-5f;
    int stack[32]; int stack_ptr = 0;
    stack[stack_ptr++] = 0;
    float inv_dir[3];
    for (int a = 0; a < 3; a++) inv_dir[a] = 1.0f / (fabsf(r->d.e[a]) > 1e-8f ? r->d.e[a] : 1e-8f);
    while (stack_ptr > 0) {
        const int node_idx = stack[--stack_ptr];
        const onode* node = &sc->nodes[node_idx];
        float t1 = (node->bmin.e[0] - r->o.e[0]) * inv_dir[0];
        float t2 = (node->bmax.e[0] - r->o.e[0]) * inv_dir[0];
        float tmin = fminf(t1, t2), tmax = fmaxf(t1, t2);
        t1 = (node->bmin.e[1] - r->o.e[1]) * inv_dir[1];
        t2 = (node->bmax.e[1] - r->o.e[1]) * inv_dir[1];
        tmin = fmaxf(tmin, fminf(t1, t2)); tmax = fminf(tmax, fmaxf(t1, t2));
        t1 = (node->bmin.e[2] - r->o.e[2]) * inv_dir[2];
        t2 = (node->bmax.e[2] - r->o.e[2]) * inv_dir[2];
        tmin = fmaxf(tmin, fminf(t1, t2)); tmax = fminf(tmax, fmaxf(t1, t2));
        if (tmax < EPSILON || tmin > max_dist || tmin > tmax) continue;
        if (node->prim_count > 0) {
            for (int i = 0; i < node->prim_count; i++) {
                const int prim_idx = sc->indices[node->left_child + i];
                if (prim_idx == source_idx || prim_idx == target_idx) continue;
                float t;
                if (prim_intersect(&sc->prims[prim_idx], r, EPSILON, max_dist, &t)) return 1;
            }
        } else if (stack_ptr < 30) {
            stack[stack_ptr++] = node->left_child;
            stack[stack_ptr++] = node->right_child;
        }
    }
    return 0;
}
int po_visibility_blocked(const po_scene* sc, const float o[3], const float d[3], float max_dist, int source_idx, int target_idx) {
    const ray_t r = make_ray(V(o[0], o[1], o[2]), V(d[0], d[1], d[2]));
    return visibility_test_anyhit(sc, &r, max_dist, source_idx, target_idx);
}

typedef struct { const float* area; const v3* centroid; const v3* radiosity; } solver_geom;

/* form_factors.h:219-366 calculate_form_factors_mc_kernel, one (i, j) pair; adds the pair's partial sums to row i's
 * grids; returns F_ij */
static float form_factor_mc_pair(const po_scene* sc, const solver_geom* g, int i, int j, int n_samples,
                                 float* grid_i, v3* rad_grid_i, uint64_t* rays) {
    const int num_primitives = sc->n_prims;
    const int ff_idx = i * num_primitives + j;
    if (i == j) return 0.0f;
    const oprim* prim_i = &sc->prims[i]; const oprim* prim_j = &sc->prims[j];
    const v3 center_i = g->centroid[i], center_j = g->centroid[j];
    const v3 normal_i = prim_i->normal, normal_j = prim_j->normal;
    const v3 dir_ij = vsub(center_j, center_i);
    const float dist_sq = dir_ij.e[0] * dir_ij.e[0] + dir_ij.e[1] * dir_ij.e[1] + dir_ij.e[2] * dir_ij.e[2];
    const float dist = sqrtf(dist_sq);
    if (dist < 1e-6f) return 0.0f;
    const v3 dir_norm = vdivs(dir_ij, dist);
    const float cos_i_approx = vdot(normal_i, dir_norm);
    const float cos_j_approx = -vdot(normal_j, dir_norm);
    if (cos_i_approx <= 0.0f || cos_j_approx <= 0.0f) return 0.0f;
    const float approx_ff = (float)((double)(cos_i_approx * cos_j_approx * g->area[j]) / (PTMI_PI_D * (double)dist_sq));
    int actual_samples = n_samples;
    if (approx_ff < 0.001f) actual_samples = n_samples / 4 > 1 ? n_samples / 4 : 1;
    else if (approx_ff < 0.01f) actual_samples = n_samples / 2 > 2 ? n_samples / 2 : 2;

    uint32_t local_state[6];
    po_rng_init(12345u + (uint64_t)(int64_t)ff_idx, (uint64_t)(int64_t)ff_idx, local_state);   /* :85-89 formfactor_rand_init */
    float visibility_sum = 0.0f, cos_i_sum = 0.0f, cos_j_sum = 0.0f, dist_sum = 0.0f;
    int valid_samples = 0;
    float local_grid[GRID_SIZE]; v3 local_rad_grid[GRID_SIZE];
    for (int k = 0; k < GRID_SIZE; k++) { local_grid[k] = 0.0f; local_rad_grid[k] = V(0.0f, 0.0f, 0.0f); }
    for (int s = 0; s < actual_samples; ++s) {
        float r1 = rng_uniform(local_state), r2 = rng_uniform(local_state);
        const v3 p_i = prim_sample_uniform(prim_i, r1, r2);
        r1 = rng_uniform(local_state); r2 = rng_uniform(local_state);
        const v3 p_j = prim_sample_uniform(prim_j, r1, r2);
        v3 sample_dir = vsub(p_j, p_i);
        const float r = vlen(sample_dir);
        if (r < 1e-6f) continue;
        sample_dir = vdivs(sample_dir, r);
        const float cos_theta_i = vdot(normal_i, sample_dir);
        const float cos_theta_j = -vdot(normal_j, sample_dir);
        if (cos_theta_i <= 0.0f || cos_theta_j <= 0.0f) continue;
        const ray_t shadow_ray = make_ray(vadd(p_i, vscale(1e-4f, normal_i)), sample_dir);
        (*rays)++;
        if (!visibility_test_anyhit(sc, &shadow_ray, r - 2e-4f, i, j)) {
            visibility_sum += 1.0f; cos_i_sum += cos_theta_i; cos_j_sum += cos_theta_j; dist_sum += r;
            valid_samples++;
            const int grid_idx = direction_to_grid_index_local(sample_dir, normal_i);
            local_grid[grid_idx] += 1.0f;
            const float geometric_weight = (cos_theta_i * cos_theta_j) / (r * r);
            const v3 contrib = vscale(g->area[j], vscale(geometric_weight, g->radiosity[j]));
            local_rad_grid[grid_idx] = vadd(local_rad_grid[grid_idx], contrib);
        }
    }
    for (int k = 0; k < GRID_SIZE; k++)
        if (local_grid[k] > 0.0f) { grid_i[k] += local_grid[k]; rad_grid_i[k] = vadd(rad_grid_i[k], local_rad_grid[k]); }
    if (valid_samples > 0) {
        const float avg_cos_i = cos_i_sum / (float)valid_samples;
        const float avg_cos_j = cos_j_sum / (float)valid_samples;
        const float avg_dist = dist_sum / (float)valid_samples;
        const float visibility_fraction = visibility_sum / (float)actual_samples;
        const float F_ij = (float)((double)(visibility_fraction * (avg_cos_i * avg_cos_j * g->area[j])) /
                                   (PTMI_PI_D * (double)avg_dist * (double)avg_dist));
        return fmaxf(0.0f, fminf(F_ij, 1.0f));
    }
    return 0.0f;
}

/* form_factors.h:368-415 calculate_form_factors_kernel (point-to-point) */
static float form_factor_p2p_pair(const po_scene* sc, const solver_geom* g, int i, int j, uint64_t* rays) {
    if (i == j) return 0.0f;
    const v3 vec_ij = vsub(g->centroid[j], g->centroid[i]);
    const float r = vlen(vec_ij);
    if (r < 1e-6f) return 0.0f;
    const v3 dir_ij = vdivs(vec_ij, r);
    const v3 normal_i = sc->prims[i].normal, normal_j = sc->prims[j].normal;
    const float cos_theta_i = vdot(normal_i, dir_ij);
    const float cos_theta_j = vdot(normal_j, vneg(dir_ij));
    if (cos_theta_i <= 0.0f || cos_theta_j <= 0.0f) return 0.0f;
    const ray_t visibility_ray = make_ray(vadd(g->centroid[i], vscale(1e-4f, normal_i)), dir_ij);
    (*rays)++;
    if (visibility_test_anyhit(sc, &visibility_ray, r - 2e-4f, i, j)) return 0.0f;
    const float ff = (float)((double)(cos_theta_i * cos_theta_j * g->area[j]) / (PTMI_PI_D * (double)r * (double)r));
    return fmaxf(0.0f, ff);
}

/* grid_filter.h:35-41 */
static float gaussian_weight(float distance, float sigma) { return ptmi_expf(-(distance * distance) / (2.0f * sigma * sigma)); }
static float luminance_from_rgb(v3 rgb) { return 0.2126f * rgb.e[0] + 0.7152f * rgb.e[1] + 0.0722f * rgb.e[2]; }
#define BILATERAL_KERNEL_RADIUS 2
/* grid_filter.h:55-101 bilateralFilterCell / :221-249 gaussianFilterCell */
static v3 filter_cell(const v3* input_grid, int center_i, int center_j, int bilateral, float sigma_spatial, float sigma_range) {
    const v3 center_val = input_grid[center_i * GRID_RES + center_j];
    const float center_lum = luminance_from_rgb(center_val);
    v3 weighted_sum = V(0.0f, 0.0f, 0.0f);
    float total_weight = 0.0f;
    for (int di = -BILATERAL_KERNEL_RADIUS; di <= BILATERAL_KERNEL_RADIUS; di++)
        for (int dj = -BILATERAL_KERNEL_RADIUS; dj <= BILATERAL_KERNEL_RADIUS; dj++) {
            const int ni = center_i + di;
            const int nj = (center_j + dj + GRID_RES) % GRID_RES;           /* phi wraps, theta does not */
            if (ni < 0 || ni >= GRID_RES) continue;
            const v3 neighbor_val = input_grid[ni * GRID_RES + nj];
            const float spatial_dist = sqrtf((float)(di * di + dj * dj));
            float weight = gaussian_weight(spatial_dist, sigma_spatial);
            if (bilateral) {
                const float range_dist = fabsf(center_lum - luminance_from_rgb(neighbor_val));
                weight = weight * gaussian_weight(range_dist, sigma_range);
            }
            weighted_sum = vadd(weighted_sum, vscale(weight, neighbor_val));
            total_weight += weight;
        }
    if (total_weight > 1e-6f) return vdivs(weighted_sum, total_weight);
    return center_val;
}

/* rows of the form-factor matrix are independent of each other: a subset, for spot checks at sizes where the whole
 * matrix is too slow on the CPU.  rows: n_rows receiver indices; out_ff n_rows*n, out_grid n_rows*256 (may be NULL) */
int po_form_factor_rows(const po_scene* sc, const po_radiosity_params* prm, int n_threads, int n_rows, const int* rows,
                        float* out_ff, float* out_grid) {
    if (!sc || !prm || !rows || !out_ff || sc->n_prims > 46340 || prm->mc_samples < 1) return -1;
    const int n = sc->n_prims;
    init_jump_tables();
#ifdef _OPENMP
    if (n_threads <= 0) n_threads = omp_get_num_procs();
#else
    n_threads = 1;
#endif
    float* area = (float*)malloc(sizeof(float) * (size_t)n);
    v3* centroid = (v3*)malloc(sizeof(v3) * (size_t)n);
    v3* radiosity = (v3*)malloc(sizeof(v3) * (size_t)n);
    for (int i = 0; i < n; i++) { area[i] = prim_area(&sc->prims[i]); centroid[i] = prim_centroid(&sc->prims[i]); radiosity[i] = sc->prims[i].Le; }
    const solver_geom g = { area, centroid, radiosity };
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads)
    for (int r = 0; r < n_rows; r++) {
        const int i = rows[r];
        float grid[GRID_SIZE]; v3 rad_grid[GRID_SIZE]; uint64_t rays = 0;
        memset(grid, 0, sizeof grid); memset(rad_grid, 0, sizeof rad_grid);
        for (int j = 0; j < n; j++)
            out_ff[(size_t)r * n + j] = prm->use_monte_carlo ? form_factor_mc_pair(sc, &g, i, j, prm->mc_samples, grid, rad_grid, &rays)
                                                             : form_factor_p2p_pair(sc, &g, i, j, &rays);
        if (out_grid) memcpy(out_grid + (size_t)r * GRID_SIZE, grid, sizeof grid);
    }
    free(area); free(centroid); free(radiosity);
    return 0;
}

int po_radiosity_solve(po_scene* sc, const po_radiosity_params* prm, int n_threads, float* out_form_factors,
                       float* out_radiosity, float* out_unshot, float* out_grid, float* out_rad_grid, uint64_t* out_rays) {
    if (!sc || !prm || sc->n_prims <= 0 || sc->n_prims > 46340 || prm->num_iterations < 0 || prm->mc_samples < 1) return -1;
    const int n = sc->n_prims;
    init_jump_tables();
#ifdef _OPENMP
    if (n_threads <= 0) n_threads = omp_get_num_procs();
#else
    n_threads = 1;
#endif
    float* area = (float*)malloc(sizeof(float) * (size_t)n);
    v3* centroid = (v3*)malloc(sizeof(v3) * (size_t)n);
    v3* radiosity = (v3*)malloc(sizeof(v3) * (size_t)n);
    v3* unshot = (v3*)malloc(sizeof(v3) * (size_t)n);
    v3* next_unshot = (v3*)malloc(sizeof(v3) * (size_t)n);
    float* ff = (float*)malloc(sizeof(float) * (size_t)n * (size_t)n);
    float* grid = (float*)calloc((size_t)n * GRID_SIZE, sizeof(float));           /* initialize_directional_grids :71-83 */
    v3* rad_grid = (v3*)calloc((size_t)n * GRID_SIZE, sizeof(v3));
    for (int i = 0; i < n; i++) {
        area[i] = prim_area(&sc->prims[i]); centroid[i] = prim_centroid(&sc->prims[i]);
        radiosity[i] = sc->prims[i].Le; unshot[i] = sc->prims[i].Le;              /* application_state.h:697-701 */
    }
    const solver_geom g = { area, centroid, radiosity };
    uint64_t rays_total = 0;
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads) reduction(+ : rays_total)
    for (int i = 0; i < n; i++) {
        uint64_t rays = 0;
        for (int j = 0; j < n; j++)
            ff[(size_t)i * n + j] = prm->use_monte_carlo
                ? form_factor_mc_pair(sc, &g, i, j, prm->mc_samples, grid + (size_t)i * GRID_SIZE, rad_grid + (size_t)i * GRID_SIZE, &rays)
                : form_factor_p2p_pair(sc, &g, i, j, &rays);
        rays_total += rays;
    }
    for (int it = 0; it < prm->num_iterations; ++it) {
        /* radiosity_iteration_kernel :441-465 */
#pragma omp parallel for schedule(static) num_threads(n_threads)
        for (int i = 0; i < n; i++) {
            v3 incident_rad = V(0.0f, 0.0f, 0.0f);
            for (int j = 0; j < n; ++j)
                if (i != j) {
                    const float F_ij = ff[(size_t)i * n + j];
                    if (F_ij > 0.0f) incident_rad = vadd(incident_rad, vscale(F_ij, unshot[j]));
                }
            const v3 bsdf = sc->prims[i].bsdf;
            const v3 reflected = V(fminf(bsdf.e[0] * incident_rad.e[0], incident_rad.e[0]),
                                   fminf(bsdf.e[1] * incident_rad.e[1], incident_rad.e[1]),
                                   fminf(bsdf.e[2] * incident_rad.e[2], incident_rad.e[2]));
            radiosity[i] = vadd(radiosity[i], reflected);
            next_unshot[i] = reflected;
        }
        { v3* t = unshot; unshot = next_unshot; next_unshot = t; }
        /* update_radiosity_grid :405-439 and the optional filter are recomputed from scratch by every iteration and
         * feed nothing but the final output: evaluated once, after the last iteration, below */
    }
    if (prm->num_iterations > 0) {
#pragma omp parallel for schedule(dynamic, 4) num_threads(n_threads)
        for (int i = 0; i < n; i++) {
            v3* rg = rad_grid + (size_t)i * GRID_SIZE;
            for (int k = 0; k < GRID_SIZE; k++) rg[k] = V(0.0f, 0.0f, 0.0f);
            for (int j = 0; j < n; ++j) {
                if (i == j) continue;
                const float F_ij = ff[(size_t)i * n + j];
                if (F_ij <= 0.0f) continue;
                v3 dir_ij = vsub(centroid[j], centroid[i]);
                const float r = vlen(dir_ij);
                if (r < 1e-6f) continue;
                dir_ij = vdivs(dir_ij, r);
                const int grid_idx = direction_to_grid_index_local(dir_ij, sc->prims[i].normal);
                rg[grid_idx] = vadd(rg[grid_idx], vscale(F_ij, radiosity[j]));
            }
            if (prm->enable_filtering) {                                            /* application_state.h:759-767 */
                v3 tmp[GRID_SIZE];
                for (int c = 0; c < GRID_SIZE; c++)
                    tmp[c] = filter_cell(rg, c / GRID_RES, c % GRID_RES, prm->use_bilateral, prm->filter_sigma_spatial, prm->filter_sigma_range);
                memcpy(rg, tmp, sizeof tmp);
            }
        }
    }
    if (out_form_factors) memcpy(out_form_factors, ff, sizeof(float) * (size_t)n * (size_t)n);
    if (out_radiosity) memcpy(out_radiosity, radiosity, sizeof(v3) * (size_t)n);
    if (out_unshot) memcpy(out_unshot, unshot, sizeof(v3) * (size_t)n);
    if (out_grid) memcpy(out_grid, grid, sizeof(float) * (size_t)n * GRID_SIZE);
    if (out_rad_grid) memcpy(out_rad_grid, rad_grid, sizeof(v3) * (size_t)n * GRID_SIZE);
    if (out_rays) *out_rays = rays_total;
    /* ui_windows.h:185-192: runSolver; precomputeCDFs(); upload primitives (radiosity now visible to render_radiosity) */
    po_scene_set_radiosity(sc, (const float*)radiosity);
    po_scene_set_radiosity_grids(sc, (const float*)rad_grid);
    free(sc->count_grid); sc->count_grid = grid; grid = NULL;
    free(area); free(centroid); free(radiosity); free(unshot); free(next_unshot); free(ff); free(grid); free(rad_grid);
    return 0;
}

/* "Apply Filter & Rebuild CDFs" (ui_windows.h:154-167): filter_pdfs_for_primitives (grid_filter.h:329-507) on the
 * count grids and on the luminance of the radiosity grids - 5x5 bilateral or gaussian on floats, then each primitive's
 * 256 values divided by their sum - followed by precomputeCDFsFromFiltered (application_state.h:587-680), which builds
 * the CDF records from the filtered luminance.  out_*: n * 256 floats, may be NULL. */
static float filter_cell_float(const float* src, int ci, int cj, int bilateral, float sigma_spatial, float sigma_range) {
    const float center = src[ci * GRID_RES + cj];                               /* grid_filter.h:352-367, 381-406 */
    float weighted_sum = 0.0f, total_weight = 0.0f;
    for (int di = -BILATERAL_KERNEL_RADIUS; di <= BILATERAL_KERNEL_RADIUS; di++)
        for (int dj = -BILATERAL_KERNEL_RADIUS; dj <= BILATERAL_KERNEL_RADIUS; dj++) {
            const int ni = ci + di, nj = (cj + dj + GRID_RES) % GRID_RES;
            if (ni < 0 || ni >= GRID_RES) continue;
            float w = gaussian_weight(sqrtf((float)(di * di + dj * dj)), sigma_spatial);
            if (bilateral) w = w * gaussian_weight(fabsf(center - src[ni * GRID_RES + nj]), sigma_range);
            weighted_sum += src[ni * GRID_RES + nj] * w;
            total_weight += w;
        }
    if (total_weight > 1e-6f) return weighted_sum / total_weight;
    return center;
}
static void filter_and_normalize(const float* in, float* out, int bilateral, float sigma_spatial, float sigma_range) {
    for (int c = 0; c < GRID_SIZE; c++) out[c] = filter_cell_float(in, c / GRID_RES, c % GRID_RES, bilateral, sigma_spatial, sigma_range);
    float sum = 0.0f;                                                           /* normalize_pdf_kernel :409-418 */
    for (int c = 0; c < GRID_SIZE; c++) sum += out[c];
    if (sum <= 1e-12f) return;
    for (int c = 0; c < GRID_SIZE; c++) out[c] = out[c] / sum;
}
int po_scene_apply_grid_filter(po_scene* s, int use_bilateral, float sigma_spatial, float sigma_range,
                               float* out_formfactor, float* out_radiosity) {
    if (!s || !s->rad_grid) return -1;
    const int n = s->n_prims;
    float* ff = (float*)malloc(sizeof(float) * (size_t)n * GRID_SIZE);
    float* rad = (float*)malloc(sizeof(float) * (size_t)n * GRID_SIZE);
    for (int p = 0; p < n; p++) {
        float lum[GRID_SIZE], cnt[GRID_SIZE];
        for (int c = 0; c < GRID_SIZE; c++) {
            lum[c] = luminance_from_rgb(s->rad_grid[(size_t)p * GRID_SIZE + c]);   /* compute_radiosity_luminance_kernel :324-336 */
            cnt[c] = s->count_grid ? s->count_grid[(size_t)p * GRID_SIZE + c] : 0.0f;
        }
        filter_and_normalize(cnt, ff + (size_t)p * GRID_SIZE, use_bilateral, sigma_spatial, sigma_range);
        filter_and_normalize(lum, rad + (size_t)p * GRID_SIZE, use_bilateral, sigma_spatial, sigma_range);
    }
    build_cdf_records(s, rad);
    if (out_formfactor) memcpy(out_formfactor, ff, sizeof(float) * (size_t)n * GRID_SIZE);
    if (out_radiosity) memcpy(out_radiosity, rad, sizeof(float) * (size_t)n * GRID_SIZE);
    free(ff); free(rad);
    return 0;
}

static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

int po_render(const po_scene* sc, const po_camera* cam, int width, int height, int spp, int max_depth, int sampling_mode,
              uint64_t seed_base, int reset_rng, uint32_t* rng_state,
              int y0, int y1, int n_threads,
              unsigned char* out_rgb8, float* out_radiance, po_stats* stats) {
    if (!sc || width <= 0 || height <= 0 || spp <= 0 || y0 < 0 || y1 > height || y0 > y1) return -1;
    init_jump_tables();
    po_camera_frame cf; po_camera_frame_setup(cam, width, height, &cf);
    counters total = {0, 0, 0, 0};
#ifdef _OPENMP
    if (n_threads <= 0) n_threads = omp_get_num_procs();
#else
    n_threads = 1;
#endif
    /* render_init (integrator.h:274-280) is a separate, untimed pass, as in the
     * reference where it runs once per allocateBuffers(). */
    uint32_t* own_state = NULL;
    if (!rng_state) { own_state = (uint32_t*)malloc(sizeof(uint32_t) * 6 * (size_t)width * (size_t)(y1 - y0)); reset_rng = 1; }
    if (reset_rng) {
#pragma omp parallel for schedule(static) num_threads(n_threads)
        for (int y = y0; y < y1; y++)
            for (int x = 0; x < width; x++) {
                int pixel_index = y * width + x;
                uint32_t* st = rng_state ? &rng_state[(size_t)pixel_index * 6]
                                         : &own_state[((size_t)(y - y0) * (size_t)width + (size_t)x) * 6];
                po_rng_init(seed_base + (uint64_t)pixel_index, (uint64_t)pixel_index, st);   /* :279 */
            }
    }
    double t_start = now_s();
    uint64_t c_rays = 0, c_nodes = 0, c_tests = 0, c_hits = 0;
#pragma omp parallel for schedule(dynamic, 4) num_threads(n_threads) reduction(+ : c_rays, c_nodes, c_tests, c_hits)
    for (int y = y0; y < y1; y++) {
        counters cn = {0, 0, 0, 0};
        for (int x = 0; x < width; x++) {
            int pixel_index = y * width + x;
            uint32_t* rng = rng_state ? &rng_state[(size_t)pixel_index * 6]
                                      : &own_state[((size_t)(y - y0) * (size_t)width + (size_t)x) * 6];
            v3 color = V(0.0f, 0.0f, 0.0f);
            for (int s = 0; s < spp; s++) {
                float u = ((float)x + rng_uniform(rng)) / (float)width;    /* :384 */
                float v = ((float)y + rng_uniform(rng)) / (float)height;   /* :385 */
                ray_t ray = camera_get_ray(&cf, u, v);
                v3 sample_color = V(0.0f, 0.0f, 0.0f);
                integrator(sc, ray, &sample_color, max_depth, rng, sampling_mode, &cn);
                color = vadd(color, sample_color);
            }
            {   /* color /= float(spp) : vector.h:90-94 */
                float k = recip_via_double((float)spp);
                color = V(color.e[0] * k, color.e[1] * k, color.e[2] * k);
            }
            if (out_radiance) for (int c = 0; c < 3; c++) out_radiance[(size_t)pixel_index * 3 + c] = color.e[c];
            if (out_rgb8) {
                v3 tm = vdivv(color, vadd(color, V(1.0f, 1.0f, 1.0f)));   /* :396 */
                const float gamma = 1.0f / 2.2f;
                for (int c = 0; c < 3; c++) {
                    float g = ptmi_powf(tm.e[c], gamma);
                    out_rgb8[(size_t)pixel_index * 3 + c] = (unsigned char)(255.99f * fminf(g, 1.0f));
                }
            }
        }
        c_rays += cn.rays; c_nodes += cn.node_visits; c_tests += cn.prim_tests; c_hits += cn.hits;
    }
    total.rays = c_rays; total.node_visits = c_nodes; total.prim_tests = c_tests; total.hits = c_hits;
    double t_end = now_s();
    free(own_state);
    if (stats) {
        stats->seconds = t_end - t_start;
        stats->samples = (uint64_t)width * (uint64_t)(y1 - y0) * (uint64_t)spp;
        stats->rays = total.rays; stats->node_visits = total.node_visits;
        stats->prim_tests = total.prim_tests; stats->hits = total.hits;
    }
    return 0;
}

/* render_radiosity (integrator.h:460-504): first hit only, Le + per-primitive radiosity, sqrt "gamma".  Same frame /
 * row / RNG conventions as po_render; out_radiance receives color / spp before the sqrt. */
int po_render_radiosity(const po_scene* sc, const po_camera* cam, int width, int height, int spp,
                        uint64_t seed_base, int reset_rng, uint32_t* rng_state, int y0, int y1, int n_threads,
                        unsigned char* out_rgb8, float* out_radiance) {
    if (!sc || width <= 0 || height <= 0 || spp <= 0 || y0 < 0 || y1 > height || y0 > y1) return -1;
    init_jump_tables();
    po_camera_frame cf; po_camera_frame_setup(cam, width, height, &cf);
#ifdef _OPENMP
    if (n_threads <= 0) n_threads = omp_get_num_procs();
#else
    n_threads = 1;
#endif
#pragma omp parallel for schedule(dynamic, 4) num_threads(n_threads)
    for (int y = y0; y < y1; y++) {
        counters cn = {0, 0, 0, 0};
        for (int x = 0; x < width; x++) {
            const int pixel_index = y * width + x;
            uint32_t local[6];
            if (rng_state && !reset_rng) memcpy(local, &rng_state[(size_t)pixel_index * 6], sizeof local);   /* :468 local copy */
            else po_rng_init(seed_base + (uint64_t)pixel_index, (uint64_t)pixel_index, local);
            v3 color = V(0.0f, 0.0f, 0.0f);
            for (int s = 0; s < spp; s++) {
                const float u = ((float)x + rng_uniform(local)) / (float)width;
                const float v = ((float)y + rng_uniform(local)) / (float)height;
                const ray_t ray = camera_get_ray(&cf, u, v);
                float t; int prim = -1;
                if (scene_intersect_bvh(sc, &ray, 1e-4f, FLT_MAX, &t, &prim, &cn)) {
                    color = vadd(color, sc->prims[prim].Le);                                                 /* :481 */
                    color = vadd(color, sc->radiosity ? sc->radiosity[prim] : V(0.0f, 0.0f, 0.0f));          /* :483 */
                }
            }
            { const float k = recip_via_double((float)spp); color = V(color.e[0] * k, color.e[1] * k, color.e[2] * k); }
            if (out_radiance) for (int c = 0; c < 3; c++) out_radiance[(size_t)pixel_index * 3 + c] = color.e[c];
            if (out_rgb8) for (int c = 0; c < 3; c++)
                out_rgb8[(size_t)pixel_index * 3 + c] = (unsigned char)(255.99f * sqrtf(fminf(color.e[c], 1.0f)));   /* :492-501 */
            if (rng_state) memcpy(&rng_state[(size_t)pixel_index * 6], local, sizeof local);                 /* :503 */
        }
    }
    return 0;
}
