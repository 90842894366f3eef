// ref_pbrt_harness.cpp — thin C-ABI driver around the REFERENCE's own PBRT import.
//
// TEST INFRASTRUCTURE ONLY.  Built by oracle/Makefile into oracle/_ref/libptmi_ref_pbrt.so when /root/reference is
// present; the reference sources are included / compiled BY PATH from where they lie and are never copied into this repo.
//
// What is compiled, unmodified: include/utils/pbrt_loader.h (loadPBRT, :178-422; it includes no CUDA header, only
// <pbrtParser/Scene.h> and the geometry headers ref_harness.cpp already uses) and the vendored parser it calls,
// ext/pbrtparser/pbrtParser/impl/{syntactic,semantic}/*.cpp + impl/3rdParty/rply.c (plain C++11/C, no generated code;
// the reference's CMake only lists these files).  The __host__/__device__ qualifiers of the geometry headers come from
// ROCm's own <hip/amd_detail/host_defines.h>, as in ref_harness.cpp.  No stand-in header is written.
#include <hip/amd_detail/host_defines.h>

#include <cfloat>
#include <cstring>
#include <iostream>
#include <sstream>

#include "utils/pbrt_loader.h"

namespace {
struct Silencer {   // loadPBRT reports to std::cout / std::cerr
    std::streambuf *o, *e; std::ostringstream sink;
    Silencer() : o(std::cout.rdbuf(sink.rdbuf())), e(std::cerr.rdbuf(sink.rdbuf())) {}
    ~Silencer() { std::cout.rdbuf(o); std::cerr.rdbuf(e); }
};
struct Loaded { Primitive* prims = nullptr; int n = 0; };
}

extern "C" {

// loadPBRT(filename) -> handle (NULL where the reference returns false or throws)
void* ref_pbrt_load(const char* filename, int quiet) {
    Loaded* L = new Loaded;
    bool ok = false;
    try {
        if (quiet) { Silencer s; ok = loadPBRT(filename, &L->prims, L->n); }
        else ok = loadPBRT(filename, &L->prims, L->n);
    } catch (...) { ok = false; }
    if (!ok) { delete L; return nullptr; }
    return L;
}
int ref_pbrt_count(void* h) { return ((Loaded*)h)->n; }
// per primitive: type, 4 vertices (12 floats; the 4th zero for triangles), normal, bsdf, Le
void ref_pbrt_get(void* h, int* type, float* verts, float* normal, float* bsdf, float* Le) {
    Loaded* L = (Loaded*)h;
    for (int i = 0; i < L->n; i++) {
        const Primitive& p = L->prims[i];
        type[i] = (int)p.type;
        float* v = verts + (size_t)i * 12;
        std::memset(v, 0, 12 * sizeof(float));
        if (p.type == PRIM_TRIANGLE) {
            for (int k = 0; k < 3; k++) { v[k] = p.tri.v0[k]; v[3 + k] = p.tri.v1[k]; v[6 + k] = p.tri.v2[k];
                                          normal[3 * i + k] = p.tri.normal[k]; bsdf[3 * i + k] = p.tri.bsdf[k]; Le[3 * i + k] = p.tri.Le[k]; }
        } else {
            for (int k = 0; k < 3; k++) { v[k] = p.quad.v00[k]; v[3 + k] = p.quad.v10[k]; v[6 + k] = p.quad.v11[k]; v[9 + k] = p.quad.v01[k];
                                          normal[3 * i + k] = p.quad.normal[k]; bsdf[3 * i + k] = p.quad.bsdf[k]; Le[3 * i + k] = p.quad.Le[k]; }
        }
    }
}
void ref_pbrt_free(void* h) { Loaded* L = (Loaded*)h; if (!L) return; delete[] L->prims; delete L; }

}  // extern "C"
