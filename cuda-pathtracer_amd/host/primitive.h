// primitive.h — host-side primitive record.
//
// Mirrors what the hot path reads out of the reference's 4364-byte tagged union
// (include/rendering/primitive.h:21-29): the vertices, the stored normal, Kd
// ("bsdf") and Ke ("Le").  The radiosity grids/history that make up the other
// 4.2 KB are not on this path and are not carried.
#pragma once
#include <vector>

#include "../csrc/pt_vec.h"

namespace ptmi {

enum PrimitiveType { PRIM_TRIANGLE = 0, PRIM_QUAD = 1 };   // primitive.h:15-18

struct Primitive {
    PrimitiveType type = PRIM_TRIANGLE;
    f3 v[4] = {};            // triangle: v0 v1 v2 ; quad: v00 v10 v11 v01
    f3 bsdf = {0.8f, 0.8f, 0.8f};
    f3 normal = {0, 0, 1};
    f3 Le = {0, 0, 0};

    // Triangle(v0,v1,v2,bsdf,normal) — triangle.h:49-62 (normal taken as given)
    static Primitive triangle(f3 v0, f3 v1, f3 v2, f3 bsdf, f3 normal) {
        Primitive p; p.type = PRIM_TRIANGLE; p.v[0] = v0; p.v[1] = v1; p.v[2] = v2; p.v[3] = mk3(0, 0, 0);
        p.bsdf = bsdf; p.normal = normal; return p;
    }
    // Triangle(v0,v1,v2,bsdf) — triangle.h:23-31 (geometric normal)
    static Primitive triangle(f3 v0, f3 v1, f3 v2, f3 bsdf) {
        return triangle(v0, v1, v2, bsdf, unit_vector(cross(v1 - v0, v2 - v0)));
    }
    // Quad(v00,v10,v11,v01,bsdf) — quad.h:23-31 (normal from the v00 corner)
    static Primitive quad(f3 v00, f3 v10, f3 v11, f3 v01, f3 bsdf) {
        Primitive p; p.type = PRIM_QUAD; p.v[0] = v00; p.v[1] = v10; p.v[2] = v11; p.v[3] = v01;
        p.bsdf = bsdf; p.normal = unit_vector(cross(v10 - v00, v01 - v00)); return p;
    }
    // Triangle/Quad::area as their constructors compute it (triangle.h:28,54; quad.h:31): a pure function of the vertices
    float area() const {
        if (type == PRIM_TRIANGLE) return 0.5f * length(cross(v[1] - v[0], v[2] - v[0]));
        return 0.5f * (length(cross(v[1] - v[0], v[3] - v[0])) + length(cross(v[2] - v[1], v[2] - v[3])));
    }
    // area1 / (area1 + area2) of Primitive::sampleUniform's quad split (primitive.h:161-170); unused for triangles
    float sampleAreaRatio() const {
        if (type == PRIM_TRIANGLE) return 1.0f;
        const float area1 = 0.5f * length(cross(v[1] - v[0], v[3] - v[0]));
        const float area2 = 0.5f * length(cross(v[2] - v[1], v[2] - v[3]));
        const float total_area = area1 + area2;
        return area1 / total_area;
    }
    // Primitive::centroid — primitive.h:92-98
    f3 centroid() const {
        if (type == PRIM_TRIANGLE) return div_scalar(v[0] + v[1] + v[2], 3.0f);
        return 0.25f * (v[0] + v[1] + v[2] + v[3]);
    }
};

}  // namespace ptmi
