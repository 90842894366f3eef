#include "bvh.h"

#include <algorithm>
#include <cmath>

namespace ptmi {

void AABB::grow(const AABB& o) {                // AABB::merge, bvh.h:30-35
    min = mk3(fminf(min.x, o.min.x), fminf(min.y, o.min.y), fminf(min.z, o.min.z));
    max = mk3(fmaxf(max.x, o.max.x), fmaxf(max.y, o.max.y), fmaxf(max.z, o.max.z));
}

BVHBuilder::BVHBuilder(const Primitive* prims, int count) : primitives(prims) {
    nodes.reserve(static_cast<size_t>(count) * 2);
    primitive_indices.resize(count);
    for (int i = 0; i < count; i++) primitive_indices[i] = i;
    buildRecursive(0, count, 1);
}

static inline float axis(const f3& v, int a) { return a == 0 ? v.x : (a == 1 ? v.y : v.z); }

AABB BVHBuilder::computeBounds(int start, int end) const {   // bvh.h:108-148
    AABB bounds;
    const float eps = 1e-6f;
    for (int i = start; i < end; i++) {
        const Primitive& p = primitives[primitive_indices[i]];
        AABB b;
        float lo[3], hi[3];
        for (int a = 0; a < 3; a++) {
            const float c0 = axis(p.v[0], a), c1 = axis(p.v[1], a), c2 = axis(p.v[2], a);
            if (p.type == PRIM_TRIANGLE) {
                lo[a] = fminf(fminf(c0, c1), c2);
                hi[a] = fmaxf(fmaxf(c0, c1), c2);
            } else {
                const float c3 = axis(p.v[3], a);
                lo[a] = fminf(fminf(c0, c1), fminf(c2, c3));
                hi[a] = fmaxf(fmaxf(c0, c1), fmaxf(c2, c3));
            }
        }
        b.min = mk3(lo[0] - eps, lo[1] - eps, lo[2] - eps);
        b.max = mk3(hi[0] + eps, hi[1] + eps, hi[2] + eps);
        bounds.grow(b);
    }
    return bounds;
}

int BVHBuilder::buildRecursive(int start, int end, int depth) {   // bvh.h:154-218
    const int node_idx = static_cast<int>(nodes.size());
    nodes.emplace_back();
    max_depth = std::max(max_depth, depth);

    const AABB bbox = computeBounds(start, end);
    const int count = end - start;
    auto make_leaf = [&] {
        nodes[node_idx].bbox = bbox;
        nodes[node_idx].left_child = start;
        nodes[node_idx].prim_count = count;
        return node_idx;
    };
    if (count <= 4) return make_leaf();

    AABB cb;
    for (int i = start; i < end; i++) {
        const f3 c = primitives[primitive_indices[i]].centroid();
        AABB one; one.min = c; one.max = c;
        cb.grow(one);
    }
    const f3 extent = cb.max - cb.min;
    int best_axis = 0;
    if (extent.y > extent.x) best_axis = 1;
    if (extent.z > axis(extent, best_axis)) best_axis = 2;
    if (axis(extent, best_axis) < 1e-6f) return make_leaf();       // all centroids coincide: oversized leaf

    const float split_pos = axis(0.5f * (cb.min + cb.max), best_axis);   // AABB::center(), bvh.h:21-23
    int mid = start;
    for (int i = start; i < end; i++) {
        if (axis(primitives[primitive_indices[i]].centroid(), best_axis) < split_pos) {
            std::swap(primitive_indices[i], primitive_indices[mid]);
            mid++;
        }
    }
    if (mid == start || mid == end) mid = start + count / 2;

    const int left_idx = buildRecursive(start, mid, depth + 1);
    const int right_idx = buildRecursive(mid, end, depth + 1);
    nodes[node_idx].bbox = bbox;
    nodes[node_idx].left_child = left_idx;
    nodes[node_idx].right_child = right_idx;
    nodes[node_idx].prim_count = 0;
    return node_idx;
}

}  // namespace ptmi
