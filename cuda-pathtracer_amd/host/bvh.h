// bvh.h — host BVH builder reproducing the reference's tree (rendering/bvh.h:76-219).
//
// The tree SHAPE and the leaf order are part of the render result: rays through a
// shared edge hit two coplanar primitives at the same t and the first one visited
// wins (scene.h:90), and cbox.obj's left wall carries two different normals.  So
// the split rule, the unstable swap partition and the pre-order numbering are kept
// exactly; only the storage changes (flat arrays, later re-laid-out SoA for the GPU).
#pragma once
#include <vector>

#include "primitive.h"

namespace ptmi {

struct AABB {                                   // bvh.h:13-35
    f3 min = {1e30f, 1e30f, 1e30f};
    f3 max = {-1e30f, -1e30f, -1e30f};
    void grow(const AABB& o);
};

struct BVHNode {                                // bvh.h:63-72
    AABB bbox;
    int left_child = -1;                        // inner: child index; leaf: first slot in primitive_indices
    int right_child = -1;
    int prim_count = 0;                         // > 0 marks a leaf
    bool isLeaf() const { return prim_count > 0; }
};

class BVHBuilder {
public:
    std::vector<BVHNode> nodes;                 // pre-order
    std::vector<int> primitive_indices;         // leaf order -> load order
    int max_depth = 0;                          // root = 1; bounds the traversal stack (depth + 1 entries)

    BVHBuilder(const Primitive* prims, int count);

private:
    const Primitive* primitives;
    AABB computeBounds(int start, int end) const;
    int buildRecursive(int start, int end, int depth);
};

}  // namespace ptmi
