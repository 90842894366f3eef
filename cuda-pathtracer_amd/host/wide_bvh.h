// wide_bvh.h (host) — builder and host-side walk of the opt-in FAST tree (csrc/wide_bvh.h: layout, scope, what may differ).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../csrc/wide_bvh.h"
#include "primitive.h"

namespace ptmi {

struct WideBVHParams {
    int max_leaf = 3;          // triangles per leaf child, 1..kWideMaxLeaf
    float c_trav = 1.0f;       // SAH: cost of one more box level relative to ...
    float c_tri = 1.0f;        // ... one triangle test
    int bins = 16;
};

struct WideBVH {
    std::vector<uint32_t> nodes;        // kWideNodeDwords per node, breadth-first order
    std::vector<int> tri_load_index;    // fast order -> load-order primitive index
    std::vector<int> level_start;       // first node index of every level (+ a final entry = n_nodes)
    int n_nodes = 0, depth = 0;         // depth: levels of wide nodes (root = 1) = most entries the walk's stack ever holds + 1
    float origin_guard = 0.0f;          // the boxes are padded for ray origins with |coordinate| <= this (4 x scale)
    float scale = 0.0f;                 // the scene's largest coordinate, or 8 x the 99th percentile of its primitives' when a few lie far outside
    double sah = 0.0;                   // for inspection: sum of (child area / root area) over all children of all nodes
    int binary_nodes = 0, binary_leaves = 0;
    int fill_hist[9] = {}, leaf_hist[4] = {};   // nodes by number of children; leaf children by number of triangles
    bool empty() const { return n_nodes == 0; }
    void clear() { *this = WideBVH(); }
};

// Triangles and quads (a quad is ONE primitive of a leaf), in load order.  Throws std::invalid_argument for an empty scene or
// coordinates of 1e9 and beyond; the caller (SceneState::buildFast) turns every builder failure into ArgError.
void buildWideBVH(const std::vector<Primitive>& prims, const WideBVHParams& prm, WideBVH& out);

// The fast walk on the host, decision for decision what ptmi_bounce_wide does per lane: closest hit of one ray.
// ref_slot[load index] = reference leaf-order slot (tie rule).  Returns the load-order index of the hit triangle or -1.
struct WideWalkCounters { uint64_t node_visits = 0, prim_tests = 0, max_stack = 0; };
int wideIntersectHost(const WideBVH& bvh, const std::vector<Primitive>& prims, const std::vector<int>& ref_slot, f3 o, f3 d,
                      float t_min, float t_max, float& t_hit, WideWalkCounters* cn);

}  // namespace ptmi
