#include "application_state.h"

#include <algorithm>
#include <array>
#include <chrono>
#include <limits>
#include <cctype>
#include <cstring>
#include <map>

namespace ptmi {

#define PTMI_HIP(call)                                                                                   \
    do {                                                                                                 \
        hipError_t e_ = (call);                                                                          \
        if (e_ != hipSuccess) throw HipError(e_, std::string(#call) + ": " + hipGetErrorString(e_));     \
    } while (0)

void* hipMallocSafe(size_t bytes, const char* name) {
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes ? bytes : 16);
    if (e != hipSuccess) throw HipError(e, std::string("hipMalloc(") + name + ", " + std::to_string(bytes) + " B): " + hipGetErrorString(e));
    return p;
}

// ------------------------------------------------------------------------------------------------
// XORWOW skip-ahead matrices
// ------------------------------------------------------------------------------------------------
namespace {
using Mat = std::vector<uint32_t>;   // 160 rows x 5 words; row b = image of basis state bit b
void stepV(uint32_t v[5]) {
    const uint32_t t = v[0] ^ (v[0] >> 2);
    v[0] = v[1]; v[1] = v[2]; v[2] = v[3]; v[3] = v[4];
    v[4] = (v[4] ^ (v[4] << 4)) ^ (t ^ (t << 1));
}
void apply(const Mat& m, const uint32_t* in, uint32_t* out) {
    uint32_t r[5] = {0, 0, 0, 0, 0};
    for (int b = 0; b < 160; b++)
        if ((in[b >> 5] >> (b & 31)) & 1u) for (int c = 0; c < 5; c++) r[c] ^= m[b * 5 + c];
    std::memcpy(out, r, sizeof r);
}
Mat square(const Mat& m) {
    Mat s(160 * 5);
    for (int b = 0; b < 160; b++) apply(m, &m[b * 5], &s[b * 5]);
    return s;
}
}  // namespace

std::vector<uint32_t> buildXorwowJumpMatrices() {
    Mat m(160 * 5);
    for (int b = 0; b < 160; b++) {
        uint32_t v[5] = {0, 0, 0, 0, 0};
        v[b >> 5] = 1u << (b & 31);
        stepV(v);
        std::memcpy(&m[b * 5], v, sizeof v);
    }
    for (int s = 0; s < 67; s++) m = square(m);          // one subsequence = 2^67 draws
    std::vector<uint32_t> all;
    all.reserve(kXorwowJumpWords);
    for (int k = 0; k < 32; k++) {
        all.insert(all.end(), m.begin(), m.end());
        if (k != 31) m = square(m);
    }
    return all;
}

// ------------------------------------------------------------------------------------------------
// tiling
// ------------------------------------------------------------------------------------------------
int countLocalRows(int height, int n_ranks, int rank, int row_block) {
    int rows = 0;
    for (int y0 = rank * row_block; y0 < height; y0 += n_ranks * row_block) rows += std::min(row_block, height - y0);
    return rows;
}
std::vector<int> localRowMap(const TileMap& tm) {
    std::vector<int> rows;
    for (int y0 = tm.rank * tm.row_block; y0 < tm.height; y0 += tm.n_ranks * tm.row_block)
        for (int y = y0; y < std::min(y0 + tm.row_block, tm.height); y++) rows.push_back(y);
    return rows;
}

// ------------------------------------------------------------------------------------------------
// SceneState
// ------------------------------------------------------------------------------------------------
void SceneState::cleanup() {
    if (d_nodes) (void)hipFree(d_nodes);
    if (d_prims) (void)hipFree(d_prims);
    if (d_mats) (void)hipFree(d_mats);
    if (d_precomputed_cdfs) (void)hipFree(d_precomputed_cdfs);
    if (d_radiosity) (void)hipFree(d_radiosity);
    freePacked();
    freeFast();
    fast_declined = false;
    d_nodes = d_prims = d_mats = nullptr; d_precomputed_cdfs = nullptr; d_radiosity = nullptr;
    h_precomputed_cdfs.clear(); h_radiosity_grids.clear(); h_count_grids.clear(); h_filtered_formfactor.clear(); h_filtered_radiosity.clear();
    d_scene = DeviceScene();
    h_primitives.clear(); bvh_nodes.clear(); bvh_indices.clear();
    num_tris = num_quads = 0; bvh_depth = 0;
}

void SceneState::loadScene(const std::string& filename, int subdivision_count, bool convert_quads) {
    loadSceneHost(filename, subdivision_count, convert_quads);
    upload();
}

void SceneState::loadSceneArrays(std::vector<Primitive> prims) {
    loadSceneArraysHost(std::move(prims));
    upload();
}

void SceneState::loadSceneHost(const std::string& filename, int subdivision_count, bool convert_quads) {
    cleanup();
    const size_t dot = filename.find_last_of('.');
    if (dot == std::string::npos) throw IoError("unsupported file format (no extension): " + filename);
    std::string ext = filename.substr(dot);
    std::transform(ext.begin(), ext.end(), ext.begin(), [](unsigned char c) { return (char)std::tolower(c); });
    std::vector<Primitive> prims;
    if (ext == ".obj") {
        if (!loadOBJ(filename, prims)) throw IoError("failed to load scene: " + filename);
    } else if (ext == ".pbrt") {                                                    // USE_PBRT_LOADER, application_state.h:385-388
        std::string why;
        if (!loadPBRT(filename, prims, &why)) throw IoError("failed to load scene: " + filename + " (" + why + ")");
    } else throw IoError("unsupported file format: " + ext);
    if (convert_quads) prims = convertQuadsToTriangles(prims);
    if (subdivision_count > 0) prims = subdivide_primitives(prims, subdivision_count);
    h_primitives.swap(prims);
    scene_file = filename;
    buildBVH();
}

void SceneState::loadSceneArraysHost(std::vector<Primitive> prims) {
    cleanup();
    if (prims.empty()) throw ArgError("scene has no primitives");
    h_primitives.swap(prims);
    scene_file = "<arrays>";
    buildBVH();
}

void SceneState::buildBVH() {
    const int n = (int)h_primitives.size();
    BVHBuilder builder(h_primitives.data(), n);
    bvh_nodes = builder.nodes;
    bvh_indices = builder.primitive_indices;
    bvh_depth = builder.max_depth;
    num_tris = num_quads = 0;
    for (const Primitive& p : h_primitives) (p.type == PRIM_TRIANGLE ? num_tris : num_quads)++;
}

void SceneState::upload() {
    const int n = (int)h_primitives.size();
    // The kernels' short reciprocal (pt_vec.h: rcp_exact_normal) equals the IEEE quotient while the Moller-Trumbore
    // determinant stays below 2^126; |a| <= 2 |e1| |e2| (|d| = 1), so edge components below 2^60 are always safe.
    for (const Primitive& p : h_primitives)
        for (int k = 0; k < (p.type == PRIM_QUAD ? 4 : 3); k++) {
            const f3 e = p.v[k] - p.v[0];
            const float m = fmaxf(fabsf(e.x), fmaxf(fabsf(e.y), fabsf(e.z)));
            const float c = fabsf(p.v[k].x) + fabsf(p.v[k].y) + fabsf(p.v[k].z);
            if (!(m < 1.0e18f) || !(c < 3.0e38f)) throw ArgError("scene coordinates must be finite and primitive edges shorter than 1e18");
        }
    // ---- SoA re-layout (device_scene.h) ----
    const int stride = num_quads ? 4 : 3;
    auto bits = [](int i) { float f; std::memcpy(&f, &i, 4); return f; };
    std::vector<float4> nodes(2 * bvh_nodes.size()), prims((size_t)stride * n), mats((size_t)3 * n);
    // pre-order skip pointers: skip[i] = i + size of the subtree rooted at i
    std::vector<int> subtree(bvh_nodes.size(), 1);
    for (size_t i = bvh_nodes.size(); i-- > 0;)
        if (!bvh_nodes[i].isLeaf()) subtree[i] = 1 + subtree[bvh_nodes[i].left_child] + subtree[bvh_nodes[i].right_child];
    for (size_t i = 0; i < bvh_nodes.size(); i++) {
        const BVHNode& b = bvh_nodes[i];
        if (!b.isLeaf() && b.left_child != (int)i + 1) throw ArgError("internal: BVH is not in pre-order");
        nodes[2 * i] = make_float4(b.bbox.min.x, b.bbox.min.y, b.bbox.min.z, bits(b.isLeaf() ? b.left_child : (int)i + subtree[i]));
        nodes[2 * i + 1] = make_float4(b.bbox.max.x, b.bbox.max.y, b.bbox.max.z, bits(b.isLeaf() ? -b.prim_count : b.right_child));
    }
    for (int k = 0; k < n; k++) {                     // k = leaf-order slot
        const int src = bvh_indices[k];
        const Primitive& p = h_primitives[src];
        const f3 e1 = p.v[1] - p.v[0], e2 = p.v[2] - p.v[0];
        prims[(size_t)stride * k] = make_float4(p.v[0].x, p.v[0].y, p.v[0].z, bits(p.type == PRIM_QUAD ? 1 : 0));
        prims[(size_t)stride * k + 1] = make_float4(e1.x, e1.y, e1.z, 0.0f);
        prims[(size_t)stride * k + 2] = make_float4(e2.x, e2.y, e2.z, 0.0f);
        if (stride == 4) {
            const f3 e3 = p.type == PRIM_QUAD ? p.v[3] - p.v[0] : mk3(0, 0, 0);
            prims[(size_t)stride * k + 3] = make_float4(e3.x, e3.y, e3.z, 0.0f);
        }
        mats[(size_t)3 * k] = make_float4(p.normal.x, p.normal.y, p.normal.z, bits(src));
        mats[(size_t)3 * k + 1] = make_float4(p.bsdf.x, p.bsdf.y, p.bsdf.z, 0.0f);
        mats[(size_t)3 * k + 2] = make_float4(p.Le.x, p.Le.y, p.Le.z, 0.0f);
    }
    d_nodes = (float4*)hipMallocSafe(nodes.size() * sizeof(float4), "d_nodes");
    d_prims = (float4*)hipMallocSafe(prims.size() * sizeof(float4), "d_prims");
    d_mats = (float4*)hipMallocSafe(mats.size() * sizeof(float4), "d_mats");
    PTMI_HIP(hipMemcpy(d_nodes, nodes.data(), nodes.size() * sizeof(float4), hipMemcpyHostToDevice));
    PTMI_HIP(hipMemcpy(d_prims, prims.data(), prims.size() * sizeof(float4), hipMemcpyHostToDevice));
    PTMI_HIP(hipMemcpy(d_mats, mats.data(), mats.size() * sizeof(float4), hipMemcpyHostToDevice));

    d_scene.nodes = d_nodes; d_scene.prims = d_prims; d_scene.mats = d_mats;
    d_scene.n_nodes = (int)bvh_nodes.size(); d_scene.n_prims = n;
    d_scene.prim_stride = stride; d_scene.has_quads = num_quads ? 1 : 0;
    d_scene.stack_entries = std::min(bvh_depth + 1, 64);
    // LDS residency: the whole scene is staged per workgroup while it leaves room for >= 2 workgroups per CU
    const size_t scene_bytes = (nodes.size() + prims.size() + mats.size()) * sizeof(float4);
    d_scene.lds_resident = (scene_bytes + (size_t)d_scene.stack_entries * kBlock * sizeof(int)) <= 64 * 1024 ? 1 : 0;
    buildPacked();
    // Triangle scenes beyond the sweep's few dozen primitives: the 8-wide tree + the certificate data of TRAVERSAL_CERTIFIED - the
    // default walk there: the reference's hit for every ray, by proof or by its own walk, at 1.4 - 2.3 x the rate of the walk over
    // the reference's tree (128 ... 1 M triangles, planar scenes included; 1 M triangles: +1.3 s of loading, +145 MB)
    if (n > sweep_max_prims && bvh_depth <= 62 && certified_default) {
        try { buildFast(); }
        catch (const ArgError&) { freeFast(); }        // a scene the builder declines (tree too deep for the walk's LDS stack, coordinates of 1e9): the reference's tree is walked
        catch (const HipError&) { freeFast(); (void)hipGetLastError(); }      // no memory for the second tree: the scene still loads
    }
    chooseTraversal();
}

void SceneState::freePacked() {
    void* ptrs[] = {d_gnodes, d_gmats, d_mtab, d_gprims, d_load_index};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    d_gnodes = d_gmats = d_mtab = nullptr; d_gprims = nullptr; d_load_index = nullptr;
    d_scene.gnodes = nullptr; d_scene.n_pos = 0; d_scene.gprims = nullptr; d_scene.gmats = nullptr; d_scene.mtab = nullptr;
    d_scene.load_index = nullptr; d_scene.n_top = 0; d_scene.top_depth = 0;
}

// The node records of the packed layout (csrc/device_scene.h: PACKED LAYOUT): same tree, same visiting order, explicit links,
// sibling pairs adjacent, the nodes of depth <= D first (level by level) while they fit top_records positions.  false: a leaf
// holds more than 7 primitives (its count has 3 bits).
bool packBvhNodes(const std::vector<BVHNode>& bvh_nodes, int top_records, std::vector<float4>& g, int& n_pos_out, int& n_top_out, int& top_depth_out) {
    const int n = (int)bvh_nodes.size();
    for (const BVHNode& b : bvh_nodes) if (b.isLeaf() && b.prim_count > 7) return false;
    auto bits = [](int i) { float f; std::memcpy(&f, &i, 4); return f; };
    std::vector<int> pos((size_t)n, -1), subtree((size_t)n, 1), stack;
    for (int i = n; i-- > 0;)
        if (!bvh_nodes[i].isLeaf()) subtree[i] = 1 + subtree[bvh_nodes[i].left_child] + subtree[bvh_nodes[i].right_child];
    int n_pos = 2, top_depth = -1;                     // root at 0, position 1 pads it to a pair
    pos[0] = 0;
    {
        std::vector<int> level{0}, next_level;
        for (int d = 0; !level.empty(); d++) {
            int pairs = 0;
            for (int x : level) if (!bvh_nodes[x].isLeaf()) pairs++;
            if (n_pos + 2 * pairs > top_records) break;                   // the next level no longer fits the LDS top
            next_level.clear();
            for (int x : level) {
                if (bvh_nodes[x].isLeaf()) continue;
                const int l = bvh_nodes[x].left_child, r = bvh_nodes[x].right_child;
                pos[l] = n_pos; pos[r] = n_pos + 1; n_pos += 2;
                next_level.push_back(l); next_level.push_back(r);
            }
            top_depth = d + 1;
            level.swap(next_level);
        }
    }
    const int n_top = top_records >= 2 ? n_pos : 0;
    stack.push_back(0);
    while (!stack.empty()) {                           // the rest: pairs in pre-order of their parents
        const int x = stack.back(); stack.pop_back();
        if (bvh_nodes[x].isLeaf()) continue;
        const int l = bvh_nodes[x].left_child, r = bvh_nodes[x].right_child;
        if (pos[l] < 0) { pos[l] = n_pos; pos[r] = n_pos + 1; n_pos += 2; }
        stack.push_back(r); stack.push_back(l);
    }
    auto position_of = [&](int pre) { return pre >= n ? n_pos : pos[pre]; };
    const float inf = std::numeric_limits<float>::infinity();
    g.assign((size_t)2 * n_pos, make_float4(inf, inf, inf, 0.0f));
    g[2] = make_float4(inf, inf, inf, bits(0)); g[3] = make_float4(-inf, -inf, -inf, bits(~n_pos));   // padding: an empty leaf nobody links to
    for (int i = 0; i < n; i++) {
        const BVHNode& b = bvh_nodes[i];
        int a_, b_;
        if (b.isLeaf()) { a_ = (b.left_child << 3) | b.prim_count; b_ = ~position_of(i + 1); }
        else { a_ = position_of(i + subtree[i]); b_ = pos[b.left_child]; }
        g[2 * (size_t)pos[i]] = make_float4(b.bbox.min.x, b.bbox.min.y, b.bbox.min.z, bits(a_));
        g[2 * (size_t)pos[i] + 1] = make_float4(b.bbox.max.x, b.bbox.max.y, b.bbox.max.z, bits(b_));
    }
    n_pos_out = n_pos; n_top_out = n_top; top_depth_out = top_depth;
    return true;
}

// The packed layout of csrc/device_scene.h: same tree, same visiting order, records placed so that a ray touches fewer lines.
void SceneState::buildPacked() {
    freePacked();
    const int n = (int)bvh_nodes.size(), n_prims = (int)h_primitives.size();
    if (!d_nodes || d_scene.lds_resident || bvh_depth > 62 || n < packed_min_nodes || n_prims >= (1 << 28)) return;
    auto bits = [](int i) { float f; std::memcpy(&f, &i, 4); return f; };
    std::vector<float4> g;
    int n_pos = 0, n_top = 0, top_depth = -1;
    if (!packBvhNodes(bvh_nodes, packed_top_records, g, n_pos, n_top, top_depth)) return;
    // materials: (normal, table row) per slot + the distinct (Kd, Ke) pairs; load-order index on its own
    std::map<std::array<uint32_t, 6>, int> rows;
    std::vector<float4> gm((size_t)n_prims), tab;
    std::vector<int> load_index((size_t)n_prims);
    std::vector<float> gp;
    if (!num_quads) gp.resize((size_t)9 * n_prims);
    for (int k = 0; k < n_prims; k++) {
        const Primitive& p = h_primitives[bvh_indices[k]];
        std::array<uint32_t, 6> key;
        const float kv[6] = {p.bsdf.x, p.bsdf.y, p.bsdf.z, p.Le.x, p.Le.y, p.Le.z};
        std::memcpy(key.data(), kv, sizeof kv);
        auto it = rows.find(key);
        if (it == rows.end()) {
            it = rows.emplace(key, (int)rows.size()).first;
            tab.push_back(make_float4(p.bsdf.x, p.bsdf.y, p.bsdf.z, 0.0f)); tab.push_back(make_float4(p.Le.x, p.Le.y, p.Le.z, 0.0f));
        }
        gm[k] = make_float4(p.normal.x, p.normal.y, p.normal.z, bits(it->second));
        load_index[k] = bvh_indices[k];
        if (!num_quads) {
            const f3 e1 = p.v[1] - p.v[0], e2 = p.v[2] - p.v[0];         // the float subtraction the reference does per test
            const float rec[9] = {p.v[0].x, p.v[0].y, p.v[0].z, e1.x, e1.y, e1.z, e2.x, e2.y, e2.z};
            std::memcpy(&gp[(size_t)9 * k], rec, sizeof rec);
        }
    }
    auto upload_vec = [&](const void* src, size_t bytes, const char* name) {
        void* d = hipMallocSafe(bytes, name);
        PTMI_HIP(hipMemcpy(d, src, bytes, hipMemcpyHostToDevice));
        return d;
    };
    d_gnodes = (float4*)upload_vec(g.data(), g.size() * sizeof(float4), "d_gnodes");
    d_gmats = (float4*)upload_vec(gm.data(), gm.size() * sizeof(float4), "d_gmats");
    d_mtab = (float4*)upload_vec(tab.data(), tab.size() * sizeof(float4), "d_mtab");
    d_load_index = (int*)upload_vec(load_index.data(), load_index.size() * sizeof(int), "d_load_index");
    if (!num_quads) d_gprims = (float*)upload_vec(gp.data(), gp.size() * sizeof(float), "d_gprims");
    d_scene.gnodes = d_gnodes; d_scene.n_pos = n_pos; d_scene.gprims = d_gprims; d_scene.gmats = d_gmats; d_scene.mtab = d_mtab;
    d_scene.load_index = d_load_index; d_scene.n_top = n_top; d_scene.top_depth = top_depth;
}

void SceneState::freeFast() {
    void* ptrs[] = {d_wnodes, d_wprims, d_wmats, d_wmtab, d_wload_index, d_wref_slot, d_wanc, d_wcert, d_wfast_of_ref, d_wqprims};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    d_wqprims = nullptr; d_scene.wqprims = nullptr;
    d_wnodes = nullptr; d_wprims = nullptr; d_wmats = d_wmtab = nullptr; d_wload_index = d_wref_slot = nullptr;
    d_wanc = nullptr; d_wcert = nullptr; d_wfast_of_ref = nullptr;
    h_wide.clear();
    d_scene.wnodes = nullptr; d_scene.w_nodes = d_scene.w_top = d_scene.w_depth = 0; d_scene.wprims = nullptr; d_scene.wmats = nullptr;
    d_scene.wmtab = nullptr; d_scene.wload_index = nullptr; d_scene.wref_slot = nullptr;
    d_scene.wanc = nullptr; d_scene.wcert = nullptr; d_scene.wfast_of_ref = nullptr; d_scene.w_guard = 0.0f; d_scene.w_big = 0.0f;
}

// The opt-in fast tree (csrc/wide_bvh.h): same triangles, same hit arithmetic, own boxes.  Per-triangle arrays are re-ordered
// into the fast tree's leaf order; wref_slot keeps every triangle's reference leaf-order slot for the equal-t rule.
void SceneState::buildFast() {
    freeFast();
    if (!d_nodes) throw ArgError("fast tree: no scene loaded");
    fast_declined = true;                              // until the build has gone through
    try { buildWideBVH(h_primitives, wide_params, h_wide); }
    catch (const std::exception& e) { h_wide.clear(); throw ArgError(e.what()); }      // every builder failure: the callers' fallback catches ArgError
    // The walk's stack is (levels - 1) x 256 lanes x 8 bytes of the workgroup's LDS, next to the tree's top and - in the radiosity
    // pre-pass - 7 KB of static arrays; a launch may ask for 64 KB.  24 levels = 46 KB leaves room for both (an 8-wide tree of a
    // million triangles has 9 levels); a deeper tree is declined and the scene walks the reference's tree.
    if (h_wide.depth > kWideMaxLevels) { h_wide.clear(); throw ArgError("fast tree: deeper than " + std::to_string(kWideMaxLevels) + " levels"); }
    const int n = (int)h_primitives.size();
    auto bits = [](int i) { float f; std::memcpy(&f, &i, 4); return f; };
    std::vector<int> ref_slot_of_load((size_t)n), ref_slot((size_t)n);
    for (int k = 0; k < n; k++) ref_slot_of_load[bvh_indices[k]] = k;
    std::map<std::array<uint32_t, 6>, int> rows;
    std::vector<float4> wm((size_t)n), tab;
    std::vector<float> wp((size_t)9 * n);
    std::vector<float4> wq(num_quads ? (size_t)4 * n : 0);
    for (int k = 0; k < n; k++) {
        const int li = h_wide.tri_load_index[k];
        const Primitive& p = h_primitives[li];
        ref_slot[k] = ref_slot_of_load[li];
        std::array<uint32_t, 6> key;
        const float kv[6] = {p.bsdf.x, p.bsdf.y, p.bsdf.z, p.Le.x, p.Le.y, p.Le.z};
        std::memcpy(key.data(), kv, sizeof kv);
        auto it = rows.find(key);
        if (it == rows.end()) {
            it = rows.emplace(key, (int)rows.size()).first;
            tab.push_back(make_float4(p.bsdf.x, p.bsdf.y, p.bsdf.z, 0.0f)); tab.push_back(make_float4(p.Le.x, p.Le.y, p.Le.z, 0.0f));
        }
        wm[k] = make_float4(p.normal.x, p.normal.y, p.normal.z, bits(it->second));
        const f3 e1 = p.v[1] - p.v[0], e2 = p.v[2] - p.v[0];             // the float subtraction the reference does per test
        const float rec[9] = {p.v[0].x, p.v[0].y, p.v[0].z, e1.x, e1.y, e1.z, e2.x, e2.y, e2.z};
        std::memcpy(&wp[(size_t)9 * k], rec, sizeof rec);
        if (num_quads) {                                                  // scenes with quads: the 64-byte records of d_prims, fast order
            const f3 e3 = p.type == PRIM_QUAD ? p.v[3] - p.v[0] : mk3(0, 0, 0);
            wq[(size_t)4 * k] = make_float4(p.v[0].x, p.v[0].y, p.v[0].z, bits(p.type == PRIM_QUAD ? 1 : 0));
            wq[(size_t)4 * k + 1] = make_float4(e1.x, e1.y, e1.z, 0.0f);
            wq[(size_t)4 * k + 2] = make_float4(e2.x, e2.y, e2.z, 0.0f);
            wq[(size_t)4 * k + 3] = make_float4(e3.x, e3.y, e3.z, 0.0f);
        }
    }
    auto upload_vec = [&](const void* src, size_t bytes, const char* name) {
        void* d = hipMallocSafe(bytes, name);
        PTMI_HIP(hipMemcpy(d, src, bytes, hipMemcpyHostToDevice));
        return d;
    };
    d_wnodes = (uint4*)upload_vec(h_wide.nodes.data(), h_wide.nodes.size() * sizeof(uint32_t), "d_wnodes");
    d_wprims = (float*)upload_vec(wp.data(), wp.size() * sizeof(float), "d_wprims");
    if (num_quads) d_wqprims = (float4*)upload_vec(wq.data(), wq.size() * sizeof(float4), "d_wqprims");
    d_wmats = (float4*)upload_vec(wm.data(), wm.size() * sizeof(float4), "d_wmats");
    d_wmtab = (float4*)upload_vec(tab.data(), tab.size() * sizeof(float4), "d_wmtab");
    d_wload_index = (int*)upload_vec(h_wide.tri_load_index.data(), (size_t)n * sizeof(int), "d_wload_index");
    d_wref_slot = (int*)upload_vec(ref_slot.data(), (size_t)n * sizeof(int), "d_wref_slot");
    // TRAVERSAL_CERTIFIED: per reference leaf the pre-order indices of its ancestors (the leaf itself first, the root last) in chunks
    // of four, padded with 0xffffffff; per fast-order triangle where its leaf's list starts and how many chunks it has; and the
    // way back from a reference leaf-order slot to the fast order (for the rays that take the reference's walk)
    {
        const int nn = (int)bvh_nodes.size();
        std::vector<int> parent((size_t)nn, -1);
        for (int i = 0; i < nn; i++) if (!bvh_nodes[i].isLeaf()) { parent[bvh_nodes[i].left_child] = i; parent[bvh_nodes[i].right_child] = i; }
        std::vector<uint32_t> anc;                       // 4 per chunk
        std::vector<uint32_t> leaf_ref((size_t)nn, 0u);  // leaf node -> first chunk << 5 | chunks
        std::vector<int> path;
        bool fits = true;
        for (int i = 0; i < nn; i++) {
            if (!bvh_nodes[i].isLeaf()) continue;
            path.clear();
            for (int x = i; x >= 0; x = parent[x]) path.push_back(x);
            const size_t first_chunk = anc.size() / 4, chunks = (path.size() + 3) / 4;
            if (chunks > 31 || first_chunk >= (1u << 27)) { fits = false; break; }
            for (size_t k = 0; k < path.size(); k++) anc.push_back((uint32_t)path[k]);      // the leaf first, the root last
            while (anc.size() % 4) anc.push_back(0xffffffffu);
            leaf_ref[i] = (uint32_t)(first_chunk << 5) | (uint32_t)chunks;
        }
        if (fits && bvh_depth <= 62) {
            std::vector<int> leaf_of_slot((size_t)n, 0), fast_of_ref((size_t)n, 0);
            for (int i = 0; i < nn; i++) if (bvh_nodes[i].isLeaf()) for (int k = 0; k < bvh_nodes[i].prim_count; k++) leaf_of_slot[bvh_nodes[i].left_child + k] = i;
            // one 64-byte record per fast-order triangle - ONE line of memory per hit: the leaf's box as the reference built it + where
            // its ancestor list is (the proof reads these), then what shading reads (wmats' entry: normal, row of the material table)
            std::vector<float4> cert((size_t)kWideCertStride * n);
            for (int k = 0; k < n; k++) {
                const int leaf = leaf_of_slot[ref_slot[k]];
                const AABB& bx = bvh_nodes[leaf].bbox;
                float ref_bits; std::memcpy(&ref_bits, &leaf_ref[leaf], 4);
                cert[(size_t)kWideCertStride * k] = make_float4(bx.min.x, bx.min.y, bx.min.z, ref_bits);
                cert[(size_t)kWideCertStride * k + 1] = make_float4(bx.max.x, bx.max.y, bx.max.z, 0.0f);
                cert[(size_t)kWideCertStride * k + 2] = wm[k];
                cert[(size_t)kWideCertStride * k + 3] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                fast_of_ref[ref_slot[k]] = k;
            }
            d_wanc = (uint4*)upload_vec(anc.data(), anc.size() * sizeof(uint32_t), "d_wanc");
            d_wcert = (float4*)upload_vec(cert.data(), cert.size() * sizeof(float4), "d_wcert");
            d_wfast_of_ref = (int*)upload_vec(fast_of_ref.data(), fast_of_ref.size() * sizeof(int), "d_wfast_of_ref");
        }
    }
    // LDS of a workgroup: the walk's stack (one 8-byte entry per lane and tree level below the root: a group is pushed only
    // while a deeper one is entered) + the top of the tree, whole levels while they fit wide_top_nodes AND six workgroups
    // still share a CU's 160 KB (6 waves per SIMD, what the kernel's 80 registers allow; a 9-level tree with the 8-level
    // tree's top dropped to 5 waves: -6 %)
    const int stack_entries = std::max(h_wide.depth - 1, 1);
    const long long lds_budget = 160 * 1024 / 6 - (long long)stack_entries * kBlock * 8;
    int top = 0;
    for (size_t l = 1; l < h_wide.level_start.size(); l++)
        if (h_wide.level_start[l] <= wide_top_nodes && (long long)h_wide.level_start[l] * kWideNodeDwords * 4 <= lds_budget) top = h_wide.level_start[l];
    d_scene.wnodes = d_wnodes; d_scene.w_nodes = h_wide.n_nodes; d_scene.w_top = top; d_scene.w_depth = stack_entries;
    d_scene.wqprims = d_wqprims;
    d_scene.wprims = d_wprims; d_scene.wmats = d_wmats; d_scene.wmtab = d_wmtab; d_scene.wload_index = d_wload_index; d_scene.wref_slot = d_wref_slot;
    d_scene.wanc = d_wanc; d_scene.wcert = d_wcert; d_scene.wfast_of_ref = d_wfast_of_ref; d_scene.w_guard = h_wide.origin_guard; d_scene.w_big = h_wide.scale;
    if (bounce_lds_bytes_wide(d_scene) > 56 * 1024) { freeFast(); throw ArgError("fast tree: the walk's LDS does not fit a workgroup"); }
    fast_declined = false;
}

void SceneState::setRadiosity(const float* rgb) {
    if (!d_nodes) throw ArgError("setRadiosity: no scene loaded");
    if (d_radiosity) { (void)hipFree(d_radiosity); d_radiosity = nullptr; }
    d_scene.radiosity = nullptr;
    if (!rgb) return;
    const int n = (int)h_primitives.size();
    std::vector<float4> leaf_order((size_t)n);
    for (int k = 0; k < n; k++) {                    // same leaf-order slots as prims/mats
        const float* c = rgb + (size_t)bvh_indices[k] * 3;
        leaf_order[k] = make_float4(c[0], c[1], c[2], 0.0f);
    }
    d_radiosity = (float4*)hipMallocSafe(leaf_order.size() * sizeof(float4), "d_radiosity");
    PTMI_HIP(hipMemcpy(d_radiosity, leaf_order.data(), leaf_order.size() * sizeof(float4), hipMemcpyHostToDevice));
    d_scene.radiosity = d_radiosity;
}

// ---------------------------------------------------------------------------------------------
// RadiosityState (application_state.h:688-787)
// ---------------------------------------------------------------------------------------------
void RadiosityState::cleanup() {
    void* ptrs[] = {(void*)d.geo, (void*)d.slot_of, (void*)d.bsdf, d.radiosity, d.unshot[0], d.unshot[1], d.form_factors, d.grid, d.rad_grid, d.rays, d.row_jump};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    d = RadiosityBuffers();
    is_calculated = false; host_grids_current = false; grids_are_scene_grids = false;
}

void RadiosityState::runSolver(SceneState& scene, const uint32_t* d_jump, bool enable_filtering, bool use_bilateral,
                               float filter_sigma_spatial, float filter_sigma_range, hipStream_t stream, RadiosityStats* stats, bool fast_tree) {
    if (!scene.d_nodes) throw ArgError("runSolver: no scene loaded");
    const int n = (int)scene.h_primitives.size();
    if (n > 46340) throw ArgError("runSolver: more than 46340 primitives (the pair index i * n + j is an int in the reference too)");
    if (num_iterations < 0 || num_iterations > 1000) throw ArgError("runSolver: num_iterations out of range");
    if (mc_samples < 1 || mc_samples > 65536) throw ArgError("runSolver: mc_samples out of range");
    const bool timing = getenv("PTMI_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_start = now();
    cleanup();                                                                       // :703
    const double t_cleanup = now();
    // for (i) setRadiosity(Le), setUnshotRad(Le)  (:697-701) + the load-order geometry the kernels sample
    std::vector<float4> geo((size_t)n * 6), bsdf((size_t)n), rad((size_t)n);
    std::vector<int> slot_of((size_t)n);
    for (int k = 0; k < n; k++) slot_of[scene.bvh_indices[k]] = k;
    for (int p = 0; p < n; p++) {
        const Primitive& pr = scene.h_primitives[p];
        const f3 c = pr.centroid();
        int type = (int)pr.type; float type_bits; std::memcpy(&type_bits, &type, 4);
        geo[6 * p + 0] = make_float4(pr.v[0].x, pr.v[0].y, pr.v[0].z, type_bits);
        geo[6 * p + 1] = make_float4(pr.v[1].x, pr.v[1].y, pr.v[1].z, pr.area());
        geo[6 * p + 2] = make_float4(pr.v[2].x, pr.v[2].y, pr.v[2].z, pr.sampleAreaRatio());
        geo[6 * p + 3] = make_float4(pr.v[3].x, pr.v[3].y, pr.v[3].z, 0.0f);
        geo[6 * p + 4] = make_float4(pr.normal.x, pr.normal.y, pr.normal.z, 0.0f);
        geo[6 * p + 5] = make_float4(c.x, c.y, c.z, 0.0f);
        bsdf[p] = make_float4(pr.bsdf.x, pr.bsdf.y, pr.bsdf.z, 0.0f);
        rad[p] = make_float4(pr.Le.x, pr.Le.y, pr.Le.z, 0.0f);
    }
    auto upload = [&](const void* src, size_t bytes, const char* name) {
        void* p = hipMallocSafe(bytes, name);
        if (src) PTMI_HIP(hipMemcpy(p, src, bytes, hipMemcpyHostToDevice));
        return p;
    };
    d.n = n; d.bvh_depth = scene.bvh_depth;
    // the visibility walk: 0 the reference's own (stackless over its tree), 1 the opt-in fast tree (AppConfig::fast_tree), 2 the
    // certified walk - the fast tree + a per-ray proof that the reference's any-hit walk gives the same answer (radiosity.hip:
    // certified_blocked) - the default for triangle scenes from cert_min_prims primitives up; trees are built on first use
    {
        const bool can = scene.bvh_depth <= 30;
        int walk = 0;
        if (can && fast_tree) walk = 1;
        else if (can && (force_walk >= 2 || (force_walk < 0 && scene.certified_default && n >= cert_min_prims))) walk = 2;
        if (walk) {
            // the automatic choice falls back to the reference's walk for a scene the builder declines (and does not ask again);
            // a walk the caller asked for by name (fast_tree, force_walk) reports the failure
            const bool automatic = !fast_tree && force_walk < 0;
            if (!scene.fastReady() && !(automatic && scene.fast_declined)) {
                try { scene.buildFast(); }
                catch (const ArgError&) { if (!automatic) throw; }
            }
            if (!scene.fastReady()) walk = 0;
            if (walk == 2 && !(scene.d_scene.wcert && scene.d_scene.wanc)) walk = 0;
        }
        d.fast_tree = walk;
    }
    d.geo = (const float4*)upload(geo.data(), geo.size() * sizeof(float4), "d_radiosity_geo");
    d.slot_of = (const int*)upload(slot_of.data(), slot_of.size() * sizeof(int), "d_radiosity_slot_of");
    d.bsdf = (const float4*)upload(bsdf.data(), bsdf.size() * sizeof(float4), "d_radiosity_bsdf");
    d.radiosity = (float4*)upload(rad.data(), rad.size() * sizeof(float4), "d_radiosity_primitives");
    d.unshot[0] = (float4*)upload(rad.data(), rad.size() * sizeof(float4), "d_radiosity_unshot0");
    d.unshot[1] = (float4*)upload(nullptr, rad.size() * sizeof(float4), "d_radiosity_unshot1");
    d.form_factors = (float*)upload(nullptr, (size_t)n * (size_t)n * sizeof(float), "d_form_factors");
    d.grid = (unsigned int*)upload(nullptr, (size_t)n * kGridSize * sizeof(unsigned int), "d_radiosity_grid_counts");
    d.rad_grid = (float4*)upload(nullptr, (size_t)n * kGridSize * sizeof(float4), "d_radiosity_grids");
    d.rays = (unsigned long long*)upload(nullptr, 3 * sizeof(unsigned long long), "d_radiosity_rays");   // rays, certified: chains, fallbacks
    // the part of the pairs' XORWOW skip-ahead that a row shares: one 160 x 160 GF(2) matrix per receiver (3.2 KB; n = 8192: 26 MB;
    // form factors 82.8 -> 79.1 ms there, 19.7 -> 14.9 ms with 4 samples; at n = 2048 the extra kernel costs what it saves)
    if (use_monte_carlo && n >= 4096) d.row_jump = (uint32_t*)upload(nullptr, (size_t)n * 160 * 5 * sizeof(uint32_t), "d_radiosity_row_jump");
    PTMI_HIP(hipMemset(d.rays, 0, 3 * sizeof(unsigned long long)));

    RadiosityParams prm;
    prm.num_iterations = num_iterations; prm.mc_samples = mc_samples; prm.use_monte_carlo = use_monte_carlo ? 1 : 0;
    prm.enable_filtering = enable_filtering ? 1 : 0; prm.use_bilateral = use_bilateral ? 1 : 0;
    prm.filter_sigma_spatial = filter_sigma_spatial; prm.filter_sigma_range = filter_sigma_range;

    struct Events {                                    // destroyed on every exit path
        hipEvent_t e[4] = {nullptr, nullptr, nullptr, nullptr};
        ~Events() { for (hipEvent_t x : e) if (x) (void)hipEventDestroy(x); }
    } events;
    const double t_alloc = now();
    hipEvent_t* ev = events.e;
    for (int k = 0; k < 4; k++) PTMI_HIP(hipEventCreate(&ev[k]));
    PTMI_HIP(hipEventRecord(ev[0], stream));
    DeviceScene ff_scene = scene.d_scene;
    ff_scene.w_cert_debug = force_walk == 3 ? 1 : force_walk == 4 ? 2 : 0;
    launch_form_factors(ff_scene, d, prm, d_jump, stream);                       // :726-741
    PTMI_HIP(hipGetLastError());
    PTMI_HIP(hipEventRecord(ev[1], stream));
    for (int it = 0; it < num_iterations; ++it) launch_radiosity_iteration(d, it & 1, stream);   // :748-771
    final_unshot = num_iterations & 1;
    PTMI_HIP(hipGetLastError());
    PTMI_HIP(hipEventRecord(ev[2], stream));
    if (num_iterations > 0) launch_radiosity_grid(d, prm, stream);
    PTMI_HIP(hipGetLastError());
    PTMI_HIP(hipEventRecord(ev[3], stream));
    PTMI_HIP(hipStreamSynchronize(stream));
    const double t_kernels = now();

    // cudaMemcpy(h_primitives, d_radiosity_primitives, ...) (:773): the solution comes back to the host
    std::vector<float4> h4((size_t)n);
    auto unpack = [&](const float4* dev, size_t count, std::vector<float>& out) {
        PTMI_HIP(hipMemcpy(h4.data(), dev, count * sizeof(float4), hipMemcpyDeviceToHost));
        out.resize(count * 3);
        for (size_t k = 0; k < count; k++) { out[3 * k] = h4[k].x; out[3 * k + 1] = h4[k].y; out[3 * k + 2] = h4[k].z; }
    };
    unpack(d.radiosity, (size_t)n, h_radiosity);
    unpack(d.unshot[final_unshot], (size_t)n, h_unshot);
    h_radiosity_grid.clear(); h_grid.clear(); host_grids_current = false;             // the two big grids stay on the device until asked for
    is_calculated = true;
    if (timing) fprintf(stderr, "[ptmi] runSolver: cleanup %.1f ms, geometry+alloc+upload %.1f ms, kernels %.1f ms, download %.1f ms\n",
                        t_cleanup - t_start, t_alloc - t_cleanup, t_kernels - t_alloc, now() - t_kernels);
    if (stats) {
        float ms = 0.0f;
        *stats = RadiosityStats();
        PTMI_HIP(hipEventElapsedTime(&ms, ev[0], ev[3])); stats->seconds = ms * 1e-3;
        PTMI_HIP(hipEventElapsedTime(&ms, ev[0], ev[1])); stats->form_factor_ms = ms;
        PTMI_HIP(hipEventElapsedTime(&ms, ev[1], ev[2])); stats->iteration_ms = ms;
        PTMI_HIP(hipEventElapsedTime(&ms, ev[2], ev[3])); stats->grid_ms = ms;
        stats->pairs = (uint64_t)n * (uint64_t)n;
        unsigned long long rays[3] = {0, 0, 0};
        PTMI_HIP(hipMemcpy(rays, d.rays, sizeof rays, hipMemcpyDeviceToHost));
        stats->rays = rays[0]; stats->cert_chain = rays[1]; stats->cert_fallback = rays[2]; stats->walk = d.fast_tree;
    }
}

void RadiosityState::fetchGrids() {
    if (!is_calculated) throw ArgError("no radiosity solution");
    if (host_grids_current) return;
    const size_t cells = (size_t)d.n * kGridSize;
    std::vector<float4> h4(cells);
    PTMI_HIP(hipMemcpy(h4.data(), d.rad_grid, cells * sizeof(float4), hipMemcpyDeviceToHost));
    h_radiosity_grid.resize(cells * 3);
    for (size_t k = 0; k < cells; k++) { h_radiosity_grid[3 * k] = h4[k].x; h_radiosity_grid[3 * k + 1] = h4[k].y; h_radiosity_grid[3 * k + 2] = h4[k].z; }
    std::vector<unsigned int> counts(cells);
    PTMI_HIP(hipMemcpy(counts.data(), d.grid, cells * sizeof(unsigned int), hipMemcpyDeviceToHost));
    h_grid.resize(cells);
    for (size_t k = 0; k < cells; k++) h_grid[k] = (float)counts[k];
    host_grids_current = true;
}

void RadiosityState::readFormFactors(float* out) const {
    if (!is_calculated) throw ArgError("no radiosity solution");
    PTMI_HIP(hipMemcpy(out, d.form_factors, (size_t)d.n * (size_t)d.n * sizeof(float), hipMemcpyDeviceToHost));
}

void SceneState::precomputeCDFsDevice(const void* d_src, int src_kind, hipStream_t stream) {
    if (!d_nodes) throw ArgError("precomputeCDFs: no scene loaded");
    if (d_precomputed_cdfs) { (void)hipFree(d_precomputed_cdfs); d_precomputed_cdfs = nullptr; }
    h_precomputed_cdfs.clear();
    d_scene.cdfs = nullptr;
    const int n = (int)h_primitives.size();
    d_precomputed_cdfs = (float*)hipMallocSafe((size_t)n * kCdfDwords * sizeof(float), "d_precomputed_cdfs");
    launch_cdf_records(n, d_src, src_kind, d_precomputed_cdfs, stream);
    PTMI_HIP(hipGetLastError());
    PTMI_HIP(hipStreamSynchronize(stream));
    d_scene.cdfs = d_precomputed_cdfs;
}

const std::vector<float>& SceneState::precomputedCdfsHost() {
    if (h_precomputed_cdfs.empty() && d_precomputed_cdfs) {
        h_precomputed_cdfs.resize(h_primitives.size() * (size_t)kCdfDwords);
        PTMI_HIP(hipMemcpy(h_precomputed_cdfs.data(), d_precomputed_cdfs, h_precomputed_cdfs.size() * sizeof(float), hipMemcpyDeviceToHost));
    }
    return h_precomputed_cdfs;
}

void SceneState::precomputeCDFs(const float* rgb) {
    if (!d_nodes) throw ArgError("precomputeCDFs: no scene loaded");
    if (d_precomputed_cdfs) { (void)hipFree(d_precomputed_cdfs); d_precomputed_cdfs = nullptr; }
    h_precomputed_cdfs.clear();
    d_scene.cdfs = nullptr;
    h_filtered_formfactor.clear(); h_filtered_radiosity.clear();
    if (!rgb) { h_radiosity_grids.clear(); return; }
    const size_t cells = h_primitives.size() * (size_t)kGridSize;
    if (rgb != h_radiosity_grids.data()) h_radiosity_grids.assign(rgb, rgb + cells * 3);
    struct Tmp { void* p = nullptr; ~Tmp() { if (p) (void)hipFree(p); } } d_rgb;
    d_rgb.p = hipMallocSafe(cells * 3 * sizeof(float), "d_radiosity_grids_rgb");
    PTMI_HIP(hipMemcpy(d_rgb.p, rgb, cells * 3 * sizeof(float), hipMemcpyHostToDevice));
    precomputeCDFsDevice(d_rgb.p, 1, nullptr);
}

void SceneState::precomputeCDFsFromFiltered(bool use_bilateral, float sigma_spatial, float sigma_range, hipStream_t stream) {
    if (!d_nodes) throw ArgError("precomputeCDFsFromFiltered: no scene loaded");
    if (h_radiosity_grids.empty()) throw ArgError("precomputeCDFsFromFiltered: the scene has no radiosity grids");
    const int n = (int)h_primitives.size();
    const size_t cells = (size_t)n * kGridSize;
    struct Tmp { void* p = nullptr; ~Tmp() { if (p) (void)hipFree(p); } } rgb, cnt, off, orad;
    rgb.p = hipMallocSafe(cells * 3 * sizeof(float), "d_filter_rgb");
    off.p = hipMallocSafe(cells * sizeof(float), "d_filtered_formfactor");
    orad.p = hipMallocSafe(cells * sizeof(float), "d_filtered_radiosity");
    PTMI_HIP(hipMemcpy(rgb.p, h_radiosity_grids.data(), cells * 3 * sizeof(float), hipMemcpyHostToDevice));
    if (!h_count_grids.empty()) {
        cnt.p = hipMallocSafe(cells * sizeof(float), "d_filter_counts");
        PTMI_HIP(hipMemcpy(cnt.p, h_count_grids.data(), cells * sizeof(float), hipMemcpyHostToDevice));
    }
    launch_filter_pdfs(n, (const float*)rgb.p, (const float*)cnt.p, (float*)off.p, (float*)orad.p, use_bilateral, sigma_spatial, sigma_range, stream);
    PTMI_HIP(hipGetLastError());
    PTMI_HIP(hipStreamSynchronize(stream));
    h_filtered_formfactor.resize(cells); h_filtered_radiosity.resize(cells);
    PTMI_HIP(hipMemcpy(h_filtered_formfactor.data(), off.p, cells * sizeof(float), hipMemcpyDeviceToHost));
    PTMI_HIP(hipMemcpy(h_filtered_radiosity.data(), orad.p, cells * sizeof(float), hipMemcpyDeviceToHost));
    precomputeCDFsDevice(orad.p, 2, stream);
}

// traversal choice (results are identical in all three; see device_scene.h)
void SceneState::chooseTraversal() {
    if (!d_nodes) return;
    if (bvh_depth > 62) d_scene.traversal = TRAVERSAL_STACK;           // the reference's stack-overflow rule can trigger
    else if ((int)h_primitives.size() <= sweep_max_prims) d_scene.traversal = TRAVERSAL_SWEEP;
    else d_scene.traversal = certified_default && d_scene.wnodes && d_scene.wcert && d_scene.wanc ? TRAVERSAL_CERTIFIED : (d_scene.gnodes ? TRAVERSAL_PACKED : TRAVERSAL_PHASED);
    // PHASED: measured faster than the segment-synchronous LANE walk from 128 primitives up (LDS-resident or not); PACKED: the
    // phased walk over the packed layout of scenes too large for LDS; LANE stays available through the override
    if (force_traversal >= 0 && !(force_traversal != TRAVERSAL_STACK && bvh_depth > 62)) d_scene.traversal = force_traversal;
    if (d_scene.traversal == TRAVERSAL_SWEEP && !d_scene.lds_resident) d_scene.traversal = TRAVERSAL_LANE;   // the sweep reads through LDS
    if (d_scene.traversal == TRAVERSAL_PACKED && !d_scene.gnodes) d_scene.traversal = TRAVERSAL_PHASED;
    // CERTIFIED needs the fast tree and the ancestor lists (triangle scenes of depth <= 62); otherwise the exact walk it stands for
    if (d_scene.traversal == TRAVERSAL_CERTIFIED && !(d_scene.wnodes && d_scene.wanc))
        d_scene.traversal = (int)h_primitives.size() <= sweep_max_prims && d_scene.lds_resident ? TRAVERSAL_SWEEP : (d_scene.gnodes ? TRAVERSAL_PACKED : TRAVERSAL_PHASED);
}

// ------------------------------------------------------------------------------------------------
// RenderState
// ------------------------------------------------------------------------------------------------
void RenderState::freeBuffers() {
    void* ptrs[] = {d_state.A, d_state.B, d_state.C, d_state.D, d_state.E, d_state.F, d_image, d_radiance, d_stats, d_frame_color,
                    d_cost[0], d_cost[1], d_cost_max, d_cost_hist, d_queue_ordered};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    d_cost[0] = d_cost[1] = nullptr; d_cost_max = nullptr; d_cost_hist = nullptr; d_queue_ordered = nullptr; cost_valid = false; cost_frame = 0;
    d_frame_color = nullptr; frame_color_frames = 0; batch_frames = 1; batch_spp = 0;
    if (h_image) { (void)hipHostFree(h_image); h_image = nullptr; }
    freeChunks();
    d_state = PathState();
    d_image = nullptr; d_radiance = nullptr; d_stats = nullptr;
    n_local = 0;
}

void RenderState::freeChunks() {
    for (Chunk& c : chunk) {
        void* cp[] = {c.d_queue_init, c.d_queue[0], c.d_queue[1], c.d_count};
        for (void* p : cp) if (p) (void)hipFree(p);
        if (c.h_count) (void)hipHostFree(c.h_count);
        c.d_queue_init = c.d_queue[0] = c.d_queue[1] = c.d_count = nullptr; c.h_count = nullptr; c.d_hcount = nullptr; c.n = 0;
    }
    n_chunks = 0;
}

// chunks: 256-slot blocks dealt round-robin, so a workgroup still reads 256 consecutive state records
void RenderState::setupChunks(int n) {
    freeChunks();
    n_chunks = std::max(1, std::min(n, (int)kMaxChunks));
    std::vector<std::vector<int>> slots(n_chunks);
    for (size_t b = 0; b * kBlock < n_local; b++) {
        std::vector<int>& v = slots[b % n_chunks];
        for (size_t i = b * kBlock; i < std::min(n_local, (b + 1) * (size_t)kBlock); i++) v.push_back((int)i);
    }
    for (int c = 0; c < n_chunks; c++) {
        Chunk& ch = chunk[c];
        ch.n = (int)slots[c].size();
        const size_t cap = std::max<size_t>(slots[c].size(), 1) * sizeof(int);
        ch.d_queue_init = (int*)hipMallocSafe(cap, "chunk.queue_init");
        ch.d_queue[0] = (int*)hipMallocSafe(cap, "chunk.queue0");
        ch.d_queue[1] = (int*)hipMallocSafe(cap, "chunk.queue1");
        // per ring slot i: [4 i] the launch's output count, [4 i + 1] its refill cursor, [4 i + 2] its finished pixels; [4 kCountRing]: the launch's arrival counter
        ch.d_count = (int*)hipMallocSafe((4 * kCountRing + 1) * sizeof(int), "chunk.count");
        PTMI_HIP(hipMemset(ch.d_count, 0, (4 * kCountRing + 1) * sizeof(int)));
        // coherent (fine-grained) host memory: with count publishing the device stores into it while the kernel runs
        PTMI_HIP(hipHostMalloc((void**)&ch.h_count, kCountRing * sizeof(int), hipHostMallocMapped | hipHostMallocCoherent));
        PTMI_HIP(hipHostGetDevicePointer((void**)&ch.d_hcount, ch.h_count, 0));
        if (ch.n) PTMI_HIP(hipMemcpy(ch.d_queue_init, slots[c].data(), slots[c].size() * sizeof(int), hipMemcpyHostToDevice));
    }
}

void RenderState::allocateBuffers() {
    freeBuffers();
    tile.width = width; tile.height = height;
    tile.local_rows = countLocalRows(height, tile.n_ranks, tile.rank, tile.row_block);
    n_local = (size_t)tile.local_rows * (size_t)width;
    tile.tile8 = (allow_tile8 && width % 8 == 0 && tile.local_rows % 8 == 0 && tile.row_block % 8 == 0) ? 1 : 0;
    const size_t n = std::max<size_t>(n_local, 1);
    d_state.A = (float4*)hipMallocSafe(n * sizeof(float4), "state.A");
    d_state.B = (float4*)hipMallocSafe(n * sizeof(float4), "state.B");
    d_state.C = (float4*)hipMallocSafe(n * sizeof(float4), "state.C");
    d_state.D = (float4*)hipMallocSafe(n * sizeof(float4), "state.D");
    d_state.E = (uint4*)hipMallocSafe(n * sizeof(uint4), "state.E");
    d_state.F = (uint2*)hipMallocSafe(n * sizeof(uint2), "state.F");
    d_image = (unsigned char*)hipMallocSafe(n * 3, "d_image");
    d_radiance = (float*)hipMallocSafe(n * 3 * sizeof(float), "d_radiance");
    d_stats = (StatCounters*)hipMallocSafe(sizeof(StatCounters), "d_stats");
    for (int k = 0; k < 2; k++) d_cost[k] = (unsigned int*)hipMallocSafe(n * sizeof(unsigned int), "d_cost");
    d_cost_max = (unsigned int*)hipMallocSafe(2 * sizeof(unsigned int), "d_cost_max");
    d_cost_hist = (int*)hipMallocSafe(512 * sizeof(int), "d_cost_hist");
    d_queue_ordered = (int*)hipMallocSafe(n * sizeof(int), "d_queue_ordered");
    PTMI_HIP(hipHostMalloc((void**)&h_image, n * 3));                     // h_image = new unsigned char[img_size], application_state.h:99
    setupChunks(want_chunks > 0 ? std::min(want_chunks, (int)kMaxChunks) : (n_local >= (size_t)(1 << 18) ? 2 : 1));

    // camera: image size + aspect, then updateCamera (application_state.h:106-109)
    h_camera.image_width = width; h_camera.image_height = height;
    h_camera.aspect = (float)width / (float)height;
    h_camera.updateCamera();

    // render_init (application_state.h:120-122): streams are re-seeded on every (re)allocation
    launch_render_init(tile, d_state, d_jump, seed_base, stream);
    PTMI_HIP(hipGetLastError());
    PTMI_HIP(hipStreamSynchronize(stream));
}

void RenderState::updateResolution(int w, int h, const TileMap* tiling) {
    if (w <= 0 || h <= 0) throw ArgError("width and height must be positive");
    if ((long long)w * h > (1ll << 31) - 1) throw ArgError("frame has more than 2^31-1 pixels");
    if (tiling) {
        if (tiling->n_ranks < 1 || tiling->rank < 0 || tiling->rank >= tiling->n_ranks || tiling->row_block < 1)
            throw ArgError("bad tiling (need n_ranks >= 1, 0 <= rank < n_ranks, row_block >= 1)");
        tile.n_ranks = tiling->n_ranks; tile.rank = tiling->rank; tile.row_block = tiling->row_block;
    } else { tile.n_ranks = 1; tile.rank = 0; tile.row_block = 8; }
    width = w; height = h;
    allocateBuffers();
}

// ------------------------------------------------------------------------------------------------
// ApplicationState / renderFrame
// ------------------------------------------------------------------------------------------------
ApplicationState::ApplicationState(int device) : device_id(device) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) throw HipError(e == hipSuccess ? hipErrorNoDevice : e, "no HIP device available (libptmi has no CPU fallback)");
    if (device < 0 || device >= count) throw ArgError("device_id out of range");
    PTMI_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    PTMI_HIP(hipGetDeviceProperties(&prop, device));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
        throw HipError(hipErrorInvalidDevice, std::string("libptmi is built for gfx950 only; device is ") + prop.gcnArchName);
    n_cus = prop.multiProcessorCount;
    PTMI_HIP(hipStreamCreateWithFlags(&render.stream, hipStreamNonBlocking));
    for (RenderState::Chunk& c : render.chunk) PTMI_HIP(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
    h_jump = buildXorwowJumpMatrices();
    render.d_jump = (uint32_t*)hipMallocSafe(h_jump.size() * sizeof(uint32_t), "d_jump");
    PTMI_HIP(hipMemcpy(render.d_jump, h_jump.data(), h_jump.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    render.h_camera = Sensor(config.camera_origin, config.look_at, config.up, config.fov, 1.0f);   // application.h:107-113
}

ApplicationState::~ApplicationState() {
    (void)hipSetDevice(device_id);
    render.resolve_gate = nullptr;
    dist.finalize();
    for (hipEvent_t ev : event_pool) (void)hipEventDestroy(ev);
    scene.cleanup();
    render.freeBuffers();
    if (render.d_jump) (void)hipFree(render.d_jump);
    if (render.stream) (void)hipStreamDestroy(render.stream);
    for (RenderState::Chunk& c : render.chunk) if (c.stream) (void)hipStreamDestroy(c.stream);
}

void renderFrame(ApplicationState& g, FrameStats* stats) { renderFrames(g, 1, stats); }

void selectFrame(ApplicationState& g, int frame) {
    RenderState& r = g.render;
    if (!r.d_state.A || r.batch_spp <= 0) throw ArgError("selectFrame: nothing rendered yet");
    if (frame < 0 || frame >= r.batch_frames) throw ArgError("selectFrame: frame index outside the last batch");
    PTMI_HIP(hipSetDevice(g.device_id));
    if (r.resolve_gate) PTMI_HIP(hipStreamWaitEvent(r.stream, r.resolve_gate, 0));
    const float4* src = frame == r.batch_frames - 1 ? nullptr : r.d_frame_color + (size_t)frame * r.n_local;
    launch_resolve(r.tile, r.d_state, r.batch_spp, r.d_image, r.d_radiance, r.stream, src);
    PTMI_HIP(hipGetLastError());
    if (r.download_image && r.n_local) PTMI_HIP(hipMemcpyAsync(r.h_image, r.d_image, r.n_local * 3, hipMemcpyDeviceToHost, r.stream));
    PTMI_HIP(hipStreamSynchronize(r.stream));
}

void renderFrames(ApplicationState& g, int n_frames, FrameStats* stats) {
    RenderState& r = g.render;
    if (n_frames < 1 || n_frames > 256) throw ArgError("renderFrames: n_frames must be in [1, 256]");
    if (n_frames > 1 && g.config.spp >= (1 << 16)) throw ArgError("renderFrames: a batch needs spp < 65536");
    if (n_frames > 1 && (unsigned long long)(n_frames - 1) * r.n_local >= (1ull << 31))
        throw ArgError("renderFrames: (n_frames - 1) * local pixels must stay below 2^31");
    if (n_frames > 1 && g.config.current_integrator == IntegratorType::Radiosity)
        throw ArgError("renderFrames: the Radiosity integrator renders one frame per call");
    if (!g.scene.d_nodes) throw ArgError("renderFrame: no scene loaded");
    if (!r.d_state.A) throw ArgError("renderFrame: buffers not allocated (call updateResolution first)");
    if (g.config.spp < 1 || g.config.spp >= (1 << 24)) throw ArgError("spp must be in [1, 2^24)");
    if (g.config.max_depth < 1 || g.config.max_depth > 255) throw ArgError("max_depth must be in [1, 255]");
    PTMI_HIP(hipSetDevice(g.device_id));

    // camera update (application.h:161-163)
    if (g.config.orbit) r.h_camera.updateCameraOrbit(); else r.h_camera.updateCamera();
    const CameraFrame cf = r.h_camera.frame();
    FrameParams fp;
    const f3* src[4] = {&cf.origin, &cf.lower_left_corner, &cf.horizontal, &cf.vertical};
    float* dst[4] = {fp.cam_origin, fp.cam_llc, fp.cam_hor, fp.cam_ver};
    for (int i = 0; i < 4; i++) { dst[i][0] = src[i]->x; dst[i][1] = src[i]->y; dst[i][2] = src[i]->z; }
    fp.spp = g.config.spp; fp.max_depth = g.config.max_depth;
    fp.sampling_mode = (int)g.config.sampling_mode; fp.mis_bsdf_fraction = g.config.mis_bsdf_fraction;
    PTMI_HIP(hipSetDevice(g.device_id));
    if (n_frames > 1) {
        if (r.frame_color_frames < (size_t)(n_frames - 1)) {
            if (r.d_frame_color) { (void)hipFree(r.d_frame_color); r.d_frame_color = nullptr; r.frame_color_frames = 0; }
            r.d_frame_color = (float4*)hipMallocSafe((size_t)(n_frames - 1) * std::max<size_t>(r.n_local, 1) * sizeof(float4), "d_frame_color");
            r.frame_color_frames = (size_t)(n_frames - 1);
        }
        fp.n_frames = n_frames; fp.sample_mask = 0xffffu; fp.frame_color = r.d_frame_color; fp.n_local = (int)r.n_local;
    }
    // what ptmi_select_frame may resolve: nothing until THIS path-tracing run has completed (a Radiosity frame or a run that
    // threw leaves the colour sums of some earlier run behind, and resolving those would overwrite a valid image)
    r.batch_frames = 1; r.batch_spp = 0;

    const int n_local = (int)r.n_local;
    // segments per launch: 32 while the device has more waves to run than it holds at once; once the pixels still active fit
    // (an eighth of a 2048^2 frame per GPU from its first launch on; the last stretch of any other frame), the frame is as
    // long as the chain of its heaviest pixels, every launch boundary makes it wait for the slowest wave once more, and the
    // rest of the frame goes into ONE launch per chunk (1 M-triangle scene, 1/8 of the frame: 47 instead of 59 ms; with more
    // waves than slots one launch is slower, the second round starts a whole chain late) - DESIGN.md 5
    // (the phased kernels: spp / 4, see below)
    int segments = g.config.segments_per_launch > 0 ? g.config.segments_per_launch : 32;
    // "the rest of the frame": bounded so that one launch stays well below a minute (a lane of the 1 M-triangle scene does
    // ~4 000 segments per second; BASELINE's 2048 spp x 8 bounces = 16 384 segments at most = ONE 1.45 s launch per chunk.  A
    // boundary in mid-frame is dear: with 8 192 the same frame took 1.58 s)
    constexpr int kRestOfFrameSegments = 65536;
    // The sweep's cost per segment does not shrink with its living lanes, so its waves want the compaction of every 32nd
    // segment for longer: with the phased kernels' threshold c2 loses 7 %, c3 8 %; at 0.3 x the wave slots a small frame
    // gains (cbox 256^2 +14 %, 362^2 +15 %; 512^2 = 0.5 x the slots -11 % with one launch) and c2's last stretch +0.5 %.
    // the opt-in fast tree: built at the first frame that asks for it; quad scenes keep the exact walk
    if (g.config.fast_tree && !g.scene.fastReady() && g.config.current_integrator == IntegratorType::PathTracing)
        g.scene.buildFast();
    DeviceScene scene = g.scene.d_scene;
    if (g.config.fast_tree && g.scene.fastReady()) scene.traversal = TRAVERSAL_WIDE;
    if (scene.traversal == TRAVERSAL_CERTIFIED && !(g.scene.fastReady() && scene.wanc)) scene.traversal = scene.gnodes ? TRAVERSAL_PACKED : TRAVERSAL_PHASED;
    const int trav = scene.traversal;
    const bool phased = trav == TRAVERSAL_PHASED || trav == TRAVERSAL_PACKED || trav == TRAVERSAL_WIDE || trav == TRAVERSAL_CERTIFIED;
    // The phased kernels keep their lanes busy across sample boundaries, so a launch boundary buys them only the compaction; with
    // many samples per pixel fewer, longer launches win: spp / 4 segments, at least 32, at most 512 (1 M-triangle scene, certified
    // walk, Msamples/s with 32 / 128 / 512 segments: an eighth of the frame at 2048 spp 1 787 / 1 895 / 1 943, at 256 spp 1 706 /
    // 1 801 / -; the whole frame at 512 spp 2 444 / 2 531 / 2 514 (256), at 64 spp 2 341 / 2 267 / -)
    if (phased && g.config.segments_per_launch <= 0) segments = std::min(512, std::max(32, g.config.spp / 4));
    // "the rest of the frame" for the 8-wide walks: at most 512 segments, then compaction - what a 64-spp frame has left at that
    // point anyway; with BASELINE's 2048 spp the pixels of a tile finish far apart, and re-packing the living ones every 512
    // segments beats one launch to the end (an eighth of the 1 M-triangle frame, 2048 spp: 1 943 against 1 824 Msamples/s)
    const int rest_segments = trav == TRAVERSAL_WIDE || trav == TRAVERSAL_CERTIFIED ? std::max(segments, 512) : kRestOfFrameSegments;
    const long long fit_pct = phased ? 120 : 30;
    const long long wave_slots = g.config.segments_per_launch > 0 || !(phased || trav == TRAVERSAL_SWEEP)
                                     ? 0 : bounce_resident_waves(scene, fp, g.config.collect_stats, g.n_cus);
    hipStream_t s = r.stream;
    // whatever way this function is left, nothing of this frame is still in flight (an exception thrown between two
    // launches must not let the next frame start on top of the chunk streams' queued work)
    struct Drain {
        RenderState& r; bool armed = true;
        ~Drain() {
            if (!armed) return;
            for (int c = 0; c < RenderState::kMaxChunks; c++) if (r.chunk[c].stream) (void)hipStreamSynchronize(r.chunk[c].stream);
            (void)hipStreamSynchronize(r.stream);
        }
    } drain{r};
    const bool want_stats = g.config.collect_stats;
    // Count publishing (device_scene.h: CountPublish): the launch's last wave stores the output count to
    // pinned host memory and zeroes the next counter, so a chunk's stream carries kernels only - no fill and no 4-byte copy
    // between two launches (profiles/r02_kernel_stats_c5frame.csv: copyBuffer 11.2 % of the summed GPU time).  Built, parity
    // green (100 GPU tests with it on), and measured SLOWER where it was meant to help: whole 1 M-triangle frame, exact walk
    // 245.5 -> 262.0 ms, fast tree 120.5 -> 125.4 ms; an eighth of the frame +-0.  The blits' time is waiting
    // time behind the other chunk's kernel, not GPU time the frame is short of.  Off by default; PTMI_PUBLISH=1 turns it on.
    bool publish = false;
    if (const char* e = getenv("PTMI_PUBLISH")) publish = e[0] == '1';
    // Refill (device_scene.h: LaunchSchedule): a launch of the 8-wide walks has at most as many waves as the device holds at once, and
    // a lane whose pixel has had its visit takes the next queued pixel - no wave waits for a slot, no lane idles while pixels are
    // queued, and the frame needs no launch boundary to re-pack its lanes before the queue has run dry
    // (the sweep of the small LDS-resident scenes does not gain: with refill its waves lose the coherence their shared walk lives on -
    // c2 5 685 Msamples/s in image order, 6 790 in cost order, against 6 717 for its 32-segment launches on the same box)
    bool refill = (trav == TRAVERSAL_WIDE || trav == TRAVERSAL_CERTIFIED) && g.config.segments_per_launch <= 0 && wave_slots > 0;
    const int refill_segments = kRestOfFrameSegments;
    // launch order by last frame's cost (below): in 16 classes, and only while the frame has at most three pixels per lane of the
    // launch - an eighth of the 1 M-triangle frame (1.3 per lane) gains 9 % at 64 spp and 12 % at 2048 spp; the whole frame (10.7
    // per lane) has no tail to speak of and loses 6 % with its launch order torn from the image order (neighbouring waves share
    // the lines of the scene they fetch).  Classes 2 / 4 / 8 / 16 / 32 / 64 / 256 on the eighth at 64 spp: 1 681 / 1 685 / 1 879 / 1 897 /
    // 1 875 / 1 860 / 1 819 Msamples/s against 1 735 in image order.
    bool order_by_cost = (long long)r.n_local <= 3ll * 64 * wave_slots;
    int order_classes = 16;
    if (const char* e = getenv("PTMI_REFILL")) refill = refill && e[0] != '0';                      // A/B hooks of round 4 (tools/occupancy_probe.py)
    if (const char* e = getenv("PTMI_ORDER")) { order_by_cost = e[0] != '0'; if (atoi(e) > 1) order_classes = atoi(e); }
    // With refill one launch keeps every wave slot busy by itself: a second chunk's kernel only competes with it (an eighth of the
    // 1 M-triangle frame 1 661 -> 1 716 Msamples/s with one chunk, the whole frame 2 600 -> 2 795; three chunks: 1 628 / 2 501).  The
    // automatic choice follows the walk; a forced count (config.streams) stays.
    if (r.want_chunks == 0 && g.config.current_integrator == IntegratorType::PathTracing) {
        const int want = refill ? 1 : (r.n_local >= (size_t)(1 << 18) ? 2 : 1);
        if (want != r.n_chunks) { PTMI_HIP(hipStreamSynchronize(r.stream)); r.setupChunks(want); }
    }

    auto event = [&](size_t i) {
        while (g.event_pool.size() <= i) { hipEvent_t ev; PTMI_HIP(hipEventCreate(&ev)); g.event_pool.push_back(ev); }
        return g.event_pool[i];
    };
    size_t n_ev = 0;
    const hipEvent_t ev_begin = event(n_ev++);
    PTMI_HIP(hipEventRecord(ev_begin, s));
    if (want_stats) PTMI_HIP(hipMemsetAsync(r.d_stats, 0, sizeof(StatCounters), s));

    if (g.config.current_integrator == IntegratorType::Radiosity) {       // application.h:193-197
        if (r.resolve_gate) PTMI_HIP(hipStreamWaitEvent(s, r.resolve_gate, 0));
        launch_render_radiosity(g.scene.d_scene, r.tile, r.d_state, fp, r.d_image, r.d_radiance, s);
        PTMI_HIP(hipGetLastError());
        const hipEvent_t ev_done = event(n_ev++);
        PTMI_HIP(hipEventRecord(ev_done, s));
        if (r.download_image && n_local) PTMI_HIP(hipMemcpyAsync(r.h_image, r.d_image, (size_t)n_local * 3, hipMemcpyDeviceToHost, s));
        PTMI_HIP(hipStreamSynchronize(s));
        drain.armed = false;
        if (stats) {
            float ms = 0.0f;
            PTMI_HIP(hipEventElapsedTime(&ms, ev_begin, ev_done));
            *stats = FrameStats();
            stats->seconds = ms * 1e-3;
            stats->samples = (uint64_t)n_local * (uint64_t)g.config.spp;
        }
        return;
    }

    launch_frame_begin(r.tile, r.d_state, fp, s);

    // Launch order of a refill frame: the pixels that took the most segments in the last frame first.  A pixel has ONE path in
    // flight, so the heaviest pixels are as long as the frame whatever the GPU does meanwhile; started last - in image order the
    // top rows of the tile are taken when the first background pixels are through - they end a background pixel's length after
    // the rest.  The costs are the last frame's (same pixels, usually the same view); a first frame runs in image order.
    const int* first_queue = nullptr;
    unsigned int *cost_now = nullptr, *cost_max_now = nullptr;
    if (refill && r.n_chunks == 1 && n_frames == 1 && order_by_cost) {
        const int cur = (int)(r.cost_frame & 1), prev = cur ^ 1;
        if (r.cost_valid) {
            launch_order_by_cost(r.chunk[0].d_queue_init, r.chunk[0].n, r.d_cost[prev], r.d_cost_max + prev, r.d_cost_hist, r.d_queue_ordered, order_classes, s);
            first_queue = r.d_queue_ordered;
        }
        PTMI_HIP(hipMemsetAsync(r.d_cost[cur], 0, std::max<size_t>(r.n_local, 1) * sizeof(unsigned int), s));
        PTMI_HIP(hipMemsetAsync(r.d_cost_max + cur, 0, sizeof(unsigned int), s));
        cost_now = r.d_cost[cur]; cost_max_now = r.d_cost_max + cur;
        r.cost_valid = false;                          // until this frame is through
    }

    // queue-driven loop: every launch advances each active pixel of its chunk by `segments` ray segments and compacts.
    // A launch reads its exact input count from device memory (the previous launch's output counter), so the host
    // does not wait for each count: per chunk it runs kRunAhead launches ahead, sizing grids with the newest count it
    // HAS seen (counts only shrink), and stops once a count of 0 has come back; the launches already queued behind it
    // find an empty queue and exit.  The chunks' streams run concurrently.
    constexpr int kRunAhead = 2, kRing = RenderState::kCountRing;   // count slots: launch i reads (i-1) % kRing, writes i % kRing
    struct Run { int bound, last_out = 0, issued = 0, retired = 0; bool finished; hipEvent_t done[kRing]; };
    Run run[RenderState::kMaxChunks];
    const hipEvent_t ev_ready = event(n_ev++);
    PTMI_HIP(hipEventRecord(ev_ready, s));             // frame_begin (and the stats reset) precede every chunk's first launch
    for (int c = 0; c < r.n_chunks; c++) {
        run[c].bound = r.chunk[c].n; run[c].finished = r.chunk[c].n == 0;
        for (int i = 0; i < kRing; i++) run[c].done[i] = event(n_ev++);
        PTMI_HIP(hipStreamWaitEvent(r.chunk[c].stream, ev_ready, 0));
        if (publish) PTMI_HIP(hipMemsetAsync(r.chunk[c].d_count, 0, (4 * kRing + 1) * sizeof(int), r.chunk[c].stream));   // once per frame
    }
    uint64_t launches = 0, visits = 0;
    const size_t first_pair_event = n_ev;
    auto busy = [&] { for (int c = 0; c < r.n_chunks; c++) if (!run[c].finished || run[c].retired < run[c].issued) return true; return false; };
    while (busy()) {
        for (int c = 0; c < r.n_chunks; c++) {
            Run& u = run[c]; RenderState::Chunk& ch = r.chunk[c];
            // (a refill launch takes the frame to its end: nothing to run ahead with)
            while (!u.finished && u.issued - u.retired < (refill ? 1 : kRunAhead)) {
                const int slot_out = u.issued % kRing;
                CountPublish pub;
                if (publish) {
                    pub.done_count = ch.d_count + 4 * kRing; pub.next_count = ch.d_count + 4 * ((slot_out + 1) % kRing); pub.host_count = ch.d_hcount + slot_out;
                    __atomic_store_n(ch.h_count + slot_out, -1, __ATOMIC_RELEASE);      // "not there yet"
                } else PTMI_HIP(hipMemsetAsync(ch.d_count + 4 * slot_out, 0, 4 * sizeof(int), ch.stream));     // output count, refill cursor, finished pixels
                const hipEvent_t e0 = stats ? event(n_ev++) : nullptr;
                if (stats) PTMI_HIP(hipEventRecord(e0, ch.stream));
                long long active = 0;                   // pixels still in flight, as far as the host has seen (counts only shrink)
                for (int k = 0; k < r.n_chunks; k++) active += run[k].finished ? 0 : run[k].bound;
                const bool fits = wave_slots > 0 && (active + 63) / 64 * 100 <= wave_slots * fit_pct;
                LaunchSchedule sched;
                if (refill && active > 0) {
                    // this chunk's share of the device's wave slots, by its share of the pixels still in flight (rounded down: the
                    // chunks together must not ask for more waves than fit at once, or the surplus starts a whole launch late)
                    sched.max_waves = std::max(4, (int)(wave_slots * (long long)u.bound / active) & ~3);
                    sched.cursor = ch.d_count + 4 * slot_out + 1;
                    sched.cost = cost_now; sched.cost_max = cost_max_now;
                }
                launch_bounce(scene, r.tile, r.d_state, fp, u.issued == 0 ? (first_queue ? first_queue : ch.d_queue_init) : ch.d_queue[(u.issued - 1) & 1], u.bound,
                              u.issued == 0 ? nullptr : ch.d_count + 4 * ((u.issued - 1) % kRing), ch.d_queue[u.issued & 1], ch.d_count + 4 * slot_out,
                              refill ? refill_segments : fits ? rest_segments : segments, want_stats ? r.d_stats : nullptr,
                              phased && wave_slots > 0 && (active + 63) / 64 >= 2 * wave_slots, ch.stream, pub, sched);
                PTMI_HIP(hipGetLastError());           // launch-time failures (bad LDS size, ...) surface here, not a frame later
                const hipEvent_t e1 = stats ? event(n_ev++) : nullptr;
                if (stats) PTMI_HIP(hipEventRecord(e1, ch.stream));
                if (!publish) {
                    PTMI_HIP(hipMemcpyAsync(ch.h_count + slot_out, ch.d_count + 4 * slot_out, sizeof(int), hipMemcpyDeviceToHost, ch.stream));
                    PTMI_HIP(hipEventRecord(u.done[slot_out], ch.stream));
                }
                u.issued++;
            }
        }
        for (int c = 0; c < r.n_chunks; c++) {         // retire each chunk's oldest outstanding launch: its count is the new bound
            Run& u = run[c]; RenderState::Chunk& ch = r.chunk[c];
            if (u.retired == u.issued) continue;
            const int slot = u.retired % kRing;
            if (publish) {
                // the launch's last workgroup stores the count here; the stream is asked now and then so that a launch that
                // died (or never ran) surfaces as an error instead of an endless wait
                unsigned long long spins = 0;
                while (__atomic_load_n(ch.h_count + slot, __ATOMIC_ACQUIRE) < 0) {
                    if ((++spins & 0xffffu) == 0) {
                        const hipError_t q = hipStreamQuery(ch.stream);
                        if (q != hipErrorNotReady && __atomic_load_n(ch.h_count + slot, __ATOMIC_ACQUIRE) < 0) {
                            if (q != hipSuccess) throw HipError(q, std::string("bounce launch failed: ") + hipGetErrorString(q));
                            throw HipError(hipErrorUnknown, "bounce launch finished without publishing its count");
                        }
                    }
                    __builtin_ia32_pause();
                }
            } else PTMI_HIP(hipEventSynchronize(u.done[slot]));
            const int out_count = ch.h_count[slot];
            launches++;                                // every issued launch counts (rocprof sees the trailing empty ones too)
            if (u.retired == 0 || u.last_out > 0) visits += (uint64_t)(u.retired == 0 ? ch.n : u.last_out);
            u.last_out = out_count; u.bound = out_count;
            u.retired++;
            if (out_count == 0) u.finished = true;
        }
    }
    for (int c = 0; c < r.n_chunks; c++) {             // the resolve pass waits for every chunk
        const hipEvent_t ev = event(n_ev++);
        PTMI_HIP(hipEventRecord(ev, r.chunk[c].stream));
        PTMI_HIP(hipStreamWaitEvent(s, ev, 0));
    }
    const size_t after_pairs = n_ev;
#ifndef PTMI_EXPERIMENT_NO_RESOLVE_GATE      // (never defined in the shipped build: make ab-host-lib, to see the asynchronous-exchange test fail without it)
    if (r.resolve_gate) PTMI_HIP(hipStreamWaitEvent(s, r.resolve_gate, 0));   // a gather of the previous frame may still read the tile
#endif
    launch_resolve(r.tile, r.d_state, g.config.spp, r.d_image, r.d_radiance, s);
    const hipEvent_t ev_end = event(n_ev++);
    PTMI_HIP(hipEventRecord(ev_end, s));
    // cudaMemcpy(h_image, d_image, img_size, DeviceToHost), application.h:211 ("Memory Transfer" stage)
    if (r.download_image && n_local) PTMI_HIP(hipMemcpyAsync(r.h_image, r.d_image, (size_t)n_local * 3, hipMemcpyDeviceToHost, s));
    PTMI_HIP(hipStreamSynchronize(s));                 // cudaDeviceSynchronize, application.h:199
    PTMI_HIP(hipGetLastError());
    drain.armed = false;
    r.batch_frames = n_frames; r.batch_spp = g.config.spp;
    if (cost_now) { r.cost_valid = true; r.cost_frame++; }

    if (stats) {
        float ms = 0.0f;
        PTMI_HIP(hipEventElapsedTime(&ms, ev_begin, ev_end));
        stats->seconds = ms * 1e-3;
        double kms = 0.0;
        for (size_t i = first_pair_event; i + 1 < after_pairs - (size_t)r.n_chunks; i += 2) {
            float k = 0.0f;
            PTMI_HIP(hipEventElapsedTime(&k, g.event_pool[i], g.event_pool[i + 1]));
            kms += k;
        }
        stats->bounce_kernel_ms = kms;
        stats->bounce_launches = launches;
        stats->path_visits = visits;
        stats->samples = (uint64_t)n_local * (uint64_t)g.config.spp * (uint64_t)n_frames;
        if (want_stats) {
            StatCounters c;
            PTMI_HIP(hipMemcpy(&c, r.d_stats, sizeof c, hipMemcpyDeviceToHost));
            stats->rays = c.rays; stats->node_visits = c.node_visits; stats->prim_tests = c.prim_tests; stats->hits = c.hits;
            stats->top_node_visits = c.top_node_visits; stats->cert_chain = c.cert_chain; stats->cert_fallback = c.cert_fallback;
        }
    }
}

}  // namespace ptmi
