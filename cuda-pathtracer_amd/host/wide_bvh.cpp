// wide_bvh.cpp — builder of the opt-in FAST tree: binned-SAH binary tree -> 8-wide collapse -> slot assignment by octant ->
// breadth-first numbering -> conservative 8-bit quantisation (csrc/wide_bvh.h).  Own code; nothing here has a counterpart in
// the reference, whose only tree is the midpoint-split binary one of rendering/bvh.h (host/bvh.cpp reproduces that one).
#include "wide_bvh.h"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstring>
#include <limits>
#include <stdexcept>

namespace ptmi {

namespace {

struct Box {
    float lo[3] = {std::numeric_limits<float>::infinity(), std::numeric_limits<float>::infinity(), std::numeric_limits<float>::infinity()};
    float hi[3] = {-std::numeric_limits<float>::infinity(), -std::numeric_limits<float>::infinity(), -std::numeric_limits<float>::infinity()};
    void grow(const Box& o) { for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], o.lo[a]); hi[a] = std::max(hi[a], o.hi[a]); } }
    void grow(const float* p) { for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], p[a]); hi[a] = std::max(hi[a], p[a]); } }
    double area() const {
        const double x = (double)hi[0] - lo[0], y = (double)hi[1] - lo[1], z = (double)hi[2] - lo[2];
        return x < 0 ? 0.0 : 2.0 * (x * y + y * z + z * x);
    }
};

struct Node2 { Box box; int left = -1, right = -1, first = 0, count = 0, r_first = 0, r_count = 0; };   // count > 0: leaf; r_*: the subtree's range of order[]

struct Builder2 {
    const std::vector<Box>& pbox;
    const std::vector<std::array<float, 3>>& pcen;
    const WideBVHParams& prm;
    std::vector<int> order;
    std::vector<Node2> nodes;

    Builder2(const std::vector<Box>& b, const std::vector<std::array<float, 3>>& c, const WideBVHParams& p) : pbox(b), pcen(c), prm(p) {
        order.resize(b.size());
        for (size_t i = 0; i < b.size(); i++) order[i] = (int)i;
        nodes.reserve(2 * b.size());
    }

    void build() {
        struct Job { int node, first, count; };
        std::vector<Job> jobs;
        nodes.emplace_back();
        jobs.push_back({0, 0, (int)order.size()});
        const int nb = std::max(4, std::min(prm.bins, 64));
        std::vector<Box> bin_box((size_t)3 * nb), right_acc((size_t)nb);
        std::vector<int> bin_cnt((size_t)3 * nb);
        while (!jobs.empty()) {
            const Job j = jobs.back(); jobs.pop_back();
            Box bounds, cb;
            for (int i = j.first; i < j.first + j.count; i++) { bounds.grow(pbox[order[i]]); cb.grow(pcen[order[i]].data()); }
            nodes[j.node].box = bounds; nodes[j.node].r_first = j.first; nodes[j.node].r_count = j.count;
            auto make_leaf = [&] { nodes[j.node].first = j.first; nodes[j.node].count = j.count; };
            if (j.count == 1) { make_leaf(); continue; }
            // binned SAH over the three axes in one pass
            for (auto& b : bin_box) b = Box();
            std::fill(bin_cnt.begin(), bin_cnt.end(), 0);
            float k1[3], k0[3];
            bool usable[3];
            for (int a = 0; a < 3; a++) {
                const float ext = cb.hi[a] - cb.lo[a];
                usable[a] = ext > 0.0f && std::isfinite(ext);
                k0[a] = cb.lo[a]; k1[a] = usable[a] ? (float)nb * (1.0f - 1e-6f) / ext : 0.0f;
            }
            for (int i = j.first; i < j.first + j.count; i++) {
                const int p = order[i];
                for (int a = 0; a < 3; a++) {
                    if (!usable[a]) continue;
                    const int b = std::min(nb - 1, std::max(0, (int)((pcen[p][a] - k0[a]) * k1[a])));
                    bin_box[(size_t)a * nb + b].grow(pbox[p]); bin_cnt[(size_t)a * nb + b]++;
                }
            }
            double best = std::numeric_limits<double>::infinity();
            int best_axis = -1, best_split = -1;
            for (int a = 0; a < 3; a++) {
                if (!usable[a]) continue;
                Box acc; int cnt = 0;
                std::vector<int> rc((size_t)nb);
                for (int b = nb - 1; b > 0; b--) { acc.grow(bin_box[(size_t)a * nb + b]); cnt += bin_cnt[(size_t)a * nb + b]; right_acc[b] = acc; rc[b] = cnt; }
                Box l; int lc = 0;
                for (int b = 0; b + 1 < nb; b++) {
                    l.grow(bin_box[(size_t)a * nb + b]); lc += bin_cnt[(size_t)a * nb + b];
                    if (lc == 0 || rc[b + 1] == 0) continue;
                    const double c = l.area() * lc + right_acc[b + 1].area() * rc[b + 1];
                    if (c < best) { best = c; best_axis = a; best_split = b + 1; }
                }
            }
            const double area = std::max(bounds.area(), 1e-300);
            if (j.count <= prm.max_leaf) {
                const double leaf_cost = (double)prm.c_tri * j.count;
                const double split_cost = best_axis < 0 ? std::numeric_limits<double>::infinity() : (double)prm.c_trav + (double)prm.c_tri * best / area;
                if (leaf_cost <= split_cost) { make_leaf(); continue; }
            }
            int mid;
            if (best_axis < 0) {                                  // all centroids coincide: split the index range
                mid = j.first + j.count / 2;
            } else {
                const int a = best_axis;
                int* lo = order.data() + j.first; int* hi = lo + j.count;
                int* m = std::partition(lo, hi, [&](int p) { return std::min(nb - 1, std::max(0, (int)((pcen[p][a] - k0[a]) * k1[a]))) < best_split; });
                mid = (int)(m - order.data());
                if (mid == j.first || mid == j.first + j.count) mid = j.first + j.count / 2;      // cannot happen (both sides counted non-empty)
            }
            const int l = (int)nodes.size(); nodes.emplace_back(); nodes.emplace_back();
            nodes[j.node].left = l; nodes[j.node].right = l + 1;
            jobs.push_back({l + 1, mid, j.first + j.count - mid});
            jobs.push_back({l, j.first, mid - j.first});
        }
    }
};

uint32_t float_bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }

}  // namespace

void buildWideBVH(const std::vector<Primitive>& prims, const WideBVHParams& prm_in, WideBVH& out) {
    out.clear();
    WideBVHParams prm = prm_in;
    prm.max_leaf = std::max(1, std::min(prm.max_leaf, kWideMaxLeaf));
    const int n = (int)prims.size();
    if (n == 0) throw std::invalid_argument("fast tree: scene has no primitives");
    auto n_verts = [](const Primitive& p) { return p.type == PRIM_QUAD ? 4 : 3; };

    // triangle boxes, padded.  The node test computes a plane distance as fma(q, 2^e / d, (p - o) / d): its error is a few
    // 2^-24 of |p - o| + the node's extent, i.e. of |o| + the largest coordinate X of the NODE the child sits in.  With S the
    // scale of the scene - its largest coordinate, unless a few far-away primitives stick out (below) - origins are good up
    // to |coordinate| <= 4 S (the certified walk sends any other ray through the reference's walk), so a node with X <= S is
    // off by less than 5 S * 4 * 2^-24 = 2^-19.7 S; every primitive's box is widened by 2^-16 max(S, its own largest
    // coordinate) (cbox: 9e-5, a two-hundredth of the 1 M-triangle scene's cell) plus the reference's own 1e-6
    // (bvh.h:108-114), and a child box stored in a node that reaches beyond S by a further 2^-16 (X - S): an order of magnitude
    // of slack, so that no triangle the ray hits is ever culled.
    // S: round 3 took the largest coordinate of the scene, so ONE primitive a million units away widened every box of a
    // ten-unit scene by 15 units and the 8-wide tree decided nothing any more (the frames stayed right: everything was tested).
    // Now S = min(largest coordinate, 8 x the 99th percentile of the primitives' largest coordinates): the same value for
    // every scene whose primitives lie within 8 x the bulk's range, and the bulk's scale when a few lie far outside.
    std::vector<Box> pbox((size_t)n);
    std::vector<std::array<float, 3>> pcen((size_t)n);
    std::vector<float> pmag((size_t)n, 0.0f);
    float big = 0.0f;
    for (int i = 0; i < n; i++) {
        const Primitive& p = prims[i];
        for (int k = 0; k < n_verts(p); k++) pmag[i] = std::max(pmag[i], std::max(std::fabs(p.v[k].x), std::max(std::fabs(p.v[k].y), std::fabs(p.v[k].z))));
        big = std::max(big, pmag[i]);
    }
    if (!(big < 1.0e9f)) throw std::invalid_argument("fast tree: coordinates must stay below 1e9");
    float scale = big;
    {
        std::vector<float> sorted(pmag);
        const size_t k99 = (size_t)(0.99 * (double)(n - 1));
        std::nth_element(sorted.begin(), sorted.begin() + (std::ptrdiff_t)k99, sorted.end());
        if (sorted[k99] > 0.0f) scale = std::min(big, 8.0f * sorted[k99]);
    }
    out.origin_guard = 4.0f * scale;
    out.scale = scale;
    for (int i = 0; i < n; i++) {
        const Primitive& p = prims[i];
        const float pad = std::max(scale, pmag[i]) * (1.0f / 65536.0f) + 1e-6f;
        Box b;
        for (int k = 0; k < n_verts(p); k++) { const float v[3] = {p.v[k].x, p.v[k].y, p.v[k].z}; b.grow(v); }
        for (int a = 0; a < 3; a++) { pcen[i][a] = 0.5f * (b.lo[a] + b.hi[a]); b.lo[a] -= pad; b.hi[a] += pad; }
        pbox[i] = b;
    }
    Builder2 b2(pbox, pcen, prm);
    b2.build();
    const std::vector<Node2>& N = b2.nodes;
    out.binary_nodes = (int)N.size();
    for (const Node2& x : N) if (x.count > 0) out.binary_leaves++;

    // ---- collapse by dynamic programming over the binary tree (the published scheme of Ylitie, Karras, Laine 2017, sec. 3.1) ----
    // cost(n, i) = least SAH cost of standing the subtree of binary node n on at most i slots of a wide node:
    //   as ONE leaf child (all its triangles, if there are at most max_leaf):  area(n) * triangles(n) * c_tri
    //   as ONE inner child (a wide node of its own):                            area(n) * c_trav + distribute(n, 8)
    //   distribute(n, j) = min over k of cost(left, k) + cost(right, j - k);    cost(n, i) = min(distribute(n, i), cost(n, i - 1))
    // The greedy alternative (always open the largest child) left 35 % of the 1 M-triangle scene's nodes with two children.
    const int nb2 = (int)N.size();
    const double INF = std::numeric_limits<double>::infinity();
    struct Dp { double c[8]; uint8_t split[9]; uint8_t leaf1; };        // c[i], i = 1..7; split[j], j = 2..8: slots given to the left child
    std::vector<Dp> dp((size_t)nb2);
    for (int x = nb2 - 1; x >= 0; x--) {
        Dp& d = dp[(size_t)x];
        const double area = N[x].box.area();
        const int tris = N[x].r_count;
        const double as_leaf = tris <= prm.max_leaf ? area * tris * (double)prm.c_tri : INF;
        if (N[x].count > 0) {                                            // binary leaf
            for (int i = 1; i <= 7; i++) d.c[i] = as_leaf;
            d.leaf1 = 1;
            continue;
        }
        const Dp &l = dp[(size_t)N[x].left], &r = dp[(size_t)N[x].right];
        double dist[9];
        for (int j = 2; j <= 8; j++) {
            dist[j] = INF; d.split[j] = 1;
            for (int k = 1; k < j; k++) {
                const double c = l.c[std::min(k, 7)] + r.c[std::min(j - k, 7)];
                if (c < dist[j]) { dist[j] = c; d.split[j] = (uint8_t)k; }
            }
        }
        const double as_inner = area * (double)prm.c_trav + dist[8];
        d.leaf1 = as_leaf <= as_inner ? 1 : 0;
        d.c[1] = std::min(as_leaf, as_inner);
        for (int i = 2; i <= 7; i++) d.c[i] = std::min(dist[i], d.c[i - 1]);
    }
    struct WNode { int bin; int level; };
    std::vector<WNode> W;
    W.reserve(N.size() / 4 + 2);
    W.push_back({0, 1});
    out.level_start.push_back(0);
    out.tri_load_index.reserve((size_t)n);
    std::vector<uint32_t>& nodes = out.nodes;
    const double root_area = std::max(N[0].box.area(), 1e-300);
    std::vector<std::pair<int, int>> todo;
    for (size_t wi = 0; wi < W.size(); wi++) {
        if (W[wi].level > (int)out.level_start.size()) out.level_start.push_back((int)wi);
        const WNode w = W[wi];
        int set[8], ns = 0;
        bool set_leaf[8];
        if (N[w.bin].count > 0) { set[ns] = w.bin; set_leaf[ns++] = true; }       // a scene of <= max_leaf triangles: the root's only child
        else {
            // the children of this wide node: walk the recorded decisions from distribute(bin, 8) down
            todo.clear();
            const int k = dp[(size_t)w.bin].split[8];
            todo.push_back({N[w.bin].right, 8 - k}); todo.push_back({N[w.bin].left, k});
            while (!todo.empty()) {
                const int x = todo.back().first; int i = std::min(todo.back().second, 7); todo.pop_back();
                const Dp& d = dp[(size_t)x];
                if (N[x].count == 0) while (i > 1 && d.c[i] == d.c[i - 1]) i--;      // cost(n, i) came from fewer slots
                if (i == 1 || N[x].count > 0) { set[ns] = x; set_leaf[ns++] = N[x].count > 0 || d.leaf1; continue; }
                const int kk = d.split[i];
                todo.push_back({N[x].right, i - kk}); todo.push_back({N[x].left, kk});
            }
        }
        // ---- slots: child c goes to the slot whose corner direction its centre lies towards (greedy on the best pairs) ----
        // the children's boxes as this node stores them: widened once more where the node reaches beyond the scene's scale
        Box cbox[8], nb;
        {
            Box raw;
            for (int k = 0; k < ns; k++) raw.grow(N[set[k]].box);
            float reach = 0.0f;
            for (int a = 0; a < 3; a++) reach = std::max(reach, std::max(std::fabs(raw.lo[a]), std::fabs(raw.hi[a])));
            const float extra = reach > scale ? (reach - scale) * (1.0f / 65536.0f) : 0.0f;
            for (int k = 0; k < ns; k++) {
                cbox[k] = N[set[k]].box;
                for (int a = 0; a < 3; a++) { cbox[k].lo[a] -= extra; cbox[k].hi[a] += extra; }
                nb.grow(cbox[k]);
            }
        }
        int slot_of[8]; bool slot_used[8] = {}, placed[8] = {};
        for (int round = 0; round < ns; round++) {
            double bestc = -std::numeric_limits<double>::infinity(); int bc = -1, bs = -1;
            for (int k = 0; k < ns; k++) {
                if (placed[k]) continue;
                const Box& cbx = N[set[k]].box;
                for (int s = 0; s < 8; s++) {
                    if (slot_used[s]) continue;
                    double c = 0.0;
                    for (int a = 0; a < 3; a++) {
                        const double rel = 0.5 * ((double)cbx.lo[a] + cbx.hi[a]) - 0.5 * ((double)nb.lo[a] + nb.hi[a]);
                        c += ((s >> a) & 1) ? rel : -rel;
                    }
                    if (c > bestc) { bestc = c; bc = k; bs = s; }
                }
            }
            placed[bc] = true; slot_used[bs] = true; slot_of[bc] = bs;
        }
        int child_in_slot[8], k_in_slot[8]; bool leaf_in_slot[8];
        for (int s = 0; s < 8; s++) { child_in_slot[s] = -1; k_in_slot[s] = -1; leaf_in_slot[s] = false; }
        for (int k = 0; k < ns; k++) { child_in_slot[slot_of[k]] = set[k]; k_in_slot[slot_of[k]] = k; leaf_in_slot[slot_of[k]] = set_leaf[k]; }

        // ---- record ----
        const size_t base = nodes.size();
        nodes.resize(base + kWideNodeDwords, 0u);
        uint32_t* rec = nodes.data() + base;
        uint32_t ebias[3];
        double step[3];
        for (int a = 0; a < 3; a++) {
            const double ext = (double)nb.hi[a] - (double)nb.lo[a];
            int e = ext > 0.0 ? (int)std::ceil(std::log2(ext / 255.0)) : -126;
            while (std::ldexp(255.0, e) < ext) e++;                 // 255 steps must reach the far side
            e = std::max(-126, std::min(e, 100));
            ebias[a] = (uint32_t)(e + 127); step[a] = std::ldexp(1.0, e);
            rec[a] = float_bits(nb.lo[a]);
        }
        uint32_t imask = 0;
        const uint32_t child_base = (uint32_t)W.size(), tri_base = (uint32_t)out.tri_load_index.size();
        uint8_t q[6][8];
        for (int s = 0; s < 8; s++) { for (int pl = 0; pl < 3; pl++) { q[pl][s] = 255; q[3 + pl][s] = 0; } }
        for (int s = 0; s < 8; s++) {
            const int c = child_in_slot[s];
            if (c < 0) continue;
            const Box& cbx = cbox[k_in_slot[s]];
            for (int a = 0; a < 3; a++) {
                // outward rounding, checked in the arithmetic the kernel's planes stand for: p + q * 2^e as exact reals
                int lo = (int)std::floor(((double)cbx.lo[a] - (double)nb.lo[a]) / step[a]);
                int hi = (int)std::ceil(((double)cbx.hi[a] - (double)nb.lo[a]) / step[a]);
                lo = std::max(0, std::min(lo, 255)); hi = std::max(0, std::min(hi, 255));
                while (lo > 0 && (double)nb.lo[a] + lo * step[a] > (double)cbx.lo[a]) lo--;
                while (hi < 255 && (double)nb.lo[a] + hi * step[a] < (double)cbx.hi[a]) hi++;
                q[a][s] = (uint8_t)lo; q[3 + a][s] = (uint8_t)hi;
            }
            out.sah += cbx.area() / root_area;
            if (leaf_in_slot[s]) {
                const uint32_t off = (uint32_t)out.tri_load_index.size() - tri_base;
                for (int i = 0; i < N[c].r_count; i++) out.tri_load_index.push_back(b2.order[N[c].r_first + i]);
                rec[WN_WORD0 + s] = ((1u << N[c].r_count) - 1u) << off;
                out.leaf_hist[N[c].r_count]++;
            } else {
                imask |= 1u << s;
                rec[WN_WORD0 + s] = 1u << (24 + s);
                W.push_back({c, w.level + 1});
            }
        }
        rec[3] = ebias[0] | (ebias[1] << 8) | (ebias[2] << 16) | (imask << 24);
        rec[4] = child_base; rec[5] = tri_base;
        for (int pl = 0; pl < 6; pl++) {
            rec[WN_PLANES0 + 2 * pl] = (uint32_t)q[pl][0] | ((uint32_t)q[pl][1] << 8) | ((uint32_t)q[pl][2] << 16) | ((uint32_t)q[pl][3] << 24);
            rec[WN_PLANES0 + 2 * pl + 1] = (uint32_t)q[pl][4] | ((uint32_t)q[pl][5] << 8) | ((uint32_t)q[pl][6] << 16) | ((uint32_t)q[pl][7] << 24);
        }
        out.depth = std::max(out.depth, w.level);
        out.fill_hist[ns]++;
    }
    out.n_nodes = (int)W.size();
    out.level_start.push_back(out.n_nodes);
    if ((int)out.tri_load_index.size() != n) throw std::logic_error("fast tree: triangle count mismatch");
}

namespace {
// One half of Quad::intersect (quad.h:56-87 / 90-121) without its upper bound: t of the half (v0; v0 + edge1, v0 + edge2) if it
// is accepted, else +inf.  The inclusive forms of the quad (|a| > eps, u >= 0 && u <= 1, ...), t > eps && t >= t_min as t >= t_lo.
float quad_half_host(f3 v0, f3 edge1, f3 edge2, f3 o, f3 d, float t_lo) {
    const float inf = std::numeric_limits<float>::infinity();
    const f3 h = cross(d, edge2);
    const float a = dot(edge1, h);
    if (!(std::fabs(a) > 1e-8f)) return inf;
    const float f = 1.0f / a;
    const f3 s = o - v0;
    const float u = f * dot(s, h);
    if (!(u >= 0.0f && u <= 1.0f)) return inf;
    const f3 q = cross(s, edge1);
    const float v = f * dot(d, q);
    if (!(v >= 0.0f && u + v <= 1.0f)) return inf;
    const float t = f * dot(edge2, q);
    return t >= t_lo ? t : inf;
}
// Triangle::intersect (triangle.h:64-96) with precomputed edges, accept test as pt_device.h: mt_accept states it; a quad: the
// smaller t of its two halves (Quad::intersect under an upper bound c returns exactly that whenever it is below c)
bool mt_host(const Primitive& p, f3 o, f3 d, float t_lo, float& t) {
    if (p.type == PRIM_QUAD) {
        const f3 e1 = p.v[1] - p.v[0], e2 = p.v[2] - p.v[0], e3 = p.v[3] - p.v[0];
        t = std::fmin(quad_half_host(p.v[0], e1, e2, o, d, t_lo), quad_half_host(p.v[0], e2, e3, o, d, t_lo));
        return t < std::numeric_limits<float>::infinity();
    }
    const f3 edge1 = p.v[1] - p.v[0], edge2 = p.v[2] - p.v[0];
    const f3 h = cross(d, edge2);
    const float a = dot(edge1, h);
    if (std::fabs(a) < 1e-8f) return false;
    const float f = 1.0f / a;
    const f3 s = o - p.v[0];
    const float u = f * dot(s, h);
    if (u < 0.0f || u > 1.0f) return false;
    const f3 q = cross(s, edge1);
    const float v = f * dot(d, q);
    if (v < 0.0f || u + v > 1.0f) return false;
    t = f * dot(edge2, q);
    return t >= t_lo;
}
}  // namespace

int wideIntersectHost(const WideBVH& bvh, const std::vector<Primitive>& prims, const std::vector<int>& ref_slot, f3 o, f3 d,
                      float t_min, float t_max, float& t_hit, WideWalkCounters* cn) {
    const f3 inv = mk3(wide_inv(d.x), wide_inv(d.y), wide_inv(d.z));
    const uint32_t octinv = wide_octinv(inv);
    const float eps_up = wb_as_float(float_bits(1e-8f) + 1u);
    const float t_lo = t_min > 1e-8f ? t_min : eps_up;
    float closest = t_max;
    int hit = -1;
    struct Group { uint32_t base, bits; };
    std::vector<Group> stack;
    Group g{0u, (1u << 8) | (1u << octinv)};                    // the root: slot 0 of a virtual parent with imask 1
    const uint4* recs = reinterpret_cast<const uint4*>(bvh.nodes.data());
    while (true) {
        if ((g.bits & 0xffu) == 0u) {
            if (stack.empty()) break;
            g = stack.back(); stack.pop_back();
        }
        const uint32_t hits = g.bits & 0xffu;
        const int bit = 31 - __builtin_clz(hits);
        g.bits ^= 1u << bit;
        const uint32_t slot = (uint32_t)bit ^ octinv, imask = g.bits >> 8;
        const uint32_t ni = g.base + (uint32_t)__builtin_popcount(imask & ((1u << slot) - 1u));
        if (g.bits & 0xffu) stack.push_back(g);
        if (cn) { cn->node_visits++; cn->max_stack = std::max<uint64_t>(cn->max_stack, stack.size()); }
        const uint4* q = recs + (size_t)ni * (kWideNodeDwords / 4);
        const WideStep st = wide_node_test(q[0], q[1], q[2], q[3], q[4], q[5], q[6], o, inv, octinv, t_min, closest);
        uint32_t tris = st.tris;
        while (tris) {
            const int k = (int)st.tri_base + __builtin_ctz(tris);
            tris &= tris - 1u;
            if (cn) cn->prim_tests++;
            const int li = bvh.tri_load_index[(size_t)k];
            float t;
            if (!mt_host(prims[(size_t)li], o, d, t_lo, t)) continue;
            if (t < closest || (t == closest && hit >= 0 && ref_slot[(size_t)li] < ref_slot[(size_t)hit])) { closest = t; hit = li; }
        }
        g = Group{st.child_base, (st.imask << 8) | st.inner};
    }
    t_hit = closest;
    return hit;
}

}  // namespace ptmi
