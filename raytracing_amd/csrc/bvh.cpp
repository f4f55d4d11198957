// bvh.cpp -- binned-SAH binary build (parallel over subtrees) collapsed to 128-byte 4-wide nodes.
// See bvh.h for why the tree's shape is free (reference call site: src/scene/object.rs:156).
#include "bvh.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <thread>

namespace trr {
namespace {

constexpr int MAX_BINS = 64;
int NBINS = 32;              // (tunable for experiments: TRR_BVH_BINS = 4..64)
uint32_t SWEEP_MAX = 0;      // ranges of at most this many triangles are split by an exact SAH sweep (TRR_BVH_SWEEP)
uint32_t MAX_LEAF = 4;       // (tunable for experiments: TRR_BVH_MAX_LEAF = 1..4)
uint32_t ALWAYS_LEAF = 2;    // ranges of at most this many triangles are never split (TRR_BVH_ALWAYS_LEAF)
float LEAF_BIAS = 0.5f;      // leaf preferred while leaf_cost <= split_cost + LEAF_BIAS * area (TRR_BVH_LEAF_BIAS); round 3: 1.5 -> 0.5
                             // and 16 -> 32 bins lower the wavefront steps per ray by 3 % (scripts/experiments/tree_quality.py)
uint32_t MEDIAN_BELOW = 0;   // experiment (TRR_BVH_MEDIAN_BELOW): ranges of at most this many triangles are split at the spatial
                             // middle of their widest axis, as a Morton-order build would, instead of by SAH
constexpr uint32_t PAR_MIN = 1u << 15;   // subtrees above this size may get their own thread

struct BNode {
    Box3 box;
    uint32_t left, right;    // children (valid when count == 0)
    uint32_t first, count;   // leaf range in idx (count > 0)
    uint32_t range_first, range_count;   // the range of idx below the node, leaf or not (the collapse may close a small subtree into a leaf)
};

inline Box3 empty_box() {
    const float inf = std::numeric_limits<float>::infinity();
    return Box3{{inf, inf, inf}, {-inf, -inf, -inf}};
}
inline void grow(Box3& a, const Box3& b) {
    for (int k = 0; k < 3; ++k) {
        a.lo[k] = std::fmin(a.lo[k], b.lo[k]);
        a.hi[k] = std::fmax(a.hi[k], b.hi[k]);
    }
}
inline float half_area(const Box3& b) {
    float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    if (!(dx >= 0.0f) || !(dy >= 0.0f) || !(dz >= 0.0f)) return 0.0f;
    float a = dx * dy + dy * dz + dz * dx;
    return std::isfinite(a) ? a : std::numeric_limits<float>::max();
}

struct Builder {
    const Box3* boxes;
    uint32_t n;
    std::vector<float> cent[3];
    std::vector<uint32_t> idx;
    std::vector<BNode> nodes;
    std::atomic<uint32_t> n_nodes{0};
    int par_depth = 0;

    uint32_t alloc2() { return n_nodes.fetch_add(2); }

    void make_leaf(BNode& nd, uint32_t lo, uint32_t hi) {
        nd.first = lo;
        nd.count = hi - lo;
        nd.left = nd.right = 0;
    }

    void build(uint32_t me, uint32_t lo, uint32_t hi, int depth) {
        BNode& nd = nodes[me];
        Box3 box = empty_box();
        float cmin[3] = {INFINITY, INFINITY, INFINITY}, cmax[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (uint32_t i = lo; i < hi; ++i) {
            uint32_t t = idx[i];
            grow(box, boxes[t]);
            for (int k = 0; k < 3; ++k) {
                float c = cent[k][t];
                cmin[k] = std::fmin(cmin[k], c);
                cmax[k] = std::fmax(cmax[k], c);
            }
        }
        nd.box = box;
        uint32_t cnt = hi - lo;
        nd.range_first = lo;
        nd.range_count = cnt;
        if (cnt <= ALWAYS_LEAF) { make_leaf(nd, lo, hi); return; }

        // exact SAH sweep for small ranges: every split position of the centroid order on every axis
        if (cnt <= SWEEP_MAX && depth < 56) {
            float sw_cost = INFINITY;
            int sw_axis = -1;
            uint32_t sw_pos = 0;
            std::vector<uint32_t> order(idx.begin() + lo, idx.begin() + hi);
            std::vector<float> rarea(cnt);
            for (int ax = 0; ax < 3; ++ax) {
                const std::vector<float>& ca = cent[ax];
                std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return ca[a] < ca[b] || (ca[a] == ca[b] && a < b); });
                Box3 acc = empty_box();
                for (uint32_t i = cnt; i-- > 1;) { grow(acc, boxes[order[i]]); rarea[i] = half_area(acc); }
                acc = empty_box();
                for (uint32_t i = 1; i < cnt; ++i) {
                    grow(acc, boxes[order[i - 1]]);
                    float cost = half_area(acc) * (float)i + rarea[i] * (float)(cnt - i);
                    if (cost < sw_cost) { sw_cost = cost; sw_axis = ax; sw_pos = i; }
                }
            }
            if (sw_axis >= 0) {
                if (cnt <= MAX_LEAF) {
                    float leaf_cost = half_area(box) * (float)cnt;
                    if (leaf_cost <= sw_cost + LEAF_BIAS * half_area(box)) { make_leaf(nd, lo, hi); return; }
                }
                const std::vector<float>& ca = cent[sw_axis];
                std::sort(idx.begin() + lo, idx.begin() + hi, [&](uint32_t a, uint32_t b) { return ca[a] < ca[b] || (ca[a] == ca[b] && a < b); });
                uint32_t mid = lo + sw_pos;
                uint32_t kids = alloc2();
                nd.count = 0; nd.first = 0; nd.left = kids; nd.right = kids + 1;
                build(kids, lo, mid, depth + 1);
                build(kids + 1, mid, hi, depth + 1);
                return;
            }
        }

        // binned SAH over the three axes
        float best_cost = INFINITY;
        int best_axis = -1, best_bin = -1;
        float best_scale = 0.0f;
        if (depth < 56) {
            for (int ax = 0; ax < 3; ++ax) {
                float ext = cmax[ax] - cmin[ax];
                if (!(ext > 0.0f) || !std::isfinite(ext)) continue;
                float scale = (float)NBINS / ext;
                Box3 bb[MAX_BINS];
                uint32_t bc[MAX_BINS];
                for (int b = 0; b < NBINS; ++b) { bb[b] = empty_box(); bc[b] = 0; }
                for (uint32_t i = lo; i < hi; ++i) {
                    uint32_t t = idx[i];
                    int b = (int)((cent[ax][t] - cmin[ax]) * scale);
                    b = b < 0 ? 0 : (b >= NBINS ? NBINS - 1 : b);
                    bc[b]++;
                    grow(bb[b], boxes[t]);
                }
                float right_area[MAX_BINS];
                uint32_t right_cnt[MAX_BINS];
                Box3 acc = empty_box();
                uint32_t c = 0;
                for (int b = NBINS - 1; b > 0; --b) {
                    if (bc[b]) grow(acc, bb[b]);
                    c += bc[b];
                    right_area[b] = half_area(acc);
                    right_cnt[b] = c;
                }
                acc = empty_box();
                c = 0;
                for (int b = 0; b < NBINS - 1; ++b) {
                    if (bc[b]) grow(acc, bb[b]);
                    c += bc[b];
                    if (c == 0 || right_cnt[b + 1] == 0) continue;
                    float cost = half_area(acc) * (float)c + right_area[b + 1] * (float)right_cnt[b + 1];
                    if (cost < best_cost) { best_cost = cost; best_axis = ax; best_bin = b; best_scale = scale; }
                }
            }
        }
        if (cnt <= MAX_LEAF) {
            float leaf_cost = half_area(box) * (float)cnt;
            if (best_axis < 0 || leaf_cost <= best_cost + LEAF_BIAS * half_area(box)) { make_leaf(nd, lo, hi); return; }
        }
        uint32_t mid = lo;
        if (cnt <= MEDIAN_BELOW) {
            int ax = 0;
            float e0 = cmax[0] - cmin[0], e1 = cmax[1] - cmin[1], e2 = cmax[2] - cmin[2];
            if (e1 > e0 && e1 >= e2) ax = 1; else if (e2 > e0 && e2 > e1) ax = 2;
            const float m = 0.5f * cmin[ax] + 0.5f * cmax[ax];
            const std::vector<float>& ca = cent[ax];
            auto it = std::partition(idx.begin() + lo, idx.begin() + hi, [&](uint32_t t) { return ca[t] < m; });
            mid = (uint32_t)(it - idx.begin());
        } else if (best_axis >= 0) {
            const float c0 = cmin[best_axis];
            const std::vector<float>& ca = cent[best_axis];
            auto it = std::partition(idx.begin() + lo, idx.begin() + hi, [&](uint32_t t) {
                int b = (int)((ca[t] - c0) * best_scale);
                b = b < 0 ? 0 : (b >= NBINS ? NBINS - 1 : b);
                return b <= best_bin;
            });
            mid = (uint32_t)(it - idx.begin());
        }
        if (mid == lo || mid == hi) {
            // all centroids coincide (or the tree got too deep): split by count on the widest axis
            int ax = 0;
            float e0 = cmax[0] - cmin[0], e1 = cmax[1] - cmin[1], e2 = cmax[2] - cmin[2];
            if (e1 > e0 && e1 >= e2) ax = 1; else if (e2 > e0 && e2 > e1) ax = 2;
            mid = lo + cnt / 2;
            const std::vector<float>& ca = cent[ax];
            std::nth_element(idx.begin() + lo, idx.begin() + mid, idx.begin() + hi,
                             [&](uint32_t a, uint32_t b) { return ca[a] < ca[b] || (ca[a] == ca[b] && a < b); });
        }
        uint32_t kids = alloc2();
        nd.count = 0;
        nd.first = 0;
        nd.left = kids;
        nd.right = kids + 1;
        if (depth < par_depth && cnt >= PAR_MIN) {
            std::thread th([this, kids, lo, mid, depth] { build(kids, lo, mid, depth + 1); });
            build(kids + 1, mid, hi, depth + 1);
            th.join();
        } else {
            build(kids, lo, mid, depth + 1);
            build(kids + 1, mid, hi, depth + 1);
        }
    }
};

// Which (grand)children of a binary node become the <= 4 children of its 4-wide node.  Greedy: open the child with the
// largest box until there are four.  Optimal (the default, after Ylitie et al. 2017, "Efficient Incoherent Ray
// Traversal on GPUs Through Compressed Wide BVHs", section 3.1): minimise the expected number of wavefront steps, with a
// subtree's box surface as the chance of entering it --
//   T(b)    = cost of b's subtree when b fills ONE child slot = A(b) * leaf_steps(count) for a leaf, A(b) * C_NODE + G(b, 4) else
//   G(b, k) = cheapest way to hand b's two children i + j <= k slots = min F(l, i) + F(r, j)
//   F(b, k) = min(T(b), G(b, k)), F(b, 1) = T(b)
// A leaf's cost counts the kernel's triangle steps (two triangles per step).
int COLLAPSE_DP = 1;         // TRR_BVH_COLLAPSE=greedy for the old rule
float C_TRI_STEP = 2.0f;     // cost of a triangle step relative to a node step (TRR_BVH_TRI_STEP; measured: 1.5-2 is flat, 1 is worse)

struct CollapseDP {
    const std::vector<BNode>& bn;
    uint32_t n_nodes;
    std::vector<float> T, G;   // G[3*b + (k-2)], k = 2..4
    std::vector<uint8_t> closed;   // the node ends up as a leaf reference (a leaf of the binary tree, or a subtree closed here)
    explicit CollapseDP(const std::vector<BNode>& b, uint32_t n) : bn(b), n_nodes(n), T(n), G((size_t)3 * n, 0.0f), closed(n, 0) {
        for (uint32_t b_ = n; b_-- > 0;) {   // children were allocated after their parents: descending order is bottom-up
            const BNode& nd = bn[b_];
            const float a = half_area(nd.box);
            if (nd.count > 0) { T[b_] = a * C_TRI_STEP * (float)((nd.count + 1) / 2); closed[b_] = 1; continue; }
            for (int k = 2; k <= 4; ++k) {
                float best = INFINITY;
                for (int i = 1; i < k; ++i) best = std::fmin(best, F(nd.left, i) + F(nd.right, k - i));
                G[(size_t)3 * b_ + (k - 2)] = best;
            }
            T[b_] = a + G[(size_t)3 * b_ + 2];
            if (nd.range_count <= MAX_LEAF) {   // a small subtree may be closed into one leaf
                const float as_leaf = a * C_TRI_STEP * (float)((nd.range_count + 1) / 2);
                if (as_leaf <= T[b_]) { T[b_] = as_leaf; closed[b_] = 1; }
            }
        }
    }
    bool is_leaf(uint32_t b) const { return closed[b] != 0; }
    float F(uint32_t b, int k) const {
        if (k == 1 || closed[b]) return T[b];
        return std::fmin(T[b], G[(size_t)3 * b + (k - 2)]);
    }
    // the slots b's subtree takes when it may use k of them
    void assign(uint32_t b, int k, uint32_t* kids, int* nk) const {
        if (k == 1 || closed[b] || T[b] <= G[(size_t)3 * b + (k - 2)]) { kids[(*nk)++] = b; return; }
        split(b, k, kids, nk);
    }
    void split(uint32_t b, int k, uint32_t* kids, int* nk) const {
        const BNode& nd = bn[b];
        int bi = 1;
        float best = INFINITY;
        for (int i = 1; i < k; ++i) {
            const float c = F(nd.left, i) + F(nd.right, k - i);
            if (c < best) { best = c; bi = i; }
        }
        assign(nd.left, bi, kids, nk);
        assign(nd.right, k - bi, kids, nk);
    }
};

struct Collapser {
    const std::vector<BNode>& bn;
    std::vector<Node4>& out;
    const CollapseDP* dp = nullptr;
    uint32_t max_depth = 0;
    uint32_t stack_need = 0;

    uint32_t ref_of(uint32_t b, uint32_t depth, uint32_t pushed) {
        const BNode& nd = bn[b];
        if (nd.count > 0) return make_leaf_ref(nd.first, nd.count);
        if (dp && dp->is_leaf(b)) return make_leaf_ref(nd.range_first, nd.range_count);
        return emit(b, depth, pushed);
    }

    // emits the 4-wide node whose children are (grand)children of binary node b; returns its index
    // pushed: entries the traversal may hold on its stack when it arrives at this node
    uint32_t emit(uint32_t b, uint32_t depth, uint32_t pushed) {
        if (depth > max_depth) max_depth = depth;
        uint32_t me = (uint32_t)out.size();
        out.emplace_back();
        uint32_t kids[4];
        int nk = 2;
        kids[0] = bn[b].left;
        kids[1] = bn[b].right;
        if (dp) {
            nk = 0;
            dp->split(b, 4, kids, &nk);
        }
        while (!dp && nk < 4) {
            int pick = -1;
            float pa = -1.0f;
            for (int k = 0; k < nk; ++k) {
                if (bn[kids[k]].count > 0) continue;
                float a = half_area(bn[kids[k]].box);
                if (a > pa) { pa = a; pick = k; }
            }
            if (pick < 0) break;
            uint32_t p = kids[pick];
            kids[pick] = bn[p].left;
            kids[nk++] = bn[p].right;
        }
        Node4 nd;
        std::memset(&nd, 0, sizeof nd);
        const float inf = std::numeric_limits<float>::infinity();
        for (int k = 0; k < 4; ++k) {
            if (k < nk) {
                const Box3& bx = bn[kids[k]].box;
                nd.lox[k] = bx.lo[0]; nd.loy[k] = bx.lo[1]; nd.loz[k] = bx.lo[2];
                nd.hix[k] = bx.hi[0]; nd.hiy[k] = bx.hi[1]; nd.hiz[k] = bx.hi[2];
            } else {
                // (+inf .. +inf): (inf - o) * inv is +inf or -inf on both planes of every axis, so tmax > max(tmin, 0) is
                // false for every ray and the kernel needs no separate "slot in use" test
                nd.lox[k] = nd.loy[k] = nd.loz[k] = inf;
                nd.hix[k] = nd.hiy[k] = nd.hiz[k] = inf;
                nd.child[k] = REF_EMPTY;
            }
        }
        pushed += (uint32_t)nk - 1u;   // the node pushes all but the child it descends into
        if (pushed > stack_need) stack_need = pushed;
        for (int k = 0; k < nk; ++k) nd.child[k] = ref_of(kids[k], depth + 1, pushed);
        out[me] = nd;
        return me;
    }
};

}   // namespace

void build_bvh4(const Box3* boxes, uint32_t n, uint32_t n_threads, BuildResult* res) {
    res->nodes.clear();
    res->order.clear();
    res->root_ref = REF_EMPTY;
    res->max_depth = 0;
    res->stack_need = 0;
    res->bounds = empty_box();
    if (n == 0) return;
    if (const char* e = std::getenv("TRR_BVH_MAX_LEAF")) MAX_LEAF = (uint32_t)std::min(4, std::max(1, std::atoi(e)));
    if (const char* e = std::getenv("TRR_BVH_ALWAYS_LEAF")) ALWAYS_LEAF = (uint32_t)std::min(4, std::max(1, std::atoi(e)));
    if (const char* e = std::getenv("TRR_BVH_LEAF_BIAS")) LEAF_BIAS = (float)std::atof(e);
    if (const char* e = std::getenv("TRR_BVH_BINS")) NBINS = std::min(MAX_BINS, std::max(4, std::atoi(e)));
    if (const char* e = std::getenv("TRR_BVH_MEDIAN_BELOW")) MEDIAN_BELOW = (uint32_t)std::max(0, std::atoi(e));
    if (const char* e = std::getenv("TRR_BVH_COLLAPSE")) COLLAPSE_DP = std::strcmp(e, "dp") == 0;
    if (const char* e = std::getenv("TRR_BVH_TRI_STEP")) C_TRI_STEP = (float)std::atof(e);
    if (const char* e = std::getenv("TRR_BVH_SWEEP")) SWEEP_MAX = (uint32_t)std::max(0, std::atoi(e));
    if (ALWAYS_LEAF > MAX_LEAF) ALWAYS_LEAF = MAX_LEAF;
    if (n_threads == 0) n_threads = std::max(1u, std::thread::hardware_concurrency());
    n_threads = std::min(n_threads, 64u);

    Builder b;
    b.boxes = boxes;
    b.n = n;
    for (int k = 0; k < 3; ++k) b.cent[k].resize(n);
    b.idx.resize(n);
    for (uint32_t i = 0; i < n; ++i) {
        b.idx[i] = i;
        for (int k = 0; k < 3; ++k) {
            float c = 0.5f * boxes[i].lo[k] + 0.5f * boxes[i].hi[k];
            b.cent[k][i] = std::isfinite(c) ? c : 0.0f;   // NaN / infinite vertices must not poison the binning
        }
    }
    b.nodes.resize((size_t)2 * n + 2);
    b.n_nodes = 1;
    b.par_depth = 0;
    while ((1u << b.par_depth) < n_threads) b.par_depth++;
    b.build(0, 0, n, 0);

    res->order = b.idx;
    res->bounds = b.nodes[0].box;
    if (b.nodes[0].count > 0) {
        res->root_ref = make_leaf_ref(b.nodes[0].first, b.nodes[0].count);
        return;
    }
    res->nodes.reserve((size_t)n / 2 + 16);
    Collapser c{b.nodes, res->nodes};
    std::unique_ptr<CollapseDP> dp;
    if (COLLAPSE_DP) {
        dp.reset(new CollapseDP(b.nodes, b.n_nodes.load()));
        c.dp = dp.get();
    }
    if (dp && dp->is_leaf(0)) {   // (n <= MAX_LEAF and one leaf is cheapest)
        res->root_ref = make_leaf_ref(b.nodes[0].range_first, b.nodes[0].range_count);
        return;
    }
    res->root_ref = c.emit(0, 1, 0);
    res->max_depth = c.max_depth;
    res->stack_need = c.stack_need;
}

}   // namespace trr
