// Runs the per-element bodies of the device BVH build (raytracing_amd/csrc/lbvh_core.h) serially on the CPU, in the
// order of the kernels of build_device.hip, and checks the resulting 4-wide tree: every sorted position in exactly one
// leaf, every child box containing every triangle box below it.  Built and run by tests/test_lbvh_core.py.
//   usage: lbvh_core_check <n> <seed> <mode> [treelet_max]   mode 0: uniform, 1: many identical centroids, 2: on a line,
//   3: clustered, 4: the triangles of a tessellated bumpy sphere; treelet_max >= 3: subtrees of at most that many triangles
//   are rebuilt by lbvh::treelet_rebuild (what k_treelet does; k_treelet_top, the wavefront form for big ranges, makes the
//   same kind of tree and is checked on the GPU by trr_scene_validate in tests/test_gpu_build_device.py)
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <random>
#include <vector>

#include "../raytracing_amd/csrc/lbvh_core.h"

using namespace trr;

static int fail(const char* what, long a = 0, long b = 0) {
    std::printf("FAIL %s %ld %ld\n", what, a, b);
    return 1;
}

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    const uint32_t n = (uint32_t)std::atol(argv[1]);
    const uint32_t seed = (uint32_t)std::atol(argv[2]);
    const int mode = std::atoi(argv[3]);
    const uint32_t treelet_max = argc > 4 ? (uint32_t)std::atol(argv[4]) : 0u;
    std::mt19937 rng(seed);
    std::uniform_real_distribution<float> U(-10.0f, 10.0f), S(0.0f, 0.3f);
    std::vector<Box3> boxes(n);
    for (uint32_t i = 0; i < n; ++i) {
        float c[3] = {U(rng), U(rng), U(rng)};
        if (mode == 1 && (i % 3)) { c[0] = 1.0f; c[1] = 2.0f; c[2] = 3.0f; }
        if (mode == 2) { c[1] = 0.0f; c[2] = 0.0f; }
        if (mode == 3) { float k = (float)(i % 7); c[0] = k + 0.001f * c[0]; c[1] = k + 0.001f * c[1]; c[2] = 0.001f * c[2]; }
        float h = (mode == 1 && (i % 3)) ? 0.25f : S(rng);
        for (int k = 0; k < 3; ++k) { boxes[i].lo[k] = c[k] - h; boxes[i].hi[k] = c[k] + h; }
    }
    if (mode == 4) {   // boxes of the triangles of a (u, v) grid over a bumpy sphere, two per cell
        const uint32_t nv = (uint32_t)std::max(2.0, std::sqrt(n / 4.0)), nu = std::max(3u, n / (2 * nv));
        auto P = [&](uint32_t iu, uint32_t iv, float p[3]) {
            const float u = 6.2831853f * (float)iu / (float)nu, v = 3.1415926f * (float)iv / (float)nv;
            const float r = 5.0f + 0.4f * std::sin(7.0f * u) * std::sin(5.0f * v);
            p[0] = r * std::sin(v) * std::cos(u); p[1] = r * std::cos(v); p[2] = r * std::sin(v) * std::sin(u);
        };
        for (uint32_t i = 0; i < n; ++i) {
            const uint32_t cell = i / 2, iu = cell % nu, iv = (cell / nu) % nv;
            float a[3], b[3], c[3];
            if (i & 1) { P(iu, iv, a); P(iu + 1, iv, b); P(iu + 1, iv + 1, c); } else { P(iu, iv, a); P(iu + 1, iv + 1, b); P(iu, iv + 1, c); }
            for (int k = 0; k < 3; ++k) { boxes[i].lo[k] = std::min({a[k], b[k], c[k]}); boxes[i].hi[k] = std::max({a[k], b[k], c[k]}) + 1e-4f; }
        }
    }
    // k_prep / k_keys
    float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
    std::vector<float> cent(3 * (size_t)n);
    for (uint32_t i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) {
            float c = 0.5f * boxes[i].lo[k] + 0.5f * boxes[i].hi[k];
            cent[3 * (size_t)i + k] = c;
            lo[k] = std::min(lo[k], c);
            hi[k] = std::max(hi[k], c);
        }
    float ext[3] = {hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]};
    std::vector<uint64_t> key_of(n);
    for (uint32_t i = 0; i < n; ++i) key_of[i] = lbvh::morton63(&cent[3 * (size_t)i], lo, ext);
    std::vector<uint32_t> val(n);
    std::iota(val.begin(), val.end(), 0u);
    std::stable_sort(val.begin(), val.end(), [&](uint32_t a, uint32_t b) { return key_of[a] < key_of[b]; });
    std::vector<uint64_t> keys(n);
    for (uint32_t s = 0; s < n; ++s) keys[s] = key_of[val[s]];
    for (uint32_t s = 0; s < n; ++s)
        if (keys[s] >> 63) return fail("key uses bit 63", s);
    if (n <= lbvh::MAX_LEAF) { std::printf("ok n=%u single leaf\n", n); return 0; }

    // k_gather: boxes in sorted order = level 0 of the pyramid; k_pyramid: the levels above
    lbvh::Pyramid pyr{};
    const uint32_t pyr_total = lbvh::pyramid_layout(n, pyr.off, &pyr.levels);
    std::vector<Box3> pyr_store(pyr_total);
    for (uint32_t s = 0; s < n; ++s) pyr_store[s] = boxes[val[s]];
    {
        uint32_t m = n;
        for (uint32_t k = 1; k < pyr.levels; ++k) {
            uint32_t m2 = (m + 1) >> 1;
            for (uint32_t j = 0; j < m2; ++j) pyr_store[pyr.off[k] + j] = lbvh::pyramid_pair(&pyr_store[pyr.off[k - 1]], m, j);
            m = m2;
        }
        if (m != 1) return fail("pyramid does not end in one entry", m);
    }
    pyr.base = pyr_store.data();

    // k_hier
    std::vector<uint32_t> child_l(n - 1), child_r(n - 1), first(n - 1), last(n - 1);
    std::vector<Box3> ibox(n - 1);
    lbvh::Tree t{child_l.data(), child_r.data(), first.data(), last.data(), ibox.data(), pyr_store.data(), lbvh::MAX_LEAF};
    for (uint32_t i = 0; i + 1 < n; ++i) lbvh::build_inner(keys.data(), n, i, t);
    if (first[0] != 0 || last[0] != n - 1) return fail("root range", first[0], last[0]);
    size_t n_treelets = 0;
    if (treelet_max >= 3) {
        // k_hier's second half (which children root a treelet), then k_treelet on every root
        std::vector<uint32_t> roots, stack(n, 0);
        for (uint32_t i = 0; i + 1 < n; ++i) {
            const uint32_t f = first[i], l = last[i], cl = child_l[i], cr = child_r[i];
            if (l - f + 1 > treelet_max) {
                if (!(cl & lbvh::LEAF_BIT)) {
                    uint32_t c = cl - f + 1;
                    if (c != last[cl] - first[cl] + 1) return fail("left count", i);
                    if (c >= 3 && c <= treelet_max) roots.push_back(cl);
                }
                if (!(cr & lbvh::LEAF_BIT)) {
                    uint32_t c = l - cr + 1;
                    if (c != last[cr] - first[cr] + 1) return fail("right count", i);
                    if (c >= 3 && c <= treelet_max) roots.push_back(cr);
                }
            } else if (i == 0 && l - f + 1 >= 3) roots.push_back(0);
        }
        std::vector<uint32_t> f0(first), l0(last);
        for (uint32_t r : roots) lbvh::treelet_rebuild(t, r, val.data(), pyr_store.data(), stack.data());
        for (uint32_t r : roots)
            if (first[r] != f0[r] || last[r] != l0[r]) return fail("treelet root range changed", r);
        for (uint32_t s = 0; s < n; ++s)
            if (std::memcmp(&pyr_store[s], &boxes[val[s]], sizeof(Box3)) != 0) return fail("boxes and ids permuted apart", s);
        n_treelets = roots.size();
        // the levels of the pyramid above the (permuted) sorted boxes
        uint32_t m = n;
        for (uint32_t k = 1; k < pyr.levels; ++k) {
            uint32_t m2 = (m + 1) >> 1;
            for (uint32_t j = 0; j < m2; ++j) pyr_store[pyr.off[k] + j] = lbvh::pyramid_pair(&pyr_store[pyr.off[k - 1]], m, j);
            m = m2;
        }
    }
    std::vector<uint32_t> parents(n - 1, 0), leaf_parents(n, 0);
    for (uint32_t i = 0; i + 1 < n; ++i) {
        // children partition the range
        uint32_t l = child_l[i], r = child_r[i];
        uint32_t lf = (l & lbvh::LEAF_BIT) ? (l & ~lbvh::LEAF_BIT) : first[l], ll = (l & lbvh::LEAF_BIT) ? (l & ~lbvh::LEAF_BIT) : last[l];
        uint32_t rf = (r & lbvh::LEAF_BIT) ? (r & ~lbvh::LEAF_BIT) : first[r], rl = (r & lbvh::LEAF_BIT) ? (r & ~lbvh::LEAF_BIT) : last[r];
        if (lf != first[i] || rl != last[i] || ll + 1 != rf) return fail("children do not partition the range", i);
        for (uint32_t c : {l, r}) {
            if (c & lbvh::LEAF_BIT) leaf_parents[c & ~lbvh::LEAF_BIT]++;
            else parents[c]++;
        }
    }
    if (parents[0] != 0) return fail("root has a parent");
    for (uint32_t i = 1; i + 1 < n; ++i)
        if (parents[i] != 1) return fail("inner node without exactly one parent", i, parents[i]);
    for (uint32_t s = 0; s < n; ++s)
        if (leaf_parents[s] != 1) return fail("leaf without exactly one parent", s, leaf_parents[s]);
    // k_fit: range unions; must equal the exact union of the range, computed here the slow way for a sample of nodes
    for (uint32_t i = 0; i + 1 < n; ++i) ibox[i] = lbvh::range_box(pyr, first[i], last[i]);
    for (uint32_t i = 0; i + 1 < n; i += (n > 5000 ? 97 : 1)) {
        if (last[i] - first[i] > 20000 && i != 0) continue;
        Box3 e = pyr_store[first[i]];
        for (uint32_t s = first[i] + 1; s <= last[i]; ++s) e = lbvh::box_union(e, pyr_store[s]);
        if (std::memcmp(&e, &ibox[i], sizeof e) != 0) return fail("range union differs from the exact union", i);
    }
    double cost_inner = 0.0;   // surface-area cost of the binary tree's inner nodes, in root areas (what the treelets lower)
    for (uint32_t i = 0; i + 1 < n; ++i) cost_inner += lbvh::box_half_area(ibox[i]);
    cost_inner /= lbvh::box_half_area(ibox[0]);
    // k_collapse, level by level
    uint32_t n_big = 0;
    for (uint32_t i = 0; i + 1 < n; ++i) n_big += (last[i] - first[i] + 1 > lbvh::MAX_LEAF);
    std::vector<Node4> nodes(n_big);
    std::vector<std::pair<uint32_t, uint32_t>> q{{0u, 0u}}, qn;
    uint32_t n_nodes = 1, depth = 0;
    const uint32_t slot_base = 1000, node_base = 77;
    while (!q.empty()) {
        ++depth;
        qn.clear();
        for (auto item : q) {
            uint32_t kids[4];
            int nk = lbvh::collapse_children(t, item.first, kids);
            Node4 nd;
            std::memset(&nd, 0, sizeof nd);
            for (int k = 0; k < 4; ++k) {
                if (k >= nk) { lbvh::node4_set_empty(nd, k); continue; }
                lbvh::node4_set_box(nd, k, lbvh::ref_box(t, kids[k]));
                if (lbvh::leaf_like(t, kids[k])) nd.child[k] = lbvh::leaf_ref_of(t, kids[k], slot_base);
                else {
                    if (n_nodes >= n_big) return fail("more nodes than n_big", n_nodes, n_big);
                    qn.push_back({kids[k], n_nodes});
                    nd.child[k] = node_base + n_nodes++;
                }
            }
            nodes[item.second] = nd;
        }
        q.swap(qn);
    }
    // the 4-wide tree
    std::vector<uint8_t> seen(n, 0);
    struct It { uint32_t ref; Box3 box; };
    std::vector<It> st;
    st.push_back({node_base + 0u, Box3{{-1e30f, -1e30f, -1e30f}, {1e30f, 1e30f, 1e30f}}});
    size_t leaves = 0, tri_in_leaves = 0;
    while (!st.empty()) {
        It it = st.back();
        st.pop_back();
        if (it.ref & REF_LEAF) {
            uint32_t f = ((it.ref & 0x7FFFFFFFu) >> 2) - slot_base, c = (it.ref & 3u) + 1;
            ++leaves;
            tri_in_leaves += c;
            for (uint32_t s = f; s < f + c; ++s) {
                if (s >= n) return fail("slot out of range", s);
                if (seen[s]++) return fail("slot in two leaves", s);
                const Box3& b = boxes[val[s]];
                for (int k = 0; k < 3; ++k)
                    if (b.lo[k] < it.box.lo[k] || b.hi[k] > it.box.hi[k]) return fail("leaf box too small", s);
            }
            continue;
        }
        uint32_t idx = it.ref - node_base;
        if (idx >= n_nodes) return fail("node out of range", idx);
        const Node4& nd = nodes[idx];
        int used = 0;
        for (int k = 0; k < 4; ++k) {
            if (nd.child[k] == REF_EMPTY) continue;
            ++used;
            Box3 b{{nd.lox[k], nd.loy[k], nd.loz[k]}, {nd.hix[k], nd.hiy[k], nd.hiz[k]}};
            for (int a = 0; a < 3; ++a)
                if (b.lo[a] < it.box.lo[a] || b.hi[a] > it.box.hi[a]) return fail("child box sticks out", idx, k);
            st.push_back({nd.child[k], b});
        }
        if (used < 2) return fail("node with fewer than two children", idx);
    }
    for (uint32_t s = 0; s < n; ++s)
        if (!seen[s]) return fail("slot in no leaf", s);
    std::printf("ok n=%u nodes=%u (bound %u) depth=%u leaves=%zu tris/leaf=%.2f treelets=%zu inner_cost=%.2f\n", n, n_nodes, n_big, depth, leaves,
                (double)tri_in_leaves / (double)leaves, n_treelets, cost_inner);
    return 0;
}
