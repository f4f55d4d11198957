// Runs the per-element bodies of the device BVH build (raytracing_amd/csrc/lbvh_core.h) serially on the CPU, in the
// order of the kernels of build_device.hip, and checks the resulting 4-wide tree: every sorted position in exactly one
// leaf, every child box containing every triangle box below it.  Built and run by tests/test_lbvh_core.py.
//   usage: lbvh_core_check <n> <seed> <mode>     mode 0: uniform, 1: many identical centroids, 2: on a line, 3: clustered
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
    std::printf("ok n=%u nodes=%u (bound %u) depth=%u leaves=%zu tris/leaf=%.2f\n", n, n_nodes, n_big, depth, leaves,
                (double)tri_in_leaves / (double)leaves);
    return 0;
}
