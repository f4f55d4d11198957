// lbvh_core.h -- per-element bodies of the device BVH build (build_device.hip), written as plain host/device
// functions so that tests/test_lbvh_core.py can run the same code serially on the CPU.
//
// Stands in, like bvh.cpp, for bvh::Bvh::build (reference call site: src/scene/object.rs:156); the tree's shape is free
// (bvh.h, "candidate-set invariant").  Algorithm: 63-bit Morton keys of the box centroids, radix sort, the radix-tree
// construction of Karras 2012 ("Maximizing Parallelism in the Construction of BVHs, Octrees, and k-d Trees": every
// inner node finds its key range and split independently), box fit, then a top-down collapse of the binary tree into
// the 128-byte 4-wide nodes of bvh.h with <= 4 triangles per leaf.
//
// Box fit: an inner node covers a contiguous range of the sorted triangles, so its box is a range union.  A pyramid of
// pairwise unions over the sorted boxes (level k entry j = boxes [j*2^k, (j+1)*2^k)) answers every range with at most
// 2*log2(range) reads and needs no communication between threads (the usual bottom-up walk with arrival counters
// needs a release/acquire fence per step, which on gfx950 writes back and invalidates a whole per-XCD L2: measured
// 63 ms for 10 M triangles against 1 ms for this).
#pragma once
#include <cstdint>

#include "bvh.h"

#ifndef TRR_HD
#define TRR_HD
#endif

namespace trr {
namespace lbvh {

constexpr uint32_t LEAF_BIT = 0x80000000u;   // in child references of the BINARY tree: low bits = sorted position
constexpr uint32_t NONE = 0xFFFFFFFFu;
constexpr uint64_t KEY_DEAD = ~0ull;         // key of a triangle left out of the tree (sorts last; live keys use 63 bits)
constexpr uint32_t MAX_LEAF = 4;             // most triangles a leaf reference can hold (bvh.h)

// spreads the low 21 bits of v to every third bit
TRR_HD inline uint64_t spread3(uint32_t v) {
    uint64_t x = v & 0x1FFFFFu;
    x = (x | (x << 32)) & 0x001F00000000FFFFull;
    x = (x | (x << 16)) & 0x001F0000FF0000FFull;
    x = (x | (x << 8)) & 0x100F00F00F00F00Full;
    x = (x | (x << 4)) & 0x10C30C30C30C30C3ull;
    x = (x | (x << 2)) & 0x1249249249249249ull;
    return x;
}

// c: centroid, lo/ext: bounds of all live centroids.  21 bits per axis.
TRR_HD inline uint64_t morton63(const float c[3], const float lo[3], const float ext[3]) {
    uint32_t q[3];
    for (int k = 0; k < 3; ++k) {
        float f = ext[k] > 0.0f ? (c[k] - lo[k]) / ext[k] : 0.0f;
        f = f * 2097152.0f;
        q[k] = !(f > 0.0f) ? 0u : (f >= 2097151.0f ? 2097151u : (uint32_t)f);
    }
    return (spread3(q[0]) << 2) | (spread3(q[1]) << 1) | spread3(q[2]);
}

// length of the common prefix of keys i and j (ties broken by the positions, so equal keys still form a tree)
TRR_HD inline int common_prefix(const uint64_t* keys, int64_t n, int64_t i, int64_t j) {
    if (j < 0 || j >= n) return -1;
    uint64_t a = keys[i], b = keys[j];
    if (a != b) return __builtin_clzll(a ^ b);
    return 64 + __builtin_clz((uint32_t)i ^ (uint32_t)j);
}

constexpr int MAX_LEVELS = 32;
struct Pyramid {              // pairwise-union pyramid over the n sorted triangle boxes
    const Box3* base;         // all levels behind one another
    uint32_t off[MAX_LEVELS]; // first entry of level k (level 0 = the sorted boxes themselves)
    uint32_t levels;
};

struct Tree {                 // the binary radix tree over n sorted leaves (n >= 2): inner nodes 0..n-2, root = 0
    uint32_t* child_l;        // [n-1]  reference: LEAF_BIT | position, or inner index
    uint32_t* child_r;        // [n-1]
    uint32_t* first;          // [n-1]  key range covered by the inner node
    uint32_t* last;           // [n-1]
    Box3* ibox;               // [n-1]  boxes of the inner nodes (fit pass)
    const Box3* leaf_box;     // [n]    triangle boxes in sorted order (= level 0 of the pyramid)
    uint32_t max_leaf;        // subtrees of at most this many triangles become one leaf (1..MAX_LEAF)
};

// Karras 2012, section 4 ("determine range", "find split") for inner node i.
TRR_HD inline void build_inner(const uint64_t* keys, int64_t n, int64_t i, const Tree& t) {
    const int d = (common_prefix(keys, n, i, i + 1) - common_prefix(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = common_prefix(keys, n, i, i - d);
    int64_t lmax = 2;
    while (common_prefix(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int64_t l = 0;
    for (int64_t s = lmax / 2; s >= 1; s /= 2)
        if (common_prefix(keys, n, i, i + (l + s) * d) > dmin) l += s;
    const int64_t j = i + l * d;
    const int dnode = common_prefix(keys, n, i, j);
    int64_t s = 0, step = l;
    do {
        step = (step + 1) >> 1;
        if (common_prefix(keys, n, i, i + (s + step) * d) > dnode) s += step;
    } while (step > 1);
    const int64_t gamma = i + s * d + (d < 0 ? -1 : 0);
    const int64_t lo = i < j ? i : j, hi = i < j ? j : i;
    const uint32_t cl = (lo == gamma) ? (LEAF_BIT | (uint32_t)gamma) : (uint32_t)gamma;
    const uint32_t cr = (hi == gamma + 1) ? (LEAF_BIT | (uint32_t)(gamma + 1)) : (uint32_t)(gamma + 1);
    t.child_l[i] = cl;
    t.child_r[i] = cr;
    t.first[i] = (uint32_t)lo;
    t.last[i] = (uint32_t)hi;
}

TRR_HD inline float fmin_(float a, float b) { return __builtin_fminf(a, b); }
TRR_HD inline float fmax_(float a, float b) { return __builtin_fmaxf(a, b); }
TRR_HD inline Box3 box_union(const Box3& a, const Box3& b) {
    Box3 r;
    for (int k = 0; k < 3; ++k) { r.lo[k] = fmin_(a.lo[k], b.lo[k]); r.hi[k] = fmax_(a.hi[k], b.hi[k]); }
    return r;
}
// level sizes of the pyramid over n boxes: n, ceil(n/2), ... 1; returns the total entry count
TRR_HD inline uint32_t pyramid_layout(uint32_t n, uint32_t off[MAX_LEVELS], uint32_t* levels) {
    uint32_t total = 0, m = n, k = 0;
    for (;;) {
        off[k++] = total;
        total += m;
        if (m <= 1 || k == MAX_LEVELS) break;
        m = (m + 1) >> 1;
    }
    *levels = k;
    return total;
}
// entry j of the next level from a level of n_src entries
TRR_HD inline Box3 pyramid_pair(const Box3* src, uint32_t n_src, uint32_t j) {
    return (2 * j + 1 < n_src) ? box_union(src[2 * j], src[2 * j + 1]) : src[2 * j];
}
// union of the sorted boxes a..b (inclusive): whole pyramid blocks inside [a, b], bottom-up
TRR_HD inline Box3 range_box(const Pyramid& p, uint32_t a, uint32_t b) {
    const float inf = __builtin_inff();
    Box3 acc{{inf, inf, inf}, {-inf, -inf, -inf}};
    uint32_t l = a, r = b + 1, k = 0;
    while (l < r && k < p.levels) {
        if (l & 1u) acc = box_union(acc, p.base[p.off[k] + l++]);
        if (r & 1u) acc = box_union(acc, p.base[p.off[k] + --r]);
        l >>= 1;
        r >>= 1;
        ++k;
    }
    return acc;
}

TRR_HD inline float box_half_area(const Box3& b) {   // bvh.cpp half_area
    float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    if (!(dx >= 0.0f) || !(dy >= 0.0f) || !(dz >= 0.0f)) return 0.0f;
    float a = dx * dy + dy * dz + dz * dx;
    return (a - a == 0.0f) ? a : 3.402823466e+38f;
}

// a child of the binary tree that becomes a LEAF of the 4-wide tree: a single triangle or a subtree of <= 4
TRR_HD inline bool leaf_like(const Tree& t, uint32_t ref) {
    return (ref & LEAF_BIT) || (t.last[ref] - t.first[ref] + 1u <= t.max_leaf);
}
TRR_HD inline Box3 ref_box(const Tree& t, uint32_t ref) {
    return (ref & LEAF_BIT) ? t.leaf_box[ref & ~LEAF_BIT] : t.ibox[ref];
}

// Children of the 4-wide node rooted at binary inner node b: its two children, then repeatedly the child with the
// largest box that is not leaf-like is replaced by its own two children (bvh.cpp's greedy rule; its default, a dynamic
// programme over the whole binary tree, needs a bottom-up pass that this top-down, level-by-level collapse does not have).
// Returns the number of children (2..4) and fills kids[] with binary-tree references.
TRR_HD inline int collapse_children(const Tree& t, uint32_t b, uint32_t kids[4]) {
    int nk = 2;
    kids[0] = t.child_l[b];
    kids[1] = t.child_r[b];
    while (nk < 4) {
        int pick = -1;
        float pa = -1.0f;
        for (int k = 0; k < nk; ++k) {
            if (leaf_like(t, kids[k])) continue;
            float a = box_half_area(t.ibox[kids[k]]);
            if (a > pa) { pa = a; pick = k; }
        }
        if (pick < 0) break;
        uint32_t p = kids[pick];
        kids[pick] = t.child_l[p];
        kids[nk++] = t.child_r[p];
    }
    return nk;
}

// Leaf reference of the 4-wide tree for a leaf-like child (slots are sorted positions + slot_base).
TRR_HD inline uint32_t leaf_ref_of(const Tree& t, uint32_t ref, uint32_t slot_base) {
    if (ref & LEAF_BIT) return REF_LEAF | (((ref & ~LEAF_BIT) + slot_base) << 2);
    return REF_LEAF | ((t.first[ref] + slot_base) << 2) | (t.last[ref] - t.first[ref]);
}

// ---------------------------------------------------------------------------------------------------------------
// Treelets: the bottom of the radix tree rebuilt with surface-area splits.  Measured on the host builder:
// most of what a Morton-order tree loses against a SAH tree it loses in its small subtrees.  A subtree of at most
// `treelet_max` triangles (a "treelet": a contiguous range [a, b] of the sorted order) is rebuilt by ONE thread, top-down,
// by binned SAH over the three axes like bvh.cpp's Builder::build, permuting the range in place.  A rebuilt node still
// covers a contiguous range, so the box fit and the collapse work on it unchanged, and it keeps Karras' numbering -- a left
// child has the index of its last leaf, a right child of its first -- so the rebuilt subtree uses exactly the node slots of
// the old one and needs no allocation.
constexpr int TL_BINS = 8;

TRR_HD inline float centroid_of(const Box3& b, int k) {
    const float c = 0.5f * b.lo[k] + 0.5f * b.hi[k];
    return (c - c == 0.0f) ? c : 0.0f;   // NaN / infinite vertices must not poison the binning (as bvh.cpp)
}

// vals / sbox: the sorted triangle ids and their boxes (both permuted together); stack: one word per sorted position, the
// thread uses the slice [first, last] of its treelet.
TRR_HD inline void treelet_rebuild(const Tree& t, uint32_t root, uint32_t* vals, Box3* sbox, uint32_t* stack) {
    const float inf = __builtin_inff();
    const uint32_t base = t.first[root];
    uint32_t sp = 0;
    stack[base + sp++] = root;
    while (sp > 0) {
        const uint32_t node = stack[base + --sp];
        const uint32_t f = t.first[node], l = t.last[node], cnt = l - f + 1u;
        uint32_t mid = f + (cnt - 1u) / 2u;   // last position of the left part; the default splits by count
        if (cnt > 2u) {
            float cmin[3] = {inf, inf, inf}, cmax[3] = {-inf, -inf, -inf};
            for (uint32_t i = f; i <= l; ++i) {
                const Box3 b = sbox[i];
                for (int k = 0; k < 3; ++k) {
                    const float c = centroid_of(b, k);
                    cmin[k] = fmin_(cmin[k], c);
                    cmax[k] = fmax_(cmax[k], c);
                }
            }
            float best_cost = inf, best_scale = 0.0f;
            int best_axis = -1, best_bin = -1;
            for (int ax = 0; ax < 3; ++ax) {
                const float ext = cmax[ax] - cmin[ax];
                if (!(ext > 0.0f) || !(ext - ext == 0.0f)) continue;
                const float scale = (float)TL_BINS / ext;
                Box3 bb[TL_BINS];
                uint32_t bc[TL_BINS];
                for (int b = 0; b < TL_BINS; ++b) { bb[b] = Box3{{inf, inf, inf}, {-inf, -inf, -inf}}; bc[b] = 0; }
                for (uint32_t i = f; i <= l; ++i) {
                    const Box3 bx = sbox[i];
                    int b = (int)((centroid_of(bx, ax) - cmin[ax]) * scale);
                    b = b < 0 ? 0 : (b >= TL_BINS ? TL_BINS - 1 : b);
                    bc[b]++;
                    bb[b] = box_union(bb[b], bx);
                }
                float right_area[TL_BINS];
                uint32_t right_cnt[TL_BINS];
                Box3 acc{{inf, inf, inf}, {-inf, -inf, -inf}};
                uint32_t c = 0;
                for (int b = TL_BINS - 1; b > 0; --b) {
                    acc = box_union(acc, bb[b]);
                    c += bc[b];
                    right_area[b] = box_half_area(acc);
                    right_cnt[b] = c;
                }
                acc = Box3{{inf, inf, inf}, {-inf, -inf, -inf}};
                c = 0;
                for (int b = 0; b < TL_BINS - 1; ++b) {
                    acc = box_union(acc, bb[b]);
                    c += bc[b];
                    if (c == 0 || right_cnt[b + 1] == 0) continue;
                    const float cost = box_half_area(acc) * (float)c + right_area[b + 1] * (float)right_cnt[b + 1];
                    if (cost < best_cost) { best_cost = cost; best_axis = ax; best_bin = b; best_scale = scale; }
                }
            }
            if (best_axis >= 0) {   // partition [f, l] in place: bins <= best_bin first
                const float c0 = cmin[best_axis];
                uint32_t i = f, j = l;
                for (;;) {
                    while (i <= j) {
                        int b = (int)((centroid_of(sbox[i], best_axis) - c0) * best_scale);
                        b = b < 0 ? 0 : (b >= TL_BINS ? TL_BINS - 1 : b);
                        if (b > best_bin) break;
                        ++i;
                    }
                    while (i < j) {
                        int b = (int)((centroid_of(sbox[j], best_axis) - c0) * best_scale);
                        b = b < 0 ? 0 : (b >= TL_BINS ? TL_BINS - 1 : b);
                        if (b <= best_bin) break;
                        --j;
                    }
                    if (i >= j) break;
                    const Box3 tb = sbox[i]; sbox[i] = sbox[j]; sbox[j] = tb;
                    const uint32_t tv = vals[i]; vals[i] = vals[j]; vals[j] = tv;
                    ++i;
                    --j;
                }
                if (i > f && i <= l) mid = i - 1u;   // (both sides non-empty: the cost loop skipped empty sides)
            }
        }
        const uint32_t lc = (mid == f) ? (LEAF_BIT | f) : mid;
        const uint32_t rc = (mid + 1u == l) ? (LEAF_BIT | l) : mid + 1u;
        t.child_l[node] = lc;
        t.child_r[node] = rc;
        if (!(lc & LEAF_BIT)) { t.first[lc] = f; t.last[lc] = mid; stack[base + sp++] = lc; }
        if (!(rc & LEAF_BIT)) { t.first[rc] = mid + 1u; t.last[rc] = l; stack[base + sp++] = rc; }
    }
}

TRR_HD inline void node4_set_box(Node4& nd, int k, const Box3& bx) {
    nd.lox[k] = bx.lo[0]; nd.loy[k] = bx.lo[1]; nd.loz[k] = bx.lo[2];
    nd.hix[k] = bx.hi[0]; nd.hiy[k] = bx.hi[1]; nd.hiz[k] = bx.hi[2];
}
TRR_HD inline void node4_set_empty(Node4& nd, int k) {   // never hit: see bvh.cpp
    const float inf = __builtin_inff();
    nd.lox[k] = nd.loy[k] = nd.loz[k] = inf;
    nd.hix[k] = nd.hiy[k] = nd.hiz[k] = inf;
    nd.child[k] = REF_EMPTY;
}

}   // namespace lbvh
}   // namespace trr
