// build_device.hip -- BVH4 build on the GPU for gfx950.
//
// Stands in, like bvh.cpp, for bvh::Bvh::build (reference call site: src/scene/object.rs:156) and for the per-triangle
// values Triangle::intersect / Triangle::aabb recompute on every call (src/scene/triangle.rs:54-56, 88-95).  The tree's
// shape is free (bvh.h); what has to hold is (i) every child box contains the reference's own box of every triangle
// below it and (ii) the TriRec / InterpRec values equal the host builder's bit for bit -- both are checked by
// trr_scene_validate and by the parity tests, which render the same scene with both builders.
//
// Pipeline (all HBM-bound, one pass each over n triangles): prep (raw 96-B triangles -> box, InterpRec, live flag,
// centroid bounds) -> 63-bit Morton keys -> hipCUB radix sort of (key, triangle) -> gather of the 64-B TriRecs and the
// boxes into leaf order -> Karras radix tree -> its small subtrees ("treelets") rebuilt with surface-area splits
// (lbvh_core.h) -> box fit by range unions over a pairwise-union pyramid (no atomics, no fences) -> level-by-level
// collapse into 128-B 4-wide nodes.
#include "build_device.h"

#include <hipcub/hipcub.hpp>

#include <atomic>
#include <chrono>
#include <cstring>
#include <mutex>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define TRR_HD __host__ __device__
#include "lbvh_core.h"

namespace trr {
namespace {

constexpr int BT = 256;

// order-preserving map float -> uint32 (for atomicMin / atomicMax on floats of either sign)
__host__ __device__ inline uint32_t f2ord(float f) {
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline float ord2f(uint32_t o) {
    uint32_t u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}

struct Globals {            // one small device record per build, zero-initialised except cmin
    uint32_t cmin[3];       // f2ord of the centroid bounds of the live triangles
    uint32_t cmax[3];
    uint32_t maxabs;        // bits of the largest finite |coordinate| (non-negative floats order like their bits)
    uint32_t n_live;
    uint32_t n_big;         // inner nodes of the binary tree over more than MAX_LEAF triangles (bound on the 4-wide nodes)
    uint32_t n_nodes;       // 4-wide nodes allocated so far
    uint32_t stack_need;    // traversal stack entries the tree can need (device_types.h, TRACE_STACK_ENTRIES)
    uint32_t n_roots;       // small treelet roots (k_hier, k_treelet_top): one thread rebuilds each
    uint32_t n_roots_big;   // big treelet roots (k_hier): one wavefront splits each down to small ones
};

__device__ inline void load_raw(const float* raw, uint32_t t, float f[24]) {
    const float4* p = reinterpret_cast<const float4*>(raw + (size_t)t * 24);   // 96-B rows of a hipMalloc'd array
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        float4 v = p[k];
        f[4 * k] = v.x; f[4 * k + 1] = v.y; f[4 * k + 2] = v.z; f[4 * k + 3] = v.w;
    }
}

// triangle.rs:54-56: n = (ab x ac) / |ab x ac|, len = |ab x ac|, with the operation order of api.hip's hcross/hdot/hnorm
__device__ inline void normal_of(const float f[24], float n[3], float* len) {
    float abx = f[3] - f[0], aby = f[4] - f[1], abz = f[5] - f[2];
    float acx = f[6] - f[0], acy = f[7] - f[1], acz = f[8] - f[2];
    float cx = aby * acz - abz * acy, cy = abz * acx - abx * acz, cz = abx * acy - aby * acx;
    float l = sqrtf((cx * cx + cy * cy) + cz * cz);
    n[0] = cx / l; n[1] = cy / l; n[2] = cz / l;
    *len = l;
}

__device__ inline float wave_min(float v) {
    for (int o = 32; o >= 1; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ inline float wave_max(float v) {
    for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Grid-stride over the triangles with a bounded grid: the eight reductions end in same-address atomics (one set per
// wavefront), which serialise -- one set per 64 triangles cost 14 ms for 10 M triangles, one set per resident wavefront
// costs nothing.
__global__ void __launch_bounds__(BT) k_prep(const float* __restrict__ raw, uint32_t nt, InterpRec* __restrict__ interp,
                                             Box3* __restrict__ boxes, uint8_t* __restrict__ live, Globals* g) {
    const float inf = __builtin_inff();
    float cmin[3] = {inf, inf, inf}, cmax[3] = {-inf, -inf, -inf};
    float ma = 0.0f;
    uint32_t n_live = 0;
    for (uint64_t t64 = (uint64_t)blockIdx.x * BT + threadIdx.x; t64 < nt; t64 += (uint64_t)gridDim.x * BT) {
        const uint32_t t = (uint32_t)t64;
        float f[24];
        load_raw(raw, t, f);
        float n[3], len;
        normal_of(f, n, &len);
        const bool is_live = !(len <= 0.0f);   // |ab x ac| == 0 can never be hit for any delta >= 0 (triangle.rs:56)
        // triangle.rs:88-95: Aabb::empty().grow(a).grow(b).grow(c + 1e-4)
        Box3 bx;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float c2 = f[6 + k] + 0.0001f;
            bx.lo[k] = fminf(fminf(f[k], f[3 + k]), c2);
            bx.hi[k] = fmaxf(fmaxf(f[k], f[3 + k]), c2);
        }
        boxes[t] = bx;
        live[t] = is_live ? 1 : 0;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            float v = fabsf(f[k]);
            if ((v - v == 0.0f) && v > ma) ma = v;
        }
        if (is_live) {
            ++n_live;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                float c = 0.5f * bx.lo[k] + 0.5f * bx.hi[k];
                if (!(c - c == 0.0f)) c = 0.0f;   // NaN / infinite vertices must not poison the keys
                cmin[k] = fminf(cmin[k], c);
                cmax[k] = fmaxf(cmax[k], c);
            }
        }
        float4* ip = reinterpret_cast<float4*>(interp + t);
        ip[0] = make_float4(f[9], f[10], f[11], f[12]);     // na, nb.x
        ip[1] = make_float4(f[13], f[14], f[15], f[16]);    // nb.yz, nc.xy
        ip[2] = make_float4(f[17], f[18], f[19], f[20]);    // nc.z, uva, uvb.x
        ip[3] = make_float4(f[21], f[22], f[23], 0.0f);     // uvb.y, uvc, pad
    }
    // every lane of the wavefront takes part in the reductions
#pragma unroll
    for (int k = 0; k < 3; ++k) { cmin[k] = wave_min(cmin[k]); cmax[k] = wave_max(cmax[k]); }
    ma = wave_max(ma);
    for (int o = 32; o >= 1; o >>= 1) n_live += __shfl_xor(n_live, o, 64);
    if ((threadIdx.x & 63) == 0) {
        if (n_live) {
            for (int k = 0; k < 3; ++k) {
                atomicMin(&g->cmin[k], f2ord(cmin[k]));
                atomicMax(&g->cmax[k], f2ord(cmax[k]));
            }
            atomicAdd(&g->n_live, n_live);
        }
        if (ma > 0.0f) atomicMax(&g->maxabs, __float_as_uint(ma));
    }
}

template <bool UNIFORM_PER_AXIS>
__global__ void __launch_bounds__(BT) k_keys(const Box3* __restrict__ boxes, const uint8_t* __restrict__ live, uint32_t nt,
                                             const Globals* g, uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const uint32_t t = blockIdx.x * BT + threadIdx.x;
    if (t >= nt) return;
    uint64_t key = lbvh::KEY_DEAD;
    if (live[t]) {
        float lo[3], ext[3], c[3];
        const Box3 bx = boxes[t];
        for (int k = 0; k < 3; ++k) {
            lo[k] = ord2f(g->cmin[k]);
            ext[k] = ord2f(g->cmax[k]) - lo[k];
            c[k] = 0.5f * bx.lo[k] + 0.5f * bx.hi[k];
            if (!(c[k] - c[k] == 0.0f)) c[k] = 0.0f;
        }
        // one scale for the three axes (cubic Morton cells): a flat object (a terrain) is then not split along its thin
        // axis as often as along the others (measured: 10 M-triangle heightfield frame 5.2 -> see DESIGN.md)
        if (!UNIFORM_PER_AXIS) ext[0] = ext[1] = ext[2] = fmaxf(fmaxf(ext[0], ext[1]), ext[2]);
        key = lbvh::morton63(c, lo, ext);
    }
    keys[t] = key;
    vals[t] = t;
}

// The radix tree; with treelet_max >= 3 every inner node also reports the children that root a treelet (a subtree of at most
// treelet_max triangles below a bigger node; lbvh_core.h) -- the parent knows its children's ranges from its own split.
// Treelets of more than small_max triangles go to the `big` list (k_treelet_top), the others to `roots` (k_treelet).
__global__ void __launch_bounds__(BT) k_hier(const uint64_t* __restrict__ keys, uint32_t n, lbvh::Tree tree, uint32_t treelet_max, uint32_t small_max,
                                             uint32_t* __restrict__ roots, uint32_t* __restrict__ big, Globals* g) {
    const uint32_t i = blockIdx.x * BT + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t mine[2];
    uint32_t n_mine = 0;
    if (i + 1 < n) {
        lbvh::build_inner(keys, (int64_t)n, (int64_t)i, tree);
        if (treelet_max >= 3u) {
            const uint32_t f = tree.first[i], l = tree.last[i], cl = tree.child_l[i], cr = tree.child_r[i];
            if (l - f + 1u > treelet_max) {
                if (!(cl & lbvh::LEAF_BIT)) { const uint32_t c = cl - f + 1u; if (c >= 3u && c <= treelet_max) mine[n_mine++] = cl | (c > small_max ? 0x80000000u : 0u); }
                if (!(cr & lbvh::LEAF_BIT)) { const uint32_t c = l - cr + 1u; if (c >= 3u && c <= treelet_max) mine[n_mine++] = cr | (c > small_max ? 0x80000000u : 0u); }
            } else if (i == 0u && l - f + 1u >= 3u) {
                mine[n_mine++] = (l - f + 1u > small_max) ? 0x80000000u : 0u;   // the whole object is one treelet
            }
        }
    }
    // one same-address atomic per wavefront and list
    for (uint32_t which = 0; which < 2u; ++which) {
        uint32_t cnt = 0;
        for (uint32_t k = 0; k < n_mine; ++k) cnt += ((mine[k] >> 31) == which) ? 1u : 0u;
        uint32_t incl = cnt;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t v = __shfl_up(incl, o, 64);
            if (lane >= (uint32_t)o) incl += v;
        }
        const uint32_t total = __shfl(incl, 63, 64);
        uint32_t base = 0;
        if (lane == 63u && total) base = atomicAdd(which ? &g->n_roots_big : &g->n_roots, total);
        base = __shfl(base, 63, 64);
        uint32_t at = base + incl - cnt;
        for (uint32_t k = 0; k < n_mine; ++k)
            if ((mine[k] >> 31) == which) (which ? big : roots)[at++] = mine[k] & 0x7FFFFFFFu;
    }
}

// One wavefront per BIG treelet: the same binned-SAH splits as lbvh::treelet_rebuild, made by 64 lanes together (bins in LDS,
// partition through a second buffer), down to ranges of at most small_max triangles, which are appended to `roots` for
// k_treelet.  The smaller child is always processed first, so the LDS stack holds at most log2(range) pending ranges.
__global__ void __launch_bounds__(64) k_treelet_top(lbvh::Tree tree, const uint32_t* __restrict__ big, uint32_t n_big_roots, uint32_t small_max,
                                                    uint32_t* __restrict__ vals, Box3* __restrict__ sbox, uint32_t* __restrict__ vals_tmp,
                                                    Box3* __restrict__ sbox_tmp, uint32_t* __restrict__ roots, uint32_t roots_cap, Globals* g) {
    constexpr int NB = lbvh::TL_BINS;
    __shared__ uint32_t s_lo[3][NB][3], s_hi[3][NB][3], s_cnt[3][NB];
    constexpr uint32_t STACK = 40;
    __shared__ uint32_t s_stack[STACK][3];   // node, first, last
    __shared__ uint32_t s_small[64];
    __shared__ uint32_t s_nsmall, s_base;
    if (blockIdx.x >= n_big_roots) return;
    const uint32_t lane = threadIdx.x;
    const float inf = __builtin_inff();
    uint32_t sp = 0;
    if (lane == 0) {
        const uint32_t r = big[blockIdx.x];
        s_stack[0][0] = r; s_stack[0][1] = tree.first[r]; s_stack[0][2] = tree.last[r];
        s_nsmall = 0;
    }
    sp = 1;
    __syncthreads();
    auto flush_small = [&]() {   // (called by all lanes)
        __syncthreads();
        const uint32_t m = s_nsmall;
        if (m) {
            if (lane == 0) s_base = atomicAdd(&g->n_roots, m);
            __syncthreads();
            if (lane < m && s_base + lane < roots_cap) roots[s_base + lane] = s_small[lane];
            __syncthreads();
            if (lane == 0) s_nsmall = 0;
            __syncthreads();
        }
    };
    while (sp > 0) {
        --sp;
        const uint32_t node = s_stack[sp][0], f = s_stack[sp][1], l = s_stack[sp][2], cnt = l - f + 1u;
        if (cnt <= small_max || sp + 2u > STACK) {   // (the stack cannot run out: see above; if it did, k_treelet takes the range)
            if (lane == 0) s_small[s_nsmall++] = node;
            __syncthreads();
            if (s_nsmall >= 63u) flush_small();
            continue;
        }
        // centroid bounds of the range
        float cmin[3] = {inf, inf, inf}, cmax[3] = {-inf, -inf, -inf};
        for (uint32_t i = f + lane; i <= l; i += 64u) {
            const Box3 b = sbox[i];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float c = lbvh::centroid_of(b, k);
                cmin[k] = fminf(cmin[k], c);
                cmax[k] = fmaxf(cmax[k], c);
            }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) { cmin[k] = wave_min(cmin[k]); cmax[k] = wave_max(cmax[k]); }
        float scale[3];
        bool axis_ok[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float ext = cmax[k] - cmin[k];
            axis_ok[k] = (ext > 0.0f) && (ext - ext == 0.0f);
            scale[k] = axis_ok[k] ? (float)NB / ext : 0.0f;
        }
        for (uint32_t w = lane; w < 3u * NB * 3u; w += 64u) { (&s_lo[0][0][0])[w] = 0xFFFFFFFFu; (&s_hi[0][0][0])[w] = 0u; }
        if (lane < 3u * NB) (&s_cnt[0][0])[lane] = 0u;
        __syncthreads();
        for (uint32_t i = f + lane; i <= l; i += 64u) {
            const Box3 bx = sbox[i];
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
                if (!axis_ok[ax]) continue;
                int b = (int)((lbvh::centroid_of(bx, ax) - cmin[ax]) * scale[ax]);
                b = b < 0 ? 0 : (b >= NB ? NB - 1 : b);
                atomicAdd(&s_cnt[ax][b], 1u);
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    if (bx.lo[k] == bx.lo[k]) atomicMin(&s_lo[ax][b][k], f2ord(bx.lo[k]));   // (NaN bounds are left out, as fmin / fmax do)
                    if (bx.hi[k] == bx.hi[k]) atomicMax(&s_hi[ax][b][k], f2ord(bx.hi[k]));
                }
            }
        }
        __syncthreads();
        // lane c < 21 prices the split of axis c / 7 after bin c % 7
        float cost = inf;
        if (lane < 3u * (NB - 1)) {
            const int ax = (int)lane / (NB - 1), plane = (int)lane % (NB - 1);
            if (axis_ok[ax]) {
                Box3 L{{inf, inf, inf}, {-inf, -inf, -inf}}, R{{inf, inf, inf}, {-inf, -inf, -inf}};
                uint32_t nl = 0, nr = 0;
                for (int b = 0; b < NB; ++b) {
                    const uint32_t c = s_cnt[ax][b];
                    if (!c) continue;
                    Box3 bb;
                    for (int k = 0; k < 3; ++k) { bb.lo[k] = ord2f(s_lo[ax][b][k]); bb.hi[k] = ord2f(s_hi[ax][b][k]); }
                    if (b <= plane) { L = lbvh::box_union(L, bb); nl += c; } else { R = lbvh::box_union(R, bb); nr += c; }
                }
                if (nl && nr) cost = lbvh::box_half_area(L) * (float)nl + lbvh::box_half_area(R) * (float)nr;
            }
        }
        const float best_cost = wave_min(cost);
        const unsigned long long winners = __ballot(cost == best_cost && cost < inf);
        uint32_t mid = f + (cnt - 1u) / 2u;   // no split found (all centroids equal): by count, in the order as it is
        if (winners) {
            const int c = __ffsll((long long)winners) - 1;
            const int ax = c / (NB - 1), best_bin = c % (NB - 1);
            const float c0 = cmin[ax], sc = scale[ax];
            // stable partition through the second buffer: left part from f upwards, right part from l downwards
            uint32_t n_left = 0, n_right = 0;
            for (uint32_t i0 = f; i0 <= l; i0 += 64u) {
                const uint32_t i = i0 + lane;
                const bool in = i <= l;
                Box3 bx{};
                uint32_t v = 0;
                bool left = false;
                if (in) {
                    bx = sbox[i];
                    v = vals[i];
                    int b = (int)((lbvh::centroid_of(bx, ax) - c0) * sc);
                    b = b < 0 ? 0 : (b >= NB ? NB - 1 : b);
                    left = b <= best_bin;
                }
                const unsigned long long ml = __ballot(in && left), mr = __ballot(in && !left);
                const unsigned long long below = (1ull << lane) - 1ull;
                if (in) {
                    const uint32_t at = left ? f + n_left + (uint32_t)__popcll(ml & below) : l - n_right - (uint32_t)__popcll(mr & below);
                    sbox_tmp[at] = bx;
                    vals_tmp[at] = v;
                }
                n_left += (uint32_t)__popcll(ml);
                n_right += (uint32_t)__popcll(mr);
            }
            __syncthreads();
            for (uint32_t i = f + lane; i <= l; i += 64u) { sbox[i] = sbox_tmp[i]; vals[i] = vals_tmp[i]; }
            __syncthreads();
            if (n_left > 0 && n_right > 0) mid = f + n_left - 1u;
        }
        const uint32_t lc = (mid == f) ? (lbvh::LEAF_BIT | f) : mid;
        const uint32_t rc = (mid + 1u == l) ? (lbvh::LEAF_BIT | l) : mid + 1u;
        if (lane == 0) {
            tree.child_l[node] = lc;
            tree.child_r[node] = rc;
            if (!(lc & lbvh::LEAF_BIT)) { tree.first[lc] = f; tree.last[lc] = mid; }
            if (!(rc & lbvh::LEAF_BIT)) { tree.first[rc] = mid + 1u; tree.last[rc] = l; }
            // the bigger child first, so that it is taken up last
            const bool left_bigger = (mid - f) >= (l - mid - 1u);
            const uint32_t a = left_bigger ? lc : rc, b = left_bigger ? rc : lc;
            uint32_t w = sp;
            for (const uint32_t c : {a, b}) {
                if (c & lbvh::LEAF_BIT) continue;
                s_stack[w][0] = c;
                s_stack[w][1] = (c == lc) ? f : mid + 1u;
                s_stack[w][2] = (c == lc) ? mid : l;
                ++w;
            }
        }
        sp += ((lc & lbvh::LEAF_BIT) ? 0u : 1u) + ((rc & lbvh::LEAF_BIT) ? 0u : 1u);
        __threadfence_block();
        __syncthreads();
    }
    flush_small();
}

// one thread per treelet (sequential binned-SAH rebuild, lbvh_core.h); 64-thread blocks: the threads' bins live in scratch
__global__ void __launch_bounds__(64) k_treelet(lbvh::Tree tree, const uint32_t* __restrict__ roots, uint32_t n_roots, uint32_t* __restrict__ vals,
                                                Box3* __restrict__ sbox, uint32_t* __restrict__ stack) {
    const uint32_t r = blockIdx.x * 64u + threadIdx.x;
    if (r < n_roots) lbvh::treelet_rebuild(tree, roots[r], vals, sbox, stack);
}

// inner nodes over more than max_leaf triangles: the bound on the 4-wide nodes the collapse can make
__global__ void __launch_bounds__(BT) k_count_big(uint32_t n, lbvh::Tree tree, Globals* g) {
    __shared__ uint32_t s_big;
    if (threadIdx.x == 0) s_big = 0;
    __syncthreads();
    const uint32_t i = blockIdx.x * BT + threadIdx.x;
    const bool big = (i + 1 < n) && tree.last[i] - tree.first[i] + 1u > tree.max_leaf;
    const unsigned long long m = __ballot(big);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(&s_big, (uint32_t)__popcll(m));
    __syncthreads();
    if (threadIdx.x == 0 && s_big) atomicAdd(&g->n_big, s_big);   // one same-address atomic per workgroup
}

__global__ void __launch_bounds__(BT) k_sorted_boxes(const uint32_t* __restrict__ vals, const Box3* __restrict__ boxes, uint32_t n, Box3* __restrict__ out) {
    const uint32_t s = blockIdx.x * BT + threadIdx.x;
    if (s < n) out[s] = boxes[vals[s]];
}

// next level of the pairwise-union pyramid
__global__ void __launch_bounds__(BT) k_pyramid(const Box3* __restrict__ src, uint32_t n_src, Box3* __restrict__ dst, uint32_t n_dst) {
    const uint32_t j = blockIdx.x * BT + threadIdx.x;
    if (j < n_dst) dst[j] = lbvh::pyramid_pair(src, n_src, j);
}

// box of every inner node = union of the sorted boxes first..last
__global__ void __launch_bounds__(BT) k_fit(uint32_t n, lbvh::Tree tree, lbvh::Pyramid pyr) {
    const uint32_t i = blockIdx.x * BT + threadIdx.x;
    if (i + 1 < n) tree.ibox[i] = lbvh::range_box(pyr, tree.first[i], tree.last[i]);
}

// One level of the top-down collapse: every queue item (binary inner node, index of its 4-wide node) writes its node
// and queues the children that are inner nodes of the 4-wide tree.  Node indices are handed out in level order, one
// atomic per wavefront (a prefix sum over the lanes' child counts), and a child's place in the next level's queue is
// its node index minus the first index of that level -- no second counter.
__global__ void __launch_bounds__(BT) k_collapse(const uint2* __restrict__ q_in, const uint32_t* __restrict__ pushed_in, uint32_t n_in,
                                                 uint2* __restrict__ q_out, uint32_t* __restrict__ pushed_out, uint32_t level_base, lbvh::Tree tree, Node4* __restrict__ nodes, uint32_t node_cap,
                                                 uint32_t node_base, uint32_t slot_base, Globals* g) {
    const uint32_t i = blockIdx.x * BT + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    const bool valid = i < n_in;
    uint2 item = make_uint2(0u, 0u);
    uint32_t kids[4] = {0u, 0u, 0u, 0u};
    int nk = 0;
    uint32_t c = 0;        // children of mine that become 4-wide nodes
    uint32_t pushed = 0;   // stack entries a traversal may hold while it is below this node (bvh.cpp, Collapser)
    if (valid) {
        item = q_in[i];
        nk = lbvh::collapse_children(tree, item.x, kids);
        for (int k = 0; k < nk; ++k) c += lbvh::leaf_like(tree, kids[k]) ? 0u : 1u;
        pushed = pushed_in[i] + (uint32_t)nk - 1u;
    }
    uint32_t wmax = pushed;
    for (int o = 32; o >= 1; o >>= 1) wmax = max(wmax, (uint32_t)__shfl_xor(wmax, o, 64));
    if (lane == 0u && wmax) atomicMax(&g->stack_need, wmax);
    uint32_t incl = c;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = __shfl_up(incl, o, 64);
        if (lane >= (uint32_t)o) incl += v;
    }
    const uint32_t total = __shfl(incl, 63, 64);
    uint32_t base = 0;
    if (lane == 63u && total) base = atomicAdd(&g->n_nodes, total);
    base = __shfl(base, 63, 64);
    if (!valid) return;
    uint32_t next = base + incl - c;
    Node4 nd;
    for (int k = 0; k < 4; ++k) {
        nd.pad[k] = 0;
        if (k >= nk) { lbvh::node4_set_empty(nd, k); continue; }
        lbvh::node4_set_box(nd, k, lbvh::ref_box(tree, kids[k]));
        if (lbvh::leaf_like(tree, kids[k])) {
            nd.child[k] = lbvh::leaf_ref_of(tree, kids[k], slot_base);
        } else {
            const uint32_t idx = next++;
            if (idx < node_cap && idx >= level_base) {
                q_out[idx - level_base] = make_uint2(kids[k], idx);
                pushed_out[idx - level_base] = pushed;
                nd.child[k] = node_base + idx;
            } else {
                nd.child[k] = REF_EMPTY;   // cannot happen (node_cap = n_big bounds the node count); the host checks n_nodes
            }
        }
    }
    if (item.y < node_cap) {
        float4* dst = reinterpret_cast<float4*>(nodes + item.y);
        const float4* src = reinterpret_cast<const float4*>(&nd);
#pragma unroll
        for (int k = 0; k < 8; ++k) dst[k] = src[k];
    }
}

__global__ void __launch_bounds__(BT) k_gather(const float* __restrict__ raw, const int32_t* __restrict__ matidx,
                                               const uint32_t* __restrict__ vals, uint32_t n_live, const MatRec* __restrict__ mats,
                                               uint32_t mat_base, uint32_t n_mats_obj, uint32_t tri_base, const Box3* __restrict__ boxes,
                                               TriRec* __restrict__ out) {
    const uint32_t s = blockIdx.x * BT + threadIdx.x;
    if (s >= n_live) return;
    const uint32_t t = vals[s];
    float f[24];
    load_raw(raw, t, f);
    float n[3], len;
    normal_of(f, n, &len);
    int32_t mi = matidx ? matidx[t] : -1;
    int32_t mat = (mi >= 0 && (uint32_t)mi < n_mats_obj) ? (int32_t)(mat_base + (uint32_t)mi) : -1;
    uint32_t flags = 0;
    if (mat >= 0) {
        int il = mats[mat].illum;
        if (il == 6 || il == 7) flags |= 1u;   // material.rs:51-53
        if (il == 3 || il == 4) flags |= 2u;   // material.rs:47-49
    }
    float4* dst = reinterpret_cast<float4*>(out + s);
    dst[0] = make_float4(f[0], f[1], f[2], f[3]);
    dst[1] = make_float4(f[4], f[5], f[6], f[7]);
    dst[2] = make_float4(f[8], n[0], n[1], n[2]);
    dst[3] = make_float4(__uint_as_float(tri_base + t), __int_as_float(mat), len, __uint_as_float(flags));
}

struct Tmp {   // device allocations of one build, freed on every exit path
    std::vector<void*> ptrs;
    ~Tmp() { for (void* p : ptrs) (void)hipFree(p); }
    template <class T>
    hipError_t get(T** out, size_t count) {
        void* p = nullptr;
        hipError_t e = hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T));
        if (e == hipSuccess) ptrs.push_back(p);
        *out = static_cast<T*>(p);
        return e;
    }
    void keep(void* p) {   // ownership passes to the caller
        for (auto& q : ptrs) if (q == p) q = nullptr;
    }
};

inline uint32_t blocks_for(uint64_t n) { return (uint32_t)((n + BT - 1) / BT); }

}   // namespace

// Host-to-device copy of a big pageable array (an object's raw triangles: 0.96 GB for 10 M of them).  hipMemcpy from pageable
// memory runs at ~29 GB/s on this box (one staging thread inside the runtime); here six host threads copy 8-MB chunks into a
// ring of pinned buffers (64 MB, allocated once per process) and queue one DMA per chunk on `stream`, so that the host-side
// copies and the DMA overlap.  A buffer is re-used only after the DMA of its previous chunk has finished (the event of that
// chunk, which the chunk's thread publishes through `recorded`).  Falls back to the plain copy for small arrays, when the
// pinned ring cannot be had, or when another build is using it.
namespace {
struct PinnedRing {
    static constexpr size_t CHUNK = 8u << 20;
    static constexpr int SLOTS = 8;
    std::mutex mu;
    void* buf[SLOTS] = {};
    hipEvent_t ev[SLOTS] = {};
    bool ok = false, tried = false;
    bool init() {
        if (tried) return ok;
        tried = true;
        for (int k = 0; k < SLOTS; ++k)
            if (hipHostMalloc(&buf[k], CHUNK, hipHostMallocDefault) != hipSuccess || hipEventCreateWithFlags(&ev[k], hipEventDisableTiming) != hipSuccess) return false;
        ok = true;
        return true;
    }
};
PinnedRing g_ring;

hipError_t staged_h2d(void* dst, const void* src, size_t bytes, hipStream_t stream) {
    const char* off = std::getenv("TRR_BUILD_STAGING");
    if (bytes < (64u << 20) || (off && off[0] == '0')) return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream);
    std::unique_lock<std::mutex> lock(g_ring.mu, std::try_to_lock);
    if (!lock.owns_lock() || !g_ring.init()) return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream);
    int dev = 0;
    (void)hipGetDevice(&dev);
    const size_t C = PinnedRing::CHUNK;
    const int64_t n_chunks = (int64_t)((bytes + C - 1) / C);
    std::atomic<int64_t> next{0};
    std::atomic<int64_t> recorded[PinnedRing::SLOTS];
    for (auto& r : recorded) r.store(-1);
    std::atomic<int> failed{0};
    auto worker = [&]() {
        if (hipSetDevice(dev) != hipSuccess) { failed.store((int)hipErrorInvalidDevice); return; }
        for (;;) {
            const int64_t i = next.fetch_add(1);
            if (i >= n_chunks || failed.load() != 0) return;
            const int slot = (int)(i % PinnedRing::SLOTS);
            if (i >= PinnedRing::SLOTS) {
                while (recorded[slot].load(std::memory_order_acquire) != i - PinnedRing::SLOTS && failed.load() == 0) std::this_thread::yield();
                if (failed.load() != 0) return;
                hipError_t e = hipEventSynchronize(g_ring.ev[slot]);   // the DMA out of this buffer has finished
                if (e != hipSuccess) { failed.store((int)e); return; }
            }
            const size_t o = (size_t)i * C, len = std::min(C, bytes - o);
            std::memcpy(g_ring.buf[slot], static_cast<const char*>(src) + o, len);
            hipError_t e = hipMemcpyAsync(static_cast<char*>(dst) + o, g_ring.buf[slot], len, hipMemcpyHostToDevice, stream);
            if (e == hipSuccess) e = hipEventRecord(g_ring.ev[slot], stream);
            if (e != hipSuccess) { failed.store((int)e); return; }
            recorded[slot].store(i, std::memory_order_release);
        }
    };
    std::vector<std::thread> th;
    const int n_threads = (int)std::min<int64_t>(6, n_chunks);
    for (int k = 0; k < n_threads; ++k) th.emplace_back(worker);
    for (auto& t : th) t.join();
    if (failed.load() != 0) return (hipError_t)failed.load();
    // the ring is free for the next build once its DMAs are done: the pinned buffers must not be overwritten before
    return hipStreamSynchronize(stream);
}
}   // namespace

#define BCK(expr)                                                                                 \
    do {                                                                                          \
        hipError_t e__ = (expr);                                                                  \
        if (e__ != hipSuccess) {                                                                  \
            if (err) *err = std::string(#expr) + ": " + hipGetErrorString(e__);                   \
            return e__;                                                                           \
        }                                                                                         \
    } while (0)

hipError_t build_object_device(const float* h_tris, const int32_t* h_matidx, uint32_t nt, const MatRec* d_mats, uint32_t mat_base,
                               uint32_t n_mats_obj, uint32_t node_base, uint32_t slot_base, uint32_t tri_base, InterpRec* d_interp, hipStream_t stream,
                               DeviceBuildOut* out, std::string* err) {
    *out = DeviceBuildOut{};
    if (nt == 0) return hipSuccess;
    // TRR_DEBUG=1: wall-clock marks of the stages (each mark waits for the stream)
    const bool debug = std::getenv("TRR_DEBUG") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto mark = [&](const char* what) {
        if (!debug) return;
        (void)hipStreamSynchronize(stream);
        auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[trr build] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
        t_prev = now;
    };
    Tmp tmp;
    float* raw = nullptr;
    int32_t* matidx = nullptr;
    Box3* boxes = nullptr;
    uint8_t* live = nullptr;
    Globals* g = nullptr;
    uint64_t *keys_a = nullptr, *keys_b = nullptr;
    uint32_t *vals_a = nullptr, *vals_b = nullptr;
    BCK(tmp.get(&raw, (size_t)nt * 24));
    BCK(tmp.get(&boxes, nt));
    BCK(tmp.get(&live, nt));
    BCK(tmp.get(&g, 1));
    BCK(tmp.get(&keys_a, nt));
    BCK(tmp.get(&keys_b, nt));
    BCK(tmp.get(&vals_a, nt));
    BCK(tmp.get(&vals_b, nt));
    mark("allocations");
    BCK(staged_h2d(raw, h_tris, (size_t)nt * 24 * sizeof(float), stream));
    if (h_matidx) {
        BCK(tmp.get(&matidx, nt));
        BCK(hipMemcpyAsync(matidx, h_matidx, (size_t)nt * sizeof(int32_t), hipMemcpyHostToDevice, stream));
    }
    Globals h{};
    for (int k = 0; k < 3; ++k) { h.cmin[k] = 0xFFFFFFFFu; h.cmax[k] = 0u; }
    BCK(hipMemcpyAsync(g, &h, sizeof h, hipMemcpyHostToDevice, stream));
    mark("host to device");

    hipLaunchKernelGGL(k_prep, dim3(std::min(blocks_for(nt), 2048u)), dim3(BT), 0, stream, raw, nt, d_interp, boxes, live, g);
    BCK(hipGetLastError());
    if (std::getenv("TRR_LBVH_PER_AXIS"))   // experiments: every axis scaled to its own extent
        hipLaunchKernelGGL(k_keys<true>, dim3(blocks_for(nt)), dim3(BT), 0, stream, boxes, live, nt, g, keys_a, vals_a);
    else
        hipLaunchKernelGGL(k_keys<false>, dim3(blocks_for(nt)), dim3(BT), 0, stream, boxes, live, nt, g, keys_a, vals_a);
    BCK(hipGetLastError());
    BCK(hipMemcpyAsync(&h, g, sizeof h, hipMemcpyDeviceToHost, stream));
    BCK(hipStreamSynchronize(stream));
    mark("prep + keys");
    const uint32_t n = h.n_live;
    if (n > nt) {
        if (err) *err = "device build: live-triangle count out of range";
        return hipErrorUnknown;
    }
    __builtin_memcpy(&out->maxabs, &h.maxabs, 4);
    out->n_live = n;
    if (n == 0) return hipSuccess;

    // (key, triangle) sorted by key; dead triangles (KEY_DEAD) end up behind the n live ones
    size_t sort_bytes = 0;
    BCK(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, keys_a, keys_b, vals_a, vals_b, (int)nt, 0, 64, stream));
    uint8_t* sort_tmp = nullptr;
    BCK(tmp.get(&sort_tmp, sort_bytes));
    BCK(hipcub::DeviceRadixSort::SortPairs(sort_tmp, sort_bytes, keys_a, keys_b, vals_a, vals_b, (int)nt, 0, 64, stream));
    const uint64_t* keys = keys_b;
    uint32_t* vals = vals_b;
    mark("sort");

    TriRec* tris = nullptr;
    BCK(tmp.get(&tris, n));
    uint32_t max_leaf = 2;   // measured: leaves of <= 2 trace 3-10 % faster than <= 4 (more, smaller boxes; as bvh.cpp's ALWAYS_LEAF)
    if (const char* e = std::getenv("TRR_LBVH_MAX_LEAF")) max_leaf = (uint32_t)std::min(4, std::max(1, std::atoi(e)));   // experiments
    if (n <= max_leaf) {   // the whole object is one leaf
        hipLaunchKernelGGL(k_gather, dim3(blocks_for(n)), dim3(BT), 0, stream, raw, matidx, (const uint32_t*)vals, n, d_mats, mat_base, n_mats_obj, tri_base, boxes, tris);
        BCK(hipGetLastError());
        BCK(hipStreamSynchronize(stream));
        out->root_ref = make_leaf_ref(slot_base, n);
        out->tris = tris;
        tmp.keep(tris);
        return hipSuccess;
    }

    lbvh::Tree tree{};
    BCK(tmp.get(&tree.child_l, n - 1));
    BCK(tmp.get(&tree.child_r, n - 1));
    BCK(tmp.get(&tree.first, n - 1));
    BCK(tmp.get(&tree.last, n - 1));
    BCK(tmp.get(&tree.ibox, n - 1));
    tree.max_leaf = max_leaf;
    Box3* pyr_base = nullptr;
    lbvh::Pyramid pyr{};
    const uint32_t pyr_total = lbvh::pyramid_layout(n, pyr.off, &pyr.levels);
    BCK(tmp.get(&pyr_base, pyr_total));
    pyr.base = pyr_base;
    tree.leaf_box = pyr_base;
    // treelets (lbvh_core.h): subtrees of at most treelet_max triangles are rebuilt with SAH splits -- ranges above
    // small_max by one wavefront each, the ranges below by one thread each.  0: plain radix tree.  Measured:
    // 2048 / 8 brings the 4K frames of the three BASELINE meshes to within 0-3 % of the host SAH tree's for 3-10 ms.
    uint32_t treelet_max = 2048, small_max = 8;
    if (const char* e = std::getenv("TRR_TREELET_MAX")) treelet_max = (uint32_t)std::max(0, std::atoi(e));   // experiments
    if (const char* e = std::getenv("TRR_TREELET_SMALL")) small_max = (uint32_t)std::max(3, std::atoi(e));
    uint32_t *roots = nullptr, *big_roots = nullptr, *tl_stack = nullptr, *vals_tmp = nullptr;
    Box3* sbox_tmp = nullptr;
    const uint32_t roots_cap = n / 2 + 1;
    if (treelet_max >= 3u) {
        BCK(tmp.get(&roots, roots_cap));
        BCK(tmp.get(&big_roots, roots_cap));
        BCK(tmp.get(&tl_stack, n));
        BCK(tmp.get(&vals_tmp, n));
        BCK(tmp.get(&sbox_tmp, n));
    }
    // level 0 of the pyramid = the boxes in sorted order (the treelet pass permutes them together with vals)
    hipLaunchKernelGGL(k_sorted_boxes, dim3(blocks_for(n)), dim3(BT), 0, stream, (const uint32_t*)vals, (const Box3*)boxes, n, pyr_base);
    BCK(hipGetLastError());
    hipLaunchKernelGGL(k_hier, dim3(blocks_for(n - 1)), dim3(BT), 0, stream, keys, n, tree, treelet_max, small_max, roots, big_roots, g);
    BCK(hipGetLastError());
    if (treelet_max >= 3u) {
        BCK(hipMemcpyAsync(&h, g, sizeof h, hipMemcpyDeviceToHost, stream));
        BCK(hipStreamSynchronize(stream));
        if (h.n_roots > roots_cap || h.n_roots_big > roots_cap) {
            if (err) *err = "device build: treelet count out of range";
            return hipErrorUnknown;
        }
        mark("sorted boxes + radix tree");
        if (h.n_roots_big) {
            hipLaunchKernelGGL(k_treelet_top, dim3(h.n_roots_big), dim3(64), 0, stream, tree, (const uint32_t*)big_roots, h.n_roots_big, small_max, vals,
                               pyr_base, vals_tmp, sbox_tmp, roots, roots_cap, g);
            BCK(hipGetLastError());
            BCK(hipMemcpyAsync(&h, g, sizeof h, hipMemcpyDeviceToHost, stream));
            BCK(hipStreamSynchronize(stream));
            if (h.n_roots > roots_cap) {
                if (err) *err = "device build: treelet count out of range";
                return hipErrorUnknown;
            }
            mark("big treelets");
        }
        if (h.n_roots) {
            hipLaunchKernelGGL(k_treelet, dim3((h.n_roots + 63u) / 64u), dim3(64), 0, stream, tree, (const uint32_t*)roots, h.n_roots, vals, pyr_base, tl_stack);
            BCK(hipGetLastError());
        }
        mark("small treelets");
    }
    hipLaunchKernelGGL(k_count_big, dim3(blocks_for(n - 1)), dim3(BT), 0, stream, n, tree, g);
    BCK(hipGetLastError());
    hipLaunchKernelGGL(k_gather, dim3(blocks_for(n)), dim3(BT), 0, stream, raw, matidx, (const uint32_t*)vals, n, d_mats, mat_base, n_mats_obj, tri_base, boxes, tris);
    BCK(hipGetLastError());
    {
        uint32_t m = n;
        for (uint32_t k = 1; k < pyr.levels; ++k) {
            const uint32_t m2 = (m + 1) >> 1;
            hipLaunchKernelGGL(k_pyramid, dim3(blocks_for(m2)), dim3(BT), 0, stream, pyr_base + pyr.off[k - 1], m, pyr_base + pyr.off[k], m2);
            BCK(hipGetLastError());
            m = m2;
        }
    }
    hipLaunchKernelGGL(k_fit, dim3(blocks_for(n - 1)), dim3(BT), 0, stream, n, tree, pyr);
    BCK(hipGetLastError());
    BCK(hipMemcpyAsync(&h, g, sizeof h, hipMemcpyDeviceToHost, stream));
    BCK(hipStreamSynchronize(stream));
    mark("gather + fit");
    const uint32_t n_big = h.n_big;   // the root is one of them (n > max_leaf)
    if (n_big == 0 || n_big > n - 1) {
        if (err) *err = "device build: inner-node count out of range";
        return hipErrorUnknown;
    }

    Node4* nodes = nullptr;
    uint2* q[2] = {nullptr, nullptr};
    uint32_t* q_pushed[2] = {nullptr, nullptr};
    BCK(tmp.get(&nodes, n_big));
    BCK(tmp.get(&q[0], n_big));
    BCK(tmp.get(&q[1], n_big));
    BCK(tmp.get(&q_pushed[0], n_big));
    BCK(tmp.get(&q_pushed[1], n_big));
    BCK(hipMemsetAsync(q_pushed[0], 0, sizeof(uint32_t), stream));
    // level 1: the root of the binary tree becomes 4-wide node 0
    const uint2 root_item = make_uint2(0u, 0u);
    BCK(hipMemcpyAsync(q[0], &root_item, sizeof root_item, hipMemcpyHostToDevice, stream));
    h.n_nodes = 1;
    BCK(hipMemcpyAsync(g, &h, sizeof h, hipMemcpyHostToDevice, stream));
    uint32_t n_in = 1, depth = 0, level_base = 1;   // level_base: index of the first node the level's children get
    int cur = 0;
    while (n_in > 0) {
        ++depth;
        if (depth > 4096) {
            if (err) *err = "device build: collapse did not terminate";
            return hipErrorUnknown;
        }
        hipLaunchKernelGGL(k_collapse, dim3(blocks_for(n_in)), dim3(BT), 0, stream, q[cur], q_pushed[cur], n_in, q[cur ^ 1], q_pushed[cur ^ 1],
                           level_base, tree, nodes, n_big, node_base, slot_base, g);
        BCK(hipGetLastError());
        BCK(hipMemcpyAsync(&h, g, sizeof h, hipMemcpyDeviceToHost, stream));
        BCK(hipStreamSynchronize(stream));
        if (h.n_nodes > n_big || h.n_nodes < level_base) {
            if (err) *err = "device build: node count out of range";
            return hipErrorUnknown;
        }
        n_in = h.n_nodes - level_base;
        level_base = h.n_nodes;
        cur ^= 1;
    }
    mark("collapse");
    out->n_nodes = h.n_nodes;
    out->stack_need = h.stack_need;
    out->depth = depth;
    out->root_ref = node_base;   // 4-wide node 0 of this object
    out->nodes = nodes;
    out->tris = tris;
    tmp.keep(nodes);
    tmp.keep(tris);
    return hipSuccess;
}

}   // namespace trr
