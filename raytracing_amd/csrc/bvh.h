// bvh.h -- host-side BVH4 build for the gfx950 traversal kernel.
//
// Stands in for bvh::Bvh::build (reference call site: src/scene/object.rs:156).  The reference's crate
// builds a binary one-shape-per-leaf tree; its traversal returns every triangle whose OWN box
// (src/scene/triangle.rs:88-95) passes the slab test, and the caller takes a min over exact tests
// (object.rs:191-206).  That candidate set does not depend on the tree ("candidate-set
// invariant"), so this builder is free to produce the layout the GPU wants: 4-wide nodes of exactly one
// 128-B cache line, <= 4 triangles per leaf, triangles re-ordered so that every leaf is contiguous.
#pragma once
#include <cstdint>
#include <vector>

namespace trr {

struct Box3 {
    float lo[3], hi[3];
};

// One 128-byte node: SoA child boxes (lo/hi per axis for the 4 children), child references, padding.
struct alignas(128) Node4 {
    float lox[4], loy[4], loz[4], hix[4], hiy[4], hiz[4];
    uint32_t child[4];
    uint32_t pad[4];
};
static_assert(sizeof(Node4) == 128, "Node4 must be one cache line");

// The same node in 64 bytes, what the trace kernels read when the scene allows it (kernels.hip: k_quantise_nodes): the child
// boxes as 8-bit offsets from the node's own lower corner, in units of a per-axis power of two, rounded OUTWARDS -- child
// k spans [org + qlo_k * s, org + qhi_k * s] per axis, a superset of its Node4 box.  Byte k of each q word is child k.
#ifdef TRR_WIDE8
// experiment (-DTRR_WIDE8, scripts/experiments/build_wide8.sh): the slot of the quantised nodes holds an 8-WIDE exact node instead --
// a Node4 with its largest inner children opened while the result has at most eight children (kernels.hip: k_quantise_nodes).
// Node i of this array stands for the same subtree as Node4 i, so references stay what they are.
struct alignas(128) QNode4 {
    float lox[8], loy[8], loz[8], hix[8], hiy[8], hiz[8];
    uint32_t child[8];
    uint32_t pad[8];
};
static_assert(sizeof(QNode4) == 256, "the 8-wide node is two cache lines");
#else
struct alignas(64) QNode4 {
    float org[3];
    float sx;
    float sy, sz;
    uint32_t qlox, qloy;
    uint32_t qloz, qhix, qhiy, qhiz;
    uint32_t child[4];
};
static_assert(sizeof(QNode4) == 64, "QNode4 must be half a cache line");
#endif

constexpr uint32_t REF_EMPTY = 0xFFFFFFFFu;
constexpr uint32_t REF_LEAF = 0x80000000u;
// leaf reference: REF_LEAF | (first_slot << 2) | (count - 1), count in 1..4, first_slot < 2^29
inline uint32_t make_leaf_ref(uint32_t first, uint32_t count) { return REF_LEAF | (first << 2) | (count - 1); }

struct BuildResult {
    std::vector<Node4> nodes;      // nodes[0] is the root when root_ref is an inner reference
    std::vector<uint32_t> order;   // slot -> index into the input box array
    uint32_t root_ref = REF_EMPTY; // inner node index, leaf reference, or REF_EMPTY
    uint32_t max_depth = 0;        // depth of the 4-wide tree (root = 1)
    uint32_t stack_need = 0;       // traversal stack entries the tree can need (device_types.h, TRACE_STACK_ENTRIES)
    Box3 bounds{};                 // union of all input boxes
};

// boxes: one conservative box per primitive.  n_threads = 0 -> hardware concurrency.
void build_bvh4(const Box3* boxes, uint32_t n, uint32_t n_threads, BuildResult* out);

}   // namespace trr
