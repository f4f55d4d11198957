// launch.h -- kernel I/O descriptors and host-callable launchers (implemented in kernels.hip, used by api.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "device_types.h"

namespace trr {

struct TraceIO {
    uint32_t* work_counter;       // camera rays claimed so far (zeroed per batch)
    HitRec* hits;                 // [cap] the transparency march's "last hit so far" (written only for illum 6/7 surfaces)
    float4* level;                // [max_depth][cap][LEVEL_F4] shaded-hit records (48 B each)
    float4* term;                 // [cap] xyz = direction whose skybox() value ends a reflective chain (image sky only)
    uint8_t* nlev;                // [cap] shaded levels of the sample, 0 = the camera ray missed
    int32_t* occ;                 // [max_depth][n_lights][cap] occluder code: material id, -1 none, -2 material None
    float4* wqueue;               // [wavefronts of the grid][qcap][2] wave-local ray queues (32-B entries)
    uint32_t qcap;                // power of two >= 64 * (n_lights + 1)
    const float4* src_q;          // nullptr: the work items are the batch's camera rays; else queue entries spilled by the launch before
    const uint32_t* src_n;        // number of entries in src_q (device memory, written by the launch before)
    uint32_t src_cap;
    const float4* src_q2;         // nullable: a second source queue, taken after src_q (levels-then-drain: shadow + chain queues)
    const uint32_t* src_n2;
    uint32_t src_cap2;
    uint32_t src_compact;         // 1: src_q holds 16-B compact shadow entries of level src_level (kernels.hip: shadow entries)
    uint32_t src_level;
    float4* spill_q;              // nullable: where a wavefront pushes once it can claim no more work items
    uint32_t* spill_n;
    uint32_t spill_cap;
    uint32_t chunk_min;           // smallest claim of work items per wavefront (>= 1)
    uint32_t cull;                // 1: no shadow ray for a light that adds exactly nothing at a hit (kernels.hip: k_setup)
    uint32_t* first_obj;          // nullable: id buffers of the camera rays' first closest hits
    uint32_t* first_tri;
    unsigned long long* counters; // kernel counters (device_types.h: KC_*)
    uint2* spill;                 // traversal-stack spill, per thread of the grid
    const uint32_t* groups;       // nullable (camera-ray cull): the 64-sample groups that are traced; work item i is sample
    const uint32_t* n_groups;     //   groups[i / 64] * 64 + i % 64, and there are *n_groups * 64 work items
};
// the level-synchronous schedule's kernels
struct LevelIO {
    const float4* shadow_q;       // QUEUE: the level's shadow rays (32-B ray records, or 16-B compact entries) ...
    const float4* chain_q;        // ... and the next level's reflection rays (32-B ray records)
    uint32_t shadow_compact;      // 1: shadow_q holds compact entries, all of level shadow_level
    uint32_t shadow_level;
    const uint32_t* n_shadow;
    const uint32_t* n_chain;
    uint32_t* work_counter;
    HitRec* hits;                 // closest hit per chain ray, by queue position (camera rays: by sample)
    uint8_t* nlev;                // RAYGEN: 0 = the camera ray missed, 0xFF = hit, to be shaded by k_setup
    int32_t* occ;
    uint32_t* first_obj;
    uint32_t* first_tri;
    unsigned long long* counters;
    uint2* spill;
    const uint32_t* groups;       // RAYGEN, nullable (camera-ray cull): see TraceIO
    const uint32_t* n_groups;
};
// Camera-ray cull (kernels.hip: k_cover_*): which 8x8-pixel tiles of the frame can see a box of the scene at all.
// A scene's "cover boxes" are made once, when it is created: the child boxes of all nodes of one tree level (plus the leaves
// above it), at most COVER_BOX_BUDGET of them -- every triangle's own box lies inside one of them.
struct alignas(16) CoverBox {
    float lo[3];
    uint32_t obj;
    float hi[3];
    uint32_t pad;
};
constexpr uint32_t COVER_BOX_BUDGET = 1u << 16;
struct CoverIO {
    uint8_t* mask;                // [tiles_y][tiles_x] 1 = some leaf box projects onto the tile (or too near to tell)
    uint32_t tiles_x, tiles_y;
    uint32_t* flags;              // [0] != 0: no statement possible this frame (a box at the camera plane, a bound not finite): all tiles count
    const CoverBox* boxes;        // the scene's cover boxes (object space)
    uint32_t n_boxes;
    float pad_x_lo, pad_x_hi, pad_y_lo, pad_y_hi;   // pixels a box's projection is widened by: the samples of a pixel lie around its corner
    uint4* big;                   // rectangles of more than COVER_SMALL tiles, filled by k_cover_big
    uint32_t* n_big;
    uint32_t big_cap;
    uint32_t* groups;             // out: the batch's 64-sample groups with a covered pixel, n_groups of them
    uint32_t* n_groups;
    uint8_t* nlev;                // groups left out: their samples are camera rays that miss (nlev 0)
    unsigned long long* counters;
};
struct SetupIO {
    const float4* chain_in;
    const uint32_t* n_chain_in;
    const HitRec* hits;
    float4* level;
    float4* term;
    uint8_t* nlev;
    float4* shadow_out;
    uint32_t* n_shadow_out;
    uint32_t shadow_compact;      // 1: write 16-B compact shadow entries
    int32_t* occ;                 // occluder codes; a culled (hit, light) pair gets OCC_CULLED here instead of a shadow ray
    uint32_t cull;                // 1: cull shadow rays whose light contributes exactly nothing (kernels.hip)
    float4* chain_out;
    uint32_t* n_chain_out;
    unsigned long long* counters;
};
struct ResolveIO {
    const float4* level;
    const float4* term;
    const uint8_t* nlev;
    float* color;      // [3][cap] planes: the samples' colours, between the two resolve passes
    const int32_t* occ;
    uint8_t* strip;
};
void launch_quantise_nodes(hipStream_t st, const Node4* nodes, QNode4* out, uint32_t n, uint32_t* flag);
int trace_blocks_per_cu(bool count);
size_t trace_spill_bytes(uint32_t grid);
size_t trace_queue_bytes(uint32_t grid, uint32_t qcap);
void launch_trace(bool count, uint32_t grid, hipStream_t st, const FrameParams& fp, const BatchParams& bp, const TraceIO& io);
int level_blocks_per_cu(bool raygen, bool count);
void launch_trace_level(bool raygen, bool count, uint32_t grid, hipStream_t st, const FrameParams& fp, const BatchParams& bp,
                        const LevelIO& io);
void launch_cover_mask(hipStream_t st, const FrameParams& fp, const CoverIO& io);
void launch_cover_groups(hipStream_t st, const FrameParams& fp, const BatchParams& bp, const CoverIO& io);
// One round of the top-down expansion that makes a scene's cover boxes: the children of the nodes in frontier_in go to
// frontier_out (inner nodes) or to boxes (leaves; with `last`, every child).  node_begin: first node of every object.
void launch_cover_expand(hipStream_t st, const Node4* nodes, const uint32_t* node_begin, uint32_t n_objs, const uint32_t* frontier_in,
                         uint32_t n_in, uint32_t* frontier_out, uint32_t* n_out, CoverBox* boxes, uint32_t* n_boxes, uint32_t box_cap, bool last);
void launch_setup(bool level0, uint32_t grid, hipStream_t st, const FrameParams& fp, const BatchParams& bp, const SetupIO& io,
                  uint32_t lvl);
void launch_resolve(uint32_t grid_samples, uint32_t grid_pixels, hipStream_t st, const FrameParams& fp, const BatchParams& bp,
                    const ResolveIO& io);
void launch_assemble(hipStream_t st, uint32_t width, uint32_t height, uint32_t bw, uint32_t bh, uint32_t nbx, uint32_t nby,
                     uint32_t n_shards, size_t strip_bytes, const uint8_t* strips, uint8_t* frame);


}   // namespace trr
