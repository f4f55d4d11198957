// device_types.h -- HBM layouts shared by the host uploader and the gfx950 kernels.
#pragma once
#include <cstdint>

#include "bvh.h"

namespace trr {

// Entries of the trace kernel's per-lane traversal stack (LDS part + HBM spill part, kernels.hip).  A 4-wide node pushes
// at most children-1 entries before descending, so a tree needs max over its inner nodes of (sum of children-1 over the
// node's ancestors and itself) entries; both builders compute that figure and trr_scene_create refuses a tree above it.
constexpr int TRACE_STACK_ENTRIES = 64;

// 64-byte leaf-ordered triangle record = what one exact test reads (half a 128-B line; a 4-triangle leaf is
// two whole lines).  n = (b-a)x(c-a) / |.| and len = |(b-a)x(c-a)| are the values Triangle::intersect
// recomputes on every call (src/scene/triangle.rs:54-56), precomputed once with identical fp32 operations.
struct alignas(64) TriRec {
    float ax, ay, az, bx;
    float by, bz, cx, cy;
    float cz, nx, ny, nz;
    uint32_t tri_id;    // obj.tri_base + row in the object's triangle array (the reference's Vec order) = index of the InterpRec
    int32_t mat;        // global material id, -1 = None
    float len;          // |ab x ac|; the kernel culls len <= delta (triangle.rs:56)
    uint32_t flags;     // bit0: material is transparent (illum 6/7, material.rs:51-53); bit1: reflective (illum 3/4, :47-49)
};
static_assert(sizeof(TriRec) == 64, "TriRec must be 64 bytes");

// 64-byte per-triangle interpolation record, indexed by obj.tri_base + tri_id; read once per shaded hit
// (src/scene/object.rs:201-203).
struct alignas(64) InterpRec {
    float na[3], nb[3], nc[3];
    float uva[2], uvb[2], uvc[2];
    float pad;
};
static_assert(sizeof(InterpRec) == 64, "InterpRec must be 64 bytes");

// Per-object record.  inv/fwd: rows of the 3x4 affine part of Object::transform()^-1 / Object::transform().
struct ObjRec {
    float inv[12];      // inv[r*4 + c], c = 3 is the translation column
    float fwd[12];
    uint32_t root;      // inner node index (absolute), leaf ref (absolute slots) or REF_EMPTY
    uint32_t tri_base;  // first InterpRec of this object
    uint32_t n_tris;    // triangles in the reference's Vec (degenerate ones included)
    uint32_t flags;     // bit0: single-triangle object -> the crate's traverse() returns it with no box test
    float slack_abs;    // absolute pruning slack in object units (1e-4 * largest |coordinate|)
    uint32_t pad[3];
};

struct MatRec {   // src/scene/material.rs:5-15
    float kd[3];
    float ks[3];
    float ns;
    float dissolve;
    int32_t illum;
    int32_t tex;        // texture id, -1 = none
    uint32_t flags;     // TRR_MAT_HAS_*
    uint32_t pad;
};

struct TexRec {
    uint64_t offset;    // byte offset into the texel atlas
    uint32_t w, h;
};

struct LightRec {   // src/scene/light.rs:5-9
    float pos[3];
    float color[3];
    float intensity;
    float pad;
};

// "Last hit so far" of a chain (camera / reflection) ray inside the transparency march (mod.rs:100-111), indexed by sample:
// written only when a raycast ends on a transparent surface, read back only when the next raycast finds nothing.
struct alignas(16) HitRec {
    float ox, oy, oz;   // origin of the raycast that found the hit (advances during the march, mod.rs:106); the hit set-up
                        // re-evaluates the barycentric weights from it, the chain ray's direction and the slot
    uint32_t slot;      // TriRec index, 0xFFFFFFFF = miss
    uint32_t obj;
    uint32_t tri;       // TriRec.tri_id (global) of the LAST hit of the march = index of its InterpRec
    int32_t mat;        // TriRec.mat
    uint32_t flags;     // TriRec.flags
};
static_assert(sizeof(HitRec) == 32, "HitRec must be 32 bytes");

// Shaded-hit record of one chain level (src/raytracer/mod.rs:124-146 inputs): 48 bytes per (level, sample), written by the
// lane that shaded the hit as three 16-byte stores and read back as 16-byte loads (one for a shadow ray's origin):
// (Px Py Pz Nx) (Ny Nz Dx Dy) (Dz tex_u tex_v mat); the diffuse colour is looked up by the resolve pass.
constexpr int LEVEL_F4 = 3;

// 64-bit kernel counters of one batch (behind the work counter in the batch's counter block)
enum { KC_NODES = 0, KC_BOXES, KC_TRIS, KC_MARCH, KC_WNODE, KC_WTRI, KC_WOUTER, KC_WLANES, KC_WTRANS, KC_LTRANS, KC_WUNI,
       KC_SHADOW_RAYS, KC_CHAIN_RAYS, KC_QUEUE_OVERFLOW, KC_SHADOW_CULLED, KC_PRIMARY_CULLED, KC_COUNT };

// Everything a kernel needs about the frame (passed by value).
struct FrameParams {
    // scene
    const Node4* nodes;
    const QNode4* qnodes;     // nullptr: traverse the exact 128-B nodes
    const TriRec* tris;
    const InterpRec* interp;
    const ObjRec* objs;
    const MatRec* mats;
    const TexRec* texs;
    const uint8_t* texels;
    const LightRec* lights;
    uint32_t n_objs, n_lights;
    // settings
    float ambient[3];
    float ambient_intensity;
    float sky_color[3];
    const uint8_t* sky;       // nullptr = constant colour
    uint32_t sky_w, sky_h;
    uint32_t aa;              // anti_aliasing
    uint32_t spp;             // samples per pixel actually traced (1 when !aa)
    uint32_t sqrt_spp;        // floor(sqrt(samples)) (mod.rs:226)
    uint64_t seed;
    // camera (basis precomputed on the host from the same inputs Camera::ray uses per call)
    float cam_pos[3];
    float cam_x[3], cam_y[3], cam_z[3];
    float cam_focal;          // -1 / tan(fov/2)
    uint32_t width, height;
    // tiling (render.rs:103-110)
    uint32_t bw, bh, nbx, nby;
    uint32_t shard, n_shards;
    // tracer
    float delta;
    uint32_t max_depth;
};

// Work description of one batch of owned blocks.
struct BatchParams {
    uint32_t first_owned;     // index (among this shard's blocks) of the batch's first block
    uint32_t n_blocks;        // blocks in the batch
    uint32_t n_pixels;        // n_blocks * bw * bh
    uint32_t n_samples;       // n_pixels * spp
    uint32_t cap;             // plane stride of the per-sample arrays (>= n_samples)
};

}   // namespace trr
