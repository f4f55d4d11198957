// build_device.h -- BVH4 build on the GPU: the host passes one object's raw triangles, the
// device produces the same three arrays the host builder (bvh.cpp + the prep loop of trr_scene_create) produces:
// 4-wide nodes, leaf-ordered TriRecs, InterpRecs.  See lbvh_core.h for the algorithm.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

#include "device_types.h"

namespace trr {

struct DeviceBuildOut {
    Node4* nodes = nullptr;    // hipMalloc'd by the build, n_nodes used entries; the caller hipFree's it
    TriRec* tris = nullptr;    // hipMalloc'd by the build, n_live entries; the caller hipFree's it
    uint32_t n_nodes = 0;
    uint32_t n_live = 0;
    uint32_t root_ref = REF_EMPTY;   // absolute reference (node_base / slot_base applied)
    uint32_t depth = 0;              // levels of the 4-wide tree (root = 1)
    uint32_t stack_need = 0;         // traversal stack entries the tree can need (device_types.h)
    float maxabs = 0.0f;             // largest finite |coordinate| (ObjRec::slack_abs)
};

// h_tris: nt x 24 floats (Triangle layout of trayracer_hip.h); h_matidx: nt object-local material indices or nullptr.
// d_mats: the scene's device material table (for the transparent flag), this object's materials start at mat_base.
// d_interp: where this object's nt InterpRecs go (final location).  Synchronous; returns hipSuccess or the first error
// (err gets a description).
hipError_t build_object_device(const float* h_tris, const int32_t* h_matidx, uint32_t nt, const MatRec* d_mats, uint32_t mat_base,
                               uint32_t n_mats_obj, uint32_t node_base, uint32_t slot_base, uint32_t tri_base, InterpRec* d_interp, hipStream_t stream,
                               DeviceBuildOut* out, std::string* err);

}   // namespace trr
