// launch.h -- kernel I/O descriptors and host-callable launchers (implemented in kernels.hip, used by api.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "device_types.h"

namespace trr {

struct TraceIO {
    const RayRec* shadow_q;
    const RayRec* chain_q;
    const uint32_t* n_shadow;
    const uint32_t* n_chain;
    uint32_t* work_counter;
    HitRec* hits;
    int32_t* occ;
    uint32_t* first_obj;
    uint32_t* first_tri;
    unsigned long long* counters;
    uint2* spill;
    float* term;      // RAYGEN: terminal colour planes (a camera ray that misses is finished by the trace kernel)
    uint8_t* nlev;    // RAYGEN: 0 = miss (done), 0xFF = hit, to be shaded by k_setup
};
struct SetupIO {
    const RayRec* chain_in;
    const uint32_t* n_chain_in;
    const HitRec* hits;
    float* level;
    float* term;
    uint8_t* nlev;
    RayRec* shadow_out;
    uint32_t* n_shadow_out;
    RayRec* chain_out;
    uint32_t* n_chain_out;
};
struct ResolveIO {
    const float* level;
    const float* term;
    float* term_out;   // same planes as term: k_resolve_samples replaces a hit sample's terminal colour by its colour
    const uint8_t* nlev;
    const int32_t* occ;
    uint8_t* strip;
};
int trace_blocks_per_cu(bool raygen, bool count);
size_t trace_spill_bytes(uint32_t grid);
void launch_trace(bool raygen, bool count, uint32_t grid, hipStream_t st, const FrameParams& fp, const BatchParams& bp,
                  const TraceIO& io);
void launch_setup(bool level0, uint32_t grid, hipStream_t st, const FrameParams& fp, const BatchParams& bp, const SetupIO& io,
                  uint32_t lvl);
void launch_resolve(uint32_t grid_samples, uint32_t grid_pixels, hipStream_t st, const FrameParams& fp, const BatchParams& bp,
                    const ResolveIO& io);
void launch_assemble(hipStream_t st, uint32_t width, uint32_t height, uint32_t bw, uint32_t bh, uint32_t nbx, uint32_t nby,
                     uint32_t n_shards, size_t strip_bytes, const uint8_t* strips, uint8_t* frame);


}   // namespace trr
