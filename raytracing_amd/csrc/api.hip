// api.hip -- host side of libtrayracer_hip: context, scene flattening/upload, the per-batch launch
// sequence, and the extern "C" boundary declared in include/trayracer_hip.h.
//
// Mirrors (file:line relative to /root/reference/src):
//   scene upload      <- Scene::clone + Bvh::build                      raytracer/render.rs:60, scene/object.rs:156
//   trr_render*       <- RenderingThread::run block loop                raytracer/render.rs:95-193
//   host math helpers <- Object::transform, Camera::ray (nalgebra)      scene/object.rs:177-182, scene/camera.rs:30-42
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/trayracer_hip.h"
#include "bvh.h"
#include "build_device.h"
#include "device_types.h"
#include "launch.h"

namespace trr {

// ---- small host-side fp32 helpers with nalgebra's operation order ([crate]) -------------
struct H3 {
    float x, y, z;
};
static inline H3 h3(float x, float y, float z) { return H3{x, y, z}; }
static inline H3 hsub(H3 a, H3 b) { return h3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline H3 hcross(H3 a, H3 b) { return h3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
static inline float hdot(H3 a, H3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline float hnorm(H3 a) { return std::sqrt(hdot(a, a)); }
static inline H3 hdiv(H3 a, float s) { return h3(a.x / s, a.y / s, a.z / s); }

thread_local std::string g_tls_error;

}   // namespace trr

using namespace trr;

// ---- the no-allocation entry point's work memory (trr_render_shard_device_noalloc) ---------------------------------------
// One file-scope device array per GPU the library runs on, held until the process ends.  `scratch` is a host-side stand-in
// here, NOT a device address: trr_create asks hipGetSymbolAddress for the array's address on its device (every GPU holds
// its own copy at its own address) and everything below uses that pointer (ctx->scratch), never the name.
static constexpr size_t SCRATCH_BYTES = 256ull << 20;   // 256 MiB of the 288 GB
static __device__ __align__(256) unsigned char scratch[SCRATCH_BYTES];
// Every context and stream of a GPU carves its work set from offset 0 of that GPU's array, so two calls must not overlap on
// the device.  One event per device, created with the device's first context (nothing is allocated at call time): a call
// makes its stream wait for the event before it touches the array and records it after its last batch, so callers on
// different streams or contexts of one GPU queue up behind each other instead of racing.
static std::mutex g_scratch_mu;
static hipEvent_t g_scratch_ev[64] = {};

// ---- opaque handles --------------------------------------------------------------------------------------
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

struct trr_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    int n_cu = 0;
    std::string err;
    std::string desc;
    // work buffers (grown on demand, kept between renders)
    // Batches of blocks are pipelined over MAX_LANES work sets, each with its own stream: the persistent trace kernels
    // of one batch drain (few long rays left) while the next batch's kernels fill the freed CUs.
    struct WorkSet {
        DevBuf hits, level, term, nlev, color, occ, wqueue, gqueue, shadow_q, shadow_deep[2], chain_q[2], counters, spill;
        DevBuf groups;                             // camera-ray cull: the batch's traced 64-sample groups (kernels.hip: k_cover_groups)
        hipStream_t stream = nullptr;
        hipStream_t aux[2] = {nullptr, nullptr};   // split schedule: level 0's shadow rays / the deeper levels' shadow rays
        std::vector<hipEvent_t> sync_ev;           // split schedule: [2 l] level l set up, [2 l + 1] level l's shadow rays traced
        hipEvent_t done = nullptr;
    };
    static constexpr int MAX_LANES = 6;
    WorkSet ws[MAX_LANES];
    hipEvent_t fork = nullptr;
    DevBuf strip, frame, ids_obj, ids_tri;
    // camera-ray cull: the tile mask of a FRAME (kernels.hip: k_cover_boxes), made once on the caller's stream before the
    // frame's batches fork.  Two sets taken in turn, each with an event that its last user records when the frame's work has
    // joined the caller's stream again and that the next user waits for: frames in flight on different streams do not share one.
    struct CoverSet {
        DevBuf mask, big, words;   // words: [0] give-up flag, [1] big rectangles
        hipEvent_t free_ev = nullptr;
    };
    CoverSet cover[2];
    uint32_t cover_turn = 0;
    std::vector<hipEvent_t> events;
    int bpc[2] = {0, 0};                 // resident 256-thread blocks per CU of k_trace<COUNT> (occupancy query, once)
    int bpc_level[2][2] = {{0, 0}, {0, 0}};   // ... of k_trace_level<RAYGEN, COUNT>
    unsigned char* scratch = nullptr;    // this device's copy of the static `scratch` array (hipGetSymbolAddress in trr_create)
    hipEvent_t scratch_ev = nullptr;     // this DEVICE's event that orders the users of that array (shared by its contexts)
    // a multi-device context (trr_create_multi) owns one plain context per device and drives them; it has no streams or
    // work buffers of its own.  subs[0] is the root: it receives the gathered strips and assembles the frame.
    std::vector<trr_ctx*> subs;
    bool duplicate_devices = false;      // (rehearsal on one GPU) a device is listed twice: strips are exchanged by peer copies
    bool rccl_failed = false;            // RCCL reported an error once: this context exchanges strips by peer copies from then on
    std::vector<void*> comms;            // ncclComm_t per device (RCCL, loaded on demand); empty while no gather is needed
    DevBuf gathered;                     // on subs[0]'s device: the strips of all devices
    uint32_t rotate = 0;                 // round-robin start lane of single-batch frames
    uint64_t budget_total = 0;           // bytes the work sets of all lanes may take together (set at the first render)
    size_t events_used = 0;              // events handed to frames whose statistics are still pending
    // pinned read-back chunks (queue counters + kernel counters of one frame each), reused once collected
    struct Pinned {
        void* p = nullptr;
        size_t bytes = 0;
        bool in_use = false;
    };
    std::vector<Pinned> pinned;
    // a frame rendered with TRR_FLAG_DEFER_STATS: everything trr_collect_stats needs to fill *out later
    struct PendingStats {
        trr_stats* out = nullptr;
        size_t ev_first = 0, ev_count = 0;
        std::vector<uint8_t> ev_kind;
        int pinned_slot = -1;
        uint32_t batches = 0, blocks_per_batch = 0, n_owned = 0, bpix = 0, spp = 0, max_depth = 0, blocks_done = 0;
        hipStream_t caller_stream = nullptr;
    };
    std::vector<PendingStats> pending;
};

// HIP's default of 4 hardware queues per process is fewer than the streams a frame uses (up to three lanes, each with two
// more streams in the split schedule): ask for 16 when this library is loaded before the HIP runtime starts and the host has
// not chosen a value itself.
// The value is recorded ONCE, here, before anything of this library can have started the HIP runtime: what the host had set
// when the library was loaded, else the 16 this initialiser sets itself (documented in INTEGRATION.md; a host that wants HIP's
// default exports GPU_MAX_HW_QUEUES=4 before loading the library).  The render path reads the recorded value, never the
// environment again: a change made after HIP has started says nothing about the queues HIP really has.
static const int g_hw_queues = [] {
    if (const char* e = std::getenv("GPU_MAX_HW_QUEUES")) {
        const int v = std::atoi(e);
        return v > 0 ? v : 4;
    }
    (void)::setenv("GPU_MAX_HW_QUEUES", "16", 0);
    return 16;
}();

struct trr_scene {
    trr_ctx* ctx = nullptr;
    int device = 0;                      // ctx->device, kept so that trr_scene_destroy does not touch the context
    std::vector<trr_scene*> replicas;    // scene of a multi-device context: one uploaded copy per device, nothing else
    DevBuf nodes, qnodes, tris, interp, objs, mats, texs, texels, lights, sky;
    DevBuf node_begin;                   // [n_objs + 1] first node of every object (camera-ray cull)
    DevBuf cover_boxes;                  // camera-ray cull: boxes that hold every triangle's own box (launch.h: CoverBox)
    uint32_t n_cover = 0;
    bool cover_ok = false;               // the cover boxes exist (every object's root is an inner node or empty)
    bool quantised = false;              // qnodes holds the 64-B form of every node (kernels.hip: k_quantise_nodes)
    bool any_reflective = true;          // some material is a mirror (illum 3 / 4, material.rs:47-49): reflection rays can exist
    std::vector<ObjRec> h_objs;
    std::vector<uint32_t> h_node_begin;  // host copy of node_begin
    uint32_t n_objs = 0, n_lights = 0, n_mats = 0, n_tex = 0;
    uint64_t n_nodes = 0, n_tris_total = 0, n_tris_live = 0;
    uint32_t bvh_depth = 0;
    uint32_t stack_need = 0;
    uint32_t build = TRR_BUILD_HOST;
    double ms_build = 0.0;
    trr_settings settings{};
    bool have_settings = false;
    uint32_t sky_w = 0, sky_h = 0;
    trr_camera camera{};
    bool have_camera = false;
    float cam_x[3], cam_y[3], cam_z[3], cam_focal = 0.0f;
};

static int fail(trr_ctx* ctx, int code, const std::string& msg) {
    if (ctx) ctx->err = msg;
    g_tls_error = msg;
    return code;
}

#define HIPCHK(ctx, expr)                                                                                     \
    do {                                                                                                      \
        hipError_t e_ = (expr);                                                                               \
        if (e_ != hipSuccess)                                                                                 \
            return fail((ctx), TRR_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));            \
    } while (0)

static int ensure(trr_ctx* ctx, DevBuf& b, size_t bytes) {
    if (b.bytes >= bytes && b.p) return TRR_OK;
    if (b.p) {
        HIPCHK(ctx, hipDeviceSynchronize());
        HIPCHK(ctx, hipFree(b.p));
        b.p = nullptr;
        b.bytes = 0;
    }
    if (bytes == 0) bytes = 256;
    hipError_t e = hipMalloc(&b.p, bytes);
    if (e != hipSuccess) {
        b.p = nullptr;
        return fail(ctx, TRR_ERR_NOMEM, "hipMalloc(" + std::to_string(bytes) + " bytes): " + hipGetErrorString(e));
    }
    b.bytes = bytes;
    return TRR_OK;
}
static void release(DevBuf& b) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.bytes = 0;
}
static int upload(trr_ctx* ctx, DevBuf& b, const void* src, size_t bytes) {
    int rc = ensure(ctx, b, bytes);
    if (rc != TRR_OK) return rc;
    if (bytes) HIPCHK(ctx, hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
    return TRR_OK;
}

static void finish_stats(trr_ctx* ctx, const trr_ctx::PendingStats& f);

// Restores the calling thread's current device when an entry point returns (every entry point selects its context's).
struct DeviceGuard {
    int prev = -1;
    DeviceGuard() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

static bool is_multi(const trr_ctx* ctx) { return ctx && !ctx->subs.empty(); }
static void multi_destroy(trr_ctx* ctx);
static int multi_scene_create(trr_ctx* ctx, const trr_object_desc* objects, uint32_t n_objects, const trr_image_rgb8* textures,
                              uint32_t n_textures, uint32_t build, trr_scene** out);
static int multi_render(trr_ctx* ctx, trr_scene* scene, float delta, uint32_t max_depth, uint32_t flags, uint8_t* rgb_out,
                        const volatile int32_t* cancel_flag, trr_progress_cb progress, void* user, trr_stats* stats_out);
static int multi_render_blocks(trr_ctx* ctx, trr_scene* scene, float delta, uint32_t max_depth, uint32_t flags, uint32_t blocks_per_batch,
                               const volatile int32_t* cancel_flag, trr_block_cb sink, trr_progress_cb progress, void* user, trr_stats* stats_out);
static void add_stats(trr_stats& sum, const trr_stats& a);

// ---- extern "C" ---------------------------------------------------------------------------------------------
extern "C" {

int trr_hw_queues(void) { return g_hw_queues; }

const char* trr_version(void) { return "trayracer_hip 0.1 (gfx950, wavefront BVH4 tracer)"; }

const char* trr_last_error(const trr_ctx* ctx) { return ctx ? ctx->err.c_str() : g_tls_error.c_str(); }

int trr_create(int device_id, trr_ctx** out) {
    if (!out) return fail(nullptr, TRR_ERR_INVALID, "trr_create: out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, TRR_ERR_DEVICE, std::string("no HIP device available: ") + hipGetErrorString(e));
    if (device_id < 0 || device_id >= n) return fail(nullptr, TRR_ERR_INVALID, "trr_create: device_id out of range");
    trr_ctx* ctx = new trr_ctx();
    ctx->device = device_id;
    e = hipSetDevice(device_id);
    if (e != hipSuccess) {
        delete ctx;
        return fail(nullptr, TRR_ERR_DEVICE, std::string("hipSetDevice: ") + hipGetErrorString(e));
    }
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device_id);
    if (e != hipSuccess) {
        delete ctx;
        return fail(nullptr, TRR_ERR_DEVICE, std::string("hipGetDeviceProperties: ") + hipGetErrorString(e));
    }
    ctx->n_cu = prop.multiProcessorCount;
    ctx->desc = std::string(prop.name) + " " + prop.gcnArchName + " CUs=" + std::to_string(prop.multiProcessorCount);
    e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete ctx;
        return fail(nullptr, TRR_ERR_DEVICE, std::string("hipStreamCreate: ") + hipGetErrorString(e));
    }
    for (auto& w : ctx->ws) {
        if (hipStreamCreateWithFlags(&w.stream, hipStreamNonBlocking) != hipSuccess ||
            hipStreamCreateWithFlags(&w.aux[0], hipStreamNonBlocking) != hipSuccess ||
            hipStreamCreateWithFlags(&w.aux[1], hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&w.done, hipEventDisableTiming) != hipSuccess) {
            trr_destroy(ctx);
            return fail(nullptr, TRR_ERR_DEVICE, "hipStreamCreate / hipEventCreate for a work set failed");
        }
    }
    if (hipEventCreateWithFlags(&ctx->fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->cover[0].free_ev, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->cover[1].free_ev, hipEventDisableTiming) != hipSuccess) {
        trr_destroy(ctx);
        return fail(nullptr, TRR_ERR_DEVICE, "hipEventCreate failed");
    }
    // this device's copy of the static scratch array, once per context (= per device), with the device current
    e = hipGetSymbolAddress((void**)&ctx->scratch, HIP_SYMBOL(scratch));
    if (e != hipSuccess || ctx->scratch == nullptr) {
        trr_destroy(ctx);
        return fail(nullptr, TRR_ERR_DEVICE, std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e));
    }
    if (device_id < 64) {
        std::lock_guard<std::mutex> lock(g_scratch_mu);
        if (!g_scratch_ev[device_id] && hipEventCreateWithFlags(&g_scratch_ev[device_id], hipEventDisableTiming) != hipSuccess) {
            g_scratch_ev[device_id] = nullptr;
            trr_destroy(ctx);
            return fail(nullptr, TRR_ERR_DEVICE, "hipEventCreate (scratch order) failed");
        }
        ctx->scratch_ev = g_scratch_ev[device_id];   // kept for the life of the process, like the array
    }
    // resident 256-thread blocks per CU of the two k_trace instantiations (each has its own register count)
    for (int cn = 0; cn < 2; ++cn) ctx->bpc[cn] = std::max(1, std::min(trace_blocks_per_cu(cn != 0), 8));
    for (int rg = 0; rg < 2; ++rg)
        for (int cn = 0; cn < 2; ++cn) ctx->bpc_level[rg][cn] = std::max(1, std::min(level_blocks_per_cu(rg != 0, cn != 0), 8));
    *out = ctx;
    return TRR_OK;
}

void trr_destroy(trr_ctx* ctx) {
    if (!ctx) return;
    DeviceGuard guard;
    if (is_multi(ctx)) { multi_destroy(ctx); return; }
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    for (auto& w : ctx->ws) {
        for (DevBuf* b : {&w.hits, &w.level, &w.term, &w.nlev, &w.color, &w.occ, &w.wqueue, &w.gqueue, &w.shadow_q, &w.shadow_deep[0], &w.shadow_deep[1],
                          &w.chain_q[0], &w.chain_q[1], &w.counters, &w.spill, &w.groups})
            release(*b);
        for (hipEvent_t ev : w.sync_ev) (void)hipEventDestroy(ev);
        if (w.done) (void)hipEventDestroy(w.done);
        for (hipStream_t a : w.aux) if (a) (void)hipStreamDestroy(a);
        if (w.stream) (void)hipStreamDestroy(w.stream);
    }
    if (ctx->fork) (void)hipEventDestroy(ctx->fork);
    for (auto& cs : ctx->cover) {
        for (DevBuf* b : {&cs.mask, &cs.big, &cs.words}) release(*b);
        if (cs.free_ev) (void)hipEventDestroy(cs.free_ev);
    }
    for (DevBuf* b : {&ctx->strip, &ctx->frame, &ctx->ids_obj, &ctx->ids_tri}) release(*b);
    for (hipEvent_t ev : ctx->events) (void)hipEventDestroy(ev);
    for (auto& pn : ctx->pinned)
        if (pn.p) (void)hipHostFree(pn.p);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int trr_collect_stats(trr_ctx* ctx) {
    if (!ctx) return fail(nullptr, TRR_ERR_INVALID, "trr_collect_stats: ctx is NULL");
    DeviceGuard guard;
    if (is_multi(ctx)) {
        int rc = TRR_OK;
        for (trr_ctx* c : ctx->subs) { int r = trr_collect_stats(c); if (r != TRR_OK) { rc = fail(ctx, r, c->err); } }
        return rc;
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc = TRR_OK;
    for (auto& f : ctx->pending) {
        // the frame's last work is on its caller's stream (the join), its read-backs on the lane streams before that
        hipError_t e = hipStreamSynchronize(f.caller_stream);
        for (auto& w : ctx->ws)
            if (e == hipSuccess) e = hipStreamSynchronize(w.stream);
        if (e == hipSuccess) finish_stats(ctx, f);
        else rc = fail(ctx, TRR_ERR_DEVICE, std::string("trr_collect_stats: ") + hipGetErrorString(e));
        ctx->pinned[f.pinned_slot].in_use = false;
    }
    ctx->pending.clear();
    ctx->events_used = 0;
    return rc;
}

int trr_synchronize(trr_ctx* ctx, void* stream) {
    if (!ctx) return fail(nullptr, TRR_ERR_INVALID, "trr_synchronize: ctx is NULL");
    DeviceGuard guard;
    if (is_multi(ctx)) {
        for (trr_ctx* c : ctx->subs) { int r = trr_synchronize(c, nullptr); if (r != TRR_OK) return fail(ctx, r, c->err); }
        return TRR_OK;
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(stream ? (hipStream_t)stream : ctx->stream));
    return TRR_OK;
}

// ---- host math -----------------------------------------------------------------------------------------------
// Object::transform() = Isometry3::from_parts(t, q).to_homogeneous() * Scale3.to_homogeneous() (object.rs:177-182),
// inverse by nalgebra's 4x4 cofactor routine [crate: do_inverse4 == MESA gluInvertMatrix].
int trr_object_matrices(const float translation[3], const float rotation[4], const float scale3[3], float transform[16],
                        float inverse[16]) {
    // [crate] UnitQuaternion::to_rotation_matrix
    const float i = rotation[0], j = rotation[1], k = rotation[2], w = rotation[3];
    const float ww = w * w, ii = i * i, jj = j * j, kk = k * k;
    const float ij = i * j * 2.0f, wk = w * k * 2.0f, wj = w * j * 2.0f, ik = i * k * 2.0f, jk = j * k * 2.0f, wi = w * i * 2.0f;
    float iso[16] = {0}, sc[16] = {0};   // column-major: element (r, c) at [c*4 + r]
    iso[0] = ww + ii - jj - kk; iso[4] = ij - wk;           iso[8] = wj + ik;
    iso[1] = wk + ij;           iso[5] = ww - ii + jj - kk; iso[9] = jk - wi;
    iso[2] = ik - wj;           iso[6] = wi + jk;           iso[10] = ww - ii - jj + kk;
    iso[12] = translation[0]; iso[13] = translation[1]; iso[14] = translation[2]; iso[15] = 1.0f;
    sc[0] = scale3[0]; sc[5] = scale3[1]; sc[10] = scale3[2]; sc[15] = 1.0f;
    // [crate] small static Matrix*Matrix: per output column a gemv accumulating column by column
    float* m = transform;
    for (int c = 0; c < 4; ++c) {
        for (int r = 0; r < 4; ++r) m[c * 4 + r] = iso[0 * 4 + r] * sc[c * 4 + 0];
        for (int q = 1; q < 4; ++q)
            for (int r = 0; r < 4; ++r) m[c * 4 + r] = iso[q * 4 + r] * sc[c * 4 + q] + m[c * 4 + r];
    }
    float o[16];
    o[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] -
           m[13] * m[7] * m[10];
    o[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] +
           m[12] * m[7] * m[10];
    o[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] -
           m[12] * m[7] * m[9];
    o[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] +
            m[12] * m[6] * m[9];
    o[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] +
           m[13] * m[3] * m[10];
    o[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] -
           m[12] * m[3] * m[10];
    o[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] +
           m[12] * m[3] * m[9];
    o[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] -
            m[12] * m[2] * m[9];
    o[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] -
           m[13] * m[3] * m[6];
    o[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] +
           m[12] * m[3] * m[6];
    o[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] -
            m[12] * m[3] * m[5];
    o[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] +
            m[12] * m[2] * m[5];
    o[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] +
           m[9] * m[3] * m[6];
    o[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] -
           m[8] * m[3] * m[6];
    o[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] +
            m[8] * m[3] * m[5];
    o[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] -
            m[8] * m[2] * m[5];
    float det = m[0] * o[0] + m[1] * o[4] + m[2] * o[8] + m[3] * o[12];
    if (det == 0.0f) {
        std::memcpy(inverse, transform, 16 * sizeof(float));
        return 0;
    }
    float inv_det = 1.0f / det;
    for (int q = 0; q < 16; ++q) inverse[q] = o[q] * inv_det;
    return 1;
}

// camera basis of Rotation3::look_at_rh(look_at - position, up) [crate] (camera.rs:35)
static void camera_basis(const trr_camera* c, float x[3], float y[3], float z[3], float* focal) {
    H3 dir = hsub(h3(c->look_at[0], c->look_at[1], c->look_at[2]), h3(c->position[0], c->position[1], c->position[2]));
    H3 nd = h3(-dir.x, -dir.y, -dir.z);
    H3 zax = hdiv(nd, hnorm(nd));
    H3 xc = hcross(h3(c->up[0], c->up[1], c->up[2]), zax);
    H3 xax = hdiv(xc, hnorm(xc));
    H3 yc = hcross(zax, xax);
    H3 yax = hdiv(yc, hnorm(yc));
    x[0] = xax.x; x[1] = xax.y; x[2] = xax.z;
    y[0] = yax.x; y[1] = yax.y; y[2] = yax.z;
    z[0] = zax.x; z[1] = zax.y; z[2] = zax.z;
    *focal = -1.0f / std::tan(c->fov / 2.0f);   // camera.rs:32
}

void trr_camera_ray(const trr_camera* camera, float x, float y, float origin[3], float direction[3]) {
    float bx[3], by[3], bz[3], focal;
    camera_basis(camera, bx, by, bz, &focal);
    H3 dc = h3(x, -y, focal);
    H3 d = h3(hdot(h3(bx[0], by[0], bz[0]), dc), hdot(h3(bx[1], by[1], bz[1]), dc), hdot(h3(bx[2], by[2], bz[2]), dc));
    d = hdiv(d, hnorm(d));
    origin[0] = camera->position[0]; origin[1] = camera->position[1]; origin[2] = camera->position[2];
    direction[0] = d.x; direction[1] = d.y; direction[2] = d.z;
}

// ---- scene ----------------------------------------------------------------------------------------------------
static void fill_obj_matrices(ObjRec& o, const float* fwd_cm, const float* inv_cm) {
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 4; ++c) {
            o.fwd[r * 4 + c] = fwd_cm[c * 4 + r];
            o.inv[r * 4 + c] = inv_cm[c * 4 + r];
        }
}

// TRR_BUILD_DEVICE: every object's tree is built on the GPU (build_device.hip) into arrays of its own, which are then
// packed behind one another.  sc->mats is already on the device; fills sc->nodes / tris / interp / h_objs and the counts.
// The 64-B form of the scene's nodes, made on the device from the exact ones (host- or device-built alike), for the level
// kernels.  Opt-in (TRR_NODES=quantised): same image bit for bit, four 16-B loads per lane and node step instead of seven,
// and no faster -- 31.2 against 31.3 ms on the 4K x 16 spp frame, 1.7 % more node visits (profiles/r02o_qnodes*.txt;
// DESIGN.md section 4).  A scene with a bound that is not finite or beyond 2^20 keeps to its exact nodes regardless.
static int quantise_scene_nodes(trr_ctx* ctx, trr_scene* sc) {
    sc->quantised = false;
    if (sc->n_nodes == 0) return TRR_OK;
    const char* mode = std::getenv("TRR_NODES");
    if (!mode || mode[0] != 'q') return TRR_OK;
    int rc;
    DevBuf flag;
    if ((rc = ensure(ctx, sc->qnodes, (size_t)sc->n_nodes * sizeof(QNode4))) != TRR_OK || (rc = ensure(ctx, flag, 256)) != TRR_OK) {
        release(flag);
        return rc;
    }
    uint32_t bad = 1;
    hipError_t e = hipMemsetAsync(flag.p, 0, 4, ctx->stream);
    if (e == hipSuccess) {
        launch_quantise_nodes(ctx->stream, (const Node4*)sc->nodes.p, (QNode4*)sc->qnodes.p, (uint32_t)sc->n_nodes, (uint32_t*)flag.p);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&bad, flag.p, 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    release(flag);
    if (e != hipSuccess) return fail(ctx, TRR_ERR_DEVICE, std::string("quantising the BVH nodes: ") + hipGetErrorString(e));
    sc->quantised = bad == 0;
    if (!sc->quantised) release(sc->qnodes);
    return TRR_OK;
}

static int build_scene_device(trr_ctx* ctx, trr_scene* sc, const trr_object_desc* objects, uint32_t n_objects,
                              const std::vector<uint32_t>& mat_bases) {
    std::vector<DeviceBuildOut> parts(n_objects);
    auto free_parts = [&]() {
        for (DeviceBuildOut& p : parts) {
            if (p.nodes) (void)hipFree(p.nodes);
            if (p.tris) (void)hipFree(p.tris);
            p.nodes = nullptr;
            p.tris = nullptr;
        }
    };
    int rc = ensure(ctx, sc->interp, (size_t)sc->n_tris_total * sizeof(InterpRec));
    if (rc != TRR_OK) return rc;
    uint64_t node_base = 0, slot_base = 0, interp_base = 0;
    uint32_t depth_max = 0, stack_max = 0;
    sc->h_objs.assign(n_objects, ObjRec{});
    sc->h_node_begin.clear();
    for (uint32_t oi = 0; oi < n_objects; ++oi) {
        const trr_object_desc& d = objects[oi];
        // leaf references hold the first triangle slot in 29 bits (see trr_scene_create_ex)
        if (slot_base + d.n_triangles >= (1ull << 29) - 1 || node_base + d.n_triangles >= (1ull << 31)) {
            free_parts();
            return fail(ctx, TRR_ERR_INVALID, "trr_scene_create: more than 2^29 - 2 triangles in one scene");
        }
        std::string msg;
        hipError_t e = build_object_device(d.triangles, d.material_index, d.n_triangles, static_cast<const MatRec*>(sc->mats.p),
                                           mat_bases[oi], d.n_materials, (uint32_t)node_base, (uint32_t)slot_base, (uint32_t)interp_base,
                                           static_cast<InterpRec*>(sc->interp.p) + interp_base, ctx->stream, &parts[oi], &msg);
        if (e != hipSuccess) {
            free_parts();
            return fail(ctx, e == hipErrorOutOfMemory ? TRR_ERR_NOMEM : TRR_ERR_DEVICE, "device BVH build of object " + std::to_string(oi) + ": " + msg);
        }
        sc->h_node_begin.push_back((uint32_t)node_base);
        ObjRec& o = sc->h_objs[oi];
        fill_obj_matrices(o, d.transform, d.inverse);
        o.root = parts[oi].root_ref;
        o.tri_base = (uint32_t)interp_base;
        o.n_tris = d.n_triangles;
        o.flags = (d.n_triangles == 1) ? 1u : 0u;   // [crate] a root Leaf is returned by traverse() without a box test
        o.slack_abs = 1.0e-4f * parts[oi].maxabs;
        depth_max = std::max(depth_max, parts[oi].depth);
        stack_max = std::max(stack_max, parts[oi].stack_need);
        node_base += parts[oi].n_nodes;
        slot_base += parts[oi].n_live;
        interp_base += d.n_triangles;
    }
    if ((rc = ensure(ctx, sc->nodes, (size_t)node_base * sizeof(Node4))) != TRR_OK ||
        (rc = ensure(ctx, sc->tris, (size_t)slot_base * sizeof(TriRec))) != TRR_OK) {
        free_parts();
        return rc;
    }
    uint64_t nb = 0, sb = 0;
    hipError_t e = hipSuccess;
    for (uint32_t oi = 0; oi < n_objects && e == hipSuccess; ++oi) {
        const DeviceBuildOut& p = parts[oi];
        if (p.n_nodes)
            e = hipMemcpyAsync(static_cast<Node4*>(sc->nodes.p) + nb, p.nodes, (size_t)p.n_nodes * sizeof(Node4), hipMemcpyDeviceToDevice, ctx->stream);
        if (e == hipSuccess && p.n_live)
            e = hipMemcpyAsync(static_cast<TriRec*>(sc->tris.p) + sb, p.tris, (size_t)p.n_live * sizeof(TriRec), hipMemcpyDeviceToDevice, ctx->stream);
        nb += p.n_nodes;
        sb += p.n_live;
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    free_parts();
    if (std::getenv("TRR_DEBUG")) std::fprintf(stderr, "[trr build] packed %llu nodes, %llu triangles\n", (unsigned long long)node_base, (unsigned long long)slot_base);
    if (e != hipSuccess) return fail(ctx, TRR_ERR_DEVICE, std::string("device BVH build: packing the objects: ") + hipGetErrorString(e));
    sc->n_nodes = node_base;
    sc->h_node_begin.push_back((uint32_t)node_base);
    sc->n_tris_live = slot_base;
    sc->bvh_depth = depth_max;
    sc->stack_need = stack_max;
    return TRR_OK;
}

// Camera-ray cull: the scene's cover boxes (launch.h), made on the device from the packed node array of either builder.  Top
// down from every object's root, one tree level per round: leaves met on the way become boxes, inner nodes go on to the next
// round, until the next level would exceed the budget -- then every child box of the level's nodes is taken.  A few tiny
// launches and two words read back per round, once per scene.
static int build_cover_boxes(trr_ctx* ctx, trr_scene* sc) {
    std::vector<uint32_t> frontier;
    for (const ObjRec& o : sc->h_objs)
        if (o.root != REF_EMPTY) frontier.push_back(o.root);   // (cover_ok: no root is a leaf)
    sc->n_cover = 0;
    int rc = ensure(ctx, sc->cover_boxes, (size_t)COVER_BOX_BUDGET * sizeof(CoverBox));
    if (rc != TRR_OK || frontier.empty()) return rc;
    DevBuf fr[2], cnt;
    auto done = [&](int code) { release(fr[0]); release(fr[1]); release(cnt); return code; };
    if ((rc = ensure(ctx, fr[0], (size_t)COVER_BOX_BUDGET * 4)) || (rc = ensure(ctx, fr[1], (size_t)COVER_BOX_BUDGET * 4)) || (rc = ensure(ctx, cnt, 256)))
        return done(rc);
    if (frontier.size() * 4 > COVER_BOX_BUDGET) { sc->cover_ok = false; return done(TRR_OK); }   // (thousands of objects: no cull)
    hipError_t e = hipMemcpy(fr[0].p, frontier.data(), frontier.size() * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(cnt.p, 0, 256);
    uint32_t n_in = (uint32_t)frontier.size(), n_boxes = 0;
    uint32_t* d_cnt = static_cast<uint32_t*>(cnt.p);   // [0] boxes so far, [1 + round % 2] next frontier
    for (uint32_t round = 0; e == hipSuccess && n_in != 0; ++round) {
        // the level below this one holds at most 4 n_in nodes with 16 n_in child boxes
        const bool last = (uint64_t)n_boxes + 16ull * n_in > COVER_BOX_BUDGET || 4ull * n_in > COVER_BOX_BUDGET;
        uint32_t* n_out = d_cnt + 1 + (round & 1u);
        e = hipMemsetAsync(n_out, 0, 4, ctx->stream);
        if (e != hipSuccess) break;
        launch_cover_expand(ctx->stream, static_cast<const Node4*>(sc->nodes.p), static_cast<const uint32_t*>(sc->node_begin.p), sc->n_objs,
                            static_cast<const uint32_t*>(fr[round & 1u].p), n_in, static_cast<uint32_t*>(fr[(round + 1) & 1u].p), n_out,
                            static_cast<CoverBox*>(sc->cover_boxes.p), d_cnt, COVER_BOX_BUDGET, last);
        uint32_t back[3] = {0, 0, 0};
        e = hipMemcpyAsync(back, d_cnt, 12, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        n_boxes = back[0];
        n_in = last ? 0u : back[1 + (round & 1u)];
    }
    if (e != hipSuccess) return done(fail(ctx, TRR_ERR_DEVICE, std::string("cover boxes: ") + hipGetErrorString(e)));
    if (n_boxes > COVER_BOX_BUDGET) sc->cover_ok = false;   // cannot happen (the rounds stop before the budget); no cull rather than a partial list
    sc->n_cover = std::min(n_boxes, COVER_BOX_BUDGET);
    if (std::getenv("TRR_DEBUG")) std::fprintf(stderr, "[trr] camera-ray cull: %u cover boxes\n", sc->n_cover);
    return done(TRR_OK);
}

int trr_scene_create(trr_ctx* ctx, const trr_object_desc* objects, uint32_t n_objects, const trr_image_rgb8* textures,
                     uint32_t n_textures, trr_scene** out) {
    return trr_scene_create_ex(ctx, objects, n_objects, textures, n_textures, TRR_BUILD_HOST, out);
}

int trr_scene_create_ex(trr_ctx* ctx, const trr_object_desc* objects, uint32_t n_objects, const trr_image_rgb8* textures,
                        uint32_t n_textures, uint32_t build, trr_scene** out) {
    if (!ctx || !out) return fail(ctx, TRR_ERR_INVALID, "trr_scene_create: NULL argument");
    *out = nullptr;
    if (n_objects && !objects) return fail(ctx, TRR_ERR_INVALID, "trr_scene_create: objects is NULL");
    if (n_textures && !textures) return fail(ctx, TRR_ERR_INVALID, "trr_scene_create: textures is NULL");
    if (build != TRR_BUILD_HOST && build != TRR_BUILD_DEVICE) return fail(ctx, TRR_ERR_INVALID, "trr_scene_create_ex: unknown build mode");
    DeviceGuard guard;
    if (is_multi(ctx)) return multi_scene_create(ctx, objects, n_objects, textures, n_textures, build, out);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const auto t_start = std::chrono::steady_clock::now();

    std::vector<Node4> nodes;
    std::vector<TriRec> tris;
    std::vector<InterpRec> interp;
    std::vector<ObjRec> objs(n_objects);
    std::vector<MatRec> mats;
    uint32_t depth_max = 0, stack_max = 0;
    uint64_t n_total = 0;
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());

    // material table of all objects (object-local indices become global ones: mat_bases[object] + index)
    std::vector<uint32_t> mat_bases(n_objects, 0);
    for (uint32_t oi = 0; oi < n_objects; ++oi) {
        const trr_object_desc& d = objects[oi];
        if (d.n_triangles && !d.triangles) return fail(ctx, TRR_ERR_INVALID, "trr_scene_create: object without triangles pointer");
        if (d.n_materials && !d.materials) return fail(ctx, TRR_ERR_INVALID, "trr_scene_create: object without materials pointer");
        mat_bases[oi] = (uint32_t)mats.size();
        for (uint32_t m = 0; m < d.n_materials; ++m) {
            const trr_material_desc& s = d.materials[m];
            MatRec r{};
            for (int k = 0; k < 3; ++k) { r.kd[k] = s.kd[k]; r.ks[k] = s.ks[k]; }
            r.ns = s.ns; r.dissolve = s.dissolve; r.illum = s.illum;
            r.tex = (s.texture >= 0 && (uint32_t)s.texture < n_textures) ? s.texture : -1;
            r.flags = s.flags;
            mats.push_back(r);
        }
        n_total += d.n_triangles;
    }

    std::vector<uint32_t> node_begin_host;   // first node of every object (camera-ray cull)
    for (uint32_t oi = 0; build == TRR_BUILD_HOST && oi < n_objects; ++oi) {
        const trr_object_desc& d = objects[oi];
        const uint32_t nt = d.n_triangles;
        const uint32_t mat_base = mat_bases[oi];
        // per-triangle: normal/len (triangle.rs:54-56), own box (triangle.rs:88-95), interpolation record
        std::vector<TriRec> recs(nt);
        std::vector<Box3> boxes(nt);
        std::vector<uint8_t> live(nt);
        const size_t interp_base = interp.size();
        interp.resize(interp_base + nt);
        float maxabs = 0.0f;
        auto prep = [&](uint32_t lo, uint32_t hi, float* maxabs_out) {
            float ma = 0.0f;
            for (uint32_t t = lo; t < hi; ++t) {
                const float* f = d.triangles + (size_t)t * 24;
                H3 a = h3(f[0], f[1], f[2]), b = h3(f[3], f[4], f[5]), c = h3(f[6], f[7], f[8]);
                H3 cr = hcross(hsub(b, a), hsub(c, a));
                float len = hnorm(cr);
                H3 n = hdiv(cr, len);
                TriRec& r = recs[t];
                r.ax = a.x; r.ay = a.y; r.az = a.z; r.bx = b.x; r.by = b.y; r.bz = b.z; r.cx = c.x; r.cy = c.y; r.cz = c.z;
                r.nx = n.x; r.ny = n.y; r.nz = n.z;
                r.tri_id = (uint32_t)interp_base + t;
                int32_t mi = d.material_index ? d.material_index[t] : -1;
                r.mat = (mi >= 0 && (uint32_t)mi < d.n_materials) ? (int32_t)(mat_base + (uint32_t)mi) : -1;
                r.len = len;
                r.flags = 0;
                if (r.mat >= 0) {
                    int il = mats[r.mat].illum;
                    if (il == 6 || il == 7) r.flags |= 1u;   // material.rs:51-53
                    if (il == 3 || il == 4) r.flags |= 2u;   // material.rs:47-49
                }
                // a triangle with |ab x ac| == 0 can never be hit for any delta >= 0 (triangle.rs:56): left out of the tree
                live[t] = !(len <= 0.0f);
                H3 c2 = h3(c.x + 0.0001f, c.y + 0.0001f, c.z + 0.0001f);
                Box3& bx = boxes[t];
                bx.lo[0] = std::fmin(std::fmin(a.x, b.x), c2.x); bx.lo[1] = std::fmin(std::fmin(a.y, b.y), c2.y);
                bx.lo[2] = std::fmin(std::fmin(a.z, b.z), c2.z);
                bx.hi[0] = std::fmax(std::fmax(a.x, b.x), c2.x); bx.hi[1] = std::fmax(std::fmax(a.y, b.y), c2.y);
                bx.hi[2] = std::fmax(std::fmax(a.z, b.z), c2.z);
                for (int k = 0; k < 9; ++k) {
                    float v = std::fabs(f[k]);
                    if (std::isfinite(v) && v > ma) ma = v;
                }
                InterpRec& ip = interp[interp_base + t];
                for (int k = 0; k < 3; ++k) { ip.na[k] = f[9 + k]; ip.nb[k] = f[12 + k]; ip.nc[k] = f[15 + k]; }
                ip.uva[0] = f[18]; ip.uva[1] = f[19]; ip.uvb[0] = f[20]; ip.uvb[1] = f[21]; ip.uvc[0] = f[22]; ip.uvc[1] = f[23];
                ip.pad = 0.0f;
            }
            *maxabs_out = ma;
        };
        {
            unsigned nth = nt > (1u << 16) ? std::min(hw, 32u) : 1u;
            std::vector<float> mas(nth, 0.0f);
            std::vector<std::thread> th;
            for (unsigned k = 0; k < nth; ++k) {
                uint32_t lo = (uint32_t)((uint64_t)nt * k / nth), hi = (uint32_t)((uint64_t)nt * (k + 1) / nth);
                if (nth == 1) prep(lo, hi, &mas[k]);
                else th.emplace_back(prep, lo, hi, &mas[k]);
            }
            for (auto& t : th) t.join();
            for (float v : mas) maxabs = std::max(maxabs, v);
        }
        std::vector<Box3> live_boxes;
        std::vector<uint32_t> live_ids;
        live_boxes.reserve(nt);
        live_ids.reserve(nt);
        for (uint32_t t = 0; t < nt; ++t)
            if (live[t]) { live_boxes.push_back(boxes[t]); live_ids.push_back(t); }
        BuildResult br;
        build_bvh4(live_boxes.data(), (uint32_t)live_boxes.size(), 0, &br);
        const uint32_t node_base = (uint32_t)nodes.size();
        node_begin_host.push_back(node_base);
        const uint32_t slot_base = (uint32_t)tris.size();
        if ((uint64_t)slot_base + live_boxes.size() >= (1ull << 29))
            return fail(ctx, TRR_ERR_INVALID, "trr_scene_create: more than 2^29 triangles");
        auto fix = [&](uint32_t ref) -> uint32_t {
            if (ref == REF_EMPTY) return ref;
            if (ref & REF_LEAF) {
                uint32_t first = (ref & 0x7FFFFFFFu) >> 2, cnt = (ref & 3u) + 1u;
                return make_leaf_ref(first + slot_base, cnt);
            }
            return ref + node_base;
        };
        for (Node4 nd : br.nodes) {
            for (int k = 0; k < 4; ++k) nd.child[k] = fix(nd.child[k]);
            nodes.push_back(nd);
        }
        for (uint32_t s = 0; s < br.order.size(); ++s) tris.push_back(recs[live_ids[br.order[s]]]);
        ObjRec& o = objs[oi];
        std::memset(&o, 0, sizeof o);
        fill_obj_matrices(o, d.transform, d.inverse);
        o.root = fix(br.root_ref);
        o.tri_base = (uint32_t)interp_base;
        o.n_tris = nt;
        o.flags = (nt == 1) ? 1u : 0u;   // [crate] a root Leaf is returned by traverse() without a box test
        o.slack_abs = 1.0e-4f * maxabs;
        depth_max = std::max(depth_max, br.max_depth);
        stack_max = std::max(stack_max, br.stack_need);
    }
    // leaf references hold the first triangle slot in 29 bits, and the two highest 32-bit values are REF_EMPTY and the
    // trace kernel's "idle lane" marker; node indices must stay below REF_LEAF
    if (tris.size() >= (1ull << 29) - 1 || nodes.size() >= (1ull << 31))
        return fail(ctx, TRR_ERR_INVALID, "trr_scene_create: more than 2^29 - 2 triangles in one scene");

    trr_scene* sc = new trr_scene();
    sc->ctx = ctx;
    sc->device = ctx->device;
    sc->n_objs = n_objects;
    sc->h_objs = objs;
    node_begin_host.push_back((uint32_t)nodes.size());
    sc->h_node_begin = node_begin_host;
    sc->n_nodes = nodes.size();
    sc->n_tris_total = n_total;
    sc->n_tris_live = tris.size();
    sc->bvh_depth = depth_max;
    sc->stack_need = stack_max;
    sc->n_mats = (uint32_t)mats.size();
    sc->any_reflective = false;
    for (const MatRec& m : mats) sc->any_reflective = sc->any_reflective || m.illum == 3 || m.illum == 4;
    sc->n_tex = n_textures;
    // texture atlas
    std::vector<TexRec> texs(n_textures);
    std::vector<uint8_t> texels;
    for (uint32_t t = 0; t < n_textures; ++t) {
        if (!textures[t].rgb || textures[t].width == 0 || textures[t].height == 0) {
            delete sc;
            return fail(ctx, TRR_ERR_INVALID, "trr_scene_create: empty texture");
        }
        texs[t].offset = texels.size();
        texs[t].w = textures[t].width;
        texs[t].h = textures[t].height;
        size_t nb = (size_t)textures[t].width * textures[t].height * 3;
        texels.insert(texels.end(), textures[t].rgb, textures[t].rgb + nb);
    }
    int rc = TRR_OK;
    if ((rc = upload(ctx, sc->mats, mats.data(), mats.size() * sizeof(MatRec))) != TRR_OK ||
        (rc = upload(ctx, sc->texs, texs.data(), texs.size() * sizeof(TexRec))) != TRR_OK ||
        (rc = upload(ctx, sc->texels, texels.data(), texels.size())) != TRR_OK) {
        trr_scene_destroy(sc);
        return rc;
    }
    if (build == TRR_BUILD_DEVICE) {
        rc = build_scene_device(ctx, sc, objects, n_objects, mat_bases);
    } else {
        if ((rc = upload(ctx, sc->nodes, nodes.data(), nodes.size() * sizeof(Node4))) == TRR_OK &&
            (rc = upload(ctx, sc->tris, tris.data(), tris.size() * sizeof(TriRec))) == TRR_OK)
            rc = upload(ctx, sc->interp, interp.data(), interp.size() * sizeof(InterpRec));
    }
    if (rc == TRR_OK && sc->stack_need > (uint32_t)TRACE_STACK_ENTRIES)
        rc = fail(ctx, TRR_ERR_INVALID, "trr_scene_create: the BVH can need a traversal stack of " + std::to_string(sc->stack_need) +
                                            " entries, the trace kernel holds " + std::to_string(TRACE_STACK_ENTRIES) +
                                            (build == TRR_BUILD_DEVICE ? " (build this scene with TRR_BUILD_HOST)" : ""));
    if (rc == TRR_OK) rc = upload(ctx, sc->objs, sc->h_objs.data(), sc->h_objs.size() * sizeof(ObjRec));
    if (rc == TRR_OK) {
        // camera-ray cull: usable when every leaf box of the scene is a child box of some node, i.e. no object's root is a leaf
        // (an object of <= 4 triangles; of ONE triangle, the crate even skips the box test)
        sc->cover_ok = n_objects != 0 && sc->h_node_begin.size() == (size_t)n_objects + 1;
        for (const ObjRec& o : sc->h_objs) sc->cover_ok = sc->cover_ok && (o.root == REF_EMPTY || !(o.root & REF_LEAF)) && !(o.flags & 1u);
        if (sc->cover_ok) rc = upload(ctx, sc->node_begin, sc->h_node_begin.data(), sc->h_node_begin.size() * sizeof(uint32_t));
        if (rc == TRR_OK && sc->cover_ok) rc = build_cover_boxes(ctx, sc);
    }
    if (rc == TRR_OK) rc = quantise_scene_nodes(ctx, sc);
    if (rc != TRR_OK) {
        trr_scene_destroy(sc);
        return rc;
    }
    sc->build = build;
    sc->ms_build = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count();
    *out = sc;
    return TRR_OK;
}

int trr_scene_get_info(const trr_scene* sc, trr_scene_info* out) {
    if (!sc || !out) return fail(sc ? sc->ctx : nullptr, TRR_ERR_INVALID, "trr_scene_get_info: NULL argument");
    if (!sc->replicas.empty()) return trr_scene_get_info(sc->replicas[0], out);
    out->n_triangles = sc->n_tris_total;
    out->n_triangles_live = sc->n_tris_live;
    out->n_nodes = sc->n_nodes;
    out->bvh_depth = sc->bvh_depth;
    out->stack_need = sc->stack_need;
    out->build = sc->build;
    out->ms_build = sc->ms_build;
    return TRR_OK;
}

int trr_scene_validate(trr_scene* sc) {
    if (!sc) return fail(nullptr, TRR_ERR_INVALID, "trr_scene_validate: NULL argument");
    DeviceGuard guard;
    if (!sc->replicas.empty()) {
        for (trr_scene* r : sc->replicas) { int rc = trr_scene_validate(r); if (rc != TRR_OK) return fail(sc->ctx, rc, r->ctx->err); }
        return TRR_OK;
    }
    trr_ctx* ctx = sc->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    std::vector<Node4> nodes(sc->n_nodes);
    std::vector<TriRec> tris(sc->n_tris_live);
    if (!nodes.empty()) HIPCHK(ctx, hipMemcpy(nodes.data(), sc->nodes.p, nodes.size() * sizeof(Node4), hipMemcpyDeviceToHost));
    if (!tris.empty()) HIPCHK(ctx, hipMemcpy(tris.data(), sc->tris.p, tris.size() * sizeof(TriRec), hipMemcpyDeviceToHost));
    std::vector<uint8_t> node_seen(nodes.size(), 0), slot_seen(tris.size(), 0);
    auto bad = [&](uint32_t oi, const std::string& what) {
        return fail(ctx, TRR_ERR_INVALID, "trr_scene_validate: object " + std::to_string(oi) + ": " + what);
    };
    // a >= b unless either is NaN (NaN coordinates make no promise in either builder)
    auto ge = [](float a, float b) { return !(a < b); };
    uint32_t depth_seen = 0;
    for (uint32_t oi = 0; oi < sc->n_objs; ++oi) {
        const ObjRec& o = sc->h_objs[oi];
        std::vector<uint8_t> tri_seen(o.n_tris, 0);
        struct Item { uint32_t ref; Box3 box; bool has_box; uint32_t depth; };
        std::vector<Item> stack;
        stack.push_back(Item{o.root, Box3{}, false, 1});
        while (!stack.empty()) {
            Item it = stack.back();
            stack.pop_back();
            if (it.ref == REF_EMPTY) continue;
            if (it.ref & REF_LEAF) {
                uint32_t first = (it.ref & 0x7FFFFFFFu) >> 2, cnt = (it.ref & 3u) + 1u;
                if ((uint64_t)first + cnt > tris.size()) return bad(oi, "leaf slots out of range");
                for (uint32_t s = first; s < first + cnt; ++s) {
                    if (slot_seen[s]++) return bad(oi, "triangle slot " + std::to_string(s) + " is in two leaves");
                    const TriRec& r = tris[s];
                    const uint32_t tid = r.tri_id - o.tri_base;   // the record holds tri_base + row
                    if (r.tri_id < o.tri_base || tid >= o.n_tris) return bad(oi, "triangle id out of range in slot " + std::to_string(s));
                    if (tri_seen[tid]++) return bad(oi, "triangle " + std::to_string(tid) + " is in two slots");
                    if (!it.has_box) continue;
                    const float a[3] = {r.ax, r.ay, r.az}, b[3] = {r.bx, r.by, r.bz}, c[3] = {r.cx, r.cy, r.cz};
                    for (int k = 0; k < 3; ++k) {   // triangle.rs:88-95
                        float c2 = c[k] + 0.0001f;
                        float lo = std::fmin(std::fmin(a[k], b[k]), c2), hi = std::fmax(std::fmax(a[k], b[k]), c2);
                        if (!ge(lo, it.box.lo[k]) || !ge(it.box.hi[k], hi))
                            return bad(oi, "leaf box does not contain triangle " + std::to_string(tid));
                    }
                }
                continue;
            }
            if (it.ref >= nodes.size()) return bad(oi, "node reference out of range");
            if (node_seen[it.ref]++) return bad(oi, "node " + std::to_string(it.ref) + " has two parents");
            depth_seen = std::max(depth_seen, it.depth);
            if (it.depth > 4096) return bad(oi, "tree deeper than 4096 levels");
            const Node4& nd = nodes[it.ref];
            for (int k = 0; k < 4; ++k) {
                if (nd.child[k] == REF_EMPTY) continue;
                Box3 bx{{nd.lox[k], nd.loy[k], nd.loz[k]}, {nd.hix[k], nd.hiy[k], nd.hiz[k]}};
                if (it.has_box)
                    for (int a = 0; a < 3; ++a)
                        if (!ge(bx.lo[a], it.box.lo[a]) || !ge(it.box.hi[a], bx.hi[a]))
                            return bad(oi, "child box of node " + std::to_string(it.ref) + " sticks out of its parent's");
                stack.push_back(Item{nd.child[k], bx, true, it.depth + 1});
            }
        }
    }
    for (size_t s = 0; s < slot_seen.size(); ++s)
        if (!slot_seen[s]) return fail(ctx, TRR_ERR_INVALID, "trr_scene_validate: triangle slot " + std::to_string(s) + " is in no leaf");
    for (size_t i = 0; i < node_seen.size(); ++i)
        if (!node_seen[i]) return fail(ctx, TRR_ERR_INVALID, "trr_scene_validate: node " + std::to_string(i) + " is unreachable");
    if (depth_seen > sc->bvh_depth)
        return fail(ctx, TRR_ERR_INVALID, "trr_scene_validate: tree depth " + std::to_string(depth_seen) + " exceeds the recorded " +
                                              std::to_string(sc->bvh_depth));
    return TRR_OK;
}

void trr_scene_destroy(trr_scene* sc) {
    if (!sc) return;
    DeviceGuard guard;
    for (trr_scene* r : sc->replicas) trr_scene_destroy(r);
    sc->replicas.clear();
    // the device id is the scene's own copy: a scene may be destroyed after its context (the buffers are plain hipMalloc
    // memory of the device; destroying scenes first is still the documented order)
    (void)hipSetDevice(sc->device);
    (void)hipDeviceSynchronize();
    for (DevBuf* b : {&sc->nodes, &sc->qnodes, &sc->tris, &sc->interp, &sc->objs, &sc->mats, &sc->texs, &sc->texels, &sc->lights, &sc->sky, &sc->node_begin, &sc->cover_boxes})
        release(*b);
    delete sc;
}

int trr_scene_set_instances(trr_scene* sc, const float* transforms, const float* inverses, uint32_t n_objects) {
    if (!sc || !transforms || !inverses) return fail(sc ? sc->ctx : nullptr, TRR_ERR_INVALID, "trr_scene_set_instances: NULL argument");
    DeviceGuard guard;
    if (!sc->replicas.empty()) {
        for (trr_scene* r : sc->replicas) { int rc = trr_scene_set_instances(r, transforms, inverses, n_objects); if (rc != TRR_OK) return fail(sc->ctx, rc, r->ctx->err); }
        return TRR_OK;
    }
    if (n_objects != sc->n_objs) return fail(sc->ctx, TRR_ERR_INVALID, "trr_scene_set_instances: object count mismatch");
    HIPCHK(sc->ctx, hipSetDevice(sc->ctx->device));
    HIPCHK(sc->ctx, hipDeviceSynchronize());
    for (uint32_t i = 0; i < n_objects; ++i) fill_obj_matrices(sc->h_objs[i], transforms + 16 * i, inverses + 16 * i);
    return upload(sc->ctx, sc->objs, sc->h_objs.data(), sc->h_objs.size() * sizeof(ObjRec));
}

int trr_scene_set_lights(trr_scene* sc, const trr_light* lights, uint32_t n_lights) {
    if (!sc || (n_lights && !lights)) return fail(sc ? sc->ctx : nullptr, TRR_ERR_INVALID, "trr_scene_set_lights: NULL argument");
    if (n_lights > 0xFFFFu) return fail(sc->ctx, TRR_ERR_INVALID, "trr_scene_set_lights: more than 65535 lights");
    DeviceGuard guard;
    if (!sc->replicas.empty()) {
        for (trr_scene* r : sc->replicas) { int rc = trr_scene_set_lights(r, lights, n_lights); if (rc != TRR_OK) return fail(sc->ctx, rc, r->ctx->err); }
        sc->n_lights = n_lights;
        return TRR_OK;
    }
    HIPCHK(sc->ctx, hipSetDevice(sc->ctx->device));
    HIPCHK(sc->ctx, hipDeviceSynchronize());
    std::vector<LightRec> l(n_lights);
    for (uint32_t i = 0; i < n_lights; ++i) {
        for (int k = 0; k < 3; ++k) { l[i].pos[k] = lights[i].position[k]; l[i].color[k] = lights[i].color[k]; }
        l[i].intensity = lights[i].intensity;
        l[i].pad = 0.0f;
    }
    sc->n_lights = n_lights;
    return upload(sc->ctx, sc->lights, l.data(), l.size() * sizeof(LightRec));
}

int trr_scene_set_settings(trr_scene* sc, const trr_settings* s) {
    if (!sc || !s) return fail(sc ? sc->ctx : nullptr, TRR_ERR_INVALID, "trr_scene_set_settings: NULL argument");
    DeviceGuard guard;
    if (!sc->replicas.empty()) {
        for (trr_scene* r : sc->replicas) { int rc = trr_scene_set_settings(r, s); if (rc != TRR_OK) return fail(sc->ctx, rc, r->ctx->err); }
        sc->settings = *s; sc->settings.skybox_image.rgb = nullptr; sc->have_settings = true;
        return TRR_OK;
    }
    HIPCHK(sc->ctx, hipSetDevice(sc->ctx->device));
    HIPCHK(sc->ctx, hipDeviceSynchronize());
    sc->settings = *s;
    sc->settings.skybox_image.rgb = nullptr;   // borrowed pointer: not kept
    sc->sky_w = sc->sky_h = 0;
    if (s->skybox_is_image) {
        if (!s->skybox_image.rgb || !s->skybox_image.width || !s->skybox_image.height)
            return fail(sc->ctx, TRR_ERR_INVALID, "trr_scene_set_settings: empty skybox image");
        sc->sky_w = s->skybox_image.width;
        sc->sky_h = s->skybox_image.height;
        int rc = upload(sc->ctx, sc->sky, s->skybox_image.rgb, (size_t)sc->sky_w * sc->sky_h * 3);
        if (rc != TRR_OK) return rc;
    }
    sc->have_settings = true;
    return TRR_OK;
}

int trr_scene_set_camera(trr_scene* sc, const trr_camera* c) {
    if (!sc || !c) return fail(sc ? sc->ctx : nullptr, TRR_ERR_INVALID, "trr_scene_set_camera: NULL argument");
    for (trr_scene* r : sc->replicas) (void)trr_scene_set_camera(r, c);
    sc->camera = *c;
    camera_basis(c, sc->cam_x, sc->cam_y, sc->cam_z, &sc->cam_focal);
    sc->have_camera = true;
    return TRR_OK;
}

// ---- render -----------------------------------------------------------------------------------------------------
struct Tiling {
    uint32_t w, h, bw, bh, nbx, nby, nblocks;
};
static bool tiling_of(uint32_t w, uint32_t h, Tiling* t) {
    t->w = w; t->h = h;
    t->bw = w / 20; t->bh = h / 20;               // render.rs:103
    if (t->bw == 0 || t->bh == 0) return false;   // render.rs:109 would divide by zero
    t->nbx = w / t->bw; t->nby = h / t->bh;       // render.rs:109-110
    t->nblocks = t->nbx * t->nby;
    return true;
}
static uint32_t owned_blocks(const Tiling& t, uint32_t shard, uint32_t n_shards) {
    return shard < t.nblocks ? (t.nblocks - shard + n_shards - 1) / n_shards : 0;
}
static uint32_t max_owned_blocks(const Tiling& t, uint32_t n_shards) { return (t.nblocks + n_shards - 1) / n_shards; }

size_t trr_strip_bytes(uint32_t width, uint32_t height, uint32_t n_shards) {
    Tiling t;
    if (n_shards == 0 || !tiling_of(width, height, &t)) return 0;
    return (size_t)max_owned_blocks(t, n_shards) * t.bw * t.bh * 3;
}

// Everything the kernels need about the frame, from the scene's current camera / settings / lights.
static FrameParams frame_params_of(const trr_scene* sc, const Tiling& tl, uint32_t shard, uint32_t n_shards, float delta,
                                   uint32_t max_depth) {
    const bool aa = sc->settings.anti_aliasing != 0;
    const uint32_t spp = aa ? sc->settings.samples : 1u;
    const uint32_t L = sc->n_lights;
    FrameParams fp{};
    fp.nodes = (const Node4*)sc->nodes.p; fp.qnodes = sc->quantised ? (const QNode4*)sc->qnodes.p : nullptr; fp.tris = (const TriRec*)sc->tris.p; fp.interp = (const InterpRec*)sc->interp.p;
    fp.objs = (const ObjRec*)sc->objs.p; fp.mats = (const MatRec*)sc->mats.p; fp.texs = (const TexRec*)sc->texs.p;
    fp.texels = (const uint8_t*)sc->texels.p; fp.lights = (const LightRec*)sc->lights.p;
    fp.n_objs = sc->n_objs; fp.n_lights = L;
    for (int k = 0; k < 3; ++k) {
        fp.ambient[k] = sc->settings.ambient_color[k];
        fp.sky_color[k] = sc->settings.skybox_color[k];
        fp.cam_pos[k] = sc->camera.position[k];
        fp.cam_x[k] = sc->cam_x[k]; fp.cam_y[k] = sc->cam_y[k]; fp.cam_z[k] = sc->cam_z[k];
    }
    fp.ambient_intensity = sc->settings.ambient_intensity;
    fp.sky = sc->settings.skybox_is_image ? (const uint8_t*)sc->sky.p : nullptr;
    fp.sky_w = sc->sky_w; fp.sky_h = sc->sky_h;
    fp.aa = aa ? 1u : 0u;
    fp.spp = spp;
    {
        float sq = std::sqrt((float)sc->settings.samples);   // mod.rs:226
        fp.sqrt_spp = sq >= 4294967296.0f ? 0xFFFFFFFFu : (sq > 0.0f ? (uint32_t)sq : 0u);
    }
    fp.seed = sc->settings.seed;
    fp.cam_focal = sc->cam_focal;
    fp.width = tl.w; fp.height = tl.h; fp.bw = tl.bw; fp.bh = tl.bh; fp.nbx = tl.nbx; fp.nby = tl.nby;
    fp.shard = shard; fp.n_shards = n_shards;
    fp.delta = delta; fp.max_depth = max_depth;
    return fp;
}

static hipEvent_t get_event(trr_ctx* ctx, size_t i) {
    while (ctx->events.size() <= i) {
        hipEvent_t ev;
        if (hipEventCreate(&ev) != hipSuccess) return nullptr;
        ctx->events.push_back(ev);
    }
    return ctx->events[i];
}

// counters block of one batch (uint32 words, cleared by ONE memset): [CW_WORK + 8 i ..] work counters (one per XCD segment) of the batch's trace launch number i (split schedule: camera
// rays 0, reflection rays of level l 2 l - 1, shadow rays of level l 2 l + 2; up to 520), [CW_ZERO] a word nobody writes,
// launch (work items claimed), [CW_NSHADOW + lvl] / [CW_NCHAIN + lvl] shadow rays emitted at / chain rays entering level
// lvl (level-synchronous schedule), [CW_SPILL + k] entries spilled by launch k (persistent schedule); the 64-bit kernel
// counters (device_types.h KC_*) follow at byte offset CW_WORDS*4.
enum { CW_WORK = 0, CW_WORK_STRIDE = 8, CW_NSHADOW = 4160, CW_NCHAIN = 4448, CW_SPILL = 4736, CW_ZERO = 5024,
       CW_NGROUPS = 5027,   // camera-ray cull: 64-sample groups of the batch that are traced
       CW_WORDS = 5120 };
static constexpr uint32_t COVER_BIG_CAP = 1u << 16;
static constexpr uint32_t MAX_FOLLOWUPS = 7;
static constexpr size_t RB_STRIDE = CW_WORDS * 4 + 8 * 16;   // per batch: launch / queue counters + up to 16 kernel counters
static_assert(KC_COUNT <= 16, "kernel counters outgrew the read-back block");

static int acquire_pinned(trr_ctx* ctx, size_t bytes, int* slot) {
    for (size_t i = 0; i < ctx->pinned.size(); ++i)
        if (!ctx->pinned[i].in_use && ctx->pinned[i].bytes >= bytes) {
            ctx->pinned[i].in_use = true;
            *slot = (int)i;
            return TRR_OK;
        }
    trr_ctx::Pinned pn;
    HIPCHK(ctx, hipHostMalloc(&pn.p, bytes, hipHostMallocDefault));
    pn.bytes = bytes;
    pn.in_use = true;
    ctx->pinned.push_back(pn);
    *slot = (int)ctx->pinned.size() - 1;
    return TRR_OK;
}

// Fills *f.out from the frame's read-back counters and events; the frame's work must have completed.
static void finish_stats(trr_ctx* ctx, const trr_ctx::PendingStats& f) {
    trr_stats* stats = f.out;
    const char* base = (const char*)ctx->pinned[f.pinned_slot].p;
    for (uint32_t bi = 0; bi < f.batches; ++bi) {
        const unsigned long long* k = (const unsigned long long*)(base + RB_STRIDE * bi + CW_WORDS * 4);
        uint32_t first = bi * f.blocks_per_batch;
        uint32_t nb = std::min(f.blocks_per_batch, f.n_owned - first);
        stats->rays_primary += (uint64_t)nb * f.bpix * f.spp;
        stats->rays_shadow += k[KC_SHADOW_RAYS];
        stats->rays_reflection += k[KC_CHAIN_RAYS];
        stats->queue_overflows += k[KC_QUEUE_OVERFLOW];
        stats->rays_shadow_culled += k[KC_SHADOW_CULLED];
        stats->rays_primary_culled += k[KC_PRIMARY_CULLED];
        stats->node_visits += k[KC_NODES]; stats->box_tests += k[KC_BOXES]; stats->tri_tests += k[KC_TRIS]; stats->rays_march += k[KC_MARCH];
        stats->wave_node_steps += k[KC_WNODE]; stats->wave_tri_steps += k[KC_WTRI]; stats->wave_outer_steps += k[KC_WOUTER];
        stats->wave_lane_steps += k[KC_WLANES]; stats->wave_trans_steps += k[KC_WTRANS]; stats->trans_lanes += k[KC_LTRANS];
        stats->wave_uniform_node_steps += k[KC_WUNI];
    }
    float span = 0.0f;
    for (size_t i = 0; i + 1 < f.ev_count; i += 2) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, ctx->events[f.ev_first + i], ctx->events[f.ev_first + i + 1]) == hipSuccess) {
            if (f.ev_kind[i / 2] == 0) { stats->ms_trace += ms; stats->trace_launches++; }
            else stats->ms_shade += ms;
        }
        if (hipEventElapsedTime(&ms, ctx->events[f.ev_first], ctx->events[f.ev_first + i + 1]) == hipSuccess) span = std::max(span, ms);
    }
    stats->blocks_rendered = f.blocks_done;
    stats->ms_total = span;   // first kernel start to last kernel end of the frame, on the device clock
}

// Raw device pointers of one work set: the buffers of one batch in flight (hipMalloc'ed per context by the ordinary entry
// points, carved out of the library's static scratch array by trr_render_shard_device_noalloc).
struct WorkView {
    HitRec* hits = nullptr;
    float4* level = nullptr;
    float4* term = nullptr;
    uint8_t* nlev = nullptr;
    float* color = nullptr;
    int32_t* occ = nullptr;
    // persistent schedule
    float4* wqueue = nullptr;   // wave-local ray queues of the trace grid (32-B entries)
    float4* gqueue = nullptr;   // two device-wide spill queues of spill_capacity() entries each (ping-pong between launches)
    // level-synchronous schedule
    float4* shadow_q = nullptr;     // [cap * L] one level's shadow rays (16-B compact entries or 32-B ray records)
    float4* shadow_deep[2] = {nullptr, nullptr};   // split schedule: the shadow rays of the odd / even levels >= 1
    hipStream_t aux[2] = {nullptr, nullptr};       // split schedule (nullptr: not available, e.g. the no-alloc entry point)
    hipEvent_t* sync_ev = nullptr;                 // split schedule: 2 * max_depth events
    float4* chain_q[2] = {nullptr, nullptr};   // [cap] reflection rays entering the even / odd levels
    void* counters = nullptr;   // RB_STRIDE bytes: work counter + kernel counters
    uint2* spill = nullptr;
    // camera-ray cull (nullptr: off)
    const uint8_t* cover_mask = nullptr;   // the frame's tile mask
    const uint32_t* cover_flags = nullptr; // the frame's give-up flag
    uint32_t* groups = nullptr;
};

static size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

// Capacity of a wavefront's ray queue.  A wavefront takes camera rays only when its queue is empty, so at most 64 chains
// are alive in it, and the queue is FIFO, so of each chain it holds at most the L shadow rays of its last hit plus its
// next reflection ray: 64 * (L + 1) entries, rounded up to a power of two (the kernel masks indices).
static uint32_t queue_capacity(uint32_t n_lights) {
    uint32_t need = 64u * (n_lights + 1u), c = 128u;
    while (c < need) c <<= 1;
    return c;
}

// Capacity of one spill queue: from the moment a wavefront can claim no more work items, every ray it would queue is
// spilled -- the children of its at most 64 rays in flight, L + 1 each (its own queue is empty at that moment, because
// it claims only when the queue is drained).
static uint32_t spill_capacity(uint32_t trace_grid, uint32_t n_lights) { return trace_grid * 4u * 64u * (n_lights + 1u); }

// Two schedules of one batch, same results bit for bit (same arithmetic, same records):
//  * SCHED_PERSISTENT: ONE trace launch; every wavefront shades the hits of its own rays and drains the shadow / reflection
//    rays they spawn from a queue of its own before it claims more camera rays (k_trace).  No level barriers: best where
//    the ramp-up and tail of a launch weigh more than the work in it (small frames, thin deep levels).
//  * SCHED_LEVELS: one launch per ray population -- camera rays, then per level hit set-up + (shadow rays of the level and
//    reflection rays of the next) -- with specialised kernels (k_trace_level, k_setup).  A ray costs less (thin ray start /
//    end blocks, hit set-up with full wavefronts); 2 * max_depth + 1 dependent launches per batch.
// choose_schedule() picks per call, from measurements.
enum { SCHED_PERSISTENT = 0, SCHED_LEVELS = 1 };
struct BatchGrids {
    uint32_t schedule;
    uint32_t trace, aux_max;   // persistent trace grid; largest grid of the set-up / resolve kernels
    uint32_t followups;        // persistent: launches after the first that drain what the launch before spilled (0 = none)
    uint32_t chunk_min[2];     // persistent: smallest claim of work items per wavefront in the first / the follow-up launches
    uint32_t raygen, secondary;   // levels: grids of the camera-ray and the secondary-ray launches
    uint32_t chain;               // levels, split: grid of a launch that traces reflection rays only (the batch's dependent chain)
    uint32_t aux_deep;            // largest grid of the set-up passes of levels >= 1
    uint32_t levels;              // levels: chain levels that can hold a hit (1 when no material is a mirror, else max_depth)
    uint32_t cull;                // levels: 1 = k_setup culls the shadow rays of lights that add exactly nothing (kernels.hip)
    uint32_t split;               // levels: 1 = reflection rays and shadow rays in launches of their own, on streams of their own
    uint32_t split_drain;         // levels, split: 1 = after level 0's set-up ONE k_trace launch follows every reflection chain to its
                                  // end (levels >= 1: their hits, shadow rays and further reflections) while level 0's shadow rays
                                  // are traced on their own stream -- the batch's dependent chain is then four launches long
    uint32_t drain_level;         // levels: from this level's set-up on, ONE k_trace launch (grid `trace`) drains the rest of the
                                  // batch -- the level's shadow rays, the next level's reflection rays and all they spawn
                                  // (NO_DRAIN: never)
};
constexpr uint32_t NO_DRAIN = 0xFFFFFFFFu;

// Queues the launch sequence of one batch on `s_` (no host synchronisation, no allocation), then the two resolve passes
// into the strip.
extern "C++" {
template <class EvBegin, class EvEnd>
static void queue_batch(const FrameParams& fp, const BatchParams& bp, const WorkView& w, const BatchGrids& g, bool count,
                        uint8_t* strip_dev, uint32_t* ids_obj_dev, uint32_t* ids_tri_dev, hipStream_t s_,
                        EvBegin&& ev_begin, EvEnd&& ev_end) {
    uint32_t* cw = (uint32_t*)w.counters;
    unsigned long long* kc = (unsigned long long*)((char*)w.counters + CW_WORDS * 4);
    // ---- camera-ray cull: the batch's 64-sample groups that own a pixel of a tile the frame's mask marks ----
    const bool camcull = w.cover_mask != nullptr && w.groups != nullptr && bp.n_samples != 0u;
    if (camcull) {
        CoverIO cio{};
        cio.mask = const_cast<uint8_t*>(w.cover_mask);
        cio.tiles_x = (fp.width + 7u) / 8u; cio.tiles_y = (fp.height + 7u) / 8u;
        cio.flags = const_cast<uint32_t*>(w.cover_flags);
        cio.groups = w.groups; cio.n_groups = cw + CW_NGROUPS;
        cio.nlev = w.nlev;
        cio.counters = kc;
        ev_begin(1, s_);
        launch_cover_groups(s_, fp, bp, cio);
        ev_end(s_);
    }
    if (g.schedule == SCHED_PERSISTENT) {
        TraceIO tio{};
        tio.hits = w.hits;
        tio.level = w.level;
        tio.term = w.term;
        tio.nlev = w.nlev;
        tio.occ = w.occ;
        tio.wqueue = w.wqueue;
        tio.qcap = queue_capacity(fp.n_lights);
        tio.first_obj = ids_obj_dev;
        tio.first_tri = ids_tri_dev;
        tio.counters = kc;
        tio.spill = w.spill;
        tio.cull = g.cull;
        if (camcull) { tio.groups = w.groups; tio.n_groups = cw + CW_NGROUPS; }
        // Launch 0 takes the camera rays.  With follow-ups, a wavefront that can claim no more work items spills the rays
        // it would queue, and launch k >= 1 deals what launch k - 1 spilled evenly over all wavefronts (the last one spills
        // nothing; a chain is at most max_depth reflection rays long, so deeper follow-ups would find nothing).
        const uint32_t scap = spill_capacity(g.trace, fp.n_lights);
        const uint32_t n_follow = fp.max_depth == 0u ? 0u : std::min(std::min(g.followups, fp.max_depth), MAX_FOLLOWUPS);
        for (uint32_t k = 0; k <= n_follow; ++k) {
            tio.work_counter = cw + CW_WORK + CW_WORK_STRIDE * k;
            tio.src_q = k == 0 ? nullptr : w.gqueue + (size_t)((k - 1) & 1u) * scap * 2u;
            tio.src_n = k == 0 ? nullptr : cw + CW_SPILL + (k - 1);
            tio.src_cap = scap;
            tio.spill_q = k == n_follow ? nullptr : w.gqueue + (size_t)(k & 1u) * scap * 2u;
            tio.spill_n = k == n_follow ? nullptr : cw + CW_SPILL + k;
            tio.spill_cap = scap;
            tio.chunk_min = std::max(1u, g.chunk_min[k == 0 ? 0 : 1]);
            ev_begin(0, s_);
            launch_trace(count, g.trace, s_, fp, bp, tio);
            ev_end(s_);
        }
    } else {
        const uint32_t aux_grid = std::min<uint32_t>(g.aux_max, (bp.n_samples + 255) / 256);
        LevelIO tio{};
        tio.hits = w.hits;
        tio.nlev = w.nlev;
        tio.occ = w.occ;
        tio.counters = kc;
        tio.spill = w.spill;
        tio.work_counter = cw + CW_WORK + 0;
        tio.first_obj = ids_obj_dev;
        tio.first_tri = ids_tri_dev;
        if (camcull) { tio.groups = w.groups; tio.n_groups = cw + CW_NGROUPS; }
        ev_begin(0, s_);
        launch_trace_level(true, count, g.raygen, s_, fp, bp, tio);
        ev_end(s_);
        tio.groups = nullptr; tio.n_groups = nullptr;   // (the secondary-ray launches copy this descriptor)
        const uint32_t n_levels = std::min(fp.max_depth, std::max(1u, g.levels));   // deeper levels would find empty queues
        if (g.split != 0u && n_levels >= 1u && w.aux[0] != nullptr && w.sync_ev != nullptr) {
            // ---- split schedule.  Only the reflection rays carry the chain on: level l + 1's hit set-up needs their hits
            // and nothing of level l's shadow rays, which are the bulk.  So the main stream runs the thin dependent chain
            //     camera rays -> set-up 0 -> reflection rays 1 -> set-up 1 -> reflection rays 2 -> ... -> resolve
            // and every level's shadow rays go to a launch of their own on another stream as soon as that level is set
            // up (level 0's on aux[0], the deeper levels' on aux[1]): the deep levels finish while level 0's shadow rays
            // are still being traced, instead of after them.  Shadow queues: level 0 has its own, the deeper levels
            // alternate between two (level l + 2's set-up waits for level l's shadow launch).
            const char* e_shadow = std::getenv("TRR_SHADOW");
            const bool wide_shadow = e_shadow && e_shadow[0] == 'w';
            const uint32_t compact = (!wide_shadow && fp.n_lights <= 128u && bp.cap <= (1u << 25)) ? 1u : 0u;
            hipEvent_t last_aux[2] = {nullptr, nullptr};
            for (uint32_t lvl = 0; lvl < n_levels; ++lvl) {
                float4* sq = lvl == 0 ? w.shadow_q : w.shadow_deep[(lvl - 1) & 1];
                if (lvl >= 3) (void)hipStreamWaitEvent(s_, w.sync_ev[2 * (lvl - 2) + 1], 0);   // the queue's previous user is done
                SetupIO sio{};
                sio.chain_in = w.chain_q[lvl & 1];
                sio.n_chain_in = cw + CW_NCHAIN + lvl;
                sio.hits = w.hits; sio.level = w.level; sio.term = w.term; sio.nlev = w.nlev;
                sio.shadow_compact = compact;
                sio.occ = w.occ; sio.cull = g.cull;
                sio.shadow_out = sq;
                sio.n_shadow_out = cw + CW_NSHADOW + lvl;
                sio.chain_out = w.chain_q[(lvl + 1) & 1];
                sio.n_chain_out = cw + CW_NCHAIN + lvl + 1;
                sio.counters = kc;
                ev_begin(1, s_);
                launch_setup(lvl == 0, lvl == 0 ? aux_grid : std::min(aux_grid, g.aux_deep), s_, fp, bp, sio, lvl);
                ev_end(s_);
                (void)hipEventRecord(w.sync_ev[2 * lvl], s_);
                // this level's shadow rays, off the chain
                hipStream_t sa = w.aux[lvl == 0 ? 0 : 1];
                (void)hipStreamWaitEvent(sa, w.sync_ev[2 * lvl], 0);
                LevelIO ts = tio;
                ts.shadow_q = sq; ts.shadow_compact = compact; ts.shadow_level = lvl;
                ts.n_shadow = cw + CW_NSHADOW + lvl;
                ts.chain_q = nullptr; ts.n_chain = cw + CW_ZERO;
                ts.work_counter = cw + CW_WORK + CW_WORK_STRIDE * (2 * lvl + 2);
                ts.first_obj = nullptr; ts.first_tri = nullptr;
                ev_begin(0, sa);
                launch_trace_level(false, count, g.secondary, sa, fp, bp, ts);
                ev_end(sa);
                (void)hipEventRecord(w.sync_ev[2 * lvl + 1], sa);
                last_aux[lvl == 0 ? 0 : 1] = w.sync_ev[2 * lvl + 1];
                if (g.split_drain != 0u && lvl == 0 && n_levels >= 2 && w.wqueue != nullptr) {
                    // everything deeper than level 0 in one persistent launch: level 1's reflection rays are its work items
                    TraceIO d{};
                    d.hits = w.hits; d.level = w.level; d.term = w.term; d.nlev = w.nlev; d.occ = w.occ;
                    d.wqueue = w.wqueue;
                    d.qcap = queue_capacity(fp.n_lights);
                    d.counters = kc;
                    d.spill = w.spill;
                    d.cull = g.cull;
                    d.work_counter = cw + CW_WORK + CW_WORK_STRIDE * 1;
                    d.src_q = w.chain_q[1]; d.src_n = cw + CW_NCHAIN + 1; d.src_cap = bp.cap;
                    d.src_compact = 0; d.src_level = 0;
                    d.chunk_min = std::max(1u, g.chunk_min[1]);
                    ev_begin(0, s_);
                    launch_trace(count, g.trace, s_, fp, bp, d);
                    ev_end(s_);
                    break;
                }
                // the next level's reflection rays, on the chain
                if (lvl + 1 < n_levels) {
                    LevelIO tc = tio;
                    tc.shadow_q = nullptr; tc.n_shadow = cw + CW_ZERO; tc.shadow_compact = 0; tc.shadow_level = 0;
                    tc.chain_q = w.chain_q[(lvl + 1) & 1];
                    tc.n_chain = cw + CW_NCHAIN + lvl + 1;
                    tc.work_counter = cw + CW_WORK + CW_WORK_STRIDE * (2 * lvl + 1);
                    tc.first_obj = nullptr; tc.first_tri = nullptr;
                    ev_begin(0, s_);
                    launch_trace_level(false, count, g.chain, s_, fp, bp, tc);
                    ev_end(s_);
                }
            }
            for (hipEvent_t e : last_aux)
                if (e) (void)hipStreamWaitEvent(s_, e, 0);   // the resolve pass needs every occluder code
        } else
        // max_depth == 0: one set-up pass turns every sample into skybox(ray.direction) (mod.rs:125-127), no secondary rays
        for (uint32_t lvl = 0; lvl < std::max(1u, n_levels); ++lvl) {
            SetupIO sio{};
            sio.chain_in = w.chain_q[lvl & 1];
            sio.n_chain_in = cw + CW_NCHAIN + lvl;
            sio.hits = w.hits;
            sio.level = w.level;
            sio.term = w.term;
            sio.nlev = w.nlev;
            // 16-B compact shadow entries whenever they can name every light and sample (kernels.hip); TRR_SHADOW=wide keeps
            // the 32-B ray records (experiments, tests of that path)
            const char* e_shadow = std::getenv("TRR_SHADOW");
            const bool wide_shadow = e_shadow && e_shadow[0] == 'w';
            const uint32_t compact = (!wide_shadow && fp.n_lights <= 128u && bp.cap <= (1u << 25)) ? 1u : 0u;
            sio.shadow_compact = compact;
            sio.occ = w.occ; sio.cull = g.cull;
            sio.shadow_out = w.shadow_q;
            sio.n_shadow_out = cw + CW_NSHADOW + lvl;
            sio.chain_out = w.chain_q[(lvl + 1) & 1];
            sio.n_chain_out = cw + CW_NCHAIN + lvl + 1;
            sio.counters = kc;
            ev_begin(1, s_);
            launch_setup(lvl == 0, lvl == 0 ? aux_grid : std::min(aux_grid, g.aux_deep), s_, fp, bp, sio, lvl);
            ev_end(s_);
            if (fp.max_depth == 0) break;
            if (lvl >= g.drain_level && lvl + 1 < n_levels) {
                // levels-then-drain: the deep levels hold few rays each, and a launch per level costs a ramp-up and a tail
                // as long as its slowest ray (a march through glass, a long reflection path) whatever it holds.  k_trace
                // takes this level's shadow rays and the next level's reflection rays as its work items and follows
                // every chain to its end inside the one launch.
                TraceIO d{};
                d.hits = w.hits; d.level = w.level; d.term = w.term; d.nlev = w.nlev; d.occ = w.occ;
                d.wqueue = w.wqueue;
                d.qcap = queue_capacity(fp.n_lights);
                d.counters = kc;
                d.spill = w.spill;
                d.cull = g.cull;
                d.work_counter = cw + CW_WORK + CW_WORK_STRIDE * (lvl + 1);
                d.src_q = w.shadow_q; d.src_n = cw + CW_NSHADOW + lvl; d.src_cap = bp.cap * std::max(1u, fp.n_lights);
                d.src_q2 = w.chain_q[(lvl + 1) & 1]; d.src_n2 = cw + CW_NCHAIN + lvl + 1; d.src_cap2 = bp.cap;
                d.src_compact = compact; d.src_level = lvl;
                d.chunk_min = std::max(1u, g.chunk_min[1]);
                ev_begin(0, s_);
                launch_trace(count, g.trace, s_, fp, bp, d);
                ev_end(s_);
                break;
            }
            LevelIO t2 = tio;
            t2.shadow_q = w.shadow_q;
            t2.shadow_compact = compact; t2.shadow_level = lvl;
            t2.n_shadow = cw + CW_NSHADOW + lvl;
            t2.chain_q = w.chain_q[(lvl + 1) & 1];
            t2.n_chain = cw + CW_NCHAIN + lvl + 1;
            t2.work_counter = cw + CW_WORK + CW_WORK_STRIDE * (lvl + 1);
            t2.first_obj = nullptr;
            t2.first_tri = nullptr;
            ev_begin(0, s_);
            launch_trace_level(false, count, g.secondary, s_, fp, bp, t2);
            ev_end(s_);
        }
    }
    ResolveIO rio{};
    rio.level = w.level; rio.term = w.term; rio.nlev = w.nlev; rio.color = w.color;
    rio.occ = w.occ; rio.strip = strip_dev;
    ev_begin(1, s_);
    launch_resolve(std::min<uint32_t>(g.aux_max * 4, (bp.n_samples + 255) / 256),
                   std::min<uint32_t>(g.aux_max, (bp.n_pixels + 255) / 256), s_, fp, bp, rio);
    ev_end(s_);
}
}   // extern "C++"

// Which schedule a call uses.  Measured on one MI355X (scripts/experiments/quick.py, two frames in flight, ms per frame,
// levels / persistent): 800x600 2 objects max_depth 2: 1.42 / 1.64; 1080p max_depth 5: 1.64 / 1.33; 4K terrain 10 M
// triangles: 2.23 / 2.32; 4K mirrors max_depth 9: 12.2 / 15.1; an eighth of the 4K x 16 spp frame: 4.9 / 6.1; that whole
// frame: 30.8 / 39.  The persistent schedule wins only where many levels hold few rays each (deep max_depth, a frame of a
// few million rays): the launches it saves then outweigh its dearer rays.  TRR_SCHEDULE=levels|persistent overrides.
static uint32_t choose_schedule(uint64_t samples, uint32_t n_lights, uint32_t max_depth) {
    if (const char* e = std::getenv("TRR_SCHEDULE")) return e[0] == 'l' ? SCHED_LEVELS : SCHED_PERSISTENT;
    return (max_depth >= 4 && samples * (1ull + n_lights) <= (12ull << 20)) ? SCHED_PERSISTENT : SCHED_LEVELS;
}

// trr_render_blocks: batches of at most blocks_per_batch blocks, rendered in block order; after_batch runs on the host
// once a batch's blocks are final in the strip (its stream has been waited for) and may end the render with an error.
struct BatchHook {
    uint32_t blocks_per_batch = 0;
    std::function<int(uint32_t first_owned, uint32_t n_blocks, hipStream_t stream)> after_batch;
};

static int render_shard_impl(trr_ctx* ctx, trr_scene* sc, float delta, uint32_t max_depth, uint32_t flags, uint32_t shard,
                             uint32_t n_shards, uint8_t* strip_dev, size_t strip_bytes, hipStream_t st, uint32_t* ids_obj_dev,
                             uint32_t* ids_tri_dev, const volatile int32_t* cancel_flag, trr_progress_cb progress, void* user,
                             trr_stats* stats, const BatchHook* hook = nullptr) {
    if (!ctx || !sc) return fail(ctx, TRR_ERR_INVALID, "render: NULL context or scene");
    if (sc->ctx != ctx) return fail(ctx, TRR_ERR_INVALID, "render: scene belongs to another context");
    if (!sc->have_camera || !sc->have_settings) return fail(ctx, TRR_ERR_INVALID, "render: camera/settings not set");
    if (!(delta >= 0.0f)) return fail(ctx, TRR_ERR_INVALID, "render: delta must be >= 0");
    if (max_depth > 254) return fail(ctx, TRR_ERR_INVALID, "render: max_depth above 254");
    if (n_shards == 0 || shard >= n_shards) return fail(ctx, TRR_ERR_INVALID, "render: bad shard / n_shards");
    Tiling tl;
    if (!tiling_of(sc->camera.width, sc->camera.height, &tl))
        return fail(ctx, TRR_ERR_RESOLUTION, "render: width and height must be >= 20 (reference divides by zero, render.rs:109)");
    const size_t need_strip = (size_t)max_owned_blocks(tl, n_shards) * tl.bw * tl.bh * 3;
    if (!strip_dev || strip_bytes < need_strip) return fail(ctx, TRR_ERR_INVALID, "render: strip buffer too small");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    auto wall0 = std::chrono::steady_clock::now();

    const bool aa = sc->settings.anti_aliasing != 0;
    const uint32_t spp = aa ? sc->settings.samples : 1u;
    const uint32_t L = sc->n_lights;
    const uint32_t n_owned = owned_blocks(tl, shard, n_shards);
    const uint32_t bpix = tl.bw * tl.bh;
    const bool count = (flags & TRR_FLAG_COUNTERS) != 0;

    HIPCHK(ctx, hipMemsetAsync(strip_dev, 0, need_strip, st));
    trr_stats zero{};
    if (stats) *stats = zero;
    if (spp == 0 || n_owned == 0) {
        // mod.rs:228-249 with samples == 0: the mean of nothing is NaN, which quantises to 0 (render.rs:136)
        if (stats) stats->blocks_rendered = n_owned;
        return TRR_OK;
    }
    if ((uint64_t)bpix * spp > (1ull << 31)) return fail(ctx, TRR_ERR_INVALID, "render: block too large for one batch");

    // ---- batch sizing: batches are pipelined over `lanes` work sets (each its own stream + buffers) -------------
    const bool interactive = progress != nullptr || cancel_flag != nullptr || hook != nullptr;   // polled per batch: keep batches in order
    const uint64_t spb = (uint64_t)bpix * spp;
    const uint32_t schedule = choose_schedule((uint64_t)n_owned * spb, L, max_depth);
    const uint64_t per_sample = sizeof(HitRec) + 16ull * LEVEL_F4 * std::max(1u, max_depth) + 16 + 1 + 12 +
                                4ull * std::max(1u, L) * std::max(1u, max_depth) + 16 +
                                (schedule == SCHED_LEVELS ? 3 * 32ull * std::max(1u, L) + 64 : 0);   // three shadow queues (split schedule)
    // Batch count (measured on MI355X, scripts/experiments/sweep_batches.py): few, large batches -- every persistent launch pays a
    // ramp-up and a tail of fixed length -- but at least two, on two lanes, so that one batch's tails and small deep-level
    // launches overlap the other's bulk; three lanes once there are four or more batches.  An odd batch per lane keeps
    // the lanes out of lock-step.  TRR_LANES / TRR_BATCHES_PER_LANE / TRR_BATCHES override (experiments).
    const uint64_t total_samples = (uint64_t)n_owned * spb;
    uint64_t want_batches = std::max<uint64_t>(2, (total_samples + (1ull << 25) - 1) >> 25);
    uint32_t lanes_want = want_batches >= 4 ? 3u : 2u;
    // A lone frame (the caller waits for it: no TRR_FLAG_PIPELINED) of at most 2^23 samples -- 4K at 1 spp and below -- is
    // all ramp-ups and tails: four batches on three lanes balance them better than two on two (measured, lone frame,
    // two / four batches: 1080p max_depth 5 2.23 / 1.93 ms, 4K terrain 3.18 / 2.90, 800x600 2.25 / 2.17, 4K mirrors
    // max_depth 9 15.6 / 15.4; from 2^24 samples on -- an eighth of the 4K x 16 spp frame -- two batches are better again).
    if ((flags & TRR_FLAG_PIPELINED) == 0 && total_samples <= (1ull << 23)) { want_batches = 4; lanes_want = 3; }
    const char* e_lanes = std::getenv("TRR_LANES");
    const char* e_bpl = std::getenv("TRR_BATCHES_PER_LANE");
    if (e_lanes || e_bpl) {
        lanes_want = e_lanes ? (uint32_t)std::min(trr_ctx::MAX_LANES, std::max(1, std::atoi(e_lanes))) : 3u;
        want_batches = (uint64_t)lanes_want * (e_bpl ? (uint32_t)std::max(1, std::atoi(e_bpl)) : 2u);
    }
    if (const char* e = std::getenv("TRR_BATCHES")) {
        want_batches = (uint64_t)std::max(1, std::atoi(e));
        lanes_want = e_lanes ? lanes_want : (uint32_t)std::min<uint64_t>(3, want_batches);
    }
    // A caller that keeps two frames in flight (TRR_FLAG_PIPELINED) and whose shard fits one batch: one batch per frame,
    // frames alternating between two work sets (measured: -9 % per frame for a quarter / an eighth of the 4K frame
    // against two half-size batches per frame; a lone frame would be slower this way, hence the flag).
    const bool pipelined_single = !interactive && !e_lanes && !e_bpl && !std::getenv("TRR_BATCHES") &&
                                  (flags & TRR_FLAG_PIPELINED) != 0 && total_samples <= (1ull << 25);
    if (pipelined_single) { want_batches = 1; lanes_want = 1; }
    if (interactive) want_batches = 1;   // polled per batch: keep batches in order, as large as memory allows
    want_batches = std::min<uint64_t>(want_batches, n_owned);
    // A block sink (trr_render_blocks) gets two lanes: batch k+1 is queued before batch k is waited for and delivered, so
    // the read-back, the sink's work and batch k's tail overlap batch k+1 (measured at 4K, 20 blocks per batch: 18.5 ->
    // 11.0 ms at 1 spp, 59.6 -> 41.8 ms at 16 spp).  Plain cancel / progress polling stays on one lane.
    const bool lagged = hook != nullptr && hook->after_batch && n_owned > 1;
    const uint32_t lanes = lagged ? 2u : interactive ? 1u : (uint32_t)std::min<uint64_t>(lanes_want, want_batches);
    // work-buffer budget: a third of the device memory that is free now, at most 96 GB (MI355X: 288 GB HBM3E per GPU)
    if (ctx->budget_total == 0) {   // asked once per context: the query is an ioctl
        ctx->budget_total = 24ull << 30;
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
            ctx->budget_total = std::max<uint64_t>(1ull << 30, std::min<uint64_t>(96ull << 30, (uint64_t)free_b / 3));
    }
    const uint64_t budget_total = ctx->budget_total;
    const uint64_t budget = budget_total / lanes;
    const uint64_t max_samples = std::min<uint64_t>(budget / per_sample, 1ull << 25);
    uint32_t blocks_per_batch = (uint32_t)std::max<uint64_t>(
        1, std::min<uint64_t>((n_owned + want_batches - 1) / want_batches, max_samples / spb));
    if (hook && hook->blocks_per_batch) blocks_per_batch = std::min(blocks_per_batch, hook->blocks_per_batch);
    const uint32_t cap = (uint32_t)(((uint64_t)blocks_per_batch * spb + 63) & ~63ull);
    const uint32_t D = std::max(1u, max_depth);
    // Persistent trace grid: every CU gets as many 256-thread blocks as the kernel's registers / LDS admit, four at most (a
    // fifth was measured neutral to slightly slower).  Level-synchronous schedule: secondary-ray launches the same,
    // camera-ray launches two blocks per CU -- most camera rays are short and coherent, and the slots they leave free let
    // the other lanes' secondary launches in (grid sweep on MI355X: 4 -> 2 blocks per CU is -4 % frame time for the whole
    // 4K frame, -9 % for an eighth of it).  TRR_GRID_BPC="persistent,raygen,secondary" overrides (experiments).
    const int cb = count ? 1 : 0;
    int bpc_p = std::min(4, ctx->bpc[cb]), bpc_r = std::min(2, ctx->bpc_level[1][cb]);
    int bpc_s = std::min(4, std::min(ctx->bpc_level[0][cb], ctx->bpc_level[1][cb]));
    if (const char* e = std::getenv("TRR_GRID_BPC")) {
        int a = 0, b = 0, c = 0;
        if (std::sscanf(e, "%d,%d,%d", &a, &b, &c) == 3 && a > 0 && b > 0 && c > 0) {
            bpc_p = std::min(a, ctx->bpc[cb]); bpc_r = std::min(b, ctx->bpc_level[1][cb]); bpc_s = std::min(c, ctx->bpc_level[0][cb]);
        }
    }
    const uint32_t n_cu = (uint32_t)std::max(1, ctx->n_cu);
    BatchGrids grids{};
    grids.schedule = schedule;
    grids.trace = n_cu * (uint32_t)bpc_p;
    grids.aux_max = n_cu * 8;
    grids.followups = 0;
    grids.chunk_min[0] = 64; grids.chunk_min[1] = 16;
    grids.raygen = n_cu * (uint32_t)bpc_r;
    grids.secondary = n_cu * (uint32_t)bpc_s;
    // Launches of the dependent chain that hold few rays -- reflection rays only (split schedule), set-up passes of the levels
    // >= 1 -- get small grids: their blocks have to find slots on CUs that the bulk launches fill, and a grid of a thousand
    // blocks that exit at once still has to be dealt through them.  Measured (TRR_GRID_CHAIN_BPC / TRR_GRID_SETUP_DEEP 4 / 8 ->
    // 2 / 2): an eighth of the 4K x 16 spp frame 4.96 -> 4.90 ms, 4K mirrors max_depth 9 14.5 -> 14.1 ms, the whole frame unchanged.
    grids.chain = n_cu * (uint32_t)std::min(bpc_s, 2);
    grids.aux_deep = n_cu * 2;
    if (const char* e = std::getenv("TRR_GRID_CHAIN_BPC")) grids.chain = n_cu * (uint32_t)std::min(bpc_s, std::max(1, std::atoi(e)));
    if (const char* e = std::getenv("TRR_GRID_SETUP_DEEP")) grids.aux_deep = n_cu * (uint32_t)std::min(8, std::max(1, std::atoi(e)));
    if (const char* e = std::getenv("TRR_FOLLOWUPS")) grids.followups = (uint32_t)std::max(0, std::atoi(e));
    // Levels-then-drain (measured, ms per frame without / with drain from level 2: a lone 4K x 16 spp frame 32.3 / 31.5, an
    // eighth of it 5.64 / 5.42, 4K mirrors max_depth 9 15.8 / 14.9): a caller that waits for every frame gains what the
    // thin deep levels' launch tails cost; one that keeps two frames in flight (TRR_FLAG_PIPELINED) already hides those
    // tails behind the other frame and only pays k_trace's dearer rays (30.6 / 31.3, 12.3 / 13.4), so it keeps the levels.
    grids.levels = sc->any_reflective ? max_depth : std::min(max_depth, 1u);
    {
        const char* ce = std::getenv("TRR_CULL");
        grids.cull = (ce && ce[0] == '0') ? 0u : 1u;
    }
    // Split schedule (queue_batch): every level's shadow rays on a stream of their own, off the chain of reflection rays.  It
    // needs more concurrent streams than HIP's default of 4 hardware queues carries -- with 4, streams of different lanes
    // share a queue and whole batches serialise (an eighth of the 4K x 16 spp frame 5.7 instead of 5.4 ms) -- so it is
    // used only when GPU_MAX_HW_QUEUES says there are at least 8 (the library asks for 16 when it is loaded before HIP
    // starts; bench.py and the tests set it), and for frames or shards of at most 2^26 samples: measured with 16 queues,
    // without / with the split, lone frame: an eighth of the 4K x 16 spp frame 5.44 / 5.17 ms, a quarter 9.05 / 8.71, a
    // half 16.3 / 16.0, 4K mirrors max_depth 9 15.4 / 14.5; the whole frame 29.9 / 30.8 (nine persistent launches at once
    // crowd each other), so it keeps the plain levels.  TRR_SPLIT=0|1 overrides.
    {
        grids.split = (g_hw_queues >= 8 && total_samples <= (1ull << 26) && grids.levels >= 2) ? 1u : 0u;
    }
    if (const char* e = std::getenv("TRR_SPLIT")) grids.split = std::atoi(e) != 0 ? 1u : 0u;
    grids.split_drain = 0;
    if (const char* e = std::getenv("TRR_SPLIT_DRAIN")) grids.split_drain = std::atoi(e) != 0 ? 1u : 0u;
    // a pipelined caller's small frames hide their thin deep levels behind the other frame; a big frame (four batches on three
    // lanes, e.g. 4K x 16 spp: 29.4 ms without a drain, 28.2 / 27.8 with one from level 2 / 3) gains from it like a lone one
    grids.drain_level = (flags & TRR_FLAG_PIPELINED) ? (total_samples > (1ull << 26) ? 3u : NO_DRAIN) : 2u;
    if (const char* e = std::getenv("TRR_DRAIN_LEVEL")) grids.drain_level = std::atoi(e) < 0 ? NO_DRAIN : (uint32_t)std::atoi(e);
    const bool drains = schedule == SCHED_LEVELS && ((grids.split == 0 && grids.drain_level != NO_DRAIN && grids.drain_level + 1 < max_depth) ||
                                                     (grids.split != 0 && grids.split_drain != 0 && max_depth >= 2));
    const uint32_t trace_grid = schedule == SCHED_PERSISTENT ? grids.trace
                                                             : std::max(std::max(grids.raygen, grids.secondary), drains ? grids.trace : 0u);   // sizes the stack-spill area
    if (std::getenv("TRR_DEBUG"))
        std::fprintf(stderr, "trr: schedule %s%s, levels %u, resident blocks per CU persistent %d raygen %d secondary %d, lanes %u batches %llu\n",
                     schedule == SCHED_LEVELS ? "levels" : "persistent", grids.split ? " (split)" : "", grids.levels, ctx->bpc[cb],
                     ctx->bpc_level[1][cb], ctx->bpc_level[0][cb], lanes, (unsigned long long)want_batches);
    const uint32_t qcap = queue_capacity(L);
    int rc;
    // camera-ray cull (kernels.hip: k_cover_*): whenever the scene's leaf boxes are all in its node array; TRR_CAMCULL=0 switches it off
    bool camcull_on = sc->cover_ok && sc->cover_boxes.p != nullptr;
    if (const char* e = std::getenv("TRR_CAMCULL")) camcull_on = camcull_on && e[0] != '0';
    // `pool` work sets may be used by this call: its `lanes` consecutive ones starting at lane_base.  A single-batch frame
    // of a pipelined caller (TRR_FLAG_PIPELINED) takes the two sets in turn, so that consecutive frames, queued on
    // different streams, overlap each other instead of two half-size batches of the same frame.
    uint32_t pool = lanes, lane_base = 0;
    if (pipelined_single) {
        pool = 2;
        if (const char* e = std::getenv("TRR_POOL")) pool = (uint32_t)std::min(trr_ctx::MAX_LANES, std::max(2, std::atoi(e)));
        lane_base = ctx->rotate++ % pool;
    }
    for (uint32_t k = 0; k < pool; ++k) {
        trr_ctx::WorkSet& w = ctx->ws[k];
        if ((rc = ensure(ctx, w.hits, (size_t)cap * sizeof(HitRec))) || (rc = ensure(ctx, w.level, (size_t)cap * 16 * LEVEL_F4 * D)) ||
            (rc = ensure(ctx, w.term, (size_t)cap * 16)) || (rc = ensure(ctx, w.nlev, (size_t)cap)) || (rc = ensure(ctx, w.color, (size_t)cap * 12)) ||
            (rc = ensure(ctx, w.occ, (size_t)cap * 4 * std::max(1u, L) * D)) || (rc = ensure(ctx, w.counters, RB_STRIDE)) ||
            (rc = ensure(ctx, w.spill, trace_spill_bytes(trace_grid))))
            return rc;
        if (camcull_on && (rc = ensure(ctx, w.groups, (size_t)(cap / 64 + 1) * 4))) return rc;
        if (schedule == SCHED_PERSISTENT) {
            if ((rc = ensure(ctx, w.wqueue, trace_queue_bytes(trace_grid, qcap))) ||
                (rc = ensure(ctx, w.gqueue, grids.followups ? 2 * (size_t)spill_capacity(trace_grid, L) * 32 : 256)))
                return rc;
        } else {
            const bool split_bufs = grids.split != 0 && max_depth >= 2;
            if (grids.split != 0)
                while (w.sync_ev.size() < 2 * (size_t)D + 2) {
                    hipEvent_t ev = nullptr;
                    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return fail(ctx, TRR_ERR_DEVICE, "hipEventCreate failed");
                    w.sync_ev.push_back(ev);
                }
            if ((rc = ensure(ctx, w.shadow_q, (size_t)cap * 32 * std::max(1u, L))) || (rc = ensure(ctx, w.chain_q[0], (size_t)cap * 32)) ||
                (rc = ensure(ctx, w.chain_q[1], (size_t)cap * 32)) ||
                (split_bufs && ((rc = ensure(ctx, w.shadow_deep[0], (size_t)cap * 32 * std::max(1u, L))) ||
                                (max_depth >= 3 && (rc = ensure(ctx, w.shadow_deep[1], (size_t)cap * 32 * std::max(1u, L)))))) ||
                (drains && (rc = ensure(ctx, w.wqueue, trace_queue_bytes(grids.trace, qcap)))))
                return rc;
        }
    }

    const FrameParams fp = frame_params_of(sc, tl, shard, n_shards, delta, max_depth);

    const uint32_t n_batches = (n_owned + blocks_per_batch - 1) / blocks_per_batch;
    // statistics: per-launch events on the launch streams + one read-back of the queue / kernel counters per batch.
    // TRR_FLAG_DEFER_STATS leaves both pending (no host synchronisation here); trr_collect_stats fills *stats later.
    const bool defer = stats != nullptr && (flags & TRR_FLAG_DEFER_STATS) != 0;
    const size_t rb_stride = RB_STRIDE;
    int pinned_slot = -1;
    if (stats && (rc = acquire_pinned(ctx, rb_stride * n_batches, &pinned_slot))) return rc;
    char* pinned_base = stats ? (char*)ctx->pinned[pinned_slot].p : nullptr;
    if (ctx->pending.empty()) ctx->events_used = 0;
    const size_t ev_first = ctx->events_used;
    const bool timed = stats != nullptr;
    size_t n_events = 0;
    std::vector<uint8_t> ev_kind;   // 0 trace, 1 shade
    auto ev_begin = [&](uint8_t kind, hipStream_t s_) {
        if (!timed) return;
        hipEvent_t e = get_event(ctx, ev_first + n_events);
        if (e) (void)hipEventRecord(e, s_);
        ++n_events;
        ev_kind.push_back(kind);
    };
    auto ev_end = [&](hipStream_t s_) {
        if (!timed) return;
        hipEvent_t e = get_event(ctx, ev_first + n_events);
        if (e) (void)hipEventRecord(e, s_);
        ++n_events;
    };

    // fork: with one lane everything stays on the caller's stream; with several, every lane stream waits for what
    // the caller's stream has queued so far (the strip clear) and the caller's stream later waits for every lane
    // From here on a failure must not return at once: the lane streams have to be joined onto the caller's stream, the
    // work sets' `done` events recorded and the pinned read-back slot released, whatever happened in between.  HIP errors
    // end the batch loop through `chk` and are reported after the join.
    int result = TRR_OK;
    auto chk = [&](hipError_t e_, const char* what) {
        if (e_ == hipSuccess) return true;
        if (result == TRR_OK || result == TRR_ERR_CANCELLED) result = fail(ctx, TRR_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e_));
        return false;
    };
    // camera-ray cull: this frame's tile mask, on the caller's stream, before the batches fork from it
    trr_ctx::CoverSet* cover_set = nullptr;
    if (camcull_on) {
        trr_ctx::CoverSet& cs = ctx->cover[ctx->cover_turn++ & 1u];
        const uint32_t tiles_x = (tl.w + 7) / 8, tiles_y = (tl.h + 7) / 8;
        if ((rc = ensure(ctx, cs.mask, (size_t)tiles_x * tiles_y)) || (rc = ensure(ctx, cs.big, (size_t)COVER_BIG_CAP * sizeof(uint4))) ||
            (rc = ensure(ctx, cs.words, 256))) {
            if (stats) ctx->pinned[pinned_slot].in_use = false;
            return rc;
        }
        CoverIO cio{};
        cio.mask = (uint8_t*)cs.mask.p; cio.tiles_x = tiles_x; cio.tiles_y = tiles_y;
        cio.flags = (uint32_t*)cs.words.p;
        cio.boxes = (const CoverBox*)sc->cover_boxes.p; cio.n_boxes = sc->n_cover;
        // where the samples of pixel x lie (mod.rs:233-240, 251-252): at x exactly without anti-aliasing; else from x - 1/s to
        // below x + 1, vertically to below y + (yi_max + 1)/s with yi_max = (samples - 1) / s; one more pixel on every side
        const float sq = (float)std::max(1u, fp.sqrt_spp);
        const float below = fp.aa ? 1.0f / sq : 0.0f, above_x = fp.aa ? 1.0f : 0.0f;
        const float above_y = fp.aa ? (float)((fp.spp - 1u) / std::max(1u, fp.sqrt_spp) + 1u) / sq : 0.0f;
        cio.pad_x_lo = above_x + 1.0f; cio.pad_x_hi = below + 1.0f;   // a box edge at px reaches the pixels x with x + above >= px, x - below <= px
        cio.pad_y_lo = above_y + 1.0f; cio.pad_y_hi = below + 1.0f;
        cio.big = (uint4*)cs.big.p; cio.n_big = (uint32_t*)cs.words.p + 1; cio.big_cap = COVER_BIG_CAP;
        if (chk(hipStreamWaitEvent(st, cs.free_ev, 0), "hipStreamWaitEvent") &&
            chk(hipMemsetAsync(cs.mask.p, 0, (size_t)tiles_x * tiles_y, st), "hipMemsetAsync") &&
            chk(hipMemsetAsync(cs.words.p, 0, 256, st), "hipMemsetAsync")) {
            launch_cover_mask(st, fp, cio);
            cover_set = &cs;
        }
    }
    const bool forked = pool > 1;
    if (forked) {
        // every lane stream waits for what the caller's stream has queued so far (the strip clear) AND for the previous
        // user of its work set, which may have been a single-lane frame on another caller stream
        if (chk(hipEventRecord(ctx->fork, st), "hipEventRecord"))
            for (uint32_t k = 0; k < lanes; ++k) {
                trr_ctx::WorkSet& w = ctx->ws[(lane_base + k) % pool];
                chk(hipStreamWaitEvent(w.stream, ctx->fork, 0), "hipStreamWaitEvent");
                chk(hipStreamWaitEvent(w.stream, w.done, 0), "hipStreamWaitEvent");
            }
    } else {
        // one work set, used directly on the caller's stream: order this frame after the set's previous user, which may
        // have been queued on another stream
        chk(hipStreamWaitEvent(st, ctx->ws[0].done, 0), "hipStreamWaitEvent");
    }

    uint32_t blocks_done = 0;
    uint32_t batches_run = 0;
    // lagged delivery: the batch that has been queued but not yet handed to the hook
    struct Queued { bool valid = false; uint32_t first = 0, n = 0, done_after = 0; hipStream_t stream = nullptr; } queued;
    auto deliver = [&](const Queued& q) -> int {
        hipError_t e_ = hipStreamSynchronize(q.stream);   // blocks of this batch are final: render.rs:143-149
        if (e_ != hipSuccess) return fail(ctx, TRR_ERR_DEVICE, std::string("hipStreamSynchronize: ") + hipGetErrorString(e_));
        if (hook && hook->after_batch) {
            int hr = hook->after_batch(q.first, q.n, q.stream);
            if (hr != TRR_OK) return hr;
        }
        if (progress) progress(user, q.done_after, n_owned);
        return TRR_OK;
    };
    for (uint32_t bi = 0; bi < n_batches && result == TRR_OK; ++bi) {
        if (cancel_flag && *cancel_flag) { result = TRR_ERR_CANCELLED; break; }   // render.rs:113
        trr_ctx::WorkSet& w = ctx->ws[(lane_base + bi % lanes) % pool];
        hipStream_t s_ = forked ? w.stream : st;
        BatchParams bp{};
        bp.first_owned = bi * blocks_per_batch;
        bp.n_blocks = std::min(blocks_per_batch, n_owned - bp.first_owned);
        bp.n_pixels = bp.n_blocks * bpix;
        bp.n_samples = bp.n_pixels * spp;
        bp.cap = cap;
        if (!chk(hipMemsetAsync(w.counters.p, 0, RB_STRIDE, s_), "hipMemsetAsync")) break;
        WorkView wv;
        wv.hits = (HitRec*)w.hits.p; wv.level = (float4*)w.level.p; wv.term = (float4*)w.term.p; wv.nlev = (uint8_t*)w.nlev.p; wv.color = (float*)w.color.p;
        wv.occ = (int32_t*)w.occ.p; wv.wqueue = (float4*)w.wqueue.p; wv.gqueue = (float4*)w.gqueue.p; wv.shadow_q = (float4*)w.shadow_q.p;
        wv.chain_q[0] = (float4*)w.chain_q[0].p; wv.chain_q[1] = (float4*)w.chain_q[1].p; wv.counters = w.counters.p; wv.spill = (uint2*)w.spill.p;
        wv.shadow_deep[0] = (float4*)w.shadow_deep[0].p; wv.shadow_deep[1] = (float4*)w.shadow_deep[1].p;
        wv.aux[0] = w.aux[0]; wv.aux[1] = w.aux[1];
        wv.sync_ev = w.sync_ev.empty() ? nullptr : w.sync_ev.data();
        if (cover_set) {
            wv.cover_mask = (const uint8_t*)cover_set->mask.p; wv.cover_flags = (const uint32_t*)cover_set->words.p; wv.groups = (uint32_t*)w.groups.p;
        }
        queue_batch(fp, bp, wv, grids, count, strip_dev, ids_obj_dev, ids_tri_dev, s_, ev_begin, ev_end);
        if (!chk(hipGetLastError(), "kernel launch")) break;
        if (stats && !chk(hipMemcpyAsync(pinned_base + rb_stride * bi, w.counters.p, rb_stride, hipMemcpyDeviceToHost, s_), "hipMemcpyAsync")) break;
        ++batches_run;
        blocks_done += bp.n_blocks;
        if (interactive) {
            Queued now;
            now.valid = true; now.first = bp.first_owned; now.n = bp.n_blocks; now.done_after = blocks_done; now.stream = s_;
            if (lagged) {   // hand over the previous batch while this one runs
                if (queued.valid) {
                    Queued q = queued;
                    queued = now;
                    int dr = deliver(q);
                    if (dr != TRR_OK) { result = dr; break; }
                } else {
                    queued = now;
                }
            } else {
                int dr = deliver(now);
                if (dr != TRR_OK) { result = dr; break; }
            }
        }
    }
    if (queued.valid) {
        if (result == TRR_OK) {
            int dr = deliver(queued);
            if (dr != TRR_OK) result = dr;
        } else {
            // cancelled (or failed) with a batch in flight: it is waited for and dropped, as the reference drops the blocks
            // that finish after the cancel (render.rs:154)
            (void)hipStreamSynchronize(queued.stream);
            blocks_done -= queued.n;
        }
    }
    if (forked) {   // join (also after a failure: whatever was queued on the lanes is ordered before the caller's next work)
        for (uint32_t k = 0; k < lanes; ++k) {
            trr_ctx::WorkSet& w = ctx->ws[(lane_base + k) % pool];
            if (chk(hipEventRecord(w.done, w.stream), "hipEventRecord")) chk(hipStreamWaitEvent(st, w.done, 0), "hipStreamWaitEvent");
        }
    } else {
        chk(hipEventRecord(ctx->ws[0].done, st), "hipEventRecord");
    }
    if (cover_set) chk(hipEventRecord(cover_set->free_ev, st), "hipEventRecord");   // the mask's next user waits for this frame
    if (stats && result != TRR_OK && result != TRR_ERR_CANCELLED) {
        ctx->pinned[pinned_slot].in_use = false;   // a failed frame leaves no statistics pending
    } else if (stats) {
        trr_ctx::PendingStats f;
        f.out = stats;
        f.ev_first = ev_first; f.ev_count = n_events; f.ev_kind = ev_kind;
        f.pinned_slot = pinned_slot;
        f.batches = batches_run; f.blocks_per_batch = blocks_per_batch; f.n_owned = n_owned; f.bpix = bpix; f.spp = spp;
        f.max_depth = max_depth; f.blocks_done = blocks_done;
        f.caller_stream = st;
        if (defer) {
            ctx->events_used = ev_first + n_events;
            ctx->pending.push_back(std::move(f));
        } else {
            hipError_t e_ = hipStreamSynchronize(st);
            if (e_ == hipSuccess) {
                finish_stats(ctx, f);
                stats->ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
            }
            ctx->pinned[pinned_slot].in_use = false;
            if (e_ != hipSuccess) return fail(ctx, TRR_ERR_DEVICE, std::string("hipStreamSynchronize: ") + hipGetErrorString(e_));
        }
    }
    if (result == TRR_ERR_CANCELLED) return fail(ctx, result, "render cancelled");
    return result;
}

int trr_render_shard_device(trr_ctx* ctx, trr_scene* scene, float delta, uint32_t max_depth, uint32_t flags, uint32_t shard,
                            uint32_t n_shards, void* strip_dev, size_t strip_bytes, void* stream,
                            const volatile int32_t* cancel_flag, trr_progress_cb progress, void* user, trr_stats* stats_out) {
    if (!ctx) return fail(nullptr, TRR_ERR_INVALID, "trr_render_shard_device: ctx is NULL");
    if (is_multi(ctx)) return fail(ctx, TRR_ERR_INVALID, "trr_render_shard_device: a multi-device context shards by itself (use trr_render / trr_render_blocks)");
    DeviceGuard guard;
    return render_shard_impl(ctx, scene, delta, max_depth, flags, shard, n_shards, (uint8_t*)strip_dev, strip_bytes,
                             stream ? (hipStream_t)stream : ctx->stream, nullptr, nullptr, cancel_flag, progress, user, stats_out);
}

// ---- the entry point that allocates nothing ------------------------------------------------------------------------------
// Work set of `cap` samples carved out of a flat byte range: returns the bytes needed; fills *w when base != nullptr.
static size_t carve_work_set(unsigned char* base, uint32_t cap, uint32_t D, uint32_t L, uint32_t schedule, uint32_t trace_grid,
                             WorkView* w) {
    size_t off = 0;
    auto take = [&](size_t bytes) {
        unsigned char* p = base ? base + off : nullptr;
        off += align256(bytes);
        return p;
    };
    const uint32_t Lq = std::max(1u, L);
    unsigned char* counters = take(RB_STRIDE);
    unsigned char* spill = take(trace_spill_bytes(trace_grid));
    unsigned char* wqueue = schedule == SCHED_PERSISTENT ? take(trace_queue_bytes(trace_grid, queue_capacity(L))) : nullptr;
    unsigned char* hits = take((size_t)cap * sizeof(HitRec));
    unsigned char* level = take((size_t)cap * 16 * LEVEL_F4 * D);
    unsigned char* term = take((size_t)cap * 16);
    unsigned char* nlev = take((size_t)cap);
    unsigned char* color = take((size_t)cap * 12);
    unsigned char* occ = take((size_t)cap * 4 * Lq * D);
    unsigned char* shadow_q = schedule == SCHED_LEVELS ? take((size_t)cap * 32 * Lq) : nullptr;
    unsigned char* chain0 = schedule == SCHED_LEVELS ? take((size_t)cap * 32) : nullptr;
    unsigned char* chain1 = schedule == SCHED_LEVELS ? take((size_t)cap * 32) : nullptr;
    if (w) {
        w->counters = counters; w->spill = (uint2*)spill; w->wqueue = (float4*)wqueue; w->gqueue = nullptr; w->hits = (HitRec*)hits;
        w->level = (float4*)level; w->term = (float4*)term; w->nlev = (uint8_t*)nlev; w->color = (float*)color; w->occ = (int32_t*)occ;
        w->shadow_q = (float4*)shadow_q; w->chain_q[0] = (float4*)chain0; w->chain_q[1] = (float4*)chain1;
    }
    return off;
}

size_t trr_noalloc_scratch_bytes(void) { return SCRATCH_BYTES; }

int trr_render_shard_device_noalloc(trr_ctx* ctx, trr_scene* sc, float delta, uint32_t max_depth, uint32_t shard, uint32_t n_shards,
                                    void* strip_dev, size_t strip_bytes, void* stream) {
    if (!ctx || !sc) return fail(ctx, TRR_ERR_INVALID, "trr_render_shard_device_noalloc: NULL context or scene");
    if (is_multi(ctx)) return fail(ctx, TRR_ERR_INVALID, "trr_render_shard_device_noalloc: needs a single-device context");
    DeviceGuard guard;
    if (sc->ctx != ctx) return fail(ctx, TRR_ERR_INVALID, "trr_render_shard_device_noalloc: scene belongs to another context");
    if (!sc->have_camera || !sc->have_settings) return fail(ctx, TRR_ERR_INVALID, "trr_render_shard_device_noalloc: camera/settings not set");
    if (!(delta >= 0.0f)) return fail(ctx, TRR_ERR_INVALID, "trr_render_shard_device_noalloc: delta must be >= 0");
    if (max_depth > 254) return fail(ctx, TRR_ERR_INVALID, "trr_render_shard_device_noalloc: max_depth above 254");
    if (n_shards == 0 || shard >= n_shards) return fail(ctx, TRR_ERR_INVALID, "trr_render_shard_device_noalloc: bad shard / n_shards");
    if (!ctx->scratch) return fail(ctx, TRR_ERR_DEVICE, "trr_render_shard_device_noalloc: the scratch array has no device address");
    Tiling tl;
    if (!tiling_of(sc->camera.width, sc->camera.height, &tl))
        return fail(ctx, TRR_ERR_RESOLUTION, "trr_render_shard_device_noalloc: width and height must be >= 20 (render.rs:109)");
    const size_t need_strip = (size_t)max_owned_blocks(tl, n_shards) * tl.bw * tl.bh * 3;
    if (!strip_dev || strip_bytes < need_strip) return fail(ctx, TRR_ERR_INVALID, "trr_render_shard_device_noalloc: strip buffer too small");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    if (!ctx->scratch_ev) return fail(ctx, TRR_ERR_DEVICE, "trr_render_shard_device_noalloc: no ordering event for this device's scratch array");

    const bool aa = sc->settings.anti_aliasing != 0;
    const uint32_t spp = aa ? sc->settings.samples : 1u;
    const uint32_t L = sc->n_lights, D = std::max(1u, max_depth);
    const uint32_t n_owned = owned_blocks(tl, shard, n_shards);
    const uint64_t spb = (uint64_t)tl.bw * tl.bh * spp;   // samples per block
    HIPCHK(ctx, hipMemsetAsync(strip_dev, 0, need_strip, st));
    if (spp == 0 || n_owned == 0) return TRR_OK;   // mod.rs:228-249 with samples == 0: NaN mean -> 0 (render.rs:136)

    // two resident blocks per CU for every trace launch: the stack-spill area (and, in the persistent schedule, the ray
    // queues) of the whole grid must fit the scratch array too; no follow-up launches (no spill queues)
    const uint32_t schedule = choose_schedule((uint64_t)n_owned * spb, L, max_depth);
    const uint32_t n_cu = (uint32_t)std::max(1, ctx->n_cu);
    BatchGrids grids{};
    grids.schedule = schedule;
    grids.trace = n_cu * (uint32_t)std::min(2, ctx->bpc[0]);
    grids.raygen = grids.secondary = n_cu * (uint32_t)std::min(2, std::min(ctx->bpc_level[0][0], ctx->bpc_level[1][0]));
    grids.chain = grids.secondary;
    grids.aux_max = grids.aux_deep = n_cu * 8;
    grids.followups = 0;
    grids.drain_level = NO_DRAIN;   // (the drain's ray queues would take scratch from the batch)
    grids.levels = sc->any_reflective ? max_depth : std::min(max_depth, 1u);
    {
        const char* ce = std::getenv("TRR_CULL");
        grids.cull = (ce && ce[0] == '0') ? 0u : 1u;
    }
    grids.chunk_min[0] = 64; grids.chunk_min[1] = 16;
    const uint32_t spill_grid = schedule == SCHED_PERSISTENT ? grids.trace : grids.raygen;
    // the largest whole number of blocks per batch whose work set fits the scratch array
    const size_t fixed = carve_work_set(nullptr, 0, D, L, schedule, spill_grid, nullptr);
    const size_t per_64 = carve_work_set(nullptr, 64, D, L, schedule, spill_grid, nullptr) - fixed;   // an upper bound per 64 samples (alignment included)
    if (fixed >= SCRATCH_BYTES || spb > (1ull << 25))
        return fail(ctx, TRR_ERR_INVALID, "trr_render_shard_device_noalloc: one block of this frame exceeds the static scratch array");
    const uint64_t max_samples = (uint64_t)((SCRATCH_BYTES - fixed) / per_64) * 64;
    uint32_t blocks_per_batch = (uint32_t)std::min<uint64_t>(std::min<uint64_t>(max_samples, 1ull << 25) / spb, n_owned);
    if (blocks_per_batch == 0)
        return fail(ctx, TRR_ERR_INVALID,
                    "trr_render_shard_device_noalloc: one block (" + std::to_string(spb) + " samples at " + std::to_string(L) + " lights, max_depth " +
                        std::to_string(max_depth) + ") needs more than the " + std::to_string(SCRATCH_BYTES >> 20) +
                        " MiB static scratch array holds (" + std::to_string(max_samples) + " samples); use trr_render_shard_device");
    const uint32_t cap = (uint32_t)(((uint64_t)blocks_per_batch * spb + 63) & ~63ull);
    WorkView wv;
    if (carve_work_set(ctx->scratch, cap, D, L, schedule, spill_grid, &wv) > SCRATCH_BYTES)
        return fail(ctx, TRR_ERR_INVALID, "trr_render_shard_device_noalloc: work set does not fit the scratch array");

    const FrameParams fp = frame_params_of(sc, tl, shard, n_shards, delta, max_depth);
    const uint32_t bpix = tl.bw * tl.bh;
    auto no_ev_begin = [](uint8_t, hipStream_t) {};
    auto no_ev_end = [](hipStream_t) {};
    // batches run one after the other on the caller's stream and share the one work set; the array keeps the last call's
    // contents, so what must start at zero (queue sizes, work counters) is cleared before every batch.  The array's previous
    // user on this GPU (any stream, any context) is waited for first, on the device.
    {
        std::lock_guard<std::mutex> lock(g_scratch_mu);   // wait + record of one call are not interleaved with another call's
        HIPCHK(ctx, hipStreamWaitEvent(st, ctx->scratch_ev, 0));
    for (uint32_t first = 0; first < n_owned; first += blocks_per_batch) {
        BatchParams bp{};
        bp.first_owned = first;
        bp.n_blocks = std::min(blocks_per_batch, n_owned - first);
        bp.n_pixels = bp.n_blocks * bpix;
        bp.n_samples = bp.n_pixels * spp;
        bp.cap = cap;
        HIPCHK(ctx, hipMemsetAsync(wv.counters, 0, RB_STRIDE, st));
        queue_batch(fp, bp, wv, grids, false, (uint8_t*)strip_dev, nullptr, nullptr, st, no_ev_begin, no_ev_end);
        HIPCHK(ctx, hipGetLastError());
    }
        HIPCHK(ctx, hipEventRecord(ctx->scratch_ev, st));
    }
    return TRR_OK;
}

int trr_assemble_device(trr_ctx* ctx, uint32_t width, uint32_t height, uint32_t n_shards, const void* strips_dev, void* frame_dev,
                        void* stream) {
    if (!ctx || !strips_dev || !frame_dev) return fail(ctx, TRR_ERR_INVALID, "trr_assemble_device: NULL argument");
    if (is_multi(ctx)) ctx = ctx->subs[0];   // the root device assembles
    DeviceGuard guard;
    Tiling t;
    if (n_shards == 0) return fail(ctx, TRR_ERR_INVALID, "trr_assemble_device: n_shards is 0");
    if (!tiling_of(width, height, &t)) return fail(ctx, TRR_ERR_RESOLUTION, "trr_assemble_device: width and height must be >= 20");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    launch_assemble(stream ? (hipStream_t)stream : ctx->stream, width, height, t.bw, t.bh, t.nbx, t.nby, n_shards,
                    trr_strip_bytes(width, height, n_shards), (const uint8_t*)strips_dev, (uint8_t*)frame_dev);
    HIPCHK(ctx, hipGetLastError());
    return TRR_OK;
}

int trr_render_blocks(trr_ctx* ctx, trr_scene* scene, float delta, uint32_t max_depth, uint32_t flags, uint32_t blocks_per_batch,
                      const volatile int32_t* cancel_flag, trr_block_cb sink, trr_progress_cb progress, void* user, trr_stats* stats_out) {
    if (!ctx || !scene || !sink) return fail(ctx, TRR_ERR_INVALID, "trr_render_blocks: NULL argument");
    if (!scene->have_camera) return fail(ctx, TRR_ERR_INVALID, "trr_render_blocks: camera not set");
    const uint32_t w = scene->camera.width, h = scene->camera.height;
    Tiling t;
    if (!tiling_of(w, h, &t)) return fail(ctx, TRR_ERR_RESOLUTION, "trr_render_blocks: width and height must be >= 20 (render.rs:109)");
    DeviceGuard guard;
    if (is_multi(ctx) && ctx->subs.size() > 1)
        return multi_render_blocks(ctx, scene, delta, max_depth, flags & ~(TRR_FLAG_DEFER_STATS | TRR_FLAG_PIPELINED), blocks_per_batch, cancel_flag,
                                   sink, progress, user, stats_out);
    if (is_multi(ctx)) { ctx = ctx->subs[0]; scene = scene->replicas[0]; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t sb = trr_strip_bytes(w, h, 1), block_bytes = (size_t)t.bw * t.bh * 3;
    int rc;
    if ((rc = ensure(ctx, ctx->strip, sb))) return rc;
    BatchHook hook;
    hook.blocks_per_batch = blocks_per_batch ? blocks_per_batch : t.nbx;   // default: one row of blocks per batch
    hook.blocks_per_batch = std::min(hook.blocks_per_batch, t.nblocks);
    int stage_slot = -1;
    if ((rc = acquire_pinned(ctx, block_bytes * hook.blocks_per_batch, &stage_slot))) return rc;
    uint8_t* stage = static_cast<uint8_t*>(ctx->pinned[stage_slot].p);
    const uint8_t* strip = static_cast<const uint8_t*>(ctx->strip.p);
    hook.after_batch = [&](uint32_t first, uint32_t n, hipStream_t s) -> int {
        hipError_t e = hipMemcpyAsync(stage, strip + (size_t)first * block_bytes, (size_t)n * block_bytes, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) return fail(ctx, TRR_ERR_DEVICE, std::string("trr_render_blocks: block read-back: ") + hipGetErrorString(e));
        for (uint32_t j = 0; j < n; ++j) {   // shard 0 of 1: the j-th owned block is block first + j of the row-major grid
            const uint32_t b = first + j;
            sink(user, b % t.nbx, b / t.nbx, t.bw, t.bh, stage + (size_t)j * block_bytes);
        }
        return TRR_OK;
    };
    int rres = render_shard_impl(ctx, scene, delta, max_depth, flags & ~(TRR_FLAG_DEFER_STATS | TRR_FLAG_PIPELINED), 0, 1,
                                 (uint8_t*)ctx->strip.p, sb, ctx->stream, nullptr, nullptr, cancel_flag, progress, user, stats_out, &hook);
    ctx->pinned[stage_slot].in_use = false;
    return rres;
}

int trr_render(trr_ctx* ctx, trr_scene* scene, float delta, uint32_t max_depth, uint32_t flags, uint8_t* rgb_out,
               uint32_t* hit_obj_out, uint32_t* hit_tri_out, const volatile int32_t* cancel_flag, trr_progress_cb progress,
               void* user, trr_stats* stats_out) {
    if (!ctx || !scene || !rgb_out) return fail(ctx, TRR_ERR_INVALID, "trr_render: NULL argument");
    if (!scene->have_camera) return fail(ctx, TRR_ERR_INVALID, "trr_render: camera not set");
    DeviceGuard guard;
    if (is_multi(ctx)) {
        if (hit_obj_out || hit_tri_out) {
            if (ctx->subs.size() > 1) return fail(ctx, TRR_ERR_INVALID, "trr_render: hit-id buffers need a single-device context");
            return trr_render(ctx->subs[0], scene->replicas[0], delta, max_depth, flags, rgb_out, hit_obj_out, hit_tri_out, cancel_flag, progress, user, stats_out);
        }
        return multi_render(ctx, scene, delta, max_depth, flags & ~(TRR_FLAG_DEFER_STATS | TRR_FLAG_PIPELINED), rgb_out, cancel_flag, progress, user, stats_out);
    }
    const uint32_t w = scene->camera.width, h = scene->camera.height;
    Tiling t;
    if (!tiling_of(w, h, &t)) return fail(ctx, TRR_ERR_RESOLUTION, "trr_render: width and height must be >= 20 (render.rs:109)");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t sb = trr_strip_bytes(w, h, 1), fb = (size_t)w * h * 3;
    int rc;
    if ((rc = ensure(ctx, ctx->strip, sb)) || (rc = ensure(ctx, ctx->frame, fb))) return rc;
    const bool ids = hit_obj_out && hit_tri_out;
    if (ids) {
        if ((rc = ensure(ctx, ctx->ids_obj, (size_t)w * h * 4)) || (rc = ensure(ctx, ctx->ids_tri, (size_t)w * h * 4))) return rc;
        HIPCHK(ctx, hipMemsetAsync(ctx->ids_obj.p, 0xFF, (size_t)w * h * 4, ctx->stream));
        HIPCHK(ctx, hipMemsetAsync(ctx->ids_tri.p, 0xFF, (size_t)w * h * 4, ctx->stream));
    }
    int rres = render_shard_impl(ctx, scene, delta, max_depth, flags, 0, 1, (uint8_t*)ctx->strip.p, sb, ctx->stream,
                                 ids ? (uint32_t*)ctx->ids_obj.p : nullptr, ids ? (uint32_t*)ctx->ids_tri.p : nullptr, cancel_flag,
                                 progress, user, stats_out);
    if (rres != TRR_OK && rres != TRR_ERR_CANCELLED) return rres;
    std::string keep = ctx->err;
    if ((rc = trr_assemble_device(ctx, w, h, 1, ctx->strip.p, ctx->frame.p, ctx->stream))) return rc;
    HIPCHK(ctx, hipMemcpyAsync(rgb_out, ctx->frame.p, fb, hipMemcpyDeviceToHost, ctx->stream));
    if (ids) {
        HIPCHK(ctx, hipMemcpyAsync(hit_obj_out, ctx->ids_obj.p, (size_t)w * h * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(hit_tri_out, ctx->ids_tri.p, (size_t)w * h * 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (rres == TRR_ERR_CANCELLED) return fail(ctx, rres, keep);
    return TRR_OK;
}

}   // extern "C"

// ---- multi-device contexts --------------------------------------------------------------------------------------------------
// One process, one render thread (the reference's host: render.rs:39-71, statusbar.rs:142-157) and N GPUs: the context owns a
// plain context per device, every scene is uploaded to each of them, a frame's blocks are dealt round-robin (block id % N ==
// device index, interleaved so that sky / object imbalance averages out), every device renders its blocks into a compact strip
// and ONE ncclGather (RCCL over xGMI) brings the strips to the first device, which de-interleaves them into the frame.
// RCCL is loaded on demand (dlopen), so that a single-GPU host -- and the static archive a build.rs links -- needs no librccl.
namespace {
struct Rccl {
    void* handle = nullptr;
    int (*CommInitAll)(void** comms, int ndev, const int* devlist) = nullptr;
    int (*CommDestroy)(void* comm) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Gather)(const void* send, void* recv, size_t count, int datatype, int root, void* comm, hipStream_t stream) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string error;
};
constexpr int kNcclUint8 = 1;   // ncclDataType_t (rccl.h)

Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // the copy a host process already holds (PyTorch-ROCm bundles one) first, so that one RCCL serves the whole process
        const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names)
            if ((r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
        for (const char* n : names) {
            if (r.handle) break;
            r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        }
        if (!r.handle) { r.error = std::string("RCCL is not loadable: ") + dlerror(); return; }
        auto sym = [&](const char* name) {
            void* p = dlsym(r.handle, name);
            if (!p && r.error.empty()) r.error = std::string("RCCL lacks ") + name;
            return p;
        };
        r.CommInitAll = (decltype(r.CommInitAll))sym("ncclCommInitAll");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
        r.Gather = (decltype(r.Gather))sym("ncclGather");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    });
    return &r;
}
}   // namespace

static void multi_destroy(trr_ctx* ctx) {
    Rccl* r = rccl();
    for (void* c : ctx->comms)
        if (c && r->CommDestroy) (void)r->CommDestroy(c);
    if (!ctx->subs.empty()) {
        (void)hipSetDevice(ctx->subs[0]->device);
        release(ctx->gathered);
    }
    for (trr_ctx* c : ctx->subs) trr_destroy(c);
    delete ctx;
}

static int multi_scene_create(trr_ctx* ctx, const trr_object_desc* objects, uint32_t n_objects, const trr_image_rgb8* textures,
                              uint32_t n_textures, uint32_t build, trr_scene** out) {
    trr_scene* sc = new trr_scene();
    sc->ctx = ctx;
    sc->device = ctx->subs[0]->device;
    // replicated scene: the same upload on every device (devices build / receive their copies one after the other)
    for (trr_ctx* c : ctx->subs) {
        trr_scene* r = nullptr;
        int rc = trr_scene_create_ex(c, objects, n_objects, textures, n_textures, build, &r);
        if (rc != TRR_OK) {
            std::string msg = c->err;
            trr_scene_destroy(sc);
            return fail(ctx, rc, msg);
        }
        sc->replicas.push_back(r);
    }
    sc->n_objs = n_objects;
    *out = sc;
    return TRR_OK;
}

// One frame on all devices of the context: shard renders (one host thread per device while cancel / progress need polling,
// plain asynchronous queueing otherwise), one gather, assemble and read-back on the first device.
static int multi_render(trr_ctx* ctx, trr_scene* scene, float delta, uint32_t max_depth, uint32_t flags, uint8_t* rgb_out,
                        const volatile int32_t* cancel_flag, trr_progress_cb progress, void* user, trr_stats* stats_out) {
    if (scene->ctx != ctx || scene->replicas.size() != ctx->subs.size()) return fail(ctx, TRR_ERR_INVALID, "render: scene belongs to another context");
    const uint32_t n = (uint32_t)ctx->subs.size();
    const uint32_t w = scene->camera.width, h = scene->camera.height;
    Tiling t;
    if (!tiling_of(w, h, &t)) return fail(ctx, TRR_ERR_RESOLUTION, "trr_render: width and height must be >= 20 (render.rs:109)");
    const auto wall0 = std::chrono::steady_clock::now();
    const size_t sb = trr_strip_bytes(w, h, n), fb = (size_t)w * h * 3;
    trr_ctx* root = ctx->subs[0];
    int rc;
    for (trr_ctx* c : ctx->subs) {
        HIPCHK(ctx, hipSetDevice(c->device));
        if ((rc = ensure(c, c->strip, sb))) return fail(ctx, rc, c->err);
    }
    HIPCHK(ctx, hipSetDevice(root->device));
    if ((rc = ensure(root, root->frame, fb))) return fail(ctx, rc, root->err);
    // Strip exchange: one RCCL gather (default), or -- TRR_MULTI_TRANSPORT=copy, and always when a device is listed twice
    // (rehearsals on one GPU: RCCL refuses duplicate devices) -- one peer copy per device into the root's gather buffer.
    // The RCCL path is three calls -- ncclCommInitAll, grouped ncclGather, ncclCommDestroy -- and has only ever run with one rank
    // (no multi-GPU node was available to this build): any error it reports is logged and answered with the peer-copy exchange,
    // for this frame and for the rest of the context's life, never with a failed render.  TRR_MULTI_INJECT_RCCL_FAILURE=init|gather
    // makes the corresponding call "fail" (tests of that branch).
    const char* e_tr = std::getenv("TRR_MULTI_TRANSPORT");
    const char* e_inject = std::getenv("TRR_MULTI_INJECT_RCCL_FAILURE");
    const bool force_rccl = std::getenv("TRR_MULTI_FORCE_RCCL") != nullptr;
    const bool want_copy = ctx->duplicate_devices || (e_tr && e_tr[0] == 'c');
    bool use_rccl = !ctx->rccl_failed && !want_copy && (n > 1 || force_rccl);
    auto rccl_fallback = [&](const std::string& why) {
        std::fprintf(stderr, "trayracer_hip: warning: %s; exchanging the strips by peer copies instead\n", why.c_str());
        ctx->rccl_failed = true;
        use_rccl = false;
    };
    if (use_rccl && ctx->comms.empty()) {
        Rccl* r = rccl();
        if (!r->error.empty()) rccl_fallback(r->error);
        else {
            std::vector<int> devs;
            for (trr_ctx* c : ctx->subs) devs.push_back(c->device);
            std::vector<void*> comms(n, nullptr);
            int e = (e_inject && e_inject[0] == 'i') ? -1 : r->CommInitAll(comms.data(), (int)n, devs.data());
            if (e != 0) rccl_fallback(std::string("ncclCommInitAll: ") + (e < 0 ? "injected failure" : r->GetErrorString(e)));
            else ctx->comms = comms;
        }
    }
    // peer copies: asked for, needed (duplicates), or the answer to an RCCL error (also with one device, where the "peer" is itself)
    bool use_copy = !use_rccl && (n > 1 || (force_rccl && ctx->rccl_failed));
    if ((use_copy || use_rccl) && (rc = ensure(root, ctx->gathered, sb * n))) return fail(ctx, rc, root->err);

    // ---- shard renders ----
    std::vector<trr_stats> st(n);
    std::vector<int> res(n, TRR_OK);
    const bool polled = cancel_flag != nullptr || progress != nullptr;
    std::mutex mu;
    std::vector<uint32_t> done(n, 0);
    struct ProgressCtx { std::mutex* mu; std::vector<uint32_t>* done; uint32_t dev, total; trr_progress_cb cb; void* user; };
    auto on_progress = +[](void* u, uint32_t blocks_done, uint32_t) {
        ProgressCtx* p = static_cast<ProgressCtx*>(u);
        std::lock_guard<std::mutex> lock(*p->mu);   // the caller's callback runs on one thread at a time (render.rs:143-149)
        (*p->done)[p->dev] = blocks_done;
        uint32_t sum = 0;
        for (uint32_t v : *p->done) sum += v;
        p->cb(p->user, sum, p->total);
    };
    std::vector<ProgressCtx> pc(n);
    auto render_one = [&](uint32_t d) {
        trr_ctx* c = ctx->subs[d];
        if (hipSetDevice(c->device) != hipSuccess) { res[d] = fail(c, TRR_ERR_DEVICE, "hipSetDevice failed"); return; }
        pc[d] = ProgressCtx{&mu, &done, d, t.nblocks, progress, user};
        res[d] = render_shard_impl(c, scene->replicas[d], delta, max_depth, flags, d, n, (uint8_t*)c->strip.p, sb, c->stream, nullptr, nullptr,
                                   cancel_flag, progress ? on_progress : nullptr, &pc[d], stats_out ? &st[d] : nullptr);
    };
    (void)polled;
    if (n > 1) {
        // one host thread per device, also when nothing is polled: queueing a shard's launches takes ~0.12 ms of host time
        // (measured, an eighth of the 4K x 16 spp frame), which from one thread would start the eighth device 0.85 ms late
        std::vector<std::thread> th;
        for (uint32_t d = 0; d < n; ++d) th.emplace_back(render_one, d);
        for (auto& x : th) x.join();
    } else {
        for (uint32_t d = 0; d < n; ++d) render_one(d);
    }
    int result = TRR_OK;
    for (uint32_t d = 0; d < n; ++d)
        if (res[d] == TRR_ERR_CANCELLED && result == TRR_OK) result = TRR_ERR_CANCELLED;
        else if (res[d] != TRR_OK && res[d] != TRR_ERR_CANCELLED) result = res[d];
    if (result != TRR_OK && result != TRR_ERR_CANCELLED) {
        // a shard failed: the other devices may still be writing their strips -- wait for all of them, then report every
        // failed device's message
        std::string msg;
        for (uint32_t d = 0; d < n; ++d) {
            trr_ctx* c = ctx->subs[d];
            if (hipSetDevice(c->device) == hipSuccess) (void)hipStreamSynchronize(c->stream);
            if (res[d] != TRR_OK && res[d] != TRR_ERR_CANCELLED) msg += (msg.empty() ? "device " : "; device ") + std::to_string(c->device) + ": " + c->err;
        }
        return fail(ctx, result, msg);
    }

    // ---- the single exchange of the path: finished RGB8 strips to the first device ----
    const void* strips = root->strip.p;
    if (use_rccl) {
        Rccl* r = rccl();
        std::string why;
        int e = (e_inject && e_inject[0] == 'g') ? -1 : r->GroupStart();
        const bool opened = e == 0;
        for (uint32_t d = 0; d < n && e == 0; ++d) {
            trr_ctx* c = ctx->subs[d];
            if (hipSetDevice(c->device) != hipSuccess) { why = "hipSetDevice failed before ncclGather"; e = -2; break; }
            e = r->Gather(c->strip.p, d == 0 ? ctx->gathered.p : nullptr, sb, kNcclUint8, 0, ctx->comms[d], c->stream);
        }
        if (opened) {   // a group that was opened is always closed, whatever happened inside it
            const int e2 = r->GroupEnd();
            if (e == 0) e = e2;
        }
        if (e != 0) {
            // nothing of a failed group can be trusted: drain every stream, then exchange the (still intact) strips by copies
            for (trr_ctx* c : ctx->subs)
                if (hipSetDevice(c->device) == hipSuccess) (void)hipStreamSynchronize(c->stream);
            rccl_fallback(why.empty() ? std::string("ncclGather: ") + (e < 0 ? "injected failure" : r->GetErrorString(e)) : why);
            use_copy = true;
        } else strips = ctx->gathered.p;
    }
    if (use_copy) {
        for (uint32_t d = 0; d < n; ++d) {
            trr_ctx* c = ctx->subs[d];
            HIPCHK(ctx, hipSetDevice(c->device));
            HIPCHK(ctx, hipMemcpyPeerAsync((uint8_t*)ctx->gathered.p + (size_t)d * sb, root->device, c->strip.p, c->device, sb, c->stream));
            HIPCHK(ctx, hipEventRecord(c->fork, c->stream));
            HIPCHK(ctx, hipStreamWaitEvent(root->stream, c->fork, 0));   // (a no-op for the root's own stream)
        }
        strips = ctx->gathered.p;
    }
    HIPCHK(ctx, hipSetDevice(root->device));
    launch_assemble(root->stream, w, h, t.bw, t.bh, t.nbx, t.nby, n, sb, (const uint8_t*)strips, (uint8_t*)root->frame.p);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(rgb_out, root->frame.p, fb, hipMemcpyDeviceToHost, root->stream));
    HIPCHK(ctx, hipStreamSynchronize(root->stream));
    for (uint32_t d = 1; d < n; ++d) {   // the other devices' gather sends have completed with the root's receive; make it explicit
        HIPCHK(ctx, hipSetDevice(ctx->subs[d]->device));
        HIPCHK(ctx, hipStreamSynchronize(ctx->subs[d]->stream));
    }
    if (stats_out) {
        trr_stats sum{};
        for (const trr_stats& a : st) add_stats(sum, a);
        sum.ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
        *stats_out = sum;
    }
    if (result == TRR_ERR_CANCELLED) return fail(ctx, result, "render cancelled");
    return TRR_OK;
}

static void add_stats(trr_stats& sum, const trr_stats& a) {
    sum.rays_primary += a.rays_primary; sum.rays_shadow += a.rays_shadow; sum.rays_reflection += a.rays_reflection;
    sum.rays_march += a.rays_march; sum.node_visits += a.node_visits; sum.tri_tests += a.tri_tests; sum.box_tests += a.box_tests;
    sum.ms_trace += a.ms_trace; sum.ms_shade += a.ms_shade; sum.trace_launches += a.trace_launches;
    sum.blocks_rendered += a.blocks_rendered; sum.wave_node_steps += a.wave_node_steps; sum.wave_tri_steps += a.wave_tri_steps;
    sum.wave_outer_steps += a.wave_outer_steps; sum.wave_lane_steps += a.wave_lane_steps; sum.wave_trans_steps += a.wave_trans_steps;
    sum.trans_lanes += a.trans_lanes; sum.wave_uniform_node_steps += a.wave_uniform_node_steps; sum.queue_overflows += a.queue_overflows;
    sum.rays_shadow_culled += a.rays_shadow_culled;
    sum.rays_primary_culled += a.rays_primary_culled;
}

// trr_render_blocks on several devices: the reference's sink sees blocks as they finish, in whatever order its worker threads
// finish them (render.rs:109-113 `par_bridge`, :151-186).  Every device renders its blocks (block id % N == device index) in
// batches on a host thread of its own; a finished batch is read back into that device's pinned strip copy and announced on a
// queue, which the CALLING thread drains in arrival order: `sink` runs once per block, on the calling thread only, while the
// devices go on rendering.  No gather and no assembly: the blocks go to the host anyway.  `progress` runs on the calling
// thread too, after each delivered batch.  A cancel ends every device's loop at its next batch; blocks that were announced
// before are still delivered (they are final), the batches in flight are dropped (render.rs:154).
static int multi_render_blocks(trr_ctx* ctx, trr_scene* scene, float delta, uint32_t max_depth, uint32_t flags, uint32_t blocks_per_batch,
                               const volatile int32_t* cancel_flag, trr_block_cb sink, trr_progress_cb progress, void* user, trr_stats* stats_out) {
    if (scene->ctx != ctx || scene->replicas.size() != ctx->subs.size()) return fail(ctx, TRR_ERR_INVALID, "render: scene belongs to another context");
    const uint32_t n = (uint32_t)ctx->subs.size();
    const uint32_t w = scene->camera.width, h = scene->camera.height;
    Tiling t;
    if (!tiling_of(w, h, &t)) return fail(ctx, TRR_ERR_RESOLUTION, "trr_render_blocks: width and height must be >= 20 (render.rs:109)");
    const auto wall0 = std::chrono::steady_clock::now();
    const size_t sb = trr_strip_bytes(w, h, n), block_bytes = (size_t)t.bw * t.bh * 3;
    // a row of the block grid per round of batches by default, split over the devices
    const uint32_t per_frame = blocks_per_batch ? blocks_per_batch : t.nbx;
    const uint32_t per_device = std::max(1u, (std::min(per_frame, t.nblocks) + n - 1) / n);
    int rc;
    std::vector<int> stage_slot(n, -1);
    for (uint32_t d = 0; d < n; ++d) {
        trr_ctx* c = ctx->subs[d];
        HIPCHK(ctx, hipSetDevice(c->device));
        if ((rc = ensure(c, c->strip, sb)) || (rc = acquire_pinned(c, sb, &stage_slot[d]))) {
            for (uint32_t k = 0; k < d; ++k) ctx->subs[k]->pinned[stage_slot[k]].in_use = false;
            return fail(ctx, rc, c->err);
        }
    }
    struct Item { uint32_t dev, first, count; };
    std::mutex mu;
    std::condition_variable cv;
    std::vector<Item> queue;
    uint32_t running = n;
    std::vector<trr_stats> st(n);
    std::vector<int> res(n, TRR_OK);
    auto render_one = [&](uint32_t d) {
        trr_ctx* c = ctx->subs[d];
        if (hipSetDevice(c->device) != hipSuccess) res[d] = fail(c, TRR_ERR_DEVICE, "hipSetDevice failed");
        else {
            uint8_t* stage = static_cast<uint8_t*>(c->pinned[stage_slot[d]].p);
            BatchHook hook;
            hook.blocks_per_batch = per_device;
            hook.after_batch = [&, c, d, stage](uint32_t first, uint32_t count, hipStream_t s_) -> int {
                hipError_t e = hipMemcpyAsync(stage + (size_t)first * block_bytes, (const uint8_t*)c->strip.p + (size_t)first * block_bytes,
                                              (size_t)count * block_bytes, hipMemcpyDeviceToHost, s_);
                if (e == hipSuccess) e = hipStreamSynchronize(s_);
                if (e != hipSuccess) return fail(c, TRR_ERR_DEVICE, std::string("trr_render_blocks: block read-back: ") + hipGetErrorString(e));
                { std::lock_guard<std::mutex> lock(mu); queue.push_back(Item{d, first, count}); }
                cv.notify_one();
                return TRR_OK;
            };
            res[d] = render_shard_impl(c, scene->replicas[d], delta, max_depth, flags, d, n, (uint8_t*)c->strip.p, sb, c->stream, nullptr, nullptr,
                                       cancel_flag, nullptr, nullptr, stats_out ? &st[d] : nullptr, &hook);
        }
        { std::lock_guard<std::mutex> lock(mu); --running; }
        cv.notify_one();
    };
    std::vector<std::thread> th;
    for (uint32_t d = 0; d < n; ++d) th.emplace_back(render_one, d);
    // the calling thread: hand over what has been announced, in arrival order, until every device's loop has ended
    uint32_t delivered = 0;
    size_t next = 0;
    for (;;) {
        Item it;
        {
            std::unique_lock<std::mutex> lock(mu);
            cv.wait(lock, [&] { return next < queue.size() || running == 0; });
            if (next >= queue.size()) break;   // running == 0 and nothing left
            it = queue[next++];
        }
        const uint8_t* stage = static_cast<const uint8_t*>(ctx->subs[it.dev]->pinned[stage_slot[it.dev]].p);
        for (uint32_t j = 0; j < it.count; ++j) {
            const uint32_t b = it.dev + (it.first + j) * n;   // the device's (first + j)-th owned block
            sink(user, b % t.nbx, b / t.nbx, t.bw, t.bh, stage + (size_t)(it.first + j) * block_bytes);
        }
        delivered += it.count;
        if (progress) progress(user, delivered, t.nblocks);
    }
    for (auto& x : th) x.join();
    for (uint32_t d = 0; d < n; ++d) ctx->subs[d]->pinned[stage_slot[d]].in_use = false;
    int result = TRR_OK;
    std::string msg;
    for (uint32_t d = 0; d < n; ++d) {
        if (res[d] == TRR_ERR_CANCELLED && result == TRR_OK) result = TRR_ERR_CANCELLED;
        else if (res[d] != TRR_OK && res[d] != TRR_ERR_CANCELLED) {
            result = res[d];
            msg += (msg.empty() ? "device " : "; device ") + std::to_string(ctx->subs[d]->device) + ": " + ctx->subs[d]->err;
        }
    }
    if (stats_out) {
        trr_stats sum{};
        for (const trr_stats& a : st) add_stats(sum, a);
        sum.blocks_rendered = delivered;
        sum.ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
        *stats_out = sum;
    }
    if (result == TRR_ERR_CANCELLED) return fail(ctx, result, "render cancelled");
    if (result != TRR_OK) return fail(ctx, result, msg);
    return TRR_OK;
}

extern "C" {

int trr_create_multi(const int* device_ids, int n_devices, trr_ctx** out) {
    if (!out) return fail(nullptr, TRR_ERR_INVALID, "trr_create_multi: out is NULL");
    *out = nullptr;
    if (!device_ids || n_devices <= 0 || n_devices > 64) return fail(nullptr, TRR_ERR_INVALID, "trr_create_multi: need 1..64 device ids");
    // a device listed twice: refused, unless TRR_MULTI_ALLOW_DUPLICATES is set -- several sub-contexts on one GPU rehearse the
    // N-device logic (sharding, host threads, strip exchange by peer copies, assembly) where only one GPU exists
    bool dup = false;
    for (int a = 0; a < n_devices; ++a)
        for (int b = a + 1; b < n_devices; ++b)
            if (device_ids[a] == device_ids[b]) dup = true;
    if (dup && !std::getenv("TRR_MULTI_ALLOW_DUPLICATES")) return fail(nullptr, TRR_ERR_INVALID, "trr_create_multi: a device is listed twice");
    DeviceGuard guard;
    trr_ctx* ctx = new trr_ctx();
    ctx->duplicate_devices = dup;
    ctx->device = device_ids[0];
    for (int k = 0; k < n_devices; ++k) {
        trr_ctx* c = nullptr;
        int rc = trr_create(device_ids[k], &c);
        if (rc != TRR_OK) {
            std::string msg = g_tls_error;
            for (trr_ctx* x : ctx->subs) trr_destroy(x);
            delete ctx;
            return fail(nullptr, rc, msg);
        }
        ctx->subs.push_back(c);
    }
    ctx->n_cu = ctx->subs[0]->n_cu;
    ctx->desc = std::to_string(n_devices) + " x " + ctx->subs[0]->desc;
    *out = ctx;
    return TRR_OK;
}

int trr_device_count(const trr_ctx* ctx) {
    if (!ctx) return 0;
    return ctx->subs.empty() ? 1 : (int)ctx->subs.size();
}

uint32_t trr_shard_blocks(uint32_t width, uint32_t height, uint32_t shard, uint32_t n_shards, uint32_t* block_ids, uint32_t capacity) {
    Tiling t;
    if (n_shards == 0 || shard >= n_shards || !tiling_of(width, height, &t)) return 0;
    const uint32_t n = owned_blocks(t, shard, n_shards);
    for (uint32_t j = 0; j < n && j < capacity && block_ids; ++j) block_ids[j] = shard + j * n_shards;   // the j-th owned block
    return n;
}

}   // extern "C"
