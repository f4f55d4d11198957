/* trayracer_hip.h -- C ABI of the MI355X (gfx950) render backend for Trayracer's src/raytracer hot path.
 *
 * The reference has no FFI/plugin interface.  The seam this library replaces is the pair of Rust calls
 *     Raytracer::new(scene, delta, max_depth)                 (src/raytracer/mod.rs:37)
 *     Raytracer::render(&self, (x,y), (w,h), anti_aliasing)   (src/raytracer/mod.rs:218)
 * made only from RenderingThread::run (src/raytracer/render.rs:105,132), i.e. the body of the block loop
 * (render.rs:103-152).  Each entry point below cites what it stands in for.  INTEGRATION.md shows the Rust
 * shim + build.rs lines a maintainer would add.
 *
 * Rules of the boundary: plain pointers and sizes only; every pointer passed IN is borrowed for the
 * duration of the call; opaque handles are owned by the library; no entry point aborts -- all return
 * TRR_OK (0) or a negative trr_status and leave a message for trr_last_error().  Multi-GPU rendering shards the
 * reference's own 20x20 block grid (render.rs:103-110) over the GPUs (block id % N), scene replicated, one gather of
 * the RGB8 strips.  Two forms: inside ONE process -- the reference's host has one process and one render thread
 * (render.rs:39-71) -- a context made by trr_create_multi drives all its devices behind trr_render /
 * trr_render_blocks; or one process (rank) per GPU, each calling trr_render_shard_device, the strips gathered by the
 * caller (torch.distributed / RCCL) and assembled by trr_assemble_device.
 * Destroy scenes before their context.  Every entry point leaves the calling thread's current HIP device as it was.
 */
#ifndef TRAYRACER_HIP_H
#define TRAYRACER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct trr_ctx trr_ctx;       /* device, stream, work buffers */
typedef struct trr_scene trr_scene;   /* device-resident flattened scene (BVH nodes, triangle records, materials ...) */

typedef enum trr_status {
    TRR_OK = 0,
    TRR_ERR_INVALID = -1,      /* bad argument */
    TRR_ERR_RESOLUTION = -2,   /* width or height < 20: the reference divides by zero at render.rs:109 */
    TRR_ERR_DEVICE = -3,       /* HIP error (message has the HIP error string) */
    TRR_ERR_CANCELLED = -4,    /* cancel flag observed; blocks rendered so far are valid, the rest are black */
    TRR_ERR_NOMEM = -5
} trr_status;

/* src/scene/material.rs:5-15.  has_* flags are the reference's Option<..>::is_some(). */
#define TRR_MAT_HAS_KD 1u
#define TRR_MAT_HAS_KS 2u
#define TRR_MAT_HAS_NS 4u
#define TRR_MAT_HAS_DISSOLVE 8u
typedef struct trr_material_desc {
    float kd[3];          /* diffuse_color */
    float ks[3];          /* specular_color */
    float ns;             /* specular_exponent */
    float dissolve;       /* 1 - d  (object.rs:106) */
    int32_t illum;        /* IlluminationModel 0..10 (material.rs:31-53) */
    int32_t texture;      /* index into the textures array, -1 = no map_Kd */
    uint32_t flags;       /* TRR_MAT_HAS_* */
} trr_material_desc;

/* 8-bit RGB image, row-major, as image::RgbImage after into_rgb8() (object.rs:31-38, skybox.rs:57). */
typedef struct trr_image_rgb8 {
    const uint8_t* rgb;
    uint32_t width, height;
} trr_image_rgb8;

/* One Object (src/scene/object.rs:19-29).  `triangles` is n_triangles x 24 floats laid out like Triangle
 * (src/scene/triangle.rs:9-21): a, b, c, a_normal, b_normal, c_normal (3 each), a_uv, b_uv, c_uv (2 each).
 * `material_index` (nullable = all None) holds the object-local index into `materials`, -1 = None.
 * Triangle ID = (object index, row in `triangles`).
 * `transform` / `inverse` are Object::transform().to_homogeneous() (object.rs:177-182) and its inverse,
 * column-major 4x4 exactly as nalgebra stores them (Matrix4::as_slice()), computed by the HOST's own math
 * so the device inherits the host's bits; trr_object_matrices() is a restatement for non-Rust hosts. */
typedef struct trr_object_desc {
    const float* triangles;
    const int32_t* material_index;
    uint32_t n_triangles;
    const trr_material_desc* materials;
    uint32_t n_materials;
    float transform[16];
    float inverse[16];
} trr_object_desc;

/* src/scene/light.rs:5-9 (color is the already-normalised Ke, light.rs:33). */
typedef struct trr_light {
    float position[3];
    float color[3];
    float intensity;
} trr_light;

/* src/scene/settings.rs:4-12 + skybox.rs:6-9.  `seed` feeds the counter RNG that replaces the
 * irreproducible thread-local rand::random() of raytracer/mod.rs:234,238. */
typedef struct trr_settings {
    float ambient_color[3];      /* normalised (settings.rs:51-54) */
    float ambient_intensity;
    int32_t skybox_is_image;     /* 0: skybox_color, 1: skybox_image */
    float skybox_color[3];
    trr_image_rgb8 skybox_image;
    int32_t anti_aliasing;
    uint32_t samples;
    uint64_t seed;
} trr_settings;

/* src/scene/camera.rs:5-11 (fov in radians). */
typedef struct trr_camera {
    float position[3];
    float look_at[3];
    float up[3];
    float fov;
    uint32_t width, height;
} trr_camera;

/* Per-render statistics (the reference only keeps wall ms, render.rs:189-190). */
typedef struct trr_stats {
    uint64_t rays_primary;         /* closest-hit queries for camera rays */
    uint64_t rays_shadow;          /* one per (shaded hit x light) */
    uint64_t rays_reflection;      /* one per reflective shaded hit below max_depth */
    uint64_t rays_march;           /* extra queries from the 0.05-step transparency march (mod.rs:100-111) */
    uint64_t node_visits;          /* filled only when counters were requested */
    uint64_t tri_tests;
    uint64_t box_tests;
    double ms_total;               /* host wall clock around the block loop (what render.rs:96,189 times) */
    double ms_trace;               /* HIP-event time of all trace-kernel launches */
    double ms_shade;               /* HIP-event time of the hit-setup + resolve launches */
    uint32_t trace_launches;
    uint32_t blocks_rendered;
    /* SIMD-efficiency diagnostics of the trace kernel, filled only with TRR_FLAG_COUNTERS: steps counted once per
     * WAVEFRONT (a step runs for all 64 lanes whether 1 or 64 of them need it) */
    uint64_t wave_node_steps;      /* inner-node iterations; node_visits / (64 * this) = lane utilisation */
    uint64_t wave_tri_steps;       /* triangle-slot iterations of the leaf loop; box_tests-4*node_visits over 64*this */
    uint64_t wave_outer_steps;     /* outer (refill / transition / descend / leaf) iterations */
    uint64_t wave_lane_steps;      /* sum over outer iterations of lanes that still carry a ray */
    uint64_t wave_trans_steps;     /* executions of the transition block (ray start / object change / ray end) per wavefront */
    uint64_t trans_lanes;          /* lanes that took part in them */
    uint64_t wave_uniform_node_steps; /* inner-node iterations in which every participating lane stood on the same node */
    uint64_t queue_overflows;      /* pushes beyond a wavefront's ray-queue capacity: always 0 (the capacity is a proven bound) */
    uint64_t rays_shadow_culled;   /* of rays_shadow: answered without a traversal, because the light's diffuse and specular
                                    * intensities at the hit are exactly 0.0 and the material is no mirror -- whatever the ray
                                    * found, mod.rs:171-196 would add exactly nothing */
    uint64_t rays_primary_culled;  /* of rays_primary: camera rays answered "no hit" (mod.rs:119) without a ray of their own, because
                                    * no leaf box of the scene projects onto their 8x8-pixel tile */
} trr_stats;

typedef void (*trr_progress_cb)(void* user, uint32_t blocks_done, uint32_t blocks_total);

/* ---- context -------------------------------------------------------------------------------- */
/* Replaces nothing in the reference (it has no device).  Selects GPU `device_id` for this process. */
int trr_create(int device_id, trr_ctx** out);
/* The same for N GPUs of this process (1 <= n_devices <= 64, distinct ids; the first is the root that assembles the
 * frame).  Scenes created on such a context are uploaded to every device, the setters reach every copy, and trr_render /
 * trr_render_blocks render block b on device b % N and bring the strips to the root with ONE ncclGather (RCCL over
 * xGMI; librccl is loaded on first use, a single-device context never needs it).  The Rust shim of INTEGRATION.md uses
 * all GPUs of a node by changing its trr_create call into this one.  trr_render_shard_device* and the hit-id buffers of
 * trr_render need a single-device context.  Environment: TRR_MULTI_TRANSPORT=copy exchanges the strips by peer copies
 * instead of RCCL; TRR_MULTI_ALLOW_DUPLICATES=1 lets a device be listed twice (a rehearsal of the N-device path on one
 * GPU; always peer copies).  An error reported by RCCL (ncclCommInitAll, ncclGather) costs a warning on stderr, not the
 * frame: the context exchanges its strips by peer copies from then on.  No counterpart in the reference (it has no device). */
int trr_create_multi(const int* device_ids, int n_devices, trr_ctx** out);
int trr_device_count(const trr_ctx* ctx);   /* devices the context drives (1 for trr_create) */
void trr_destroy(trr_ctx* ctx);
/* Last error text of `ctx` (or of the calling thread when ctx is NULL).  Never NULL. */
const char* trr_last_error(const trr_ctx* ctx);

/* ---- scene upload --------------------------------------------------------------------------- */
/* Replaces Scene::clone() + Bvh::build (render.rs:60, object.rs:156): flattens all objects into one
 * BVH4 node array (128-B nodes), leaf-ordered 64-B triangle records, a 64-B interpolation record per
 * triangle, a material table and a texture atlas in HBM. */
int trr_scene_create(trr_ctx* ctx, const trr_object_desc* objects, uint32_t n_objects,
                     const trr_image_rgb8* textures, uint32_t n_textures, trr_scene** out);
void trr_scene_destroy(trr_scene* scene);

/* The same with the BVH built where `build` says.  The reference rebuilds an object's Bvh on the CPU whenever the UI
 * adds an object (object.rs:156 via ui/properties.rs:307-415); TRR_BUILD_DEVICE does the whole flattening on the GPU
 * (Morton keys, radix sort, radix tree with its subtrees of up to 2048 triangles rebuilt by SAH splits, box fit, 4-wide
 * collapse: tens of milliseconds for 10 M triangles instead of seconds) for a tree whose frames are within a few per
 * cent of the host SAH tree's.  Both builders produce the same
 * triangle / interpolation records, and -- because the closest hit is a min over every triangle whose own box the ray
 * passes (object.rs:191-206), whatever the tree -- the same image. */
#define TRR_BUILD_HOST 0u     /* binned-SAH build on the host cores (what trr_scene_create does) */
#define TRR_BUILD_DEVICE 1u   /* Morton-order build on the GPU, SAH-rebuilt treelets */
int trr_scene_create_ex(trr_ctx* ctx, const trr_object_desc* objects, uint32_t n_objects,
                        const trr_image_rgb8* textures, uint32_t n_textures, uint32_t build, trr_scene** out);

/* What the upload produced (no counterpart in the reference, whose Bvh is opaque). */
typedef struct trr_scene_info {
    uint64_t n_triangles;        /* rows of all objects' triangle arrays */
    uint64_t n_triangles_live;   /* of those, in the tree (|ab x ac| > 0, triangle.rs:56) */
    uint64_t n_nodes;            /* 128-B 4-wide nodes */
    uint32_t bvh_depth;          /* levels of the deepest object tree (root = 1) */
    uint32_t stack_need;         /* traversal stack entries the tree can need; trr_scene_create* refuses more than 64 */
    uint32_t build;              /* TRR_BUILD_* */
    uint32_t pad;
    double ms_build;             /* wall time of the flattening + BVH build + upload */
} trr_scene_info;
int trr_scene_get_info(const trr_scene* scene, trr_scene_info* out);
/* Diagnostic: downloads the node and triangle arrays and checks the tree on the host -- every live triangle in exactly
 * one leaf, every child box containing the reference's own box (triangle.rs:88-95) of every triangle below it, all
 * references in range.  TRR_OK or TRR_ERR_INVALID with the first violation in trr_last_error(). */
int trr_scene_validate(trr_scene* scene);

/* Cheap per-render state the UI edits between renders (ui/properties.rs). */
int trr_scene_set_instances(trr_scene* scene, const float* transforms, const float* inverses, uint32_t n_objects);
int trr_scene_set_lights(trr_scene* scene, const trr_light* lights, uint32_t n_lights);
int trr_scene_set_settings(trr_scene* scene, const trr_settings* settings);
int trr_scene_set_camera(trr_scene* scene, const trr_camera* camera);

/* ---- render --------------------------------------------------------------------------------- */
#define TRR_FLAG_COUNTERS 1u   /* also count node visits / box tests / triangle tests (slower kernel build) */
#define TRR_FLAG_DEFER_STATS 2u /* trr_render_shard_device with stats_out: do not wait for the frame; *stats_out is
                                 * filled by the next trr_collect_stats (stats_out must stay valid until then).  Lets
                                 * a caller keep several frames in flight and still get per-launch timings. */
#define TRR_FLAG_PIPELINED 4u   /* trr_render_shard_device: the caller keeps two frames in flight, consecutive frames on
                                 * different streams and strips.  A shard that fits one batch is then rendered as one
                                 * batch, frames alternating between two work sets, instead of two batches per frame. */

/* Replaces the block loop of RenderingThread::run (render.rs:103-152) for the whole frame: writes
 * row-major RGB8 (width*height*3, remainder strip left 0,0,0 as render.rs:100-102) into host memory.
 * hit_obj_out / hit_tri_out (nullable, width*height u32 each, 0xFFFFFFFF = miss) receive the primary ray's
 * first closest hit for the parity tests (1 spp only).  `cancel_flag` (nullable) is polled between block
 * batches like render.rs:113,154; `progress` (nullable) is called after each batch (render.rs:143-149). */
int trr_render(trr_ctx* ctx, trr_scene* scene, float delta, uint32_t max_depth, uint32_t flags,
               uint8_t* rgb_out, uint32_t* hit_obj_out, uint32_t* hit_tri_out,
               const volatile int32_t* cancel_flag, trr_progress_cb progress, void* user, trr_stats* stats_out);

/* Progressive form of trr_render: the reference's sink (render.rs:155-186) receives each finished block as
 * (pixels, x_block, y_block) and paints it into the texture and the image while the render goes on.  The frame is
 * rendered in batches of `blocks_per_batch` blocks (0 = one row of the 20x20 grid) in row-major block order, and after
 * each batch `sink` is called once per block of it, on the calling thread: (x_block, y_block) are block coordinates as
 * in render.rs:129-130, `rgb` holds block_w*block_h RGB8 pixels row-major inside the block and is valid only during the
 * call.  Pixels of the remainder strip (render.rs:100-102) belong to no block and are never delivered.  `cancel_flag`
 * is polled and `progress` (nullable) called between batches as in trr_render.  Smaller batches mean earlier blocks
 * and a slower frame (every batch ends with the latency of its longest ray).
 * On a context of several devices (trr_create_multi) every device renders its blocks (b % N) in batches of
 * blocks_per_batch / N and the calling thread hands the finished batches to `sink` in ARRIVAL order, while the devices go on
 * rendering: each block exactly once, still only on the calling thread, but not in row-major order -- the reference's
 * `par_bridge` (render.rs:109-113) delivers in completion order too.  `progress` then also runs on the calling thread. */
typedef void (*trr_block_cb)(void* user, uint32_t x_block, uint32_t y_block, uint32_t block_w, uint32_t block_h, const uint8_t* rgb);
int trr_render_blocks(trr_ctx* ctx, trr_scene* scene, float delta, uint32_t max_depth, uint32_t flags, uint32_t blocks_per_batch,
                      const volatile int32_t* cancel_flag, trr_block_cb sink, trr_progress_cb progress, void* user,
                      trr_stats* stats_out);

/* Multi-GPU form: renders the blocks b with b % n_shards == shard (row-major block ids over the grid of
 * render.rs:109-110) into a compact DEVICE strip: block j-th owned -> bytes [j*bw*bh*3, (j+1)*bw*bh*3),
 * row-major inside the block.  All strips have trr_strip_bytes() bytes (the last block slot of a short
 * shard is zero).  Asynchronous on `stream` (a hipStream_t, NULL = the context's stream) except for the
 * cancel/progress polling between batches.  One rank per GPU calls this, the strips are gathered to the
 * root over RCCL (a single ncclGather / torch.distributed.gather), and the root calls trr_assemble_device. */
int trr_render_shard_device(trr_ctx* ctx, trr_scene* scene, float delta, uint32_t max_depth, uint32_t flags,
                            uint32_t shard, uint32_t n_shards, void* strip_dev, size_t strip_bytes, void* stream,
                            const volatile int32_t* cancel_flag, trr_progress_cb progress, void* user,
                            trr_stats* stats_out);
size_t trr_strip_bytes(uint32_t width, uint32_t height, uint32_t n_shards);
/* The partition itself (host arithmetic, no GPU): writes the row-major block ids (render.rs:109-110, 129-130) that `shard`
 * of `n_shards` renders, in strip order, into block_ids (up to `capacity`; NULL to only count) and returns their number. */
uint32_t trr_shard_blocks(uint32_t width, uint32_t height, uint32_t shard, uint32_t n_shards, uint32_t* block_ids, uint32_t capacity);

/* The entry point that allocates nothing.  It has NO counterpart in the reference; it exists for callers that cannot call
 * hipMalloc: code under HIP graph capture, tight serving loops, harnesses that forbid allocation.  Same arguments, strip
 * layout and bytes as trr_render_shard_device without its statistics / cancel / progress parameters, but every work
 * buffer (ray queues, hit and shaded-hit records, occluder codes, traversal-stack spill, queue counters) lives in ONE
 * file-scope `static __device__` array of trr_noalloc_scratch_bytes() = 256 MiB that each GPU the library runs on holds
 * until the process ends; the call makes no hipMalloc / hipFree / hipHostMalloc / event creation, only memsets and
 * kernel launches on `stream` (NULL = the context's stream), one batch of whole blocks after the other.
 * Limit: the work set of ONE block (block_w*block_h*samples samples; the bytes per sample grow with lights x max_depth)
 * must fit the array, else TRR_ERR_INVALID with the figures in trr_last_error() -- e.g. a 1080p block at 16 spp, 4 lights,
 * max_depth 5 fits (about 4 blocks per batch); a 4K block at 16 spp does not.  Every call on a GPU shares that GPU's
 * array and keeps the last call's contents.  Calls from different streams or contexts of one GPU are ordered ON THE
 * DEVICE by one event per GPU (created with the GPU's first context; the call waits for it and records it again), so they
 * queue up behind each other instead of racing -- they never overlap, which is the price of the one array.  Two host
 * THREADS calling at the same moment are serialised by a mutex for the few microseconds of queueing. */
int trr_render_shard_device_noalloc(trr_ctx* ctx, trr_scene* scene, float delta, uint32_t max_depth, uint32_t shard,
                                    uint32_t n_shards, void* strip_dev, size_t strip_bytes, void* stream);
size_t trr_noalloc_scratch_bytes(void);
/* De-interleaves n_shards gathered strips (contiguous, trr_strip_bytes each) into the row-major RGB8 frame
 * (width*height*3 device bytes; unrendered remainder pixels are zeroed). */
int trr_assemble_device(trr_ctx* ctx, uint32_t width, uint32_t height, uint32_t n_shards,
                        const void* strips_dev, void* frame_dev, void* stream);
/* Blocks until everything queued on the context's stream (or `stream`) has finished. */
int trr_synchronize(trr_ctx* ctx, void* stream);
/* Waits for every frame rendered with TRR_FLAG_DEFER_STATS and fills their trr_stats (ms_total = device time from
 * the frame's first kernel start to its last kernel end).  The reference's only statistic is Render.time
 * (render.rs:188-190); per-launch timing exists for the roofline figures of bench.py. */
int trr_collect_stats(trr_ctx* ctx);

/* ---- host math helpers (restatements of nalgebra for non-Rust hosts) -------------------------- */
/* Object::transform() (object.rs:177-182) and its 4x4 inverse, column-major; rotation = unit quaternion
 * (i, j, k, w).  Returns 1 when invertible. */
int trr_object_matrices(const float translation[3], const float rotation[4], const float scale[3],
                        float transform[16], float inverse[16]);
/* Camera::ray (camera.rs:30-42) for normalised device coordinates (x, y). */
void trr_camera_ray(const trr_camera* camera, float x, float y, float origin[3], float direction[3]);

/* Library / device description for logs: "trayracer_hip <version> gfx950 ...". */
const char* trr_version(void);
/* The number of HIP hardware queues the library plans its streams for: the value of GPU_MAX_HW_QUEUES that the host had set
 * when the library was loaded, else the 16 the library's initialiser sets itself (a frame uses up to nine streams; with HIP's
 * default of 4 the library does not use its split schedule -- DESIGN.md section 4).  Measurements quote it. */
int trr_hw_queues(void);

#ifdef __cplusplus
}
#endif
#endif /* TRAYRACER_HIP_H */
