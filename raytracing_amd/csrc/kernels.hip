// kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the Trayracer hot path.
//
// Wavefront (queue-based) organisation, TWO kernels per batch of image blocks, no host round trip inside a batch:
//
//   k_trace              ONE persistent launch traces every ray of the batch: camera rays generated in-kernel
//                        (mod.rs:250-255, camera.rs:30-42), closest hits with the transparency march (mod.rs:45-51, 96-114;
//                        object.rs:184-220), and -- at the end of each chain ray -- the hit set-up (mod.rs:124-153, 199-206)
//                        whose L shadow rays and <= 1 reflection ray go into the wavefront's OWN ray queue, which the same
//                        wavefront drains before it claims more camera rays.  No level barriers, no ray records in HBM.
//   k_resolve_samples    deepest-first evaluation of shade_impl per sample, incl. sky / texture look-ups  (mod.rs:57-93, 141-211)
//   k_resolve_pixels     mean over a pixel's samples, RGB8                                     (mod.rs:248-249, render.rs:134-140)
//
// Arithmetic is IEEE fp32 with the reference's operation order; build with -ffp-contract=off (Rust never
// fuses a*b+c) and correctly rounded division / sqrt (hipcc default).  Comments cite the reference lines
// (paths relative to /root/reference/src).  "[crate]" marks semantics of un-vendored crates (nalgebra, bvh)
// restated from their public sources.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "device_types.h"
#include "launch.h"

namespace trr {

#define DEV __device__ __forceinline__

// ------------------------------------------------------------------------------------------------
// fp32 vector helpers with nalgebra's operation order [crate]
// ------------------------------------------------------------------------------------------------
struct F3 {
    float x, y, z;
};
DEV F3 f3(float x, float y, float z) { return F3{x, y, z}; }
DEV F3 operator+(F3 a, F3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
DEV F3 operator-(F3 a, F3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
DEV F3 neg(F3 a) { return f3(-a.x, -a.y, -a.z); }
DEV F3 scale(F3 a, float s) { return f3(a.x * s, a.y * s, a.z * s); }
DEV F3 lscale(float s, F3 a) { return f3(s * a.x, s * a.y, s * a.z); }
DEV F3 unscale(F3 a, float s) { return f3(a.x / s, a.y / s, a.z / s); }
DEV F3 cmul(F3 a, F3 b) { return f3(a.x * b.x, a.y * b.y, a.z * b.z); }
DEV float dot(F3 a, F3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
DEV F3 cross(F3 a, F3 b) { return f3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
DEV float norm_squared(F3 a) { return dot(a, a); }
DEV float norm(F3 a) { return sqrtf(norm_squared(a)); }
DEV F3 normalize(F3 a) { return unscale(a, norm(a)); }
DEV float rmax(float a, float b) { return __builtin_fmaxf(a, b); }   // Rust f32::max: NaN loses
DEV float rmin(float a, float b) { return __builtin_fminf(a, b); }
DEV bool is_nan(float a) { return a != a; }

// rows of a 3x4 affine matrix times vector / point: ((m0*x + m1*y) + m2*z) [+ m3]   [crate: gemv order]
DEV F3 xform_vec(const float* m, F3 v) {
    return f3(m[2] * v.z + (m[1] * v.y + m[0] * v.x), m[6] * v.z + (m[5] * v.y + m[4] * v.x),
              m[10] * v.z + (m[9] * v.y + m[8] * v.x));
}
DEV F3 xform_point(const float* m, F3 p) {
    F3 r = xform_vec(m, p);
    return f3(r.x + m[3], r.y + m[7], r.z + m[11]);
}

// Rust `f32 as u32`: saturating, NaN -> 0
DEV uint32_t f32_as_u32(float f) {
    if (!(f > 0.0f)) return 0u;
    if (f >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)f;
}

// raytracer/mod.rs:53-55
DEV F3 reflect(F3 incoming, F3 normal) { return incoming - lscale(2.0f * dot(incoming, normal), normal); }

// raytracer/mod.rs:57-82
DEV F3 skybox(const FrameParams& fp, F3 direction) {
    if (fp.sky == nullptr) return f3(fp.sky_color[0], fp.sky_color[1], fp.sky_color[2]);
    F3 d;
    float n = norm(direction);
    if (n <= 1.1920929e-07f) d = f3(0.0f, 1.0f, 0.0f);   // try_normalize(f32::EPSILON).unwrap_or(Vector3::y)
    else d = unscale(direction, n);
    const float PI = 3.14159265358979323846f;
    float fx = (0.5f + atan2f(d.z, d.x) / (2.0f * PI)) * (float)fp.sky_w;
    float fy = (0.5f - asinf(d.y) / PI) * (float)fp.sky_h;
    uint32_t x = f32_as_u32(fx) % fp.sky_w;
    uint32_t y = f32_as_u32(fy) % fp.sky_h;
    const uint8_t* p = fp.sky + ((size_t)y * fp.sky_w + x) * 3;
    return f3((float)p[0] / 255.0f, (float)p[1] / 255.0f, (float)p[2] / 255.0f);
}

// raytracer/mod.rs:84-93
DEV F3 texture_lookup(const FrameParams& fp, int tex, float u, float v) {
    TexRec t = fp.texs[tex];
    uint32_t x = f32_as_u32(u * (float)t.w) % t.w;
    uint32_t y = f32_as_u32((1.0f - v) * (float)t.h) % t.h;
    const uint8_t* p = fp.texels + t.offset + ((size_t)y * t.w + x) * 3;
    return f3((float)p[0] / 255.0f, (float)p[1] / 255.0f, (float)p[2] / 255.0f);
}

// counter RNG standing in for rand::random::<f32>() (raytracer/mod.rs:234,238): (u32 >> 8) * 2^-24 like
// rand 0.9's StandardUniform, the u32 from a splitmix64-style hash of (seed, pixel, sample, dimension).
DEV float rand_f32(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t dim) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * ((((uint64_t)pixel << 32) | ((uint64_t)sample << 1) | dim) + 1ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (float)((uint32_t)(z >> 32) >> 8) * (1.0f / 16777216.0f);
}

// ------------------------------------------------------------------------------------------------
// work index -> pixel.  Pixels of a block are walked in 8-row strips, column-major inside a strip, so 64
// consecutive work items form an 8x8 screen tile (coherent rays per wavefront).  With spp > 1 the samples
// of one pixel are consecutive.
// ------------------------------------------------------------------------------------------------
struct PixelPos {
    uint32_t x, y;        // frame coordinates
    uint32_t strip_off;   // pixel offset inside this shard's strip buffer
};
DEV PixelPos pixel_of(const FrameParams& fp, const BatchParams& bp, uint32_t pix) {
    uint32_t bpix = fp.bw * fp.bh;
    uint32_t jb = pix / bpix, r = pix - jb * bpix;
    uint32_t strip = r / (8u * fp.bw);
    uint32_t q = r - strip * 8u * fp.bw;
    uint32_t sh = min(8u, fp.bh - strip * 8u);
    uint32_t lx = q / sh, ly = strip * 8u + (q - lx * sh);
    uint32_t j = bp.first_owned + jb;
    uint32_t b = fp.shard + j * fp.n_shards;
    uint32_t xb = b % fp.nbx, yb = b / fp.nbx;
    PixelPos p;
    p.x = xb * fp.bw + lx;
    p.y = yb * fp.bh + ly;
    p.strip_off = j * bpix + ly * fp.bw + lx;
    return p;
}

// scene/camera.rs:30-42 with the look_at basis hoisted to the host (same inputs -> same bits)
DEV F3 camera_dir(const FrameParams& fp, float x, float y) {
    F3 dc = f3(x, -y, fp.cam_focal);   // camera.rs:32
    F3 d = f3(dot(f3(fp.cam_x[0], fp.cam_y[0], fp.cam_z[0]), dc), dot(f3(fp.cam_x[1], fp.cam_y[1], fp.cam_z[1]), dc),
              dot(f3(fp.cam_x[2], fp.cam_y[2], fp.cam_z[2]), dc));   // camera.rs:36 (tr_mul)
    return normalize(d);   // camera.rs:40
}

// raytracer/mod.rs:224-256: camera ray of sample `si` of pixel (x, y)
DEV F3 primary_dir(const FrameParams& fp, uint32_t x, uint32_t y, uint32_t si) {
    float w = (float)fp.width, h = (float)fp.height;
    float fx, fy;
    if (fp.aa) {
        uint32_t s = fp.sqrt_spp;
        uint32_t xi = si % s, yi = si / s;
        uint32_t pixel = y * fp.width + x;
        float rx = rand_f32(fp.seed, pixel, si, 0), ry = rand_f32(fp.seed, pixel, si, 1);
        float jx = ((float)x + ((float)xi + (rx * 2.0f + -1.0f)) / (float)s) / w;   // mod.rs:233-236
        float jy = ((float)y + ((float)yi + (ry * 2.0f + -1.0f)) / (float)s) / h;   // mod.rs:237-240
        fx = (jx * 2.0f + -1.0f) * (w / h);                                          // mod.rs:241
        fy = jy * 2.0f + -1.0f;                                                      // mod.rs:242
    } else {
        fx = ((float)x / w * 2.0f + -1.0f) * (w / h);   // mod.rs:251
        fy = (float)y / h * 2.0f + -1.0f;               // mod.rs:252
    }
    return camera_dir(fp, fx, fy);
}

// ------------------------------------------------------------------------------------------------
// k_trace: closest-hit queries with the transparency march (mod.rs:45-51, 96-114; object.rs:184-220;
// triangle.rs:53-85).  Persistent wavefronts pull rays from a device-side work counter; lanes whose ray
// has finished are re-filled (ballot + one atomic per wavefront) once enough of them are idle.
// ------------------------------------------------------------------------------------------------
// Threads per block of the trace kernels.  The host side counts blocks in units of 256 threads (grid sizes, resident blocks
// per CU, spill / queue sizes): the launchers below convert.
#ifndef TRR_TRACE_THREADS
#define TRR_TRACE_THREADS 256
#endif
constexpr int TRACE_THREADS = TRR_TRACE_THREADS;
static_assert(TRACE_THREADS == 64 || TRACE_THREADS == 128 || TRACE_THREADS == 256, "trace blocks are 1, 2 or 4 wavefronts");
constexpr int BLOCKS_PER_256 = 256 / TRACE_THREADS;
#ifndef TRR_RESOLVE_THREADS
#define TRR_RESOLVE_THREADS 256
#endif
constexpr int RESOLVE_THREADS = TRR_RESOLVE_THREADS;
#ifdef TRR_WAVES_PER_EU
#define TRR_OCCUPANCY_ATTR __attribute__((amdgpu_waves_per_eu(TRR_WAVES_PER_EU, 8)))
#else
#define TRR_OCCUPANCY_ATTR
#endif
#ifndef TRR_STACK_LDS
#define TRR_STACK_LDS 12
#endif
constexpr int STACK_LDS = TRR_STACK_LDS;       // per-lane (ref, tmin) entries kept in LDS
constexpr int STACK_SPILL = TRACE_STACK_ENTRIES - STACK_LDS;     // further entries in a per-thread HBM spill area (rarely touched)
#ifndef TRR_REFILL_MIN
#define TRR_REFILL_MIN 42   // swept 24..48 with TRR_DEFER_END: 31.4 / 30.9 / 30.6 (36) / 30.25 (42) / 30.5 ms
#endif
constexpr int REFILL_MIN = TRR_REFILL_MIN;      // re-fill when at least this many of the 64 lanes are idle
#ifndef TRR_END_MIN
#define TRR_END_MIN 16
#endif
constexpr int END_MIN = TRR_END_MIN;            // staged ray starts (k_trace_level<.., ST>): ray ends run once this many lanes wait
#ifndef TRR_PRUNE_REL
#define TRR_PRUNE_REL 1.0e-3f
#endif
constexpr float PRUNE_REL = TRR_PRUNE_REL;
#ifndef TRR_NODE_BIAS
#define TRR_NODE_BIAS 8
#endif
constexpr int NODE_BIAS = TRR_NODE_BIAS;
#ifndef TRR_TRIS_PER_STEP
#define TRR_TRIS_PER_STEP 2   // 2: -4 % frame time against 1 (fewer, fatter leaf iterations); whole leaves cost a wave of occupancy
#endif
#ifndef TRR_DEFER_END
#define TRR_DEFER_END 1   // end-of-ray transitions wait for company: 22.0 M -> 9.4 M executions of the block per 4K frame, -2.2 %
#endif
#ifndef TRR_CHUNK_DIV
#define TRR_CHUNK_DIV 8
#endif
#ifndef TRR_CHUNK_MAX
#define TRR_CHUNK_MAX 512
#endif
constexpr uint32_t CHUNK_MAX = TRR_CHUNK_MAX;
constexpr uint32_t CHUNK_DIV = TRR_CHUNK_DIV;   // a claim takes 1/CHUNK_DIV of the remaining work per wavefront   // a node step runs when 8ths-weighted inner lanes >= leaf lanes


// triangle.rs:53-85 + object.rs:201,206 for one 64-B TriRec (normal and |ab x ac| precomputed with the same
// fp32 operations).  One function for the leaf loop AND the end-of-object re-evaluation of the winner, so both
// produce the same bits.
struct TriHit {
    float u, v, w;   // weights of (a, b, c)
    F3 point;        // a*u + b*v + c*w (object space)
    float key;       // |o' - point|^2 (object.rs:206)
};
DEV bool tri_test(float4 q0, float4 q1, float4 q2, float len, F3 oo, F3 od, float delta, TriHit& h) {
    F3 a = f3(q0.x, q0.y, q0.z), b = f3(q0.w, q1.x, q1.y), c = f3(q1.z, q1.w, q2.x), n = f3(q2.y, q2.z, q2.w);
    if (len <= delta) return false;                              // :56  try_normalize(delta)
    float t = dot(a - oo, n) / dot(od, n);                       // :58
    if (t < delta) return false;                                 // :60 (NaN falls through)
    F3 p = oo + scale(od, t);                                    // :64
    F3 ab = b - a;
    F3 ap = p - a, bp_ = p - b, cp = p - c;                      // :66-68
    float e0 = dot(cross(ab, ap), n);                            // :70,74
    float e1 = dot(cross(c - b, bp_), n);                        // :71,75
    float e2 = dot(cross(a - c, cp), n);                         // :72,76
    if (e0 < 0.0f || e1 < 0.0f || e2 < 0.0f) return false;       // :78
    float area = e0 + e1 + e2;                                   // :82
    h.u = e1 / area; h.v = e2 / area; h.w = e0 / area;           // :84
    h.point = (scale(a, h.u) + scale(b, h.v)) + scale(c, h.w);   // object.rs:201
    h.key = norm_squared(oo - h.point);                          // object.rs:206
    return true;
}

// min_by_key(OrderedFloat) over candidates: NaN is the greatest key; equal keys keep the first candidate, which
// here (as in the oracle) is the lowest triangle index.
DEV bool candidate_better(bool have, float key, uint32_t tri, float best_key, uint32_t best_tri) {
    if (!have) return true;
    if (is_nan(key)) return is_nan(best_key) && tri < best_tri;
    if (is_nan(best_key)) return true;
    return key < best_key || (key == best_key && tri < best_tri);
}

// One entry of a ray queue (32 B = two 16-byte words): origin, direction, sample index inside the batch, meta (bits 0..7
// level, bit 8 shadow flag, bits 16..31 light index).  A shadow entry of level k is the ray (P + L^*delta, L^) of
// mod.rs:149-153 from the shaded hit of level k; a chain entry of level k >= 1 the reflection ray (P + N*delta,
// reflect(D, N)) of mod.rs:203-206 from the hit of level k - 1.  The whole ray is in the entry so that a ray STARTS with one
// memory round trip: the kernel is bound by the latency of dependent loads, not by bytes.
// Compact shadow entry (16 B, level-synchronous schedule, scenes of at most COMPACT_MAX_LIGHTS lights): (Px Py Pz, sample |
// light << 25) -- the shaded point itself.  The trace kernel rebuilds the ray of mod.rs:149-153 from it with the very
// operations the hit set-up would have used (L^ = normalize(light - P), origin = P + L^ * delta), so the same bits; the
// lights' positions sit in LDS for that.  Halves what the set-up pass writes per hit and what a shadow ray's start reads.
constexpr uint32_t META_SHADOW = 0x100u;
constexpr uint32_t COMPACT_MAX_LIGHTS = 128u, COMPACT_SAMPLE_BITS = 25u;
constexpr uint32_t TRI_FLAG_TRANSPARENT = 1u, TRI_FLAG_REFLECTIVE = 2u;   // TriRec.flags
// occluder codes of a (level, light, sample): a material id, -1 nothing hit, -2 a hit without material, and
constexpr int32_t OCC_CULLED = -3;   // no shadow ray traced: the light adds exactly nothing at this hit (k_setup)
constexpr uint8_t NLEV_PENDING = 0xFF;   // nlev of a camera sample whose hit waits for k_setup (max_depth <= 254 keeps it free)
constexpr uint32_t MARCH_CAP = 0xFFFFu;   // raycasts of one transparency march (mod.rs:100-111); the reference has no cap and
                                          // would spin for ever on a ray that cannot advance (zero / NaN direction) -- the
                                          // oracle shares this cap, which 3 276 scene units of glass would be needed to reach

// Work items of a trace launch are claimed per wavefront in chunks (guided self-scheduling: a fraction of what is left
// per wavefront, between chunk_min and CHUNK_MAX items -- one same-address atomic per chunk).  The list can be cut into
// WORK_SEGMENTS contiguous segments, one per XCD (workgroups are dealt to the 8 XCDs round-robin, so blockIdx & 7 names
// the XCD): a wavefront then takes chunks from its own XCD's segment first and from the others' once that is used up, so
// that an XCD's 4 MB L2 sees ONE image region at a time.  Measured with 8 segments (profiles/r02o_xcd_segments.txt):
// +3 % on the 4K x 16 spp frame, +20..50 % on small frames (unequal regions, up to 8 failed claims per wavefront at the
// end) -- L2 locality is not what limits the kernel.  Built with one segment (= one device-wide counter).
#ifndef TRR_WORK_SEGMENTS
#define TRR_WORK_SEGMENTS 1
#endif
constexpr uint32_t WORK_SEGMENTS = TRR_WORK_SEGMENTS;   // power of two; the host reserves 8 counters per launch
struct WorkClaim {
    uint32_t seg, tries, chunk, chunk_min, seg_len, waves_per_seg;   // all wave-uniform
};
DEV WorkClaim work_claim_init(uint32_t total, uint32_t n_waves, uint32_t chunk_min) {
    WorkClaim c;
    c.seg = blockIdx.x & (WORK_SEGMENTS - 1u);
    c.tries = 0;
    c.chunk_min = chunk_min;
    c.seg_len = (((total + WORK_SEGMENTS - 1u) / WORK_SEGMENTS) + 63u) & ~63u;
    c.waves_per_seg = max(1u, n_waves / WORK_SEGMENTS);
    c.chunk = max(chunk_min, min(CHUNK_MAX, c.seg_len / (c.waves_per_seg * CHUNK_DIV)));
    return c;
}
// Claims the wavefront's next range [w_next, w_end) of work items; false once every segment is used up (each wavefront finds
// that out with at most WORK_SEGMENTS failed claims, its exit condition).
DEV bool work_claim_next(WorkClaim& c, uint32_t* counters, uint32_t total, uint32_t lane, uint32_t& w_next, uint32_t& w_end) {
    while (c.tries < WORK_SEGMENTS) {
        const uint32_t seg_base = c.seg * c.seg_len;
        if (seg_base < total) {
            const uint32_t seg_end = min(seg_base + c.seg_len, total);
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(counters + c.seg, c.chunk);
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            if (base < seg_end - seg_base) {
                w_next = seg_base + base;
                w_end = min(w_next + c.chunk, seg_end);
                c.chunk = max(c.chunk_min, min(CHUNK_MAX, (seg_end - w_end) / (c.waves_per_seg * CHUNK_DIV)));
                return true;
            }
        }
        c.seg = (c.seg + 1u) & (WORK_SEGMENTS - 1u);
        c.tries++;
    }
    return false;
}

// The kernel's one argument.  The traversal steps of the loop need four of its scalars (nodes, tris, delta, n_objs); the
// rest -- some 70 dwords -- is read only by the blocks that start and end rays.  Kept as ordinary kernel arguments they
// all stay in scalar registers for the whole loop, which overflowed the 102 SGPRs (19 spilled into vector-register lanes,
// 249 v_readlane / v_writelane in the loop, 134 VGPRs).  The ray start / end blocks therefore read their parameters from
// the kernel-argument segment through a pointer that the optimiser cannot see through (cold_args), so those scalar loads
// stay inside the blocks.
struct TraceArgs {
    FrameParams fp;
    BatchParams bp;
    TraceIO io;
};
DEV const TraceArgs* cold_args() {
    auto p = (const TraceArgs __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return (const TraceArgs*)p;
}

template <bool COUNT>
__global__ void __launch_bounds__(TRACE_THREADS) TRR_OCCUPANCY_ATTR k_trace(TraceArgs args) {
    __shared__ uint2 s_stack[STACK_LDS * TRACE_THREADS];
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave_global = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (TRACE_THREADS / 64) + tid / 64));
    // the only parameters the traversal steps use
    const Node4* const g_nodes = args.fp.nodes;
    const TriRec* const g_tris = args.fp.tris;
    const float delta = args.fp.delta;
    const uint32_t n_objs = args.fp.n_objs;
    constexpr bool QN = false;                      // this kernel reads the exact nodes (trace_step.inl)
    const QNode4* const g_qnodes = nullptr;
    const float e_ray = 0.0f;
    uint2* my_spill;
    uint32_t total, chunk_min;
    WorkClaim claim;
    bool from_q;
    {
        const TraceArgs* ca = cold_args();
        my_spill = ca->io.spill + ((size_t)blockIdx.x * TRACE_THREADS + tid) * STACK_SPILL;
        // Work items of the launch: the batch's camera rays, or -- in a follow-up launch -- the queue entries that the
        // launch before it spilled (io.src_q).  Once a wavefront can claim no more work items, everything it would push
        // onto its own queue goes to the device-wide spill queue instead (io.spill_q, when present): the rays of the
        // launch's tail are then dealt evenly over all wavefronts by the next launch (a kernel boundary makes them visible
        // for free), instead of each wavefront finishing its own, very unequal, backlog.
        from_q = ca->io.src_q != nullptr;
        total = from_q ? min(*ca->io.src_n, ca->io.src_cap) + (ca->io.src_q2 ? min(*ca->io.src_n2, ca->io.src_cap2) : 0u)
                       : (ca->io.groups != nullptr ? *ca->io.n_groups * 64u : ca->bp.n_samples);
        chunk_min = ca->io.chunk_min;
        claim = work_claim_init(total, gridDim.x * (TRACE_THREADS / 64), chunk_min);
    }
    // This wavefront's ray queue is a ring of io.qcap entries at io.wqueue + wave_global * io.qcap that only this wavefront
    // reads and writes, so no store has to become visible to another CU inside the launch (per-XCD L2s are not coherent
    // with each other).
    uint32_t q_head = 0, q_tail = 0;   // wave-uniform, free-running

    // a launch with little (or no) work: wavefronts beyond one per chunk_min items leave before touching the work counter
    // (5120 same-address atomics cost ~40 us, which is most of an empty launch)
    if (wave_global * chunk_min >= total) return;

    // ---- per-lane state (results go to memory as soon as a raycast ends; only the winner's slot/key stay live) ----
    // Per-lane booleans live in ONE register (bits of `fl`), and "this lane has no ray" is a value of `cur` (REF_IDLE):
    // as separate bools they become SGPR lane masks that hipcc re-merges at every join of the loop (3 scalar
    // instructions per bool and join, ~100 per iteration).  Bits 16..31 of `fl` count the raycasts of the march.
    constexpr uint32_t F_FIRST = 1u, F_GHAVE = 2u, F_OHAVE = 4u, F_SKIPBOX = 8u;
    constexpr uint32_t REF_IDLE = 0xFFFFFFFEu;           // never a leaf reference: slots stay below 2^29 - 1 (api.hip)
    uint32_t fl = 0;
    bool exhausted = false;                              // wave-uniform
    uint32_t w_next = 0, w_end = 0;                      // wave-uniform: this wavefront's claimed range of camera rays
    F3 ro = f3(0, 0, 0), rd = f3(0, 0, 0);      // world ray (origin advances during the march)
    uint32_t r_sample = 0, r_meta = 0;
    // one raycast (all objects)
    uint32_t obj_i = 0;
    float g_key = 0, t_lim = INFINITY;
    uint32_t g_slot = 0, g_obj = 0;
    // one object
    F3 oo = f3(0, 0, 0), od = f3(0, 0, 0), oi = f3(0, 0, 0);
    float len_d = 1.0f, lim_n = INFINITY, slack_abs = 0.0f;
    float o_key = 0;
    uint32_t o_slot = 0, o_tri = 0;
    uint32_t cur = REF_IDLE;
    int sp = 0;
    unsigned long long c_nodes = 0, c_boxes = 0, c_tris = 0, c_march = 0;
    unsigned long long c_wnode = 0, c_wtri = 0, c_wouter = 0, c_lanes = 0, c_wtrans = 0, c_ltrans = 0, c_wuni = 0;   // wave-level step counts (SIMD efficiency)
    uint32_t n_pushed_shadow = 0, n_pushed_chain = 0, n_overflow = 0, n_culled = 0;   // wave-uniform

    auto push = [&](uint32_t ref, float tmin) {
        uint2 e = make_uint2(ref, __float_as_uint(tmin));
        if (sp < STACK_LDS) s_stack[sp * TRACE_THREADS + tid] = e;
        else if (sp < STACK_LDS + STACK_SPILL) my_spill[sp - STACK_LDS] = e;
        else return;   // unreachable: trr_scene_create refuses trees whose stack need exceeds TRACE_STACK_ENTRIES
        ++sp;
    };
    auto pop = [&]() -> uint32_t {
        while (sp > 0) {
            --sp;
            uint2 e;
            if (sp < STACK_LDS) e = s_stack[sp * TRACE_THREADS + tid];
            else {
                e = my_spill[sp - STACK_LDS];
                // an opaque use keeps the two branches apart (ds_read above, global_load here); merged into a select of
                // addresses they become one flat_load, which cost 13 % of the frame time
                asm volatile("" : "+v"(e.x), "+v"(e.y));
            }
            if (!(__uint_as_float(e.y) > lim_n)) return e.x;
        }
        return REF_EMPTY;
    };

    // a ray from a queue entry
    auto start_secondary = [&](const float4* e) {
        const float4 a = e[0], b = e[1];
        ro = f3(a.x, a.y, a.z);
        rd = f3(a.w, b.x, b.y);
        r_sample = __float_as_uint(b.z);
        r_meta = __float_as_uint(b.w);
        fl = F_FIRST;          // no raycast of this ray has finished yet; no hit of any kind
        obj_i = 0xFFFFFFFFu;   // "before the first object"
        t_lim = INFINITY;
        cur = REF_EMPTY;
        sp = 0;
    };

#ifdef TRR_PAD_VGPRS
    float pad_regs[TRR_PAD_VGPRS];
#pragma unroll
    for (int k = 0; k < TRR_PAD_VGPRS; ++k) pad_regs[k] = (float)(tid + k);
#endif
    for (;;) {
#ifdef TRR_PAD_VGPRS
#pragma unroll
        for (int k = 0; k < TRR_PAD_VGPRS; ++k) asm volatile("" : "+v"(pad_regs[k]));
#endif
        // ---- re-fill idle lanes: first from this wavefront's own queue (secondary rays of hits it shaded), then, once
        // that is drained, with camera rays claimed from the device-wide counter ---------------------------------------
        const unsigned long long idle = __ballot(cur == REF_IDLE);
        const uint32_t nq = q_tail - q_head;   // (nothing is pushed onto the own queue any more once `exhausted` spills)
        if (idle != 0ull && (nq != 0u || !exhausted)) {
            const int nidle = __popcll(idle);
            if (nidle >= REFILL_MIN) {
                const TraceArgs* ca = cold_args();
                const FrameParams& fp = ca->fp;
                const BatchParams& bp = ca->bp;
                const TraceIO& io = ca->io;
                const uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
                const uint32_t take_q = min((uint32_t)nidle, nq);
                if (take_q != 0u) {
                    // entries of this queue were stored by this wavefront (other lanes): its stores have left the CU
                    // before the loads below are issued
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    const float4* my_q = io.wqueue + (size_t)wave_global * io.qcap * 2u;
                    if (cur == REF_IDLE && rank < take_q) start_secondary(my_q + (size_t)((q_head + rank) & (io.qcap - 1u)) * 2u);
                    q_head += take_q;
                }
                const uint32_t rest = (uint32_t)nidle - take_q;   // > 0 only when the queue has just been drained
                if (rest != 0u && !exhausted) {
                    // camera rays are claimed from the device counter in chunks per wavefront (one same-address atomic per
                    // chunk, not per re-fill: 10^6s of atomics on one word serialise in L2) and dealt to idle lanes
                    if (w_next >= w_end && !work_claim_next(claim, io.work_counter, total, lane, w_next, w_end)) {
                        exhausted = true;
                        w_next = w_end = 0;
                    }
                    const uint32_t take = min(rest, w_end - w_next);
                    const uint32_t idx = w_next + (rank - take_q);
                    if (from_q) {
                        const uint32_t n1 = min(*io.src_n, io.src_cap);   // entries of the first source queue
                        if (cur == REF_IDLE && rank >= take_q && rank - take_q < take) {
                            if (io.src_compact != 0u && idx < n1) {
                                // compact shadow entry (levels-then-drain): rebuild the ray of mod.rs:149-153
                                const float4 a = io.src_q[idx];
                                const uint32_t wd = __float_as_uint(a.w), l = wd >> COMPACT_SAMPLE_BITS;
                                const LightRec* lt = fp.lights + l;
                                const F3 P = f3(a.x, a.y, a.z);
                                const F3 ld = normalize(f3(lt->pos[0], lt->pos[1], lt->pos[2]) - P);
                                ro = P + scale(ld, delta);
                                rd = ld;
                                r_sample = wd & ((1u << COMPACT_SAMPLE_BITS) - 1u);
                                r_meta = io.src_level | META_SHADOW | (l << 16);
                                fl = F_FIRST;
                                obj_i = 0xFFFFFFFFu;
                                t_lim = INFINITY;
                                cur = REF_EMPTY;
                                sp = 0;
                            } else {
                                start_secondary(idx < n1 ? io.src_q + (size_t)idx * 2u : io.src_q2 + (size_t)(idx - n1) * 2u);
                            }
                        }
                    } else if (cur == REF_IDLE && rank >= take_q && rank - take_q < take &&
                               (io.groups == nullptr || io.groups[idx >> 6] * 64u + (idx & 63u) < bp.n_samples)) {
                        const uint32_t sidx = io.groups == nullptr ? idx : io.groups[idx >> 6] * 64u + (idx & 63u);   // camera-ray cull
                        const uint32_t pix = sidx / fp.spp, si = sidx - pix * fp.spp;
                        const PixelPos pp = pixel_of(fp, bp, pix);
                        ro = f3(fp.cam_pos[0], fp.cam_pos[1], fp.cam_pos[2]);
                        rd = primary_dir(fp, pp.x, pp.y, si);
                        r_sample = sidx;
                        r_meta = 0u;           // chain ray of level 0
                        fl = F_FIRST;
                        obj_i = 0xFFFFFFFFu;
                        t_lim = INFINITY;
                        cur = REF_EMPTY;
                        sp = 0;
                    }
                    w_next += take;
                }
            }
        }
        if (__ballot(cur != REF_IDLE) == 0ull) {
            // every lane idle: 64 >= REFILL_MIN, so the re-fill above has run; it found nothing only if the queue is empty
            // and the camera rays are used up -- the exit every wavefront reaches
            if (q_tail == q_head && exhausted) break;
            continue;
        }
        if (COUNT) {
            if (lane == 0) c_wouter++;
            if (cur != REF_IDLE) c_lanes++;
        }

        // ---- transition: object finished / raycast finished / ray finished ----------------------
        // The transition block is long (winner re-evaluation, hit set-up, object-space ray, divisions) and costs a
        // wavefront the same whether 1 or 64 lanes need it.  Lanes that only have to END their ray (last object done) can
        // wait: they would sit idle until the next re-fill anyway.  The block runs when a lane needs it to go on (fresh
        // ray, next object), when the waiting + idle lanes reach the re-fill threshold, or when no lane has traversal
        // work left.
        bool run_trans;
        {
            const bool is_empty = cur == REF_EMPTY;
#if TRR_DEFER_END
            const bool can_wait = is_empty && obj_i != 0xFFFFFFFFu && obj_i + 1u >= n_objs;
#else
            const bool can_wait = false;
#endif
            const unsigned long long m_empty = __ballot(is_empty);
            const unsigned long long m_wait = __ballot(can_wait);
            const unsigned long long m_work = __ballot(cur < REF_IDLE);
            run_trans = m_empty != 0ull && (m_empty != m_wait || m_work == 0ull ||
                                            __popcll(m_wait) + __popcll(idle) >= REFILL_MIN);
        }
        if (run_trans) {
            const TraceArgs* ca = cold_args();
            const FrameParams& fp = ca->fp;
            const BatchParams& bp = ca->bp;
            const TraceIO& io = ca->io;
            const uint32_t L = fp.n_lights;
            const size_t cap = bp.cap;
            if (COUNT) {
                unsigned long long tm = __ballot(cur == REF_EMPTY);
                if (tm != 0ull && lane == (uint32_t)__ffsll((long long)tm) - 1u) c_wtrans++;
                if (cur == REF_EMPTY) c_ltrans++;
            }
            // what the end of a chain ray leaves for the hit set-up below
            bool final_hit = false;          // this lane's chain ray ended on a hit that is to be shaded
            F3 h_org = f3(0, 0, 0);          // origin of the raycast that found it
            uint32_t h_slot = 0, h_obj = 0, h_gid = 0, h_flags = 0;   // its triangle: slot, object, global id, TriRec.flags
            int32_t h_mat = -1;
            if (cur == REF_EMPTY) {
                // object.rs:206-219: best candidate of the object -> world space, then mod.rs:50 ranks objects by
                // |point_world - ray.origin| (first object wins ties: strict <, NaN greatest)
                // With a single object there is nothing to rank: the winner of the object is the winner of the raycast.
                const bool rank_free = n_objs == 1u;
                if ((fl & F_OHAVE) && rank_free) {
                    fl = (fl | F_GHAVE) & ~F_OHAVE;
                    g_slot = o_slot;
                    g_obj = 0u;
                }
                if (fl & F_OHAVE) {
                    const ObjRec* ob = fp.objs + obj_i;
                    const float4* tp = reinterpret_cast<const float4*>(g_tris + o_slot);
                    float4 q0 = tp[0], q1 = tp[1], q2 = tp[2], q3 = tp[3];
                    TriHit h;
                    h.u = h.v = h.w = 0.0f; h.point = f3(0, 0, 0); h.key = 0.0f;
                    (void)tri_test(q0, q1, q2, q3.z, oo, od, delta, h);   // same inputs as in the leaf loop -> same bits
                    F3 pw = xform_point(ob->fwd, h.point);
                    float key = norm(pw - ro);
                    bool better;
                    if (!(fl & F_GHAVE)) better = true;
                    else if (is_nan(key)) better = false;
                    else if (is_nan(g_key)) better = true;
                    else better = key < g_key;
                    if (better) {
                        fl |= F_GHAVE; g_key = key; g_slot = o_slot; g_obj = obj_i;
                        // later objects only need boxes nearer than this hit (ray parameter + slack)
                        float th = sqrtf(h.key) / len_d;
                        if (th == th) t_lim = th * (1.0f + PRUNE_REL);
                    }
                    fl &= ~F_OHAVE;
                }
                ++obj_i;
                if (obj_i >= n_objs) {
                    // raycast done (mod.rs:45-51); raycast_transparent loop (mod.rs:100-111): only .last() is consumed
                    const bool is_shadow = (r_meta & META_SHADOW) != 0u;
                    const uint32_t level = r_meta & 0xFFu;
                    const bool first_cast = (fl & F_FIRST) != 0u;
                    bool again = false, final_miss = false;
                    if (fl & F_GHAVE) {
                        const float4 q3 = reinterpret_cast<const float4*>(g_tris + g_slot)[3];
                        const uint32_t gid = __float_as_uint(q3.x);
                        const int32_t mat = (int32_t)__float_as_uint(q3.y);
                        const bool transp = (__float_as_uint(q3.w) & TRI_FLAG_TRANSPARENT) != 0u;
                        if (r_meta == 0u && first_cast && io.first_obj != nullptr) {
                            // id buffers: the camera ray's FIRST closest hit, per pixel (1 spp only); triangle id = row in
                            // the object's triangle array
                            const PixelPos pp = pixel_of(fp, bp, r_sample / fp.spp);
                            io.first_obj[pp.y * fp.width + pp.x] = g_obj;
                            io.first_tri[pp.y * fp.width + pp.x] = gid - fp.objs[g_obj].tri_base;
                        }
                        again = transp && (fl >> 16) < MARCH_CAP;
                        if (is_shadow) {
                            // only the last occluder of the march counts: every raycast with a hit overwrites the code
                            io.occ[((size_t)level * L + (r_meta >> 16)) * cap + r_sample] = mat >= 0 ? mat : -2;
                        } else if (again) {
                            // the next raycast may find nothing, and then THIS hit is the ray's result: keep it in memory
                            // (rare: illum 6/7 surfaces only), with the origin this raycast used
                            float4* dst = reinterpret_cast<float4*>(io.hits + r_sample);
                            dst[0] = make_float4(ro.x, ro.y, ro.z, __uint_as_float(g_slot));
                            dst[1] = make_float4(__uint_as_float(g_obj), __uint_as_float(gid), 0.0f, 0.0f);
                        } else {
                            final_hit = true;
                            h_org = ro; h_slot = g_slot; h_obj = g_obj;
                            h_gid = gid; h_mat = mat; h_flags = __float_as_uint(q3.w);
                        }
                        if (again) {
                            ro = ro + scale(rd, 0.05f);   // mod.rs:106
                            fl += 0x10000u;
                            if (COUNT) c_march++;
                        }
                    } else if (first_cast) {
                        if (is_shadow) io.occ[((size_t)level * L + (r_meta >> 16)) * cap + r_sample] = -1;
                        else {
                            final_miss = true;
                            if (r_meta == 0u && io.first_obj != nullptr) {
                                const PixelPos pp = pixel_of(fp, bp, r_sample / fp.spp);
                                io.first_obj[pp.y * fp.width + pp.x] = 0xFFFFFFFFu;
                                io.first_tri[pp.y * fp.width + pp.x] = 0xFFFFFFFFu;
                            }
                        }
                    } else if (!is_shadow) {
                        // the march ended on a raycast that found nothing: the previous (transparent) hit is the result
                        const float4* src = reinterpret_cast<const float4*>(io.hits + r_sample);
                        const float4 a = src[0], b = src[1];
                        final_hit = true;
                        h_org = f3(a.x, a.y, a.z); h_slot = __float_as_uint(a.w); h_obj = __float_as_uint(b.x);
                        const float4 q3 = reinterpret_cast<const float4*>(g_tris + h_slot)[3];
                        h_gid = __float_as_uint(q3.x); h_mat = (int32_t)__float_as_uint(q3.y); h_flags = __float_as_uint(q3.w);
                    }
                    // mod.rs:125-127: with max_depth == 0 even a hit is shaded as skybox(ray.direction)
                    if (final_hit && fp.max_depth == 0u) { final_hit = false; final_miss = true; }
                    if (final_miss) {
                        // mod.rs:119: no hit -> skybox(ray.direction) ends the chain; the resolve pass looks the sky up with
                        // full wavefronts.  A camera ray that misses only records the fact (the resolve pass rebuilds its
                        // direction from the sample index); a reflection ray that misses leaves its direction.
                        if (level == 0u) io.nlev[r_sample] = 0;
                        else if (fp.sky != nullptr) io.term[r_sample] = make_float4(rd.x, rd.y, rd.z, 0.0f);
                    }
                    fl &= ~(F_FIRST | F_GHAVE);
                    if (again) {
                        obj_i = 0;
                        t_lim = INFINITY;
                    } else {
                        cur = REF_IDLE;
                    }
                }
                if (cur != REF_IDLE) {
                    // object.rs:186-195: ray into object space (direction NOT renormalised), bvh::Ray::new
                    // normalises its own copy and keeps 1/direction [crate]
                    const ObjRec* ob = fp.objs + obj_i;
                    oo = xform_point(ob->inv, ro);
                    od = xform_vec(ob->inv, rd);
                    len_d = norm(od);
                    F3 dn = unscale(od, len_d);
                    oi = f3(1.0f / dn.x, 1.0f / dn.y, 1.0f / dn.z);
                    slack_abs = ob->slack_abs;
                    lim_n = t_lim * len_d + slack_abs;
                    if (!(lim_n == lim_n)) lim_n = INFINITY;
                    fl = (fl & ~(F_OHAVE | F_SKIPBOX)) | ((ob->flags & 1u) ? F_SKIPBOX : 0u);
                    cur = ob->root;
                    sp = 0;
                }
            }

            // ---- hit set-up (mod.rs:124-153, 199-206; object.rs:198-219) of the chain rays that have just ended on a hit:
            // shaded-hit record of the level, then L shadow rays and at most one reflection ray into this wavefront's queue
            const unsigned long long hm = __ballot(final_hit);
            if (hm != 0ull) {
                bool spawn = false;
                const uint32_t level = r_meta & 0xFFu;
                F3 P = f3(0, 0, 0), N = f3(0, 0, 0);
                if (final_hit) {
                    // three independent loads (triangle, object, interpolation record): ONE memory round trip
                    const ObjRec* ob = fp.objs + h_obj;
                    const float4* tp = reinterpret_cast<const float4*>(g_tris + h_slot);
                    const float4* ipp = reinterpret_cast<const float4*>(fp.interp + h_gid);
                    float4 q0 = tp[0], q1 = tp[1], q2 = tp[2];
                    float4 i0 = ipp[0], i1 = ipp[1], i2 = ipp[2], i3 = ipp[3];
                    const float len = reinterpret_cast<const float*>(tp)[14];
                    F3 a = f3(q0.x, q0.y, q0.z), b = f3(q0.w, q1.x, q1.y), c = f3(q1.z, q1.w, q2.x);
                    // the winner's barycentrics (triangle.rs:84) from the object-space ray of the raycast that found it
                    // (object.rs:186-189): the same tri_test on the same inputs as in the leaf loop, so the same bits
                    TriHit th;
                    th.u = th.v = th.w = 0.0f; th.point = f3(0, 0, 0); th.key = 0.0f;
                    (void)tri_test(q0, q1, q2, len, xform_point(ob->inv, h_org), xform_vec(ob->inv, rd), delta, th);
                    const float u = th.u, v = th.v, w = th.w;
                    F3 na = f3(i0.x, i0.y, i0.z), nb = f3(i0.w, i1.x, i1.y), nc = f3(i1.z, i1.w, i2.x);
                    // object.rs:201-203
                    F3 point = (scale(a, u) + scale(b, v)) + scale(c, w);
                    F3 normal = (scale(na, u) + scale(nb, v)) + scale(nc, w);
                    float tu = (i2.y * u + i2.w * v) + i3.y * w;
                    float tv = (i2.z * u + i3.x * v) + i3.z * w;
                    // object.rs:209-210 (forward matrix for the normal too)
                    P = xform_point(ob->fwd, point);
                    N = xform_vec(ob->fwd, normal);
                    const bool reflective = (h_flags & TRI_FLAG_REFLECTIVE) != 0u;   // material.rs:47-49
                    float4* lv = io.level + ((size_t)level * cap + r_sample) * LEVEL_F4;
                    lv[0] = make_float4(P.x, P.y, P.z, N.x);
                    lv[1] = make_float4(N.y, N.z, rd.x, rd.y);
                    lv[2] = make_float4(rd.z, tu, tv, __int_as_float(h_mat));
                    io.nlev[r_sample] = (uint8_t)(level + 1u);
                    if (reflective) {
                        if (level + 1u < fp.max_depth) spawn = true;
                        else if (fp.sky != nullptr) {
                            // mod.rs:125-127: the nested shade() is at max_depth -> skybox(reflection direction)
                            F3 r2 = reflect(rd, N);
                            io.term[r_sample] = make_float4(r2.x, r2.y, r2.z, 0.0f);
                        }
                    }
                }
                const unsigned long long sm = __ballot(spawn);
                const uint32_t nh = (uint32_t)__popcll(hm), ns = (uint32_t)__popcll(sm);
                // shadow-ray cull (see k_setup): lights whose two intensities are exactly 0.0 at a non-reflective hit
                const bool do_cull = io.cull != 0u && L <= 64u;
                unsigned long long keep = 0ull;   // per lane: bit l = light l's shadow ray is traced
                uint32_t n_kept = nh * L;
                if (do_cull) {
                    if (final_hit) {
                        bool specular = false;
                        float ns_exp = 1.0f;
                        if (h_mat >= 0) {
                            const MatRec* m = fp.mats + h_mat;
                            if (m->flags & 4u) ns_exp = m->ns;
                            specular = (m->illum == 2);
                        }
                        const bool reflective = (h_flags & TRI_FLAG_REFLECTIVE) != 0u;
                        const F3 rv = reflect(neg(rd), N);
                        for (uint32_t l = 0; l < L; ++l) {
                            const LightRec* lt = fp.lights + l;
                            const F3 lp = f3(lt->pos[0], lt->pos[1], lt->pos[2]);
                            const F3 ld = normalize(lp - P);
                            const float li = lt->intensity / norm_squared(lp - P);
                            const float di = rmax(dot(ld, N), 0.0f) * li;
                            bool dark = !reflective && di == 0.0f;
                            if (dark && specular) {
                                const float sint = powf(rmax(dot(ld, rv), 0.0f), ns_exp) * li;
                                dark = sint == 0.0f;
                            }
                            if (dark) io.occ[((size_t)level * L + l) * cap + r_sample] = OCC_CULLED;
                            else keep |= 1ull << l;
                        }
                    }
                    n_kept = 0;
                    for (uint32_t l = 0; l < L; ++l) n_kept += (uint32_t)__popcll(__ballot((keep >> l) & 1ull));
                }
                const uint32_t need = n_kept + ns;
                // destination: the wavefront's own ring, or -- once it can claim no more work items -- the spill queue of
                // the launch (one returning atomic per wavefront and set-up)
                const bool spill = exhausted && io.spill_q != nullptr;
                float4* dst = io.wqueue + (size_t)wave_global * io.qcap * 2u;
                uint32_t base = q_tail, mask = io.qcap - 1u;
                bool room = true;
                if (spill) {
                    uint32_t b0 = 0;
                    if (lane == 0) b0 = atomicAdd(io.spill_n, need);
                    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)b0);
                    dst = io.spill_q;
                    mask = 0xFFFFFFFFu;
                    room = base + need <= io.spill_cap;
                } else {
                    room = (q_tail - q_head) + need <= io.qcap;
                    q_tail += need;
                }
                if (!room) n_overflow++;   // cannot happen (capacities are proven bounds, see api.hip); counted, never out of bounds
                if (do_cull && room) {
                    uint32_t off = 0;
                    for (uint32_t l = 0; l < L; ++l) {
                        const bool mine = ((keep >> l) & 1ull) != 0ull;
                        const unsigned long long km = __ballot(mine);
                        if (mine) {
                            const LightRec* lt = fp.lights + l;
                            const F3 ld = normalize(f3(lt->pos[0], lt->pos[1], lt->pos[2]) - P);   // mod.rs:149
                            const F3 o = P + scale(ld, delta);                                     // mod.rs:151
                            float4* e = dst + (size_t)((base + off + (uint32_t)__popcll(km & ((1ull << lane) - 1ull))) & mask) * 2u;
                            e[0] = make_float4(o.x, o.y, o.z, ld.x);
                            e[1] = make_float4(ld.y, ld.z, __uint_as_float(r_sample), __uint_as_float(level | META_SHADOW | (l << 16)));
                        }
                        off += (uint32_t)__popcll(km);
                    }
                } else if (final_hit && room) {
                    // shadow rays: mod.rs:149-153, one per light, grouped per light (coherent re-fills)
                    const uint32_t rk = (uint32_t)__popcll(hm & ((1ull << lane) - 1ull));
                    for (uint32_t l = 0; l < L; ++l) {
                        const LightRec* lt = fp.lights + l;
                        const F3 ld = normalize(f3(lt->pos[0], lt->pos[1], lt->pos[2]) - P);   // mod.rs:149
                        const F3 o = P + scale(ld, delta);                                     // mod.rs:151
                        float4* e = dst + (size_t)((base + l * nh + rk) & mask) * 2u;
                        e[0] = make_float4(o.x, o.y, o.z, ld.x);
                        e[1] = make_float4(ld.y, ld.z, __uint_as_float(r_sample), __uint_as_float(level | META_SHADOW | (l << 16)));
                    }
                }
                if (spawn && room) {
                    // reflection ray: mod.rs:203-206 (traced once per hit; the value is light-independent)
                    const uint32_t rk = (uint32_t)__popcll(sm & ((1ull << lane) - 1ull));
                    const F3 o = P + scale(N, delta), d = reflect(rd, N);
                    float4* e = dst + (size_t)((base + n_kept + rk) & mask) * 2u;
                    e[0] = make_float4(o.x, o.y, o.z, d.x);
                    e[1] = make_float4(d.y, d.z, __uint_as_float(r_sample), __uint_as_float(level + 1u));
                }
                n_pushed_shadow += nh * L;
                n_culled += nh * L - n_kept;
                n_pushed_chain += ns;
            }
        }

#include "trace_step.inl"
    }

#ifdef TRR_PAD_VGPRS
#pragma unroll
    for (int k = 0; k < TRR_PAD_VGPRS; ++k) asm volatile("" ::"v"(pad_regs[k]));
#endif
    // rays this wavefront queued (the ray statistics of trr_stats) and the queue's overflow count (always 0)
    const TraceIO& io = cold_args()->io;
    if (lane == 0) {
        atomicAdd(io.counters + 11, (unsigned long long)n_pushed_shadow);
        atomicAdd(io.counters + 12, (unsigned long long)n_pushed_chain);
        if (n_overflow) atomicAdd(io.counters + 13, (unsigned long long)n_overflow);
        if (n_culled) atomicAdd(io.counters + KC_SHADOW_CULLED, (unsigned long long)n_culled);
    }
    if (COUNT) {
        // wave reduction, then one atomic per wavefront and counter
        for (int off = 32; off > 0; off >>= 1) {
            c_nodes += __shfl_down(c_nodes, off);
            c_boxes += __shfl_down(c_boxes, off);
            c_tris += __shfl_down(c_tris, off);
            c_march += __shfl_down(c_march, off);
            c_wnode += __shfl_down(c_wnode, off);
            c_wtri += __shfl_down(c_wtri, off);
            c_wouter += __shfl_down(c_wouter, off);
            c_lanes += __shfl_down(c_lanes, off);
            c_wtrans += __shfl_down(c_wtrans, off);
            c_ltrans += __shfl_down(c_ltrans, off);
            c_wuni += __shfl_down(c_wuni, off);
        }
        if (lane == 0) {
            atomicAdd(io.counters + 0, c_nodes);
            atomicAdd(io.counters + 1, c_boxes);
            atomicAdd(io.counters + 2, c_tris);
            atomicAdd(io.counters + 3, c_march);
            atomicAdd(io.counters + 4, c_wnode);
            atomicAdd(io.counters + 5, c_wtri);
            atomicAdd(io.counters + 6, c_wouter);
            atomicAdd(io.counters + 7, c_lanes);
            atomicAdd(io.counters + 8, c_wtrans);
            atomicAdd(io.counters + 9, c_ltrans);
            atomicAdd(io.counters + 10, c_wuni);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The level-synchronous schedule (big batches): specialised kernels, one launch per ray population.
//   k_trace_level<RAYGEN>   camera rays                      -> HitRec per sample
//   per level: k_setup      HitRec -> shaded-hit record, L shadow rays + <= 1 reflection ray into device-wide queues
//              k_trace_level<QUEUE>  shadow rays of the level + reflection rays of the next in ONE launch
// Every wavefront does one thing (thin ray start / end blocks, hit set-up with full wavefronts), which makes a ray cheaper
// than in k_trace; the price is 2 * max_depth + 1 dependent launches per batch, each with a ramp-up and a tail.
// ------------------------------------------------------------------------------------------------
// ST ("staged ray starts", secondary-ray launches only): a wavefront turns its next 64 queue entries into started rays with
// ALL 64 lanes at once -- the ray of mod.rs:149-153 / 203-206, its direction in object 0's space, the norm and the six
// divisions of bvh::Ray::new (object.rs:186-195) -- and parks them in LDS (48 B per ray); a lane whose ray has ended then
// STARTS its next ray with three LDS reads and the two matrix products, no division, no square root, no memory round trip.
// The arithmetic of a ray is the same as in the transition block below (same functions, same operands), so the same bits.
// The idea: ray ends no longer have to wait for 42 lanes of company to make the long start block worth its while (END_MIN
// instead of REFILL_MIN) -- finished lanes are 25 % of all lane-steps of the 4K x 16 spp frame.  Measured: no gain on its
// own (lanes that start together stay in step, which the wave-uniform vote needs more than it needs busy lanes) and 8 %
// slower with three batches in flight (its 12 KB of LDS crowd out the other batches' blocks): opt-in, TRR_STAGE=1.
template <bool RAYGEN, bool COUNT, bool QN, bool ST>
__global__ void __launch_bounds__(TRACE_THREADS) TRR_OCCUPANCY_ATTR k_trace_level(FrameParams fp, BatchParams bp, LevelIO io) {
    static_assert(!(RAYGEN && ST), "staged ray starts are for the secondary-ray launches");
    __shared__ uint2 s_stack[STACK_LDS * TRACE_THREADS];
    __shared__ float4 s_stage[ST ? 3 * TRACE_THREADS : 1];   // [3][wavefronts of the block][64 slots]
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    uint2* my_spill = io.spill + ((size_t)blockIdx.x * TRACE_THREADS + tid) * STACK_SPILL;
    const Node4* const g_nodes = fp.nodes;
    const QNode4* const g_qnodes = fp.qnodes;
    const TriRec* const g_tris = fp.tris;
    const float delta = fp.delta;

    uint32_t n_shadow = 0, total;
    if (RAYGEN) total = io.groups != nullptr ? *io.n_groups * 64u : bp.n_samples;
    else {
        n_shadow = *io.n_shadow;
        total = n_shadow + *io.n_chain;
    }
    // compact shadow entries need the lights' positions at every ray start: one LDS read instead of a dependent load
    __shared__ float s_light_pos[RAYGEN ? 1 : 3 * COMPACT_MAX_LIGHTS];
    if (!RAYGEN && io.shadow_compact != 0u) {
        for (uint32_t k = tid; k < 3u * fp.n_lights; k += TRACE_THREADS) s_light_pos[k] = fp.lights[k / 3u].pos[k % 3u];
        __syncthreads();
    }

    // a launch with little (or no) work: wavefronts beyond one per 64 items leave before touching the work counter
    // (5120 same-address atomics cost ~40 us, which is most of an empty launch)
    if ((blockIdx.x * (TRACE_THREADS / 64) + tid / 64) * 64u >= total) return;

    // per-lane state: see k_trace
    constexpr uint32_t F_FIRST = 1u, F_GHAVE = 2u, F_OHAVE = 4u, F_SKIPBOX = 8u;
    constexpr uint32_t REF_IDLE = 0xFFFFFFFEu;
    uint32_t fl = 0;
    bool exhausted = false;                              // wave-uniform
    uint32_t w_next = 0, w_end = 0;                      // wave-uniform: this wavefront's claimed range of work items
    WorkClaim claim = work_claim_init(total, gridDim.x * (TRACE_THREADS / 64), 64u);
    F3 ro = f3(0, 0, 0), rd = f3(0, 0, 0);      // world ray (origin advances during the march)
    uint32_t r_out = 0, r_sample = 0, r_meta = 0;
    uint32_t obj_i = 0;
    float g_key = 0, t_lim = INFINITY;
    uint32_t g_slot = 0, g_obj = 0;
    F3 oo = f3(0, 0, 0), od = f3(0, 0, 0), oi = f3(0, 0, 0);
    float len_d = 1.0f, lim_n = INFINITY, slack_abs = 0.0f;
    float e_ray = 0.0f;   // QN: how far the quantised boxes are pushed out for this ray (trace_step.inl)
    float o_key = 0;
    uint32_t o_slot = 0, o_tri = 0;
    uint32_t cur = REF_IDLE;
    int sp = 0;
    unsigned long long c_nodes = 0, c_boxes = 0, c_tris = 0, c_march = 0;
    unsigned long long c_wnode = 0, c_wtri = 0, c_wouter = 0, c_lanes = 0, c_wtrans = 0, c_ltrans = 0, c_wuni = 0;

    auto push = [&](uint32_t ref, float tmin) {
        uint2 e = make_uint2(ref, __float_as_uint(tmin));
        if (sp < STACK_LDS) s_stack[sp * TRACE_THREADS + tid] = e;
        else if (sp < STACK_LDS + STACK_SPILL) my_spill[sp - STACK_LDS] = e;
        else return;   // unreachable: trr_scene_create refuses trees whose stack need exceeds TRACE_STACK_ENTRIES
        ++sp;
    };
    auto pop = [&]() -> uint32_t {
        while (sp > 0) {
            --sp;
            uint2 e;
            if (sp < STACK_LDS) e = s_stack[sp * TRACE_THREADS + tid];
            else {
                e = my_spill[sp - STACK_LDS];
                asm volatile("" : "+v"(e.x), "+v"(e.y));   // keeps ds_read and global_load apart (see k_trace)
            }
            if (!(__uint_as_float(e.y) > lim_n)) return e.x;
        }
        return REF_EMPTY;
    };

    // ST: object 0, where every ray starts (wave-uniform); the staging slots of this wavefront
    uint32_t root0 = REF_EMPTY, skip0 = 0u;
    float slack0 = 0.0f;
    if (ST) {
        root0 = fp.objs[0].root;
        skip0 = (fp.objs[0].flags & 1u) ? F_SKIPBOX : 0u;
        slack0 = fp.objs[0].slack_abs;
    }
    float4* const my_stage = s_stage + (ST ? (tid & ~63u) : 0u);
    uint32_t stage_head = 0, stage_n = 0;   // wave-uniform: staged rays not yet handed to a lane

    for (;;) {
        // ---- re-fill idle lanes -------------------------------------------------------------
        unsigned long long idle = __ballot(cur == REF_IDLE);
        if (ST && idle != 0ull) {
            if (stage_n == 0u && !exhausted) {
                // stage the wavefront's next (up to) 64 work items, one per lane, busy lanes included
                if (w_next >= w_end && !work_claim_next(claim, io.work_counter, total, lane, w_next, w_end)) {
                    exhausted = true;
                    w_next = w_end = 0;
                }
                const uint32_t n = min(64u, w_end - w_next);
                if (lane < n) {
                    const uint32_t idx = w_next + lane;
                    F3 o, d;
                    uint32_t rid, meta;
                    if (io.shadow_compact != 0u && idx < n_shadow) {
                        // compact shadow entry: the ray of mod.rs:149-153 from the shaded point, same operations as k_setup's
                        const float4 a = io.shadow_q[idx];
                        const uint32_t wd = __float_as_uint(a.w), l = wd >> COMPACT_SAMPLE_BITS;
                        const F3 P = f3(a.x, a.y, a.z);
                        const F3 ld = normalize(f3(s_light_pos[3u * l], s_light_pos[3u * l + 1u], s_light_pos[3u * l + 2u]) - P);   // mod.rs:149
                        o = P + scale(ld, fp.delta);                                                                             // mod.rs:151
                        d = ld;
                        rid = wd & ((1u << COMPACT_SAMPLE_BITS) - 1u);
                        meta = io.shadow_level | META_SHADOW | (l << 16);
                    } else {
                        const float4* q = (idx < n_shadow) ? (io.shadow_q + (size_t)idx * 2u) : (io.chain_q + (size_t)(idx - n_shadow) * 2u);
                        const float4 a = q[0], b = q[1];
                        o = f3(a.x, a.y, a.z);
                        d = f3(a.w, b.x, b.y);
                        meta = __float_as_uint(b.w);
                        // a shadow ray is known by its sample, a chain ray by its queue position (its hit record's index)
                        rid = (meta & META_SHADOW) ? __float_as_uint(b.z) : idx - n_shadow;
                    }
                    // object.rs:186-195 for object 0: bvh::Ray::new normalises its own copy and keeps 1/direction [crate]
                    const F3 od0 = xform_vec(fp.objs[0].inv, d);
                    const float ld0 = norm(od0);
                    const F3 dn = unscale(od0, ld0);
                    my_stage[lane] = make_float4(o.x, o.y, o.z, d.x);
                    my_stage[TRACE_THREADS + lane] = make_float4(d.y, d.z, 1.0f / dn.x, 1.0f / dn.y);
                    my_stage[2 * TRACE_THREADS + lane] = make_float4(1.0f / dn.z, ld0, __uint_as_float(rid), __uint_as_float(meta));
                }
                w_next += n;
                stage_head = 0;
                stage_n = n;
            }
            if (stage_n != 0u) {
                const uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
                const uint32_t take = min((uint32_t)__popcll(idle), stage_n);
                if (cur == REF_IDLE && rank < take) {
                    const uint32_t slot = stage_head + rank;
                    const float4 a = my_stage[slot], b = my_stage[TRACE_THREADS + slot], c = my_stage[2 * TRACE_THREADS + slot];
                    ro = f3(a.x, a.y, a.z);
                    rd = f3(a.w, b.x, b.y);
                    oi = f3(b.z, b.w, c.x);
                    len_d = c.y;
                    r_meta = __float_as_uint(c.w);
                    r_sample = r_out = __float_as_uint(c.z);
                    oo = xform_point(fp.objs[0].inv, ro);   // object.rs:187
                    od = xform_vec(fp.objs[0].inv, rd);     // object.rs:188
                    slack_abs = slack0;
                    t_lim = INFINITY;
                    lim_n = INFINITY;   // = t_lim * len_d + slack_abs, or its NaN fallback (the transition block's rule)
                    if (QN) e_ray = 0x1p-20f * rmax(rmax(fabsf(oo.x), fabsf(oo.y)), fabsf(oo.z)) + 0x1p-6f * slack_abs;
                    fl = F_FIRST | skip0;
                    obj_i = 0;
                    cur = root0;
                    sp = 0;
                }
                stage_head += take;
                stage_n -= take;
            }
        }
        if (!ST && idle != 0ull && !exhausted) {
            int nidle = __popcll(idle);
            if (nidle >= REFILL_MIN) {
                if (w_next >= w_end && !work_claim_next(claim, io.work_counter, total, lane, w_next, w_end)) {
                    exhausted = true;
                    w_next = w_end = 0;
                }
                const uint32_t take = min((uint32_t)nidle, w_end - w_next);
                uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
                uint32_t idx = w_next + rank;
                w_next += take;
                if (RAYGEN && io.groups != nullptr && rank < take) {
                    idx = io.groups[idx >> 6] * 64u + (idx & 63u);   // camera-ray cull: work item -> sample
                    if (idx >= bp.n_samples) rank = 0xFFFFFFFFu;     // the batch's last group may be short
                }
                if (cur == REF_IDLE && rank < take) {
                    if (RAYGEN) {
                        uint32_t pix = idx / fp.spp, si = idx - pix * fp.spp;
                        PixelPos pp = pixel_of(fp, bp, pix);
                        ro = f3(fp.cam_pos[0], fp.cam_pos[1], fp.cam_pos[2]);
                        rd = primary_dir(fp, pp.x, pp.y, si);
                        r_out = idx;
                        r_sample = pp.y * fp.width + pp.x;   // frame pixel index, used for the id buffers
                        r_meta = 0u;
                    } else if (io.shadow_compact != 0u && idx < n_shadow) {
                        // compact shadow entry: the ray of mod.rs:149-153 from the shaded point, same operations as k_setup's
                        const float4 a = io.shadow_q[idx];
                        const uint32_t wd = __float_as_uint(a.w), l = wd >> COMPACT_SAMPLE_BITS;
                        const F3 P = f3(a.x, a.y, a.z);
                        const F3 ld = normalize(f3(s_light_pos[3u * l], s_light_pos[3u * l + 1u], s_light_pos[3u * l + 2u]) - P);   // mod.rs:149
                        ro = P + scale(ld, fp.delta);                                                                            // mod.rs:151
                        rd = ld;
                        r_sample = wd & ((1u << COMPACT_SAMPLE_BITS) - 1u);
                        r_meta = io.shadow_level | META_SHADOW | (l << 16);
                        r_out = 0;
                    } else {
                        const float4* q = (idx < n_shadow) ? (io.shadow_q + (size_t)idx * 2u) : (io.chain_q + (size_t)(idx - n_shadow) * 2u);
                        float4 a = q[0], b = q[1];
                        ro = f3(a.x, a.y, a.z);
                        rd = f3(a.w, b.x, b.y);
                        r_sample = __float_as_uint(b.z);
                        r_meta = __float_as_uint(b.w);
                        r_out = idx - n_shadow;   // meaningful for chain rays only
                    }
                    fl = F_FIRST;
                    obj_i = 0xFFFFFFFFu;
                    t_lim = INFINITY;
                    cur = REF_EMPTY;
                    sp = 0;
                }
            }
        }
        if (__ballot(cur != REF_IDLE) == 0ull) break;
        if (COUNT) {
            if (lane == 0) c_wouter++;
            if (cur != REF_IDLE) c_lanes++;
#ifdef TRR_COUNT_WAIT
            if (cur == REF_EMPTY) c_wuni++;   // experiment: lanes whose traversal has ended and that wait for the transition block
#endif
        }

        // ---- transition: object finished / raycast finished / ray finished (deferred ends: see k_trace) ------------------
        bool run_trans;
        {
            const bool is_empty = cur == REF_EMPTY;
#if TRR_DEFER_END
            const bool can_wait = is_empty && obj_i != 0xFFFFFFFFu && obj_i + 1u >= fp.n_objs;
#else
            const bool can_wait = false;
#endif
            const unsigned long long m_empty = __ballot(is_empty);
            const unsigned long long m_wait = __ballot(can_wait);
            const unsigned long long m_work = __ballot(cur < REF_IDLE);
            run_trans = m_empty != m_wait || m_work == 0ull || __popcll(m_wait) + __popcll(__ballot(cur == REF_IDLE)) >= (ST ? END_MIN : REFILL_MIN);
        }
        if (COUNT && run_trans) {
            unsigned long long tm = __ballot(cur == REF_EMPTY);
            if (tm != 0ull && lane == (uint32_t)__ffsll((long long)tm) - 1u) c_wtrans++;
            if (cur == REF_EMPTY) c_ltrans++;
        }
        if (run_trans && cur == REF_EMPTY) {
            // object.rs:206-219 / mod.rs:50: see k_trace.  With one object the object's winner is the raycast's winner.
            const bool rank_free = fp.n_objs == 1u;
            if ((fl & F_OHAVE) && rank_free) {
                fl = (fl | F_GHAVE) & ~F_OHAVE;
                g_slot = o_slot;
                g_obj = 0u;
            }
            if (fl & F_OHAVE) {
                const ObjRec* ob = fp.objs + obj_i;
                const float4* tp = reinterpret_cast<const float4*>(fp.tris + o_slot);
                float4 q0 = tp[0], q1 = tp[1], q2 = tp[2], q3 = tp[3];
                TriHit h;
                h.u = h.v = h.w = 0.0f; h.point = f3(0, 0, 0); h.key = 0.0f;
                (void)tri_test(q0, q1, q2, q3.z, oo, od, fp.delta, h);   // same inputs as in the leaf loop -> same bits
                F3 pw = xform_point(ob->fwd, h.point);
                float key = norm(pw - ro);
                bool better;
                if (!(fl & F_GHAVE)) better = true;
                else if (is_nan(key)) better = false;
                else if (is_nan(g_key)) better = true;
                else better = key < g_key;
                if (better) {
                    fl |= F_GHAVE; g_key = key; g_slot = o_slot; g_obj = obj_i;
                    float th = sqrtf(h.key) / len_d;
                    if (th == th) t_lim = th * (1.0f + PRUNE_REL);
                }
                fl &= ~F_OHAVE;
            }
            ++obj_i;
            if (obj_i >= fp.n_objs) {
                // raycast done (mod.rs:45-51); raycast_transparent loop (mod.rs:100-111): only .last() is consumed,
                // so each finished raycast with a hit overwrites the ray's result record
                const bool is_shadow = !RAYGEN && (r_meta & META_SHADOW) != 0u;
                bool again = false;
                const bool first_cast = (fl & F_FIRST) != 0u;
                if (fl & F_GHAVE) {
                    const float4 q3 = reinterpret_cast<const float4*>(fp.tris + g_slot)[3];
                    const uint32_t gid = __float_as_uint(q3.x);
                    const int32_t mat = (int32_t)__float_as_uint(q3.y);
                    const bool transp = (__float_as_uint(q3.w) & TRI_FLAG_TRANSPARENT) != 0u;
                    if (is_shadow) {
                        uint32_t level = r_meta & 0xFFu, light = r_meta >> 16;
                        io.occ[((size_t)level * fp.n_lights + light) * bp.cap + r_sample] = mat >= 0 ? mat : -2;
                    } else {
                        float4* dst = reinterpret_cast<float4*>(io.hits + r_out);
                        dst[0] = make_float4(ro.x, ro.y, ro.z, __uint_as_float(g_slot));   // the origin THIS raycast used (march)
                        dst[1] = make_float4(__uint_as_float(g_obj), __uint_as_float(gid), __uint_as_float((uint32_t)mat), q3.w);
                        if (RAYGEN && first_cast) io.nlev[r_out] = NLEV_PENDING;   // k_setup shades it at level 0
                    }
                    if (RAYGEN && first_cast && io.first_obj != nullptr) {
                        io.first_obj[r_sample] = g_obj;
                        io.first_tri[r_sample] = gid - fp.objs[g_obj].tri_base;
                    }
                    if (transp && (fl >> 16) < MARCH_CAP) {
                        ro = ro + scale(rd, 0.05f);   // mod.rs:106
                        fl += 0x10000u;
                        again = true;
                        if (COUNT) c_march++;
                    }
                } else if (first_cast) {
                    if (is_shadow) {
                        uint32_t level = r_meta & 0xFFu, light = r_meta >> 16;
                        io.occ[((size_t)level * fp.n_lights + light) * bp.cap + r_sample] = -1;
                    } else if (RAYGEN) {
                        // mod.rs:119: a camera ray that hits nothing is skybox(ray.direction).  Only the fact is recorded (one
                        // byte, no hit record); the resolve pass evaluates the sky with full wavefronts
                        io.nlev[r_out] = 0;
                    } else {
                        // a reflection ray that hits nothing: k_setup leaves its direction for the resolve pass
                        reinterpret_cast<float4*>(io.hits + r_out)[0] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(0xFFFFFFFFu));
                    }
                    if (RAYGEN && io.first_obj != nullptr) {
                        io.first_obj[r_sample] = 0xFFFFFFFFu;
                        io.first_tri[r_sample] = 0xFFFFFFFFu;
                    }
                }
                fl &= ~(F_FIRST | F_GHAVE);
                if (again) {
                    obj_i = 0;
                    t_lim = INFINITY;
                } else {
                    cur = REF_IDLE;
                }
            }
            if (cur != REF_IDLE) {
                // object.rs:186-195: ray into object space (direction NOT renormalised), bvh::Ray::new
                // normalises its own copy and keeps 1/direction [crate]
                const ObjRec* ob = fp.objs + obj_i;
                oo = xform_point(ob->inv, ro);
                od = xform_vec(ob->inv, rd);
                len_d = norm(od);
                F3 dn = unscale(od, len_d);
                oi = f3(1.0f / dn.x, 1.0f / dn.y, 1.0f / dn.z);
                slack_abs = ob->slack_abs;
                lim_n = t_lim * len_d + slack_abs;
                if (!(lim_n == lim_n)) lim_n = INFINITY;
                // QN: 2^-20 of (largest |origin coordinate| + largest |scene coordinate|, which is slack_abs * 1e4)
                if (QN) e_ray = 0x1p-20f * rmax(rmax(fabsf(oo.x), fabsf(oo.y)), fabsf(oo.z)) + 0x1p-6f * slack_abs;
                fl = (fl & ~(F_OHAVE | F_SKIPBOX)) | ((ob->flags & 1u) ? F_SKIPBOX : 0u);
                cur = ob->root;
                sp = 0;
            }
        }

#include "trace_step.inl"
    }

    if (COUNT) {
        for (int off = 32; off > 0; off >>= 1) {
            c_nodes += __shfl_down(c_nodes, off);
            c_boxes += __shfl_down(c_boxes, off);
            c_tris += __shfl_down(c_tris, off);
            c_march += __shfl_down(c_march, off);
            c_wnode += __shfl_down(c_wnode, off);
            c_wtri += __shfl_down(c_wtri, off);
            c_wouter += __shfl_down(c_wouter, off);
            c_lanes += __shfl_down(c_lanes, off);
            c_wtrans += __shfl_down(c_wtrans, off);
            c_ltrans += __shfl_down(c_ltrans, off);
            c_wuni += __shfl_down(c_wuni, off);
        }
        if (lane == 0) {
            atomicAdd(io.counters + KC_NODES, c_nodes);
            atomicAdd(io.counters + KC_BOXES, c_boxes);
            atomicAdd(io.counters + KC_TRIS, c_tris);
            atomicAdd(io.counters + KC_MARCH, c_march);
            atomicAdd(io.counters + KC_WNODE, c_wnode);
            atomicAdd(io.counters + KC_WTRI, c_wtri);
            atomicAdd(io.counters + KC_WOUTER, c_wouter);
            atomicAdd(io.counters + KC_WLANES, c_lanes);
            atomicAdd(io.counters + KC_WTRANS, c_wtrans);
            atomicAdd(io.counters + KC_LTRANS, c_ltrans);
            atomicAdd(io.counters + KC_WUNI, c_wuni);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Camera-ray cull.  A camera ray that passes no triangle's own box has an empty candidate set in the reference
// (object.rs:191-195) and is skybox(direction) (mod.rs:119) -- most of a frame, where the scene fills little of it.  Per
// FRAME: (1) k_cover_boxes projects the scene's cover boxes (made once per scene by k_cover_expand: the child boxes of one
// level of every object's tree, each the union of the own boxes -- triangle.rs:88-95 -- of the triangles below it)
// onto the image and marks the 8x8-pixel tiles whose pixels have a sample position inside the projection's bounding
// rectangle, widened by a pixel on every side on top of the samples' own spread around the pixel corner (mod.rs:233-240: from
// -1/s to +1 horizontally, to (yi_max + 1)/s vertically); rectangles of many tiles go through k_cover_big.  A ray through an
// unmarked tile misses every leaf box by at least that pixel, four orders of magnitude beyond the rounding of the slab tests
// it would have failed.  Boxes wholly behind the camera plane cannot be met by a camera ray (all of which point forward,
// camera.rs:32); a box AT the plane, or a bound that is not finite, has no bounded projection: the frame then counts as
// covered everywhere (flags[0]).  Per BATCH: (2) k_cover_groups lists the batch's 64-sample groups that own a pixel of a marked tile; the
// camera-ray launch takes its work items from that list, the samples of the other groups are recorded as misses (nlev 0)
// without a ray.  TRR_CAMCULL=0 switches it off; scenes with an object of at most four triangles (whose root is a leaf, and
// which bvh::traverse returns without a box test when it is one triangle) never use it.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t COVER_SMALL = 16u;   // tiles a thread marks itself

__global__ void __launch_bounds__(256) k_cover_expand(const Node4* nodes, const uint32_t* node_begin, uint32_t n_objs, const uint32_t* frontier_in,
                                                      uint32_t n_in, uint32_t* frontier_out, uint32_t* n_out, CoverBox* boxes, uint32_t* n_boxes,
                                                      uint32_t box_cap, uint32_t last) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if ((i >> 2) >= n_in) return;
    const uint32_t node = frontier_in[i >> 2], k = i & 3u;
    const Node4* nd = nodes + node;
    const uint32_t ref = nd->child[k];
    if (ref == REF_EMPTY) return;
    if (!last && !(ref & REF_LEAF)) {
        frontier_out[atomicAdd(n_out, 1u)] = ref;   // at most 4 * n_in entries: the host sizes the buffer for that
        return;
    }
    uint32_t a = 0, b = n_objs;   // the node's object: node_begin[a] <= node < node_begin[b]
    while (b - a > 1u) {
        const uint32_t m = (a + b) >> 1;
        if (node_begin[m] <= node) a = m; else b = m;
    }
    const uint32_t slot = atomicAdd(n_boxes, 1u);
    if (slot >= box_cap) return;   // (the host stops expanding before this can happen and checks the count)
    CoverBox cb;
    cb.lo[0] = nd->lox[k]; cb.lo[1] = nd->loy[k]; cb.lo[2] = nd->loz[k];
    cb.hi[0] = nd->hix[k]; cb.hi[1] = nd->hiy[k]; cb.hi[2] = nd->hiz[k];
    cb.obj = a; cb.pad = 0;
    boxes[slot] = cb;
}

void launch_cover_expand(hipStream_t st, const Node4* nodes, const uint32_t* node_begin, uint32_t n_objs, const uint32_t* frontier_in,
                         uint32_t n_in, uint32_t* frontier_out, uint32_t* n_out, CoverBox* boxes, uint32_t* n_boxes, uint32_t box_cap, bool last) {
    if (n_in == 0) return;
    hipLaunchKernelGGL(k_cover_expand, dim3((n_in * 4u + 255u) / 256u), dim3(256), 0, st, nodes, node_begin, n_objs, frontier_in, n_in,
                       frontier_out, n_out, boxes, n_boxes, box_cap, last ? 1u : 0u);
}

__global__ void __launch_bounds__(256) k_cover_boxes(FrameParams fp, CoverIO io) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= io.n_boxes) return;
    const CoverBox cbx = io.boxes[i];
    const float lo[3] = {cbx.lo[0], cbx.lo[1], cbx.lo[2]}, hi[3] = {cbx.hi[0], cbx.hi[1], cbx.hi[2]};
    bool finite = true;
    for (int a = 0; a < 3; ++a) finite = finite && fabsf(lo[a]) < INFINITY && fabsf(hi[a]) < INFINITY;   // false for NaN too
    if (!finite) { atomicOr(io.flags, 1u); return; }
    const float* fwd = fp.objs[cbx.obj].fwd;
    const F3 cx = f3(fp.cam_x[0], fp.cam_x[1], fp.cam_x[2]), cy = f3(fp.cam_y[0], fp.cam_y[1], fp.cam_y[2]),
             cz = f3(fp.cam_z[0], fp.cam_z[1], fp.cam_z[2]), cp = f3(fp.cam_pos[0], fp.cam_pos[1], fp.cam_pos[2]);
    float A[8], B[8], C[8];
    float zmin = INFINITY, zmax = -INFINITY, size = 0.0f;
    for (int c = 0; c < 8; ++c) {
        const F3 v = xform_point(fwd, f3((c & 1) ? hi[0] : lo[0], (c & 2) ? hi[1] : lo[1], (c & 4) ? hi[2] : lo[2])) - cp;
        A[c] = dot(v, cx); B[c] = dot(v, cy); C[c] = dot(v, cz);   // camera space: x right, y up, z BACKWARD (in front: C < 0)
        zmin = fminf(zmin, C[c]); zmax = fmaxf(zmax, C[c]);
        size = fmaxf(size, fmaxf(fmaxf(fabsf(A[c]), fabsf(B[c])), fabsf(C[c])));
    }
    if (!(size < INFINITY)) { atomicOr(io.flags, 1u); return; }
    const float eps = 1.0e-5f * size + 1.0e-30f;
    if (zmin > eps) return;                                  // wholly behind the camera plane: no forward ray meets it
    if (!(zmax < -eps)) { atomicOr(io.flags, 1u); return; }  // at the plane: no bounded projection
    // ray through (X, Y) (mod.rs:251-252): d_cam = (X, -Y, focal), so a point (A, B, C) lies on the ray X = focal A / C, Y = -focal B / C
    const float w = (float)fp.width, h = (float)fp.height, aspect = w / h;
    float px0 = INFINITY, px1 = -INFINITY, py0 = INFINITY, py1 = -INFINITY;
    for (int c = 0; c < 8; ++c) {
        const float X = fp.cam_focal * A[c] / C[c], Y = -fp.cam_focal * B[c] / C[c];
        const float px = (X / aspect + 1.0f) * 0.5f * w, py = (Y + 1.0f) * 0.5f * h;   // continuous pixel coordinates of the sample
        px0 = fminf(px0, px); px1 = fmaxf(px1, px); py0 = fminf(py0, py); py1 = fmaxf(py1, py);
    }
    if (!(px0 <= px1) || !(py0 <= py1)) { atomicOr(io.flags, 1u); return; }   // NaN: cannot happen after the checks above
    // pixels with a sample position inside the rectangle, the rectangle widened by the pads
    const float fx0 = floorf(fmaxf(px0 - io.pad_x_lo, -1.0f)), fx1 = ceilf(fminf(px1 + io.pad_x_hi, w));
    const float fy0 = floorf(fmaxf(py0 - io.pad_y_lo, -1.0f)), fy1 = ceilf(fminf(py1 + io.pad_y_hi, h));
    if (fx1 < 0.0f || fy1 < 0.0f || fx0 > w - 1.0f || fy0 > h - 1.0f) return;   // off the image
    const uint32_t x0 = (uint32_t)fmaxf(fx0, 0.0f), x1 = (uint32_t)fminf(fx1, w - 1.0f), y0 = (uint32_t)fmaxf(fy0, 0.0f), y1 = (uint32_t)fminf(fy1, h - 1.0f);
    const uint32_t tx0 = x0 >> 3, tx1 = x1 >> 3, ty0 = y0 >> 3, ty1 = y1 >> 3;
    if ((tx1 - tx0 + 1u) * (ty1 - ty0 + 1u) <= COVER_SMALL) {
        for (uint32_t ty = ty0; ty <= ty1; ++ty)
            for (uint32_t tx = tx0; tx <= tx1; ++tx) io.mask[(size_t)ty * io.tiles_x + tx] = 1;
    } else {
        const uint32_t slot = atomicAdd(io.n_big, 1u);
        if (slot < io.big_cap) io.big[slot] = make_uint4(tx0, ty0, tx1, ty1);
        else atomicOr(io.flags, 1u);
    }
}

__global__ void __launch_bounds__(256) k_cover_big(CoverIO io) {
    const uint32_t n = min(*io.n_big, io.big_cap);
    for (uint32_t e = blockIdx.x; e < n; e += gridDim.x) {
        const uint4 r = io.big[e];
        const uint32_t bw = r.z - r.x + 1u, cnt = bw * (r.w - r.y + 1u);
        for (uint32_t t = threadIdx.x; t < cnt; t += 256u) io.mask[(size_t)(r.y + t / bw) * io.tiles_x + r.x + t % bw] = 1;
    }
}

__global__ void __launch_bounds__(256) k_cover_groups(FrameParams fp, BatchParams bp, CoverIO io) {
    __shared__ uint32_t s_cnt[4], s_base;
    const uint32_t g = blockIdx.x * 256u + threadIdx.x;
    const uint32_t n_groups = (bp.n_samples + 63u) / 64u;
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    bool active = false;
    uint32_t skipped = 0;
    if (g < n_groups) {
        const uint32_t s0 = g * 64u, s1 = min(s0 + 63u, bp.n_samples - 1u);
        active = *io.flags != 0u;
        const uint32_t p0 = s0 / fp.spp, p1 = s1 / fp.spp;
        for (uint32_t p = p0; p <= p1 && !active; ++p) {
            const PixelPos pp = pixel_of(fp, bp, p);
            active = io.mask[(size_t)(pp.y >> 3) * io.tiles_x + (pp.x >> 3)] != 0;
        }
        if (!active) {
            // its samples are camera rays that meet nothing: nlev 0 (the buffer's size is a multiple of 64)
            uint4* z = reinterpret_cast<uint4*>(io.nlev + (size_t)s0);
            z[0] = z[1] = z[2] = z[3] = make_uint4(0, 0, 0, 0);
            skipped = s1 - s0 + 1u;
        }
    }
    // the list keeps the groups' order inside a block; one atomic per block
    const unsigned long long am = __ballot(active);
    if (lane == 0) s_cnt[wv] = (uint32_t)__popcll(am);
    for (int off = 32; off > 0; off >>= 1) skipped += __shfl_down(skipped, off);
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t t = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        s_base = t ? atomicAdd(io.n_groups, t) : 0u;
    }
    __syncthreads();
    if (active) {
        uint32_t base = s_base;
        for (uint32_t k = 0; k < wv; ++k) base += s_cnt[k];
        io.groups[base + (uint32_t)__popcll(am & ((1ull << lane) - 1ull))] = g;
    }
    if (lane == 0 && skipped != 0u) atomicAdd(io.counters + KC_PRIMARY_CULLED, (unsigned long long)skipped);
}

// once per frame: the tile mask (its memory cleared by the caller)
void launch_cover_mask(hipStream_t st, const FrameParams& fp, const CoverIO& io) {
    if (io.n_boxes == 0u) return;
    hipLaunchKernelGGL(k_cover_boxes, dim3((io.n_boxes + 255u) / 256u), dim3(256), 0, st, fp, io);
    hipLaunchKernelGGL(k_cover_big, dim3(256), dim3(256), 0, st, io);
}
// once per batch: the list of its groups that are traced
void launch_cover_groups(hipStream_t st, const FrameParams& fp, const BatchParams& bp, const CoverIO& io) {
    const uint32_t n_groups = (bp.n_samples + 63u) / 64u;
    hipLaunchKernelGGL(k_cover_groups, dim3((n_groups + 255u) / 256u), dim3(256), 0, st, fp, bp, io);
}

// k_setup: one chain ray's closest hit -> shaded-hit record + secondary rays (mod.rs:116-153, 199-206; object.rs:198-219)
// with full wavefronts.  Queue slots are claimed with ONE atomic per block, round and queue: the wavefronts of a block put
// their counts into LDS, thread 0 claims the block's range, every wavefront takes its part of it.  Same-address atomics
// serialise at about 10 ns each on this device, and one per wavefront and round (55 k per queue in a level-0 launch of the
// 4K x 16 spp frame) was most of this kernel's time.  The barriers are reached by every wavefront in every round (the
// round count is uniform over the grid); the LDS cells alternate between rounds, so a wavefront that runs ahead cannot
// overwrite what a slower one still reads.
#ifndef TRR_SETUP_THREADS
#define TRR_SETUP_THREADS 128   // same box, lone 4K x 16 spp frame: one atomic per wavefront 27.75 ms | blocks of 128: 26.7 | 256: 27.15 | 512: 28.8 | 1024: 29.2 (profiles/r03_setup_block_claims_ab.txt)
#endif
constexpr int SETUP_THREADS = TRR_SETUP_THREADS;
template <bool LEVEL0>
__global__ void __launch_bounds__(SETUP_THREADS) k_setup(FrameParams fp, BatchParams bp, SetupIO io, uint32_t lvl) {
    constexpr uint32_t NW = SETUP_THREADS / 64;
    __shared__ uint32_t s_cnt[2][2][NW];    // [round parity][shadow | chain][wavefront]
    __shared__ uint32_t s_base[2][2];       // [round parity][shadow | chain]
    const uint32_t wv = threadIdx.x >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t total = LEVEL0 ? bp.n_samples : *io.n_chain_in;
    const uint32_t L = fp.n_lights;
    const size_t cap = bp.cap;
    // whole wavefronts iterate together (the ballots below need every lane present)
    const uint32_t stride = gridDim.x * blockDim.x;
    const uint32_t rounds = (total + stride - 1) / stride;
    uint32_t n_sh = 0, n_ch = 0, n_cull = 0;   // wave-uniform: rays queued by this wavefront (n_sh includes the n_cull culled ones)
    for (uint32_t it = 0; it < rounds; ++it) {
        uint32_t i = it * stride + blockIdx.x * blockDim.x + threadIdx.x;
        bool valid = i < total;
        bool hit = false, spawn = false;
        F3 P = f3(0, 0, 0), N = f3(0, 0, 0), D = f3(0, 0, 0);
        uint32_t sample = 0;
        int32_t hit_mat = -1;
        bool hit_reflective = false;
        if (LEVEL0 && valid && io.nlev[i] != NLEV_PENDING) valid = false;   // a camera ray that missed: nothing to set up
        if (valid) {
            if (LEVEL0) {
                sample = i;
                uint32_t pix = i / fp.spp, si = i - pix * fp.spp;
                PixelPos pp = pixel_of(fp, bp, pix);
                D = primary_dir(fp, pp.x, pp.y, si);
            } else {
                const float4* q = io.chain_in + (size_t)i * 2u;
                float4 a = q[0], b = q[1];
                D = f3(a.w, b.x, b.y);
                sample = __float_as_uint(b.z);
            }
            const float4* hp = reinterpret_cast<const float4*>(io.hits + i);
            float4 h0 = hp[0];
            uint32_t slot = __float_as_uint(h0.w);
            if (slot == 0xFFFFFFFFu || fp.max_depth == 0u) {
                // mod.rs:119: no hit -> skybox(ray.direction) ends the chain; mod.rs:125-127: with max_depth == 0
                // even a hit is shaded as skybox(ray.direction).  The resolve pass looks the sky up.
                if (LEVEL0) io.nlev[sample] = 0;
                else if (fp.sky != nullptr) io.term[sample] = make_float4(D.x, D.y, D.z, 0.0f);
            } else {
                hit = true;
                float4 h1 = hp[1];
                const ObjRec* ob = fp.objs + __float_as_uint(h1.x);
                const float4* tp = reinterpret_cast<const float4*>(fp.tris + slot);
                const float4* ipp = reinterpret_cast<const float4*>(fp.interp + __float_as_uint(h1.y));
                float4 q0 = tp[0], q1 = tp[1], q2 = tp[2];
                const float len = reinterpret_cast<const float*>(tp)[14];
                float4 i0 = ipp[0], i1 = ipp[1], i2 = ipp[2], i3 = ipp[3];
                F3 a = f3(q0.x, q0.y, q0.z), b = f3(q0.w, q1.x, q1.y), c = f3(q1.z, q1.w, q2.x);
                // the winner's barycentrics (triangle.rs:84): the trace kernel kept only the slot and the origin its
                // last raycast used; object-space ray and exact test are the ones of the trace kernel (object.rs:186-189)
                TriHit th;
                th.u = th.v = th.w = 0.0f; th.point = f3(0, 0, 0); th.key = 0.0f;
                (void)tri_test(q0, q1, q2, len, xform_point(ob->inv, f3(h0.x, h0.y, h0.z)), xform_vec(ob->inv, D), fp.delta, th);
                const float u = th.u, v = th.v, w = th.w;
                F3 na = f3(i0.x, i0.y, i0.z), nb = f3(i0.w, i1.x, i1.y), nc = f3(i1.z, i1.w, i2.x);
                // object.rs:201-203
                F3 point = (scale(a, u) + scale(b, v)) + scale(c, w);
                F3 normal = (scale(na, u) + scale(nb, v)) + scale(nc, w);
                float tu = (i2.y * u + i2.w * v) + i3.y * w;
                float tv = (i2.z * u + i3.x * v) + i3.z * w;
                // object.rs:209-210 (forward matrix for the normal too)
                P = xform_point(ob->fwd, point);
                N = xform_vec(ob->fwd, normal);
                const bool reflective = (__float_as_uint(h1.w) & TRI_FLAG_REFLECTIVE) != 0u;   // material.rs:47-49
                hit_mat = __float_as_int(h1.z);
                hit_reflective = reflective;
                float4* lv = io.level + ((size_t)lvl * cap + sample) * LEVEL_F4;
                lv[0] = make_float4(P.x, P.y, P.z, N.x);
                lv[1] = make_float4(N.y, N.z, D.x, D.y);
                lv[2] = make_float4(D.z, tu, tv, h1.z);
                io.nlev[sample] = (uint8_t)(lvl + 1u);
                if (reflective) {
                    if (lvl + 1 < fp.max_depth) spawn = true;
                    else if (fp.sky != nullptr) {
                        // mod.rs:125-127: the nested shade() is at max_depth -> skybox(reflection direction)
                        F3 r2 = reflect(D, N);
                        io.term[sample] = make_float4(r2.x, r2.y, r2.z, 0.0f);
                    }
                }
            }
        }
        // ---- shadow rays: mod.rs:149-153, one per light, grouped per light inside the wavefront's chunk ----
        const unsigned long long hm = __ballot(hit);
        const unsigned long long sm = __ballot(spawn);
        const uint32_t nh = (uint32_t)__popcll(hm), ns = (uint32_t)__popcll(sm);
        const bool culling = io.cull != 0u && L <= 64u;
        unsigned long long keep = 0ull;   // per lane: bit l = light l's shadow ray is traced
        uint32_t total_rays = nh * L;
        if (hm != 0ull && L > 0 && culling) {
            // Shadow-ray cull.  A light whose diffuse AND specular intensities (mod.rs:176-177, 184-195) are exactly 0.0
            // adds exactly nothing to the colour whatever its shadow ray finds (mod.rs:171-173 skips it, or :178 / :196
            // add a colour times 0.0), unless the material is reflective (the `continue` of :171-173 would also skip the
            // reflection term, :207) -- so that ray is not traced: the pair's occluder code says OCC_CULLED and the
            // resolve pass skips the light.  The two intensities are evaluated here with the resolve pass's own
            // operations; 0.0 * li is 0.0 only for a finite li, NaN otherwise (then the ray is traced).
            if (hit) {
                bool specular = false;
                float ns_exp = 1.0f;
                if (hit_mat >= 0) {
                    const MatRec* m = fp.mats + hit_mat;
                    if (m->flags & 4u) ns_exp = m->ns;
                    specular = (m->illum == 2);
                }
                const bool reflective = hit_reflective;
                const F3 rv = reflect(neg(D), N);
                for (uint32_t l = 0; l < L; ++l) {
                    const LightRec* lt = fp.lights + l;
                    const F3 lp = f3(lt->pos[0], lt->pos[1], lt->pos[2]);
                    const F3 ld = normalize(lp - P);
                    const float li = lt->intensity / norm_squared(lp - P);
                    const float di = rmax(dot(ld, N), 0.0f) * li;
                    bool dark = !reflective && di == 0.0f;
                    if (dark && specular) {
                        const float sint = powf(rmax(dot(ld, rv), 0.0f), ns_exp) * li;
                        dark = sint == 0.0f;
                    }
                    if (dark) io.occ[((size_t)lvl * L + l) * cap + sample] = OCC_CULLED;
                    else keep |= 1ull << l;
                }
            }
            total_rays = 0;
            for (uint32_t l = 0; l < L; ++l) total_rays += (uint32_t)__popcll(__ballot((keep >> l) & 1ull));
        }
        n_sh += nh * L;   // the statistics count the algorithm's rays, traced or culled
        n_cull += nh * L - total_rays;
        n_ch += ns;
        // ---- the block's queue ranges: one atomic per queue for all its wavefronts -------------------------------------
        const uint32_t par = it & 1u;
        if (lane == 0) { s_cnt[par][0][wv] = total_rays; s_cnt[par][1][wv] = ns; }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t t_sh = 0, t_ch = 0;
            for (uint32_t k = 0; k < NW; ++k) { t_sh += s_cnt[par][0][k]; t_ch += s_cnt[par][1][k]; }
            s_base[par][0] = t_sh != 0u ? atomicAdd(io.n_shadow_out, t_sh) : 0u;
            s_base[par][1] = t_ch != 0u ? atomicAdd(io.n_chain_out, t_ch) : 0u;
        }
        __syncthreads();
        uint32_t base = s_base[par][0], base_ch = s_base[par][1];
        for (uint32_t k = 0; k < NW; ++k)
            if (k < wv) { base += s_cnt[par][0][k]; base_ch += s_cnt[par][1][k]; }
        if (hm != 0ull && L > 0 && culling) {
            for (uint32_t l = 0; l < L && total_rays != 0u; ++l) {
                const bool mine = ((keep >> l) & 1ull) != 0ull;
                const unsigned long long km = __ballot(mine);
                const uint32_t rank = (uint32_t)__popcll(km & ((1ull << lane) - 1ull));
                if (mine) {
                    if (io.shadow_compact != 0u) {
                        io.shadow_out[base + rank] = make_float4(P.x, P.y, P.z, __uint_as_float(sample | (l << COMPACT_SAMPLE_BITS)));
                    } else {
                        const LightRec* lt = fp.lights + l;
                        F3 ld = normalize(f3(lt->pos[0], lt->pos[1], lt->pos[2]) - P);   // mod.rs:149
                        F3 o = P + scale(ld, fp.delta);                                  // mod.rs:151
                        float4* e = io.shadow_out + (size_t)(base + rank) * 2u;
                        e[0] = make_float4(o.x, o.y, o.z, ld.x);
                        e[1] = make_float4(ld.y, ld.z, __uint_as_float(sample), __uint_as_float(lvl | META_SHADOW | (l << 16)));
                    }
                }
                base += (uint32_t)__popcll(km);
            }
        } else if (hm != 0ull && L > 0) {
            uint32_t rank = (uint32_t)__popcll(hm & ((1ull << lane) - 1ull));
            if (hit && io.shadow_compact != 0u) {
                // compact entries: the shaded point and (sample, light); the trace kernel rebuilds the ray
                for (uint32_t l = 0; l < L; ++l)
                    io.shadow_out[base + l * nh + rank] = make_float4(P.x, P.y, P.z, __uint_as_float(sample | (l << COMPACT_SAMPLE_BITS)));
            } else if (hit) {
                for (uint32_t l = 0; l < L; ++l) {
                    const LightRec* lt = fp.lights + l;
                    F3 ld = normalize(f3(lt->pos[0], lt->pos[1], lt->pos[2]) - P);   // mod.rs:149
                    F3 o = P + scale(ld, fp.delta);                                  // mod.rs:151
                    float4* e = io.shadow_out + (size_t)(base + l * nh + rank) * 2u;
                    e[0] = make_float4(o.x, o.y, o.z, ld.x);
                    e[1] = make_float4(ld.y, ld.z, __uint_as_float(sample), __uint_as_float(lvl | META_SHADOW | (l << 16)));
                }
            }
        }
        // ---- reflection ray: mod.rs:203-206 (traced once per hit; the value is light-independent) ----------
        if (spawn) {
            const uint32_t rank = (uint32_t)__popcll(sm & ((1ull << lane) - 1ull));
            F3 o = P + scale(N, fp.delta), d = reflect(D, N);
            float4* e = io.chain_out + (size_t)(base_ch + rank) * 2u;
            e[0] = make_float4(o.x, o.y, o.z, d.x);
            e[1] = make_float4(d.y, d.z, __uint_as_float(sample), __uint_as_float(lvl + 1u));
        }
    }
    if (lane == 0 && (n_sh | n_ch) != 0u) {   // the ray statistics of trr_stats
        atomicAdd(io.counters + KC_SHADOW_RAYS, (unsigned long long)n_sh);
        atomicAdd(io.counters + KC_CHAIN_RAYS, (unsigned long long)n_ch);
        if (n_cull != 0u) atomicAdd(io.counters + KC_SHADOW_CULLED, (unsigned long long)n_cull);
    }
}

// ------------------------------------------------------------------------------------------------
// k_resolve: deepest-first evaluation of shade_impl (mod.rs:124-212) for every sample, mean over the samples
// of a pixel (mod.rs:248-249), quantise (render.rs:134-140).
// ------------------------------------------------------------------------------------------------

DEV uint8_t quantize(float c) {   // render.rs:136-138: (c * 255.0) as u8
    float v = c * 255.0f;
    if (!(v > 0.0f)) return 0;
    if (v >= 255.0f) return 255;
    return (uint8_t)v;
}

// Pass 1, one thread per SAMPLE.  The trace / set-up kernels left, per sample: nlev = the number of shaded levels (0 = the
// camera ray missed), a 48-byte record per shaded level (P, N, D, texture u v, material), the occluder code of every (level,
// light) and -- where a reflective chain ends on an image sky (mod.rs:119, 125-127) -- the direction of that look-up in
// `term`.  Levels are evaluated deepest first; the sample's colour goes to the `color` planes.
DEV F3 resolve_sample(const FrameParams& fp, const BatchParams& bp, const ResolveIO& io, uint32_t s) {
    const uint32_t L = fp.n_lights;
    const size_t cap = bp.cap;
    {
        const int n = io.nlev[s];
        F3 R = f3(fp.sky_color[0], fp.sky_color[1], fp.sky_color[2]);
        if (n == 0) {
            // mod.rs:119: the camera ray missed -> skybox(ray.direction), the direction rebuilt from the sample index
            if (fp.sky != nullptr) {
                const uint32_t pix = s / fp.spp, si = s - pix * fp.spp;
                const PixelPos pp = pixel_of(fp, bp, pix);
                R = skybox(fp, primary_dir(fp, pp.x, pp.y, si));
            }
        } else {
            bool need_term = false;
            int32_t dm = __float_as_int(io.level[((size_t)(n - 1) * cap + s) * LEVEL_F4 + 2].w);
            if (dm >= 0) {
                int il = fp.mats[dm].illum;
                need_term = (il == 3 || il == 4);
            }
            if (!need_term) R = f3(0.0f, 0.0f, 0.0f);
            else if (fp.sky != nullptr) {
                const float4 tr = io.term[s];
                R = skybox(fp, f3(tr.x, tr.y, tr.z));
            }
        }
        for (int k = n - 1; k >= 0; --k) {
            const float4* lv = io.level + ((size_t)k * cap + s) * LEVEL_F4;
            const float4 l0 = lv[0], l1 = lv[1], l2 = lv[2];
            F3 P = f3(l0.x, l0.y, l0.z), N = f3(l0.w, l1.x, l1.y), D = f3(l1.z, l1.w, l2.x);
            const float tu = l2.y, tv = l2.z;
            int32_t mat = __float_as_int(l2.w);
            F3 diff = f3(0.9f, 0.9f, 0.9f);   // mod.rs:129-134
            F3 spec = f3(0.9f, 0.9f, 0.9f);   // mod.rs:136-139
            bool specular = false, reflective = false;
            float ns = 1.0f;
            if (mat >= 0) {
                const MatRec* m = fp.mats + mat;
                if (m->tex >= 0) diff = texture_lookup(fp, m->tex, tu, tv);
                else if (m->flags & 1u) diff = f3(m->kd[0], m->kd[1], m->kd[2]);
                if (m->flags & 2u) spec = f3(m->ks[0], m->ks[1], m->ks[2]);
                if (m->flags & 4u) ns = m->ns;
                specular = (m->illum == 2);
                reflective = (m->illum == 3 || m->illum == 4);
            }
            // mod.rs:141-146
            F3 c = scale(cmul(f3(fp.ambient[0], fp.ambient[1], fp.ambient[2]), diff), fp.ambient_intensity);
            for (uint32_t l = 0; l < L; ++l) {   // mod.rs:148 (order matters: c is read and written)
                const LightRec* lt = fp.lights + l;
                F3 lp = f3(lt->pos[0], lt->pos[1], lt->pos[2]);
                F3 ld = normalize(lp - P);   // mod.rs:149
                int32_t oc = io.occ[((size_t)k * L + l) * cap + s];
                if (oc == OCC_CULLED) continue;   // k_setup found both intensities of this light exactly 0.0: it adds nothing
                F3 trans;
                if (oc == -1) trans = f3(1.0f, 1.0f, 1.0f);   // mod.rs:160
                else {
                    F3 okd = f3(1.0f, 1.0f, 1.0f);
                    float odis = 1.0f;
                    if (oc >= 0) {
                        const MatRec* om = fp.mats + oc;
                        if (om->flags & 1u) okd = f3(om->kd[0], om->kd[1], om->kd[2]);
                        if (om->flags & 8u) odis = om->dissolve;
                    }
                    trans = scale(cmul(c, okd), odis);   // mod.rs:162-166
                }
                trans = cmul(trans, f3(lt->color[0], lt->color[1], lt->color[2]));   // mod.rs:169
                if (norm(trans) < 0.01f) continue;                                   // mod.rs:171-173
                float li = lt->intensity / norm_squared(lp - P);                     // mod.rs:176
                float di = rmax(dot(ld, N), 0.0f) * li;                              // mod.rs:177
                c = c + scale(cmul(diff, trans), di);                                // mod.rs:178
                if (specular) {                                                      // mod.rs:181-196
                    float sint = powf(rmax(dot(ld, reflect(neg(D), N)), 0.0f), ns) * li;
                    c = c + scale(cmul(spec, trans), sint);
                }
                if (reflective) c = c + cmul(c, R);                                  // mod.rs:207
            }
            R = c;
        }
        return R;
    }
}

__global__ void __launch_bounds__(RESOLVE_THREADS) k_resolve_samples(FrameParams fp, BatchParams bp, ResolveIO io) {
    const size_t cap = bp.cap;
    for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < bp.n_samples; s += gridDim.x * blockDim.x) {
        const F3 R = resolve_sample(fp, bp, io, s);
        io.color[s] = R.x;
        io.color[cap + s] = R.y;
        io.color[2 * cap + s] = R.z;
    }
}

// Both passes in one, for 1 or a power of two <= 64 samples per pixel (the reference's default is 16): one lane per SAMPLE
// evaluates it, then the spp consecutive lanes of a pixel add their values in ascending sample order -- the order of the
// sequential sum (mod.rs:248-249) -- and the first lane of the group writes the pixel.  The samples' colours never go to
// memory (12 B written + 12 B read per sample saved, and a launch).  Whole wavefronts iterate together: the shuffles need
// every lane present.
__global__ void __launch_bounds__(RESOLVE_THREADS) k_resolve_fused(FrameParams fp, BatchParams bp, ResolveIO io) {
    const uint32_t spp = fp.spp;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t base = lane & ~(spp - 1u);
    const uint32_t stride = gridDim.x * blockDim.x;
    const uint32_t rounds = (bp.n_samples + stride - 1) / stride;
    for (uint32_t it = 0; it < rounds; ++it) {
        const uint32_t s = it * stride + blockIdx.x * blockDim.x + threadIdx.x;
        const bool valid = s < bp.n_samples;   // n_samples is a multiple of spp: a group is valid or invalid as a whole
        F3 v = f3(0.0f, 0.0f, 0.0f);
        if (valid) v = resolve_sample(fp, bp, io, s);
        F3 sum = f3(0.0f, 0.0f, 0.0f);
        for (uint32_t j = 0; j < spp; ++j) {
            F3 c = f3(__shfl(v.x, (int)(base + j)), __shfl(v.y, (int)(base + j)), __shfl(v.z, (int)(base + j)));
            sum = sum + c;
        }
        if (valid && lane == base) {
            const F3 color = fp.aa ? unscale(sum, (float)spp) : v;   // mod.rs:248-249 (spp > 1 implies aa)
            const PixelPos pp = pixel_of(fp, bp, s / spp);
            uint8_t* o = io.strip + (size_t)pp.strip_off * 3;
            o[0] = quantize(color.x);
            o[1] = quantize(color.y);
            o[2] = quantize(color.z);
        }
    }
}

// Pass 2, one thread per PIXEL: sum of the pixel's samples in ascending order, mean (mod.rs:248-249), quantise
// (render.rs:134-140), store into the shard's strip.
__global__ void __launch_bounds__(RESOLVE_THREADS) k_resolve_pixels(FrameParams fp, BatchParams bp, ResolveIO io) {
    const size_t cap = bp.cap;
    const uint32_t spp = fp.spp;
    if (spp >= 2u && spp <= 64u && (spp & (spp - 1u)) == 0u) {
        // One lane per SAMPLE (coalesced reads of the three planes), spp consecutive lanes per pixel; every lane of a
        // group adds the group's values in ascending sample order -- the order of the sequential sum -- and the first
        // lane writes the pixel.  Whole wavefronts iterate together (the shuffles need every lane present).
        const uint32_t lane = threadIdx.x & 63u;
        const uint32_t base = lane & ~(spp - 1u);
        const uint32_t stride = gridDim.x * blockDim.x;
        const uint32_t rounds = (bp.n_samples + stride - 1) / stride;
        for (uint32_t it = 0; it < rounds; ++it) {
            uint32_t s = it * stride + blockIdx.x * blockDim.x + threadIdx.x;
            bool valid = s < bp.n_samples;   // n_samples is a multiple of spp: a group is valid or invalid as a whole
            F3 v = f3(0.0f, 0.0f, 0.0f);
            if (valid) v = f3(io.color[s], io.color[cap + s], io.color[2 * cap + s]);
            F3 sum = f3(0.0f, 0.0f, 0.0f);
            for (uint32_t j = 0; j < spp; ++j) {
                F3 c = f3(__shfl(v.x, (int)(base + j)), __shfl(v.y, (int)(base + j)), __shfl(v.z, (int)(base + j)));
                sum = sum + c;
            }
            if (valid && lane == base) {
                F3 color = fp.aa ? unscale(sum, (float)spp) : v;   // mod.rs:248-249 (spp > 1 implies aa)
                PixelPos pp = pixel_of(fp, bp, s / spp);
                uint8_t* o = io.strip + (size_t)pp.strip_off * 3;
                o[0] = quantize(color.x);
                o[1] = quantize(color.y);
                o[2] = quantize(color.z);
            }
        }
        return;
    }
    for (uint32_t pix = blockIdx.x * blockDim.x + threadIdx.x; pix < bp.n_pixels; pix += gridDim.x * blockDim.x) {
        F3 sum = f3(0.0f, 0.0f, 0.0f);
        F3 color = f3(0.0f, 0.0f, 0.0f);
        for (uint32_t si = 0; si < spp; ++si) {
            uint32_t s = pix * spp + si;
            color = f3(io.color[s], io.color[cap + s], io.color[2 * cap + s]);
            sum = sum + color;
        }
        if (fp.aa) color = unscale(sum, (float)spp);   // mod.rs:248-249
        PixelPos pp = pixel_of(fp, bp, pix);
        uint8_t* o = io.strip + (size_t)pp.strip_off * 3;
        o[0] = quantize(color.x);
        o[1] = quantize(color.y);
        o[2] = quantize(color.z);
    }
}

// ------------------------------------------------------------------------------------------------
// k_assemble: gathered per-shard strips -> row-major RGB8 frame; remainder strip stays black (render.rs:100-102)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_assemble(uint32_t width, uint32_t height, uint32_t bw, uint32_t bh, uint32_t nbx,
                                                   uint32_t nby, uint32_t n_shards, size_t strip_bytes,
                                                   const uint8_t* strips, uint8_t* frame) {
    size_t npix = (size_t)width * height;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (size_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)(p % width), y = (uint32_t)(p / width);
        uint8_t r = 0, g = 0, b = 0;
        if (x < nbx * bw && y < nby * bh) {
            uint32_t xb = x / bw, yb = y / bh;
            uint32_t blk = yb * nbx + xb;
            uint32_t shard = blk % n_shards, j = blk / n_shards;
            size_t off = (size_t)shard * strip_bytes + ((size_t)j * bw * bh + (size_t)(y - yb * bh) * bw + (x - xb * bw)) * 3;
            r = strips[off]; g = strips[off + 1]; b = strips[off + 2];
        }
        frame[p * 3] = r; frame[p * 3 + 1] = g; frame[p * 3 + 2] = b;
    }
}

// ------------------------------------------------------------------------------------------------
// Node4 -> QNode4 (bvh.h), one thread per node, once per scene.  Every child box is rounded outwards onto the node's own
// 8-bit grid (checked in double precision, in which every quantity here is exact: the grid step is at least 1/254 of the
// node's extent, so org / step < 2^32).  *flag is set when a node cannot be quantised -- a bound that is not finite or
// above 2^20 in magnitude -- and the scene then keeps to its exact nodes.
// ------------------------------------------------------------------------------------------------
#ifdef TRR_WIDE8
__global__ void __launch_bounds__(256) k_quantise_nodes(const Node4* __restrict__ nodes, QNode4* __restrict__ out, uint32_t n,
                                                        uint32_t* __restrict__ flag) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    float lo[8][3], hi[8][3];
    uint32_t ch[8];
    int nk = 0;
    auto take = [&](const Node4& nd, int k, int at) {
        lo[at][0] = nd.lox[k]; lo[at][1] = nd.loy[k]; lo[at][2] = nd.loz[k];
        hi[at][0] = nd.hix[k]; hi[at][1] = nd.hiy[k]; hi[at][2] = nd.hiz[k];
        ch[at] = nd.child[k];
    };
    {
        const Node4 nd = nodes[i];
        for (int k = 0; k < 4; ++k)
            if (nd.child[k] != REF_EMPTY) take(nd, k, nk++);
    }
    for (;;) {   // open the inner child with the largest box whose children still fit
        int pick = -1;
        float pa = -1.0f;
        for (int k = 0; k < nk; ++k) {
            if (ch[k] >= REF_LEAF) continue;
            const Node4& c = nodes[ch[k]];
            int m = 0;
            for (int j = 0; j < 4; ++j) m += c.child[j] != REF_EMPTY ? 1 : 0;
            if (m == 0 || nk - 1 + m > 8) continue;
            const float dx = hi[k][0] - lo[k][0], dy = hi[k][1] - lo[k][1], dz = hi[k][2] - lo[k][2];
            float a = dx * dy + dy * dz + dz * dx;
            if (!(a >= 0.0f)) a = 0.0f;
            if (a > pa) { pa = a; pick = k; }
        }
        if (pick < 0) break;
        const Node4 c = nodes[ch[pick]];
        bool first = true;
        for (int j = 0; j < 4; ++j) {
            if (c.child[j] == REF_EMPTY) continue;
            if (first) { take(c, j, pick); first = false; }
            else take(c, j, nk++);
        }
    }
    QNode4 q;
    for (int k = 0; k < 8; ++k) {
        if (k < nk) {
            q.lox[k] = lo[k][0]; q.loy[k] = lo[k][1]; q.loz[k] = lo[k][2];
            q.hix[k] = hi[k][0]; q.hiy[k] = hi[k][1]; q.hiz[k] = hi[k][2];
            q.child[k] = ch[k];
        } else {
            q.lox[k] = q.loy[k] = q.loz[k] = INFINITY;
            q.hix[k] = q.hiy[k] = q.hiz[k] = INFINITY;
            q.child[k] = REF_EMPTY;
        }
        q.pad[k] = 0;
    }
    out[i] = q;
    (void)flag;
}
#else
__global__ void __launch_bounds__(256) k_quantise_nodes(const Node4* __restrict__ nodes, QNode4* __restrict__ out, uint32_t n,
                                                        uint32_t* __restrict__ flag) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const Node4 nd = nodes[i];
    QNode4 q;
    bool bad = false;
    uint32_t qlo[3] = {0, 0, 0}, qhi[3] = {0, 0, 0};
    float org[3], sc[3];
    for (int a = 0; a < 3; ++a) {
        const float* lo = a == 0 ? nd.lox : a == 1 ? nd.loy : nd.loz;
        const float* hi = a == 0 ? nd.hix : a == 1 ? nd.hiy : nd.hiz;
        float lo_min = INFINITY, hi_max = -INFINITY;
        for (int k = 0; k < 4; ++k) {
            if (nd.child[k] == REF_EMPTY) continue;
            if (!(fabsf(lo[k]) <= 0x1p20f) || !(fabsf(hi[k]) <= 0x1p20f)) bad = true;   // also catches NaN
            lo_min = fminf(lo_min, lo[k]);
            hi_max = fmaxf(hi_max, hi[k]);
        }
        if (!(lo_min <= hi_max)) { lo_min = 0.0f; hi_max = 0.0f; }   // no child (or nothing sane): everything below is moot
        const double ext = (double)hi_max - (double)lo_min;
        int e2 = -100;   // step 2^e2, at least 2^-100
        if (ext > 0.0) {
            int ex;
            (void)frexp(ext / 254.0, &ex);   // ext / 254 = m * 2^ex, m in [0.5, 1)  ->  2^ex >= ext / 254
            e2 = max(ex, -100);
        }
        double step = ldexp(1.0, e2);
        while (ceil(ext / step) > 255.0) step *= 2.0;   // (never taken; guards the rounding of the line above)
        org[a] = lo_min;
        sc[a] = (float)step;
        for (int k = 0; k < 4; ++k) {
            uint32_t l = 255u, h = 0u;   // an unused slot: inverted, and its reference is REF_EMPTY anyway
            if (nd.child[k] != REF_EMPTY && !bad) {
                const double dl = floor(((double)lo[k] - (double)lo_min) / step), dh = ceil(((double)hi[k] - (double)lo_min) / step);
                l = (uint32_t)fmin(fmax(dl, 0.0), 255.0);
                h = (uint32_t)fmin(fmax(dh, 0.0), 255.0);
                if ((double)lo_min + l * step > (double)lo[k] || (double)lo_min + h * step < (double)hi[k]) bad = true;   // cannot happen
            }
            qlo[a] |= l << (8 * k);
            qhi[a] |= h << (8 * k);
        }
    }
    q.org[0] = org[0]; q.org[1] = org[1]; q.org[2] = org[2];
    q.sx = sc[0]; q.sy = sc[1]; q.sz = sc[2];
    q.qlox = qlo[0]; q.qloy = qlo[1]; q.qloz = qlo[2];
    q.qhix = qhi[0]; q.qhiy = qhi[1]; q.qhiz = qhi[2];
    for (int k = 0; k < 4; ++k) q.child[k] = nd.child[k];
    out[i] = q;
    if (bad) atomicOr(flag, 1u);
}

#endif
void launch_quantise_nodes(hipStream_t st, const Node4* nodes, QNode4* out, uint32_t n, uint32_t* flag) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_quantise_nodes, dim3((n + 255) / 256), dim3(256), 0, st, nodes, out, n, flag);
}

// ------------------------------------------------------------------------------------------------
// host-callable launchers (used by render.cpp)
// ------------------------------------------------------------------------------------------------
int trace_blocks_per_cu(bool count) {
    int nb = 0;
    hipError_t e = count ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_trace<true>, TRACE_THREADS, 0)
                         : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_trace<false>, TRACE_THREADS, 0);
    if (e != hipSuccess || nb < 1) nb = 1;
    return std::max(1, nb / BLOCKS_PER_256);   // in 256-thread units
}

// `grid` in 256-thread units
size_t trace_spill_bytes(uint32_t grid) { return (size_t)grid * 256 * STACK_SPILL * sizeof(uint2); }
size_t trace_queue_bytes(uint32_t grid, uint32_t qcap) { return (size_t)grid * 4 * qcap * 32; }

void launch_trace(bool count, uint32_t grid, hipStream_t st, const FrameParams& fp, const BatchParams& bp, const TraceIO& io) {
    dim3 g(grid * BLOCKS_PER_256), b(TRACE_THREADS);
    TraceArgs a;
    a.fp = fp; a.bp = bp; a.io = io;
    if (count) hipLaunchKernelGGL((k_trace<true>), g, b, 0, st, a);
    else hipLaunchKernelGGL((k_trace<false>), g, b, 0, st, a);
}

// the level kernels read the 64-B quantised nodes when the frame has them (fp.qnodes), the exact ones otherwise.
// TRR_STAGE=1 switches the secondary-ray launches to the staged ray starts (k_trace_level<.., ST>): same image, measured
// 8 % slower on the 4K x 16 spp frame (profiles/r03_staged_starts_ab.txt), so it is opt-in, for experiments and its test.
static bool staged_starts() {
    const char* e = std::getenv("TRR_STAGE");
    return e && e[0] == '1';
}
template <bool RAYGEN, bool COUNT, bool QN, bool ST>
static int level_bpc_one() {
    int nb = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_trace_level<RAYGEN, COUNT, QN, ST>, TRACE_THREADS, 0);
    return (e != hipSuccess || nb < 1) ? 1 : std::max(1, nb / BLOCKS_PER_256);   // in 256-thread units
}
int level_blocks_per_cu(bool raygen, bool count) {
    if (raygen) return count ? std::min(level_bpc_one<true, true, false, false>(), level_bpc_one<true, true, true, false>())
                             : std::min(level_bpc_one<true, false, false, false>(), level_bpc_one<true, false, true, false>());
    // asked once per context: the smallest over both ray-start forms, either of which a later launch may use
    const int st = count ? std::min(level_bpc_one<false, true, false, true>(), level_bpc_one<false, true, true, true>())
                         : std::min(level_bpc_one<false, false, false, true>(), level_bpc_one<false, false, true, true>());
    const int pl = count ? std::min(level_bpc_one<false, true, false, false>(), level_bpc_one<false, true, true, false>())
                         : std::min(level_bpc_one<false, false, false, false>(), level_bpc_one<false, false, true, false>());
    return std::min(st, pl);
}

template <bool RAYGEN, bool QN, bool ST>
static void launch_trace_level_t(bool count, dim3 g, dim3 b, hipStream_t st, const FrameParams& fp, const BatchParams& bp, const LevelIO& io) {
    if (count) hipLaunchKernelGGL((k_trace_level<RAYGEN, true, QN, ST>), g, b, 0, st, fp, bp, io);
    else hipLaunchKernelGGL((k_trace_level<RAYGEN, false, QN, ST>), g, b, 0, st, fp, bp, io);
}
void launch_trace_level(bool raygen, bool count, uint32_t grid, hipStream_t st, const FrameParams& fp, const BatchParams& bp,
                        const LevelIO& io) {
    dim3 g(grid * BLOCKS_PER_256), b(TRACE_THREADS);
    const bool qn = fp.qnodes != nullptr;
    if (raygen) {
        if (qn) launch_trace_level_t<true, true, false>(count, g, b, st, fp, bp, io);
        else launch_trace_level_t<true, false, false>(count, g, b, st, fp, bp, io);
    } else if (staged_starts()) {
        if (qn) launch_trace_level_t<false, true, true>(count, g, b, st, fp, bp, io);
        else launch_trace_level_t<false, false, true>(count, g, b, st, fp, bp, io);
    } else {
        if (qn) launch_trace_level_t<false, true, false>(count, g, b, st, fp, bp, io);
        else launch_trace_level_t<false, false, false>(count, g, b, st, fp, bp, io);
    }
}

void launch_setup(bool level0, uint32_t grid, hipStream_t st, const FrameParams& fp, const BatchParams& bp, const SetupIO& io,
                  uint32_t lvl) {
    // `grid` counts 256-thread blocks (the caller's unit for all the streaming kernels)
    dim3 g(std::max(1u, (grid * 256u + SETUP_THREADS - 1) / SETUP_THREADS)), b(SETUP_THREADS);
    if (level0) hipLaunchKernelGGL((k_setup<true>), g, b, 0, st, fp, bp, io, lvl);
    else hipLaunchKernelGGL((k_setup<false>), g, b, 0, st, fp, bp, io, lvl);
}

// TRR_RESOLVE=two_pass keeps the two resolve kernels for every sample count (experiments, tests of that path)
static bool resolve_two_pass() {
    const char* e = std::getenv("TRR_RESOLVE");
    return e && e[0] == 't';
}
void launch_resolve(uint32_t grid_samples, uint32_t grid_pixels, hipStream_t st, const FrameParams& fp, const BatchParams& bp,
                    const ResolveIO& io) {
    const bool pow2 = fp.spp >= 1u && fp.spp <= 64u && (fp.spp & (fp.spp - 1u)) == 0u;
    constexpr uint32_t RB = 256 / RESOLVE_THREADS;   // the grids arrive in 256-thread units
    if (pow2 && !resolve_two_pass()) {
        hipLaunchKernelGGL(k_resolve_fused, dim3(grid_samples * RB), dim3(RESOLVE_THREADS), 0, st, fp, bp, io);
        return;
    }
    hipLaunchKernelGGL(k_resolve_samples, dim3(grid_samples * RB), dim3(RESOLVE_THREADS), 0, st, fp, bp, io);
    const bool per_sample = pow2 && fp.spp >= 2u;   // k_resolve_pixels' coalesced form
    hipLaunchKernelGGL(k_resolve_pixels, dim3((per_sample ? grid_samples : grid_pixels) * RB), dim3(RESOLVE_THREADS), 0, st, fp, bp, io);
}

void launch_assemble(hipStream_t st, uint32_t width, uint32_t height, uint32_t bw, uint32_t bh, uint32_t nbx, uint32_t nby,
                     uint32_t n_shards, size_t strip_bytes, const uint8_t* strips, uint8_t* frame) {
    size_t npix = (size_t)width * height;
    uint32_t grid = (uint32_t)std::min<size_t>((npix + 255) / 256, 4096);
    hipLaunchKernelGGL(k_assemble, dim3(grid), dim3(256), 0, st, width, height, bw, bh, nbx, nby, n_shards, strip_bytes, strips,
                       frame);
}

}   // namespace trr
