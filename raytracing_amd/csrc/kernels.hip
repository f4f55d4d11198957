// kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the Trayracer hot path.
//
// Wavefront (queue-based) organisation, one HIP launch sequence per batch of image blocks, no host
// round trip inside a batch (queue sizes stay in device memory):
//
//   k_trace<RAYGEN>      camera rays generated in-kernel  -> HitRec per sample        (mod.rs:250-255, camera.rs:30-42)
//   for level = 0 .. max_depth-1:
//     k_setup            HitRec -> shaded-hit record; emits L shadow rays + <=1 reflection ray   (mod.rs:124-153, 199-206)
//     k_trace<QUEUE>     shadow rays + next level's reflection rays in ONE launch   (mod.rs:45-51, 96-114; object.rs:184-220)
//   k_resolve_samples    deepest-first evaluation of shade_impl per sample that hit something   (mod.rs:141-211)
//   k_resolve_pixels     mean over a pixel's samples, RGB8                                     (mod.rs:248-249, render.rs:134-140)
//
// Arithmetic is IEEE fp32 with the reference's operation order; build with -ffp-contract=off (Rust never
// fuses a*b+c) and correctly rounded division / sqrt (hipcc default).  Comments cite the reference lines
// (paths relative to /root/reference/src).  "[crate]" marks semantics of un-vendored crates (nalgebra, bvh)
// restated from their public sources.
#include <hip/hip_runtime.h>

#include <algorithm>

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
constexpr int TRACE_THREADS = 256;
constexpr uint8_t NLEV_PENDING = 0xFF;   // nlev value of a camera sample whose hit waits for k_setup (max_depth <= 255 keeps it free)
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

template <bool RAYGEN, bool COUNT>
__global__ void __launch_bounds__(TRACE_THREADS) TRR_OCCUPANCY_ATTR k_trace(FrameParams fp, BatchParams bp, TraceIO io) {
    __shared__ uint2 s_stack[STACK_LDS * TRACE_THREADS];
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    uint2* my_spill = io.spill + ((size_t)blockIdx.x * TRACE_THREADS + tid) * STACK_SPILL;

    uint32_t n_shadow = 0, total;
    if (RAYGEN) total = bp.n_samples;
    else {
        n_shadow = *io.n_shadow;
        total = n_shadow + *io.n_chain;
    }

    // a launch with little (or no) work: wavefronts beyond one per 64 items leave before touching the work counter
    // (5120 same-address atomics cost ~40 us, which is most of an empty launch)
    if ((blockIdx.x * (TRACE_THREADS / 64) + tid / 64) * 64u >= total) return;

    // ---- per-lane state (results go to memory as soon as a raycast ends; only the winner's slot/key stay live) ----
    // Per-lane booleans live in ONE register (bits of `fl`), and "this lane has no ray" is a value of `cur` (REF_IDLE):
    // as separate bools they become SGPR lane masks that hipcc re-merges at every join of the loop (3 scalar
    // instructions per bool and join, ~100 per iteration).
    constexpr uint32_t F_FIRST = 1u, F_GHAVE = 2u, F_OHAVE = 4u, F_SKIPBOX = 8u;
    constexpr uint32_t REF_IDLE = 0xFFFFFFFEu;           // never a leaf reference: slots stay below 2^29 - 1 (api.hip)
    uint32_t fl = 0;
    bool exhausted = false, last_chunk = false;          // wave-uniform
    uint32_t w_next = 0, w_end = 0;                      // wave-uniform: this wavefront's claimed range of work items
    // chunk (guided self-scheduling): a fraction of what is left per wavefront, between one wavefront's worth and CHUNK_MAX
    // items, so that claims are rare while there is plenty of work and fine-grained near the end of the launch
    const uint32_t n_waves = gridDim.x * (TRACE_THREADS / 64);
    uint32_t chunk = max(64u, min(CHUNK_MAX, total / (n_waves * CHUNK_DIV)));
    F3 ro = f3(0, 0, 0), rd = f3(0, 0, 0);      // world ray (origin advances during the march)
    uint32_t r_out = 0, r_sample = 0, r_meta = 0;
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

#ifdef TRR_PAD_VGPRS
    // experiment: TRR_PAD_VGPRS registers kept live through the loop without a single instruction.  The register count
    // matters even below 128 (the grid's four blocks per CU are resident either way): +8 registers +5.6 % frame time,
    // +19 +10.8 % -- the blocks of the OTHER work set's launches share the CU only while registers are left.
    float pad_regs[TRR_PAD_VGPRS];
#pragma unroll
    for (int k = 0; k < TRR_PAD_VGPRS; ++k) pad_regs[k] = (float)(tid + k);
#endif
    for (;;) {
#ifdef TRR_PAD_VGPRS
#pragma unroll
        for (int k = 0; k < TRR_PAD_VGPRS; ++k) asm volatile("" : "+v"(pad_regs[k]));
#endif
        // ---- re-fill idle lanes -------------------------------------------------------------
        unsigned long long idle = __ballot(cur == REF_IDLE);
        if (idle != 0ull && !exhausted) {
            int nidle = __popcll(idle);
            if (nidle >= REFILL_MIN) {
                // work is claimed from the device counter in chunks per wavefront (one same-address atomic per chunk,
                // not per re-fill: 10^6s of atomics on one word serialise in L2) and dealt to idle lanes from the chunk
                if (w_next >= w_end) {
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(io.work_counter, chunk);
                    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                    w_next = min(base, total);
                    w_end = min(base + chunk, total);
                    if (base + chunk >= total) last_chunk = true;
                    chunk = max(64u, min(CHUNK_MAX, (total - w_end) / (n_waves * CHUNK_DIV)));
                }
                const uint32_t take = min((uint32_t)nidle, w_end - w_next);
                uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
                uint32_t idx = w_next + rank;
                w_next += take;
                if (last_chunk && w_next >= w_end) exhausted = true;
                if (cur == REF_IDLE && rank < take) {
                    if (RAYGEN) {
                        uint32_t pix = idx / fp.spp, si = idx - pix * fp.spp;
                        PixelPos pp = pixel_of(fp, bp, pix);
                        ro = f3(fp.cam_pos[0], fp.cam_pos[1], fp.cam_pos[2]);
                        rd = primary_dir(fp, pp.x, pp.y, si);
                        r_out = idx;
                        r_sample = pp.y * fp.width + pp.x;   // frame pixel index, used for the id buffers
                        r_meta = idx;                        // RAYGEN: the sample index (term / nlev planes)
                    } else {
                        const RayRec* q = (idx < n_shadow) ? (io.shadow_q + idx) : (io.chain_q + (idx - n_shadow));
                        float4 a = *reinterpret_cast<const float4*>(q);
                        float4 b = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(q) + 16);
                        ro = f3(a.x, a.y, a.z);
                        rd = f3(a.w, b.x, b.y);
                        r_sample = __float_as_uint(b.z);
                        r_meta = __float_as_uint(b.w);
                        r_out = idx - n_shadow;   // meaningful for chain rays only
                    }
                    fl = F_FIRST;          // no raycast of this ray has finished yet; no hit of any kind
                    obj_i = 0xFFFFFFFFu;   // "before the first object"
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
        }

        // ---- transition: object finished / raycast finished / ray finished ----------------------
#if !TRR_DEFER_END
        if (COUNT) {
            unsigned long long tm = __ballot(cur == REF_EMPTY);
            if (tm != 0ull && lane == (uint32_t)__ffsll((long long)tm) - 1u) c_wtrans++;
            if (cur == REF_EMPTY) c_ltrans++;
        }
#endif
#if TRR_DEFER_END
        // The transition block is long (winner re-evaluation, object-space ray, divisions) and costs a wavefront the same
        // whether 1 or 64 lanes need it.  Lanes that only have to END their ray (last object done) can wait: they would
        // sit idle until the next re-fill anyway.  The block runs when a lane needs it to go on (fresh ray, next object),
        // when the waiting + idle lanes reach the re-fill threshold, or when no lane has traversal work left.
        bool run_trans;
        {
            const bool is_empty = cur == REF_EMPTY;
            const bool can_wait = is_empty && obj_i != 0xFFFFFFFFu && obj_i + 1u >= fp.n_objs;
            const unsigned long long m_empty = __ballot(is_empty);
            const unsigned long long m_wait = __ballot(can_wait);
            const unsigned long long m_work = __ballot(cur < REF_IDLE);
            run_trans = m_empty != m_wait || m_work == 0ull ||
                        __popcll(m_wait) + __popcll(__ballot(cur == REF_IDLE)) >= REFILL_MIN;
        }
        if (COUNT && run_trans) {
            unsigned long long tm = __ballot(cur == REF_EMPTY);
            if (tm != 0ull && lane == (uint32_t)__ffsll((long long)tm) - 1u) c_wtrans++;
            if (cur == REF_EMPTY) c_ltrans++;
        }
        if (run_trans && cur == REF_EMPTY) {
#else
        if (cur == REF_EMPTY) {
#endif
            // object.rs:206-219: best candidate of the object -> world space, then mod.rs:50 ranks objects by
            // |point_world - ray.origin| (first object wins ties: strict <, NaN greatest)
            // With a single object there is nothing to rank: the winner of the object is the winner of the raycast.  Its
            // barycentrics are not needed here either: a shadow ray only reports the occluder's material, and k_setup
            // re-evaluates a chain ray's winner from (slot, ray) with full wavefronts -- the same tri_test on the same
            // inputs, so the same bits -- instead of the few lanes that end their object in this iteration.
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
                    // later objects only need boxes nearer than this hit (ray parameter + slack)
                    float th = sqrtf(h.key) / len_d;
                    if (th == th) t_lim = th * (1.0f + PRUNE_REL);
                }
                fl &= ~F_OHAVE;
            }
            ++obj_i;
            if (obj_i >= fp.n_objs) {
                // raycast done (mod.rs:45-51); raycast_transparent loop (mod.rs:100-111): only .last() is consumed,
                // so each finished raycast with a hit overwrites the ray's result record
                const bool is_shadow = !RAYGEN && (r_meta & 0x100u) != 0u;
                bool again = false;
                const bool first_cast = (fl & F_FIRST) != 0u;
                if (fl & F_GHAVE) {
                    const float4 q3 = reinterpret_cast<const float4*>(fp.tris + g_slot)[3];
                    const uint32_t tri_id = __float_as_uint(q3.x);
                    const int32_t mat = (int32_t)__float_as_uint(q3.y);
                    const bool transp = (__float_as_uint(q3.w) & 1u) != 0u;
                    if (is_shadow) {
                        uint32_t level = r_meta & 0xFFu, light = r_meta >> 16;
                        io.occ[((size_t)level * fp.n_lights + light) * bp.cap + r_sample] = mat >= 0 ? mat : -2;
                    } else {
                        float4* dst = reinterpret_cast<float4*>(io.hits + r_out);
                        dst[0] = make_float4(ro.x, ro.y, ro.z, __uint_as_float(g_slot));   // the origin THIS raycast used (march)
                        dst[1] = make_float4(__uint_as_float(g_obj), __uint_as_float(tri_id), 0.0f, 0.0f);
                        if (RAYGEN && first_cast) io.nlev[r_meta] = NLEV_PENDING;   // k_setup shades it at level 0
                    }
                    if (RAYGEN && first_cast && io.first_obj != nullptr) {
                        io.first_obj[r_sample] = g_obj;
                        io.first_tri[r_sample] = tri_id;
                    }
                    if (transp) {
                        ro = ro + scale(rd, 0.05f);   // mod.rs:106
                        again = true;
                        if (COUNT) c_march++;
                    }
                } else if (first_cast) {
                    if (is_shadow) {
                        uint32_t level = r_meta & 0xFFu, light = r_meta >> 16;
                        io.occ[((size_t)level * fp.n_lights + light) * bp.cap + r_sample] = -1;
                    } else if (RAYGEN) {
                        // mod.rs:119: a camera ray that hits nothing is skybox(ray.direction).  Only the fact is recorded
                        // here (no hit record); k_setup<LEVEL0> evaluates the sky for these samples with full wavefronts
                        // instead of the few lanes that end their ray in this iteration
                        io.nlev[r_meta] = 0;
                    } else {
                        float4* dst = reinterpret_cast<float4*>(io.hits + r_out);
                        dst[0] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(0xFFFFFFFFu));
                        dst[1] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
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
                fl = (fl & ~(F_OHAVE | F_SKIPBOX)) | ((ob->flags & 1u) ? F_SKIPBOX : 0u);
                cur = ob->root;
                sp = 0;
            }
        }

        // ---- one step per iteration, chosen for the WHOLE wavefront: either every lane that stands on an inner node
        // visits it, or every lane that stands on a leaf tests ONE of its triangles.  The kind with more waiting lanes
        // runs (node steps weighted by NODE_BIAS/8); the other lanes keep their state for a later iteration.  A step
        // costs the same whether 1 or 64 lanes need it, so this keeps the SIMD lanes of divergent rays busy.
        const unsigned long long m_inner = __ballot(cur < REF_LEAF);
        const unsigned long long m_leaf = __ballot(cur >= REF_LEAF && cur < REF_IDLE);
        const bool node_step = m_inner != 0ull && (__popcll(m_inner) * NODE_BIAS >= __popcll(m_leaf) * 8);

        if (node_step) {
            if (cur < REF_LEAF) {
                const float4* nd = reinterpret_cast<const float4*>(fp.nodes + cur);
                float4 lox = nd[0], loy = nd[1], loz = nd[2], hix = nd[3], hiy = nd[4], hiz = nd[5];
                uint4 ch = *reinterpret_cast<const uint4*>(nd + 6);
#ifdef TRR_DUMMY_LOAD
                {   // experiment: one more 16-B load per lane from the node's own cache line (is the L1 / address pipe a limit?)
                    float4 extra = nd[7];
                    asm volatile("" ::"v"(extra.x), "v"(extra.y), "v"(extra.z), "v"(extra.w));
                }
#endif
#ifdef TRR_DUMMY_VALU
                {   // experiment: TRR_DUMMY_VALU extra dependent vector instructions per node step (is the kernel VALU-bound?)
                    float dv = lox.x;
#pragma unroll
                    for (int k = 0; k < TRR_DUMMY_VALU; ++k) asm volatile("v_add_f32 %0, %0, %0" : "+v"(dv));
                    asm volatile("" ::"v"(dv));
                }
#endif
                if (COUNT) {
                    c_nodes++; c_boxes += 4;
                    const unsigned long long am = __ballot(true);
                    const uint32_t first_lane = (uint32_t)__ffsll((long long)am) - 1u;
                    if (lane == first_lane) c_wnode++;
                    // node steps in which every participating lane stands on the SAME node
                    const uint32_t cur0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)cur);
                    if (__ballot(cur == cur0) == am && lane == first_lane) c_wuni++;
                }
                // [crate] Ray::intersects_aabb per child, same arithmetic as the per-triangle test below, so a parent
                // box (exact union of its children) can never reject what a triangle's own box accepts.  Written with
                // scalars and selects only (no arrays, no short-circuit branches): keeps the step branch-free.
                // An unused child slot holds the box (+inf .. +inf), which fails the test for every ray; a NaN tmin keeps
                // the child (NaN > x is false) and simply does not take part in the ordering.
#define TRR_BOX(T, R, LX, LY, LZ, HX, HY, HZ, CH)                                                          \
    float T;                                                                                               \
    uint32_t R;                                                                                            \
    {                                                                                                      \
        float ax = ((LX) - oo.x) * oi.x, bx = ((HX) - oo.x) * oi.x;                                        \
        float ay = ((LY) - oo.y) * oi.y, by = ((HY) - oo.y) * oi.y;                                        \
        float az = ((LZ) - oo.z) * oi.z, bz = ((HZ) - oo.z) * oi.z;                                        \
        float tmin = rmax(rmax(rmin(ax, bx), rmin(ay, by)), rmin(az, bz));                                 \
        float tmax = rmin(rmin(rmax(ax, bx), rmax(ay, by)), rmax(az, bz));                                 \
        bool hit = (tmax > rmax(tmin, 0.0f)) & !(tmin > lim_n);                                            \
        T = hit ? tmin : INFINITY;                                                                         \
        R = hit ? (CH) : REF_EMPTY;                                                                        \
    }
                TRR_BOX(t0, r0, lox.x, loy.x, loz.x, hix.x, hiy.x, hiz.x, ch.x)
                TRR_BOX(t1, r1, lox.y, loy.y, loz.y, hix.y, hiy.y, hiz.y, ch.y)
                TRR_BOX(t2, r2, lox.z, loy.z, loz.z, hix.z, hiy.z, hiz.z, ch.z)
                TRR_BOX(t3, r3, lox.w, loy.w, loz.w, hix.w, hiy.w, hiz.w, ch.w)
#undef TRR_BOX
                // sort the four (tmin, ref) pairs ascending: 5-comparator network of selects
#define TRR_CSWAP(TI, RI, TJ, RJ)                                                  \
    {                                                                              \
        bool sw_ = TJ < TI;                                                        \
        float ta_ = sw_ ? TJ : TI, tb_ = sw_ ? TI : TJ;                            \
        uint32_t ra_ = sw_ ? RJ : RI, rb_ = sw_ ? RI : RJ;                         \
        TI = ta_; TJ = tb_; RI = ra_; RJ = rb_;                                    \
    }
                TRR_CSWAP(t0, r0, t1, r1) TRR_CSWAP(t2, r2, t3, r3) TRR_CSWAP(t0, r0, t2, r2)
                TRR_CSWAP(t1, r1, t3, r3) TRR_CSWAP(t1, r1, t2, r2)
#undef TRR_CSWAP
                // far children go on the stack, farthest first.  Fast path: three unconditional LDS stores, the stack
                // pointer advances only past valid entries (a slot above the top may hold garbage).
                if (sp + 3 <= STACK_LDS) {
                    const int i3 = sp, i2 = i3 + (r3 != REF_EMPTY ? 1 : 0), i1 = i2 + (r2 != REF_EMPTY ? 1 : 0);
                    s_stack[i3 * TRACE_THREADS + tid] = make_uint2(r3, __float_as_uint(t3));
                    s_stack[i2 * TRACE_THREADS + tid] = make_uint2(r2, __float_as_uint(t2));
                    s_stack[i1 * TRACE_THREADS + tid] = make_uint2(r1, __float_as_uint(t1));
                    sp = i1 + (r1 != REF_EMPTY ? 1 : 0);
                } else {
                    if (r3 != REF_EMPTY) push(r3, t3);
                    if (r2 != REF_EMPTY) push(r2, t2);
                    if (r1 != REF_EMPTY) push(r1, t1);
                }
                cur = (r0 != REF_EMPTY) ? r0 : pop();
            }
        } else if (m_leaf != 0ull) {
            // ---- leaf: own-box test + exact test of the leaf's next TRR_TRIS_PER_STEP remaining triangles ------------
            if (cur >= REF_LEAF && cur < REF_IDLE) {
                const uint32_t slot = (cur & 0x7FFFFFFFu) >> 2;
                if (COUNT) {
                    if (lane == (uint32_t)__ffsll((long long)__ballot(true)) - 1u) c_wtri++;
                }
                // one triangle: triangle.rs:88-95 box (empty.grow(a).grow(b).grow(c + 1e-4)) + [crate] slab test, then the
                // exact test; candidates are ranked by (key, triangle id), so the order of the tests does not matter
                auto test_one = [&](uint32_t sl, float4 q0, float4 q1, float4 q2, float4 q3) {
                    if (COUNT) c_boxes++;
                    bool cand = true;
                    if (!(fl & F_SKIPBOX)) {
                        F3 a = f3(q0.x, q0.y, q0.z), b = f3(q0.w, q1.x, q1.y), c = f3(q1.z, q1.w, q2.x);
                        F3 c2 = f3(c.x + 0.0001f, c.y + 0.0001f, c.z + 0.0001f);
                        F3 lo = f3(rmin(rmin(a.x, b.x), c2.x), rmin(rmin(a.y, b.y), c2.y), rmin(rmin(a.z, b.z), c2.z));
                        F3 hi = f3(rmax(rmax(a.x, b.x), c2.x), rmax(rmax(a.y, b.y), c2.y), rmax(rmax(a.z, b.z), c2.z));
                        float ax = (lo.x - oo.x) * oi.x, bx = (hi.x - oo.x) * oi.x;
                        float ay = (lo.y - oo.y) * oi.y, by = (hi.y - oo.y) * oi.y;
                        float az = (lo.z - oo.z) * oi.z, bz = (hi.z - oo.z) * oi.z;
                        float tmin = rmax(rmax(rmin(ax, bx), rmin(ay, by)), rmin(az, bz));
                        float tmax = rmin(rmin(rmax(ax, bx), rmax(ay, by)), rmax(az, bz));
                        // the second condition: cannot beat the current best (slack keeps this conservative)
                        cand = (tmax > rmax(tmin, 0.0f)) && !(tmin > lim_n);
                    }
                    if (cand) {
                        if (COUNT) c_tris++;
                        TriHit h;
                        if (tri_test(q0, q1, q2, q3.z, oo, od, fp.delta, h)) {
                            uint32_t tri_id = __float_as_uint(q3.x);
                            if (candidate_better((fl & F_OHAVE) != 0u, h.key, tri_id, o_key, o_tri)) {
                                fl |= F_OHAVE; o_key = h.key; o_slot = sl; o_tri = tri_id;
                                float dist = sqrtf(h.key);
                                if (dist == dist) lim_n = rmin(lim_n, dist * (1.0f + PRUNE_REL) + slack_abs);
                            }
                        }
                    }
                };
                const float4* tp = reinterpret_cast<const float4*>(fp.tris + slot);
#if TRR_TRIS_PER_STEP >= 2
                // both records are requested before the first is used: the second one's latency hides behind the first test
                uint32_t sl = slot;
                const bool two = (cur & 3u) != 0u;
                const float4* tq = two ? tp + 4 : tp;
                float4 a0 = tp[0], a1 = tp[1], a2 = tp[2], a3 = tp[3];
                float4 b0 = tq[0], b1 = tq[1], b2 = tq[2], b3 = tq[3];
                test_one(sl, a0, a1, a2, a3);
                if (two) test_one(sl + 1u, b0, b1, b2, b3);
                // ref = LEAF | first << 2 | count-1: two triangles consumed -> first+2, count-2
                cur = ((cur & 3u) >= 2u) ? cur + 6u : pop();
#else
                test_one(slot, tp[0], tp[1], tp[2], tp[3]);
                // next triangle of the leaf (ref = LEAF | first << 2 | count-1  ->  first+1, count-2), else pop
                cur = (cur & 3u) ? cur + 3u : pop();
#endif
            }
        }
    }

#ifdef TRR_PAD_VGPRS
#pragma unroll
    for (int k = 0; k < TRR_PAD_VGPRS; ++k) asm volatile("" ::"v"(pad_regs[k]));
#endif
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
// k_setup: one chain ray's closest hit -> shaded-hit record + secondary rays (mod.rs:116-153, 199-206;
// object.rs:198-219).  Queue slots are claimed with one atomic per wavefront and queue.
// ------------------------------------------------------------------------------------------------

DEV void store_ray(RayRec* q, F3 o, F3 d, uint32_t sample, uint32_t meta) {
    float4* p = reinterpret_cast<float4*>(q);
    p[0] = make_float4(o.x, o.y, o.z, d.x);
    p[1] = make_float4(d.y, d.z, __uint_as_float(sample), __uint_as_float(meta));
}

template <bool LEVEL0>
__global__ void __launch_bounds__(256) k_setup(FrameParams fp, BatchParams bp, SetupIO io, uint32_t lvl) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t total = LEVEL0 ? bp.n_samples : *io.n_chain_in;
    const uint32_t L = fp.n_lights;
    const uint32_t cap = bp.cap;
    // whole wavefronts iterate together (the ballots below need every lane present)
    const uint32_t stride = gridDim.x * blockDim.x;
    const uint32_t rounds = (total + stride - 1) / stride;
    for (uint32_t it = 0; it < rounds; ++it) {
        uint32_t i = it * stride + blockIdx.x * blockDim.x + threadIdx.x;
        bool valid = i < total;
        bool hit = false, reflective = false, spawn = false;
        F3 P = f3(0, 0, 0), N = f3(0, 0, 0), D = f3(0, 0, 0);
        uint32_t sample = 0;
        if (LEVEL0 && valid && io.nlev[i] != NLEV_PENDING) {
            // a camera ray that missed everything (k_trace<RAYGEN> left nlev == 0): mod.rs:119, skybox(ray.direction)
            F3 c = f3(fp.sky_color[0], fp.sky_color[1], fp.sky_color[2]);
            if (fp.sky != nullptr) {   // (a constant sky needs no direction)
                uint32_t pix = i / fp.spp, si = i - pix * fp.spp;
                PixelPos pp = pixel_of(fp, bp, pix);
                c = skybox(fp, primary_dir(fp, pp.x, pp.y, si));
            }
            io.term[0 * (size_t)cap + i] = c.x;
            io.term[1 * (size_t)cap + i] = c.y;
            io.term[2 * (size_t)cap + i] = c.z;
            valid = false;
        }
        if (valid) {
            if (LEVEL0) {
                sample = i;
                uint32_t pix = i / fp.spp, si = i - pix * fp.spp;
                PixelPos pp = pixel_of(fp, bp, pix);
                D = primary_dir(fp, pp.x, pp.y, si);
            } else {
                const float4* q = reinterpret_cast<const float4*>(io.chain_in + i);
                float4 a = q[0], b = q[1];
                D = f3(a.w, b.x, b.y);
                sample = __float_as_uint(b.z);
            }
            const float4* hp = reinterpret_cast<const float4*>(io.hits + i);
            float4 h0 = hp[0], h1 = hp[1];
            uint32_t slot = __float_as_uint(h0.w);
            if (slot == 0xFFFFFFFFu || fp.max_depth == 0u) {
                // mod.rs:119: no hit -> skybox(ray.direction) ends the chain; mod.rs:125-127: with max_depth == 0
                // even a hit is shaded as skybox(ray.direction)
                F3 c = skybox(fp, D);
                io.term[0 * (size_t)cap + sample] = c.x;
                io.term[1 * (size_t)cap + sample] = c.y;
                io.term[2 * (size_t)cap + sample] = c.z;
                io.nlev[sample] = (uint8_t)lvl;
            } else {
                hit = true;
                const ObjRec* ob = fp.objs + __float_as_uint(h1.x);
                const float4* tp = reinterpret_cast<const float4*>(fp.tris + slot);
                float4 q0 = tp[0], q1 = tp[1], q2 = tp[2], q3 = tp[3];
                F3 a = f3(q0.x, q0.y, q0.z), b = f3(q0.w, q1.x, q1.y), c = f3(q1.z, q1.w, q2.x);
                // the winner's barycentrics (triangle.rs:84): the trace kernel kept only the slot and the origin its
                // last raycast used; object-space ray and exact test are the ones of k_trace (object.rs:186-189)
                TriHit th;
                th.u = th.v = th.w = 0.0f; th.point = f3(0, 0, 0); th.key = 0.0f;
                (void)tri_test(q0, q1, q2, q3.z, xform_point(ob->inv, f3(h0.x, h0.y, h0.z)), xform_vec(ob->inv, D), fp.delta, th);
                const float u = th.u, v = th.v, w = th.w;
                const InterpRec* ip = fp.interp + ob->tri_base + __float_as_uint(q3.x);
                const float4* ipp = reinterpret_cast<const float4*>(ip);
                float4 i0 = ipp[0], i1 = ipp[1], i2 = ipp[2], i3 = ipp[3];
                F3 na = f3(i0.x, i0.y, i0.z), nb = f3(i0.w, i1.x, i1.y), nc = f3(i1.z, i1.w, i2.x);
                // object.rs:201-203
                F3 point = (scale(a, u) + scale(b, v)) + scale(c, w);
                F3 normal = (scale(na, u) + scale(nb, v)) + scale(nc, w);
                float tu = (i2.y * u + i2.w * v) + i3.y * w;
                float tv = (i2.z * u + i3.x * v) + i3.z * w;
                // object.rs:209-210 (forward matrix for the normal too)
                P = xform_point(ob->fwd, point);
                N = xform_vec(ob->fwd, normal);
                int32_t mat = (int32_t)__float_as_uint(q3.y);
                // mod.rs:129-134
                F3 diff = f3(0.9f, 0.9f, 0.9f);
                if (mat >= 0) {
                    const MatRec* m = fp.mats + mat;
                    if (m->tex >= 0) diff = texture_lookup(fp, m->tex, tu, tv);
                    else if (m->flags & 1u) diff = f3(m->kd[0], m->kd[1], m->kd[2]);
                    reflective = (m->illum == 3 || m->illum == 4);   // material.rs:47-49
                }
                float* lv = io.level + (size_t)lvl * LEVEL_PLANES * cap + sample;
                lv[0 * (size_t)cap] = P.x; lv[1 * (size_t)cap] = P.y; lv[2 * (size_t)cap] = P.z;
                lv[3 * (size_t)cap] = N.x; lv[4 * (size_t)cap] = N.y; lv[5 * (size_t)cap] = N.z;
                lv[6 * (size_t)cap] = D.x; lv[7 * (size_t)cap] = D.y; lv[8 * (size_t)cap] = D.z;
                lv[9 * (size_t)cap] = diff.x; lv[10 * (size_t)cap] = diff.y; lv[11 * (size_t)cap] = diff.z;
                lv[12 * (size_t)cap] = __int_as_float(mat);
                io.nlev[sample] = (uint8_t)(lvl + 1);
                if (reflective) {
                    if (lvl + 1 < fp.max_depth) spawn = true;
                    else {
                        // mod.rs:125-127: the nested shade() is at max_depth -> skybox(reflection direction)
                        F3 c2 = skybox(fp, reflect(D, N));
                        io.term[0 * (size_t)cap + sample] = c2.x;
                        io.term[1 * (size_t)cap + sample] = c2.y;
                        io.term[2 * (size_t)cap + sample] = c2.z;
                    }
                }
            }
        }
        // ---- shadow rays: mod.rs:149-153, one per light, grouped per light inside the wavefront's chunk ----
        unsigned long long hm = __ballot(hit);
        if (hm != 0ull && L > 0) {
            uint32_t nh = (uint32_t)__popcll(hm);
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(io.n_shadow_out, nh * L);
            base = __shfl(base, 0);
            uint32_t rank = (uint32_t)__popcll(hm & ((1ull << lane) - 1ull));
            if (hit) {
                for (uint32_t l = 0; l < L; ++l) {
                    const LightRec* lt = fp.lights + l;
                    F3 ld = normalize(f3(lt->pos[0], lt->pos[1], lt->pos[2]) - P);   // mod.rs:149
                    store_ray(io.shadow_out + base + l * nh + rank, P + scale(ld, fp.delta), ld, sample,
                              lvl | 0x100u | (l << 16));
                }
            }
        }
        // ---- reflection ray: mod.rs:203-206 (traced once per hit; the value is light-independent) ----------
        unsigned long long sm = __ballot(spawn);
        if (sm != 0ull) {
            uint32_t ns = (uint32_t)__popcll(sm);
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(io.n_chain_out, ns);
            base = __shfl(base, 0);
            uint32_t rank = (uint32_t)__popcll(sm & ((1ull << lane) - 1ull));
            if (spawn) store_ray(io.chain_out + base + rank, P + scale(N, fp.delta), reflect(D, N), sample, lvl + 1);
        }
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

// Pass 1, one thread per SAMPLE: samples that hit something (nlev > 0) are evaluated deepest level first and their
// colour replaces the terminal colour in `term`; misses (nlev == 0) already hold skybox(direction) there.
__global__ void __launch_bounds__(256) k_resolve_samples(FrameParams fp, BatchParams bp, ResolveIO io) {
    const uint32_t L = fp.n_lights;
    const size_t cap = bp.cap;
    for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < bp.n_samples; s += gridDim.x * blockDim.x) {
        const int n = io.nlev[s];
        if (n == 0) continue;
        F3 R = f3(0.0f, 0.0f, 0.0f);
        {
            bool need_term = false;
            int32_t dm = __float_as_int(io.level[((size_t)(n - 1) * LEVEL_PLANES + 12) * cap + s]);
            if (dm >= 0) {
                int il = fp.mats[dm].illum;
                need_term = (il == 3 || il == 4);
            }
            if (need_term) R = f3(io.term[s], io.term[cap + s], io.term[2 * cap + s]);
        }
        for (int k = n - 1; k >= 0; --k) {
            const float* lv = io.level + (size_t)k * LEVEL_PLANES * cap + s;
            F3 P = f3(lv[0], lv[cap], lv[2 * cap]);
            F3 N = f3(lv[3 * cap], lv[4 * cap], lv[5 * cap]);
            F3 D = f3(lv[6 * cap], lv[7 * cap], lv[8 * cap]);
            F3 diff = f3(lv[9 * cap], lv[10 * cap], lv[11 * cap]);
            int32_t mat = __float_as_int(lv[12 * cap]);
            F3 spec = f3(0.9f, 0.9f, 0.9f);   // mod.rs:136-139
            bool specular = false, reflective = false;
            float ns = 1.0f;
            if (mat >= 0) {
                const MatRec* m = fp.mats + mat;
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
        io.term_out[s] = R.x;
        io.term_out[cap + s] = R.y;
        io.term_out[2 * cap + s] = R.z;
    }
}

// Pass 2, one thread per PIXEL: sum of the pixel's samples in ascending order, mean (mod.rs:248-249), quantise
// (render.rs:134-140), store into the shard's strip.
__global__ void __launch_bounds__(256) k_resolve_pixels(FrameParams fp, BatchParams bp, ResolveIO io) {
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
            if (valid) v = f3(io.term[s], io.term[cap + s], io.term[2 * cap + s]);
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
            color = f3(io.term[s], io.term[cap + s], io.term[2 * cap + s]);
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
// host-callable launchers (used by render.cpp)
// ------------------------------------------------------------------------------------------------
int trace_blocks_per_cu(bool raygen, bool count) {
    int nb = 0;
    hipError_t e;
    if (raygen) e = count ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_trace<true, true>, TRACE_THREADS, 0)
                          : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_trace<true, false>, TRACE_THREADS, 0);
    else e = count ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_trace<false, true>, TRACE_THREADS, 0)
                   : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_trace<false, false>, TRACE_THREADS, 0);
    if (e != hipSuccess || nb < 1) nb = 1;
    return nb;
}

size_t trace_spill_bytes(uint32_t grid) { return (size_t)grid * TRACE_THREADS * STACK_SPILL * sizeof(uint2); }

void launch_trace(bool raygen, bool count, uint32_t grid, hipStream_t st, const FrameParams& fp, const BatchParams& bp,
                  const TraceIO& io) {
    dim3 g(grid), b(TRACE_THREADS);
    if (raygen) {
        if (count) hipLaunchKernelGGL((k_trace<true, true>), g, b, 0, st, fp, bp, io);
        else hipLaunchKernelGGL((k_trace<true, false>), g, b, 0, st, fp, bp, io);
    } else {
        if (count) hipLaunchKernelGGL((k_trace<false, true>), g, b, 0, st, fp, bp, io);
        else hipLaunchKernelGGL((k_trace<false, false>), g, b, 0, st, fp, bp, io);
    }
}

void launch_setup(bool level0, uint32_t grid, hipStream_t st, const FrameParams& fp, const BatchParams& bp, const SetupIO& io,
                  uint32_t lvl) {
    dim3 g(grid), b(256);
    if (level0) hipLaunchKernelGGL((k_setup<true>), g, b, 0, st, fp, bp, io, lvl);
    else hipLaunchKernelGGL((k_setup<false>), g, b, 0, st, fp, bp, io, lvl);
}

void launch_resolve(uint32_t grid_samples, uint32_t grid_pixels, hipStream_t st, const FrameParams& fp, const BatchParams& bp,
                    const ResolveIO& io) {
    hipLaunchKernelGGL(k_resolve_samples, dim3(grid_samples), dim3(256), 0, st, fp, bp, io);
    const bool per_sample = fp.spp >= 2u && fp.spp <= 64u && (fp.spp & (fp.spp - 1u)) == 0u;   // k_resolve_pixels' coalesced form
    hipLaunchKernelGGL(k_resolve_pixels, dim3(per_sample ? grid_samples : grid_pixels), dim3(256), 0, st, fp, bp, io);
}

void launch_assemble(hipStream_t st, uint32_t width, uint32_t height, uint32_t bw, uint32_t bh, uint32_t nbx, uint32_t nby,
                     uint32_t n_shards, size_t strip_bytes, const uint8_t* strips, uint8_t* frame) {
    size_t npix = (size_t)width * height;
    uint32_t grid = (uint32_t)std::min<size_t>((npix + 255) / 256, 4096);
    hipLaunchKernelGGL(k_assemble, dim3(grid), dim3(256), 0, st, width, height, bw, bh, nbx, nby, n_shards, strip_bytes, strips,
                       frame);
}

}   // namespace trr
