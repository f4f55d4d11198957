// trayracer_oracle.cpp -- CPU restatement of the reference ray tracer (bircni/Raytracing "Trayracer").
//
// *** TEST INFRASTRUCTURE ONLY. ***  This file is the parity ORACLE.  Only tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg may build, load or call it.  The product (raytracing_amd/) never links,
// imports or falls back to anything in oracle/.
//
// *** PARITY UNPINNED. ***  The reference is Rust and cannot be compiled here (no cargo/rustc), it ships
// no tests, golden vectors or fixtures for this path, and its vector math / BVH live in
// crates that are not vendored (nalgebra ^0.33, bvh ^0.11, ordered-float ^5, rand ^0.9; Cargo.toml:59-90,
// Cargo.lock is git-ignored so there are no exact pins).  Everything below that restates crate internals
// is RECALLED from the public crate sources and is marked "[crate]".  The restatement is pinned only by
// the hand-derived known-answer tests in tests/test_oracle_kat.py and by internal cross-checks
// (brute force == BVH, literal per-light recursion == hoisted reflection).
//
// Conventions: IEEE binary32 everywhere, no FMA contraction (build with -ffp-contract=off), the same
// operation ORDER as the Rust expressions.  Each function cites the reference file:line it follows
// (paths relative to /root/reference/).
//
// Build: see oracle/Makefile  (g++ -O2 -ffp-contract=off -fno-fast-math).

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

namespace {

// ----------------------------------------------------------------------------------------------
// nalgebra restatement [crate]
// ----------------------------------------------------------------------------------------------
struct V3 {
    float x, y, z;
};
struct V2 {
    float x, y;
};

inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
inline V3 add(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 sub(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 neg(V3 a) { return v3(-a.x, -a.y, -a.z); }
inline V3 scale(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }     // Vector * f32
inline V3 lscale(float s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }    // f32 * Vector
inline V3 unscale(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }   // Vector / f32 (true divisions)
inline V3 cmul(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }   // component_mul
// [crate] nalgebra dot() special-cases len==3: a + b + c with a,b,c the three products.
inline float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
// [crate] nalgebra cross() for 3-vectors.
inline V3 cross(V3 a, V3 b) {
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
inline float norm_squared(V3 a) { return dot(a, a); }
inline float norm(V3 a) { return std::sqrt(norm_squared(a)); }
inline V3 normalize(V3 a) { return unscale(a, norm(a)); }
// [crate] try_normalize(min_norm): None iff norm <= min_norm.
inline bool try_normalize(V3 a, float min_norm, V3* out) {
    float n = norm(a);
    if (n <= min_norm) return false;   // NaN norm: comparison false -> Some(NaN vector), as in Rust
    *out = unscale(a, n);
    return true;
}
inline V2 add2(V2 a, V2 b) { return V2{a.x + b.x, a.y + b.y}; }
inline V2 scale2(V2 a, float s) { return V2{a.x * s, a.y * s}; }

// Rust f32::max / f32::min: if one operand is NaN the other is returned.
// (written inline: without -ffinite-math-only gcc turns std::fmax into a libm call per box-test operand)
inline float rmax(float a, float b) { return (a >= b || b != b) ? a : b; }
inline float rmin(float a, float b) { return (a <= b || b != b) ? a : b; }

// Column-major 4x4, element (r,c) at m[c*4+r] (nalgebra storage order).
struct M4 {
    float m[16];
    float& at(int r, int c) { return m[c * 4 + r]; }
    float at(int r, int c) const { return m[c * 4 + r]; }
};

// [crate] nalgebra Matrix*Matrix for small static matrices: per result column j a gemv,
// y = A.col(0)*b0j ; y = A.col(k)*bkj + y  (k = 1..), with alpha = 1, beta = 1 multiplications exact.
M4 mat_mul(const M4& a, const M4& b) {
    M4 r;
    for (int j = 0; j < 4; ++j) {
        for (int i = 0; i < 4; ++i) r.at(i, j) = a.at(i, 0) * b.at(0, j);
        for (int k = 1; k < 4; ++k)
            for (int i = 0; i < 4; ++i) r.at(i, j) = a.at(i, k) * b.at(k, j) + r.at(i, j);
    }
    return r;
}

// [crate] 3x3 top-left view times vector (gemv): ((m_i0*x0) + m_i1*x1) + m_i2*x2.
inline V3 mat3_mul(const M4& a, V3 v) {
    float r[3];
    for (int i = 0; i < 3; ++i) {
        float y = a.at(i, 0) * v.x;
        y = a.at(i, 1) * v.y + y;
        y = a.at(i, 2) * v.z + y;
        r[i] = y;
    }
    return v3(r[0], r[1], r[2]);
}
// [crate] Transform<TAffine> * Point = M3*p + translation column.
inline V3 transform_point(const M4& a, V3 p) {
    V3 r = mat3_mul(a, p);
    return v3(r.x + a.at(0, 3), r.y + a.at(1, 3), r.z + a.at(2, 3));
}
inline V3 transform_vector(const M4& a, V3 v) { return mat3_mul(a, v); }

// [crate] nalgebra do_inverse4 (the MESA gluInvertMatrix cofactor expansion), linear indices column-major.
bool inverse4(const M4& in, M4* outp) {
    const float* m = in.m;
    float o[16];
    o[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] +
           m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    o[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] -
           m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    o[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] +
           m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    o[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] -
            m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    o[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] -
           m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    o[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] +
           m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    o[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] -
           m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    o[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] +
            m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    o[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] +
           m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    o[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] -
           m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    o[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] +
            m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    o[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] -
            m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    o[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] -
           m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    o[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] +
           m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    o[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] -
            m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    o[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] +
            m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    float det = m[0] * o[0] + m[1] * o[4] + m[2] * o[8] + m[3] * o[12];
    if (det == 0.0f) return false;
    float inv_det = 1.0f / det;
    for (int i = 0; i < 16; ++i) outp->m[i] = o[i] * inv_det;
    return true;
}

// [crate] UnitQuaternion -> rotation matrix (to_rotation_matrix), q = (i, j, k, w).
void quat_to_mat3(const float q[4], float r[3][3]) {
    float i = q[0], j = q[1], k = q[2], w = q[3];
    float ww = w * w, ii = i * i, jj = j * j, kk = k * k;
    float ij = i * j * 2.0f, wk = w * k * 2.0f, wj = w * j * 2.0f;
    float ik = i * k * 2.0f, jk = j * k * 2.0f, wi = w * i * 2.0f;
    r[0][0] = ww + ii - jj - kk; r[0][1] = ij - wk;           r[0][2] = wj + ik;
    r[1][0] = wk + ij;           r[1][1] = ww - ii + jj - kk; r[1][2] = jk - wi;
    r[2][0] = ik - wj;           r[2][1] = wi + jk;           r[2][2] = ww - ii - jj + kk;
}

// ----------------------------------------------------------------------------------------------
// Scene data (mirrors src/scene/*.rs)
// ----------------------------------------------------------------------------------------------
struct Triangle {   // scene/triangle.rs:9-21
    V3 a, b, c, na, nb, nc;
    V2 uva, uvb, uvc;
    int32_t material;   // object-local index, -1 = None
};

enum : uint32_t { HAS_KD = 1, HAS_KS = 2, HAS_NS = 4, HAS_DISSOLVE = 8 };

struct Material {   // scene/material.rs:5-15
    V3 kd, ks;
    float ns, dissolve;
    int32_t illum;
    int32_t texture;    // index into scene textures, -1 = None
    uint32_t flags;
    bool specular() const { return illum == 2; }                   // material.rs:43-45
    bool reflection() const { return illum == 3 || illum == 4; }   // material.rs:47-49
    bool transparency() const { return illum == 6 || illum == 7; } // material.rs:51-53
};

struct Image {
    std::vector<uint8_t> rgb;
    uint32_t w = 0, h = 0;
};

struct Aabb {
    V3 lo, hi;
};

struct BvhNode {   // [crate] bvh::BvhNode: Node{child_l_aabb, child_l, child_r_aabb, child_r} | Leaf{shape}
    Aabb lbox, rbox;
    int32_t l, r;   // >= 0: node index; < 0: leaf, triangle = ~value
};

struct Object {   // scene/object.rs:19-29
    std::vector<Triangle> tris;
    std::vector<Material> mats;
    float translation[3];
    float rotation[4];   // unit quaternion i, j, k, w
    float scl[3];
    std::vector<BvhNode> nodes;   // empty when tris.size() <= 1
    int32_t root = 0;
};

struct Light {   // scene/light.rs:5-9
    V3 pos, color;
    float intensity;
};

struct Settings {   // scene/settings.rs:4-12, skybox.rs:6-9
    V3 ambient_color;
    float ambient_intensity;
    bool sky_is_image;
    V3 sky_color;
    Image sky;
    bool aa;
    uint32_t samples;
    uint64_t seed;   // ours: counter RNG seed replacing rand::random (raytracer/mod.rs:234,238)
};

struct Camera {   // scene/camera.rs:5-11
    V3 pos, look_at, up;
    float fov;   // radians
    uint32_t w, h;
};

struct Scene {
    std::vector<Object> objects;
    std::vector<Light> lights;
    std::vector<Image> textures;
    Settings settings;
    Camera camera;
};

struct Ray {   // raytracer/mod.rs:10-14
    V3 o, d;
};

struct Hit {   // raytracer/mod.rs:16-24 (+ ids for the parity tests)
    V3 point, normal;
    V2 uv;
    const Material* material;   // nullptr = None
    uint32_t obj, tri;
};

// scene/object.rs:177-182  Object::transform(): Isometry.to_homogeneous() * Scale.to_homogeneous()
M4 object_transform(const Object& o) {
    float r[3][3];
    quat_to_mat3(o.rotation, r);
    M4 iso{}, sc{};
    for (int i = 0; i < 16; ++i) iso.m[i] = sc.m[i] = 0.0f;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) iso.at(i, j) = r[i][j];
    for (int i = 0; i < 3; ++i) iso.at(i, 3) = o.translation[i];
    iso.at(3, 3) = 1.0f;
    sc.at(0, 0) = o.scl[0]; sc.at(1, 1) = o.scl[1]; sc.at(2, 2) = o.scl[2]; sc.at(3, 3) = 1.0f;
    return mat_mul(iso, sc);
}

// scene/triangle.rs:53-85  Triangle::intersect -> barycentric weights of (a, b, c)
bool triangle_intersect(const Triangle& t, const Ray& ray, float delta, float* u, float* v, float* w) {
    V3 ab = sub(t.b, t.a);
    V3 ac = sub(t.c, t.a);
    V3 normal;
    if (!try_normalize(cross(ab, ac), delta, &normal)) return false;   // :56
    float tt = dot(sub(t.a, ray.o), normal) / dot(ray.d, normal);      // :58
    if (tt < delta) return false;                                      // :60 (NaN falls through)
    V3 p = add(ray.o, scale(ray.d, tt));                               // :64
    V3 ap = sub(p, t.a), bp = sub(p, t.b), cp = sub(p, t.c);           // :66-68
    V3 ab_ap = cross(ab, ap);                                          // :70
    V3 bc_bp = cross(sub(t.c, t.b), bp);                               // :71
    V3 ca_cp = cross(sub(t.a, t.c), cp);                               // :72
    float e0 = dot(ab_ap, normal), e1 = dot(bc_bp, normal), e2 = dot(ca_cp, normal);   // :74-76
    if (e0 < 0.0f || e1 < 0.0f || e2 < 0.0f) return false;             // :78
    float area = e0 + e1 + e2;                                         // :82
    *u = e1 / area; *v = e2 / area; *w = e0 / area;                    // :84
    return true;
}

// scene/triangle.rs:88-95  Bounded::aabb: empty().grow(a).grow(b).grow(c + 1e-4)
Aabb triangle_aabb(const Triangle& t) {
    V3 c2 = add(t.c, v3(0.0001f, 0.0001f, 0.0001f));
    Aabb b;
    b.lo = v3(rmin(rmin(t.a.x, t.b.x), c2.x), rmin(rmin(t.a.y, t.b.y), c2.y), rmin(rmin(t.a.z, t.b.z), c2.z));
    b.hi = v3(rmax(rmax(t.a.x, t.b.x), c2.x), rmax(rmax(t.a.y, t.b.y), c2.y), rmax(rmax(t.a.z, t.b.z), c2.z));
    return b;
}
inline Aabb aabb_join(const Aabb& a, const Aabb& b) {
    Aabb r;
    r.lo = v3(rmin(a.lo.x, b.lo.x), rmin(a.lo.y, b.lo.y), rmin(a.lo.z, b.lo.z));
    r.hi = v3(rmax(a.hi.x, b.hi.x), rmax(a.hi.y, b.hi.y), rmax(a.hi.z, b.hi.z));
    return r;
}

// [crate] bvh::ray::Ray::new normalises the direction and stores 1/direction;
// intersects_aabb: lbr=(min-o)*inv, rtr=(max-o)*inv, tmin=max_i(min(lbr,rtr)), tmax=min_i(max(lbr,rtr)),
// accept iff tmax > max(tmin, 0).
struct BvhRay {
    V3 o, inv;
};
inline BvhRay bvh_ray_new(V3 o, V3 d) {
    V3 dn = normalize(d);
    return BvhRay{o, v3(1.0f / dn.x, 1.0f / dn.y, 1.0f / dn.z)};
}
inline bool ray_intersects_aabb(const BvhRay& r, const Aabb& b) {
    V3 lbr = cmul(sub(b.lo, r.o), r.inv);
    V3 rtr = cmul(sub(b.hi, r.o), r.inv);
    float tmin = rmax(rmax(rmin(lbr.x, rtr.x), rmin(lbr.y, rtr.y)), rmin(lbr.z, rtr.z));
    float tmax = rmin(rmin(rmax(lbr.x, rtr.x), rmax(lbr.y, rtr.y)), rmax(lbr.z, rtr.z));
    return tmax > rmax(tmin, 0.0f);
}

// [crate] Bvh::build: binary tree, one shape per leaf, inner nodes hold both children's boxes
// (exact unions).  Split = binned SAH over the largest centroid axis (6 buckets like the crate).
// The topology does not influence results (see DESIGN.md "candidate-set invariant"), only CPU time.
struct Builder {
    const std::vector<Aabb>& boxes;
    std::vector<V3> cent;
    std::vector<BvhNode>& nodes;
    Builder(const std::vector<Aabb>& b, std::vector<BvhNode>& n) : boxes(b), nodes(n) {
        cent.resize(b.size());
        for (size_t i = 0; i < b.size(); ++i)
            cent[i] = v3(0.5f * (b[i].lo.x + b[i].hi.x), 0.5f * (b[i].lo.y + b[i].hi.y), 0.5f * (b[i].lo.z + b[i].hi.z));
    }
    static float area(const Aabb& b) {
        float dx = b.hi.x - b.lo.x, dy = b.hi.y - b.lo.y, dz = b.hi.z - b.lo.z;
        return 2.0f * (dx * dy + dy * dz + dz * dx);
    }
    static float comp(V3 v, int a) { return a == 0 ? v.x : (a == 1 ? v.y : v.z); }
    // builds the subtree over idx[lo,hi) (hi-lo >= 1); returns child reference and its box
    int32_t build(std::vector<uint32_t>& idx, size_t lo, size_t hi, Aabb* box) {
        if (hi - lo == 1) {
            *box = boxes[idx[lo]];
            return ~(int32_t)idx[lo];
        }
        V3 cmin = cent[idx[lo]], cmax = cmin;
        for (size_t i = lo + 1; i < hi; ++i) {
            V3 c = cent[idx[i]];
            cmin = v3(std::min(cmin.x, c.x), std::min(cmin.y, c.y), std::min(cmin.z, c.z));
            cmax = v3(std::max(cmax.x, c.x), std::max(cmax.y, c.y), std::max(cmax.z, c.z));
        }
        V3 ext = sub(cmax, cmin);
        int axis = (ext.x >= ext.y && ext.x >= ext.z) ? 0 : (ext.y >= ext.z ? 1 : 2);
        size_t mid = lo + (hi - lo) / 2;
        float e = comp(ext, axis);
        if (e > 0.0f && std::isfinite(e)) {
            const int NB = 6;
            Aabb bb[NB];
            size_t cnt[NB];
            for (int k = 0; k < NB; ++k) {
                cnt[k] = 0;
                bb[k].lo = v3(INFINITY, INFINITY, INFINITY);
                bb[k].hi = v3(-INFINITY, -INFINITY, -INFINITY);
            }
            float c0 = comp(cmin, axis);
            float sc = (float)NB / e;
            auto bucket = [&](uint32_t t) {
                int k = (int)((comp(cent[t], axis) - c0) * sc);
                return std::min(std::max(k, 0), NB - 1);
            };
            for (size_t i = lo; i < hi; ++i) {
                int k = bucket(idx[i]);
                cnt[k]++;
                bb[k] = aabb_join(bb[k], boxes[idx[i]]);
            }
            float best = INFINITY;
            int bestk = -1;
            for (int k = 0; k < NB - 1; ++k) {
                Aabb l{v3(INFINITY, INFINITY, INFINITY), v3(-INFINITY, -INFINITY, -INFINITY)}, r = l;
                size_t nl = 0, nr = 0;
                for (int q = 0; q <= k; ++q) if (cnt[q]) { l = aabb_join(l, bb[q]); nl += cnt[q]; }
                for (int q = k + 1; q < NB; ++q) if (cnt[q]) { r = aabb_join(r, bb[q]); nr += cnt[q]; }
                if (!nl || !nr) continue;
                float cost = area(l) * (float)nl + area(r) * (float)nr;
                if (cost < best) { best = cost; bestk = k; }
            }
            if (bestk >= 0) {
                auto it = std::partition(idx.begin() + lo, idx.begin() + hi, [&](uint32_t t) { return bucket(t) <= bestk; });
                size_t m = (size_t)(it - idx.begin());
                if (m > lo && m < hi) mid = m;
            }
        }
        int32_t me = (int32_t)nodes.size();
        nodes.push_back(BvhNode{});
        Aabb lb, rb;
        int32_t l = build(idx, lo, mid, &lb);
        int32_t r = build(idx, mid, hi, &rb);
        nodes[me].lbox = lb; nodes[me].rbox = rb; nodes[me].l = l; nodes[me].r = r;
        *box = aabb_join(lb, rb);
        return me;
    }
};

void object_build_bvh(Object& o) {   // scene/object.rs:156  Bvh::build(triangles)
    o.nodes.clear();
    size_t n = o.tris.size();
    if (n <= 1) return;
    std::vector<Aabb> boxes(n);
    for (size_t i = 0; i < n; ++i) boxes[i] = triangle_aabb(o.tris[i]);
    std::vector<uint32_t> idx(n);
    for (size_t i = 0; i < n; ++i) idx[i] = (uint32_t)i;
    o.nodes.reserve(n);
    Builder b(boxes, o.nodes);
    Aabb root;
    o.root = b.build(idx, 0, n, &root);
}

// ----------------------------------------------------------------------------------------------
// Counters (ours; the reference keeps none)
// ----------------------------------------------------------------------------------------------
struct Counters {
    uint64_t raycasts = 0;        // every Raytracer::raycast call actually made in the selected mode
    uint64_t node_visits = 0;     // BVH inner nodes visited
    uint64_t tri_tests = 0;       // Triangle::intersect calls
    uint64_t shade_calls = 0;     // shade_impl bodies evaluated past the depth check
    void add(const Counters& o) {
        raycasts += o.raycasts; node_visits += o.node_visits; tri_tests += o.tri_tests; shade_calls += o.shade_calls;
    }
};

enum Accel : int { ACCEL_BRUTE = 0, ACCEL_BVH = 1, ACCEL_BRUTE_NOBOX = 2 };
enum ShadeMode : int { SHADE_LITERAL = 0, SHADE_HOISTED = 1 };

struct Raytracer {   // raytracer/mod.rs:26-43
    const Scene& scene;
    float delta;
    uint32_t max_depth;
    int accel;
    int shade_mode;

    // scene/object.rs:184-220  Object::intersect
    bool object_intersect(uint32_t oi, const Ray& wray, Hit* out, Counters& cn) const {
        const Object& obj = scene.objects[oi];
        // :186-189 -- transform() is rebuilt and inverted for the point and again for the vector
        M4 inv_p, inv_v;
        if (!inverse4(object_transform(obj), &inv_p)) inv_p = object_transform(obj);   // [crate] non-invertible: left unchanged
        if (!inverse4(object_transform(obj), &inv_v)) inv_v = object_transform(obj);
        Ray ray{transform_point(inv_p, wray.o), transform_vector(inv_v, wray.d)};

        bool have = false;
        float best_key = 0.0f;
        uint32_t best_tri = 0;
        V3 best_point{}, best_normal{};
        V2 best_uv{};

        auto consider = [&](uint32_t ti) {
            const Triangle& t = obj.tris[ti];
            float u, v, w;
            cn.tri_tests++;
            if (!triangle_intersect(t, ray, delta, &u, &v, &w)) return;   // :197
            // :201-203 interpolate
            V3 point = add(add(scale(t.a, u), scale(t.b, v)), scale(t.c, w));
            V3 normal = add(add(scale(t.na, u), scale(t.nb, v)), scale(t.nc, w));
            V2 uv = add2(add2(scale2(t.uva, u), scale2(t.uvb, v)), scale2(t.uvc, w));
            float key = norm_squared(sub(ray.o, point));                  // :206
            // min_by_key(OrderedFloat): NaN is the greatest key, equal keys keep the FIRST candidate.
            // The crate's DFS candidate order is unknowable here, so "first" := lowest triangle index
            // (all accelerators in this file and the GPU path use the same rule).
            bool better;
            if (!have) better = true;
            else if (std::isnan(key)) better = std::isnan(best_key) && ti < best_tri;
            else if (std::isnan(best_key)) better = true;
            else better = key < best_key || (key == best_key && ti < best_tri);
            if (better) {
                have = true; best_key = key; best_tri = ti;
                best_point = point; best_normal = normal; best_uv = uv;
            }
        };

        size_t n = obj.tris.size();
        if (accel == ACCEL_BRUTE_NOBOX) {
            for (uint32_t i = 0; i < n; ++i) consider(i);
        } else if (n == 1) {
            consider(0);   // [crate] a root Leaf is returned by traverse() without any box test
        } else if (n > 1) {
            BvhRay br = bvh_ray_new(ray.o, ray.d);   // :193
            if (accel == ACCEL_BRUTE) {
                for (uint32_t i = 0; i < n; ++i)
                    if (ray_intersects_aabb(br, triangle_aabb(obj.tris[i]))) consider(i);
            } else {
                // [crate] Bvh::traverse: recursive, left child first, every leaf whose box chain passes
                int32_t stack[128];
                int sp = 0;
                stack[sp++] = obj.root;
                while (sp > 0) {
                    int32_t ref = stack[--sp];
                    if (ref < 0) { consider((uint32_t)~ref); continue; }
                    const BvhNode& nd = obj.nodes[ref];
                    cn.node_visits++;
                    bool hl = ray_intersects_aabb(br, nd.lbox), hr = ray_intersects_aabb(br, nd.rbox);
                    if (hr) stack[sp++] = nd.r;
                    if (hl) stack[sp++] = nd.l;
                    if (sp > 120) break;   // unreachable for SAH trees of sane depth
                }
            }
        }
        if (!have) return false;
        // :207-219 back to world space (transform() rebuilt twice more)
        M4 fwd = object_transform(obj);
        out->point = transform_point(fwd, best_point);
        out->normal = transform_vector(object_transform(obj), best_normal);
        out->uv = best_uv;
        int32_t mi = obj.tris[best_tri].material;
        out->material = (mi >= 0 && (size_t)mi < obj.mats.size()) ? &obj.mats[mi] : nullptr;
        out->obj = oi; out->tri = best_tri;
        return true;
    }

    // raytracer/mod.rs:45-51  raycast: min_by_key(|h| OrderedFloat((h.point - ray.origin).norm())), first wins ties
    bool raycast(const Ray& ray, Hit* out, Counters& cn) const {
        cn.raycasts++;
        bool have = false;
        float best = 0.0f;
        for (uint32_t oi = 0; oi < scene.objects.size(); ++oi) {
            Hit h;
            if (!object_intersect(oi, ray, &h, cn)) continue;
            float key = norm(sub(h.point, ray.o));
            bool better;
            if (!have) better = true;
            else if (std::isnan(key)) better = false;
            else if (std::isnan(best)) better = true;
            else better = key < best;
            if (better) { have = true; best = key; *out = h; }
        }
        return have;
    }

    // raytracer/mod.rs:53-55
    static V3 reflect(V3 incoming, V3 normal) { return sub(incoming, lscale(2.0f * dot(incoming, normal), normal)); }

    // raytracer/mod.rs:57-82
    V3 skybox(V3 direction) const {
        const Settings& s = scene.settings;
        if (!s.sky_is_image) return s.sky_color;
        V3 d;
        if (!try_normalize(direction, std::numeric_limits<float>::epsilon(), &d)) d = v3(0.0f, 1.0f, 0.0f);
        const float PI = 3.14159265358979323846f;
        float fx = (0.5f + std::atan2(d.z, d.x) / (2.0f * PI)) * (float)s.sky.w;
        float fy = (0.5f - std::asin(d.y) / PI) * (float)s.sky.h;
        uint32_t x = f32_as_u32(fx) % s.sky.w;
        uint32_t y = f32_as_u32(fy) % s.sky.h;
        const uint8_t* p = &s.sky.rgb[((size_t)y * s.sky.w + x) * 3];
        return v3((float)p[0] / 255.0f, (float)p[1] / 255.0f, (float)p[2] / 255.0f);
    }

    // Rust `f32 as u32`: saturating, NaN -> 0
    static uint32_t f32_as_u32(float f) {
        if (!(f > 0.0f)) return 0;   // NaN, negatives, zero
        if (f >= 4294967296.0f) return 0xFFFFFFFFu;
        return (uint32_t)f;
    }

    // raytracer/mod.rs:84-93
    static V3 texture(const Image& t, V2 uv) {
        uint32_t x = f32_as_u32(uv.x * (float)t.w) % t.w;
        uint32_t y = f32_as_u32((1.0f - uv.y) * (float)t.h) % t.h;
        const uint8_t* p = &t.rgb[((size_t)y * t.w + x) * 3];
        return v3((float)p[0] / 255.0f, (float)p[1] / 255.0f, (float)p[2] / 255.0f);
    }

    // raytracer/mod.rs:96-114  raycast_transparent(...).last()
    // The reference's loop has no bound and spins for ever on a ray that cannot advance (zero or NaN direction inside a
    // transparent surface).  Restatement and GPU kernels share a cap of 65535 marching steps per ray (3 276 scene units of
    // glass at |d| = 1), reachable only where the reference itself would not return.
    bool raycast_transparent_last(Ray ray, Hit* last, Counters& cn) const {
        bool have = false;
        Hit h;
        uint32_t marches = 0;
        while (raycast(ray, &h, cn)) {
            *last = h; have = true;
            if (h.material && h.material->transparency() && marches < 0xFFFFu) {
                ray.o = add(ray.o, scale(ray.d, 0.05f));   // :106
                ++marches;
                continue;
            }
            break;
        }
        return have;
    }

    // raytracer/mod.rs:116-122
    V3 shade(const Ray& ray, uint32_t depth, Counters& cn) const {
        Hit hit;
        if (!raycast_transparent_last(ray, &hit, cn)) return skybox(ray.d);
        return shade_impl(ray, hit, depth, cn);
    }

    // raytracer/mod.rs:124-212
    V3 shade_impl(const Ray& ray, const Hit& hit, uint32_t depth, Counters& cn) const {
        if (depth >= max_depth) return skybox(ray.d);   // :125-127
        cn.shade_calls++;
        const Material* m = hit.material;
        const V3 NO_MATERIAL = v3(0.9f, 0.9f, 0.9f);    // :35
        V3 diffuse_color;                               // :129-134
        if (m && m->texture >= 0) diffuse_color = texture(scene.textures[m->texture], hit.uv);
        else if (m && (m->flags & HAS_KD)) diffuse_color = m->kd;
        else diffuse_color = NO_MATERIAL;
        V3 specular_color = (m && (m->flags & HAS_KS)) ? m->ks : NO_MATERIAL;   // :136-139
        V3 color = scale(cmul(scene.settings.ambient_color, diffuse_color), scene.settings.ambient_intensity);   // :141-146

        bool reflective = m && m->reflection();
        bool have_refl = false;
        V3 refl{};
        for (const Light& light : scene.lights) {   // :148
            V3 light_direction = normalize(sub(light.pos, hit.point));   // :149
            Ray light_ray{add(hit.point, scale(light_direction, delta)), light_direction};   // :150-153
            Hit occ;
            V3 trans;
            if (!raycast_transparent_last(light_ray, &occ, cn)) {
                trans = v3(1.0f, 1.0f, 1.0f);   // :160
            } else {
                V3 okd = (occ.material && (occ.material->flags & HAS_KD)) ? occ.material->kd : v3(1.0f, 1.0f, 1.0f);
                float odis = (occ.material && (occ.material->flags & HAS_DISSOLVE)) ? occ.material->dissolve : 1.0f;
                trans = scale(cmul(color, okd), odis);   // :162-166
            }
            trans = cmul(trans, light.color);   // :169
            if (norm(trans) < 0.01f) continue;  // :171-173
            float light_intensity = light.intensity / norm_squared(sub(light.pos, hit.point));   // :176
            float diffuse_intensity = rmax(dot(light_direction, hit.normal), 0.0f) * light_intensity;   // :177
            color = add(color, scale(cmul(diffuse_color, trans), diffuse_intensity));   // :178
            if (m && m->specular()) {   // :181-196
                float ns = (m->flags & HAS_NS) ? m->ns : 1.0f;
                float si = std::pow(rmax(dot(light_direction, reflect(neg(ray.d), hit.normal)), 0.0f), ns) * light_intensity;
                color = add(color, scale(cmul(specular_color, trans), si));
            }
            if (reflective) {   // :199-208
                if (shade_mode == SHADE_LITERAL || !have_refl) {
                    Ray rr{add(hit.point, scale(hit.normal, delta)), reflect(ray.d, hit.normal)};
                    refl = shade(rr, depth + 1, cn);
                    have_refl = true;
                }
                color = add(color, cmul(color, refl));   // :207
            }
        }
        return color;
    }

    // counter RNG standing in for rand::random::<f32>() (raytracer/mod.rs:234,238): rand 0.9's
    // StandardUniform f32 = (u32 >> 8) * 2^-24.  The u32 comes from a fixed integer hash of
    // (seed, pixel, sample, dimension) so CPU and GPU, and any GPU count, draw the same numbers.
    static uint32_t hash_u32(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t dim) {
        uint64_t z = seed + 0x9E3779B97F4A7C15ull * ((((uint64_t)pixel << 32) | ((uint64_t)sample << 1) | dim) + 1ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z = z ^ (z >> 31);
        return (uint32_t)(z >> 32);
    }
    static float rand_f32(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t dim) {
        return (float)(hash_u32(seed, pixel, sample, dim) >> 8) * (1.0f / 16777216.0f);
    }

    // scene/camera.rs:30-42
    Ray camera_ray(float x, float y) const {
        const Camera& c = scene.camera;
        V3 direction = v3(x, -y, -1.0f / std::tan(c.fov / 2.0f));   // :32
        // :35 Rotation3::look_at_rh(dir, up) = face_towards(-dir, up).inverse()   [crate]
        V3 dirv = sub(c.look_at, c.pos);
        V3 zaxis = normalize(neg(dirv));
        V3 xaxis = normalize(cross(c.up, zaxis));
        V3 yaxis = normalize(cross(zaxis, xaxis));
        // :36 inverse_transform_vector = R^T * v with R = [x y z]^T, i.e. tr_mul: rows of [x y z] dotted with v
        V3 d = v3(dot(v3(xaxis.x, yaxis.x, zaxis.x), direction),
                  dot(v3(xaxis.y, yaxis.y, zaxis.y), direction),
                  dot(v3(xaxis.z, yaxis.z, zaxis.z), direction));
        return Ray{c.pos, normalize(d)};   // :38-41
    }

    // raytracer/mod.rs:218-257  (ids: first raycast of the pixel's ray, non-AA only)
    V3 render(uint32_t x, uint32_t y, uint32_t width, uint32_t height, bool aa, Counters& cn,
              uint32_t* hit_obj, uint32_t* hit_tri) const {
        if (aa) {
            uint32_t n = scene.settings.samples;
            uint32_t s = Raytracer::f32_as_u32(std::sqrt((float)n));   // :226
            V3 sum = v3(0.0f, 0.0f, 0.0f);
            uint32_t pixel = y * width + x;
            for (uint32_t i = 0; i < n; ++i) {   // :228 (rayon order is unspecified; we fix i ascending)
                uint32_t xi = i % s, yi = i / s;
                float rx = rand_f32(scene.settings.seed, pixel, i, 0);
                float ry = rand_f32(scene.settings.seed, pixel, i, 1);
                float jx = ((float)x + ((float)xi + (rx * 2.0f + -1.0f)) / (float)s) / (float)width;    // :233-236
                float jy = ((float)y + ((float)yi + (ry * 2.0f + -1.0f)) / (float)s) / (float)height;   // :237-240
                float fx = (jx * 2.0f + -1.0f) * ((float)width / (float)height);   // :241 (x*2 is exact, so fused == unfused)
                float fy = jy * 2.0f + -1.0f;                                      // :242
                Ray ray = camera_ray(fx, fy);
                // :245-246 raycast, then shade (which re-raycasts: same answer) or skybox
                sum = add(sum, shade(ray, 0, cn));
            }
            if (hit_obj) { *hit_obj = 0xFFFFFFFFu; *hit_tri = 0xFFFFFFFFu; }
            return unscale(sum, (float)n);   // :249
        }
        float fx = ((float)x / (float)width * 2.0f + -1.0f) * ((float)width / (float)height);   // :251
        float fy = (float)y / (float)height * 2.0f + -1.0f;                                     // :252
        Ray ray = camera_ray(fx, fy);
        if (hit_obj) {
            Hit h;
            Counters scratch;
            if (raycast(ray, &h, scratch)) { *hit_obj = h.obj; *hit_tri = h.tri; }
            else { *hit_obj = 0xFFFFFFFFu; *hit_tri = 0xFFFFFFFFu; }
        }
        return shade(ray, 0, cn);   // :255
    }
};

// raytracer/render.rs:136-138  (c * 255.0) as u8 : truncating, saturating, NaN -> 0
inline uint8_t quantize(float c) {
    float v = c * 255.0f;
    if (!(v > 0.0f)) return 0;
    if (v >= 255.0f) return 255;
    return (uint8_t)v;
}

}   // namespace

// ----------------------------------------------------------------------------------------------
// C ABI used by oracle/oracle.py (ctypes)
// ----------------------------------------------------------------------------------------------
extern "C" {

struct orc_object {
    const float* tri;        // n_tris x 24: a b c na nb nc (3 each) uva uvb uvc (2 each)
    const int32_t* material; // n_tris, object-local material index, -1 = None
    uint32_t n_tris;
    const void* mats;        // orc_material[n_mats]
    uint32_t n_mats;
    float translation[3];
    float rotation[4];       // unit quaternion i j k w
    float scale[3];
};
struct orc_material {
    float kd[3], ks[3];
    float ns, dissolve;
    int32_t illum, texture;
    uint32_t flags;
};
struct orc_image {
    const uint8_t* rgb;
    uint32_t w, h;
};
struct orc_light {
    float pos[3], color[3];
    float intensity;
};
struct orc_settings {
    float ambient_color[3];
    float ambient_intensity;
    int32_t sky_is_image;
    float sky_color[3];
    orc_image sky;
    int32_t aa;
    uint32_t samples;
    uint64_t seed;
};
struct orc_camera {
    float pos[3], look_at[3], up[3];
    float fov;
    uint32_t w, h;
};
struct orc_counters {
    uint64_t raycasts, node_visits, tri_tests, shade_calls;
    double seconds;   // steady clock around the block loop only (raytracer/render.rs:96,189)
};

void* orc_scene_create(const orc_object* objs, uint32_t n_objs, const orc_image* textures, uint32_t n_tex,
                       const orc_light* lights, uint32_t n_lights, const orc_settings* st, const orc_camera* cam) {
    Scene* s = new Scene();
    s->objects.resize(n_objs);
    for (uint32_t i = 0; i < n_objs; ++i) {
        Object& o = s->objects[i];
        const orc_object& d = objs[i];
        o.tris.resize(d.n_tris);
        for (uint32_t t = 0; t < d.n_tris; ++t) {
            const float* f = d.tri + (size_t)t * 24;
            Triangle& T = o.tris[t];
            T.a = v3(f[0], f[1], f[2]); T.b = v3(f[3], f[4], f[5]); T.c = v3(f[6], f[7], f[8]);
            T.na = v3(f[9], f[10], f[11]); T.nb = v3(f[12], f[13], f[14]); T.nc = v3(f[15], f[16], f[17]);
            T.uva = V2{f[18], f[19]}; T.uvb = V2{f[20], f[21]}; T.uvc = V2{f[22], f[23]};
            T.material = d.material ? d.material[t] : -1;
        }
        const orc_material* ms = (const orc_material*)d.mats;
        o.mats.resize(d.n_mats);
        for (uint32_t m = 0; m < d.n_mats; ++m) {
            Material& M = o.mats[m];
            M.kd = v3(ms[m].kd[0], ms[m].kd[1], ms[m].kd[2]);
            M.ks = v3(ms[m].ks[0], ms[m].ks[1], ms[m].ks[2]);
            M.ns = ms[m].ns; M.dissolve = ms[m].dissolve; M.illum = ms[m].illum;
            M.texture = ms[m].texture; M.flags = ms[m].flags;
        }
        for (int k = 0; k < 3; ++k) { o.translation[k] = d.translation[k]; o.scl[k] = d.scale[k]; }
        for (int k = 0; k < 4; ++k) o.rotation[k] = d.rotation[k];
        object_build_bvh(o);
    }
    s->textures.resize(n_tex);
    for (uint32_t i = 0; i < n_tex; ++i) {
        s->textures[i].w = textures[i].w; s->textures[i].h = textures[i].h;
        s->textures[i].rgb.assign(textures[i].rgb, textures[i].rgb + (size_t)textures[i].w * textures[i].h * 3);
    }
    s->lights.resize(n_lights);
    for (uint32_t i = 0; i < n_lights; ++i) {
        s->lights[i].pos = v3(lights[i].pos[0], lights[i].pos[1], lights[i].pos[2]);
        s->lights[i].color = v3(lights[i].color[0], lights[i].color[1], lights[i].color[2]);
        s->lights[i].intensity = lights[i].intensity;
    }
    s->settings.ambient_color = v3(st->ambient_color[0], st->ambient_color[1], st->ambient_color[2]);
    s->settings.ambient_intensity = st->ambient_intensity;
    s->settings.sky_is_image = st->sky_is_image != 0;
    s->settings.sky_color = v3(st->sky_color[0], st->sky_color[1], st->sky_color[2]);
    if (st->sky_is_image) {
        s->settings.sky.w = st->sky.w; s->settings.sky.h = st->sky.h;
        s->settings.sky.rgb.assign(st->sky.rgb, st->sky.rgb + (size_t)st->sky.w * st->sky.h * 3);
    }
    s->settings.aa = st->aa != 0;
    s->settings.samples = st->samples;
    s->settings.seed = st->seed;
    s->camera.pos = v3(cam->pos[0], cam->pos[1], cam->pos[2]);
    s->camera.look_at = v3(cam->look_at[0], cam->look_at[1], cam->look_at[2]);
    s->camera.up = v3(cam->up[0], cam->up[1], cam->up[2]);
    s->camera.fov = cam->fov; s->camera.w = cam->w; s->camera.h = cam->h;
    return s;
}

void orc_scene_destroy(void* s) { delete (Scene*)s; }

void orc_scene_set_camera(void* sp, const orc_camera* cam) {
    Scene* s = (Scene*)sp;
    s->camera.pos = v3(cam->pos[0], cam->pos[1], cam->pos[2]);
    s->camera.look_at = v3(cam->look_at[0], cam->look_at[1], cam->look_at[2]);
    s->camera.up = v3(cam->up[0], cam->up[1], cam->up[2]);
    s->camera.fov = cam->fov; s->camera.w = cam->w; s->camera.h = cam->h;
}

void orc_scene_set_aa(void* sp, int32_t aa, uint32_t samples, uint64_t seed) {
    Scene* s = (Scene*)sp;
    s->settings.aa = aa != 0; s->settings.samples = samples; s->settings.seed = seed;
}

// raytracer/render.rs:95-193  RenderingThread::run  (block grid, quantise, remainder strip left black).
// block_first/block_step/block_limit select a bounded sample of blocks (block ids row-major):
// ids block_first, block_first+block_step, ... < min(block_limit, n_blocks).  rgb (w*h*3) is zeroed first.
// Returns 0, or -1 when w < 20 or h < 20 (the reference divides by zero at render.rs:109).
int orc_render(void* sp, float delta, uint32_t max_depth, int accel, int shade_mode, uint32_t n_threads,
               uint32_t block_first, uint32_t block_step, uint32_t block_limit,
               uint8_t* rgb, float* rgb_f32, uint32_t* hit_obj, uint32_t* hit_tri, orc_counters* out) {
    Scene* s = (Scene*)sp;
    uint32_t w = s->camera.w, h = s->camera.h;
    uint32_t bw = w / 20, bh = h / 20;   // render.rs:103
    if (bw == 0 || bh == 0) return -1;
    uint32_t nbx = w / bw, nby = h / bh; // render.rs:109-110
    uint32_t nblocks = nbx * nby;
    if (block_step == 0) block_step = 1;
    if (block_limit > nblocks) block_limit = nblocks;
    std::memset(rgb, 0, (size_t)w * h * 3);
    if (rgb_f32) std::memset(rgb_f32, 0, (size_t)w * h * 3 * sizeof(float));
    if (hit_obj) std::memset(hit_obj, 0xFF, (size_t)w * h * 4);
    if (hit_tri) std::memset(hit_tri, 0xFF, (size_t)w * h * 4);
    Raytracer rt{*s, delta, max_depth, accel, shade_mode};   // render.rs:105
    bool aa = s->settings.aa;                                // render.rs:104
    if (n_threads == 0) n_threads = std::max(1u, std::thread::hardware_concurrency());
    std::atomic<uint32_t> next{0};
    std::vector<Counters> cns(n_threads);
    uint32_t nsel = block_first < block_limit ? (block_limit - block_first + block_step - 1) / block_step : 0;
    auto worker = [&](uint32_t tid) {
        Counters cn;
        for (;;) {
            uint32_t k = next.fetch_add(1);
            if (k >= nsel) break;
            uint32_t b = block_first + k * block_step;
            uint32_t xb = b % nbx, yb = b / nbx;
            for (uint32_t i = 0; i < bw * bh; ++i) {     // render.rs:126-133
                uint32_t x = i % bw + xb * bw, y = i / bw + yb * bh;
                size_t p = (size_t)y * w + x;
                V3 c = rt.render(x, y, w, h, aa, cn, hit_obj ? &hit_obj[p] : nullptr, hit_tri ? &hit_tri[p] : nullptr);
                rgb[p * 3 + 0] = quantize(c.x); rgb[p * 3 + 1] = quantize(c.y); rgb[p * 3 + 2] = quantize(c.z);
                if (rgb_f32) { rgb_f32[p * 3 + 0] = c.x; rgb_f32[p * 3 + 1] = c.y; rgb_f32[p * 3 + 2] = c.z; }
            }
        }
        cns[tid] = cn;
    };
    auto t0 = std::chrono::steady_clock::now();   // render.rs:96
    if (n_threads == 1) worker(0);
    else {
        std::vector<std::thread> th;
        for (uint32_t t = 0; t < n_threads; ++t) th.emplace_back(worker, t);
        for (auto& t : th) t.join();
    }
    auto t1 = std::chrono::steady_clock::now();   // render.rs:189
    if (out) {
        Counters tot;
        for (auto& c : cns) tot.add(c);
        out->raycasts = tot.raycasts; out->node_visits = tot.node_visits;
        out->tri_tests = tot.tri_tests; out->shade_calls = tot.shade_calls;
        out->seconds = std::chrono::duration<double>(t1 - t0).count();
    }
    return 0;
}

// --- unit-level entry points for the known-answer tests -----------------------------------------
int orc_triangle_intersect(const float tri9[9], const float o[3], const float d[3], float delta, float uvw[3]) {
    Triangle t{};
    t.a = v3(tri9[0], tri9[1], tri9[2]); t.b = v3(tri9[3], tri9[4], tri9[5]); t.c = v3(tri9[6], tri9[7], tri9[8]);
    Ray r{v3(o[0], o[1], o[2]), v3(d[0], d[1], d[2])};
    return triangle_intersect(t, r, delta, &uvw[0], &uvw[1], &uvw[2]) ? 1 : 0;
}
void orc_triangle_aabb(const float tri9[9], float box6[6]) {
    Triangle t{};
    t.a = v3(tri9[0], tri9[1], tri9[2]); t.b = v3(tri9[3], tri9[4], tri9[5]); t.c = v3(tri9[6], tri9[7], tri9[8]);
    Aabb b = triangle_aabb(t);
    box6[0] = b.lo.x; box6[1] = b.lo.y; box6[2] = b.lo.z; box6[3] = b.hi.x; box6[4] = b.hi.y; box6[5] = b.hi.z;
}
int orc_ray_intersects_aabb(const float o[3], const float d[3], const float box6[6]) {
    BvhRay r = bvh_ray_new(v3(o[0], o[1], o[2]), v3(d[0], d[1], d[2]));
    Aabb b{v3(box6[0], box6[1], box6[2]), v3(box6[3], box6[4], box6[5])};
    return ray_intersects_aabb(r, b) ? 1 : 0;
}
void orc_camera_ray(const orc_camera* cam, float x, float y, float o[3], float d[3]) {
    Scene s;
    s.camera.pos = v3(cam->pos[0], cam->pos[1], cam->pos[2]);
    s.camera.look_at = v3(cam->look_at[0], cam->look_at[1], cam->look_at[2]);
    s.camera.up = v3(cam->up[0], cam->up[1], cam->up[2]);
    s.camera.fov = cam->fov; s.camera.w = cam->w; s.camera.h = cam->h;
    Raytracer rt{s, 1e-5f, 5, ACCEL_BVH, SHADE_LITERAL};
    Ray r = rt.camera_ray(x, y);
    o[0] = r.o.x; o[1] = r.o.y; o[2] = r.o.z; d[0] = r.d.x; d[1] = r.d.y; d[2] = r.d.z;
}
// pixel -> NDC of raytracer/mod.rs:251-252
void orc_pixel_ndc(uint32_t x, uint32_t y, uint32_t w, uint32_t h, float out[2]) {
    out[0] = ((float)x / (float)w * 2.0f + -1.0f) * ((float)w / (float)h);
    out[1] = (float)y / (float)h * 2.0f + -1.0f;
}
uint8_t orc_quantize(float c) { return quantize(c); }
void orc_reflect(const float i[3], const float n[3], float out[3]) {
    V3 r = Raytracer::reflect(v3(i[0], i[1], i[2]), v3(n[0], n[1], n[2]));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
// Object::transform() and its inverse (column-major 4x4 each)
int orc_object_matrices(const float translation[3], const float rotation[4], const float scale3[3], float fwd[16], float inv[16]) {
    Object o;
    for (int k = 0; k < 3; ++k) { o.translation[k] = translation[k]; o.scl[k] = scale3[k]; }
    for (int k = 0; k < 4; ++k) o.rotation[k] = rotation[k];
    M4 f = object_transform(o), i;
    bool ok = inverse4(f, &i);
    std::memcpy(fwd, f.m, sizeof f.m);
    if (ok) std::memcpy(inv, i.m, sizeof i.m);
    return ok ? 1 : 0;
}
void orc_skybox(void* sp, const float d[3], float out[3]) {
    Scene* s = (Scene*)sp;
    Raytracer rt{*s, 1e-5f, 5, ACCEL_BVH, SHADE_LITERAL};
    V3 c = rt.skybox(v3(d[0], d[1], d[2]));
    out[0] = c.x; out[1] = c.y; out[2] = c.z;
}
float orc_rand_f32(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t dim) {
    return Raytracer::rand_f32(seed, pixel, sample, dim);
}
// single world-space raycast (raytracer/mod.rs:45-51); returns 1 on hit
int orc_raycast(void* sp, float delta, int accel, const float o[3], const float d[3], uint32_t* obj, uint32_t* tri,
                float point[3], float normal[3]) {
    Scene* s = (Scene*)sp;
    Raytracer rt{*s, delta, 5, accel, SHADE_LITERAL};
    Counters cn;
    Hit h;
    if (!rt.raycast(Ray{v3(o[0], o[1], o[2]), v3(d[0], d[1], d[2])}, &h, cn)) return 0;
    *obj = h.obj; *tri = h.tri;
    point[0] = h.point.x; point[1] = h.point.y; point[2] = h.point.z;
    normal[0] = h.normal.x; normal[1] = h.normal.y; normal[2] = h.normal.z;
    return 1;
}

}   // extern "C"
