// trayracer.hpp -- C++17 host-side mirror of the reference's scene / raytracer / render interface on top of the C ABI
// of trayracer_hip.h.  Header-only; link with libtrayracer_hip (.a or .so) and libamdhip64.
//
// The reference is a Rust program; with no Rust toolchain in the build image this is the compiled-language host layer a
// maintainer would otherwise write in Rust (INTEGRATION.md shows those lines).  Names and meanings follow the reference:
//
//   reference                                                     here
//   ------------------------------------------------------------  -----------------------------------------------
//   scene::{Material, IlluminationModel}   scene/material.rs:5-53   trayracer::Material, IlluminationModel
//   scene::Triangle                        scene/triangle.rs:9-21   trayracer::Triangle
//   scene::Object  (+ transform())         scene/object.rs:19-29    trayracer::Object
//   scene::Light / Camera / Settings / Skybox / Scene               trayracer::Light / Camera / Settings / Skybox / Scene
//   Raytracer::new(scene, delta, max_depth)  raytracer/mod.rs:37    trayracer::Raytracer(scene, delta, max_depth)
//   block loop of RenderingThread::run       render.rs:103-152      Raytracer::render_frame(...)
//   Render{progress,thread,cancel,image,time} render.rs:15-25      trayracer::Render (same fields, same meaning)
//
// Value conventions applied by the reference at LOAD time and expected here: Light::color is the normalised Ke
// (light.rs:33), Settings::ambient_color is normalised with ambient_intensity = its norm (settings.rs:51-55),
// Material::dissolve is 1 - d (object.rs:106), Camera::fov is in radians (camera.rs:75).
// Errors: every failing C call throws trayracer::Error (code = trr_status, what() = trr_last_error()); a resolution
// below 20 px, where the reference panics with a division by zero (render.rs:109), throws with code TRR_ERR_RESOLUTION.
#pragma once

#include <array>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <optional>
#include <stdexcept>
#include <string>
#include <thread>
#include <utility>
#include <variant>
#include <vector>

#include "trayracer_hip.h"

namespace trayracer {

using Color = std::array<float, 3>;   // scene/mod.rs:30
using Vec3 = std::array<float, 3>;

class Error : public std::runtime_error {
public:
    Error(int code_, const std::string& msg) : std::runtime_error("trayracer_hip error " + std::to_string(code_) + ": " + msg), code(code_) {}
    int code;
};

// image::RgbImage after into_rgb8(): row-major 8-bit RGB
struct Image {
    uint32_t width = 0, height = 0;
    std::vector<uint8_t> rgb;
};

// scene/material.rs:31-53
struct IlluminationModel {
    int value = 0;
    static std::optional<IlluminationModel> from_i32(int i) {
        if (i >= 0 && i <= 10) return IlluminationModel{i};
        return std::nullopt;
    }
    bool specular() const { return value == 2; }
    bool reflection() const { return value == 3 || value == 4; }
    bool transparency() const { return value == 6 || value == 7; }
};

// scene/material.rs:5-15; std::nullopt / nullptr are the reference's Option::None
struct Material {
    std::string name;
    std::optional<Color> diffuse_color;
    std::optional<Color> specular_color;
    std::optional<float> specular_exponent;
    std::shared_ptr<const Image> diffuse_texture;
    IlluminationModel illumination_model;
    std::optional<float> dissolve;
    std::optional<float> refraction_index;   // unused by the reference too
};

// scene/triangle.rs:9-21, fields in declaration order; the first 24 floats are what the C ABI takes per triangle
struct Triangle {
    Vec3 a{}, b{}, c{};
    Vec3 a_normal{}, b_normal{}, c_normal{};
    std::array<float, 2> a_uv{}, b_uv{}, c_uv{};
    std::optional<size_t> material_index;
};

// scene/object.rs:19-29; rotation is the unit quaternion (i, j, k, w)
struct Object {
    std::string name;
    std::vector<Triangle> triangles;
    std::vector<Material> materials;
    Vec3 translation{0, 0, 0};
    std::array<float, 4> rotation{0, 0, 0, 1};
    Vec3 scale{1, 1, 1};

    // Object::transform() and its inverse (object.rs:177-182) as column-major 4x4; false when not invertible
    bool transform(float fwd[16], float inv[16]) const {
        return trr_object_matrices(translation.data(), rotation.data(), scale.data(), fwd, inv) != 0;
    }
};

// scene/light.rs:5-9
struct Light {
    Vec3 position{};
    Color color{};
    float intensity = 0.0f;
    // LightDef -> Light (light.rs:26-37): colour = Ke.try_normalize(0.0).unwrap_or_default()
    static Light from_ke(Vec3 position, Color ke, float intensity) {
        float n = std::sqrt((ke[0] * ke[0] + ke[1] * ke[1]) + ke[2] * ke[2]);
        Color c{0, 0, 0};
        if (n > 0.0f) c = {ke[0] / n, ke[1] / n, ke[2] / n};
        return Light{position, c, intensity};
    }
};

// scene/skybox.rs:6-15
using Skybox = std::variant<Color, std::shared_ptr<const Image>>;

// scene/settings.rs:4-25; `seed` feeds the counter RNG that stands in for rand::random (raytracer/mod.rs:234,238)
struct Settings {
    static Color default_ambient() {
        const float x = 0.34f, y = 0.14f, z = 0.04f;
        const float n = std::sqrt((x * x + y * y) + z * z);
        return Color{x / n, y / n, z / n};
    }
    uint32_t max_bounces = 4;   // unused by the reference as well
    uint32_t samples = 1;
    Color ambient_color = default_ambient();   // Color::new(0.34, 0.14, 0.04).normalize(), settings.rs:19
    float ambient_intensity = 0.2f;
    Skybox skybox = Color{0.16f, 0.16f, 0.16f};
    bool anti_aliasing = false;
    uint64_t seed = 1;
};

// scene/camera.rs:5-23
struct Camera {
    Vec3 position{1, 1, 1};
    Vec3 look_at{0, 0, 0};
    Vec3 up{0, 1, 0};
    float fov = 1.0471976f;   // 60 degrees
    std::pair<uint32_t, uint32_t> resolution{1920, 1080};

    trr_camera desc() const {
        trr_camera c{};
        for (int k = 0; k < 3; ++k) {
            c.position[k] = position[k];
            c.look_at[k] = look_at[k];
            c.up[k] = up[k];
        }
        c.fov = fov;
        c.width = resolution.first;
        c.height = resolution.second;
        return c;
    }
    // Camera::ray (camera.rs:30-42) for normalised device coordinates
    std::pair<Vec3, Vec3> ray(float x, float y) const {
        trr_camera c = desc();
        Vec3 o{}, d{};
        trr_camera_ray(&c, x, y, o.data(), d.data());
        return {o, d};
    }
};

// scene/mod.rs:32-42
struct Scene {
    std::vector<Object> objects;
    std::vector<Light> lights;
    Camera camera;
    Settings settings;
};

// One GPU of this process (trr_ctx).  The reference has no counterpart; a process makes one and keeps it.
class Context {
public:
    explicit Context(int device = 0) {
        int rc = trr_create(device, &ctx_);
        if (rc != TRR_OK) throw Error(rc, trr_last_error(nullptr));
    }
    // several GPUs behind one context (trr_create_multi): scenes are replicated, whole-frame renders are sharded over the
    // devices inside the library (block b -> device b % n, one RCCL gather, assembled on the first device)
    explicit Context(const std::vector<int>& devices) {
        int rc = trr_create_multi(devices.data(), (int)devices.size(), &ctx_);
        if (rc != TRR_OK) throw Error(rc, trr_last_error(nullptr));
    }
    ~Context() { trr_destroy(ctx_); }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    trr_ctx* get() const { return ctx_; }
    void check(int rc) const {
        if (rc != TRR_OK) throw Error(rc, trr_last_error(ctx_));
    }

private:
    trr_ctx* ctx_ = nullptr;
};

// Raytracer::new(scene, delta, max_depth) (raytracer/mod.rs:37): uploads a flattened copy of the scene (the reference
// deep-clones it, render.rs:60) and renders it with the gfx950 kernels.  One render at a time per Context.
// Where the BVH that stands in for bvh::Bvh::build (object.rs:156) is built: Host = binned SAH on the CPU cores (best
// tree), Device = Morton-order tree with SAH-rebuilt treelets on the GPU (an order of magnitude faster to build, frames within a
// few per cent of the host tree's; for scenes edited between renders).
enum class Build : uint32_t { Host = TRR_BUILD_HOST, Device = TRR_BUILD_DEVICE };

class Raytracer {
public:
    Raytracer(std::shared_ptr<Context> ctx, const Scene& scene, float delta = 1e-5f, uint32_t max_depth = 5, Build build = Build::Host)
        : ctx_(std::move(ctx)), delta_(delta), max_depth_(max_depth), build_(build) {
        upload(scene);
    }
    ~Raytracer() { trr_scene_destroy(scene_); }
    Raytracer(const Raytracer&) = delete;
    Raytracer& operator=(const Raytracer&) = delete;

    uint32_t width() const { return width_; }
    uint32_t height() const { return height_; }
    trr_scene_info info() const {
        trr_scene_info i{};
        ctx_->check(trr_scene_get_info(scene_, &i));
        return i;
    }
    // host-side check of the uploaded tree (trr_scene_validate); throws Error with the first violation
    void validate() { ctx_->check(trr_scene_validate(scene_)); }

    // ui/properties.rs edits between renders: lights, settings, camera (cheap, no BVH rebuild)
    void set_lights(const std::vector<Light>& lights) {
        std::vector<trr_light> l(lights.size());
        for (size_t i = 0; i < lights.size(); ++i) {
            for (int k = 0; k < 3; ++k) {
                l[i].position[k] = lights[i].position[k];
                l[i].color[k] = lights[i].color[k];
            }
            l[i].intensity = lights[i].intensity;
        }
        ctx_->check(trr_scene_set_lights(scene_, l.data(), (uint32_t)l.size()));
    }
    void set_settings(const Settings& s) {
        trr_settings st{};
        for (int k = 0; k < 3; ++k) st.ambient_color[k] = s.ambient_color[k];
        st.ambient_intensity = s.ambient_intensity;
        if (auto img = std::get_if<std::shared_ptr<const Image>>(&s.skybox)) {
            if (!*img || (*img)->rgb.size() != (size_t)(*img)->width * (*img)->height * 3) throw Error(TRR_ERR_INVALID, "skybox image is empty or has the wrong size");
            st.skybox_is_image = 1;
            st.skybox_image = trr_image_rgb8{(*img)->rgb.data(), (*img)->width, (*img)->height};
        } else {
            const Color& c = std::get<Color>(s.skybox);
            for (int k = 0; k < 3; ++k) st.skybox_color[k] = c[k];
        }
        st.anti_aliasing = s.anti_aliasing ? 1 : 0;
        st.samples = s.samples;
        st.seed = s.seed;
        ctx_->check(trr_scene_set_settings(scene_, &st));
    }
    void set_camera(const Camera& cam) {
        trr_camera c = cam.desc();
        ctx_->check(trr_scene_set_camera(scene_, &c));
        width_ = c.width;
        height_ = c.height;
    }

    // The block loop of RenderingThread::run (render.rs:103-152) for the whole frame: row-major RGB8, remainder strip
    // black.  `cancel` (nullable) is polled between block batches like render.rs:113; `progress(done, total)` is called
    // per batch like render.rs:143-149.  Returns false when cancelled (blocks rendered so far are valid).
    bool render_frame(uint8_t* rgb_out, const std::atomic<int32_t>* cancel = nullptr,
                      const std::function<void(uint32_t, uint32_t)>* progress = nullptr, trr_stats* stats = nullptr,
                      uint32_t* hit_obj = nullptr, uint32_t* hit_tri = nullptr) {
        static_assert(sizeof(std::atomic<int32_t>) == sizeof(int32_t), "atomic<int32_t> must be a plain int32_t");
        auto thunk = [](void* user, uint32_t done, uint32_t total) {
            (*static_cast<const std::function<void(uint32_t, uint32_t)>*>(user))(done, total);
        };
        int rc = trr_render(ctx_->get(), scene_, delta_, max_depth_, 0, rgb_out, hit_obj, hit_tri,
                            reinterpret_cast<const volatile int32_t*>(cancel), progress ? +thunk : nullptr,
                            const_cast<void*>(static_cast<const void*>(progress)), stats);
        if (rc == TRR_ERR_CANCELLED) return false;
        ctx_->check(rc);
        return true;
    }

    // Progressive form (the sink of render.rs:155-186): sink(x_block, y_block, block_w, block_h, rgb) once per finished
    // block, batches of blocks_per_batch blocks (0 = one row of the 20x20 grid) in row-major block order; rgb is valid
    // only during the call.  Returns false when cancelled.
    using BlockSink = std::function<void(uint32_t, uint32_t, uint32_t, uint32_t, const uint8_t*)>;
    bool render_blocks(const BlockSink& sink, uint32_t blocks_per_batch = 0, const std::atomic<int32_t>* cancel = nullptr,
                       const std::function<void(uint32_t, uint32_t)>* progress = nullptr, trr_stats* stats = nullptr) {
        struct User { const BlockSink* sink; const std::function<void(uint32_t, uint32_t)>* progress; } u{&sink, progress};
        auto on_block = [](void* user, uint32_t xb, uint32_t yb, uint32_t bw, uint32_t bh, const uint8_t* rgb) {
            (*static_cast<User*>(user)->sink)(xb, yb, bw, bh, rgb);
        };
        auto on_progress = [](void* user, uint32_t done, uint32_t total) { (*static_cast<User*>(user)->progress)(done, total); };
        int rc = trr_render_blocks(ctx_->get(), scene_, delta_, max_depth_, 0, blocks_per_batch,
                                   reinterpret_cast<const volatile int32_t*>(cancel), +on_block, progress ? +on_progress : nullptr, &u, stats);
        if (rc == TRR_ERR_CANCELLED) return false;
        ctx_->check(rc);
        return true;
    }

private:
    void upload(const Scene& scene) {
        std::vector<trr_object_desc> objs(scene.objects.size());
        std::vector<std::vector<float>> tri(scene.objects.size());
        std::vector<std::vector<int32_t>> mi(scene.objects.size());
        std::vector<std::vector<trr_material_desc>> mats(scene.objects.size());
        std::vector<trr_image_rgb8> textures;
        for (size_t oi = 0; oi < scene.objects.size(); ++oi) {
            const Object& o = scene.objects[oi];
            tri[oi].resize(o.triangles.size() * 24);
            mi[oi].resize(o.triangles.size());
            for (size_t t = 0; t < o.triangles.size(); ++t) {
                const Triangle& T = o.triangles[t];
                float* d = &tri[oi][t * 24];
                const Vec3* v3[6] = {&T.a, &T.b, &T.c, &T.a_normal, &T.b_normal, &T.c_normal};
                for (int k = 0; k < 6; ++k)
                    for (int j = 0; j < 3; ++j) d[3 * k + j] = (*v3[k])[j];
                d[18] = T.a_uv[0]; d[19] = T.a_uv[1]; d[20] = T.b_uv[0]; d[21] = T.b_uv[1]; d[22] = T.c_uv[0]; d[23] = T.c_uv[1];
                if (T.material_index && *T.material_index >= o.materials.size()) throw Error(TRR_ERR_INVALID, "material_index out of range");
                mi[oi][t] = T.material_index ? (int32_t)*T.material_index : -1;
            }
            mats[oi].resize(o.materials.size());
            for (size_t k = 0; k < o.materials.size(); ++k) {
                const Material& m = o.materials[k];
                trr_material_desc& d = mats[oi][k];
                d = trr_material_desc{};
                if (m.diffuse_color) { d.flags |= TRR_MAT_HAS_KD; for (int j = 0; j < 3; ++j) d.kd[j] = (*m.diffuse_color)[j]; }
                if (m.specular_color) { d.flags |= TRR_MAT_HAS_KS; for (int j = 0; j < 3; ++j) d.ks[j] = (*m.specular_color)[j]; }
                if (m.specular_exponent) { d.flags |= TRR_MAT_HAS_NS; d.ns = *m.specular_exponent; }
                if (m.dissolve) { d.flags |= TRR_MAT_HAS_DISSOLVE; d.dissolve = *m.dissolve; }
                d.illum = m.illumination_model.value;
                d.texture = -1;
                if (m.diffuse_texture) {
                    const Image& im = *m.diffuse_texture;
                    if (im.rgb.size() != (size_t)im.width * im.height * 3) throw Error(TRR_ERR_INVALID, "texture has the wrong size");
                    d.texture = (int32_t)textures.size();
                    textures.push_back(trr_image_rgb8{im.rgb.data(), im.width, im.height});
                }
            }
            trr_object_desc& od = objs[oi];
            od.triangles = tri[oi].data();
            od.material_index = mi[oi].data();
            od.n_triangles = (uint32_t)o.triangles.size();
            od.materials = mats[oi].data();
            od.n_materials = (uint32_t)o.materials.size();
            o.transform(od.transform, od.inverse);
        }
        ctx_->check(trr_scene_create_ex(ctx_->get(), objs.data(), (uint32_t)objs.size(), textures.data(), (uint32_t)textures.size(),
                                        (uint32_t)build_, &scene_));
        set_lights(scene.lights);
        set_settings(scene.settings);
        set_camera(scene.camera);
    }

    std::shared_ptr<Context> ctx_;
    trr_scene* scene_ = nullptr;
    float delta_;
    uint32_t max_depth_;
    Build build_ = Build::Host;
    uint32_t width_ = 0, height_ = 0;
};

// raytracer/render.rs:15-25: the handle the UI holds.  progress in [0, 65535] (finished <=> 65535, render.rs:27-29),
// cancel is set by the UI thread, image is the finished (or partial) frame, time is in milliseconds (render.rs:188-190).
class Render {
public:
    explicit Render(std::shared_ptr<Context> ctx) : ctx_(std::move(ctx)) {}
    ~Render() { join(); }

    std::atomic<uint16_t> progress{0};
    std::atomic<int32_t> cancel{0};
    std::optional<std::thread> thread;
    Image image;               // Arc<Mutex<RgbImage>> in the reference: hold image_mutex (or use snapshot()) while a render runs
    std::mutex image_mutex;
    std::atomic<uint32_t> time{0};
    std::string error;   // the reference's render thread has no error channel; this one keeps the message

    bool finished() const { return progress.load() == UINT16_MAX; }
    // a copy of the (possibly partial) frame, safe while the render thread is painting blocks into `image`
    Image snapshot() {
        std::lock_guard<std::mutex> lock(image_mutex);
        return image;
    }

    // Render::render (render.rs:39-71): resets the handle and spawns the render thread on a copy of the scene
    // image fills block by block while the thread runs (the reference's sink paints every finished block into it,
    // render.rs:170-183); blocks_per_batch = blocks that appear together (0 = one row of blocks)
    void render(Scene scene, float delta = 1e-5f, uint32_t max_depth = 5, uint32_t blocks_per_batch = 0) {
        if (thread) throw std::logic_error("a render is already in flight (ui/statusbar.rs:142-157 disables the button)");
        {
            std::lock_guard<std::mutex> lock(image_mutex);
            image.width = scene.camera.resolution.first;
            image.height = scene.camera.resolution.second;
            image.rgb.assign((size_t)image.width * image.height * 3, 0);   // render.rs:51
        }
        progress = 0;                                                  // render.rs:54
        time = 0;                                                      // render.rs:55
        error.clear();
        thread.emplace([this, scene = std::move(scene), delta, max_depth, blocks_per_batch] {
            auto start = std::chrono::steady_clock::now();             // render.rs:96
            try {
                Raytracer rt(ctx_, scene, delta, max_depth);
                std::function<void(uint32_t, uint32_t)> on_progress = [this](uint32_t done, uint32_t total) {   // render.rs:143-149
                    progress = (uint16_t)std::lround(65535.0 * done / std::max(1u, total));
                };
                Raytracer::BlockSink sink = [this](uint32_t xb, uint32_t yb, uint32_t bw, uint32_t bh, const uint8_t* rgb) {
                    std::lock_guard<std::mutex> lock(image_mutex);     // `let mut image = self.image.lock()`, render.rs:170
                    for (uint32_t y = 0; y < bh; ++y)                  // render.rs:170-183
                        std::memcpy(&image.rgb[((size_t)(yb * bh + y) * image.width + (size_t)xb * bw) * 3], rgb + (size_t)y * bw * 3, (size_t)bw * 3);
                };
                rt.render_blocks(sink, blocks_per_batch, &cancel, &on_progress);
            } catch (const std::exception& e) {
                error = e.what();
            }
            time = (uint32_t)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - start).count();
            progress = UINT16_MAX;                                     // render.rs:188
        });
    }
    // ui/mod.rs:93-100: the UI joins the finished thread and clears the cancel flag
    void join() {
        if (thread) {
            if (thread->joinable()) thread->join();
            thread.reset();
            cancel = 0;
        }
    }

private:
    std::shared_ptr<Context> ctx_;
};

}   // namespace trayracer
