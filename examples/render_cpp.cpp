// render_cpp -- the C++ host layer (include/trayracer.hpp) end to end, no Python and no PyTorch in the process:
// builds a small scene from exactly representable numbers, renders it through trayracer::Render (render thread,
// progress, cancel -- the handle of raytracer/render.rs:15-25), writes the frame as PPM and the scene as a flat binary
// dump.  tests/test_gpu_cpp_host.py reads the dump back, renders the same scene through the Python binding and the CPU
// oracle and compares the three frames.
//
//   usage: render_cpp <out.ppm> <scene.bin> [width height]
#include <cstdio>
#include <cstring>
#include <fstream>

#include "scene_dump.hpp"
#include "trayracer.hpp"

using namespace trayracer;

static Triangle tri(Vec3 a, Vec3 b, Vec3 c, Vec3 n, std::array<float, 2> ua, std::array<float, 2> ub, std::array<float, 2> uc,
                    std::optional<size_t> mat) {
    Triangle t;
    t.a = a; t.b = b; t.c = c;
    t.a_normal = t.b_normal = t.c_normal = n;
    t.a_uv = ua; t.b_uv = ub; t.c_uv = uc;
    t.material_index = mat;
    return t;
}

// axis-aligned quad p0 + s*u + t*v, two triangles
static void quad(std::vector<Triangle>& out, Vec3 p, Vec3 u, Vec3 v, Vec3 n, std::optional<size_t> mat) {
    Vec3 p1{p[0] + u[0], p[1] + u[1], p[2] + u[2]}, p3{p[0] + v[0], p[1] + v[1], p[2] + v[2]};
    Vec3 p2{p1[0] + v[0], p1[1] + v[1], p1[2] + v[2]};
    out.push_back(tri(p, p1, p2, n, {0, 0}, {1, 0}, {1, 1}, mat));
    out.push_back(tri(p, p2, p3, n, {0, 0}, {1, 1}, {0, 1}, mat));
}

static void box(std::vector<Triangle>& out, Vec3 lo, Vec3 hi, std::optional<size_t> mat) {
    float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    quad(out, {lo[0], hi[1], lo[2]}, {0, 0, dz}, {dx, 0, 0}, {0, 1, 0}, mat);    // top
    quad(out, {lo[0], lo[1], lo[2]}, {dx, 0, 0}, {0, 0, dz}, {0, -1, 0}, mat);   // bottom
    quad(out, {lo[0], lo[1], hi[2]}, {dx, 0, 0}, {0, dy, 0}, {0, 0, 1}, mat);    // front
    quad(out, {lo[0], lo[1], lo[2]}, {0, dy, 0}, {dx, 0, 0}, {0, 0, -1}, mat);   // back
    quad(out, {hi[0], lo[1], lo[2]}, {0, dy, 0}, {0, 0, dz}, {1, 0, 0}, mat);    // right
    quad(out, {lo[0], lo[1], lo[2]}, {0, 0, dz}, {0, dy, 0}, {-1, 0, 0}, mat);   // left
}

static Scene make_scene(uint32_t w, uint32_t h) {
    Scene sc;
    // floor: 16 x 16 mirror tiles (illum 3), translated like res/test/config.yaml places its models
    Object floor;
    floor.name = "Floor";
    Material mirror;
    mirror.name = "floor";
    mirror.diffuse_color = Color{0.0625f, 0.0625f, 0.0625f};
    mirror.specular_color = Color{0.5f, 0.5f, 0.5f};
    mirror.specular_exponent = 64.0f;
    mirror.illumination_model = *IlluminationModel::from_i32(3);
    mirror.dissolve = 0.0f;
    floor.materials.push_back(mirror);
    for (int j = 0; j < 16; ++j)
        for (int i = 0; i < 16; ++i)
            quad(floor.triangles, {-4.0f + 0.5f * i, 0.0f, -4.0f + 0.5f * j}, {0, 0, 0.5f}, {0.5f, 0, 0}, {0, 1, 0}, 0);
    floor.translation = {0.5f, 0.0f, -0.25f};
    sc.objects.push_back(floor);

    // props: a red specular box, a textured box, a glass pane (illum 6, dissolve 0.5), a chrome box without Ks/Ns/d and
    // one face without any material; rotated a quarter turn about y and scaled non-uniformly
    Object props;
    props.name = "Props";
    auto tex = std::make_shared<Image>();
    tex->width = 4; tex->height = 4;
    for (int y = 0; y < 4; ++y)
        for (int x = 0; x < 4; ++x) {
            bool on = ((x ^ y) & 1) != 0;
            tex->rgb.insert(tex->rgb.end(), {(uint8_t)(on ? 200 : 40), (uint8_t)(on ? 60 : 40), (uint8_t)(on ? 30 : 160)});
        }
    Material red; red.name = "red"; red.diffuse_color = Color{0.75f, 0.125f, 0.125f}; red.specular_color = Color{0.5f, 0.5f, 0.5f};
    red.specular_exponent = 32.0f; red.illumination_model = *IlluminationModel::from_i32(2); red.dissolve = 0.0f;
    Material checker; checker.name = "checker"; checker.diffuse_color = Color{0.25f, 0.25f, 0.25f}; checker.diffuse_texture = tex;
    checker.illumination_model = *IlluminationModel::from_i32(2); checker.specular_exponent = 8.0f;
    Material glass; glass.name = "glass"; glass.diffuse_color = Color{0.5f, 0.75f, 1.0f}; glass.dissolve = 0.5f;
    glass.illumination_model = *IlluminationModel::from_i32(6);
    Material chrome; chrome.name = "chrome"; chrome.diffuse_color = Color{0.125f, 0.125f, 0.1875f};
    chrome.illumination_model = *IlluminationModel::from_i32(4);
    props.materials = {red, checker, glass, chrome};
    box(props.triangles, {-0.5f, 0.0f, -0.5f}, {0.5f, 1.0f, 0.5f}, 0);
    box(props.triangles, {1.0f, 0.0f, -1.5f}, {1.75f, 0.75f, -0.75f}, 1);
    quad(props.triangles, {-1.5f, 0.0f, 1.25f}, {1.5f, 0, 0}, {0, 1.25f, 0}, {0, 0, 1}, 2);
    box(props.triangles, {-2.0f, 0.0f, -1.0f}, {-1.25f, 1.5f, -0.25f}, 3);
    quad(props.triangles, {0.75f, 0.0f, 0.75f}, {0.75f, 0, 0}, {0, 0.75f, 0}, {0, 0, 1}, std::nullopt);
    props.translation = {0.25f, 0.0f, 0.5f};
    props.rotation = {0.0f, 0.70710677f, 0.0f, 0.70710677f};
    props.scale = {1.0f, 1.25f, 0.75f};
    sc.objects.push_back(props);

    sc.lights.push_back(Light::from_ke({-2.0f, 3.5f, -0.75f}, {1, 1, 1}, 40.0f));
    sc.lights.push_back(Light::from_ke({5.5f, 0.25f, -4.5f}, {0.75f, 0.25f, 0.75f}, 22.0f));
    sc.lights.push_back(Light::from_ke({2.0f, 4.0f, 1.5f}, {1, 1, 1}, 23.0f));
    sc.camera.position = {3.5f, 2.25f, 5.0f};
    sc.camera.look_at = {0.0f, 0.5f, 0.0f};
    sc.camera.up = {0, 1, 0};
    sc.camera.fov = 1.0f;
    sc.camera.resolution = {w, h};
    // a small image sky: 8 x 4 gradient
    auto sky = std::make_shared<Image>();
    sky->width = 8; sky->height = 4;
    for (int y = 0; y < 4; ++y)
        for (int x = 0; x < 8; ++x) sky->rgb.insert(sky->rgb.end(), {(uint8_t)(30 + 20 * x), (uint8_t)(60 + 40 * y), (uint8_t)(220 - 15 * x)});
    sc.settings.skybox = std::shared_ptr<const Image>(sky);
    return sc;
}

int main(int argc, char** argv) {
    if (argc < 3) {
        std::fprintf(stderr, "usage: %s <out.ppm> <scene.bin> [width height]\n", argv[0]);
        return 2;
    }
    uint32_t w = argc > 4 ? (uint32_t)std::atoi(argv[3]) : 320, h = argc > 4 ? (uint32_t)std::atoi(argv[4]) : 240;
    try {
        Scene sc = make_scene(w, h);
        dump_scene(sc, argv[2]);
        auto ctx = std::make_shared<Context>(0);
        Render r(ctx);

        // 1. a cancelled render: the flag is already set when the thread starts (render.rs:113), so no block is rendered
        r.cancel = 1;
        r.render(sc);
        r.thread->join();   // (Render::join() would also clear the flag; the check below wants to see it first)
        bool cancelled_black = true;
        for (uint8_t v : r.image.rgb) cancelled_black = cancelled_black && v == 0;
        std::printf("cancelled render: finished=%d black=%d\n", (int)r.finished(), (int)cancelled_black);
        r.join();

        // 2. the real render, polled like the UI does (ui/mod.rs:93-100)
        r.render(sc);
        uint16_t last = 0;
        int polls = 0;
        size_t painted_last = 0;
        while (!r.finished()) {
            uint16_t p = r.progress.load();
            if (p < last) { std::fprintf(stderr, "progress went backwards\n"); return 1; }
            last = p;
            // the image fills block by block (render.rs:170-183): a snapshot never loses painted pixels
            Image snap = r.snapshot();
            size_t painted = 0;
            for (uint8_t v : snap.rgb) painted += v != 0;
            if (painted < painted_last) { std::fprintf(stderr, "painted pixels went backwards\n"); return 1; }
            painted_last = painted;
            ++polls;
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
        r.join();
        if (!r.error.empty()) { std::fprintf(stderr, "render failed: %s\n", r.error.c_str()); return 1; }
        std::printf("render: %ux%u in %u ms, progress=%u, polls=%d\n", r.image.width, r.image.height, r.time.load(), (unsigned)r.progress.load(), polls);

        // 3. the error path: a resolution below 20 px is the reference's division by zero (render.rs:109)
        Scene tiny = sc;
        tiny.camera.resolution = {16, 16};
        int code = 0;
        try {
            Raytracer rt(ctx, tiny);
            std::vector<uint8_t> px(16 * 16 * 3);
            rt.render_frame(px.data());
        } catch (const Error& e) { code = e.code; }
        std::printf("16x16 render: error code %d\n", code);

        std::ofstream f(argv[1], std::ios::binary);
        f << "P6\n" << r.image.width << " " << r.image.height << "\n255\n";
        f.write(reinterpret_cast<const char*>(r.image.rgb.data()), (std::streamsize)r.image.rgb.size());
        unsigned long long sum = 0;
        for (uint8_t v : r.image.rgb) sum += v;
        std::printf("checksum %llu\n", sum);
        return (cancelled_black && code == TRR_ERR_RESOLUTION) ? 0 : 1;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
}
