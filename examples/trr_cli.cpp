// trr_cli -- headless front end in C++: the reference's scene YAML (+ OBJ/MTL/PNG) in, PNG or PPM out.  No Python and
// no PyTorch in the process.  The reference has no CLI; this is its render button (ui/statusbar.rs:142-157 ->
// raytracer/render.rs:39-71) and its export button without the window:
//   Scene::load (include/trayracer_loader.hpp) -> Raytracer(scene, 1e-5, 5) (render.rs:105) -> RGB8 frame -> file.
//
//   usage: trr_cli <scene.yaml> [-o out.png|out.ppm] [--width W] [--height H] [--max-depth D] [--delta E]
//                  [--anti-aliasing on|off] [--samples N] [--seed S] [--device K | --devices 0,1,..] [--build host|device] [--dump-scene file] [--no-render] [--stats]
#include <cstdio>
#include <cstring>
#include <string>

#include "scene_dump.hpp"
#include "trayracer_loader.hpp"

using namespace trayracer;

// 8-bit RGB, non-interlaced, filter 0 on every row: lossless
static void write_png(const std::string& path, const Image& im) {
    std::vector<unsigned char> raw((size_t)(im.width * 3 + 1) * im.height);
    for (uint32_t y = 0; y < im.height; ++y) {
        raw[(size_t)(im.width * 3 + 1) * y] = 0;
        std::memcpy(&raw[(size_t)(im.width * 3 + 1) * y + 1], &im.rgb[(size_t)im.width * 3 * y], (size_t)im.width * 3);
    }
    uLongf clen = compressBound((uLong)raw.size());
    std::vector<unsigned char> comp(clen);
    if (compress2(comp.data(), &clen, raw.data(), (uLong)raw.size(), 6) != Z_OK) throw std::runtime_error("deflate failed");
    std::ofstream f(path, std::ios::binary);
    auto be = [](uint32_t v, unsigned char* p) { p[0] = (unsigned char)(v >> 24); p[1] = (unsigned char)(v >> 16); p[2] = (unsigned char)(v >> 8); p[3] = (unsigned char)v; };
    auto chunk = [&](const char* tag, const unsigned char* data, uint32_t len) {
        unsigned char b[4];
        be(len, b);
        f.write(reinterpret_cast<const char*>(b), 4);
        f.write(tag, 4);
        if (len) f.write(reinterpret_cast<const char*>(data), len);
        uLong crc = crc32(0L, reinterpret_cast<const Bytef*>(tag), 4);
        if (len) crc = crc32(crc, data, len);
        be((uint32_t)crc, b);
        f.write(reinterpret_cast<const char*>(b), 4);
    };
    f.write("\x89PNG\r\n\x1a\n", 8);
    unsigned char ihdr[13];
    be(im.width, ihdr);
    be(im.height, ihdr + 4);
    ihdr[8] = 8; ihdr[9] = 2; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;
    chunk("IHDR", ihdr, 13);
    chunk("IDAT", comp.data(), (uint32_t)clen);
    chunk("IEND", nullptr, 0);
}

int main(int argc, char** argv) {
    // trr_cli --decode-image <in> <out.ppm>: what image::open(in)?.into_rgb8() (object.rs:31-38) yields, as a binary PPM
    if (argc == 4 && std::string(argv[1]) == "--decode-image") {
        try {
            const Image im = loader::load_image_rgb8(argv[2]);
            FILE* f = std::fopen(argv[3], "wb");
            if (!f) { std::fprintf(stderr, "error: cannot write %s\n", argv[3]); return 2; }
            std::fprintf(f, "P6\n%u %u\n255\n", im.width, im.height);
            std::fwrite(im.rgb.data(), 1, im.rgb.size(), f);
            std::fclose(f);
            return 0;
        } catch (const std::exception& e) {   // LoadError, and anything else a hostile file can raise
            std::fprintf(stderr, "error: %s\n", e.what());
            return 2;
        }
    }
    std::string scene_path, out = "render.png", dump;
    int device = 0;
    std::vector<int> devices;
    long width = 0, height = 0, samples = -1;
    uint32_t max_depth = 5;
    float delta = 1e-5f;
    int aa = -1;
    long long seed = -1;
    bool no_render = false, stats = false;
    Build build = Build::Host;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto val = [&]() -> std::string {
            if (i + 1 >= argc) { std::fprintf(stderr, "error: %s needs a value\n", a.c_str()); std::exit(2); }
            return argv[++i];
        };
        if (a == "-o" || a == "--output") out = val();
        else if (a == "--width") width = std::atol(val().c_str());
        else if (a == "--height") height = std::atol(val().c_str());
        else if (a == "--max-depth") max_depth = (uint32_t)std::atol(val().c_str());
        else if (a == "--delta") delta = std::strtof(val().c_str(), nullptr);
        else if (a == "--anti-aliasing") aa = val() == "on" ? 1 : 0;
        else if (a == "--samples") samples = std::atol(val().c_str());
        else if (a == "--seed") seed = std::atoll(val().c_str());
        else if (a == "--device") device = std::atoi(val().c_str());
        else if (a == "--devices") {   // "0,1,2,3": one context over several GPUs (trr_create_multi)
            std::string v = val();
            size_t pos = 0;
            while (pos <= v.size()) {
                size_t c = v.find(',', pos);
                if (c == std::string::npos) c = v.size();
                if (c > pos) devices.push_back(std::atoi(v.substr(pos, c - pos).c_str()));
                pos = c + 1;
            }
        }
        else if (a == "--build") {
            std::string b = val();
            if (b != "host" && b != "device") { std::fprintf(stderr, "error: --build takes host or device\n"); return 2; }
            build = b == "device" ? Build::Device : Build::Host;
        }
        else if (a == "--dump-scene") dump = val();
        else if (a == "--no-render") no_render = true;
        else if (a == "--stats") stats = true;
        else if (!a.empty() && a[0] == '-') { std::fprintf(stderr, "error: unknown option %s\n", a.c_str()); return 2; }
        else scene_path = a;
    }
    if (scene_path.empty()) {
        std::fprintf(stderr, "usage: %s <scene.yaml> [-o out.png|out.ppm] [--width W] [--height H] [--max-depth D] [--anti-aliasing on|off] "
                             "[--samples N] [--seed S] [--device K | --devices 0,1,..] [--build host|device] [--dump-scene file] [--no-render] [--stats]\n", argv[0]);
        return 2;
    }
    Scene sc;
    try {
        sc = loader::load_scene(scene_path);
    } catch (const loader::LoadError& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 2;
    }
    if (width > 0) sc.camera.resolution.first = (uint32_t)width;
    if (height > 0) sc.camera.resolution.second = (uint32_t)height;
    if (aa >= 0) sc.settings.anti_aliasing = aa != 0;
    if (samples >= 0) sc.settings.samples = (uint32_t)samples;
    if (seed >= 0) sc.settings.seed = (uint64_t)seed;
    if (!dump.empty()) dump_scene(sc, dump.c_str());
    if (no_render) return 0;
    try {
        // throws when no HIP device exists: there is no CPU fallback
        auto ctx = devices.empty() ? std::make_shared<Context>(device) : std::make_shared<Context>(devices);
        Raytracer rt(ctx, sc, delta, max_depth, build);
        Image im;
        im.width = rt.width(); im.height = rt.height();
        im.rgb.assign((size_t)im.width * im.height * 3, 0);
        trr_stats st{};
        rt.render_frame(im.rgb.data(), nullptr, nullptr, &st);
        if (out.size() > 4 && out.substr(out.size() - 4) == ".ppm") {
            std::ofstream f(out, std::ios::binary);
            f << "P6\n" << im.width << " " << im.height << "\n255\n";
            f.write(reinterpret_cast<const char*>(im.rgb.data()), (std::streamsize)im.rgb.size());
        } else write_png(out, im);
        if (stats) {
            unsigned long long rays = st.rays_primary + st.rays_shadow + st.rays_reflection + st.rays_march;
            size_t tris = 0;
            for (auto& o : sc.objects) tris += o.triangles.size();
            trr_scene_info info = rt.info();
            std::printf("{\"scene\": \"%s\", \"triangles\": %zu, \"resolution\": [%u, %u], \"rays\": %llu, \"ms_render\": %.3f, "
                        "\"build\": \"%s\", \"ms_build\": %.3f, \"bvh_nodes\": %llu, \"bvh_depth\": %u, \"output\": \"%s\"}\n",
                        scene_path.c_str(), tris, im.width, im.height, rays, st.ms_total, info.build == TRR_BUILD_DEVICE ? "device" : "host",
                        info.ms_build, (unsigned long long)info.n_nodes, info.bvh_depth, out.c_str());
        }
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
