// scene_dump.hpp -- flat little-endian dump of a trayracer::Scene ("TRRSCN01"), read back by tests/test_gpu_cpp_host.py and
// tests/test_loader_cpp.py to compare what the C++ host layer holds with what the Python host layer holds.
#pragma once
#include <fstream>

#include "trayracer.hpp"

namespace trayracer {

template <typename T> inline void put(std::ofstream& f, const T& v) { f.write(reinterpret_cast<const char*>(&v), sizeof v); }
inline void put_image(std::ofstream& f, const Image* im) {
    put<uint32_t>(f, im ? im->width : 0);
    put<uint32_t>(f, im ? im->height : 0);
    if (im) f.write(reinterpret_cast<const char*>(im->rgb.data()), (std::streamsize)im->rgb.size());
}
inline void dump_scene(const Scene& sc, const char* path) {
    std::ofstream f(path, std::ios::binary);
    f.write("TRRSCN01", 8);
    put<uint32_t>(f, (uint32_t)sc.objects.size());
    for (const Object& o : sc.objects) {
        put<uint32_t>(f, (uint32_t)o.triangles.size());
        for (const Triangle& t : o.triangles) {
            for (const Vec3* v : {&t.a, &t.b, &t.c, &t.a_normal, &t.b_normal, &t.c_normal}) f.write(reinterpret_cast<const char*>(v->data()), 12);
            for (const auto* v : {&t.a_uv, &t.b_uv, &t.c_uv}) f.write(reinterpret_cast<const char*>(v->data()), 8);
            put<int32_t>(f, t.material_index ? (int32_t)*t.material_index : -1);
        }
        put<uint32_t>(f, (uint32_t)o.materials.size());
        for (const Material& m : o.materials) {
            uint32_t flags = (m.diffuse_color ? 1u : 0u) | (m.specular_color ? 2u : 0u) | (m.specular_exponent ? 4u : 0u) | (m.dissolve ? 8u : 0u);
            put<uint32_t>(f, flags);
            Color kd = m.diffuse_color.value_or(Color{0, 0, 0}), ks = m.specular_color.value_or(Color{0, 0, 0});
            f.write(reinterpret_cast<const char*>(kd.data()), 12);
            f.write(reinterpret_cast<const char*>(ks.data()), 12);
            put<float>(f, m.specular_exponent.value_or(0.0f));
            put<float>(f, m.dissolve.value_or(0.0f));
            put<int32_t>(f, m.illumination_model.value);
            put_image(f, m.diffuse_texture.get());
        }
        f.write(reinterpret_cast<const char*>(o.translation.data()), 12);
        f.write(reinterpret_cast<const char*>(o.rotation.data()), 16);
        f.write(reinterpret_cast<const char*>(o.scale.data()), 12);
    }
    put<uint32_t>(f, (uint32_t)sc.lights.size());
    for (const Light& l : sc.lights) {
        f.write(reinterpret_cast<const char*>(l.position.data()), 12);
        f.write(reinterpret_cast<const char*>(l.color.data()), 12);
        put<float>(f, l.intensity);
    }
    const Settings& s = sc.settings;
    put<uint32_t>(f, s.samples);
    put<uint32_t>(f, s.anti_aliasing ? 1u : 0u);
    put<uint64_t>(f, s.seed);
    f.write(reinterpret_cast<const char*>(s.ambient_color.data()), 12);
    put<float>(f, s.ambient_intensity);
    if (auto img = std::get_if<std::shared_ptr<const Image>>(&s.skybox)) {
        put<uint32_t>(f, 1);
        put_image(f, img->get());
    } else {
        put<uint32_t>(f, 0);
        f.write(reinterpret_cast<const char*>(std::get<Color>(s.skybox).data()), 12);
    }
    const Camera& c = sc.camera;
    f.write(reinterpret_cast<const char*>(c.position.data()), 12);
    f.write(reinterpret_cast<const char*>(c.look_at.data()), 12);
    f.write(reinterpret_cast<const char*>(c.up.data()), 12);
    put<float>(f, c.fov);
    put<uint32_t>(f, c.resolution.first);
    put<uint32_t>(f, c.resolution.second);
}


}   // namespace trayracer
