// trayracer_loader.hpp -- scene ingestion for the C++ host layer: the reference's scene YAML + Wavefront OBJ/MTL (+ PNG /
// PPM images) -> trayracer::Scene, in the reference's triangle order.  Header-only C++17; needs zlib for PNG.
//
// Restates what the reference does before Raytracer::new sees a scene:
//   Scene::load                       src/scene/mod.rs:64-140   (models / pointLights / camera / extraArgs)
//   ObjectDef + euler degrees         src/scene/object.rs:314-357
//   Object::from_obj                  src/scene/object.rs:61-175 (all mtllibs' materials in order, triangles in
//                                     objects -> groups -> polygons order, material index = FIRST position of the name)
//   triangulate                       src/scene/object.rs:224-296 (fan (0,i,i+1), computed normal (a-b)x(a-c))
//   LightDef / SettingsDef / CameraDef / SkyboxDef   light.rs:17-37, settings.rs:34-60, camera.rs:50-79, skybox.rs:22-41
// The same rules live in raytracing_amd/loader.py (Python host); tests/test_loader_cpp.py checks that both produce the
// same scene, bit for bit, from the same files.
//
// The OBJ/MTL grammar is the un-vendored `obj` 0.10 crate (Cargo.toml:63), restated as recalled from its published source
// ([crate] below) -- PARITY UNPINNED for grammar corner cases, like the Python loader.  The YAML reader covers the subset
// the reference's serializer emits and people write by hand for this schema: block and flow mappings and sequences,
// plain / quoted scalars, comments, the tags !Color / !Path; no anchors, no multi-line scalars.
// Images: 8-bit non-interlaced PNG (all colour types) and binary PPM decode here, TIFF (strips, 8-bit; none / LZW /
// PackBits) and baseline JPEG in trayracer_image.hpp -- the formats of the reference's own test textures
// (res/test/textures/) -- and OpenEXR (scan-line and one-level tiled; NONE / RLE / ZIPS / ZIP / PIZ / PXR24), the format
// of the reference's skyboxes, PARITY UNPINNED: those files are git-LFS stubs, so the decoder is checked against our own
// encoder only (tests/test_exr_decode.py).  A texture that fails to load only warns (object.rs:92-96); a skybox that fails makes extraArgs fall back to
// the defaults (mod.rs:100-110).
#pragma once

#include <zlib.h>

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <map>
#include <sstream>

#include "trayracer.hpp"
#include "trayracer_image.hpp"

namespace trayracer {
namespace loader {

class LoadError : public std::runtime_error {
public:
    explicit LoadError(const std::string& m) : std::runtime_error(m) {}
};

// where warnings go (the reference uses log::warn!); replaceable by the host
inline std::function<void(const std::string&)>& warn_sink() {
    static std::function<void(const std::string&)> f = [](const std::string& m) { std::fprintf(stderr, "warning: %s\n", m.c_str()); };
    return f;
}
inline void warn(const std::string& m) { warn_sink()(m); }

inline std::string read_file(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw LoadError("cannot open " + path);
    std::ostringstream ss;
    ss << f.rdbuf();
    return ss.str();
}
inline std::string parent_dir(const std::string& path) {
    size_t p = path.find_last_of('/');
    return p == std::string::npos ? std::string() : path.substr(0, p);
}
inline std::string join_path(const std::string& dir, const std::string& rel) {
    if (!rel.empty() && rel[0] == '/') return rel;
    return dir.empty() ? rel : dir + "/" + rel;
}

// ---------------------------------------------------------------------------------------------------------------
// images
// ---------------------------------------------------------------------------------------------------------------
inline uint32_t be32(const unsigned char* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

inline Image decode_png(const std::string& d) {
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    if (d.size() < 8 || std::memcmp(d.data(), sig, 8) != 0) throw LoadError("not a PNG file");
    const unsigned char* p = reinterpret_cast<const unsigned char*>(d.data());
    size_t o = 8;
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = 0, interlace = 0;
    std::string idat;
    std::vector<unsigned char> plte;
    while (o + 12 <= d.size()) {
        uint32_t len = be32(p + o);
        std::string tag(d, o + 4, 4);
        if (o + 12 + (size_t)len > d.size()) throw LoadError("truncated PNG chunk");
        const unsigned char* body = p + o + 8;
        if (tag == "IHDR") {
            if (len < 13) throw LoadError("bad IHDR");
            w = be32(body); h = be32(body + 4);
            depth = body[8]; ctype = body[9]; interlace = body[12];
        } else if (tag == "PLTE") plte.assign(body, body + len);
        else if (tag == "IDAT") idat.append(reinterpret_cast<const char*>(body), len);
        else if (tag == "IEND") break;
        o += 12 + (size_t)len;
    }
    if (w == 0 || h == 0) throw LoadError("PNG without IHDR");
    if (depth != 8 || interlace != 0) throw LoadError("unsupported PNG (only 8-bit, non-interlaced)");
    int ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (ch == 0) throw LoadError("unsupported PNG colour type");
    const size_t stride = (size_t)w * ch;
    std::vector<unsigned char> raw((stride + 1) * h);
    uLongf out_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &out_len, reinterpret_cast<const Bytef*>(idat.data()), (uLong)idat.size()) != Z_OK || out_len != raw.size())
        throw LoadError("PNG inflate failed");
    std::vector<unsigned char> px(stride * h);
    for (uint32_t y = 0; y < h; ++y) {
        const unsigned char* in = &raw[(stride + 1) * y];
        unsigned char* cur = &px[stride * y];
        const unsigned char* up = y ? &px[stride * (y - 1)] : nullptr;
        int ft = in[0];
        for (size_t x = 0; x < stride; ++x) {
            int a = x >= (size_t)ch ? cur[x - ch] : 0, b = up ? up[x] : 0, c = (up && x >= (size_t)ch) ? up[x - ch] : 0;
            int v = in[1 + x], pr = 0;
            switch (ft) {
                case 0: pr = 0; break;
                case 1: pr = a; break;
                case 2: pr = b; break;
                case 3: pr = (a + b) / 2; break;
                case 4: {
                    int pa = std::abs(b - c), pb = std::abs(a - c), pc = std::abs(a + b - 2 * c);
                    pr = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
                    break;
                }
                default: throw LoadError("bad PNG filter");
            }
            cur[x] = (unsigned char)(v + pr);
        }
    }
    Image im;
    im.width = w; im.height = h;
    im.rgb.resize((size_t)w * h * 3);
    for (size_t i = 0; i < (size_t)w * h; ++i) {
        const unsigned char* s = &px[i * ch];
        unsigned char r, g, b;
        if (ctype == 0 || ctype == 4) r = g = b = s[0];           // into_rgb8(): luma replicated, alpha dropped
        else if (ctype == 3) {
            if ((size_t)s[0] * 3 + 2 >= plte.size()) throw LoadError("PNG palette index out of range");
            r = plte[s[0] * 3]; g = plte[s[0] * 3 + 1]; b = plte[s[0] * 3 + 2];
        } else { r = s[0]; g = s[1]; b = s[2]; }
        im.rgb[i * 3] = r; im.rgb[i * 3 + 1] = g; im.rgb[i * 3 + 2] = b;
    }
    return im;
}

inline Image decode_ppm(const std::string& d) {
    size_t o = 2;
    auto token = [&]() {
        while (o < d.size()) {
            if (d[o] == '#') { while (o < d.size() && d[o] != '\n') ++o; }
            else if (std::isspace((unsigned char)d[o])) ++o;
            else break;
        }
        size_t s = o;
        while (o < d.size() && !std::isspace((unsigned char)d[o])) ++o;
        return d.substr(s, o - s);
    };
    Image im;
    im.width = (uint32_t)std::atoi(token().c_str());
    im.height = (uint32_t)std::atoi(token().c_str());
    int maxv = std::atoi(token().c_str());
    ++o;
    if (maxv != 255 || im.width == 0 || im.height == 0 || d.size() < o + (size_t)im.width * im.height * 3) throw LoadError("unsupported PPM");
    im.rgb.assign(d.begin() + (long)o, d.begin() + (long)(o + (size_t)im.width * im.height * 3));
    return im;
}

// image::open(path)?.into_rgb8() (object.rs:31-38, skybox.rs:57-64)
inline Image load_image_rgb8(const std::string& path) {
    std::string d;
    try {
        d = read_file(path);
        if (d.size() >= 2 && d[0] == 'P' && d[1] == '6') return decode_ppm(d);
        const bool is_tiff = d.size() >= 4 && ((d[0] == 'I' && d[1] == 'I' && d[2] == 42 && d[3] == 0) || (d[0] == 'M' && d[1] == 'M' && d[2] == 0 && d[3] == 42));
        const bool is_jpeg = d.size() >= 3 && (unsigned char)d[0] == 0xFF && (unsigned char)d[1] == 0xD8;
        const bool is_exr = d.size() >= 4 && (unsigned char)d[0] == 0x76 && (unsigned char)d[1] == 0x2F && (unsigned char)d[2] == 0x31 && (unsigned char)d[3] == 0x01;
        if (is_tiff || is_jpeg || is_exr) {
            imgdec::Rgb8 r = is_tiff ? imgdec::tiff::decode(d) : is_jpeg ? imgdec::jpeg::decode(d) : imgdec::exr::decode(d);
            Image im;
            im.width = r.width; im.height = r.height;
            im.rgb = std::move(r.rgb);
            return im;
        }
        return decode_png(d);
    } catch (const LoadError& e) {
        throw LoadError("Failed to load image from path: " + path + ": " + e.what());
    } catch (const imgdec::DecodeError& e) {
        throw LoadError("Failed to load image from path: " + path + ": " + e.what());
    } catch (const std::exception& e) {   // bad_alloc / length_error from a hostile header: a failed image::open, not a crash
        throw LoadError("Failed to load image from path: " + path + ": " + e.what());
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Wavefront OBJ / MTL  ([crate] obj 0.10: ObjData::load_buf, Mtl::reload)
// ---------------------------------------------------------------------------------------------------------------
struct FaceVertex { int v = -1, t = -1, n = -1; };   // -1 = absent (Option::None)
struct ObjGroup {
    std::string name;
    bool has_material = false;
    std::string material;                    // ObjMaterial::Ref(name) until load_mtls resolves it
    std::vector<std::vector<FaceVertex>> polys;
};
struct ObjObject {
    std::string name;
    std::vector<ObjGroup> groups;
};
struct ObjData {
    std::vector<Vec3> position, normal;
    std::vector<std::array<float, 2>> texture;
    std::vector<ObjObject> objects;
    std::vector<std::string> material_libs;
};
struct MtlMaterial {
    std::string name;
    std::optional<Color> kd, ks;
    std::optional<float> ns, ni, d;
    std::optional<int> illum;
    std::optional<std::string> map_kd;
};

// [crate] a trailing backslash joins the next physical line; '#' starts a comment
inline std::vector<std::pair<int, std::string>> logical_lines(const std::string& text) {
    std::vector<std::pair<int, std::string>> out;
    std::string pending;
    int no = 0;
    size_t o = 0;
    while (o <= text.size()) {
        size_t e = text.find('\n', o);
        if (e == std::string::npos) e = text.size();
        std::string line = text.substr(o, e - o);
        o = e + 1;
        ++no;
        if (!line.empty() && line.back() == '\r') line.pop_back();
        size_t c = line.find('#');
        if (c != std::string::npos) line.erase(c);
        std::string stripped = line;
        while (!stripped.empty() && std::isspace((unsigned char)stripped.back())) stripped.pop_back();
        if (!stripped.empty() && stripped.back() == '\\') {
            stripped.pop_back();
            pending += stripped + " ";
            continue;
        }
        out.push_back({no, pending + line});
        pending.clear();
        if (e == text.size()) break;
    }
    if (!pending.empty()) out.push_back({no, pending});
    return out;
}
inline std::vector<std::string> split_ws(const std::string& s) {
    std::vector<std::string> w;
    std::istringstream ss(s);
    std::string t;
    while (ss >> t) w.push_back(t);
    return w;
}
// str::parse::<f32>() is correctly rounded from the decimal text: strtof
inline float parse_f32(const std::string& s, int line, const char* what) {
    char* end = nullptr;
    float v = std::strtof(s.c_str(), &end);
    if (end == s.c_str() || *end != '\0') throw LoadError("line " + std::to_string(line) + ": bad number in '" + what + "': " + s);
    return v;
}
inline void need(const std::vector<std::string>& w, size_t n, int line, const char* what) {
    if (w.size() < n + 1) throw LoadError("line " + std::to_string(line) + ": '" + what + "' needs " + std::to_string(n) + " values");
}
// [crate] 1-based; negative counts back from the elements read SO FAR
inline int face_index(const std::string& tok, size_t count, int line) {
    char* end = nullptr;
    long i = std::strtol(tok.c_str(), &end, 10);
    if (end == tok.c_str() || *end != '\0') throw LoadError("line " + std::to_string(line) + ": bad face index '" + tok + "'");
    long j = i < 0 ? (long)count + i : i - 1;
    if (i == 0 || j < 0 || j >= (long)count) throw LoadError("line " + std::to_string(line) + ": face index " + tok + " out of range");
    return (int)j;
}
inline std::string rest_after_keyword(const std::string& line) {
    size_t a = line.find_first_not_of(" \t");
    size_t b = line.find_first_of(" \t", a);
    if (b == std::string::npos) return "";
    size_t c = line.find_first_not_of(" \t", b);
    if (c == std::string::npos) return "";
    std::string r = line.substr(c);
    while (!r.empty() && std::isspace((unsigned char)r.back())) r.pop_back();
    return r;
}

// Rules restated [crate]: faces go to the current group of the current object (both "default" on demand); `o` closes the
// current group and object (an object that never received a group is dropped); `g` closes the current group and starts a
// new one WITHOUT a material; `usemtl` on a group that has a material closes it and continues in a copy, on a group
// without one it sets the material of the whole group (faces already in it included); `mtllib`: the rest of the line,
// words re-joined by single spaces, is ONE library; `s` and `l` are ignored, any other directive is an error (the crate's
// ObjError::UnexpectedCommand: a git-LFS pointer stub in place of a mesh fails the load; [crate], parity unpinned).
inline ObjData parse_obj(const std::string& text) {
    ObjData d;
    ObjObject obj{"default", {}};
    std::optional<ObjGroup> group;
    for (auto& [no, line] : logical_lines(text)) {
        std::vector<std::string> w = split_ws(line);
        if (w.empty()) continue;
        const std::string& key = w[0];
        if (key == "v") {
            need(w, 3, no, "v");
            d.position.push_back({parse_f32(w[1], no, "v"), parse_f32(w[2], no, "v"), parse_f32(w[3], no, "v")});
        } else if (key == "vt") {
            need(w, 1, no, "vt");
            d.texture.push_back({parse_f32(w[1], no, "vt"), w.size() > 2 ? parse_f32(w[2], no, "vt") : 0.0f});
        } else if (key == "vn") {
            need(w, 3, no, "vn");
            d.normal.push_back({parse_f32(w[1], no, "vn"), parse_f32(w[2], no, "vn"), parse_f32(w[3], no, "vn")});
        } else if (key == "f") {
            std::vector<FaceVertex> poly;
            for (size_t k = 1; k < w.size(); ++k) {
                std::vector<std::string> parts;
                size_t s = 0;
                for (;;) {
                    size_t e = w[k].find('/', s);
                    parts.push_back(w[k].substr(s, e == std::string::npos ? std::string::npos : e - s));
                    if (e == std::string::npos) break;
                    s = e + 1;
                }
                if (parts.size() > 3) throw LoadError("line " + std::to_string(no) + ": malformed face vertex '" + w[k] + "'");
                FaceVertex fv;
                fv.v = face_index(parts[0], d.position.size(), no);
                if (parts.size() > 1 && !parts[1].empty()) fv.t = face_index(parts[1], d.texture.size(), no);
                if (parts.size() > 2 && !parts[2].empty()) fv.n = face_index(parts[2], d.normal.size(), no);
                poly.push_back(fv);
            }
            if (!group) group = ObjGroup{"default", false, "", {}};
            group->polys.push_back(std::move(poly));
        } else if (key == "o") {
            if (group) {
                obj.groups.push_back(std::move(*group));
                d.objects.push_back(std::move(obj));
                group.reset();
            }
            obj = ObjObject{w.size() > 1 ? rest_after_keyword(line) : std::string("default"), {}};
        } else if (key == "g") {
            if (group) {
                obj.groups.push_back(std::move(*group));
                group.reset();
            }
            if (w.size() > 1) group = ObjGroup{rest_after_keyword(line), false, "", {}};
        } else if (key == "usemtl") {
            if (!group) group = ObjGroup{"default", false, "", {}};
            if (group->has_material) {
                obj.groups.push_back(*group);
                group->polys.clear();
            }
            group->has_material = w.size() > 1;
            group->material = w.size() > 1 ? w[1] : std::string();
        } else if (key == "mtllib") {
            if (w.size() < 2) throw LoadError("line " + std::to_string(no) + ": mtllib without a file name");
            std::string name = w[1];
            for (size_t k = 2; k < w.size(); ++k) name += " " + w[k];
            d.material_libs.push_back(name);
        } else if (key != "s" && key != "l") {
            throw LoadError("line " + std::to_string(no) + ": unexpected command '" + key + "'");
        }
    }
    if (group) obj.groups.push_back(std::move(*group));
    d.objects.push_back(std::move(obj));
    return d;
}

// newmtl (first word is the name), Kd, Ks, Ns, Ni, d, illum, map_Kd (first word) [crate]; other keys are never read by
// the reference (object.rs:82-108)
inline std::vector<MtlMaterial> parse_mtl(const std::string& text) {
    std::vector<MtlMaterial> mats;
    for (auto& [no, line] : logical_lines(text)) {
        std::vector<std::string> w = split_ws(line);
        if (w.empty()) continue;
        const std::string& key = w[0];
        if (key == "newmtl") {
            if (w.size() < 2) throw LoadError("line " + std::to_string(no) + ": newmtl without a name");
            mats.push_back(MtlMaterial{});
            mats.back().name = w[1];
            continue;
        }
        if (mats.empty()) continue;
        MtlMaterial& m = mats.back();
        if (key == "Kd") { need(w, 3, no, "Kd"); m.kd = Color{parse_f32(w[1], no, "Kd"), parse_f32(w[2], no, "Kd"), parse_f32(w[3], no, "Kd")}; }
        else if (key == "Ks") { need(w, 3, no, "Ks"); m.ks = Color{parse_f32(w[1], no, "Ks"), parse_f32(w[2], no, "Ks"), parse_f32(w[3], no, "Ks")}; }
        else if (key == "Ns") { need(w, 1, no, "Ns"); m.ns = parse_f32(w[1], no, "Ns"); }
        else if (key == "Ni") { need(w, 1, no, "Ni"); m.ni = parse_f32(w[1], no, "Ni"); }
        else if (key == "d") { need(w, 1, no, "d"); m.d = parse_f32(w[1], no, "d"); }
        else if (key == "illum") {
            need(w, 1, no, "illum");
            char* end = nullptr;
            long v = std::strtol(w[1].c_str(), &end, 10);
            if (end == w[1].c_str() || *end != '\0') throw LoadError("line " + std::to_string(no) + ": bad illum");
            m.illum = (int)v;
        } else if (key == "map_Kd") {
            if (w.size() < 2) throw LoadError("line " + std::to_string(no) + ": map_Kd without a file name");
            m.map_kd = w[1];
        }
    }
    return mats;
}

// filename() (object.rs:41-54): file name up to the first '.', first character ASCII-uppercased
inline std::string object_name(const std::string& path) {
    size_t p = path.find_last_of('/');
    std::string base = p == std::string::npos ? path : path.substr(p + 1);
    std::string stem = base.substr(0, base.find('.'));
    if (!stem.empty() && stem[0] >= 'a' && stem[0] <= 'z') stem[0] = (char)(stem[0] - 'a' + 'A');
    return stem;
}

// Object::from_obj (object.rs:61-175) + triangulate (object.rs:224-296).  rotation = unit quaternion (i, j, k, w).
inline Object load_object(const std::string& path, Vec3 translation = {0, 0, 0}, std::array<float, 4> rotation = {0, 0, 0, 1},
                          Vec3 scale = {1, 1, 1}) {
    ObjData data;
    try {
        data = parse_obj(read_file(path));
    } catch (const LoadError& e) {
        throw LoadError("Failed to load obj from path: " + path + ": " + e.what());
    }
    const std::string base = parent_dir(path);
    std::vector<MtlMaterial> mtl;
    for (const std::string& lib : data.material_libs) {   // load_mtls(): every library, relative to the OBJ (object.rs:72-75)
        try {
            std::vector<MtlMaterial> m = parse_mtl(read_file(join_path(base, lib)));
            mtl.insert(mtl.end(), m.begin(), m.end());
        } catch (const LoadError& e) {
            throw LoadError("Failed to load materials from obj path: " + path + ": " + lib + ": " + e.what());
        }
    }
    Object o;
    o.name = object_name(path);
    for (const MtlMaterial& m : mtl) {   // object.rs:77-109
        Material mat;
        mat.name = m.name;
        mat.diffuse_color = m.kd;
        mat.specular_color = m.ks;
        mat.specular_exponent = m.ns;
        if (m.map_kd) {
            std::string tp = join_path(base, *m.map_kd);
            try {
                mat.diffuse_texture = std::make_shared<const Image>(load_image_rgb8(tp));
            } catch (const LoadError& e) {   // object.rs:92-96: warn, carry on without the texture
                warn("Failed to load texture from path: " + tp + ": " + e.what());
            }
        }
        std::optional<IlluminationModel> il = m.illum ? IlluminationModel::from_i32(*m.illum) : std::nullopt;
        if (!il) {
            warn("Invalid illumination model: " + std::to_string(m.illum.value_or(-1)));
            il = IlluminationModel{};
        }
        mat.illumination_model = *il;
        if (m.d) mat.dissolve = 1.0f - *m.d;   // object.rs:106
        mat.refraction_index = m.ni;
        o.materials.push_back(std::move(mat));
    }
    uint32_t zero_normals = 0, no_normals = 0, no_uv = 0;
    for (const ObjObject& oo : data.objects)   // object.rs:111-142
        for (const ObjGroup& g : oo.groups) {
            std::optional<size_t> mi;
            if (g.has_material) {
                bool known = false;
                for (const MtlMaterial& m : mtl) known = known || m.name == g.material;
                if (!known) throw LoadError("Material reference not supported: " + g.material);   // object.rs:121-123 panics
                for (size_t k = 0; k < o.materials.size(); ++k)
                    if (o.materials[k].name == g.material) { mi = k; break; }                      // .position(): the first
            }
            for (const auto& poly : g.polys) {
                if (poly.empty()) throw LoadError("face without vertices (the reference underflows `len - 1`, object.rs:232)");
                for (size_t i = 1; i + 1 < poly.size(); ++i) {
                    const FaceVertex* fv[3] = {&poly[0], &poly[i], &poly[i + 1]};
                    Triangle t;
                    t.a = data.position[fv[0]->v]; t.b = data.position[fv[1]->v]; t.c = data.position[fv[2]->v];
                    // (a - b).cross(&(a - c)).try_normalize(f32::EPSILON) or zero (object.rs:237-243), fp32 in nalgebra's order
                    Vec3 e1{t.a[0] - t.b[0], t.a[1] - t.b[1], t.a[2] - t.b[2]}, e2{t.a[0] - t.c[0], t.a[1] - t.c[1], t.a[2] - t.c[2]};
                    Vec3 cr{e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
                    float n = std::sqrt((cr[0] * cr[0] + cr[1] * cr[1]) + cr[2] * cr[2]);
                    Vec3 computed{0, 0, 0};
                    if (n > 1.1920929e-07f) computed = {cr[0] / n, cr[1] / n, cr[2] / n};
                    else ++zero_normals;
                    Vec3* nout[3] = {&t.a_normal, &t.b_normal, &t.c_normal};
                    std::array<float, 2>* uvout[3] = {&t.a_uv, &t.b_uv, &t.c_uv};
                    for (int k = 0; k < 3; ++k) {
                        if (fv[k]->n >= 0) *nout[k] = data.normal[fv[k]->n];
                        else { *nout[k] = computed; ++no_normals; }
                        if (fv[k]->t >= 0) *uvout[k] = data.texture[fv[k]->t];
                        else { *uvout[k] = {0.0f, 0.0f}; ++no_uv; }
                    }
                    t.material_index = mi;
                    o.triangles.push_back(t);
                }
            }
        }
    if (zero_normals) warn("Computed normals for " + std::to_string(zero_normals) + " triangles is zero");
    if (no_normals) warn("No normals for " + std::to_string(no_normals) + " triangles");
    if (no_uv) warn("No UV for " + std::to_string(no_uv) + " triangles");
    o.translation = translation;
    o.rotation = rotation;
    o.scale = scale;
    return o;
}

// UnitQuaternion::from_euler_angles(roll, pitch, yaw) of degrees * PI * 2 / 360 (object.rs:336-340), fp32 throughout
// ([crate] nalgebra: half-angle product formula)
inline std::array<float, 4> quaternion_from_euler_degrees(float x, float y, float z) {
    const float pi = 3.14159265358979323846f;
    auto rad = [&](float d) { return d * pi * 2.0f / 360.0f; };
    float hr = rad(x) * 0.5f, hp = rad(y) * 0.5f, hy = rad(z) * 0.5f;
    // sinf/cosf as the rounded double result, like raytracing_amd/scene.py (equal to a correctly rounded sinf in all but
    // vanishingly rare double-rounding cases; libm's own sinf is only < 1 ulp)
    auto s32 = [](float a) { return (float)std::sin((double)a); };
    auto c32 = [](float a) { return (float)std::cos((double)a); };
    float sr = s32(hr), cr = c32(hr), sp = s32(hp), cp = c32(hp), sy = s32(hy), cy = c32(hy);
    float w = cr * cp * cy + sr * sp * sy;
    float i = sr * cp * cy - cr * sp * sy;
    float j = cr * sp * cy + sr * cp * sy;
    float k = cr * cp * sy - sr * sp * cy;
    return {i, j, k, w};
}

// ---------------------------------------------------------------------------------------------------------------
// YAML subset
// ---------------------------------------------------------------------------------------------------------------
struct Yaml {
    enum Kind { Null, Scalar, Map, Seq } kind = Null;
    std::string scalar, tag;
    bool quoted = false;
    std::vector<std::pair<std::string, Yaml>> map;
    std::vector<Yaml> seq;
    const Yaml* get(const std::string& key) const {
        if (kind != Map) return nullptr;
        for (auto& kv : map)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
};

class YamlParser {
public:
    explicit YamlParser(const std::string& text) {
        size_t o = 0;
        int no = 0;
        while (o <= text.size()) {
            size_t e = text.find('\n', o);
            if (e == std::string::npos) e = text.size();
            std::string line = text.substr(o, e - o);
            o = e + 1;
            ++no;
            if (!line.empty() && line.back() == '\r') line.pop_back();
            strip_comment(line);
            size_t ind = line.find_first_not_of(' ');
            if (ind != std::string::npos && line.find('\t') < ind) throw LoadError("YAML line " + std::to_string(no) + ": tab in indentation");
            if (ind != std::string::npos && line.substr(ind) != "---") lines_.push_back({(int)ind, line.substr(ind), no});
            if (e == text.size()) break;
        }
    }
    Yaml parse() {
        if (lines_.empty()) return Yaml{};
        Yaml y = block(lines_[0].indent);
        if (pos_ < lines_.size()) fail("unexpected content");
        return y;
    }

private:
    struct Line { int indent; std::string text; int no; };
    std::vector<Line> lines_;
    size_t pos_ = 0;

    [[noreturn]] void fail(const std::string& m) const {
        int no = pos_ < lines_.size() ? lines_[pos_].no : (lines_.empty() ? 0 : lines_.back().no);
        throw LoadError("YAML line " + std::to_string(no) + ": " + m);
    }
    static void strip_comment(std::string& s) {
        char q = 0;
        for (size_t i = 0; i < s.size(); ++i) {
            if (q) { if (s[i] == q) q = 0; }
            else if (s[i] == '"' || s[i] == '\'') q = s[i];
            else if (s[i] == '#' && (i == 0 || s[i - 1] == ' ')) { s.erase(i); break; }
        }
        while (!s.empty() && (s.back() == ' ' || s.back() == '\t')) s.pop_back();
    }
    // position of the ": " (or trailing ":") that ends a plain key, outside quotes and flow brackets
    static size_t key_colon(const std::string& s) {
        char q = 0;
        int depth = 0;
        for (size_t i = 0; i < s.size(); ++i) {
            char c = s[i];
            if (q) { if (c == q) q = 0; continue; }
            if (c == '"' || c == '\'') q = c;
            else if (c == '{' || c == '[') ++depth;
            else if (c == '}' || c == ']') --depth;
            else if (c == ':' && depth == 0 && (i + 1 == s.size() || s[i + 1] == ' ')) return i;
        }
        return std::string::npos;
    }
    static std::string trim(const std::string& s) {
        size_t a = s.find_first_not_of(" \t"), b = s.find_last_not_of(" \t");
        return a == std::string::npos ? "" : s.substr(a, b - a + 1);
    }
    static std::string unquote(const std::string& s, bool* quoted) {
        *quoted = s.size() >= 2 && (s[0] == '"' || s[0] == '\'') && s.back() == s[0];
        return *quoted ? s.substr(1, s.size() - 2) : s;
    }

    // inline value after "key:" or "- ": tag, flow collection or scalar
    Yaml inline_value(std::string v) {
        Yaml y;
        v = trim(v);
        if (!v.empty() && v[0] == '!') {
            size_t sp = v.find(' ');
            y.tag = v.substr(0, sp);
            v = sp == std::string::npos ? "" : trim(v.substr(sp));
        }
        std::string tag = y.tag;
        if (v.empty() || v == "~" || v == "null") { y.kind = Yaml::Null; return y; }
        if (v[0] == '{' || v[0] == '[') {
            size_t i = 0;
            y = flow(v, i);
            y.tag = tag;
            if (trim(v.substr(i)) != "") fail("trailing characters after flow collection");
            return y;
        }
        y.kind = Yaml::Scalar;
        y.scalar = unquote(v, &y.quoted);
        return y;
    }
    Yaml flow(const std::string& s, size_t& i) {
        Yaml y;
        auto skip = [&]() { while (i < s.size() && s[i] == ' ') ++i; };
        skip();
        if (i >= s.size()) fail("unterminated flow collection");
        if (s[i] == '{') {
            y.kind = Yaml::Map;
            ++i;
            for (;;) {
                skip();
                if (i < s.size() && s[i] == '}') { ++i; break; }
                size_t c = s.find(':', i);
                if (c == std::string::npos) fail("flow mapping without ':'");
                bool q;
                std::string key = unquote(trim(s.substr(i, c - i)), &q);
                i = c + 1;
                Yaml v = flow(s, i);
                y.map.push_back({key, v});
                skip();
                if (i < s.size() && s[i] == ',') { ++i; continue; }
                if (i < s.size() && s[i] == '}') { ++i; break; }
                fail("bad flow mapping");
            }
        } else if (s[i] == '[') {
            y.kind = Yaml::Seq;
            ++i;
            for (;;) {
                skip();
                if (i < s.size() && s[i] == ']') { ++i; break; }
                y.seq.push_back(flow(s, i));
                skip();
                if (i < s.size() && s[i] == ',') { ++i; continue; }
                if (i < s.size() && s[i] == ']') { ++i; break; }
                fail("bad flow sequence");
            }
        } else {
            size_t e = i;
            char q = 0;
            for (; e < s.size(); ++e) {
                if (q) { if (s[e] == q) q = 0; continue; }
                if (s[e] == '"' || s[e] == '\'') q = s[e];
                else if (s[e] == ',' || s[e] == '}' || s[e] == ']') break;
            }
            std::string v = trim(s.substr(i, e - i));
            i = e;
            if (v.empty() || v == "~" || v == "null") y.kind = Yaml::Null;
            else { y.kind = Yaml::Scalar; y.scalar = unquote(v, &y.quoted); }
        }
        return y;
    }
    // block node whose lines have exactly `indent` leading spaces
    Yaml block(int indent) {
        if (pos_ >= lines_.size()) return Yaml{};
        const std::string& first = lines_[pos_].text;
        if (first[0] == '-' && (first.size() == 1 || first[1] == ' ')) return sequence(indent);
        if (key_colon(first) != std::string::npos) return mapping(indent);
        Yaml y = inline_value(first);
        ++pos_;
        return y;
    }
    Yaml value_after_key(const std::string& rest, int indent) {
        Yaml v = inline_value(rest);
        if (v.kind != Yaml::Null) return v;
        // nothing (or only a tag) on the key's line: a nested block on deeper lines, or a sequence at the same indent
        if (pos_ < lines_.size()) {
            const Line& n = lines_[pos_];
            bool dash = n.text[0] == '-' && (n.text.size() == 1 || n.text[1] == ' ');
            if (n.indent > indent || (n.indent == indent && dash)) {
                std::string tag = v.tag;
                v = block(n.indent);
                if (!tag.empty()) v.tag = tag;
            }
        }
        return v;
    }
    Yaml mapping(int indent) {
        Yaml y;
        y.kind = Yaml::Map;
        while (pos_ < lines_.size() && lines_[pos_].indent == indent) {
            std::string t = lines_[pos_].text;
            if (t[0] == '-' && (t.size() == 1 || t[1] == ' ')) break;
            size_t c = key_colon(t);
            if (c == std::string::npos) fail("expected 'key: value'");
            bool q;
            std::string key = unquote(trim(t.substr(0, c)), &q);
            ++pos_;
            y.map.push_back({key, value_after_key(t.substr(c + 1), indent)});
        }
        if (pos_ < lines_.size() && lines_[pos_].indent > indent) fail("bad indentation");
        return y;
    }
    Yaml sequence(int indent) {
        Yaml y;
        y.kind = Yaml::Seq;
        while (pos_ < lines_.size() && lines_[pos_].indent == indent) {
            std::string t = lines_[pos_].text;
            if (!(t[0] == '-' && (t.size() == 1 || t[1] == ' '))) break;
            std::string rest = t.size() > 2 ? t.substr(2) : "";
            size_t lead = rest.find_first_not_of(' ');
            if (lead == std::string::npos) {   // "-" alone: the item is the nested block
                ++pos_;
                if (pos_ < lines_.size() && lines_[pos_].indent > indent) y.seq.push_back(block(lines_[pos_].indent));
                else y.seq.push_back(Yaml{});
                continue;
            }
            if (key_colon(rest) != std::string::npos && rest[lead] != '{' && rest[lead] != '[') {
                // "- key: value": a mapping whose first key sits on the dash line; re-indent that line and parse a mapping
                lines_[pos_].indent = indent + 2 + (int)lead;
                lines_[pos_].text = rest.substr(lead);
                y.seq.push_back(mapping(lines_[pos_].indent));
            } else {
                ++pos_;
                y.seq.push_back(inline_value(rest));
            }
        }
        return y;
    }
};

// ---------------------------------------------------------------------------------------------------------------
// Scene::load
// ---------------------------------------------------------------------------------------------------------------
// serde_yml reads floats as f64 and narrows with `as f32`: strtod, then (float)
inline float yaml_f32(const Yaml* y, const std::string& where) {
    if (!y) throw LoadError(where + ": missing field");
    if (y->kind != Yaml::Scalar || y->quoted) throw LoadError(where + ": invalid type, expected f32");
    char* end = nullptr;
    std::string s = y->scalar;
    if (s == ".inf" || s == "+.inf") return INFINITY;
    if (s == "-.inf") return -INFINITY;
    if (s == ".nan") return NAN;
    double v = std::strtod(s.c_str(), &end);
    if (end == s.c_str() || *end != '\0') throw LoadError(where + ": invalid type, expected f32");
    return (float)v;
}
inline uint32_t yaml_u32(const Yaml* y, const std::string& where) {
    if (!y) throw LoadError(where + ": missing field");
    if (y->kind != Yaml::Scalar || y->quoted) throw LoadError(where + ": invalid type, expected u32");
    char* end = nullptr;
    const std::string& s = y->scalar;
    if (s.empty() || s[0] == '-') throw LoadError(where + ": invalid type, expected u32");
    unsigned long long v = std::strtoull(s.c_str(), &end, 10);
    if (end == s.c_str() || *end != '\0' || v > 0xFFFFFFFFull) throw LoadError(where + ": invalid type, expected u32");
    return (uint32_t)v;
}
inline Vec3 yaml_xyz(const Yaml* parent, const std::string& key, const std::string& where, const char* names = "xyz") {
    const Yaml* m = parent ? parent->get(key) : nullptr;
    if (!m) throw LoadError(where + ": missing field `" + key + "`");
    Vec3 v{};
    for (int k = 0; k < 3; ++k) v[k] = yaml_f32(m->get(std::string(1, names[k])), where + "." + key + "." + names[k]);
    return v;
}
// try_normalize(0.0).unwrap_or_default() and the norm (light.rs:33, settings.rs:51-55)
inline Color normalize0(Color c, float* norm) {
    float n = std::sqrt((c[0] * c[0] + c[1] * c[1]) + c[2] * c[2]);
    if (norm) *norm = n;
    if (n > 0.0f) return {c[0] / n, c[1] / n, c[2] / n};
    return {0, 0, 0};
}

// SettingsDef (settings.rs:34-60): every field is required; any failure makes the caller fall back to the defaults
inline Settings settings_from_yaml(const Yaml& m) {
    if (m.kind != Yaml::Map) throw LoadError("extraArgs: invalid type, expected struct SettingsDef");
    Settings s;
    s.max_bounces = yaml_u32(m.get("max_bounces"), "extraArgs.max_bounces");
    s.samples = yaml_u32(m.get("samples"), "extraArgs.samples");
    float norm = 0.0f;
    s.ambient_color = normalize0(yaml_xyz(&m, "ambient_color", "extraArgs", "rgb"), &norm);
    s.ambient_intensity = norm;
    const Yaml* sb = m.get("skybox");
    if (!sb) throw LoadError("extraArgs: missing field `skybox`");
    std::string kind = sb->tag.empty() ? "" : sb->tag.substr(1);
    const Yaml* val = sb;
    if (kind.empty()) {   // the single-key map spelling of the externally tagged enum (skybox.rs:22-26)
        if (sb->kind != Yaml::Map || sb->map.size() != 1) throw LoadError("extraArgs.skybox: expected `!Color [r, g, b]` or `!Path file`");
        kind = sb->map[0].first;
        val = &sb->map[0].second;
    }
    if (kind == "Color") {
        if (val->kind != Yaml::Seq || val->seq.size() != 3) throw LoadError("extraArgs.skybox: Color needs 3 numbers");
        Color c{};
        for (int k = 0; k < 3; ++k) c[k] = yaml_f32(&val->seq[k], "extraArgs.skybox");
        s.skybox = c;
    } else if (kind == "Path") {
        if (val->kind != Yaml::Scalar) throw LoadError("extraArgs.skybox: Path needs a file name");
        // skybox.rs:57-64: the path is used AS WRITTEN (relative to the working directory, not to the YAML)
        try {
            s.skybox = std::shared_ptr<const Image>(std::make_shared<Image>(load_image_rgb8(val->scalar)));
        } catch (const LoadError& e) {
            throw LoadError(std::string("Failed to load skybox: ") + e.what());
        }
    } else throw LoadError("extraArgs.skybox: expected `!Color [r, g, b]` or `!Path file`");
    const Yaml* aa = m.get("anti_aliasing");
    if (!aa || aa->kind != Yaml::Scalar || aa->quoted || (aa->scalar != "true" && aa->scalar != "false"))
        throw LoadError("extraArgs: missing or non-bool field `anti_aliasing`");
    s.anti_aliasing = aa->scalar == "true";
    return s;
}

// WithRelativePath::deserialize (mod.rs:64-123).  `path` is the YAML's own path: model files resolve against its parent.
inline Scene scene_from_yaml(const std::string& text, const std::string& path) {
    try {
        Yaml doc = YamlParser(text).parse();
        if (doc.kind != Yaml::Map) throw LoadError("expected a mapping at the top level");
        const std::string base = parent_dir(path);
        Scene sc;
        const Yaml* models = doc.get("models");
        if (!models) throw LoadError("missing field `models`");
        if (models->kind != Yaml::Seq) throw LoadError("`models`: invalid type, expected a sequence");
        for (size_t i = 0; i < models->seq.size(); ++i) {
            const Yaml& m = models->seq[i];
            const std::string where = "models[" + std::to_string(i) + "]";
            const Yaml* fp = m.get("filePath");
            if (!fp || fp->kind != Yaml::Scalar) throw LoadError(where + ": missing field `filePath`");
            Vec3 pos = yaml_xyz(&m, "position", where), rot = yaml_xyz(&m, "rotation", where), scl = yaml_xyz(&m, "scale", where);
            sc.objects.push_back(load_object(join_path(base, fp->scalar), pos, quaternion_from_euler_degrees(rot[0], rot[1], rot[2]), scl));
        }
        const Yaml* lights = doc.get("pointLights");
        if (!lights) throw LoadError("missing field `pointLights`");
        if (lights->kind != Yaml::Seq) throw LoadError("`pointLights`: invalid type, expected a sequence");
        for (size_t i = 0; i < lights->seq.size(); ++i) {
            const Yaml& m = lights->seq[i];
            const std::string where = "pointLights[" + std::to_string(i) + "]";
            Light l;
            l.position = yaml_xyz(&m, "position", where);
            l.color = normalize0(yaml_xyz(&m, "Ke", where, "rgb"), nullptr);
            l.intensity = yaml_f32(m.get("intensity"), where + ".intensity");
            sc.lights.push_back(l);
        }
        const Yaml* cam = doc.get("camera");
        if (!cam) throw LoadError("missing field `camera`");
        sc.camera.position = yaml_xyz(cam, "position", "camera");
        sc.camera.look_at = yaml_xyz(cam, "lookAt", "camera");
        sc.camera.up = yaml_xyz(cam, "upVec", "camera");
        // f32::to_radians: self * (PI_f32 / 180_f32), the quotient rounded to f32 once (camera.rs:75)
        sc.camera.fov = yaml_f32(cam->get("fieldOfView"), "camera.fieldOfView") * (3.14159265358979323846f / 180.0f);
        sc.camera.resolution = {yaml_u32(cam->get("width"), "camera.width"), yaml_u32(cam->get("height"), "camera.height")};
        if (const Yaml* ea = doc.get("extraArgs")) {   // mod.rs:100-110: failure or absence -> Settings::default()
            try {
                sc.settings = settings_from_yaml(*ea);
            } catch (const LoadError& e) {
                warn(std::string("Failed to deserialize extraArgs: ") + e.what());
                sc.settings = Settings{};
            }
        }
        return sc;
    } catch (const LoadError& e) {
        throw LoadError("Failed to deserialize scene from path: " + path + "\n" + e.what());
    }
}

// Scene::load (mod.rs:125-140)
inline Scene load_scene(const std::string& path) {
    std::string text;
    try {
        text = read_file(path);
    } catch (const LoadError& e) {
        throw LoadError("Failed to read file from path: " + path + ": " + e.what());
    }
    return scene_from_yaml(text, path);
}

}   // namespace loader
}   // namespace trayracer
