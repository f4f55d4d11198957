"""Shared helpers of the parity tests."""
import json

import numpy as np

from raytracing_amd.scene import Camera, IlluminationModel, Light, Material, Object, Scene, Settings, Skybox


def compare_images(gpu_rgb, ref_rgb):
    """Returns (fraction of pixels whose 3 channels are all within 1 LSB, fraction bit-identical, max abs diff)."""
    d = np.abs(gpu_rgb.astype(np.int16) - ref_rgb.astype(np.int16)).max(axis=-1)
    return float((d <= 1).mean()), float((d == 0).mean()), int(d.max())


def compare_ids(g_obj, g_tri, r_obj, r_tri):
    same = (g_obj == r_obj) & (g_tri == r_tri)
    return float(same.mean())


# The bar of BASELINE.json's north_star (fractions of pixels).
ID_BAR = 0.9999
RGB_BAR = 0.999


def assert_parity(out, ref, ids=True, what=""):
    if ids:
        f = compare_ids(out["hit_obj"], out["hit_tri"], ref["hit_obj"], ref["hit_tri"])
        assert f >= ID_BAR, f"{what}: primary-hit ids identical on only {f:.6f} of pixels"
    w1, ex, mx = compare_images(out["rgb"], ref["rgb"])
    assert w1 >= RGB_BAR, f"{what}: RGB8 within 1 LSB on only {w1:.6f} of pixels (exact {ex:.6f}, max diff {mx})"
    return w1, ex


# ---- hermetic scene (de)serialisation for tests/golden: geometry generated with numpy's float64 transcendental
# loops may differ in the last bit between CPUs, so fixtures carry their inputs with them -------------------------
def scene_to_arrays(sc: Scene) -> dict:
    d = {}
    meta = {"n_objects": len(sc.objects), "objects": [], "lights": [], "settings": {}, "camera": {}}
    for i, o in enumerate(sc.objects):
        d[f"o{i}_tri"] = o.triangles
        d[f"o{i}_mi"] = o.material_index
        mats = []
        for k, m in enumerate(o.materials):
            mats.append({
                "name": m.name,
                "kd": None if m.diffuse_color is None else [float(x) for x in m.diffuse_color],
                "ks": None if m.specular_color is None else [float(x) for x in m.specular_color],
                "ns": m.specular_exponent, "illum": m.illumination_model.value, "dissolve": m.dissolve,
                "tex": m.diffuse_texture is not None})
            if m.diffuse_texture is not None:
                d[f"o{i}_m{k}_tex"] = m.diffuse_texture
        meta["objects"].append({"name": o.name, "materials": mats})
        d[f"o{i}_trs"] = np.concatenate([o.translation, o.rotation, o.scale]).astype(np.float32)
    d["lights"] = np.array([[*l.position, *l.color, l.intensity] for l in sc.lights], np.float32).reshape(-1, 7)
    s = sc.settings
    d["ambient"] = np.array([*s.ambient_color, s.ambient_intensity], np.float32)
    meta["settings"] = {"samples": s.samples, "aa": bool(s.anti_aliasing), "seed": int(s.seed), "sky_image": s.skybox.image is not None}
    if s.skybox.image is not None:
        d["sky"] = s.skybox.image
    else:
        d["sky_color"] = np.asarray(s.skybox.color, np.float32)
    c = sc.camera
    d["camera"] = np.array([*c.position, *c.look_at, *c.up, c.fov], np.float32)
    meta["camera"] = {"resolution": [int(c.resolution[0]), int(c.resolution[1])]}
    d["meta"] = np.frombuffer(json.dumps(meta).encode(), np.uint8)
    return d


def scene_from_arrays(d) -> Scene:
    meta = json.loads(bytes(d["meta"]).decode())
    objs = []
    for i, om in enumerate(meta["objects"]):
        mats = []
        for k, m in enumerate(om["materials"]):
            mats.append(Material(
                name=m["name"], diffuse_color=None if m["kd"] is None else np.array(m["kd"], np.float32),
                specular_color=None if m["ks"] is None else np.array(m["ks"], np.float32), specular_exponent=m["ns"],
                diffuse_texture=np.ascontiguousarray(d[f"o{i}_m{k}_tex"]) if m["tex"] else None,
                illumination_model=IlluminationModel(m["illum"]), dissolve=m["dissolve"]))
        trs = d[f"o{i}_trs"]
        objs.append(Object(name=om["name"], triangles=d[f"o{i}_tri"], material_index=d[f"o{i}_mi"], materials=mats,
                           translation=trs[0:3], rotation=trs[3:7], scale=trs[7:10]))
    lights = [Light(np.array(r[0:3], np.float32), np.array(r[3:6], np.float32), float(r[6])) for r in d["lights"]]
    sm = meta["settings"]
    sky = Skybox(image=np.ascontiguousarray(d["sky"])) if sm["sky_image"] else Skybox(color=np.array(d["sky_color"], np.float32))
    amb = d["ambient"]
    st = Settings(samples=sm["samples"], ambient_color=np.array(amb[0:3], np.float32), ambient_intensity=float(amb[3]),
                  skybox=sky, anti_aliasing=sm["aa"], seed=sm["seed"])
    cam = d["camera"]
    c = Camera(position=np.array(cam[0:3], np.float32), look_at=np.array(cam[3:6], np.float32), up=np.array(cam[6:9], np.float32),
               fov=float(cam[9]), resolution=tuple(meta["camera"]["resolution"]))
    return Scene(objects=objs, lights=lights, camera=c, settings=st)


# ---- writes a Scene in the reference's on-disk formats (OBJ + MTL + PNG textures + scene YAML) so that the loader
# (raytracing_amd/loader.py) can be fed real files; decimals are exact round-trip spellings of the f32 values --------
def export_scene(sc: Scene, dirpath: str, yaml_name: str = "scene.yaml", tex_formats=("png",), sky_format: str = "png") -> str:
    """Writes `sc` as the reference's files (scene YAML, OBJ, MTL, images).  ``tex_formats`` are dealt to the textured
    materials in turn ("png", "tif" = LZW TIFF, "jpg" = baseline 4:2:0 JPEG: the formats of the reference's own textures);
    ``sky_format`` "png" or "exr" (PIZ-compressed HALF channels, the reference's skybox format, via tests/exr_writer.py)."""
    import os

    from raytracing_amd.cli import write_png
    from raytracing_amd.loader import scene_to_yaml

    n_tex = [0]

    def write_texture(stem, img):
        fmt = tex_formats[n_tex[0] % len(tex_formats)]
        n_tex[0] += 1
        name = f"{stem}.{fmt}"
        if fmt == "png":
            write_png(os.path.join(dirpath, name), img)
        else:
            from PIL import Image
            im = Image.fromarray(np.ascontiguousarray(img), "RGB")
            if fmt == "tif":
                im.save(os.path.join(dirpath, name), compression="tiff_lzw")
            elif fmt == "jpg":
                im.save(os.path.join(dirpath, name), quality=92)
            else:
                raise ValueError(fmt)
        return name

    def r(x):
        return repr(float(np.float32(x)))

    os.makedirs(dirpath, exist_ok=True)
    for oi, o in enumerate(sc.objects):
        stem = f"obj{oi}"
        with open(os.path.join(dirpath, stem + ".mtl"), "w") as f:
            for k, m in enumerate(o.materials):
                f.write(f"newmtl {m.name}\n")
                if m.diffuse_color is not None:
                    f.write("Kd " + " ".join(r(x) for x in m.diffuse_color) + "\n")
                if m.specular_color is not None:
                    f.write("Ks " + " ".join(r(x) for x in m.specular_color) + "\n")
                if m.specular_exponent is not None:
                    f.write(f"Ns {r(m.specular_exponent)}\n")
                if m.dissolve is not None:
                    f.write(f"d {r(np.float32(1.0) - np.float32(m.dissolve))}\n")
                f.write(f"illum {m.illumination_model.value}\n")
                if m.diffuse_texture is not None:
                    f.write(f"map_Kd {write_texture(f'{stem}_m{k}', m.diffuse_texture)}\n")
        with open(os.path.join(dirpath, stem + ".obj"), "w") as f:
            f.write(f"mtllib {stem}.mtl\no {o.name or stem}\n")
            prev = None
            for ti, (t, mi) in enumerate(zip(o.triangles, o.material_index)):
                if prev is None or mi != prev:
                    f.write(f"g run{ti}\n")                    # a new group carries no material until `usemtl`
                    if mi >= 0:
                        f.write(f"usemtl {o.materials[mi].name}\n")
                    prev = mi
                for k in range(3):
                    f.write("v " + " ".join(r(x) for x in t[3 * k:3 * k + 3]) + "\n")
                    f.write("vn " + " ".join(r(x) for x in t[9 + 3 * k:12 + 3 * k]) + "\n")
                    f.write("vt " + " ".join(r(x) for x in t[18 + 2 * k:20 + 2 * k]) + "\n")
                f.write("f -3/-3/-3 -2/-2/-2 -1/-1/-1\n")
        o.path = f"./{stem}.obj"
    if sc.settings.skybox.image is not None and sky_format == "exr":
        from tests import exr_writer as W
        sky = os.path.join(dirpath, "sky.exr")
        f32 = sc.settings.skybox.image.astype(np.float32) / np.float32(255.0)
        W.write_exr(sky, {"R": (W.HALF, f32[..., 0]), "G": (W.HALF, f32[..., 1]), "B": (W.HALF, f32[..., 2])}, W.PIZ)
        sc.settings.skybox.path = sky
    elif sc.settings.skybox.image is not None:
        sky = os.path.join(dirpath, "sky.png")
        write_png(sky, sc.settings.skybox.image)
        sc.settings.skybox.path = sky
    path = os.path.join(dirpath, yaml_name)
    with open(path, "w") as f:
        f.write(scene_to_yaml(sc))
    return path


# ---- the flat scene dump written by the C++ host layer (examples/scene_dump.hpp, "TRRSCN01") ---------------------------
import struct  # noqa: E402


class _Reader:
    def __init__(self, data):
        self.d, self.o = data, 0

    def take(self, fmt):
        v = struct.unpack_from("<" + fmt, self.d, self.o)
        self.o += struct.calcsize("<" + fmt)
        return v if len(v) > 1 else v[0]

    def array(self, dtype, n):
        a = np.frombuffer(self.d, dtype, n, self.o).copy()
        self.o += a.nbytes
        return a

    def image(self):
        w, h = self.take("II")
        return self.array(np.uint8, w * h * 3).reshape(h, w, 3) if w * h else None


def read_scene_dump(path) -> Scene:
    r = _Reader(open(path, "rb").read())
    assert r.d[:8] == b"TRRSCN01"
    r.o = 8
    objects = []
    for _ in range(r.take("I")):
        nt = r.take("I")
        rec = r.array(np.dtype([("f", "<f4", 24), ("m", "<i4")]), nt)
        mats = []
        for _k in range(r.take("I")):
            flags = r.take("I")
            kd, ks = r.array(np.float32, 3), r.array(np.float32, 3)
            ns, dis, illum = r.take("f"), r.take("f"), r.take("i")
            tex = r.image()
            mats.append(Material(diffuse_color=kd if flags & 1 else None, specular_color=ks if flags & 2 else None,
                                 specular_exponent=ns if flags & 4 else None, dissolve=dis if flags & 8 else None,
                                 illumination_model=IlluminationModel(illum), diffuse_texture=tex))
        t, q, s = r.array(np.float32, 3), r.array(np.float32, 4), r.array(np.float32, 3)
        objects.append(Object(triangles=rec["f"], material_index=rec["m"], materials=mats, translation=t, rotation=q, scale=s))
    lights = []
    for _ in range(r.take("I")):
        p, c = r.array(np.float32, 3), r.array(np.float32, 3)
        lights.append(Light(p, c, r.take("f")))
    samples, aa, seed = r.take("I"), r.take("I"), r.take("Q")
    amb, amb_i = r.array(np.float32, 3), r.take("f")
    sky = Skybox(image=r.image()) if r.take("I") else Skybox(color=r.array(np.float32, 3))
    st = Settings(samples=samples, anti_aliasing=bool(aa), seed=seed, ambient_color=amb, ambient_intensity=amb_i, skybox=sky)
    pos, look, up = r.array(np.float32, 3), r.array(np.float32, 3), r.array(np.float32, 3)
    fov, w, h = r.take("f"), r.take("I"), r.take("I")
    assert r.o == len(r.d)
    return Scene(objects=objects, lights=lights, camera=Camera(pos, look, up, fov, (w, h)), settings=st)


