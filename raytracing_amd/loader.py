"""Scene ingestion: YAML scene file + Wavefront OBJ/MTL -> ``raytracing_amd.scene.Scene``.

Restates, as host-side data preparation (no arithmetic that reaches a pixel other than the values stored), what
the reference does before ``Raytracer::new`` sees a scene:

  * ``Scene::load``                 src/scene/mod.rs:64-140   (models / pointLights / camera / extraArgs)
  * ``ObjectDef`` + euler degrees   src/scene/object.rs:314-357
  * ``Object::from_obj``            src/scene/object.rs:61-175 (materials of every mtllib in order, triangles in
                                    objects -> groups -> polygons order, material index by NAME position)
  * ``triangulate``                 src/scene/object.rs:224-296 (fan (0, i, i+1), computed normal (a-b)x(a-c))
  * ``LightDef`` / ``SettingsDef`` / ``CameraDef`` / ``SkyboxDef``   light.rs:17-37, settings.rs:34-60,
                                    camera.rs:50-79, skybox.rs:22-41

Triangle order matters: a triangle's id is its position in ``Object.triangles`` and the nearest-hit tie-break is
"first in that order wins" (object.rs:206), so the loader reproduces the reference's order exactly.

The OBJ/MTL grammar itself lives in a third-party crate that is NOT vendored in the reference (``obj = "0.10"``,
Cargo.toml:63).  ``parse_obj`` / ``parse_mtl`` restate that crate's behaviour as recalled from its published source
(obj-rs 0.10.2, ``ObjData::load_buf`` / ``Mtl::reload``); each recalled rule is marked [crate].  With neither the crate
nor real OBJ assets here (res/**/*.obj are git-LFS stubs) the grammar corner cases are PARITY UNPINNED; the reference's
own code above the crate (this file's ``load_object`` / ``triangulate``) is restated from source.

Image decoding (``image::open(..).into_rgb8()``, object.rs:31-38, skybox.rs:57-64) uses Pillow when it is importable:
PNG and 8-bit TIFF decode losslessly and therefore identically; baseline JPEG decoders may differ by 1 LSB (unpinned);
OpenEXR skyboxes are not decodable here and raise ``SceneLoadError`` like the reference's failed ``image::open``.
"""
from __future__ import annotations

import logging
import math
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

from .scene import (Camera, F32, IlluminationModel, Light, Material, Object, Scene, Settings, Skybox, _norm3)

log = logging.getLogger("raytracing_amd.loader")

F32_EPSILON = np.float32(1.1920929e-07)


class SceneLoadError(ValueError):
    """Any failure of ``Scene::load`` / ``Object::from_obj`` (the reference returns ``anyhow::Error``)."""


# --------------------------------------------------------------------------------------------------------------
# Wavefront OBJ  ([crate] obj 0.10: ObjData::load_buf)
# --------------------------------------------------------------------------------------------------------------
@dataclass
class ObjGroup:
    name: str
    index: int = 0
    material: Optional[str] = None          # ObjMaterial::Ref(name) until load_mtls resolves it
    polys: List[List[Tuple[int, Optional[int], Optional[int]]]] = field(default_factory=list)


@dataclass
class ObjObject:
    name: str
    groups: List[ObjGroup] = field(default_factory=list)


@dataclass
class ObjData:
    position: np.ndarray                    # f32 [n, 3]
    texture: np.ndarray                     # f32 [n, 2]
    normal: np.ndarray                      # f32 [n, 3]
    objects: List[ObjObject]
    material_libs: List[str]


def _logical_lines(text: str):
    """[crate] a trailing backslash joins the next physical line; '#' starts a comment."""
    pending = ""
    for no, raw in enumerate(text.splitlines(), 1):
        line = raw.split("#", 1)[0]
        stripped = line.rstrip()
        if stripped.endswith("\\"):
            pending += stripped[:-1] + " "
            continue
        yield no, pending + line
        pending = ""
    if pending:
        yield no, pending


def _floats(words: List[str], n: int, line_no: int, what: str) -> List[float]:
    if len(words) < n:
        raise SceneLoadError(f"line {line_no}: '{what}' needs {n} numbers, got {len(words)}")
    try:
        return [float(w) for w in words[:n]]
    except ValueError as e:
        raise SceneLoadError(f"line {line_no}: bad number in '{what}': {e}") from None


def _index(tok: str, count: int, line_no: int) -> int:
    """[crate] 1-based; negative counts back from the elements read SO FAR."""
    try:
        i = int(tok)
    except ValueError:
        raise SceneLoadError(f"line {line_no}: bad face index '{tok}'") from None
    j = count + i if i < 0 else i - 1
    if i == 0 or not 0 <= j < count:
        raise SceneLoadError(f"line {line_no}: face index {i} out of range (have {count})")
    return j


def parse_obj(text: str) -> ObjData:
    """Parses OBJ text into the crate's ``ObjData`` shape.  Rules restated [crate]:

    * faces go to the current group of the current object; both are created as "default" on demand;
    * ``o`` closes the current group and object (an object that never received a group is dropped) and starts a new
      object; ``g`` closes the current group and starts a new one -- the NEW GROUP HAS NO MATERIAL until the next
      ``usemtl``;
    * ``usemtl`` on a group that already has a material closes it and continues in a copy with ``index + 1``; on a group
      WITHOUT a material it sets the material of the whole group, faces already in it included;
    * ``mtllib``: the rest of the line, words re-joined by single spaces, is ONE library name;
    * ``s`` and ``l`` are ignored; any other directive is an error (the crate's ``ObjError::UnexpectedCommand``), which is
      how a git-LFS pointer stub in place of a mesh ("version https://git-lfs...") fails the load of the whole scene
      (scene/mod.rs:78-96).  [crate], parity unpinned.
    """
    pos: List[List[float]] = []
    tex: List[List[float]] = []
    nrm: List[List[float]] = []
    objects: List[ObjObject] = []
    libs: List[str] = []
    obj = ObjObject("default")
    group: Optional[ObjGroup] = None

    for no, line in _logical_lines(text):
        words = line.split()
        if not words:
            continue
        key, args = words[0], words[1:]
        if key == "v":
            pos.append(_floats(args, 3, no, "v"))
        elif key == "vt":
            uv = _floats(args, 1, no, "vt") if len(args) == 1 else _floats(args, 2, no, "vt")
            tex.append([uv[0], uv[1] if len(uv) > 1 else 0.0])
        elif key == "vn":
            nrm.append(_floats(args, 3, no, "vn"))
        elif key == "f":
            poly = []
            for tok in args:
                parts = tok.split("/")
                if len(parts) > 3:
                    raise SceneLoadError(f"line {no}: malformed face vertex '{tok}'")
                vi = _index(parts[0], len(pos), no)
                ti = _index(parts[1], len(tex), no) if len(parts) > 1 and parts[1] != "" else None
                ni = _index(parts[2], len(nrm), no) if len(parts) > 2 and parts[2] != "" else None
                poly.append((vi, ti, ni))
            if group is None:
                group = ObjGroup("default")
            group.polys.append(poly)
        elif key == "o":
            if group is not None:
                obj.groups.append(group)
                objects.append(obj)
                group = None
            name = line.split(None, 1)[1].strip() if args else "default"
            obj = ObjObject(name)
        elif key == "g":
            if group is not None:
                obj.groups.append(group)
                group = None
            if args:
                group = ObjGroup(line.split(None, 1)[1].strip())
        elif key == "usemtl":
            if group is None:
                group = ObjGroup("default")
            if group.material is not None:
                obj.groups.append(group)
                group = ObjGroup(group.name, group.index + 1)
            group.material = args[0] if args else None
        elif key == "mtllib":
            if not args:
                raise SceneLoadError(f"line {no}: mtllib without a file name")
            libs.append(" ".join(args))
        elif key in ("s", "l"):
            pass
        else:
            raise SceneLoadError(f"line {no}: unexpected command '{key}'")
    if group is not None:
        obj.groups.append(group)
    objects.append(obj)

    def arr(rows, k):
        return np.asarray(rows, dtype=np.float64).astype(F32).reshape(-1, k)

    return ObjData(arr(pos, 3), arr(tex, 2), arr(nrm, 3), objects, libs)


# --------------------------------------------------------------------------------------------------------------
# MTL  ([crate] obj 0.10: Mtl::reload)
# --------------------------------------------------------------------------------------------------------------
@dataclass
class MtlMaterial:
    name: str
    kd: Optional[List[float]] = None
    ks: Optional[List[float]] = None
    ns: Optional[float] = None
    ni: Optional[float] = None
    d: Optional[float] = None
    illum: Optional[int] = None
    map_kd: Optional[str] = None


def parse_mtl(text: str) -> List[MtlMaterial]:
    """``newmtl`` (first word is the name), Kd, Ks, Ns, Ni, d, illum, map_Kd (first word) [crate]; every other key is
    parsed by the crate but never read by the reference (object.rs:82-108) and is skipped here."""
    mats: List[MtlMaterial] = []
    cur: Optional[MtlMaterial] = None
    for no, line in _logical_lines(text):
        words = line.split()
        if not words:
            continue
        key, args = words[0], words[1:]
        if key == "newmtl":
            if not args:
                raise SceneLoadError(f"line {no}: newmtl without a name")
            cur = MtlMaterial(args[0])
            mats.append(cur)
            continue
        if cur is None:
            continue
        if key == "Kd":
            cur.kd = _floats(args, 3, no, "Kd")
        elif key == "Ks":
            cur.ks = _floats(args, 3, no, "Ks")
        elif key == "Ns":
            cur.ns = _floats(args, 1, no, "Ns")[0]
        elif key == "Ni":
            cur.ni = _floats(args, 1, no, "Ni")[0]
        elif key == "d":
            cur.d = _floats(args, 1, no, "d")[0]
        elif key == "illum":
            try:
                cur.illum = int(args[0])
            except (IndexError, ValueError):
                raise SceneLoadError(f"line {no}: bad illum") from None
        elif key == "map_Kd":
            if not args:
                raise SceneLoadError(f"line {no}: map_Kd without a file name")
            cur.map_kd = args[0]
    return mats


# --------------------------------------------------------------------------------------------------------------
# images
# --------------------------------------------------------------------------------------------------------------
def load_image_rgb8(path: str) -> np.ndarray:
    """``image::open(path)?.into_rgb8()`` -> uint8 [h, w, 3] (object.rs:31-38, skybox.rs:57-64)."""
    try:
        with open(path, "rb") as f:
            magic = f.read(4)
    except OSError as e:
        raise SceneLoadError(f"Failed to load image from path: {path}: {e}") from None
    if magic == b"\x76\x2f\x31\x01":
        return _load_exr_rgb8(path)
    try:
        from PIL import Image
    except ImportError as e:   # pragma: no cover - Pillow is present in the build image
        raise SceneLoadError(f"cannot decode {path}: Pillow is not importable") from e
    try:
        with Image.open(path) as im:
            return np.ascontiguousarray(np.asarray(im.convert("RGB"), dtype=np.uint8))
    except Exception as e:   # noqa: BLE001 - any decoder failure is one load failure
        raise SceneLoadError(f"Failed to load image from path: {path}: {e}") from None


def _load_exr_rgb8(path: str) -> np.ndarray:
    """OpenEXR (the reference's skyboxes): Pillow does not read it, so the C++ loader's decoder does the work
    (include/trayracer_image.hpp, through `examples/trr_cli --decode-image`, which touches no GPU)."""
    import subprocess
    import tempfile
    cli = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "trr_cli")
    if not os.path.exists(cli):
        raise SceneLoadError(f"Failed to load image from path: {path}: EXR needs examples/trr_cli (make -C raytracing_amd/csrc)")
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "image.ppm")
        r = subprocess.run([cli, "--decode-image", path, out], capture_output=True, text=True)
        if r.returncode != 0:
            raise SceneLoadError(f"Failed to load image from path: {path}: {(r.stderr or r.stdout).strip()}")
        with open(out, "rb") as f:
            d = f.read()
    head = d.split(b"\n", 3)   # "P6", "w h", "255", pixels
    if len(head) != 4 or head[0] != b"P6":
        raise SceneLoadError(f"Failed to load image from path: {path}: unexpected decoder output")
    w, h = (int(x) for x in head[1].split())
    return np.frombuffer(head[3], np.uint8, w * h * 3).reshape(h, w, 3).copy()


# --------------------------------------------------------------------------------------------------------------
# Object::from_obj
# --------------------------------------------------------------------------------------------------------------
def object_name(path: str) -> str:
    """``filename`` (object.rs:41-54): file name up to the first '.', first character ASCII-uppercased."""
    stem = os.path.basename(path).split(".")[0]
    return stem[:1].upper() + stem[1:] if stem[:1].isascii() else stem


def triangulate(data: ObjData, polys, poly_material: np.ndarray) -> Tuple[np.ndarray, np.ndarray, Tuple[int, int, int]]:
    """object.rs:224-296 for a list of polygons at once: fan (0, i, i+1) per polygon in order; vertex normals
    where the face names them, else the computed normal ((a-b)x(a-c), normalised, zero if its norm <= f32::EPSILON);
    uv where named, else (0, 0).  fp32 in nalgebra's operation order.  Returns ([n, 24] f32, [n] i32 material, warnings)."""
    i0, i1, i2, mat = [], [], [], []
    for poly, m in zip(polys, poly_material):
        n = len(poly)
        if n == 0:
            raise SceneLoadError("face without vertices (the reference underflows `len - 1`, object.rs:232)")
        for i in range(1, n - 1):
            i0.append(poly[0])
            i1.append(poly[i])
            i2.append(poly[i + 1])
            mat.append(m)
    nt = len(i0)
    out = np.zeros((nt, 24), F32)
    if nt == 0:
        return out, np.zeros(0, np.int32), (0, 0, 0)

    def col(rows, k):
        return np.array([-1 if r[k] is None else r[k] for r in rows], np.int64)

    corners = (i0, i1, i2)
    P = [data.position[col(c, 0)] for c in corners]
    a, b, c = P
    e1 = (a - b).astype(F32)
    e2 = (a - c).astype(F32)
    cr = np.stack([e1[:, 1] * e2[:, 2] - e1[:, 2] * e2[:, 1],
                   e1[:, 2] * e2[:, 0] - e1[:, 0] * e2[:, 2],
                   e1[:, 0] * e2[:, 1] - e1[:, 1] * e2[:, 0]], axis=1).astype(F32)
    # norm(): sqrt(x*x + y*y + z*z) left to right in f32 (nalgebra norm_squared fold)
    n2 = (cr[:, 0] * cr[:, 0] + cr[:, 1] * cr[:, 1]).astype(F32)
    n2 = (n2 + cr[:, 2] * cr[:, 2]).astype(F32)
    nn = np.sqrt(n2, dtype=F32)
    ok = nn > F32_EPSILON                    # try_normalize(min_norm): None when norm <= min_norm
    comp = np.zeros_like(cr)
    comp[ok] = (cr[ok] / nn[ok, None]).astype(F32)
    zero_normals = int((~ok).sum())

    out[:, 0:3], out[:, 3:6], out[:, 6:9] = a, b, c
    no_normals = no_uv = 0
    for k, cn in enumerate(corners):
        ni = col(cn, 2)
        have = ni >= 0
        no_normals += int((~have).sum())
        nv = comp.copy()
        if have.any():
            nv[have] = data.normal[ni[have]]
        out[:, 9 + 3 * k:12 + 3 * k] = nv
        ti = col(cn, 1)
        have = ti >= 0
        no_uv += int((~have).sum())
        uv = np.zeros((nt, 2), F32)
        if have.any():
            uv[have] = data.texture[ti[have]]
        out[:, 18 + 2 * k:20 + 2 * k] = uv
    return out, np.asarray(mat, np.int32), (zero_normals, no_normals, no_uv)


def load_object(path: str, translation=(0, 0, 0), rotation=(0, 0, 0, 1), scale=(1, 1, 1)) -> Object:
    """``Object::from_obj`` (object.rs:61-175).  ``rotation`` is the unit quaternion (i, j, k, w)."""
    try:
        with open(path, "r", encoding="utf-8", errors="replace") as f:
            data = parse_obj(f.read())
    except OSError as e:
        raise SceneLoadError(f"Failed to load obj from path: {path}: {e}") from None
    except SceneLoadError as e:
        raise SceneLoadError(f"Failed to load obj from path: {path}: {e}") from None

    base = os.path.dirname(path)
    mtl: List[MtlMaterial] = []
    for lib in data.material_libs:                   # load_mtls(): every library, relative to the OBJ (object.rs:72-75)
        p = os.path.join(base, lib)
        try:
            with open(p, "r", encoding="utf-8", errors="replace") as f:
                mtl.extend(parse_mtl(f.read()))
        except (OSError, SceneLoadError) as e:
            raise SceneLoadError(f"Failed to load materials from obj path: {path}: {lib}: {e}") from None

    materials: List[Material] = []
    for m in mtl:                                    # object.rs:77-109
        tex = None
        if m.map_kd is not None:
            tp = os.path.join(base, m.map_kd)
            try:
                tex = load_image_rgb8(tp)
            except SceneLoadError as e:              # object.rs:92-96: warn, carry on without the texture
                log.warning("Failed to load texture from path: %s: %s", tp, e)
        illum = IlluminationModel.from_i32(m.illum) if m.illum is not None else None
        if illum is None:
            log.warning("Invalid illumination model: %s", -1 if m.illum is None else m.illum)
            illum = IlluminationModel()
        materials.append(Material(
            name=m.name,
            diffuse_color=None if m.kd is None else np.asarray(m.kd, np.float64).astype(F32),
            specular_color=None if m.ks is None else np.asarray(m.ks, np.float64).astype(F32),
            specular_exponent=None if m.ns is None else float(F32(m.ns)),
            diffuse_texture=tex,
            illumination_model=illum,
            dissolve=None if m.d is None else float(F32(1.0) - F32(m.d)),     # object.rs:106
            refraction_index=None if m.ni is None else float(F32(m.ni)),
        ))

    known = {m.name for m in mtl}
    first_pos: Dict[str, int] = {}
    for i, m in enumerate(materials):
        first_pos.setdefault(m.name, i)              # `.position(|mat| mat.name == m.name)`: the FIRST of that name

    polys, poly_mat = [], []
    for o in data.objects:                           # object.rs:111-142
        for g in o.groups:
            mi = -1
            if g.material is not None:
                if g.material not in known:          # stays ObjMaterial::Ref -> the reference panics (object.rs:121-123)
                    raise SceneLoadError(f"Material reference not supported: {g.material}")
                mi = first_pos[g.material]
            polys.extend(g.polys)
            poly_mat.extend([mi] * len(g.polys))
    tris, mat_idx, warn = triangulate(data, polys, np.asarray(poly_mat, np.int32))
    if warn[0]:
        log.warning("Computed normals for %d triangles is zero", warn[0])
    if warn[1]:
        log.warning("No normals for %d triangles", warn[1])
    if warn[2]:
        log.warning("No UV for %d triangles", warn[2])

    return Object(name=object_name(path), triangles=tris, material_index=mat_idx, materials=materials,
                  translation=translation, rotation=rotation, scale=scale,
                  material_name=", ".join(o.name for o in data.objects), path=path)


# --------------------------------------------------------------------------------------------------------------
# YAML  (Scene::load)
# --------------------------------------------------------------------------------------------------------------
def _yaml_loader():
    import yaml

    class L(yaml.SafeLoader):
        pass

    # serde_yml writes the externally tagged SkyboxDef (skybox.rs:22-26) as `!Color [r, g, b]` / `!Path file`
    L.add_constructor("!Color", lambda ld, node: {"Color": ld.construct_sequence(node, deep=True)})
    L.add_constructor("!Path", lambda ld, node: {"Path": ld.construct_scalar(node)})
    return yaml, L


def _num(m, key, where) -> float:
    if not isinstance(m, dict) or key not in m:
        raise SceneLoadError(f"{where}: missing field `{key}`")
    v = m[key]
    if isinstance(v, bool) or not isinstance(v, (int, float)):
        raise SceneLoadError(f"{where}.{key}: invalid type, expected f32")
    return float(v)


def _xyz(m, key, where, names=("x", "y", "z")) -> np.ndarray:
    if not isinstance(m, dict) or key not in m:
        raise SceneLoadError(f"{where}: missing field `{key}`")
    return np.array([_num(m[key], n, f"{where}.{key}") for n in names], np.float64).astype(F32)


def _uint(m, key, where) -> int:
    if not isinstance(m, dict) or key not in m:
        raise SceneLoadError(f"{where}: missing field `{key}`")
    v = m[key]
    if isinstance(v, bool) or not isinstance(v, int) or not 0 <= v < 2 ** 32:
        raise SceneLoadError(f"{where}.{key}: invalid type, expected u32")
    return v


def _normalize0(v: np.ndarray) -> Tuple[np.ndarray, np.float32]:
    """``try_normalize(0.0).unwrap_or_default()`` and the norm (light.rs:33, settings.rs:51-55)."""
    n = _norm3(v)
    return ((v / n).astype(F32) if n > 0 else np.zeros(3, F32)), n


def _settings_from_yaml(m, cwd: Optional[str]) -> Settings:
    """SettingsDef (settings.rs:34-60): every field is required; any failure makes the caller fall back to defaults."""
    where = "extraArgs"
    if not isinstance(m, dict):
        raise SceneLoadError(f"{where}: invalid type, expected struct SettingsDef")
    max_bounces = _uint(m, "max_bounces", where)
    samples = _uint(m, "samples", where)
    col, norm = _normalize0(_xyz(m, "ambient_color", where, ("r", "g", "b")))
    if "skybox" not in m:
        raise SceneLoadError(f"{where}: missing field `skybox`")
    sb = m["skybox"]
    if not isinstance(sb, dict) or len(sb) != 1 or next(iter(sb)) not in ("Color", "Path"):
        raise SceneLoadError(f"{where}.skybox: expected `!Color [r, g, b]` or `!Path file`")
    kind, val = next(iter(sb.items()))
    if kind == "Color":
        if not isinstance(val, (list, tuple)) or len(val) != 3:
            raise SceneLoadError(f"{where}.skybox: Color needs 3 numbers")
        skybox = Skybox(color=np.asarray([float(x) for x in val], np.float64).astype(F32))
    else:
        # skybox.rs:57-64: the path is used AS WRITTEN (relative to the process's working directory, not the YAML)
        p = str(val)
        try:
            skybox = Skybox(image=load_image_rgb8(p if cwd is None else os.path.join(cwd, p)), path=p)
        except SceneLoadError as e:
            raise SceneLoadError(f"Failed to load skybox: {e}") from None
    if "anti_aliasing" not in m or not isinstance(m["anti_aliasing"], bool):
        raise SceneLoadError(f"{where}: missing or non-bool field `anti_aliasing`")
    return Settings(max_bounces=max_bounces, samples=samples, ambient_color=col, ambient_intensity=float(norm),
                    skybox=skybox, anti_aliasing=m["anti_aliasing"])


def scene_from_yaml(text: str, path: str = "", cwd: Optional[str] = None) -> Scene:
    """``WithRelativePath::deserialize`` (mod.rs:64-123).  ``path`` is the YAML's own path: model files are resolved
    against its parent (object.rs:343-348)."""
    yaml, L = _yaml_loader()
    try:
        doc = yaml.load(text, Loader=L)
    except yaml.YAMLError as e:
        raise SceneLoadError(f"Failed to deserialize scene from path: {path}\n{e}") from None
    if not isinstance(doc, dict):
        raise SceneLoadError(f"Failed to deserialize scene from path: {path}\nexpected a mapping at the top level")
    base = os.path.dirname(path)

    def seq(key):
        if key not in doc:
            raise SceneLoadError(f"missing field `{key}`")
        if not isinstance(doc[key], list):
            raise SceneLoadError(f"`{key}`: invalid type, expected a sequence")
        return doc[key]

    try:
        objects = []
        for i, m in enumerate(seq("models")):
            where = f"models[{i}]"
            if not isinstance(m, dict) or "filePath" not in m:
                raise SceneLoadError(f"{where}: missing field `filePath`")
            pos = _xyz(m, "position", where)
            rot = _xyz(m, "rotation", where)
            scl = _xyz(m, "scale", where)
            file_path = str(m["filePath"])
            o = load_object(os.path.join(base, file_path), translation=pos,
                            rotation=Object.quaternion_from_euler_degrees(rot[0], rot[1], rot[2]), scale=scl)
            o.path = file_path                     # object.rs:352-355: the path as written in the YAML
            objects.append(o)

        lights = []
        for i, m in enumerate(seq("pointLights")):
            where = f"pointLights[{i}]"
            col, _ = _normalize0(_xyz(m, "Ke", where, ("r", "g", "b")))
            lights.append(Light(_xyz(m, "position", where), col, float(F32(_num(m, "intensity", where)))))

        if "camera" not in doc:
            raise SceneLoadError("missing field `camera`")
        c = doc["camera"]
        deg = F32(_num(c, "fieldOfView", "camera"))
        # f32::to_radians: self * (PI_f32 / 180_f32), the quotient rounded to f32 once (camera.rs:75)
        fov = F32(deg * F32(F32(np.pi) / F32(180.0)))
        camera = Camera(position=_xyz(c, "position", "camera"), look_at=_xyz(c, "lookAt", "camera"),
                        up=_xyz(c, "upVec", "camera"), fov=float(fov),
                        resolution=(_uint(c, "width", "camera"), _uint(c, "height", "camera")))

        settings = Settings()
        if "extraArgs" in doc:                       # mod.rs:100-110: failure or absence -> Settings::default()
            try:
                settings = _settings_from_yaml(doc["extraArgs"], cwd)
            except SceneLoadError as e:
                log.warning("Failed to deserialize extraArgs: %s", e)
    except SceneLoadError as e:
        raise SceneLoadError(f"Failed to deserialize scene from path: {path}\n{e}") from None
    return Scene(objects=objects, lights=lights, camera=camera, settings=settings, path=path)


def load_scene(path: str) -> Scene:
    """``Scene::load`` (mod.rs:125-140)."""
    try:
        with open(path, "r", encoding="utf-8") as f:
            text = f.read()
    except OSError as e:
        raise SceneLoadError(f"Failed to read file from path: {path}: {e}") from None
    return scene_from_yaml(text, path)


# --------------------------------------------------------------------------------------------------------------
# Serialize (the reference's "save scene": Serialize impls in mod.rs:32-42, object.rs:359-377, light.rs:40-52,
# camera.rs:82-97, settings.rs:63-77, skybox.rs:43-54)
# --------------------------------------------------------------------------------------------------------------
def _euler_degrees(q: np.ndarray) -> Tuple[float, float, float]:
    """UnitQuaternion::euler_angles() (roll, pitch, yaw) in degrees; editor round trip only, never reaches a pixel."""
    i, j, k, w = (float(x) for x in q)
    r00 = 1 - 2 * (j * j + k * k)
    r10 = 2 * (i * j + w * k)
    r20 = 2 * (i * k - w * j)
    r21 = 2 * (j * k + w * i)
    r22 = 1 - 2 * (i * i + j * j)
    r01 = 2 * (i * j - w * k)
    r11 = 1 - 2 * (i * i + k * k)
    if abs(r20) < 1.0:
        roll, pitch, yaw = math.atan2(r21, r22), -math.asin(r20), math.atan2(r10, r00)
    elif r20 <= -1.0:
        roll, pitch, yaw = math.atan2(r01, r11), math.pi / 2, 0.0
    else:
        roll, pitch, yaw = math.atan2(-r01, r11), -math.pi / 2, 0.0
    return tuple(float(F32(math.degrees(a))) for a in (roll, pitch, yaw))


def scene_to_yaml(scene: Scene) -> str:
    """YAML in the reference's schema; ``scene_from_yaml`` reads it back (image skyboxes need ``Skybox.path``)."""
    def f(x):
        return float(F32(x))

    def xyz(v, names=("x", "y", "z")):
        return {n: f(v[i]) for i, n in enumerate(names)}

    import yaml
    models = []
    for o in scene.objects:
        rx, ry, rz = _euler_degrees(o.rotation)
        models.append({"filePath": o.path, "position": xyz(o.translation), "rotation": {"x": rx, "y": ry, "z": rz},
                       "scale": xyz(o.scale)})
    lights = [{"position": xyz(l.position), "Ke": xyz(l.color, ("r", "g", "b")), "intensity": f(l.intensity)}
              for l in scene.lights]
    cam = scene.camera
    camera = {"position": xyz(cam.position), "lookAt": xyz(cam.look_at), "upVec": xyz(cam.up),
              "fieldOfView": f(F32(cam.fov) * F32(57.29577951308232)), "width": int(cam.resolution[0]),
              "height": int(cam.resolution[1])}
    s = scene.settings
    amb = (np.asarray(s.ambient_color, F32) * F32(s.ambient_intensity)).astype(F32)
    body = yaml.safe_dump({"models": models, "pointLights": lights, "camera": camera}, sort_keys=False)
    extra = yaml.safe_dump({"max_bounces": int(s.max_bounces), "samples": int(s.samples),
                            "ambient_color": xyz(amb, ("r", "g", "b"))}, sort_keys=False)
    if s.skybox.image is not None:
        sky = f"!Path {s.skybox.path}"
    else:
        sky = "!Color [" + ", ".join(repr(f(x)) for x in s.skybox.color) + "]"
    extra += f"skybox: {sky}\nanti_aliasing: {'true' if s.anti_aliasing else 'false'}\n"
    return body + "extraArgs:\n" + "".join("  " + ln + "\n" for ln in extra.splitlines())
