"""ctypes binding of the CPU oracle (oracle/trayracer_oracle.cpp).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
Nothing under raytracing_amd/ imports this module.  Parity is UNPINNED by the reference (it has no
tests or golden vectors, and it cannot be built here) -- see the header of trayracer_oracle.cpp.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

ACCEL_BRUTE, ACCEL_BVH, ACCEL_BRUTE_NOBOX = 0, 1, 2
SHADE_LITERAL, SHADE_HOISTED = 0, 1

HAS_KD, HAS_KS, HAS_NS, HAS_DISSOLVE = 1, 2, 4, 8


class _Material(C.Structure):
    _fields_ = [("kd", C.c_float * 3), ("ks", C.c_float * 3), ("ns", C.c_float), ("dissolve", C.c_float),
                ("illum", C.c_int32), ("texture", C.c_int32), ("flags", C.c_uint32)]


class _Object(C.Structure):
    _fields_ = [("tri", C.c_void_p), ("material", C.c_void_p), ("n_tris", C.c_uint32), ("mats", C.c_void_p),
                ("n_mats", C.c_uint32), ("translation", C.c_float * 3), ("rotation", C.c_float * 4),
                ("scale", C.c_float * 3)]


class _Image(C.Structure):
    _fields_ = [("rgb", C.c_void_p), ("w", C.c_uint32), ("h", C.c_uint32)]


class _Light(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("color", C.c_float * 3), ("intensity", C.c_float)]


class _Settings(C.Structure):
    _fields_ = [("ambient_color", C.c_float * 3), ("ambient_intensity", C.c_float), ("sky_is_image", C.c_int32),
                ("sky_color", C.c_float * 3), ("sky", _Image), ("aa", C.c_int32), ("samples", C.c_uint32),
                ("seed", C.c_uint64)]


class _Camera(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("look_at", C.c_float * 3), ("up", C.c_float * 3), ("fov", C.c_float),
                ("w", C.c_uint32), ("h", C.c_uint32)]


class Counters(C.Structure):
    _fields_ = [("raycasts", C.c_uint64), ("node_visits", C.c_uint64), ("tri_tests", C.c_uint64),
                ("shade_calls", C.c_uint64), ("seconds", C.c_double)]


def build(force: bool = False) -> str:
    """Compile liboracle.so with oracle/Makefile (gcc, -ffp-contract=off)."""
    src = os.path.join(_HERE, "trayracer_oracle.cpp")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.orc_scene_create.restype = C.c_void_p
        L.orc_scene_create.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32,
                                       C.c_void_p, C.c_void_p]
        L.orc_scene_destroy.argtypes = [C.c_void_p]
        L.orc_scene_set_camera.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_scene_set_aa.argtypes = [C.c_void_p, C.c_int32, C.c_uint32, C.c_uint64]
        L.orc_render.restype = C.c_int
        L.orc_render.argtypes = [C.c_void_p, C.c_float, C.c_uint32, C.c_int, C.c_int, C.c_uint32, C.c_uint32,
                                 C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_triangle_intersect.restype = C.c_int
        L.orc_triangle_intersect.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
        L.orc_triangle_aabb.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_ray_intersects_aabb.restype = C.c_int
        L.orc_ray_intersects_aabb.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_camera_ray.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        L.orc_pixel_ndc.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        L.orc_quantize.restype = C.c_uint8
        L.orc_quantize.argtypes = [C.c_float]
        L.orc_reflect.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_object_matrices.restype = C.c_int
        L.orc_object_matrices.argtypes = [C.c_void_p] * 5
        L.orc_skybox.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_rand_f32.restype = C.c_float
        L.orc_rand_f32.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
        L.orc_raycast.restype = C.c_int
        L.orc_raycast.argtypes = [C.c_void_p, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def _f3(dst, src):
    for i in range(len(dst)):
        dst[i] = float(src[i])


def _ptr(a: np.ndarray) -> int:
    return a.ctypes.data


def _camera_struct(cam) -> _Camera:
    c = _Camera()
    _f3(c.pos, cam.position)
    _f3(c.look_at, cam.look_at)
    _f3(c.up, cam.up)
    c.fov = float(cam.fov)
    c.w, c.h = int(cam.resolution[0]), int(cam.resolution[1])
    return c


class OracleScene:
    """Owns an oracle-side copy of a raytracing_amd.scene.Scene."""

    def __init__(self, scene):
        L = lib()
        self._keep = []
        textures = []   # list of uint8 [h, w, 3]
        objs = (_Object * max(1, len(scene.objects)))()
        for oi, o in enumerate(scene.objects):
            tri = np.ascontiguousarray(o.triangles, np.float32)
            mi = np.ascontiguousarray(o.material_index, np.int32)
            mats = (_Material * max(1, len(o.materials)))()
            for k, m in enumerate(o.materials):
                flags = 0
                if m.diffuse_color is not None:
                    flags |= HAS_KD
                    _f3(mats[k].kd, m.diffuse_color)
                if m.specular_color is not None:
                    flags |= HAS_KS
                    _f3(mats[k].ks, m.specular_color)
                if m.specular_exponent is not None:
                    flags |= HAS_NS
                    mats[k].ns = float(m.specular_exponent)
                if m.dissolve is not None:
                    flags |= HAS_DISSOLVE
                    mats[k].dissolve = float(m.dissolve)
                mats[k].illum = int(m.illumination_model.value)
                mats[k].texture = -1
                if m.diffuse_texture is not None:
                    mats[k].texture = len(textures)
                    textures.append(np.ascontiguousarray(m.diffuse_texture, np.uint8))
                mats[k].flags = flags
            self._keep += [tri, mi, mats]
            objs[oi].tri = _ptr(tri)
            objs[oi].material = _ptr(mi)
            objs[oi].n_tris = len(tri)
            objs[oi].mats = C.addressof(mats)
            objs[oi].n_mats = len(o.materials)
            _f3(objs[oi].translation, o.translation)
            _f3(objs[oi].rotation, o.rotation)
            _f3(objs[oi].scale, o.scale)
        imgs = (_Image * max(1, len(textures)))()
        for k, t in enumerate(textures):
            imgs[k].rgb = _ptr(t)
            imgs[k].h, imgs[k].w = t.shape[0], t.shape[1]
        self._keep.append(textures)
        lights = (_Light * max(1, len(scene.lights)))()
        for k, l in enumerate(scene.lights):
            _f3(lights[k].pos, l.position)
            _f3(lights[k].color, l.color)
            lights[k].intensity = float(l.intensity)
        st = _Settings()
        s = scene.settings
        _f3(st.ambient_color, s.ambient_color)
        st.ambient_intensity = float(s.ambient_intensity)
        if s.skybox.image is not None:
            sky = np.ascontiguousarray(s.skybox.image, np.uint8)
            self._keep.append(sky)
            st.sky_is_image = 1
            st.sky.rgb = _ptr(sky)
            st.sky.h, st.sky.w = sky.shape[0], sky.shape[1]
        else:
            st.sky_is_image = 0
            _f3(st.sky_color, s.skybox.color)
        st.aa = 1 if s.anti_aliasing else 0
        st.samples = int(s.samples)
        st.seed = int(s.seed)
        cam = _camera_struct(scene.camera)
        self.width, self.height = cam.w, cam.h
        self.handle = L.orc_scene_create(C.addressof(objs), len(scene.objects), C.addressof(imgs), len(textures),
                                         C.addressof(lights), len(scene.lights), C.addressof(st), C.addressof(cam))

    def set_camera(self, cam):
        c = _camera_struct(cam)
        self.width, self.height = c.w, c.h
        lib().orc_scene_set_camera(self.handle, C.addressof(c))

    def set_aa(self, aa: bool, samples: int, seed: int):
        lib().orc_scene_set_aa(self.handle, 1 if aa else 0, int(samples), int(seed))

    def render(self, delta: float = 1e-5, max_depth: int = 5, accel: int = ACCEL_BVH, shade: int = SHADE_LITERAL,
               threads: int = 0, block_first: int = 0, block_step: int = 1, block_limit: int = 0xFFFFFFFF,
               want_ids: bool = True, want_f32: bool = False):
        """Full frame like RenderingThread::run (render.rs:95-193).  Returns dict(rgb, hit_obj, hit_tri, counters)."""
        w, h = self.width, self.height
        rgb = np.zeros((h, w, 3), np.uint8)
        f32 = np.zeros((h, w, 3), np.float32) if want_f32 else None
        ho = np.zeros((h, w), np.uint32) if want_ids else None
        ht = np.zeros((h, w), np.uint32) if want_ids else None
        cn = Counters()
        rc = lib().orc_render(self.handle, float(delta), int(max_depth), int(accel), int(shade), int(threads),
                              int(block_first), int(block_step), int(block_limit), _ptr(rgb),
                              _ptr(f32) if want_f32 else None, _ptr(ho) if want_ids else None,
                              _ptr(ht) if want_ids else None, C.addressof(cn))
        if rc != 0:
            raise ZeroDivisionError("resolution below 20 pixels: the reference divides by zero (render.rs:109)")
        return {"rgb": rgb, "rgb_f32": f32, "hit_obj": ho, "hit_tri": ht, "counters": cn}

    def raycast(self, origin, direction, delta: float = 1e-5, accel: int = ACCEL_BVH):
        o = np.asarray(origin, np.float32)
        d = np.asarray(direction, np.float32)
        obj, tri = C.c_uint32(), C.c_uint32()
        p, n = np.zeros(3, np.float32), np.zeros(3, np.float32)
        hit = lib().orc_raycast(self.handle, float(delta), int(accel), _ptr(o), _ptr(d), C.addressof(obj),
                                C.addressof(tri), _ptr(p), _ptr(n))
        return (obj.value, tri.value, p, n) if hit else None

    def skybox(self, direction):
        d = np.asarray(direction, np.float32)
        out = np.zeros(3, np.float32)
        lib().orc_skybox(self.handle, _ptr(d), _ptr(out))
        return out

    def close(self):
        if self.handle:
            lib().orc_scene_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# --- unit-level helpers for the known-answer tests ------------------------------------------------
def triangle_intersect(tri9, o, d, delta=1e-5) -> Optional[np.ndarray]:
    t = np.asarray(tri9, np.float32).reshape(9)
    o = np.asarray(o, np.float32)
    d = np.asarray(d, np.float32)
    out = np.zeros(3, np.float32)
    ok = lib().orc_triangle_intersect(_ptr(t), _ptr(o), _ptr(d), float(delta), _ptr(out))
    return out if ok else None


def triangle_aabb(tri9) -> np.ndarray:
    t = np.asarray(tri9, np.float32).reshape(9)
    out = np.zeros(6, np.float32)
    lib().orc_triangle_aabb(_ptr(t), _ptr(out))
    return out


def ray_intersects_aabb(o, d, box6) -> bool:
    o = np.asarray(o, np.float32)
    d = np.asarray(d, np.float32)
    b = np.asarray(box6, np.float32)
    return bool(lib().orc_ray_intersects_aabb(_ptr(o), _ptr(d), _ptr(b)))


def camera_ray(cam, x: float, y: float):
    c = _camera_struct(cam)
    o, d = np.zeros(3, np.float32), np.zeros(3, np.float32)
    lib().orc_camera_ray(C.addressof(c), float(x), float(y), _ptr(o), _ptr(d))
    return o, d


def pixel_ndc(x, y, w, h) -> np.ndarray:
    out = np.zeros(2, np.float32)
    lib().orc_pixel_ndc(int(x), int(y), int(w), int(h), _ptr(out))
    return out


def quantize(c: float) -> int:
    return int(lib().orc_quantize(float(c)))


def reflect(i, n) -> np.ndarray:
    i = np.asarray(i, np.float32)
    n = np.asarray(n, np.float32)
    out = np.zeros(3, np.float32)
    lib().orc_reflect(_ptr(i), _ptr(n), _ptr(out))
    return out


def object_matrices(translation, rotation, scale):
    t = np.asarray(translation, np.float32)
    r = np.asarray(rotation, np.float32)
    s = np.asarray(scale, np.float32)
    f, i = np.zeros(16, np.float32), np.zeros(16, np.float32)
    ok = lib().orc_object_matrices(_ptr(t), _ptr(r), _ptr(s), _ptr(f), _ptr(i))
    return f.reshape(4, 4).T.copy(), (i.reshape(4, 4).T.copy() if ok else None)   # returned row-major [r][c]


def rand_f32(seed, pixel, sample, dim) -> float:
    return float(lib().orc_rand_f32(int(seed), int(pixel), int(sample), int(dim)))
