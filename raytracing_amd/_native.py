"""ctypes binding of libtrayracer_hip.so (the C ABI of include/trayracer_hip.h).

There is NO fallback: if the HIP library is missing or a call fails, this module raises.  Nothing here
imports or calls oracle/.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtrayracer_hip.so")

TRR_OK = 0
TRR_ERR_INVALID = -1
TRR_ERR_RESOLUTION = -2
TRR_ERR_DEVICE = -3
TRR_ERR_CANCELLED = -4
TRR_ERR_NOMEM = -5

TRR_MAT_HAS_KD, TRR_MAT_HAS_KS, TRR_MAT_HAS_NS, TRR_MAT_HAS_DISSOLVE = 1, 2, 4, 8
TRR_FLAG_COUNTERS = 1
TRR_FLAG_DEFER_STATS = 2
TRR_FLAG_PIPELINED = 4


class MaterialDesc(C.Structure):
    _fields_ = [("kd", C.c_float * 3), ("ks", C.c_float * 3), ("ns", C.c_float), ("dissolve", C.c_float),
                ("illum", C.c_int32), ("texture", C.c_int32), ("flags", C.c_uint32)]


class ImageRgb8(C.Structure):
    _fields_ = [("rgb", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32)]


class ObjectDesc(C.Structure):
    _fields_ = [("triangles", C.c_void_p), ("material_index", C.c_void_p), ("n_triangles", C.c_uint32),
                ("materials", C.c_void_p), ("n_materials", C.c_uint32), ("transform", C.c_float * 16),
                ("inverse", C.c_float * 16)]


class LightDesc(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("color", C.c_float * 3), ("intensity", C.c_float)]


class SettingsDesc(C.Structure):
    _fields_ = [("ambient_color", C.c_float * 3), ("ambient_intensity", C.c_float), ("skybox_is_image", C.c_int32),
                ("skybox_color", C.c_float * 3), ("skybox_image", ImageRgb8), ("anti_aliasing", C.c_int32),
                ("samples", C.c_uint32), ("seed", C.c_uint64)]


class CameraDesc(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("look_at", C.c_float * 3), ("up", C.c_float * 3), ("fov", C.c_float),
                ("width", C.c_uint32), ("height", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("rays_primary", C.c_uint64), ("rays_shadow", C.c_uint64), ("rays_reflection", C.c_uint64),
                ("rays_march", C.c_uint64), ("node_visits", C.c_uint64), ("tri_tests", C.c_uint64),
                ("box_tests", C.c_uint64), ("ms_total", C.c_double), ("ms_trace", C.c_double), ("ms_shade", C.c_double),
                ("trace_launches", C.c_uint32), ("blocks_rendered", C.c_uint32),
                ("wave_node_steps", C.c_uint64), ("wave_tri_steps", C.c_uint64), ("wave_outer_steps", C.c_uint64),
                ("wave_lane_steps", C.c_uint64), ("wave_trans_steps", C.c_uint64), ("trans_lanes", C.c_uint64),
                ("wave_uniform_node_steps", C.c_uint64), ("queue_overflows", C.c_uint64),
                ("rays_shadow_culled", C.c_uint64), ("rays_primary_culled", C.c_uint64)]

    def rays(self) -> int:
        return self.rays_primary + self.rays_shadow + self.rays_reflection + self.rays_march

    def as_dict(self) -> dict:
        return {n: getattr(self, n) for n, _ in self._fields_}


class SceneInfo(C.Structure):
    _fields_ = [("n_triangles", C.c_uint64), ("n_triangles_live", C.c_uint64), ("n_nodes", C.c_uint64),
                ("bvh_depth", C.c_uint32), ("stack_need", C.c_uint32), ("build", C.c_uint32), ("pad", C.c_uint32),
                ("ms_build", C.c_double)]

    def as_dict(self) -> dict:
        return {n: getattr(self, n) for n, _ in self._fields_}


TRR_BUILD_HOST = 0
TRR_BUILD_DEVICE = 1

PROGRESS_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32, C.c_uint32)
BLOCK_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint8))

# every symbol include/trayracer_hip.h declares
EXPORTS = [
    "trr_create", "trr_create_multi", "trr_device_count", "trr_shard_blocks", "trr_destroy", "trr_last_error", "trr_scene_create", "trr_scene_create_ex", "trr_scene_get_info",
    "trr_scene_validate", "trr_scene_destroy", "trr_scene_set_instances",
    "trr_scene_set_lights", "trr_scene_set_settings", "trr_scene_set_camera", "trr_render", "trr_render_blocks", "trr_render_shard_device",
    "trr_strip_bytes", "trr_render_shard_device_noalloc", "trr_noalloc_scratch_bytes", "trr_assemble_device", "trr_synchronize", "trr_collect_stats", "trr_object_matrices", "trr_camera_ray", "trr_version", "trr_hw_queues",
]


class NativeError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"trayracer_hip error {code}: {message}")
        self.code = code


_lib = None


def lib() -> C.CDLL:
    """Loads the HIP library; raises if it has not been built (python __graft_entry__.py / make -C raytracing_amd/csrc)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `make -C raytracing_amd/csrc` "
                          "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    # One HIP runtime per process: the PyTorch-ROCm wheel bundles its own libamdhip64.so.7 / libhsa-runtime64, and a
    # process that initialises /opt/rocm's copy first and torch's second ends up with two HSA runtimes (torch then
    # reports "No HIP GPUs are available").  Importing torch first makes the dynamic loader bind this library's
    # NEEDED libamdhip64.so.7 to the copy torch already loaded, so device pointers and streams can be shared.
    if os.environ.get("TRR_NO_TORCH_PRELOAD", "0") != "1":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    L = C.CDLL(LIB_PATH)
    vp, u32, i32, f32 = C.c_void_p, C.c_uint32, C.c_int, C.c_float
    L.trr_version.restype = C.c_char_p
    L.trr_hw_queues.restype = C.c_int
    L.trr_last_error.restype = C.c_char_p
    L.trr_last_error.argtypes = [vp]
    L.trr_create.restype = i32
    L.trr_create.argtypes = [i32, C.POINTER(vp)]
    if hasattr(L, "trr_create_multi"):
        L.trr_create_multi.restype = i32
        L.trr_create_multi.argtypes = [vp, i32, C.POINTER(vp)]
        L.trr_device_count.restype = i32
        L.trr_device_count.argtypes = [vp]
        L.trr_shard_blocks.restype = u32
        L.trr_shard_blocks.argtypes = [u32, u32, u32, u32, vp, u32]
    L.trr_destroy.restype = None
    L.trr_destroy.argtypes = [vp]
    L.trr_scene_create.restype = i32
    L.trr_scene_create.argtypes = [vp, vp, u32, vp, u32, C.POINTER(vp)]
    L.trr_scene_create_ex.restype = i32
    L.trr_scene_create_ex.argtypes = [vp, vp, u32, vp, u32, u32, C.POINTER(vp)]
    L.trr_scene_get_info.restype = i32
    L.trr_scene_get_info.argtypes = [vp, vp]
    L.trr_scene_validate.restype = i32
    L.trr_scene_validate.argtypes = [vp]
    L.trr_scene_destroy.restype = None
    L.trr_scene_destroy.argtypes = [vp]
    L.trr_scene_set_instances.restype = i32
    L.trr_scene_set_instances.argtypes = [vp, vp, vp, u32]
    L.trr_scene_set_lights.restype = i32
    L.trr_scene_set_lights.argtypes = [vp, vp, u32]
    L.trr_scene_set_settings.restype = i32
    L.trr_scene_set_settings.argtypes = [vp, vp]
    L.trr_scene_set_camera.restype = i32
    L.trr_scene_set_camera.argtypes = [vp, vp]
    L.trr_render.restype = i32
    L.trr_render.argtypes = [vp, vp, f32, u32, u32, vp, vp, vp, vp, vp, vp, vp]
    L.trr_render_blocks.restype = i32
    L.trr_render_blocks.argtypes = [vp, vp, f32, u32, u32, u32, vp, vp, vp, vp, vp]
    L.trr_render_shard_device.restype = i32
    L.trr_render_shard_device.argtypes = [vp, vp, f32, u32, u32, u32, u32, vp, C.c_size_t, vp, vp, vp, vp, vp]
    if hasattr(L, "trr_render_shard_device_noalloc"):   # (absent only from older builds loaded by the A/B experiment scripts)
        L.trr_render_shard_device_noalloc.restype = i32
        L.trr_render_shard_device_noalloc.argtypes = [vp, vp, f32, u32, u32, u32, vp, C.c_size_t, vp]
        L.trr_noalloc_scratch_bytes.restype = C.c_size_t
        L.trr_noalloc_scratch_bytes.argtypes = []
    L.trr_strip_bytes.restype = C.c_size_t
    L.trr_strip_bytes.argtypes = [u32, u32, u32]
    L.trr_assemble_device.restype = i32
    L.trr_assemble_device.argtypes = [vp, u32, u32, u32, vp, vp, vp]
    L.trr_synchronize.restype = i32
    L.trr_synchronize.argtypes = [vp, vp]
    L.trr_collect_stats.restype = i32
    L.trr_collect_stats.argtypes = [vp]
    L.trr_object_matrices.restype = i32
    L.trr_object_matrices.argtypes = [vp, vp, vp, vp, vp]
    L.trr_camera_ray.restype = None
    L.trr_camera_ray.argtypes = [vp, f32, f32, vp, vp]
    _lib = L
    return L


def check(rc: int, ctx=None):
    if rc != TRR_OK:
        msg = lib().trr_last_error(ctx)
        raise NativeError(rc, msg.decode("utf-8", "replace") if msg else "")
