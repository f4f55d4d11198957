"""Host mirror of the reference's ``src/raytracer`` interface on top of the C ABI (include/trayracer_hip.h).

Reference                                             here
---------------------------------------------------   ------------------------------------------------
Raytracer::new(scene, delta, max_depth)  mod.rs:37    Raytracer(scene, delta=1e-5, max_depth=5)
block loop of RenderingThread::run  render.rs:103-152 Raytracer.render_frame() / render_shard_device()
Render{progress,thread,cancel,image,time} render.rs:15 Render (same field names and meaning)

The per-pixel entry ``Raytracer::render((x,y),(w,h),aa)`` (mod.rs:218) is not offered: the boundary is the
block loop.  All arithmetic happens in the HIP library; nothing here falls back to a CPU path.
"""
from __future__ import annotations

import ctypes as C
import threading
import time
from typing import Optional

import numpy as np

from . import _native as N
from .scene import Scene

DEFAULT_DELTA = 1e-5    # render.rs:105
DEFAULT_MAX_DEPTH = 5   # render.rs:105


def _f(dst, src):
    for i in range(len(dst)):
        dst[i] = float(src[i])


def object_matrices(translation, rotation, scale):
    """Object::transform() and its inverse (object.rs:177-182), column-major float32[16] each, via the library's
    restatement of nalgebra (trr_object_matrices)."""
    t = np.ascontiguousarray(translation, np.float32)
    r = np.ascontiguousarray(rotation, np.float32)
    s = np.ascontiguousarray(scale, np.float32)
    fwd = np.zeros(16, np.float32)
    inv = np.zeros(16, np.float32)
    ok = N.lib().trr_object_matrices(t.ctypes.data, r.ctypes.data, s.ctypes.data, fwd.ctypes.data, inv.ctypes.data)
    return fwd, inv, bool(ok)


def camera_ray(camera, x: float, y: float):
    """Camera::ray (camera.rs:30-42) through the library's host restatement."""
    c = _camera_desc(camera)
    o = np.zeros(3, np.float32)
    d = np.zeros(3, np.float32)
    N.lib().trr_camera_ray(C.addressof(c), float(x), float(y), o.ctypes.data, d.ctypes.data)
    return o, d


def _camera_desc(cam) -> N.CameraDesc:
    c = N.CameraDesc()
    _f(c.position, cam.position)
    _f(c.look_at, cam.look_at)
    _f(c.up, cam.up)
    c.fov = float(cam.fov)
    c.width, c.height = int(cam.resolution[0]), int(cam.resolution[1])
    return c


class Context:
    """One GPU of this process (trr_create), or -- ``devices=[...]`` -- several behind one context (trr_create_multi): scenes
    are then replicated, ``render_frame`` / ``render_blocks`` shard the block grid over the devices and gather the strips
    with one ncclGather inside the library."""

    def __init__(self, device: int = 0, devices=None):
        self._h = C.c_void_p()
        if devices is not None:
            ids = (C.c_int * len(devices))(*[int(d) for d in devices])
            N.check(N.lib().trr_create_multi(C.addressof(ids), len(devices), C.byref(self._h)))
            self.device = int(devices[0])
        else:
            N.check(N.lib().trr_create(int(device), C.byref(self._h)))
            self.device = int(device)

    @property
    def device_count(self) -> int:
        return int(N.lib().trr_device_count(self._h))

    @property
    def handle(self):
        return self._h

    def collect_stats(self):
        """Waits for the frames rendered with ``defer_stats`` and fills their ``Stats``."""
        N.check(N.lib().trr_collect_stats(self._h), self._h)

    def synchronize(self, stream: Optional[int] = None):
        N.check(N.lib().trr_synchronize(self._h, stream), self._h)

    def close(self):
        if self._h:
            N.lib().trr_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Raytracer:
    """``Raytracer::new(scene, delta, max_depth)`` (raytracer/mod.rs:37): uploads a flattened copy of ``scene``
    (the reference deep-clones it, render.rs:60) and renders it with the gfx950 kernels."""

    def __init__(self, scene: Scene, delta: float = DEFAULT_DELTA, max_depth: int = DEFAULT_MAX_DEPTH,
                 context: Optional[Context] = None, device: int = 0, build: str = "host"):
        """``build``: "host" = binned-SAH BVH on the host cores (best tree), "device" = Morton-order tree with SAH-rebuilt
        treelets on the GPU (fastest upload, frames within a few per cent; for scenes that change between renders).  Same image either way (include/trayracer_hip.h)."""
        if build not in ("host", "device"):
            raise ValueError("build must be 'host' or 'device'")
        self.build = build
        self.ctx = context or Context(device)
        self.delta = float(delta)
        self.max_depth = int(max_depth)
        self._scene = C.c_void_p()
        self._upload(scene)
        self.width, self.height = int(scene.camera.resolution[0]), int(scene.camera.resolution[1])

    # -- upload -----------------------------------------------------------------------------------------
    def _upload(self, scene: Scene):
        L = N.lib()
        keep = []
        textures = []
        objs = (N.ObjectDesc * max(1, len(scene.objects)))()
        for oi, o in enumerate(scene.objects):
            tri = np.ascontiguousarray(o.triangles, np.float32)
            mi = np.ascontiguousarray(o.material_index, np.int32)
            mats = (N.MaterialDesc * max(1, len(o.materials)))()
            for k, m in enumerate(o.materials):
                flags = 0
                if m.diffuse_color is not None:
                    flags |= N.TRR_MAT_HAS_KD
                    _f(mats[k].kd, m.diffuse_color)
                if m.specular_color is not None:
                    flags |= N.TRR_MAT_HAS_KS
                    _f(mats[k].ks, m.specular_color)
                if m.specular_exponent is not None:
                    flags |= N.TRR_MAT_HAS_NS
                    mats[k].ns = float(m.specular_exponent)
                if m.dissolve is not None:
                    flags |= N.TRR_MAT_HAS_DISSOLVE
                    mats[k].dissolve = float(m.dissolve)
                mats[k].illum = int(m.illumination_model.value)
                mats[k].texture = -1
                if m.diffuse_texture is not None:
                    mats[k].texture = len(textures)
                    textures.append(np.ascontiguousarray(m.diffuse_texture, np.uint8))
                mats[k].flags = flags
            fwd, inv, _ok = object_matrices(o.translation, o.rotation, o.scale)
            keep += [tri, mi, mats]
            objs[oi].triangles = tri.ctypes.data
            objs[oi].material_index = mi.ctypes.data
            objs[oi].n_triangles = len(tri)
            objs[oi].materials = C.addressof(mats)
            objs[oi].n_materials = len(o.materials)
            _f(objs[oi].transform, fwd)
            _f(objs[oi].inverse, inv)
        imgs = (N.ImageRgb8 * max(1, len(textures)))()
        for k, t in enumerate(textures):
            imgs[k].rgb = t.ctypes.data
            imgs[k].height, imgs[k].width = t.shape[0], t.shape[1]
        t0 = time.perf_counter()
        N.check(L.trr_scene_create_ex(self.ctx.handle, C.addressof(objs), len(scene.objects), C.addressof(imgs), len(textures),
                                      N.TRR_BUILD_DEVICE if self.build == "device" else N.TRR_BUILD_HOST,
                                      C.byref(self._scene)), self.ctx.handle)
        self.upload_seconds = time.perf_counter() - t0
        del keep
        self.set_lights(scene.lights)
        self.set_settings(scene.settings)
        self.set_camera(scene.camera)

    def scene_info(self) -> dict:
        """Triangle / node counts, tree depth and build time of the uploaded scene (trr_scene_get_info)."""
        info = N.SceneInfo()
        N.check(N.lib().trr_scene_get_info(self._scene, C.addressof(info)), self.ctx.handle)
        return info.as_dict()

    def validate(self) -> None:
        """Checks the uploaded BVH on the host (trr_scene_validate); raises NativeError with the first violation."""
        N.check(N.lib().trr_scene_validate(self._scene), self.ctx.handle)

    def set_lights(self, lights):
        arr = (N.LightDesc * max(1, len(lights)))()
        for k, l in enumerate(lights):
            _f(arr[k].position, l.position)
            _f(arr[k].color, l.color)
            arr[k].intensity = float(l.intensity)
        N.check(N.lib().trr_scene_set_lights(self._scene, C.addressof(arr), len(lights)), self.ctx.handle)

    def set_settings(self, s):
        st = N.SettingsDesc()
        _f(st.ambient_color, s.ambient_color)
        st.ambient_intensity = float(s.ambient_intensity)
        sky = None
        if s.skybox.image is not None:
            sky = np.ascontiguousarray(s.skybox.image, np.uint8)
            st.skybox_is_image = 1
            st.skybox_image.rgb = sky.ctypes.data
            st.skybox_image.height, st.skybox_image.width = sky.shape[0], sky.shape[1]
        else:
            st.skybox_is_image = 0
            _f(st.skybox_color, s.skybox.color)
        st.anti_aliasing = 1 if s.anti_aliasing else 0
        st.samples = int(s.samples)
        st.seed = int(s.seed)
        N.check(N.lib().trr_scene_set_settings(self._scene, C.addressof(st)), self.ctx.handle)
        del sky

    def set_camera(self, cam):
        c = _camera_desc(cam)
        N.check(N.lib().trr_scene_set_camera(self._scene, C.addressof(c)), self.ctx.handle)
        self.width, self.height = c.width, c.height

    def set_instances(self, objects):
        fw = np.zeros((len(objects), 16), np.float32)
        iv = np.zeros((len(objects), 16), np.float32)
        for i, o in enumerate(objects):
            fw[i], iv[i], _ = object_matrices(o.translation, o.rotation, o.scale)
        N.check(N.lib().trr_scene_set_instances(self._scene, fw.ctypes.data, iv.ctypes.data, len(objects)), self.ctx.handle)

    # -- render -----------------------------------------------------------------------------------------
    def render_frame(self, want_ids: bool = False, counters: bool = False, cancel_flag=None, progress=None):
        """Whole frame to host memory (block loop of render.rs:103-152).  Returns dict(rgb[h,w,3] u8, hit_obj, hit_tri, stats).
        ``cancel_flag`` is a ctypes c_int32 polled between block batches; ``progress(done, total)`` is called per batch."""
        w, h = self.width, self.height
        rgb = np.zeros((h, w, 3), np.uint8)
        ho = np.full((h, w), 0xFFFFFFFF, np.uint32) if want_ids else None
        ht = np.full((h, w), 0xFFFFFFFF, np.uint32) if want_ids else None
        st = N.Stats()
        cb = N.PROGRESS_CB(lambda user, done, total: progress(done, total)) if progress else None
        rc = N.lib().trr_render(self.ctx.handle, self._scene, self.delta, self.max_depth,
                                N.TRR_FLAG_COUNTERS if counters else 0, rgb.ctypes.data,
                                ho.ctypes.data if want_ids else None, ht.ctypes.data if want_ids else None,
                                C.addressof(cancel_flag) if cancel_flag is not None else None,
                                C.cast(cb, C.c_void_p) if cb else None, None, C.addressof(st))
        if rc == N.TRR_ERR_RESOLUTION:
            raise ZeroDivisionError("resolution below 20 pixels: the reference divides by zero (render.rs:109)")
        cancelled = rc == N.TRR_ERR_CANCELLED
        if not cancelled:
            N.check(rc, self.ctx.handle)
        return {"rgb": rgb, "hit_obj": ho, "hit_tri": ht, "stats": st, "cancelled": cancelled}

    def render_blocks(self, sink, blocks_per_batch: int = 0, cancel_flag=None, progress=None, counters: bool = False):
        """Progressive render (the sink of render.rs:155-186): ``sink(x_block, y_block, block)`` is called once per finished
        block, ``block`` a [block_h, block_w, 3] uint8 array valid only during the call, batches of ``blocks_per_batch``
        blocks (0 = one row of the 20x20 grid) in row-major block order.  Returns dict(stats, cancelled)."""
        st = N.Stats()

        def on_block(user, xb, yb, bw, bh, ptr):
            sink(int(xb), int(yb), np.ctypeslib.as_array(ptr, shape=(bh, bw, 3)))

        bcb = N.BLOCK_CB(on_block)
        pcb = N.PROGRESS_CB(lambda user, done, total: progress(done, total)) if progress else None
        rc = N.lib().trr_render_blocks(self.ctx.handle, self._scene, self.delta, self.max_depth,
                                       N.TRR_FLAG_COUNTERS if counters else 0, int(blocks_per_batch),
                                       C.addressof(cancel_flag) if cancel_flag is not None else None,
                                       C.cast(bcb, C.c_void_p), C.cast(pcb, C.c_void_p) if pcb else None, None, C.addressof(st))
        if rc == N.TRR_ERR_RESOLUTION:
            raise ZeroDivisionError("resolution below 20 pixels: the reference divides by zero (render.rs:109)")
        cancelled = rc == N.TRR_ERR_CANCELLED
        if not cancelled:
            N.check(rc, self.ctx.handle)
        return {"stats": st, "cancelled": cancelled}

    def strip_bytes(self, n_shards: int) -> int:
        return int(N.lib().trr_strip_bytes(self.width, self.height, int(n_shards)))

    def render_shard_device(self, shard: int, n_shards: int, strip_ptr: int, strip_bytes: int, stream: Optional[int] = None,
                            counters: bool = False, stats: Optional[N.Stats] = None, defer_stats: bool = False,
                            pipelined: bool = False):
        """This rank's blocks (block id % n_shards == shard) into a device strip; asynchronous on ``stream``
        unless ``stats`` is given without ``defer_stats`` (statistics need the queue counters back).  With
        ``defer_stats`` the call stays asynchronous and ``stats`` is filled by ``Context.collect_stats()``.  ``pipelined``: the
        caller keeps two frames in flight on alternating streams and strips (TRR_FLAG_PIPELINED)."""
        flags = ((N.TRR_FLAG_COUNTERS if counters else 0) | (N.TRR_FLAG_DEFER_STATS if (defer_stats and stats is not None) else 0) |
                 (N.TRR_FLAG_PIPELINED if pipelined else 0))
        rc = N.lib().trr_render_shard_device(self.ctx.handle, self._scene, self.delta, self.max_depth,
                                             flags, int(shard), int(n_shards),
                                             strip_ptr, int(strip_bytes), stream, None, None, None,
                                             C.addressof(stats) if stats is not None else None)
        if rc == N.TRR_ERR_RESOLUTION:
            raise ZeroDivisionError("resolution below 20 pixels: the reference divides by zero (render.rs:109)")
        N.check(rc, self.ctx.handle)

    def render_shard_device_noalloc(self, shard: int, n_shards: int, strip_ptr: int, strip_bytes: int, stream: Optional[int] = None):
        """``render_shard_device`` for callers that cannot allocate (trr_render_shard_device_noalloc): all work buffers come
        from the library's static 256 MiB device array; raises NativeError(TRR_ERR_INVALID) when one block does not fit."""
        rc = N.lib().trr_render_shard_device_noalloc(self.ctx.handle, self._scene, self.delta, self.max_depth, int(shard),
                                                     int(n_shards), strip_ptr, int(strip_bytes), stream)
        if rc == N.TRR_ERR_RESOLUTION:
            raise ZeroDivisionError("resolution below 20 pixels: the reference divides by zero (render.rs:109)")
        N.check(rc, self.ctx.handle)

    def assemble_device(self, n_shards: int, strips_ptr: int, frame_ptr: int, stream: Optional[int] = None):
        N.check(N.lib().trr_assemble_device(self.ctx.handle, self.width, self.height, int(n_shards), strips_ptr, frame_ptr,
                                            stream), self.ctx.handle)

    def close(self):
        if self._scene:
            N.lib().trr_scene_destroy(self._scene)
            self._scene = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Render:
    """The UI-facing handle of raytracer/render.rs:15-25 with the same public fields: ``progress`` in
    [0, 65535] (finished <=> 65535), ``thread``, ``cancel``, ``image`` (h x w x 3 uint8, the RgbImage) and
    ``time`` in milliseconds.  ``render(scene)`` spawns the render thread like render.rs:39-71."""

    def __init__(self, device: int = 0):
        self.progress = 0
        self.thread: Optional[threading.Thread] = None
        self.cancel = C.c_int32(0)
        self.image = np.zeros((0, 0, 3), np.uint8)
        self.time = 0
        self.error: Optional[BaseException] = None
        self._ctx = Context(device)

    def render(self, scene: Scene, delta: float = DEFAULT_DELTA, max_depth: int = DEFAULT_MAX_DEPTH, blocks_per_batch: int = 0):
        """``self.image`` fills block by block while the thread runs, as the reference's sink paints each finished block
        into ``image`` (render.rs:170-183); ``blocks_per_batch`` = blocks that appear together (0 = one row of blocks)."""
        if self.thread is not None and self.thread.is_alive():
            raise RuntimeError("a render is already in flight (statusbar.rs:142-157 disables the button)")
        w, h = scene.camera.resolution
        self.image = np.zeros((h, w, 3), np.uint8)   # render.rs:51
        self.progress = 0                            # render.rs:54
        self.time = 0                                # render.rs:55
        self.error = None

        def run():
            start = time.perf_counter()              # render.rs:96
            try:
                rt = Raytracer(scene, delta, max_depth, context=self._ctx)
                try:
                    def on_progress(done, total):    # render.rs:143-149
                        self.progress = int(round(done / max(1, total) * 65535))
                    image = self.image

                    def sink(xb, yb, block):         # render.rs:170-183
                        bh, bw = block.shape[:2]
                        image[yb * bh:(yb + 1) * bh, xb * bw:(xb + 1) * bw] = block
                    rt.render_blocks(sink, blocks_per_batch=blocks_per_batch, cancel_flag=self.cancel, progress=on_progress)
                finally:
                    rt.close()
            except BaseException as e:   # the reference's render thread has no error channel; we keep one
                self.error = e
            self.progress = 65535                    # render.rs:188
            self.time = int((time.perf_counter() - start) * 1000)   # render.rs:189-190

        self.thread = threading.Thread(target=run, daemon=True)
        self.thread.start()

    def join(self):
        if self.thread is not None:
            self.thread.join()
            self.thread = None
            self.cancel = C.c_int32(0)   # ui/mod.rs:93-100
