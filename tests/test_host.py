"""CPU-side tests of the product's host code: the C-ABI library loads and exports every symbol the header declares
(no compute calls -- there is no GPU here), its host math helpers agree with the oracle's restatement of nalgebra,
and the tile-sharding helpers round-trip, including a world_size-2 gloo gather."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from raytracing_amd import _native as N
from raytracing_amd import multigpu
from raytracing_amd.scene import Camera, Object

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F32 = np.float32


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "trayracer_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(trr_[a-z_]+)\s*\(", header))
    declared.discard("trr_progress_cb")
    assert declared, "no declarations found in include/trayracer_hip.h"
    assert declared == set(N.EXPORTS), f"_native.EXPORTS out of sync with the header: {declared ^ set(N.EXPORTS)}"
    lib = C.CDLL(N.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"libtrayracer_hip.so does not export {name}"


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "raytracing_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "import oracle" not in text and "from oracle" not in text and "liboracle" not in text, f


def test_create_without_gpu_fails_loudly_not_silently():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    rc = N.lib().trr_create(0, C.byref(h))
    assert rc == N.TRR_ERR_DEVICE and not h
    assert b"HIP device" in N.lib().trr_last_error(None)
    with pytest.raises(N.NativeError):
        N.check(rc)


def test_null_handles_are_rejected_before_any_device_call():
    """Every entry point returns a status instead of aborting (include/trayracer_hip.h, rules of the boundary); with NULL
    handles that happens before the first HIP call, so it can be checked without a GPU."""
    L = N.lib()
    h = C.c_void_p()
    assert L.trr_scene_create(None, None, 0, None, 0, C.byref(h)) == N.TRR_ERR_INVALID and not h
    assert L.trr_scene_create_ex(None, None, 0, None, 0, N.TRR_BUILD_DEVICE, C.byref(h)) == N.TRR_ERR_INVALID and not h
    assert L.trr_scene_get_info(None, None) == N.TRR_ERR_INVALID
    assert L.trr_scene_validate(None) == N.TRR_ERR_INVALID
    assert L.trr_scene_set_lights(None, None, 0) == N.TRR_ERR_INVALID
    assert L.trr_render(None, None, 1e-5, 5, 0, None, None, None, None, None, None, None) == N.TRR_ERR_INVALID
    assert L.trr_render_blocks(None, None, 1e-5, 5, 0, 0, None, None, None, None, None) == N.TRR_ERR_INVALID
    assert L.trr_render_shard_device(None, None, 1e-5, 5, 0, 0, 1, None, 0, None, None, None, None, None) == N.TRR_ERR_INVALID
    assert L.trr_collect_stats(None) == N.TRR_ERR_INVALID
    assert b"NULL" in L.trr_last_error(None)
    L.trr_scene_destroy(None)
    L.trr_destroy(None)


def test_version_string():
    assert b"gfx950" in N.lib().trr_version()


@pytest.mark.parametrize("t,q,s", [((0, 0, 0), (0, 0, 0, 1), (1, 1, 1)), ((0.7, -0.1, -0.5), (0, 0, 0, 1), (1, 1, 1)),
                                   ((-1.2, 1.6, -2.0), None, (1.3, 0.7, 1.1)), ((1, 2, 3), None, (2, 4, 8))])
def test_object_matrices_match_oracle_bitwise(oracle_mod, t, q, s):
    from raytracing_amd.raytracer import object_matrices
    if q is None:
        q = Object.quaternion_from_euler_degrees(35.0, 10.0, 60.0)
    fwd, inv, ok = object_matrices(t, q, s)
    f2, i2 = oracle_mod.object_matrices(t, q, s)
    assert ok and i2 is not None
    assert np.array_equal(fwd.reshape(4, 4).T.view(np.uint32), f2.view(np.uint32))
    assert np.array_equal(inv.reshape(4, 4).T.view(np.uint32), i2.view(np.uint32))


def test_singular_transform_reports_not_invertible():
    from raytracing_amd.raytracer import object_matrices
    fwd, inv, ok = object_matrices((0, 0, 0), (0, 0, 0, 1), (1, 0, 1))
    assert not ok and np.array_equal(fwd, inv)   # nalgebra leaves a singular matrix untouched


def test_camera_ray_matches_oracle_bitwise(oracle_mod):
    from raytracing_amd.raytracer import camera_ray
    cam = Camera(position=np.array([3.7080245, 2.1147687, 7.0919194], F32), look_at=np.array([3.3485944, 1.793123, 6.215932], F32),
                 up=np.array([0, 1, 0], F32), fov=1.1, resolution=(320, 200))
    for x, y in [(0.0, 0.0), (-1.6, -1.0), (1.59, 0.99), (0.123, -0.456)]:
        o1, d1 = camera_ray(cam, x, y)
        o2, d2 = oracle_mod.camera_ray(cam, x, y)
        assert np.array_equal(o1.view(np.uint32), o2.view(np.uint32))
        assert np.array_equal(d1.view(np.uint32), d2.view(np.uint32))


# ---- tiling / sharding (raytracer/render.rs:103-110) ---------------------------------------------------------------------
def test_tiling_matches_reference_block_grid():
    assert multigpu.tiling(3840, 2160) == (192, 108, 20, 20)
    assert multigpu.tiling(810, 610) == (40, 30, 20, 20)       # 10-px remainder strips
    assert multigpu.tiling(50, 47) == (2, 2, 25, 23)           # w / (w / 20) can exceed 20 blocks
    with pytest.raises(ZeroDivisionError):
        multigpu.tiling(19, 100)
    for (w, h, n) in [(3840, 2160, 8), (810, 610, 3), (50, 47, 4), (1920, 1080, 1)]:
        assert multigpu.strip_bytes(w, h, n) == N.lib().trr_strip_bytes(w, h, n)
    assert N.lib().trr_strip_bytes(19, 100, 2) == 0


@pytest.mark.parametrize("w,h,n", [(80, 60, 1), (80, 60, 2), (50, 47, 3), (810, 610, 8), (64, 40, 7)])
def test_strip_roundtrip(w, h, n):
    rng = np.random.default_rng(w * 1000 + h + n)
    frame = rng.integers(1, 255, size=(h, w, 3), dtype=np.uint8)
    bw, bh, nbx, nby = multigpu.tiling(w, h)
    strips = np.stack([multigpu.strip_from_frame(frame, s, n) for s in range(n)])
    out = multigpu.assemble_host(strips, w, h, n)
    expect = frame.copy()
    expect[nby * bh:, :] = 0     # the remainder strip is never rendered (render.rs:100-102)
    expect[:, nbx * bw:] = 0
    assert np.array_equal(out, expect)
    owned = [set(multigpu.owned_blocks(w, h, s, n)) for s in range(n)]
    assert set().union(*owned) == set(range(nbx * nby)) and sum(len(o) for o in owned) == nbx * nby


def _gloo_worker(rank, world, port, w, h, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(5)
    frame = rng.integers(1, 255, size=(h, w, 3), dtype=np.uint8)     # every rank "renders" the same frame ...
    strip = multigpu.strip_from_frame(frame, rank, world)             # ... but contributes only its own blocks
    strips = multigpu.gather_strips_host(strip, rank, world)
    if rank == 0:
        q.put(multigpu.assemble_host(strips, w, h, world))
    else:
        assert strips is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_gather_assembles_the_frame():
    import torch.multiprocessing as mp
    w, h, world = 90, 64, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_gloo_worker, args=(r, world, port, w, h, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    rng = np.random.default_rng(5)
    frame = rng.integers(1, 255, size=(h, w, 3), dtype=np.uint8)
    bw, bh, nbx, nby = multigpu.tiling(w, h)
    frame[nby * bh:, :] = 0
    frame[:, nbx * bw:] = 0
    assert np.array_equal(out, frame)


def test_cpp_host_header_compiles_standalone(tmp_path):
    """include/trayracer.hpp is self-contained C++17 over the C header (syntax check only, no GPU and no linking)."""
    import subprocess
    src = os.path.join(str(tmp_path), "t.cpp")
    with open(src, "w") as f:
        f.write('#include "trayracer.hpp"\nint main() { trayracer::Scene s; trayracer::Settings st; return (int)s.objects.size() + (int)st.samples - 1; }\n')
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    p = subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I", inc, src], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr


def test_shard_partition_of_the_library_at_8_devices():
    """trr_shard_blocks (the partition a multi-device context and the rank-per-GPU path both use) at n = 8: every block of
    the 20x20 grid (render.rs:109-110) is owned exactly once, in strip order, and agrees with the numpy restatement."""
    import ctypes as C
    from raytracing_amd import _native as N
    from raytracing_amd import multigpu
    L = N.lib()
    for (w, h) in [(3840, 2160), (1920, 1080), (810, 610), (50, 47)]:
        bw, bh, nbx, nby = multigpu.tiling(w, h)
        for n in (1, 2, 3, 8):
            seen = []
            for shard in range(n):
                cnt = L.trr_shard_blocks(w, h, shard, n, None, 0)
                ids = (C.c_uint32 * max(1, cnt))()
                assert L.trr_shard_blocks(w, h, shard, n, C.addressof(ids), cnt) == cnt
                got = list(ids)[:cnt]
                assert got == multigpu.owned_blocks(w, h, shard, n)
                assert cnt * bw * bh * 3 <= L.trr_strip_bytes(w, h, n) == multigpu.strip_bytes(w, h, n)
                seen += got
            assert sorted(seen) == list(range(nbx * nby))
    assert L.trr_shard_blocks(19, 40, 0, 1, None, 0) == 0 and L.trr_shard_blocks(100, 100, 2, 2, None, 0) == 0
