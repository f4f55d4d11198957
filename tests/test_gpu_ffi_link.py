"""GPU: examples/ffi_link -- a plain C program linked by gcc against the STATIC archive with exactly the libraries
integration/rust/build.rs names for the reference's crate (libtrayracer_hip.a, -lamdhip64, -lstdc++; no hipcc at link time).
It makes the Rust shim's calls (trr_create -> trr_scene_create_ex -> setters -> trr_render_blocks with sink, progress callback
and int32 cancel flag, integration/rust/hip.rs) on a scene this test writes, and its frame is compared with the CPU oracle and
with the Python binding.  This is the evidence for SURVEY section 8(f) row N3 that can be had without cargo."""
import ctypes as C
import os
import struct
import subprocess

import numpy as np
import pytest

from raytracing_amd import _native as N
from raytracing_amd.raytracer import Raytracer, _camera_desc, object_matrices
from raytracing_amd.scene import Light, Object, Scene, Settings
from tests._util import assert_parity
from tests.test_oracle_kat import front_camera, mat, tri_rows

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "ffi_link")
F32 = np.float32


def _scene():
    z = -3.0
    wall = tri_rows(((-3, -2, z), (3, -2, z), (3, 2, z)), ((-3, -2, z), (3, 2, z), (-3, 2, z)))
    floor = tri_rows(((-3, -1, 0), (3, -1, 0), (3, -1, z)), ((-3, -1, 0), (3, -1, z), (-3, -1, z)))
    pane = tri_rows(((-1, -1, -1.5), (0, -1, -1.5), (0, 0.5, -1.5)), ((-1, -1, -1.5), (0, 0.5, -1.5), (-1, 0.5, -1.5)))
    room = Object(triangles=np.concatenate([wall, floor]), material_index=np.array([0, 0, 1, 1], np.int32),
                  materials=[mat(kd=(0.7, 0.3, 0.2), illum=2, ks=(1, 1, 1), ns=20.0, dissolve=0.0), mat(kd=(0.6, 0.6, 0.9), illum=3, dissolve=0.0)])
    glass = Object(triangles=pane, material_index=np.zeros(2, np.int32), materials=[mat(kd=(0.2, 0.9, 0.3), illum=6, dissolve=0.4)],
                   translation=np.array([0.3, 0.1, 0.0], F32), scale=np.array([1.2, 0.9, 1.0], F32))
    lights = [Light(np.array([0, 1.5, -0.5], F32), np.array([1, 1, 1], F32), 3.0), Light(np.array([-2, 0.5, -1.0], F32), np.array([1, 0.6, 0.4], F32), 1.5)]
    return Scene(objects=[room, glass], lights=lights, camera=front_camera(320, 200), settings=Settings())


def _write(sc, path, delta, max_depth):
    with open(path, "wb") as f:
        f.write(b"TRRFFI01" + struct.pack("<I", len(sc.objects)))
        for o in sc.objects:
            tri = np.ascontiguousarray(o.triangles, F32)
            f.write(struct.pack("<I", len(tri)) + tri.tobytes() + np.ascontiguousarray(o.material_index, np.int32).tobytes())
            f.write(struct.pack("<I", len(o.materials)))
            for m in o.materials:
                d = N.MaterialDesc()
                flags = 0
                if m.diffuse_color is not None:
                    flags |= N.TRR_MAT_HAS_KD
                    d.kd[:] = [float(x) for x in m.diffuse_color]
                if m.specular_color is not None:
                    flags |= N.TRR_MAT_HAS_KS
                    d.ks[:] = [float(x) for x in m.specular_color]
                if m.specular_exponent is not None:
                    flags |= N.TRR_MAT_HAS_NS
                    d.ns = float(m.specular_exponent)
                if m.dissolve is not None:
                    flags |= N.TRR_MAT_HAS_DISSOLVE
                    d.dissolve = float(m.dissolve)
                d.illum, d.texture, d.flags = int(m.illumination_model.value), -1, flags
                f.write(bytes(d))
            fwd, inv, _ = object_matrices(o.translation, o.rotation, o.scale)
            f.write(fwd.astype(F32).tobytes() + inv.astype(F32).tobytes())
        f.write(struct.pack("<I", len(sc.lights)))
        for l in sc.lights:
            f.write(np.array([*l.position, *l.color, l.intensity], F32).tobytes())
        s = sc.settings
        f.write(np.array([*s.ambient_color, s.ambient_intensity, *s.skybox.color], F32).tobytes())
        f.write(struct.pack("<iIQ", 1 if s.anti_aliasing else 0, int(s.samples), int(s.seed)))
        f.write(bytes(_camera_desc(sc.camera)))
        f.write(struct.pack("<fI", delta, max_depth))


def test_static_archive_links_with_gcc_and_renders_the_oracles_frame(tmp_path, oracle_mod):
    assert os.path.exists(EXE), "examples/ffi_link is built by `make -C raytracing_amd/csrc` (python __graft_entry__.py)"
    assert C.sizeof(N.MaterialDesc) == 44 and C.sizeof(N.LightDesc) == 28 and C.sizeof(N.CameraDesc) == 48
    sc = _scene()
    scene_bin, ppm = os.path.join(str(tmp_path), "scene.bin"), os.path.join(str(tmp_path), "frame.ppm")
    _write(sc, scene_bin, 1e-5, 4)
    p = subprocess.run([EXE, scene_bin, ppm], capture_output=True, text=True, timeout=100)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "gfx950" in p.stdout and "blocks=400 progress_calls=20 last=400/400" in p.stdout
    assert "cancelled: rc=-4 blocks=40" in p.stdout          # the flag raised inside the sink stops the block loop (render.rs:113)
    raw = open(ppm, "rb").read()
    head = b"P6\n320 200\n255\n"
    assert raw.startswith(head)
    got = np.frombuffer(raw, np.uint8, offset=len(head)).reshape(200, 320, 3)
    rt = Raytracer(sc, max_depth=4)
    try:
        out = rt.render_frame(want_ids=True)
    finally:
        rt.close()
    assert np.array_equal(got, out["rgb"]), "the C host (static link) and the Python binding (shared library) disagree"
    osc = oracle_mod.OracleScene(sc)
    ref = osc.render(shade=oracle_mod.SHADE_LITERAL, max_depth=4)
    osc.close()
    assert_parity(out, ref, what="scene rendered through the statically linked C host")
    assert got.any()
