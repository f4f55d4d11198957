"""GPU: the C++ host layer (include/trayracer.hpp, the compiled-language mirror of the reference's Scene / Raytracer /
Render interface) end to end.  examples/render_cpp is a plain g++ program: no Python, no PyTorch in its process.  It
renders its scene through trayracer::Render, writes the frame and a dump of the scene; the same scene is then rendered
through the Python binding and by the CPU oracle, and the three frames are compared."""
import os
import struct
import subprocess

import numpy as np
import pytest

from raytracing_amd.raytracer import Raytracer
from raytracing_amd.scene import Camera, IlluminationModel, Light, Material, Object, Scene, Settings, Skybox
from tests._util import assert_parity

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "render_cpp")


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


def test_cpp_host_layer_renders_the_same_frame(tmp_path, oracle_mod):
    assert os.path.exists(EXE), "examples/render_cpp is built by `make -C raytracing_amd/csrc` (python __graft_entry__.py)"
    ppm, dump = os.path.join(str(tmp_path), "frame.ppm"), os.path.join(str(tmp_path), "scene.bin")
    p = subprocess.run([EXE, ppm, dump, "320", "240"], capture_output=True, text=True, timeout=100)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "cancelled render: finished=1 black=1" in p.stdout           # render.rs:113 + render.rs:188
    assert "progress=65535" in p.stdout and "16x16 render: error code -2" in p.stdout
    raw = open(ppm, "rb").read()
    head = b"P6\n320 240\n255\n"
    assert raw.startswith(head)
    cpp = np.frombuffer(raw, np.uint8, offset=len(head)).reshape(240, 320, 3)

    sc = read_scene_dump(dump)
    # the C++ defaults are the reference's (settings.rs:14-25, light.rs:33), bit for bit the Python mirror's
    assert sc.settings.ambient_color.tobytes() == Settings().ambient_color.astype(np.float32).tobytes()
    assert sc.lights[1].color.tobytes() == Light.from_ke((0, 0, 0), (0.75, 0.25, 0.75), 1.0).color.tobytes()
    rt = Raytracer(sc, max_depth=5)
    try:
        out = rt.render_frame(want_ids=True)
    finally:
        rt.close()
    assert np.array_equal(cpp, out["rgb"]), "C++ host layer and Python binding disagree on the same scene"
    osc = oracle_mod.OracleScene(sc)
    ref = osc.render(shade=oracle_mod.SHADE_HOISTED, max_depth=5)
    osc.close()
    assert_parity(out, ref, what="scene built by the C++ host layer")
    assert (out["hit_obj"] != 0xFFFFFFFF).mean() > 0.3
