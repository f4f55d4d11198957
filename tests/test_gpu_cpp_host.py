"""GPU: the C++ host layer (include/trayracer.hpp, the compiled-language mirror of the reference's Scene / Raytracer /
Render interface) end to end.  examples/render_cpp is a plain g++ program: no Python, no PyTorch in its process.  It
renders its scene through trayracer::Render, writes the frame and a dump of the scene; the same scene is then rendered
through the Python binding and by the CPU oracle, and the three frames are compared."""
import os
import subprocess

import numpy as np
import pytest

from raytracing_amd.raytracer import Raytracer
from raytracing_amd.scene import Light, Settings
from tests._util import assert_parity, read_scene_dump

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "render_cpp")


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
