"""The ctypes mirrors of raytracing_amd/_native.py (and with them the #[repr(C)] mirrors of integration/rust/hip.rs, which
list the same fields in the same order) against include/trayracer_hip.h as a C compiler lays it out: a C program prints
sizeof and every offsetof, the test compares them with ctypes.  CPU only; also proves the header is plain C."""
import ctypes as C
import os
import subprocess

from raytracing_amd import _native as N

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

STRUCTS = {
    "trr_material_desc": (N.MaterialDesc, {}),
    "trr_image_rgb8": (N.ImageRgb8, {}),
    "trr_object_desc": (N.ObjectDesc, {}),
    "trr_light": (N.LightDesc, {}),
    "trr_settings": (N.SettingsDesc, {}),
    "trr_camera": (N.CameraDesc, {}),
    "trr_stats": (N.Stats, {}),
    "trr_scene_info": (N.SceneInfo, {"n_triangles_live": "n_triangles_live"}),
}


def test_struct_layouts_match_the_c_header(tmp_path):
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "trayracer_hip.h"', "int main(void) {"]
    for cname, (cls, _) in STRUCTS.items():
        lines.append(f'  printf("{cname} size %zu\\n", sizeof({cname}));')
        for fname, _t in cls._fields_:
            lines.append(f'  printf("{cname} {fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    seen = 0
    for line in out.splitlines():
        cname, field, value = line.split()
        cls = STRUCTS[cname][0]
        want = C.sizeof(cls) if field == "size" else getattr(cls, field).offset
        assert int(value) == want, f"{cname}.{field}: C says {value}, ctypes says {want}"
        seen += 1
    assert seen == sum(1 + len(c._fields_) for c, _ in STRUCTS.values())


def test_flag_values_match_the_c_header(tmp_path):
    names = ["TRR_OK", "TRR_ERR_INVALID", "TRR_ERR_RESOLUTION", "TRR_ERR_DEVICE", "TRR_ERR_CANCELLED", "TRR_ERR_NOMEM", "TRR_MAT_HAS_KD",
             "TRR_MAT_HAS_KS", "TRR_MAT_HAS_NS", "TRR_MAT_HAS_DISSOLVE", "TRR_FLAG_COUNTERS", "TRR_FLAG_DEFER_STATS", "TRR_FLAG_PIPELINED",
             "TRR_BUILD_HOST", "TRR_BUILD_DEVICE"]
    body = "\n".join(f'  printf("{n} %d\\n", (int){n});' for n in names)
    src = tmp_path / "flags.c"
    src.write_text('#include <stdio.h>\n#include "trayracer_hip.h"\nint main(void) {\n' + body + "\n  return 0;\n}\n")
    exe = tmp_path / "flags"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    for line in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines():
        n, v = line.split()
        assert getattr(N, n) == int(v), n
