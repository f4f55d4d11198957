"""Headless front end: scene YAML in, PNG out -- ``python -m raytracing_amd.cli scene.yaml -o frame.png``.

The reference has no CLI; its only front end is the egui window, whose render button clones the scene and starts
``Render::render`` (ui/statusbar.rs:142-157 -> raytracer/render.rs:39-71) and whose export button writes
``Render.image`` as PNG (ui/statusbar.rs export dialog).  This is the same sequence without the window:
``Scene::load`` (raytracing_amd/loader.py) -> ``Raytracer(scene, 1e-5, 5)`` (render.rs:105) -> RGB8 frame -> file.
"""
from __future__ import annotations

import argparse
import json
import struct
import sys
import time
import zlib

import numpy as np


def write_png(path: str, rgb: np.ndarray) -> None:
    """8-bit RGB, non-interlaced, filter 0 on every row: lossless, so the file holds the frame's exact bytes."""
    h, w, c = rgb.shape
    if c != 3 or rgb.dtype != np.uint8:
        raise ValueError("write_png expects uint8 [h, w, 3]")
    raw = np.zeros((h, 1 + 3 * w), np.uint8)
    raw[:, 1:] = rgb.reshape(h, 3 * w)

    def chunk(tag: bytes, data: bytes) -> bytes:
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n")
        f.write(chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)))
        f.write(chunk(b"IDAT", zlib.compress(raw.tobytes(), 6)))
        f.write(chunk(b"IEND", b""))


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="raytracing_amd.cli", description=__doc__.splitlines()[0])
    ap.add_argument("scene", help="scene YAML (the reference's format: models / pointLights / camera / extraArgs)")
    ap.add_argument("-o", "--output", default="render.png", help="PNG (or .ppm) to write")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--width", type=int, help="override camera.width (the UI's render-size field)")
    ap.add_argument("--height", type=int, help="override camera.height")
    ap.add_argument("--max-depth", type=int, default=5, help="recursion budget of Raytracer::new (render.rs:105: 5)")
    ap.add_argument("--delta", type=float, default=1e-5, help="epsilon of Raytracer::new (render.rs:105: 1e-5)")
    ap.add_argument("--anti-aliasing", choices=("on", "off"), help="override extraArgs.anti_aliasing")
    ap.add_argument("--samples", type=int, help="override extraArgs.samples (used when anti-aliasing is on)")
    ap.add_argument("--seed", type=int, help="seed of the counter RNG behind the anti-aliasing jitter")
    ap.add_argument("--stats", action="store_true", help="print ray counts and timings as one JSON line")
    args = ap.parse_args(argv)

    from .loader import SceneLoadError, load_scene
    t0 = time.perf_counter()
    try:
        scene = load_scene(args.scene)
    except SceneLoadError as e:
        print(f"error: {e}", file=sys.stderr)
        return 2
    load_s = time.perf_counter() - t0
    w, h = scene.camera.resolution
    scene.camera.resolution = (args.width or w, args.height or h)
    if args.anti_aliasing:
        scene.settings.anti_aliasing = args.anti_aliasing == "on"
    if args.samples is not None:
        scene.settings.samples = args.samples
    if args.seed is not None:
        scene.settings.seed = args.seed

    from .raytracer import Raytracer     # loads the HIP library; raises if it is missing (no CPU fallback)
    t0 = time.perf_counter()
    rt = Raytracer(scene, delta=args.delta, max_depth=args.max_depth, device=args.device)
    upload_s = time.perf_counter() - t0
    try:
        out = rt.render_frame(counters=False)
    finally:
        rt.close()
    rgb = out["rgb"]
    if args.output.lower().endswith(".ppm"):
        with open(args.output, "wb") as f:
            f.write(b"P6\n%d %d\n255\n" % (rgb.shape[1], rgb.shape[0]))
            f.write(rgb.tobytes())
    else:
        write_png(args.output, rgb)
    if args.stats:
        st = out["stats"]
        print(json.dumps({"scene": args.scene, "triangles": scene.triangle_count(), "resolution": list(scene.camera.resolution),
                          "rays": st.rays(), "ms_render": round(st.ms_total, 3), "load_s": round(load_s, 3),
                          "upload_s": round(upload_s, 3), "Mrays_per_s": round(st.rays() / max(st.ms_total, 1e-9) / 1e3, 2),
                          "output": args.output}))
    return 0


if __name__ == "__main__":
    sys.exit(main())
