"""Device BVH build of the 10 M-triangle terrain (and the 1 M-triangle car) with the pinned staging ring of build_device.hip on
and off (TRR_BUILD_STAGING=0): trr_scene_create_ex time in ms (best of 4), and the frame's digest.  Run: python scripts/experiments/build_staging.py"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from raytracing_amd import standin  # noqa: E402
from raytracing_amd.raytracer import Context, Raytracer  # noqa: E402

ctx = Context(0)
for name, make, md in (("terrain 10 M", lambda: standin.standin_terrain(1280, 720), 2), ("car 1 M", lambda: standin.standin_car(1280, 720), 2)):
    sc = make()
    for staging in ("1", "0", "1", "0"):
        os.environ["TRR_BUILD_STAGING"] = staging
        best, sha = 1e9, None
        for _ in range(4):
            rt = Raytracer(sc, max_depth=md, context=ctx, build="device")
            best = min(best, rt.scene_info()["ms_build"])
            sha = hashlib.sha1(rt.render_frame(want_ids=False)["rgb"].tobytes()).hexdigest()[:8]
            rt.close()
        print(f"{name}: staging {staging}: create {best:.1f} ms  sha {sha}", flush=True)
ctx.close()
