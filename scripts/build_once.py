#!/usr/bin/env python3
"""One device BVH build of a BASELINE.json mesh, for rocprofv3 (kernel durations of build_device.hip).
   rocprofv3 --kernel-trace --stats -d out -- python3 scripts/build_once.py terrain"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import standin  # noqa: E402
from raytracing_amd.raytracer import Raytracer  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "terrain"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
sc = {"car": lambda: standin.standin_car(3840, 2160, n_tris=1_000_000, seed=1),
      "mirrors": lambda: standin.standin_mirrors(3840, 2160),
      "terrain": lambda: standin.standin_terrain(3840, 2160)}[name]()
for k in range(reps):
    t0 = time.perf_counter()
    rt = Raytracer(sc, build="device")
    print(name, "device build", round((time.perf_counter() - t0) * 1e3, 1), "ms", rt.scene_info(), flush=True)
    rt.close()
