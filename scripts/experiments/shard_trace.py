"""Renders K frames of a 1/N shard of the bench workload, one after the other (host sync between frames); meant to run under
`rocprofv3 --kernel-trace` to see where a small shard's frame time goes.  Usage: shard_trace.py N K [config: 3 (default) | 1 | 2 | 4 | 5]"""
import os
import sys

if int(os.environ.get("GPU_MAX_HW_QUEUES", "4") or 4) < 16:
    os.environ["GPU_MAX_HW_QUEUES"] = "16"   # before the HIP runtime starts

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from raytracing_amd import standin  # noqa: E402
from raytracing_amd.raytracer import Raytracer  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
k = int(sys.argv[2]) if len(sys.argv) > 2 else 4
c = sys.argv[3] if len(sys.argv) > 3 else "3"
if c == "2":
    rt = Raytracer(standin.standin_car(1920, 1080), max_depth=5)
elif c == "1":
    rt = Raytracer(standin.standin_test(800, 600), max_depth=2)
elif c == "4":
    rt = Raytracer(standin.standin_terrain(3840, 2160), max_depth=5)
elif c == "5":
    rt = Raytracer(standin.standin_mirrors(3840, 2160), max_depth=9)
else:
    rt = Raytracer(standin.standin_car(3840, 2160, anti_aliasing=True, samples=16), max_depth=5)
sb = rt.strip_bytes(n)
strip = torch.zeros(sb, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
for _ in range(k):
    rt.render_shard_device(0, n, strip.data_ptr(), sb)
    rt.ctx.synchronize()
rt.close()
