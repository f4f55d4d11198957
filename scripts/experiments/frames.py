"""Renders K frames of the bench workload (4K, 16 spp) with the library given by TRR_LIB (default: the built one); meant to
run under `rocprofv3 --kernel-trace --stats` with TRR_LANES / TRR_BATCHES set, to compare kernel times between builds."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from raytracing_amd import _native as N  # noqa: E402
if os.environ.get("TRR_LIB"):
    N.LIB_PATH = os.environ["TRR_LIB"]
from raytracing_amd import standin  # noqa: E402
from raytracing_amd.raytracer import Raytracer  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 3
rt = Raytracer(standin.standin_car(3840, 2160, anti_aliasing=True, samples=16), max_depth=5)
sb = rt.strip_bytes(1)
strip = torch.zeros(sb, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
for _ in range(k):
    rt.render_shard_device(0, 1, strip.data_ptr(), sb)
    rt.ctx.synchronize()
rt.close()
