"""Quick timing of the stand-in configurations and of a 1/8 shard sequence (one GPU, no CPU baseline).
Usage: python scripts/experiments/quick.py [configs, e.g. 2,3,s8]  -> one line per item: ms per frame."""
import os as _os

# More hardware queues than HIP's default of 4 (which the GPU box also exports): the library runs up to nine streams per
# frame.  Must be in the environment before the HIP runtime starts, i.e. before torch touches the GPU.
# TRR_KEEP_HW_QUEUES=1 leaves the environment's own value (the GPU box exports 4) in place, for the 4-queue figures.
if not _os.environ.get("TRR_KEEP_HW_QUEUES") and int(_os.environ.get("GPU_MAX_HW_QUEUES", "4") or 4) < 16:
    _os.environ["GPU_MAX_HW_QUEUES"] = "16"

import hashlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from raytracing_amd import _native as N  # noqa: E402
if os.environ.get("TRR_LIB"):
    N.LIB_PATH = os.environ["TRR_LIB"]
from raytracing_amd import standin  # noqa: E402
from raytracing_amd.raytracer import Raytracer  # noqa: E402

which = (sys.argv[1] if len(sys.argv) > 1 else "1,2,3,4,5,s8").split(",")
cfgs = {
    "1": (lambda: standin.standin_test(800, 600), 2, 1),
    "2": (lambda: standin.standin_car(1920, 1080), 5, 1),
    "3": (lambda: standin.standin_car(3840, 2160, anti_aliasing=True, samples=16), 5, 1),
    "4": (lambda: standin.standin_terrain(3840, 2160), 5, 1),
    "5": (lambda: standin.standin_mirrors(3840, 2160), 9, 1),
    "s8": (lambda: standin.standin_car(3840, 2160, anti_aliasing=True, samples=16), 5, 8),
    "s4": (lambda: standin.standin_car(3840, 2160, anti_aliasing=True, samples=16), 5, 4),
    "s2": (lambda: standin.standin_car(3840, 2160, anti_aliasing=True, samples=16), 5, 2),
}
out = []
print(f"hardware queues the library plans for: {N.lib().trr_hw_queues()}", flush=True)
for c in which:
    make, md, n = cfgs[c]
    rt = Raytracer(make(), max_depth=md)
    sb = rt.strip_bytes(n)
    strips = [torch.zeros(sb, dtype=torch.uint8, device="cuda") for _ in range(2)]
    streams = [torch.cuda.Stream() for _ in range(2)]
    torch.cuda.synchronize()

    def run(k, pipelined):
        for i in range(k):
            j = i % 2 if pipelined else 0
            rt.render_shard_device(0, n, strips[j].data_ptr(), sb, stream=streams[j].cuda_stream, pipelined=pipelined)
            if not pipelined:
                torch.cuda.synchronize()
        torch.cuda.synchronize()

    res = []
    for pipelined in (False, True):
        t_warm = time.perf_counter()
        while time.perf_counter() - t_warm < 0.5:   # small frames: the clocks need about half a second of work to come up
            run(2, pipelined)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            run(6, pipelined)
            best = min(best, (time.perf_counter() - t0) * 1e3 / 6)
        res.append(best)
    sha = hashlib.sha1(strips[0].cpu().numpy().tobytes()).hexdigest()[:8]
    out.append(f"cfg{c}: latency {res[0]:.2f} ms  pipelined {res[1]:.2f} ms  sha {sha}")
    print(out[-1], flush=True)
    rt.close()
    del strips
    torch.cuda.empty_cache()
