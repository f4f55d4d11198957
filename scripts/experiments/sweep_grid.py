import sys, os, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
import numpy as np, torch
from raytracing_amd import _native as N
from raytracing_amd import standin
from raytracing_amd.raytracer import Raytracer
sc = standin.standin_car(3840, 2160, n_tris=1_000_000, anti_aliasing=True, samples=16)
rt = Raytracer(sc, max_depth=5)
for n in (1, 2, 4, 8):
    sb = rt.strip_bytes(n)
    strip = torch.zeros(sb, dtype=torch.uint8, device="cuda")
    s = torch.cuda.Stream(); torch.cuda.synchronize()
    res = []
    for g in ("4,4", "2,4", "1,4", "2,5", "1,5", "3,5", "2,3"):
        os.environ["TRR_GRID_BPC"] = g
        def run(k):
            for i in range(k):
                rt.render_shard_device(0, n, strip.data_ptr(), sb, stream=s.cuda_stream)
            torch.cuda.synchronize()
        run(2)
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter(); run(6); best = min(best, (time.perf_counter() - t0) * 1e3 / 6)
        res.append("bpc %s %.2f" % (g, best))
    print("N", n, "  ".join(res), flush=True)
rt.close()
