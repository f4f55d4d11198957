import sys, os, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
import numpy as np, torch
from raytracing_amd import _native as N
from raytracing_amd import standin
from raytracing_amd.raytracer import Raytracer
for name, sc, md in (("cfg2 1080p", standin.standin_car(1920, 1080), 5), ("cfg1 800x600", standin.standin_test(800, 600), 2),
                     ("cfg4 terrain 4K", standin.standin_terrain(3840, 2160, nx=1000, nz=500), 5)):
    rt = Raytracer(sc, max_depth=md)
    sb = rt.strip_bytes(1)
    strip = torch.zeros(sb, dtype=torch.uint8, device="cuda")
    s = torch.cuda.Stream(); torch.cuda.synchronize()
    res = []
    for lanes, nb in ((1, 1), (2, 2), (3, 3), (2, 4), (3, 6)):
        os.environ["TRR_BATCHES"] = str(nb); os.environ["TRR_LANES"] = str(lanes)
        def run(k):
            for i in range(k):
                rt.render_shard_device(0, 1, strip.data_ptr(), sb, stream=s.cuda_stream)
            torch.cuda.synchronize()
        run(3)
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter(); run(10); best = min(best, (time.perf_counter() - t0) * 1e3 / 10)
        res.append("L%d/B%d %.2f" % (lanes, nb, best))
    print(name, "  ".join(res), flush=True)
    rt.close()
