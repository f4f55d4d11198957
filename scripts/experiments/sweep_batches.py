import sys, os, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
import numpy as np, torch
from raytracing_amd import _native as N
from raytracing_amd import standin
from raytracing_amd.raytracer import Raytracer
sc = standin.standin_car(3840, 2160, n_tris=1_000_000, anti_aliasing=True, samples=16)
rt = Raytracer(sc, max_depth=5)
for n, counts in ((1, (4, 5, 6, 7, 8, 9, 12)), (2, (2, 3, 4, 5, 6, 8)), (4, (1, 2, 3, 4, 5, 6)), (8, (1, 2, 3, 4))):
    sb = rt.strip_bytes(n)
    strip = torch.zeros(sb, dtype=torch.uint8, device="cuda")
    s = torch.cuda.Stream(); torch.cuda.synchronize()
    res = []
    for lanes in (2, 3, 4):
        for nb in counts:
            if nb < lanes and nb != counts[0]: continue
            os.environ["TRR_BATCHES"] = str(nb); os.environ["TRR_LANES"] = str(min(lanes, nb))
            def run(k):
                for i in range(k):
                    rt.render_shard_device(0, n, strip.data_ptr(), sb, stream=s.cuda_stream)
                torch.cuda.synchronize()
            run(2)
            best = 1e9
            for rep in range(3):
                t0 = time.perf_counter(); run(6); best = min(best, (time.perf_counter() - t0) * 1e3 / 6)
            res.append("L%d/B%d %.2f" % (min(lanes, nb), nb, best))
    print("N", n, "  ".join(res), flush=True)
rt.close()
