import sys, os, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
import numpy as np, torch
from raytracing_amd import _native as N
from raytracing_amd import standin
from raytracing_amd.raytracer import Raytracer
n = int(sys.argv[1]); K = int(sys.argv[2]) if len(sys.argv) > 2 else 6
sc = standin.standin_car(3840, 2160, n_tris=1_000_000, anti_aliasing=True, samples=16)
rt = Raytracer(sc, max_depth=5)
sb = rt.strip_bytes(n)
strip = torch.zeros(sb, dtype=torch.uint8, device="cuda")
s = torch.cuda.Stream(); torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter()
    for i in range(K):
        rt.render_shard_device(0, n, strip.data_ptr(), sb, stream=s.cuda_stream)
    torch.cuda.synchronize()
    print("N", n, "ms/frame %.2f" % ((time.perf_counter() - t0) * 1e3 / K), flush=True)
rt.close()
