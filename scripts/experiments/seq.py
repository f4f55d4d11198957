import sys, os, time, hashlib
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
import numpy as np, torch
from raytracing_amd import _native as N
if len(sys.argv) > 1: N.LIB_PATH = sys.argv[1]
from raytracing_amd import standin
from raytracing_amd.raytracer import Raytracer
# throughput of a sequence of shard renders on one GPU (no gather, no host sync between frames), default batch policy
sc = standin.standin_car(3840, 2160, n_tris=1_000_000, anti_aliasing=True, samples=16)
rt = Raytracer(sc, max_depth=5)
out = []
for n in (1, 2, 4, 8):
    sb = rt.strip_bytes(n)
    strip = torch.zeros(sb, dtype=torch.uint8, device="cuda")
    s = torch.cuda.Stream(); torch.cuda.synchronize()
    def run(k):
        for i in range(k):
            rt.render_shard_device(0, n, strip.data_ptr(), sb, stream=s.cuda_stream)
        torch.cuda.synchronize()
    run(2)
    best = 1e9
    for rep in range(3):
        K = 8
        t0 = time.perf_counter(); run(K); best = min(best, (time.perf_counter() - t0) * 1e3 / K)
    out.append("N%d %.2f (x%d=%.1f)" % (n, best, n, best * n))
    if n == 1: sha = hashlib.sha1(strip.cpu().numpy().tobytes()).hexdigest()[:10]
print(os.path.basename(N.LIB_PATH), sha, "  ".join(out), flush=True)
rt.close()
