import sys, os, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
import numpy as np, torch
from raytracing_amd import _native as N
from raytracing_amd import standin
from raytracing_amd.raytracer import Raytracer
sc = standin.standin_car(3840, 2160, n_tris=1_000_000, anti_aliasing=True, samples=16)
rt = Raytracer(sc, max_depth=5)
for n in (8, 4):
    sb = rt.strip_bytes(n)
    for label, env, fif in (("default L2/B2 fif1", {}, 1), ("default L2/B2 fif2", {}, 2),
                            ("pipelined flag fif2", {"PIPE": "1"}, 2), ("pipelined pool3 fif3", {"PIPE": "1", "POOL": "3"}, 3)):
        pipe = "PIPE" in env
        os.environ.pop("TRR_POOL", None)
        if "POOL" in env: os.environ["TRR_POOL"] = env["POOL"]
        strips = [torch.zeros(sb, dtype=torch.uint8, device="cuda") for _ in range(fif)]
        streams = [torch.cuda.Stream() for _ in range(fif)]
        torch.cuda.synchronize()
        def run(k):
            for i in range(k):
                rt.render_shard_device(0, n, strips[i % fif].data_ptr(), sb, stream=streams[i % fif].cuda_stream, pipelined=pipe)
            torch.cuda.synchronize()
        run(4)
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter(); run(12); best = min(best, (time.perf_counter() - t0) * 1e3 / 12)
        ref = strips[0].cpu().numpy().copy()
        same = all(np.array_equal(ref, s_.cpu().numpy()) for s_ in strips)
        print(f"N {n} {label}: {best:.2f} ms/frame  x{n} = {best*n:.1f}  strips identical {same} sum {int(ref.sum())}", flush=True)
rt.close()
