import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from raytracing_amd import _native as N
from raytracing_amd import standin
from raytracing_amd.raytracer import Raytracer
# host time spent inside trr_render_shard_device per frame (queueing only) against the frame time, for 1/N shards
sc = standin.standin_car(3840, 2160, n_tris=1_000_000, anti_aliasing=True, samples=16)
rt = Raytracer(sc, max_depth=5)
for n in (1, 4, 8):
    sb = rt.strip_bytes(n)
    strips = [torch.zeros(sb, dtype=torch.uint8, device="cuda") for _ in range(2)]
    streams = [torch.cuda.Stream() for _ in range(2)]
    for with_stats in (False, True):
        torch.cuda.synchronize()
        K = 12
        stats = [N.Stats() for _ in range(K)]
        host = 0.0
        t0 = time.perf_counter()
        for i in range(K):
            h0 = time.perf_counter()
            rt.render_shard_device(0, n, strips[i & 1].data_ptr(), sb, stream=streams[i & 1].cuda_stream,
                                   stats=stats[i] if with_stats else None, defer_stats=True, pipelined=True)
            host += time.perf_counter() - h0
        torch.cuda.synchronize()
        total = time.perf_counter() - t0
        rt.ctx.collect_stats()
        print(f"N{n} stats={with_stats}: frame {total / K * 1e3:.2f} ms, host queueing {host / K * 1e3:.3f} ms per frame", flush=True)
rt.close()
