"""Host time to QUEUE one shard's render (no synchronisation): what a single host thread pays per device before the next
device's shard can start.  Usage: host_queue_time.py [n_shards]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from raytracing_amd import standin  # noqa: E402
from raytracing_amd.raytracer import Raytracer  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rt = Raytracer(standin.standin_car(3840, 2160, anti_aliasing=True, samples=16), max_depth=5)
sb = rt.strip_bytes(n)
strip = torch.zeros(sb, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
for _ in range(3):
    rt.render_shard_device(0, n, strip.data_ptr(), sb)
    torch.cuda.synchronize()
ts = []
for _ in range(10):
    t0 = time.perf_counter()
    rt.render_shard_device(0, n, strip.data_ptr(), sb)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    ts.append(((t1 - t0) * 1e3, (t2 - t0) * 1e3))
print(f"1/{n} shard: host queueing {min(a for a, _ in ts):.3f} ms (median {sorted(a for a, _ in ts)[5]:.3f}), frame {min(b for _, b in ts):.3f} ms")
rt.close()
