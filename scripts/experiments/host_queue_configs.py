"""Host time to QUEUE one lone frame (no synchronisation) next to the frame time, for configurations 1, 4, 5 and an eighth of the\n4K x 16 spp frame.  Usage: host_queue_configs.py 1,4,5,s8"""
import os, sys, time
os.environ["GPU_MAX_HW_QUEUES"] = "16"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from raytracing_amd import standin
from raytracing_amd.raytracer import Raytracer
cfgs = {"1": (lambda: standin.standin_test(800, 600), 2, 1), "5": (lambda: standin.standin_mirrors(3840, 2160), 9, 1),
        "s8": (lambda: standin.standin_car(3840, 2160, anti_aliasing=True, samples=16), 5, 8), "4": (lambda: standin.standin_terrain(3840, 2160), 5, 1)}
for c in sys.argv[1].split(","):
    make, md, n = cfgs[c]
    rt = Raytracer(make(), max_depth=md)
    sb = rt.strip_bytes(n)
    strip = torch.zeros(sb, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    t = time.perf_counter()
    while time.perf_counter() - t < 0.5:
        rt.render_shard_device(0, n, strip.data_ptr(), sb); torch.cuda.synchronize()
    ts = []
    for _ in range(12):
        t0 = time.perf_counter()
        rt.render_shard_device(0, n, strip.data_ptr(), sb)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        ts.append(((t1 - t0) * 1e3, (t2 - t0) * 1e3))
    print(f"cfg{c}: host queueing {min(a for a, _ in ts):.3f} ms (median {sorted(a for a, _ in ts)[6]:.3f}), frame {min(b for _, b in ts):.3f} ms", flush=True)
    rt.close()
