import sys, os, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
import numpy as np, torch
from raytracing_amd import _native as N
from raytracing_amd import standin
from raytracing_amd.raytracer import Raytracer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
os.environ["TRR_LANES"] = sys.argv[2] if len(sys.argv) > 2 else "1"
os.environ["TRR_BATCHES_PER_LANE"] = sys.argv[3] if len(sys.argv) > 3 else "1"
sc = standin.standin_car(3840, 2160, n_tris=1_000_000, anti_aliasing=True, samples=16)
rt = Raytracer(sc, max_depth=5)
sb = rt.strip_bytes(n)
strip = torch.zeros(sb, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
for i in range(3):
    st = N.Stats()
    t0 = time.perf_counter()
    rt.render_shard_device(0, n, strip.data_ptr(), sb, stats=st)
    torch.cuda.synchronize()
    print("wall %.2f" % ((time.perf_counter() - t0) * 1e3), {k: (round(v, 2) if isinstance(v, float) else v) for k, v in st.as_dict().items()}, flush=True)
rt.close()
