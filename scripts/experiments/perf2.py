import sys, time, os
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
import numpy as np, torch
from raytracing_amd import _native as N
N.LIB_PATH = sys.argv[1]
from raytracing_amd import standin
from raytracing_amd.raytracer import Raytracer
import hashlib
sc = standin.standin_car(3840, 2160, n_tris=1_000_000, anti_aliasing=True, samples=16)
rt = Raytracer(sc, max_depth=5)
sb = rt.strip_bytes(1)
strip = torch.zeros(sb, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
rt.render_shard_device(0, 1, strip.data_ptr(), sb); torch.cuda.synchronize()
ts = []
for i in range(4):
    t0 = time.perf_counter()
    rt.render_shard_device(0, 1, strip.data_ptr(), sb); torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
h = hashlib.sha1(strip.cpu().numpy().tobytes()).hexdigest()[:12]
st = N.Stats()
rt.render_shard_device(0, 1, strip.data_ptr(), sb, stats=st, counters=True); torch.cuda.synchronize()
print(os.path.basename(sys.argv[1]), "frame ms min %.2f med %.2f" % (min(ts), sorted(ts)[len(ts)//2]), "sha", h,
      "nodes/ray %.3f tris/ray %.3f" % (st.node_visits / st.rays(), st.tri_tests / st.rays()),
      "wave steps node %.1fM tri %.1fM outer %.1fM lane-steps %.0fM node-util %.2f tri-util %.2f" % (
          st.wave_node_steps / 1e6, st.wave_tri_steps / 1e6, st.wave_outer_steps / 1e6, st.wave_lane_steps / 1e6,
          st.node_visits / 64 / max(1, st.wave_node_steps), st.box_tests / 64 / max(1, st.wave_tri_steps)),
      "trans wave-steps %.1fM lanes/step %.1f" % (st.wave_trans_steps / 1e6, st.trans_lanes / max(1, st.wave_trans_steps)),
      "uniform node steps %.1fM (%.0f %%)" % (st.wave_uniform_node_steps / 1e6, 100.0 * st.wave_uniform_node_steps / max(1, st.wave_node_steps)), flush=True)
rt.close()
