"""Tree quality of one builder setting on one BASELINE.json configuration: the counter build's per-ray step counts (they
depend on the tree and not on the clock) and the best of eight lone frames.
Usage: python scripts/experiments/tree_quality.py <config 1..5> <host|device>      (builder knobs through the environment)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from raytracing_amd import _native as N  # noqa: E402
from raytracing_amd import standin  # noqa: E402
from raytracing_amd.raytracer import Raytracer  # noqa: E402

c, build = sys.argv[1], sys.argv[2]
cfgs = {
    "1": (lambda: standin.standin_test(800, 600), 2),
    "2": (lambda: standin.standin_car(1920, 1080), 5),
    "3": (lambda: standin.standin_car(3840, 2160, anti_aliasing=True, samples=16), 5),
    "4": (lambda: standin.standin_terrain(3840, 2160), 5),
    "5": (lambda: standin.standin_mirrors(3840, 2160), 9),
}
make, md = cfgs[c]
rt = Raytracer(make(), max_depth=md, build=build)
info = rt.scene_info()
sb = rt.strip_bytes(1)
strip = torch.zeros(sb, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
st = N.Stats()
rt.render_shard_device(0, 1, strip.data_ptr(), sb, stats=st, counters=True)
torch.cuda.synchronize()
rays = max(1, st.rays() - st.rays_primary_culled)
g = 64.0 / rays
line = (f"cfg{c} {build:6s} build {info['ms_build']:7.1f} ms nodes {info['n_nodes']:8d} depth {info['bvh_depth']:3d} | per traced ray: nodes {st.node_visits / rays:6.2f} "
        f"tris {st.tri_tests / rays:5.2f} | per 64 rays: node steps {st.wave_node_steps * g:6.2f} tri steps {st.wave_tri_steps * g:5.2f}")
ms = []
for _ in range(8):
    s2 = N.Stats()
    rt.render_shard_device(0, 1, strip.data_ptr(), sb, stats=s2)
    torch.cuda.synchronize()
    ms.append(s2.ms_total)
print(line + f" | frame {min(ms):6.2f} ms (median {sorted(ms)[4]:.2f})", flush=True)
rt.close()
