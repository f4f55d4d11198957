"""Counter build of one frame: per-ray and per-wavefront step counts (SIMD efficiency) of the trace kernel.
Usage: python scripts/experiments/counters.py [config [max_depth]]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from raytracing_amd import _native as N  # noqa: E402
if os.environ.get("TRR_LIB"):
    N.LIB_PATH = os.environ["TRR_LIB"]
from raytracing_amd import standin  # noqa: E402
from raytracing_amd.raytracer import Raytracer  # noqa: E402

c = sys.argv[1] if len(sys.argv) > 1 else "3"
cfgs = {
    "1": (lambda: standin.standin_test(800, 600), 2),
    "2": (lambda: standin.standin_car(1920, 1080), 5),
    "3": (lambda: standin.standin_car(3840, 2160, anti_aliasing=True, samples=16), 5),
    "4": (lambda: standin.standin_terrain(3840, 2160), 5),
    "5": (lambda: standin.standin_mirrors(3840, 2160), 9),
}
make, md = cfgs[c]
if len(sys.argv) > 2:
    md = int(sys.argv[2])   # e.g. 0: camera rays only
rt = Raytracer(make(), max_depth=md)
sb = rt.strip_bytes(1)
strip = torch.zeros(sb, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
st = N.Stats()
rt.render_shard_device(0, 1, strip.data_ptr(), sb, stats=st, counters=True)
torch.cuda.synchronize()
rays = st.rays()
g = 64.0 / max(1, rays)
print(f"cfg{c}: rays {rays} (primary {st.rays_primary} shadow {st.rays_shadow} refl {st.rays_reflection} march {st.rays_march}; shadow rays culled {st.rays_shadow_culled} = {st.rays_shadow_culled / max(1, st.rays_shadow):.3f}; camera rays culled {st.rays_primary_culled} = {st.rays_primary_culled / max(1, st.rays_primary):.3f})")
print(f"  per ray: nodes {st.node_visits / rays:.2f} tris {st.tri_tests / rays:.2f} boxes {st.box_tests / rays:.2f}")
print(f"  per 64 rays: wave node steps {st.wave_node_steps * g:.2f} tri steps {st.wave_tri_steps * g:.2f} outer {st.wave_outer_steps * g:.2f} "
      f"trans {st.wave_trans_steps * g:.2f} (lanes per trans {st.trans_lanes / max(1, st.wave_trans_steps):.1f})")
print(f"  lane utilisation: node steps {st.node_visits / max(1, 64 * st.wave_node_steps):.3f}, "
      f"lanes with a ray per outer step {st.wave_lane_steps / max(1, 64 * st.wave_outer_steps):.3f}, "
      f"uniform node steps {st.wave_uniform_node_steps / max(1, st.wave_node_steps):.3f}")
leaf_boxes = st.box_tests - 4 * st.node_visits
print(f"  leaf steps: own-box tests per wave step {leaf_boxes / max(1, st.wave_tri_steps):.1f} of 128 lane-slots = {leaf_boxes / max(1, 128 * st.wave_tri_steps):.3f}")
if os.environ.get("TRR_COUNT_WAIT"):   # a -DTRR_COUNT_WAIT build counts waiting lanes in the uniform-node-step counter
    print(f"  lanes waiting with a finished traversal per outer step {st.wave_uniform_node_steps / max(1, 64 * st.wave_outer_steps):.3f}")
if os.environ.get("TRR_COUNT_DISTINCT"):   # a -DTRR_COUNT_DISTINCT build sums the distinct nodes per node step in that counter
    print(f"  distinct nodes per wave node step {st.wave_uniform_node_steps / max(1, st.wave_node_steps):.2f} (lanes per step {st.node_visits / max(1, st.wave_node_steps):.1f})")
print(f"  ms_total {st.ms_total:.2f} trace {st.ms_trace:.2f} shade {st.ms_shade:.2f} launches {st.trace_launches} overflows {st.queue_overflows}")
rt.close()
