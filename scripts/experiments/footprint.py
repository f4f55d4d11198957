"""Does the scene's memory footprint set the cost of a traversal step?  The 4K x 16 spp stand-in frame at several triangle
counts: frame time next to the wave-level step counts of the counter build, i.e. SIMD-time per wavefront step.
Usage: python scripts/experiments/footprint.py [n_tris,...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from raytracing_amd import _native as N  # noqa: E402
from raytracing_amd import standin  # noqa: E402
from raytracing_amd.raytracer import Raytracer  # noqa: E402

sizes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1000000,250000,60000,15000").split(",")]
for n in sizes:
    rt = Raytracer(standin.standin_car(3840, 2160, n_tris=n, anti_aliasing=True, samples=16), max_depth=5)
    sb = rt.strip_bytes(1)
    strip = torch.zeros(sb, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(4):
        t0 = time.perf_counter()
        rt.render_shard_device(0, 1, strip.data_ptr(), sb)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) * 1e3)
    st = N.Stats()
    rt.render_shard_device(0, 1, strip.data_ptr(), sb, stats=st, counters=True)
    torch.cuda.synchronize()
    steps = st.wave_node_steps + st.wave_tri_steps + st.wave_trans_steps
    # 1024 SIMDs; SIMD-nanoseconds available per wavefront step
    print(f"n_tris {n}: frame {best:.2f} ms, rays {st.rays() / 1e6:.1f} M, nodes/ray {st.node_visits / st.rays():.2f} tris/ray {st.tri_tests / st.rays():.2f}, "
          f"wave steps node {st.wave_node_steps / 1e6:.1f} M tri {st.wave_tri_steps / 1e6:.1f} M trans {st.wave_trans_steps / 1e6:.1f} M, "
          f"SIMD-ns per step {best * 1e6 * 1024 / steps:.0f}, node-step lane util {st.node_visits / max(1, 64 * st.wave_node_steps):.3f}", flush=True)
    rt.close()
