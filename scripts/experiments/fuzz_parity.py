"""Randomised parity sweep (GPU vs CPU oracle) over scene seeds, cameras, depths, sample counts, sky kinds and every
schedule / layout switch of the library.  Prints one line per case and a summary; exits non-zero on any pixel or id
difference beyond the north_star bar.  Usage: python scripts/experiments/fuzz_parity.py [cases] [seed]"""
import os
import sys

if int(os.environ.get("GPU_MAX_HW_QUEUES", "4") or 4) < 16:
    os.environ["GPU_MAX_HW_QUEUES"] = "16"   # before the HIP runtime starts

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
from oracle import oracle  # noqa: E402
from raytracing_amd import standin  # noqa: E402
from raytracing_amd.raytracer import Context, Raytracer  # noqa: E402
from raytracing_amd.scene import Camera  # noqa: E402

F32 = np.float32
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
ENVS = [
    {}, {"TRR_SCHEDULE": "levels"}, {"TRR_SCHEDULE": "persistent"}, {"TRR_SCHEDULE": "levels", "TRR_DRAIN_LEVEL": "0"},
    {"TRR_SCHEDULE": "levels", "TRR_DRAIN_LEVEL": "1"}, {"TRR_SCHEDULE": "levels", "TRR_SHADOW": "wide"},
    {"TRR_SCHEDULE": "levels", "TRR_NODES": "quantised"}, {"TRR_SCHEDULE": "levels", "TRR_BATCHES": "3"},
    {"TRR_SCHEDULE": "levels", "TRR_RESOLVE": "two_pass"}, {"TRR_SCHEDULE": "levels", "TRR_SPLIT": "1"},
    {"TRR_SCHEDULE": "levels", "TRR_CULL": "0"}, {"TRR_SCHEDULE": "persistent", "TRR_CULL": "0"},
    {"TRR_SCHEDULE": "levels", "TRR_SPLIT": "0", "TRR_DRAIN_LEVEL": "1"}, {"TRR_SCHEDULE": "levels", "TRR_SPLIT": "1", "TRR_SHADOW": "wide"},
    {"TRR_CAMCULL": "0"}, {"TRR_SCHEDULE": "persistent", "TRR_CAMCULL": "0"}, {"TRR_SCHEDULE": "levels", "TRR_STAGE": "1"},
    {"TRR_SCHEDULE": "levels", "TRR_STAGE": "1", "TRR_SPLIT": "1"},
    {"TRR_BVH_COLLAPSE": "greedy"}, {"TRR_TREELET_MAX": "0"}, {"TRR_TREELET_MAX": "40", "TRR_TREELET_SMALL": "40"},
    {"TRR_TREELET_MAX": "100000", "TRR_TREELET_SMALL": "3"}, {"TRR_SPLIT_DRAIN": "1"},
]
KEYS = sorted({k for e in ENVS for k in e})
ctx = Context(0)
bad = 0
for case in range(n_cases):
    w, h = int(rng.integers(3, 12)) * 20 + int(rng.integers(0, 20)), int(rng.integers(3, 9)) * 20 + int(rng.integers(0, 20))
    sc = standin.parity_scene(w, h, detail=int(rng.integers(4, 20)), seed=int(rng.integers(0, 1000)), sky_image=bool(rng.integers(0, 2)))
    ang = rng.uniform(0, 2 * np.pi)
    dist = float(rng.choice([2.5, 6.0, 14.0, 60.0]))
    sc.camera = Camera(position=np.array([dist * np.cos(ang), rng.uniform(0.2, 6.0), dist * np.sin(ang)], F32),
                       look_at=np.array([rng.uniform(-1, 1), rng.uniform(0, 1.5), rng.uniform(-1, 1)], F32),
                       fov=float(np.deg2rad(rng.uniform(8 if dist > 10 else 30, 95))), resolution=(w, h))
    spp = int(rng.choice([1, 1, 2, 4, 5, 16]))
    sc.settings.anti_aliasing, sc.settings.samples = spp > 1, spp
    md = int(rng.choice([0, 1, 2, 3, 5, 8]))
    env = ENVS[case % len(ENVS)]
    for k in KEYS:
        os.environ.pop(k, None)
    os.environ.update(env)
    build = "device" if (case % 3 == 2 or any(k.startswith("TRR_TREELET") for k in env)) else "host"   # both BVH builders
    rt = Raytracer(sc, max_depth=md, context=ctx, build=build)
    out = rt.render_frame(want_ids=spp == 1)
    rt.close()
    osc = oracle.OracleScene(sc)
    ref = osc.render(shade=oracle.SHADE_HOISTED, max_depth=md, want_ids=spp == 1, threads=8)
    osc.close()
    d = np.abs(out["rgb"].astype(np.int16) - ref["rgb"].astype(np.int16)).max(axis=-1)
    ids = 1.0 if spp > 1 else float(((out["hit_obj"] == ref["hit_obj"]) & (out["hit_tri"] == ref["hit_tri"])).mean())
    ok = (d <= 1).mean() >= 0.999 and ids >= 0.9999
    bad += 0 if ok else 1
    print(f"case {case:2d} {w}x{h} spp {spp:2d} depth {md} cam dist {dist:4.1f} build {build} env {env or 'default'}: exact {float((d == 0).mean()):.6f} "
          f"within1 {float((d <= 1).mean()):.6f} max {int(d.max())} ids {ids:.6f} {'ok' if ok else 'FAIL'}", flush=True)
ctx.close()
print("failures:", bad)
sys.exit(1 if bad else 0)
