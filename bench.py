#!/usr/bin/env python3
"""bench.py -- Mrays/s of the src/raytracer hot path on MI355X.

Workload (BASELINE.json configs[2], the configuration the metric "Mrays/sec at 4K car_objs scene; ms/frame at
1/2/4/8 GPUs" is quoted on): the car_objs STAND-IN scene (raytracing_amd/standin.py -- the real car.obj is a
git-LFS stub), 3840x2160, anti-aliasing on with 16 samples per pixel, max_depth 5 (= "reflection depth 4", the
reference's hard-coded value, render.rs:105), 4 point lights of res/test/config.yaml.  One step = one frame:
every rank renders its blocks of the reference's 20x20 block grid (block id % N == rank) into a device strip, one
gather brings the strips to rank 0, rank 0 de-interleaves them into the row-major RGB8 frame.  The frame is the
same for every N (strong scaling); scene, BVH and all work buffers are resident in HBM before the timed region.

The reference renders ONE frame per click (raytracer/render.rs:39-71), so the timed frames run one after the other
(--frames-in-flight 1): ms_per_step is the latency of a frame and `value` = rays of a frame / that latency.  A second,
untimed-for-the-metric loop with two frames in flight gives the throughput of a frame SEQUENCE (`pipelined` in the line).

Rays are counted by the rule of SURVEY.md section 8(d) (primaries + one shadow query per shaded hit and light + one
reflection per reflective hit + transparency-march queries); the CPU baseline is credited the same rays for the
same blocks, so GPU/CPU ratio == time ratio.  `value` is that count over the frame time -- the rays of the reference
algorithm that the frame ANSWERS per second.  Two kinds of them are answered without a traversal of their own and are listed
in `config`: shadow rays whose light adds exactly nothing at the hit (round 2) and camera rays through image tiles that no
box of the scene projects onto (round 3); `config.mrays_traced_per_s` is the rate of the rays that were traced one by one,
which is what `value` meant in round 2 (BENCH_r02: 7503).

Roofline: the path is fp32 VALU work on cache-resident data -- no MFMA, 3 % of the HBM peak -- so
the roof is the fp32 vector peak.  `achieved` = ALGORITHMIC flops of a frame / frame time, the flops from the counter
build's measured unit counts times the per-unit constants below; nothing in the line can imply more than a peak.
"""
import os as _os

# More hardware queues than HIP's default of 4 (which the GPU box also exports): the library runs up to nine streams per
# frame.  Must be in the environment before the HIP runtime starts, i.e. before torch touches the GPU.
if int(_os.environ.get("GPU_MAX_HW_QUEUES", "4") or 4) < 16:
    _os.environ["GPU_MAX_HW_QUEUES"] = "16"

import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

# ---- per-unit ALGORITHMIC flops (fp32 add / mul / div / sqrt / min / max / compare / select, one each), DESIGN.md section 4 ----
NODE_FLOP = 125      # one lane visiting one 4-wide node: 4 slab tests (6 sub, 6 mul, 10 min/max, 3 compare) + 5-comparator sort of selects
BOX_FLOP = 40        # a leaf triangle's own box (triangle.rs:88-95): 3 add, 12 min/max, then the slab test
TRI_FLOP = 95        # the exact test of triangle.rs:53-85 incl. the interpolated point and its distance (object.rs:201, 206)
RAY_FLOP = 90        # a ray's start per object: object-space origin / direction, norm, 6 divisions (object.rs:186-195)
SETUP_FLOP = 200     # a shaded hit's set-up: exact test again, normal / uv interpolation, two transforms (object.rs:198-219)
LIGHT_FLOP = 60      # shade_impl per (shaded hit, light): normalise, transmission, diffuse + specular terms (mod.rs:148-208)
HIT_FLOP = 40        # shade_impl per shaded hit outside the light loop (mod.rs:129-146)
VALU_PEAK_TFLOPS = 157.3   # fp32 vector peak, /opt/skills/guides/MI355X_MICROARCH.md ("Peak FP32")
HBM_PEAK_GBS = 8000.0      # same guide
N_SIMD, CLOCK_HZ, VALU_CYCLES = 1024, 2.4e9, 2   # 256 CUs x 4 SIMDs; a wave64 VALU instruction issues over 2 cycles (same guide)


def frame_flops(st, n_lights, n_objs=1) -> float:
    leaf_boxes = max(0, st.box_tests - 4 * st.node_visits)
    shaded = st.rays_shadow / max(1, n_lights)
    traced = st.rays() - st.rays_shadow_culled - st.rays_primary_culled   # a culled ray is never started
    return (NODE_FLOP * st.node_visits + BOX_FLOP * leaf_boxes + TRI_FLOP * st.tri_tests + RAY_FLOP * traced * n_objs +
            shaded * (SETUP_FLOP + HIT_FLOP + LIGHT_FLOP * n_lights))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--spp", type=int, default=16)
    ap.add_argument("--tris", type=int, default=1_000_000)
    ap.add_argument("--max-depth", type=int, default=5)
    ap.add_argument("--frames-in-flight", type=int, default=1,
                    help="buffer sets / render streams alternated by consecutive TIMED frames (1 = every frame waits for the one before: latency)")
    ap.add_argument("--no-pipelined", action="store_true", help="skip the second loop (two frames in flight, sequence throughput)")
    ap.add_argument("--no-launch-events", action="store_true",
                    help="experiment: time the frames without per-launch HIP events (the roofline object is then null)")
    ap.add_argument("--build", choices=("host", "device"), default="host",
                    help="where the BVH is built (outside the timed region): host SAH (default, best tree) or GPU LBVH")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=int(os.environ.get("OMP_NUM_THREADS", "16")))
    ap.add_argument("--cpu-runs", type=int, default=3, help="CPU baseline runs; the median is reported")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N "
                             "--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    import torch.distributed as dist
    from raytracing_amd import _native as N
    from raytracing_amd import standin
    from raytracing_amd.multigpu import ShardedRenderer
    from raytracing_amd.raytracer import Raytracer

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    # under torch.distributed.run (RANK set) the process group is always created, also for N = 1, so that the
    # gather / all-reduce path is the one exercised whenever the launcher is used
    distributed = world > 1 or "RANK" in os.environ
    if distributed:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    aa = args.spp > 1
    scene = standin.standin_car(args.width, args.height, n_tris=args.tris, seed=1, anti_aliasing=aa, samples=max(1, args.spp))
    n_lights = len(scene.lights)
    t0 = time.perf_counter()
    rt = Raytracer(scene, max_depth=args.max_depth, device=local_rank, build=args.build)
    upload_s = time.perf_counter() - t0

    def sync_all():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
            torch.cuda.synchronize()

    def timed_loop(sr, steps, stats):
        sync_all()
        t0 = time.perf_counter()
        for k in range(steps):
            sr.render(stats=stats[k] if stats else None, defer_stats=True)   # events recorded now, read after the timed region
        sync_all()
        el = time.perf_counter() - t0
        if distributed:
            t = torch.tensor([el], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    # ---- timed region: exactly K frames; per-launch HIP events (library side, on the launch streams) give the trace
    # kernels' durations over the same frames ------------------------------------------------------------------
    sr = ShardedRenderer(rt, rank, world, use_dist=distributed, frames_in_flight=args.frames_in_flight)
    for _ in range(args.warmup):
        sr.render()
    stats = [] if args.no_launch_events else [N.Stats() for _ in range(args.steps)]
    elapsed = timed_loop(sr, args.steps, stats)
    rt.ctx.collect_stats()

    # rays of the whole frame = sum over ranks of their shard's rays (deterministic per frame)
    if args.no_launch_events:   # ray counts from one untimed frame
        stats = [N.Stats()]
        sr.render(stats=stats[-1])
        torch.cuda.synchronize()
    my_rays, my_culled, my_pculled = stats[-1].rays(), stats[-1].rays_shadow_culled, stats[-1].rays_primary_culled
    if distributed:
        t = torch.tensor([my_rays, my_culled, my_pculled], dtype=torch.int64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        frame_rays, frame_culled, frame_pculled = int(t[0].item()), int(t[1].item()), int(t[2].item())
    else:
        frame_rays, frame_culled, frame_pculled = int(my_rays), int(my_culled), int(my_pculled)
    traced_rays = frame_rays - frame_culled - frame_pculled   # rays with a traversal of their own

    # ---- untimed for the metric: the same K frames with two in flight (throughput of a frame sequence) -------------
    pipelined = None
    if not args.no_pipelined and args.steps > 0:
        sr2 = ShardedRenderer(rt, rank, world, use_dist=distributed, frames_in_flight=2)
        for _ in range(2):
            sr2.render()
        el2 = timed_loop(sr2, args.steps, None)
        pipelined = {"frames_in_flight": 2, "ms_per_frame": round(el2 / args.steps * 1e3, 3),
                     "mrays_per_s": round(frame_rays * args.steps / el2 / 1e6, 2),
                     "mrays_traced_per_s": round(traced_rays * args.steps / el2 / 1e6, 2)}
        del sr2

    # ---- untimed: counter build of rank 0's shard -> nodes visited / triangles tested (algorithmic flops) ----------
    roofline = None
    cst = N.Stats()
    sr.render(stats=cst, counters=True)
    torch.cuda.synchronize()
    if rank == 0 and args.steps > 0 and not args.no_launch_events:
        frame_s = elapsed / args.steps
        launches = sum(s.trace_launches for s in stats)
        ms_trace = sum(s.ms_trace for s in stats)
        flops = frame_flops(cst, n_lights) * (frame_rays / max(1, cst.rays()))   # rank 0's shard scaled to the frame
        achieved = flops / frame_s / 1e12
        # committed PMC passes of this same command (profiles/pmc_frame.json, made by scripts/profile_pmc.sh +
        # scripts/pmc_frame.py at the commit named in the file): VALU instructions and HBM bytes of ONE frame
        pmc = None
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_frame.json")) as f:
                pmc = json.load(f)
        except (OSError, ValueError):
            pass
        one_gpu = world == 1 and pmc is not None
        roofline = {
            "bound": "valu", "kernel": "k_trace_level (camera-ray + secondary-ray launches) + k_trace (drain of the deep levels)",
            "achieved": round(achieved, 3), "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / VALU_PEAK_TFLOPS, 5),
            "traffic": int(pmc["hbm_bytes_per_frame"]) if one_gpu else None,
            "note": "fp32 vector work on an L2 / Infinity-Cache resident scene: no MFMA, HBM far from its peak (hbm_frac); achieved = "
                    "algorithmic flops of the frame / frame time, so no figure here can exceed a peak; see DESIGN.md section 4",
            "algorithmic_gflop_per_frame": round(flops / 1e9, 1),
            "flop_constants": {"node": NODE_FLOP, "leaf_box": BOX_FLOP, "tri": TRI_FLOP, "ray_start": RAY_FLOP, "setup": SETUP_FLOP,
                               "hit": HIT_FLOP, "light": LIGHT_FLOP},
            # the contract's per-launch form: launches of up to 3 batches overlap on separate streams, so their durations add up
            # to more than the frame and this figure is BELOW `achieved`
            "per_launch": {"launches_per_frame": launches // max(1, args.steps), "avg_launch_ms": round(ms_trace / max(1, launches), 4),
                           "achieved": round(flops * args.steps / max(1e-9, ms_trace * 1e-3) / 1e12, 3)},
            "valu_issue_frac": round(pmc["valu_insts_per_frame"] * VALU_CYCLES / (N_SIMD * CLOCK_HZ * frame_s), 4) if one_gpu else None,
            "hbm_frac": round(pmc["hbm_bytes_per_frame"] / frame_s / 1e9 / HBM_PEAK_GBS, 4) if one_gpu else None,
            "pmc_source": {"file": "profiles/pmc_frame.json", "commit": pmc.get("commit"), "frames": pmc.get("frames")} if one_gpu else None,
            "lane_util_node_steps": round(cst.node_visits / max(1, 64 * cst.wave_node_steps), 4),
            "nodes_per_ray": round(cst.node_visits / max(1, cst.rays()), 2),
            "tris_per_ray": round(cst.tri_tests / max(1, cst.rays()), 2),
            "trace_ms_per_frame_summed": round(ms_trace / args.steps, 3),
            "shade_ms_per_frame_summed": round(sum(s.ms_shade for s in stats) / args.steps, 3),
        }

    # ---- untimed, rank 0 at N=1 only: the CPU oracle in reference-literal mode on a bounded block sample ---------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle
        first, step = 0, 2   # every second block: 200 of the 400 blocks, spread evenly over rows and columns (~7 s of CPU per run)
        gst = N.Stats()
        sb = rt.strip_bytes(step)
        tmp = torch.zeros(sb, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        rt.render_shard_device(first, step, tmp.data_ptr(), sb, stats=gst)
        torch.cuda.synchronize()
        osc = oracle.OracleScene(scene)
        secs, res = [], None
        for _ in range(max(1, args.cpu_runs)):
            res = osc.render(max_depth=args.max_depth, accel=oracle.ACCEL_BVH, shade=oracle.SHADE_LITERAL,
                             threads=args.cpu_threads, block_first=first, block_step=step, want_ids=False)
            secs.append(res["counters"].seconds)
        cn = res["counters"]
        med = statistics.median(secs)
        # parity spot check of the very blocks that were timed
        from raytracing_amd.multigpu import strip_from_frame
        gpu_strip = tmp.cpu().numpy()
        ref_strip = strip_from_frame(res["rgb"], first, step)
        nb = len(range(first, 400, step))
        npx = nb * (args.width // 20) * (args.height // 20)
        d = np.abs(gpu_strip[:npx * 3].astype(np.int16) - ref_strip[:npx * 3].astype(np.int16)).reshape(-1, 3).max(axis=1)
        cpu = {
            "value": round(gst.rays() / med / 1e6, 4), "unit": "Mrays/s", "cores": args.cpu_threads, "kind": "port",
            "sample": f"blocks {first}::{step} ({nb} of 400 blocks of the same {args.width}x{args.height} {args.spp}spp frame), "
                      f"C++ restatement in reference-literal mode (per-call matrix inverse, one-triangle-leaf BVH, all "
                      f"candidates tested, reflection re-traced per light), {args.cpu_threads} threads over blocks, "
                      f"median of {len(secs)} runs",
            "seconds": round(med, 3), "seconds_runs": [round(x, 3) for x in secs], "rays_credited": int(gst.rays()),
            "literal_raycasts": int(cn.raycasts), "host_cpus": os.cpu_count(),
            "parity_within_1lsb": round(float((d <= 1).mean()), 6), "parity_exact": round(float((d == 0).mean()), 6),
        }
        osc.close()

    if rank == 0:
        ms = elapsed / max(1, args.steps) * 1e3
        value = frame_rays * args.steps / elapsed / 1e6 if elapsed > 0 else 0.0
        out = {
            "metric": "Mrays/sec (whole node) at 4K car_objs scene", "value": round(value, 2), "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"car_objs stand-in (procedural, {scene.triangle_count()} triangles, one object) "
                                   f"{args.width}x{args.height}, {args.spp} spp, max_depth {args.max_depth} "
                                   f"(reflection depth {args.max_depth - 1}), {n_lights} lights, tiles sharded over {world} GPU(s)",
                       "rays_per_frame": frame_rays, "rays_traced_per_frame": traced_rays,
                       "shadow_rays_culled_per_frame": frame_culled, "camera_rays_culled_per_frame": frame_pculled,
                       "rays_note": "value counts the rays of the reference algorithm (SURVEY.md 8d: the count the CPU baseline is credited too) "
                                    "that a frame answers; shadow_rays_culled_per_frame of them need no answer (their light adds exactly nothing "
                                    "at the hit) and camera_rays_culled_per_frame are answered 'no hit' for a whole 8x8-pixel tile at once (no box "
                                    "of the scene projects onto it) -- same image bit for bit, DESIGN.md 4.  mrays_traced_per_s counts only the "
                                    "rays traced one by one: round 2's meaning of value",
                       "mrays_traced_per_s": round(traced_rays * args.steps / elapsed / 1e6, 2) if elapsed > 0 else 0.0,
                       "scene_upload_s": round(upload_s, 2), "bvh_build": args.build,
                       "frames_in_flight": args.frames_in_flight,
                       # HIP hardware queues the library planned its streams for (this script asks for 16 before HIP starts; the
                       # box's own default is 4, with which frames of <= 2^26 samples lose the split schedule -- the headline
                       # frame is above that limit and does not use it either way)
                       "gpu_max_hw_queues": int(N.lib().trr_hw_queues())},
            "pipelined": pipelined, "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    rt.close()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
