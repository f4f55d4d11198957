#!/usr/bin/env python3
"""bench.py -- Mrays/s of the src/raytracer hot path on MI355X.

Workload (BASELINE.json configs[2], the configuration the metric "Mrays/sec at 4K car_objs scene; ms/frame at
1/2/4/8 GPUs" is quoted on): the car_objs STAND-IN scene (raytracing_amd/standin.py -- the real car.obj is a
git-LFS stub), 3840x2160, anti-aliasing on with 16 samples per pixel, max_depth 5 (= "reflection depth 4", the
reference's hard-coded value, render.rs:105), 4 point lights of res/test/config.yaml.  One step = one frame:
every rank renders its blocks of the reference's 20x20 block grid (block id % N == rank) into a device strip, one
gather brings the strips to rank 0, rank 0 de-interleaves them into the row-major RGB8 frame.  The frame is the
same for every N (strong scaling); scene, BVH and all work buffers are resident in HBM before the timed region.
Consecutive frames alternate between two buffer sets / render streams (--frames-in-flight 2), so the head of frame k+1
overlaps the tail of frame k; ms_per_step is elapsed / K (throughput), roofline.frame_latency_ms is one frame's own
first-kernel-start to last-kernel-end time.

Rays are counted by the rule of SURVEY.md section 8(d) (primaries + one shadow query per shaded hit and light + one
reflection per reflective hit + transparency-march queries); the CPU baseline is credited the same rays for the
same blocks, so GPU/CPU ratio == time ratio.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

# per-unit ALGORITHMIC bytes of the trace kernel (DESIGN.md "Roofline"): one visited BVH4 node = one 128-B line,
# one tested triangle = one 64-B record, plus each ray's queue record in (32 B; camera rays are generated in
# registers) and its result out (32-B hit record for camera/reflection rays, 4-B occluder code for shadow rays).
NODE_B, TRI_B, RAY_IN_B, HIT_OUT_B, OCC_OUT_B = 128, 64, 32, 32, 4
HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md


def trace_bytes(st) -> float:
    secondary = st.rays_shadow + st.rays_reflection
    return (NODE_B * st.node_visits + TRI_B * st.tri_tests + RAY_IN_B * secondary +
            HIT_OUT_B * (st.rays_primary + st.rays_reflection) + OCC_OUT_B * st.rays_shadow)


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
    ap.add_argument("--frames-in-flight", type=int, default=2,
                    help="buffer sets / render streams alternated by consecutive frames (1 = each frame waits for the previous one)")
    ap.add_argument("--no-launch-events", action="store_true",
                    help="experiment: time the frames without per-launch HIP events (the roofline object is then null)")
    ap.add_argument("--build", choices=("host", "device"), default="host",
                    help="where the BVH is built (outside the timed region): host SAH (default, best tree) or GPU LBVH")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=int(os.environ.get("OMP_NUM_THREADS", "16")))
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
    t0 = time.perf_counter()
    rt = Raytracer(scene, max_depth=args.max_depth, device=local_rank, build=args.build)
    upload_s = time.perf_counter() - t0
    sr = ShardedRenderer(rt, rank, world, use_dist=distributed, frames_in_flight=args.frames_in_flight)

    def sync_all():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        sr.render()
    sync_all()

    # ---- timed region: exactly K frames; per-launch HIP events (library side, on the launch stream) give the
    # trace kernel's durations over the same frames ----------------------------------------------------------
    stats = [N.Stats() for _ in range(args.steps)]
    sync_all()
    t0 = time.perf_counter()
    for k in range(args.steps):
        sr.render(stats=None if args.no_launch_events else stats[k], defer_stats=True)   # events recorded now, read after the timed region
    sync_all()
    elapsed = time.perf_counter() - t0
    rt.ctx.collect_stats()
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # rays of the whole frame = sum over ranks of their shard's rays (deterministic per frame)
    if args.no_launch_events:   # ray counts from one untimed frame
        sr.render(stats=stats[-1])
        torch.cuda.synchronize()
    my_rays = stats[-1].rays() if stats else 0
    if distributed:
        t = torch.tensor([my_rays], dtype=torch.int64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        frame_rays = int(t.item())
    else:
        frame_rays = int(my_rays)

    # ---- untimed: counter build of rank 0's shard -> nodes visited / triangles tested (algorithmic bytes) -------
    roofline = None
    cst = N.Stats()
    sr.render(stats=cst, counters=True)
    torch.cuda.synchronize()
    if rank == 0 and args.steps > 0 and not args.no_launch_events:
        launches = sum(s.trace_launches for s in stats)
        ms_trace = sum(s.ms_trace for s in stats)
        avg_ms = ms_trace / max(1, launches)
        bytes_per_launch = trace_bytes(cst) / max(1, cst.trace_launches)
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        # HBM traffic per launch: not measurable from inside the process; taken from the committed PMC passes of this same
        # command (profiles/hbm_traffic.json, made by scripts/profile_pmc.sh + scripts/pmc_traffic.py), else null
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as f:
                traffic = int(json.load(f)["kernels"]["k_trace"]["hbm_bytes_per_launch"])
        except (OSError, KeyError, ValueError):
            pass
        frame_s = elapsed / args.steps
        roofline = {
            "bound": "hbm", "kernel": "k_trace", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
            # launches of up to 3 batches overlap on separate streams, so launch durations add up to more than the
            # frame; the second figure divides the same algorithmic bytes by the whole frame time instead
            "achieved_over_frame_time": round(trace_bytes(cst) / frame_s / 1e9, 2),
            "note": "algorithmic bytes; the working set is Infinity-Cache/L2 resident, see DESIGN.md section 4",
            "avg_launch_ms": round(avg_ms, 4), "launches_per_frame": launches // max(1, args.steps),
            "algorithmic_bytes_per_launch": int(bytes_per_launch),
            "nodes_per_ray": round(cst.node_visits / max(1, cst.rays()), 2),
            "tris_per_ray": round(cst.tri_tests / max(1, cst.rays()), 2),
            "frame_latency_ms": round(sum(s.ms_total for s in stats) / args.steps, 3),
            "trace_ms_per_frame": round(ms_trace / args.steps, 3),
            "shade_ms_per_frame": round(sum(s.ms_shade for s in stats) / args.steps, 3),
        }

    # ---- untimed, rank 0 at N=1 only: the CPU oracle in reference-literal mode on a bounded block sample ---------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle
        first, step = 0, 2   # every second block: 200 of the 400 blocks, spread evenly over rows and columns (~8-15 s of CPU)
        gst = N.Stats()
        sb = rt.strip_bytes(step)
        tmp = torch.zeros(sb, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        rt.render_shard_device(first, step, tmp.data_ptr(), sb, stats=gst)
        torch.cuda.synchronize()
        osc = oracle.OracleScene(scene)
        res = osc.render(max_depth=args.max_depth, accel=oracle.ACCEL_BVH, shade=oracle.SHADE_LITERAL,
                         threads=args.cpu_threads, block_first=first, block_step=step, want_ids=False)
        cn = res["counters"]
        # parity spot check of the very blocks that were timed
        from raytracing_amd.multigpu import strip_from_frame
        gpu_strip = tmp.cpu().numpy()
        ref_strip = strip_from_frame(res["rgb"], first, step)
        nb = len(range(first, 400, step))
        npx = nb * (args.width // 20) * (args.height // 20)
        d = np.abs(gpu_strip[:npx * 3].astype(np.int16) - ref_strip[:npx * 3].astype(np.int16)).reshape(-1, 3).max(axis=1)
        cpu = {
            "value": round(gst.rays() / cn.seconds / 1e6, 4), "unit": "Mrays/s", "cores": args.cpu_threads, "kind": "port",
            "sample": f"blocks {first}::{step} ({nb} of 400 blocks of the same {args.width}x{args.height} {args.spp}spp frame), "
                      f"C++ restatement in reference-literal mode (per-call matrix inverse, one-triangle-leaf BVH, all "
                      f"candidates tested, reflection re-traced per light), {args.cpu_threads} threads over blocks",
            "seconds": round(cn.seconds, 3), "rays_credited": int(gst.rays()), "literal_raycasts": int(cn.raycasts),
            "host_cpus": os.cpu_count(),
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
                                   f"(reflection depth {args.max_depth - 1}), 4 lights, tiles sharded over {world} GPU(s)",
                       "rays_per_frame": frame_rays, "scene_upload_s": round(upload_s, 2), "bvh_build": args.build,
                       "frames_in_flight": args.frames_in_flight},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    rt.close()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
