"""Two ranks of the tile-sharded render on ONE GPU (two processes, gloo with host transport): everything of the N > 1
path except the RCCL transport itself -- block ownership, strips, frames in flight on alternating streams and work sets
(TRR_FLAG_PIPELINED), the gather order, the root's assemble.  The root's frames must equal the single-rank frame byte
for byte (BASELINE.json north_star: the image is identical for any GPU count)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from raytracing_amd import standin
    from raytracing_amd.multigpu import ShardedRenderer
    from raytracing_amd.raytracer import Raytracer

    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    sc = standin.parity_scene(320, 240, detail=10)
    sc.settings.anti_aliasing, sc.settings.samples, sc.settings.seed = True, 4, 3
    rt = Raytracer(sc, max_depth=4, device=0)
    sr = ShardedRenderer(rt, rank, world, use_dist=True, frames_in_flight=2)
    assert sr.transport == "host"
    frames = []
    for _ in range(3):
        f = sr.render()
        torch.cuda.synchronize()
        if rank == 0:
            frames.append(f.cpu().numpy().copy())
    if rank == 0:
        single = rt.render_frame()["rgb"]
        q.put((frames, single))
    dist.barrier()
    rt.close()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_assemble_the_single_rank_frame():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    frames, single = q.get(timeout=150)
    for p in procs:
        p.join(timeout=150)
        assert p.exitcode == 0
    assert single.any()
    for f in frames:
        assert np.array_equal(f, single)
