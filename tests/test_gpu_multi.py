"""The multi-device context (trr_create_multi): the n = 1 instance of its code path -- per-device shard render, ncclGather
through the dlopen'ed RCCL, assemble on the root -- against the plain context's trr_render, byte for byte.  (A test box
has one GPU; the partition at n = 8 is covered on the CPU by tests/test_host.py.)"""
import os

import numpy as np
import pytest

from raytracing_amd import _native as N
from raytracing_amd import standin
from raytracing_amd.raytracer import Context, Raytracer

pytestmark = pytest.mark.gpu


def test_multi_context_of_one_device_equals_the_plain_context(monkeypatch):
    sc = standin.parity_scene(320, 240, detail=10)
    sc.settings.anti_aliasing, sc.settings.samples, sc.settings.seed = True, 4, 3
    plain = Context(0)
    rt = Raytracer(sc, max_depth=4, context=plain)
    want = rt.render_frame()["rgb"]
    rt.close()
    plain.close()

    monkeypatch.setenv("TRR_MULTI_FORCE_RCCL", "1")   # one rank still goes through ncclCommInitAll + ncclGather
    multi = Context(devices=[0])
    try:
        assert multi.device_count == 1
        rt = Raytracer(sc, max_depth=4, context=multi)
        rt.validate()
        assert rt.scene_info()["n_triangles"] == sc.triangle_count()
        calls = []
        out = rt.render_frame(progress=lambda done, total: calls.append((done, total)))
        assert np.array_equal(out["rgb"], want), "multi-device context (n = 1, RCCL gather) differs from the plain context"
        assert calls and calls[-1] == (400, 400)
        assert out["stats"].rays_primary == 320 * 240 * 4 and out["stats"].queue_overflows == 0
        # setters reach the replica; a second frame through the same communicator
        sc.lights = sc.lights[:2]
        rt.set_lights(sc.lights)
        again = rt.render_frame()["rgb"]
        assert not np.array_equal(again, want)
        blocks = {}
        rt.render_blocks(lambda xb, yb, blk: blocks.__setitem__((xb, yb), blk.copy()))
        assert len(blocks) == 400
        bh, bw = 240 // 20, 320 // 20
        for (xb, yb), blk in blocks.items():
            assert np.array_equal(blk, again[yb * bh:(yb + 1) * bh, xb * bw:(xb + 1) * bw])
        # what needs a single-device context says so
        import torch
        t = torch.zeros(rt.strip_bytes(1), dtype=torch.uint8, device="cuda")
        with pytest.raises(N.NativeError):
            rt.render_shard_device(0, 1, t.data_ptr(), t.numel())
        rt.close()
    finally:
        multi.close()


def test_multi_context_argument_checks():
    with pytest.raises(N.NativeError):
        Context(devices=[0, 0])
    with pytest.raises(N.NativeError):
        Context(devices=[])
    with pytest.raises(N.NativeError):
        Context(devices=[4096])


@pytest.mark.parametrize("n", [2, 3, 8])
def test_n_device_logic_rehearsed_on_one_gpu(monkeypatch, n):
    """TRR_MULTI_ALLOW_DUPLICATES: n sub-contexts on the one GPU of the test box run the whole N-device path of the library --
    scene replicas, shard n of N per sub-context (host threads when progress is polled), strips exchanged by peer copies
    (RCCL refuses a device listed twice), assembly on the root -- against the plain context, byte for byte."""
    import ctypes as C
    sc = standin.parity_scene(330, 250, detail=10)   # 330 x 250: a remainder strip on both edges, 400 blocks of 16 x 12
    plain = Context(0)
    rt = Raytracer(sc, max_depth=3, context=plain)
    want = rt.render_frame(want_ids=True)
    rt.close()
    plain.close()
    monkeypatch.setenv("TRR_MULTI_ALLOW_DUPLICATES", "1")
    multi = Context(devices=[0] * n)
    try:
        assert multi.device_count == n
        rt = Raytracer(sc, max_depth=3, context=multi)
        out = rt.render_frame()   # not polled: the host threads only queue the shards
        assert np.array_equal(out["rgb"], want["rgb"])
        assert out["stats"].rays() == want["stats"].rays() and out["stats"].blocks_rendered == 400
        calls = []
        out = rt.render_frame(progress=lambda done, total: calls.append((done, total)))   # polled: one host thread per device
        assert np.array_equal(out["rgb"], want["rgb"])
        assert calls and calls[-1] == (400, 400) and all(a[0] <= b[0] for a, b in zip(calls, calls[1:]))
        blocks = {}
        rt.render_blocks(lambda xb, yb, blk: blocks.__setitem__((xb, yb), blk.copy()), blocks_per_batch=37)
        assert len(blocks) == 400
        bh, bw = 250 // 20, 330 // 20
        for (xb, yb), blk in blocks.items():
            assert np.array_equal(blk, want["rgb"][yb * bh:(yb + 1) * bh, xb * bw:(xb + 1) * bw])
        flag = C.c_int32(1)   # cancelled before the first batch on every device
        assert rt.render_frame(cancel_flag=flag)["cancelled"]
        assert np.array_equal(rt.render_frame()["rgb"], want["rgb"])   # and the context is usable afterwards
        rt.close()
    finally:
        multi.close()


def test_blocks_of_a_multi_device_context_arrive_while_the_devices_still_render(monkeypatch):
    """trr_render_blocks on a context of several devices (rehearsed with three sub-contexts on the one GPU): every device's host
    thread announces its finished batches, the calling thread hands the blocks to the sink in ARRIVAL order (the reference's
    `par_bridge` has no order either, render.rs:109-113) -- each block exactly once, the first one long before the frame is
    done, the bytes those of the plain context."""
    import ctypes as C
    import time
    sc = standin.parity_scene(1280, 720, detail=14)
    sc.settings.anti_aliasing, sc.settings.samples, sc.settings.seed = True, 4, 5
    plain = Context(0)
    rt = Raytracer(sc, max_depth=4, context=plain)
    want = rt.render_frame()["rgb"]
    rt.close()
    plain.close()
    monkeypatch.setenv("TRR_MULTI_ALLOW_DUPLICATES", "1")
    multi = Context(devices=[0, 0, 0])
    try:
        rt = Raytracer(sc, max_depth=4, context=multi)
        rt.render_blocks(lambda xb, yb, blk: None)   # warm-up: work buffers, pinned staging
        bh, bw = 720 // 20, 1280 // 20
        for bpb in (0, 7, 400):
            blocks, times, prog = {}, [], []
            t0 = time.perf_counter()

            def sink(xb, yb, blk):
                assert (xb, yb) not in blocks
                blocks[(xb, yb)] = blk.copy()
                times.append(time.perf_counter() - t0)

            out = rt.render_blocks(sink, blocks_per_batch=bpb, progress=lambda done, total: prog.append((done, total)))
            t_end = time.perf_counter() - t0
            assert not out["cancelled"] and len(blocks) == 400 and out["stats"].blocks_rendered == 400
            for (xb, yb), blk in blocks.items():
                assert np.array_equal(blk, want[yb * bh:(yb + 1) * bh, xb * bw:(xb + 1) * bw]), (bpb, xb, yb)
            assert prog[-1] == (400, 400) and all(a[0] < b[0] for a, b in zip(prog, prog[1:]))
            if bpb != 400:
                # progressive: the first batch is handed over while most of the frame is still to be rendered
                assert prog[0][0] < 100 and len(prog) >= 10, prog[:3]
                assert times[0] < 0.5 * t_end, (times[0], t_end)
        # a cancel raised from inside the sink: every device stops at its next batch; what had been announced is delivered
        flag = C.c_int32(0)
        seen = []

        def cancelling_sink(xb, yb, blk):
            seen.append((xb, yb))
            flag.value = 1

        out = rt.render_blocks(cancelling_sink, blocks_per_batch=20, cancel_flag=flag)
        assert out["cancelled"] and 0 < len(seen) < 400 and len(set(seen)) == len(seen)
        # the context is usable afterwards, and whole frames still go through the gather path
        assert np.array_equal(rt.render_frame()["rgb"], want)
        rt.close()
    finally:
        multi.close()


@pytest.mark.parametrize("where", ["init", "gather"])
def test_an_rccl_error_falls_back_to_peer_copies(monkeypatch, capfd, where):
    """The RCCL path (ncclCommInitAll, grouped ncclGather) has never run between distinct devices; an error from either call must
    cost a warning, not the frame: the strips are then exchanged by peer copies, for the rest of the context's life."""
    sc = standin.parity_scene(320, 240, detail=10)
    plain = Context(0)
    rt = Raytracer(sc, max_depth=3, context=plain)
    want = rt.render_frame()["rgb"]
    rt.close()
    plain.close()
    monkeypatch.setenv("TRR_MULTI_FORCE_RCCL", "1")
    monkeypatch.setenv("TRR_MULTI_INJECT_RCCL_FAILURE", where)
    multi = Context(devices=[0])
    try:
        rt = Raytracer(sc, max_depth=3, context=multi)
        assert np.array_equal(rt.render_frame()["rgb"], want)
        err = capfd.readouterr().err
        assert "exchanging the strips by peer copies" in err and ("ncclCommInitAll" if where == "init" else "ncclGather") in err
        monkeypatch.delenv("TRR_MULTI_INJECT_RCCL_FAILURE")
        assert np.array_equal(rt.render_frame()["rgb"], want)   # stays on peer copies: no second attempt, no second warning
        assert "peer copies" not in capfd.readouterr().err
        rt.close()
    finally:
        multi.close()
