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
