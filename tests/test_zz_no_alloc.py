"""trr_render_shard_device_noalloc (the entry point that allocates nothing, include/trayracer_hip.h) against
trr_render_shard_device through the C ABI: same strip bytes on two calls in a row, the second with a smaller input (the
static scratch array keeps the first call's contents), and a clear refusal of a frame whose blocks do not fit."""
import numpy as np
import pytest

from raytracing_amd import _native as N
from raytracing_amd import standin
from raytracing_amd.raytracer import Context, Raytracer

pytestmark = pytest.mark.gpu


def _both(rt, shard, n_shards):
    import torch
    sb = rt.strip_bytes(n_shards)
    a = torch.full((sb,), 7, dtype=torch.uint8, device="cuda")
    b = torch.full((sb,), 9, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()   # the library renders on its own (non-blocking) stream
    free0 = torch.cuda.mem_get_info()[0]   # hipMemGetInfo: what the DEVICE has free, whoever allocates
    rt.render_shard_device_noalloc(shard, n_shards, a.data_ptr(), sb)
    rt.ctx.synchronize()
    assert torch.cuda.mem_get_info()[0] == free0, "the entry point that allocates nothing changed the device's free memory"
    rt.render_shard_device(shard, n_shards, b.data_ptr(), sb)
    rt.ctx.synchronize()
    return a.cpu().numpy(), b.cpu().numpy()


def test_noalloc_entry_point_matches_render_shard_device():
    ctx = Context(0)
    try:
        assert N.lib().trr_noalloc_scratch_bytes() == 256 << 20
        # first call: 3 objects, textures, mirror + transparent materials, image sky, 4 lights, 4 spp, several batches
        big = standin.parity_scene(640, 480, detail=12)
        big.settings.anti_aliasing, big.settings.samples, big.settings.seed = True, 4, 5
        rt = Raytracer(big, max_depth=4, context=ctx)
        a, b = _both(rt, 0, 1)
        assert b.any() and np.array_equal(a, b), "first call differs"
        rt.close()
        # second call, smaller input: fewer pixels, 1 spp, fewer levels, one shard of three
        small = standin.parity_scene(100, 80, detail=6)
        rt = Raytracer(small, max_depth=2, context=ctx)
        a, b = _both(rt, 1, 3)
        assert b.any() and np.array_equal(a, b), "second (smaller) call differs"
        # and the whole small frame on the same context
        a2, b2 = _both(rt, 0, 1)
        assert np.array_equal(a2, b2)
        # two callers on different streams and a second context of the same GPU share the one array: the per-device event
        # makes them queue up on the device instead of racing (no host synchronisation between the calls here)
        import torch
        ctx2 = Context(0)
        rt2 = Raytracer(small, max_depth=2, context=ctx2)
        sb = rt.strip_bytes(1)
        outs = [torch.zeros(sb, dtype=torch.uint8, device="cuda") for _ in range(6)]
        streams = [torch.cuda.Stream() for _ in range(3)]
        torch.cuda.synchronize()
        for k, o in enumerate(outs):
            (rt if k % 2 == 0 else rt2).render_shard_device_noalloc(0, 1, o.data_ptr(), sb, stream=streams[k % 3].cuda_stream)
        torch.cuda.synchronize()
        for k, o in enumerate(outs):
            assert np.array_equal(o.cpu().numpy(), b2), f"overlapping no-alloc call {k} differs"
        rt2.close()
        ctx2.close()
        rt.close()
        # a frame whose single block exceeds the static array is refused with the figures in the message
        huge = standin.parity_scene(3840, 2160, detail=4)
        huge.settings.anti_aliasing, huge.settings.samples = True, 64
        rt = Raytracer(huge, max_depth=5, context=ctx)
        import torch
        sb = rt.strip_bytes(1)
        t = torch.zeros(sb, dtype=torch.uint8, device="cuda")
        with pytest.raises(N.NativeError) as ei:
            rt.render_shard_device_noalloc(0, 1, t.data_ptr(), sb)
        assert ei.value.code == N.TRR_ERR_INVALID and "static scratch array" in str(ei.value)
        rt.close()
    finally:
        ctx.close()
