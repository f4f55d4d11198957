"""GPU tests of the progressive render (trr_render_blocks): the reference's sink receives
each finished block as (pixels, x_block, y_block) while the render goes on (render.rs:151-186).  Every block must be
delivered exactly once, in row-major order, with the bytes trr_render puts at the same place; the remainder strip
(render.rs:100-102) is never delivered; a cancel between batches stops the stream."""
import ctypes as C

import numpy as np
import pytest

from raytracing_amd import multigpu, standin
from raytracing_amd.raytracer import Raytracer, Render

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from raytracing_amd.raytracer import Context
    c = Context(0)
    yield c
    c.close()


def collect(rt, **kw):
    h, w = rt.height, rt.width
    img = np.zeros((h, w, 3), np.uint8)
    order = []

    def sink(xb, yb, block):
        bh, bw = block.shape[:2]
        order.append((yb, xb))
        img[yb * bh:(yb + 1) * bh, xb * bw:(xb + 1) * bw] = block

    progress = []
    res = rt.render_blocks(sink, progress=lambda d, t: progress.append((d, t)), **kw)
    return img, order, progress, res


@pytest.mark.parametrize("blocks_per_batch", [0, 7, 1000])
def test_blocks_cover_the_frame_once_in_order_with_the_frames_bytes(ctx, blocks_per_batch):
    sc = standin.parity_scene(205, 150, detail=10)       # blocks of 10 x 7, 20 x 21 of them, 5 px / 3 px remainder
    rt = Raytracer(sc, max_depth=4, context=ctx)
    try:
        want = rt.render_frame()["rgb"]
        img, order, progress, res = collect(rt, blocks_per_batch=blocks_per_batch)
    finally:
        rt.close()
    bw, bh, nbx, nby = multigpu.tiling(205, 150)
    assert (nbx, nby) == (20, 21) and not res["cancelled"]
    assert order == [(y, x) for y in range(nby) for x in range(nbx)]
    assert np.array_equal(img, want)
    assert (img[nby * bh:] == 0).all() and (img[:, nbx * bw:] == 0).all()
    per = blocks_per_batch or nbx
    n_batches = -(-nbx * nby // min(per, nbx * nby))
    assert len(progress) == n_batches and progress[-1] == (nbx * nby, nbx * nby)
    assert res["stats"].blocks_rendered == nbx * nby


def test_anti_aliased_blocks_equal_the_frame(ctx):
    sc = standin.parity_scene(120, 100, detail=8)
    sc.settings.anti_aliasing, sc.settings.samples, sc.settings.seed = True, 4, 5
    rt = Raytracer(sc, max_depth=3, context=ctx)
    try:
        want = rt.render_frame()["rgb"]
        img, order, _, _ = collect(rt, blocks_per_batch=33)
    finally:
        rt.close()
    assert len(order) == len(set(order)) == 400 and np.array_equal(img, want)


def test_cancel_from_the_sink_stops_after_the_batch(ctx):
    sc = standin.parity_scene(200, 160, detail=8)
    rt = Raytracer(sc, max_depth=3, context=ctx)
    cancel = C.c_int32(0)
    got = []

    def sink(xb, yb, block):
        got.append((yb, xb))
        cancel.value = 1           # the UI thread's cancel (statusbar.rs:144), observed before the next batch (render.rs:113)

    try:
        res = rt.render_blocks(sink, blocks_per_batch=20, cancel_flag=cancel)
    finally:
        rt.close()
    assert res["cancelled"] and got == [(0, x) for x in range(20)]


def test_render_handle_fills_its_image_block_by_block():
    sc = standin.parity_scene(160, 120, detail=8)
    r = Render(0)
    r.render(sc, blocks_per_batch=40)
    r.join()
    assert r.error is None and r.progress == 65535
    rt = Raytracer(sc)
    try:
        assert np.array_equal(r.image, rt.render_frame()["rgb"])
    finally:
        rt.close()


def test_small_resolution_is_refused_like_the_frame_form(ctx):
    sc = standin.parity_scene(19, 40, detail=4)
    rt = Raytracer(sc, context=ctx)
    try:
        with pytest.raises(ZeroDivisionError):
            rt.render_blocks(lambda *a: None)
    finally:
        rt.close()
