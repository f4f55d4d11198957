"""GPU parity tests: the HIP path (through the C ABI of include/trayracer_hip.h) against the CPU oracle on the same
seeded scenes, against the committed golden fixtures, and -- at BASELINE.json's full sizes -- through size-independent
properties (shard/assemble == single shard, idempotence, remainder strips).

Bar (BASELINE.json north_star): primary-hit triangle ids identical on >= 99.99 % of pixels, RGB8 within 1 LSB on
>= 99.9 % of pixels.  fp32 arithmetic, no tolerance on the ids.  "parity unpinned" by the reference: see DESIGN.md.
"""
import ctypes as C
import os

import numpy as np
import pytest

from raytracing_amd import _native as N
from raytracing_amd import multigpu, standin
from raytracing_amd.raytracer import Raytracer, Render
from raytracing_amd.scene import Light, Object, Scene, Settings, Skybox
from tests._util import assert_parity, scene_from_arrays
from tests.test_oracle_kat import front_camera, mat, single_quad_scene, tri_rows

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
F32 = np.float32


@pytest.fixture(scope="module")
def ctx():
    from raytracing_amd.raytracer import Context
    c = Context(0)
    yield c
    c.close()


def gpu_render(ctx, scene, max_depth=5, want_ids=True, **kw):
    rt = Raytracer(scene, max_depth=max_depth, context=ctx)
    try:
        return rt.render_frame(want_ids=want_ids, **kw)
    finally:
        rt.close()


def test_extension_is_loaded_in_tree():
    lib = N.lib()
    assert os.path.dirname(os.path.abspath(N.LIB_PATH)).endswith("raytracing_amd")
    assert b"gfx950" in lib.trr_version()


# ---- golden fixtures -----------------------------------------------------------------------------------------------------
def test_golden_parity_scene(ctx):
    g = np.load(os.path.join(GOLDEN, "parity_80x60_d8_md3.npz"))
    out = gpu_render(ctx, scene_from_arrays(g), max_depth=3)
    assert_parity(out, g, what="golden parity scene")


def test_golden_anti_aliasing(ctx):
    g = np.load(os.path.join(GOLDEN, "parity_aa4_60x40_d6_md2.npz"))
    out = gpu_render(ctx, scene_from_arrays(g), max_depth=2, want_ids=False)
    assert_parity(out, g, ids=False, what="golden AA scene")


# ---- oracle on the same seeded inputs ----------------------------------------------------------------------------------------
@pytest.mark.parametrize("sky_image", [True, False])
@pytest.mark.parametrize("max_depth", [0, 1, 2, 5, 9])
def test_parity_scene_all_depths(ctx, oracle_mod, sky_image, max_depth):
    sc = standin.parity_scene(200, 150, detail=14, sky_image=sky_image)
    out = gpu_render(ctx, sc, max_depth=max_depth)
    ref = oracle_mod.OracleScene(sc).render(shade=oracle_mod.SHADE_HOISTED, max_depth=max_depth)
    assert_parity(out, ref, what=f"parity scene sky_image={sky_image} max_depth={max_depth}")


def test_parity_against_literal_recursion(ctx, oracle_mod):
    """The reference re-traces the reflection once per light (mod.rs:199-208); the GPU traces it once per hit."""
    sc = standin.parity_scene(120, 90, detail=8)
    out = gpu_render(ctx, sc, max_depth=4)
    ref = oracle_mod.OracleScene(sc).render(shade=oracle_mod.SHADE_LITERAL, max_depth=4)
    assert_parity(out, ref, what="literal recursion")


@pytest.mark.parametrize("samples", [1, 4, 5, 16])
def test_anti_aliasing(ctx, oracle_mod, samples):
    sc = standin.parity_scene(100, 80, detail=8)
    sc.settings.anti_aliasing, sc.settings.samples, sc.settings.seed = True, samples, 11
    out = gpu_render(ctx, sc, max_depth=3, want_ids=False)
    ref = oracle_mod.OracleScene(sc).render(shade=oracle_mod.SHADE_HOISTED, max_depth=3, want_ids=False)
    assert_parity(out, ref, ids=False, what=f"AA {samples} spp")


def test_anti_aliasing_zero_samples_is_black(ctx):
    sc = standin.parity_scene(40, 40, detail=4)
    sc.settings.anti_aliasing, sc.settings.samples = True, 0
    out = gpu_render(ctx, sc, want_ids=False)
    assert (out["rgb"] == 0).all()   # mean of nothing = NaN -> `as u8` = 0 (mod.rs:248-249, render.rs:136)


def test_standin_test_scene_800x600_depth1(ctx, oracle_mod):
    """BASELINE.json configs[0]: the res/test stand-in (floor + car arrangement) 800x600, 1 spp, reflection depth 1."""
    sc = standin.standin_test(800, 600, n_car=120_000, n_floor=30_000)
    out = gpu_render(ctx, sc, max_depth=2)
    ref = oracle_mod.OracleScene(sc).render(shade=oracle_mod.SHADE_HOISTED, max_depth=2)
    assert_parity(out, ref, what="standin_test 800x600")


def test_standin_car_blocks_depth4(ctx, oracle_mod):
    """BASELINE.json configs[1] (1080p, depth-4 reflections) on a block sample the oracle finishes in seconds."""
    sc = standin.standin_car(1920, 1080, n_tris=300_000)
    first, step = 5, 7
    rt = Raytracer(sc, max_depth=5, context=ctx)
    import torch
    sb = rt.strip_bytes(step)
    strip = torch.zeros(sb, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()   # the library renders on its own (non-blocking) stream
    rt.render_shard_device(first, step, strip.data_ptr(), sb)
    torch.cuda.synchronize()
    rt.close()
    ref = oracle_mod.OracleScene(sc).render(shade=oracle_mod.SHADE_HOISTED, max_depth=5, block_first=first, block_step=step,
                                            want_ids=False)
    want = multigpu.strip_from_frame(ref["rgb"], first, step)
    d = np.abs(strip.cpu().numpy().astype(np.int16) - want.astype(np.int16)).reshape(-1, 3).max(axis=1)
    assert (d <= 1).mean() >= 0.999, f"within 1 LSB on {(d <= 1).mean():.6f}"


def test_mirror_scene_depth8(ctx, oracle_mod):
    """BASELINE.json configs[4] in small: mirror spheres over a mirror plane under an image sky, max_depth 9."""
    sc = standin.standin_mirrors(240, 160, n_spheres=9, tris_per_sphere=600, sky=(256, 128))
    out = gpu_render(ctx, sc, max_depth=9)
    ref = oracle_mod.OracleScene(sc).render(shade=oracle_mod.SHADE_HOISTED, max_depth=9)
    assert_parity(out, ref, what="mirrors depth 8")


def test_terrain_scene(ctx, oracle_mod):
    """BASELINE.json configs[3] in small: one heightfield object, illum 2, one light."""
    sc = standin.standin_terrain(240, 160, nx=200, nz=100)
    out = gpu_render(ctx, sc, max_depth=5)
    ref = oracle_mod.OracleScene(sc).render(shade=oracle_mod.SHADE_HOISTED, max_depth=5)
    assert_parity(out, ref, what="terrain")


# ---- edge cases the reference's semantics define ------------------------------------------------------------------------------
def test_empty_scene_and_empty_objects(ctx, oracle_mod):
    sc = Scene(objects=[], lights=[], camera=front_camera(40, 40), settings=Settings())
    out = gpu_render(ctx, sc)
    assert (out["rgb"] == int(0.16 * 255)).all() and (out["hit_obj"] == 0xFFFFFFFF).all()
    # an object without triangles next to a real one; a single-triangle object (the crate returns it with no box test)
    one = tri_rows(((-1, -1, -3), (1, -1, -3), (0, 1, -3)))
    sc = Scene(objects=[Object(), Object(triangles=one, material_index=np.array([-1], np.int32))],
               lights=[Light(np.array([0, 2, 0], F32), np.array([1, 1, 1], F32), 9.0)], camera=front_camera(60, 40), settings=Settings())
    out = gpu_render(ctx, sc)
    ref = oracle_mod.OracleScene(sc).render()
    assert_parity(out, ref, what="empty + single-triangle objects")
    assert (out["hit_obj"] == 1).any()


def test_degenerate_and_nan_triangles(ctx, oracle_mod):
    z = -2.0
    rows = tri_rows(((-1, -1, z), (1, -1, z), (1, 1, z)),            # regular
                    ((0, 0, z + 0.5), (0, 0, z + 0.5), (0, 0, z + 0.5)),     # zero area: culled by try_normalize (triangle.rs:56)
                    ((0, 0, z + 0.4), (3e-3, 0, z + 0.4), (0, 3e-3, z + 0.4)),   # |ab x ac| = 9e-6 <= delta: never hit
                    ((-1, -1, z), (1, 1, z), (-1, 1, z)))
    obj = Object(triangles=rows, material_index=np.zeros(4, np.int32), materials=[mat(kd=(0.4, 0.5, 0.6), illum=2, ks=(1, 1, 1), ns=5.0, dissolve=0.0)])
    sc = Scene(objects=[obj], lights=[Light(np.array([0.5, 0.5, 0], F32), np.array([1, 1, 1], F32), 3.0)],
               camera=front_camera(64, 64), settings=Settings())
    out = gpu_render(ctx, sc)
    ref = oracle_mod.OracleScene(sc).render()
    assert_parity(out, ref, what="degenerate triangles")
    assert not ((out["hit_tri"] == 1) | (out["hit_tri"] == 2)).any()


def test_remainder_strip_and_small_resolutions(ctx, oracle_mod):
    for (w, h) in [(50, 47), (810, 610), (20, 20), (39, 21)]:
        sc = single_quad_scene(w=w, h=h)
        out = gpu_render(ctx, sc, max_depth=1)
        ref = oracle_mod.OracleScene(sc).render(max_depth=1)
        assert np.array_equal(out["rgb"], ref["rgb"]), (w, h)
        bw, bh, nbx, nby = multigpu.tiling(w, h)
        assert (out["rgb"][nby * bh:, :] == 0).all() and (out["rgb"][:, nbx * bw:] == 0).all()
    rt = Raytracer(single_quad_scene(w=19, h=40), context=ctx)
    with pytest.raises(ZeroDivisionError):
        rt.render_frame()
    rt.close()


def test_transparency_march(ctx, oracle_mod):
    pane = tri_rows(((-5, -5, -1.0), (5, -5, -1.0), (5, 5, -1.0)), ((-5, -5, -1.0), (5, 5, -1.0), (-5, 5, -1.0)))
    wall = tri_rows(((-5, -5, -1.3), (5, -5, -1.3), (5, 5, -1.3)), ((-5, -5, -1.3), (5, 5, -1.3), (-5, 5, -1.3)))
    sc = Scene(objects=[Object(triangles=pane, material_index=np.zeros(2, np.int32), materials=[mat(kd=(1, 0, 0), illum=6, dissolve=0.5)]),
                        Object(triangles=wall, material_index=np.zeros(2, np.int32), materials=[mat(kd=(0, 1, 0), illum=0, dissolve=0.0)])],
               lights=[Light(np.array([0, 0, 0.5], F32), np.array([1, 1, 1], F32), 2.0)], camera=front_camera(48, 48), settings=Settings())
    out = gpu_render(ctx, sc, counters=True)
    ref = oracle_mod.OracleScene(sc).render()
    assert_parity(out, ref, what="transparency march")
    assert (out["hit_obj"] == 0).all()          # ids: the FIRST raycast (the pane)
    assert out["stats"].rays_march > 0


def test_transparency_march_in_a_single_object(ctx, oracle_mod):
    """The same pane and wall as ONE object with two materials and a mirror floor: with one object the trace kernel
    keeps only the winner's slot and the origin of the raycast that found it (which the march advances, mod.rs:106);
    k_setup must rebuild the barycentrics from exactly that origin."""
    rows = tri_rows(((-5, -5, -1.0), (5, -5, -1.0), (5, 5, -1.0)), ((-5, -5, -1.0), (5, 5, -1.0), (-5, 5, -1.0)),
                    ((-5, -5, -1.3), (5, -5, -1.3), (5, 5, -1.3)), ((-5, -5, -1.3), (5, 5, -1.3), (-5, 5, -1.3)),
                    ((-5, -0.4, 0.0), (5, -0.4, 0.0), (5, -0.4, -1.3)), ((-5, -0.4, 0.0), (5, -0.4, -1.3), (-5, -0.4, -1.3)))
    mats = [mat(kd=(1, 0, 0), illum=6, dissolve=0.5), mat(kd=(0, 1, 0), illum=0, dissolve=0.0), mat(kd=(0.5, 0.5, 0.9), illum=3, dissolve=0.0)]
    obj = Object(triangles=rows, material_index=np.array([0, 0, 1, 1, 2, 2], np.int32), materials=mats)
    sc = Scene(objects=[obj], lights=[Light(np.array([0, 0, 0.5], F32), np.array([1, 1, 1], F32), 2.0),
                                      Light(np.array([1, 1, -0.5], F32), np.array([1, 0.8, 0.6], F32), 1.0)],
               camera=front_camera(64, 48), settings=Settings())
    out = gpu_render(ctx, sc, counters=True, max_depth=4)
    ref = oracle_mod.OracleScene(sc).render(shade=oracle_mod.SHADE_HOISTED, max_depth=4)
    assert_parity(out, ref, what="transparency march, one object")
    assert out["stats"].rays_march > 0 and out["stats"].rays_reflection > 0


def test_update_camera_lights_instances_between_renders(ctx, oracle_mod):
    sc = standin.parity_scene(96, 72, detail=6)
    rt = Raytracer(sc, max_depth=3, context=ctx)
    a = rt.render_frame(want_ids=True)
    sc.camera.position = sc.camera.position + np.array([0.5, 0.3, -0.4], F32)
    sc.lights = sc.lights[:2]
    sc.objects[2].translation = sc.objects[2].translation + np.array([0.3, -0.2, 0.5], F32)
    rt.set_camera(sc.camera)
    rt.set_lights(sc.lights)
    rt.set_instances(sc.objects)
    b = rt.render_frame(want_ids=True)
    rt.close()
    ref = oracle_mod.OracleScene(sc).render(shade=oracle_mod.SHADE_HOISTED, max_depth=3)
    assert_parity(b, ref, what="after UI edits")
    assert not np.array_equal(a["rgb"], b["rgb"])


def test_stats_ray_accounting_matches_oracle(ctx, oracle_mod, monkeypatch):
    sc = standin.parity_scene(96, 72, detail=6)
    culled = gpu_render(ctx, sc, max_depth=4, counters=True)
    monkeypatch.setenv("TRR_CULL", "0")   # every shadow ray traced, as the oracle does
    out = gpu_render(ctx, sc, max_depth=4, counters=True)
    monkeypatch.delenv("TRR_CULL")
    st, sc_ = out["stats"], culled["stats"]
    # the shadow-ray cull changes no pixel and no count of the algorithm's rays; what it saves shows in
    # rays_shadow_culled, in the traversal counters and in the march queries the culled rays would have made
    assert np.array_equal(culled["rgb"], out["rgb"]) and np.array_equal(culled["hit_tri"], out["hit_tri"])
    assert (sc_.rays_primary, sc_.rays_shadow, sc_.rays_reflection) == (st.rays_primary, st.rays_shadow, st.rays_reflection)
    assert st.rays_shadow_culled == 0 and 0 < sc_.rays_shadow_culled < sc_.rays_shadow
    assert sc_.rays_march <= st.rays_march and sc_.node_visits < st.node_visits
    ref = oracle_mod.OracleScene(sc).render(shade=oracle_mod.SHADE_HOISTED, max_depth=4, want_ids=False)
    # The hoisted oracle makes nearly the de-duplicated closest-hit queries we count.  Two known differences, both
    # result-neutral: it still casts the reflection ray of the deepest level (whose value is skybox(direction) hit or
    # miss, mod.rs:116-127) which the GPU skips, and it casts no reflection ray when every light of a hit was skipped
    # by the 0.01 transmission threshold (mod.rs:171-173) while the GPU always queues one.
    assert abs(st.rays() - ref["counters"].raycasts) <= 0.01 * ref["counters"].raycasts
    assert st.rays_primary == 96 * 72 and st.node_visits > 0 and st.tri_tests > 0


# ---- the Render handle of render.rs:15-25 ----------------------------------------------------------------------------------------
def test_render_handle_progress_cancel_time():
    sc = standin.parity_scene(100, 80, detail=6)
    r = Render(0)
    r.render(sc, max_depth=2)
    r.join()
    assert r.error is None and r.progress == 65535 and r.image.shape == (80, 100, 3) and r.image.any()
    r.cancel = C.c_int32(1)       # set before the block loop starts: no batch is rendered (render.rs:113)
    r.render(sc, max_depth=2)
    r.join()
    assert r.error is None and r.progress == 65535 and not r.image.any()


# ---- properties at BASELINE.json's full sizes ------------------------------------------------------------------------------------
def test_full_size_shards_assemble_to_the_single_gpu_frame(ctx):
    """4K, the 1M-triangle stand-in: N shards rendered one after the other on this GPU and assembled on device equal the
    1-shard frame byte for byte (the multi-GPU test of SURVEY section 8e without N GPUs), and a second render is identical."""
    import torch
    sc = standin.standin_car(3840, 2160, n_tris=1_000_000)
    rt = Raytracer(sc, max_depth=5, context=ctx)
    whole = rt.render_frame()["rgb"]
    again = rt.render_frame()["rgb"]
    assert np.array_equal(whole, again)
    assert whole.any()
    for n in (3, 8):
        sb = rt.strip_bytes(n)
        strips = torch.zeros((n, sb), dtype=torch.uint8, device="cuda")
        frame = torch.zeros((2160, 3840, 3), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()   # the library renders on its own (non-blocking) stream
        for s in range(n):
            rt.render_shard_device(s, n, strips[s].data_ptr(), sb)
        rt.assemble_device(n, strips.data_ptr(), frame.data_ptr())
        rt.ctx.synchronize()
        assert np.array_equal(frame.cpu().numpy(), whole), f"{n} shards"
        assert np.array_equal(multigpu.assemble_host(strips.cpu().numpy(), 3840, 2160, n), whole)
    rt.close()


def test_full_size_aa_batches_are_seamless(ctx):
    """4K x 16 spp needs several block batches per frame: rendering in 1 or 5 shards (different batch boundaries) must
    give the same bytes because the counter RNG depends on (pixel, sample) only."""
    import torch
    sc = standin.standin_car(3840, 2160, n_tris=200_000, anti_aliasing=True, samples=16)
    rt = Raytracer(sc, max_depth=3, context=ctx)
    whole = rt.render_frame()["rgb"]
    n = 5
    sb = rt.strip_bytes(n)
    strips = torch.zeros((n, sb), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    for s in range(n):
        rt.render_shard_device(s, n, strips[s].data_ptr(), sb)
    rt.ctx.synchronize()
    rt.close()
    assert np.array_equal(multigpu.assemble_host(strips.cpu().numpy(), 3840, 2160, n), whole)


# ---- frames in flight / deferred statistics -----------------------------------------------------------------------------
def test_sharded_renderer_pipelined_frames_are_identical(ctx):
    """Two buffer sets alternate (frames_in_flight=2): every frame of a sequence equals the single-call frame, and the
    statistics collected after the sequence equal the synchronous ones."""
    import torch
    sc = standin.parity_scene(320, 240, detail=12)
    sc.settings.anti_aliasing, sc.settings.samples = True, 4
    rt = Raytracer(sc, max_depth=4, context=ctx)
    try:
        want = rt.render_frame()["rgb"]
        sync_stats = N.Stats()
        sb = rt.strip_bytes(1)
        tmp = torch.zeros(sb, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        rt.render_shard_device(0, 1, tmp.data_ptr(), sb, stats=sync_stats)
        sr = multigpu.ShardedRenderer(rt, 0, 1, frames_in_flight=2)
        stats = [N.Stats() for _ in range(5)]
        frames = []
        for k in range(5):
            f = sr.render(stats=stats[k], defer_stats=True)
            # a returned frame stays valid until the frames_in_flight-th call after it: copy it out in stream order
            with torch.cuda.stream(sr.comm_stream):
                frames.append(f.clone())
        torch.cuda.synchronize()
        ctx.collect_stats()
        for k, f in enumerate(frames):
            assert np.array_equal(f.cpu().numpy(), want), f"frame {k} of the pipelined sequence differs"
        for st in stats:
            assert st.rays() == sync_stats.rays() and st.rays_shadow == sync_stats.rays_shadow
            # one batch per pipelined frame (TRR_FLAG_PIPELINED) against four for the lone small frame: one (persistent) trace
            # launch per batch
            assert st.trace_launches * 4 == sync_stats.trace_launches and st.blocks_rendered == 400
            assert st.queue_overflows == 0
            assert st.ms_trace > 0 and st.ms_total > 0
    finally:
        rt.close()


def test_collect_stats_without_pending_frames_is_a_no_op(ctx):
    ctx.collect_stats()
