"""GPU: every way the library can schedule and lay out a batch gives the same bytes.  The schedule (levels, persistent,
levels-then-drain from any level), the shadow-queue entry format (16-B compact / 32-B ray records) and the resolve form
(fused / two passes) are chosen per call from the frame's size and settings; the environment switches the library offers
for experiments force each alternative here, on a scene with reflections, transparency, textures and an image sky."""
import os

import numpy as np
import pytest

from raytracing_amd import standin
from raytracing_amd.raytracer import Context, Raytracer

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = Context(0)
    yield c
    c.close()


def _render(ctx, sc, max_depth, env):
    keys = ("TRR_SCHEDULE", "TRR_DRAIN_LEVEL", "TRR_SHADOW", "TRR_RESOLVE", "TRR_BATCHES", "TRR_SPLIT", "TRR_STAGE", "TRR_SPLIT_DRAIN")
    old = {k: os.environ.get(k) for k in keys}
    for k in keys:
        os.environ.pop(k, None)
    os.environ.update(env)
    try:
        rt = Raytracer(sc, max_depth=max_depth, context=ctx)
        try:
            out = rt.render_frame()
            return out["rgb"], out["stats"]
        finally:
            rt.close()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


VARIANTS = {
    "persistent": {"TRR_SCHEDULE": "persistent"},
    "levels, split (shadow rays on streams of their own)": {"TRR_SCHEDULE": "levels", "TRR_SPLIT": "1"},
    "levels, split, wide shadow records": {"TRR_SCHEDULE": "levels", "TRR_SPLIT": "1", "TRR_SHADOW": "wide"},
    "levels, split, three batches": {"TRR_SCHEDULE": "levels", "TRR_SPLIT": "1", "TRR_BATCHES": "3"},
    "levels, no split, no drain": {"TRR_SCHEDULE": "levels", "TRR_SPLIT": "0", "TRR_DRAIN_LEVEL": "-1"},
    "levels, drain from level 0": {"TRR_SCHEDULE": "levels", "TRR_DRAIN_LEVEL": "0", "TRR_SPLIT": "0"},
    "levels, drain from level 1": {"TRR_SCHEDULE": "levels", "TRR_DRAIN_LEVEL": "1", "TRR_SPLIT": "0"},
    "levels, drain from level 3": {"TRR_SCHEDULE": "levels", "TRR_DRAIN_LEVEL": "3", "TRR_SPLIT": "0"},
    "levels, wide shadow records": {"TRR_SCHEDULE": "levels", "TRR_SHADOW": "wide", "TRR_SPLIT": "0"},
    "levels, wide shadow records, drain from level 0": {"TRR_SCHEDULE": "levels", "TRR_SHADOW": "wide", "TRR_DRAIN_LEVEL": "0", "TRR_SPLIT": "0"},
    "levels, two resolve passes": {"TRR_SCHEDULE": "levels", "TRR_RESOLVE": "two_pass"},
    "levels, five batches": {"TRR_SCHEDULE": "levels", "TRR_BATCHES": "5"},
    "levels, split, deep levels drained by one launch": {"TRR_SCHEDULE": "levels", "TRR_SPLIT": "1", "TRR_SPLIT_DRAIN": "1"},
    "levels, split, drained, wide shadow records, three batches": {"TRR_SCHEDULE": "levels", "TRR_SPLIT": "1", "TRR_SPLIT_DRAIN": "1", "TRR_SHADOW": "wide", "TRR_BATCHES": "3"},
    "levels, staged ray starts": {"TRR_SCHEDULE": "levels", "TRR_STAGE": "1", "TRR_SPLIT": "0", "TRR_DRAIN_LEVEL": "-1"},
    "levels, staged ray starts, split, wide shadow records": {"TRR_SCHEDULE": "levels", "TRR_STAGE": "1", "TRR_SPLIT": "1", "TRR_SHADOW": "wide"},
}


@pytest.mark.parametrize("spp", [1, 4, 5])
def test_every_schedule_and_layout_renders_the_same_bytes(ctx, spp):
    sc = standin.parity_scene(320, 240, detail=14)
    sc.settings.anti_aliasing, sc.settings.samples = spp > 1, spp
    want, st0 = _render(ctx, sc, 5, {"TRR_SCHEDULE": "levels", "TRR_SPLIT": "0"})
    assert want.any() and st0.rays_reflection > 0 and st0.rays_shadow > 0
    for name, env in VARIANTS.items():
        got, st = _render(ctx, sc, 5, env)
        assert np.array_equal(got, want), f"{name}, {spp} spp"
        assert st.rays() == st0.rays() and st.queue_overflows == 0, f"{name}, {spp} spp"


@pytest.mark.parametrize("n_lights", [0, 1, 129])
def test_light_counts_at_the_edges_of_the_shadow_entry_format(ctx, oracle_mod, n_lights):
    """No light at all (no shadow rays), one, and 129 -- one more than the 16-byte compact shadow entries can name, so the
    library falls back to 32-byte ray records by itself -- against the CPU oracle, levels and persistent schedules."""
    from raytracing_amd.scene import Light
    sc = standin.parity_scene(120, 80, detail=8)
    rng = np.random.default_rng(n_lights)
    sc.lights = [Light.from_ke((float(rng.uniform(-6, 6)), float(rng.uniform(1, 7)), float(rng.uniform(-6, 6))),
                               (float(rng.uniform(0.2, 1)), float(rng.uniform(0.2, 1)), float(rng.uniform(0.2, 1))), float(rng.uniform(0.2, 2.0)))
                 for _ in range(n_lights)]
    osc = oracle_mod.OracleScene(sc)
    ref = osc.render(shade=oracle_mod.SHADE_HOISTED, max_depth=3, want_ids=False, threads=8)
    osc.close()
    for env in ({"TRR_SCHEDULE": "levels", "TRR_SPLIT": "0"}, {"TRR_SCHEDULE": "levels", "TRR_SPLIT": "1"},
                {"TRR_SCHEDULE": "levels", "TRR_SPLIT": "0", "TRR_DRAIN_LEVEL": "0"}, {"TRR_SCHEDULE": "persistent"}):
        got, st = _render(ctx, sc, 3, env)
        d = np.abs(got.astype(np.int16) - ref["rgb"].astype(np.int16)).max(axis=-1)
        assert (d <= 1).mean() >= 0.999, f"{n_lights} lights, {env}: exact on {(d == 0).mean():.6f}"
        assert (st.rays_shadow == 0) == (n_lights == 0) and st.queue_overflows == 0
