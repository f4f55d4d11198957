"""GPU tests of the device BVH build (TRR_BUILD_DEVICE of include/trayracer_hip.h).

The reference builds each object's Bvh on the CPU (object.rs:156); its closest hit is a min over every triangle whose
own box the ray passes (object.rs:191-206), so the image does not depend on the tree.  The tests therefore demand that
a scene uploaded with the GPU builder renders BYTE-IDENTICAL pixels and ids to the same scene uploaded with the host
builder (which the parity tests pin against the oracle), that it matches the oracle directly, and that the tree passes
trr_scene_validate (every live triangle in one leaf, boxes nested, references in range).
"""
import numpy as np
import pytest

from raytracing_amd import _native as N
from raytracing_amd import standin
from raytracing_amd.raytracer import Raytracer
from raytracing_amd.scene import Light, Object, Scene, Settings
from tests._util import assert_parity
from tests.test_oracle_kat import front_camera, mat, tri_rows

pytestmark = pytest.mark.gpu
F32 = np.float32


@pytest.fixture(scope="module")
def ctx():
    from raytracing_amd.raytracer import Context
    c = Context(0)
    yield c
    c.close()


def render_both(ctx, scene, max_depth=5, **kw):
    out = {}
    for build in ("host", "device"):
        rt = Raytracer(scene, max_depth=max_depth, context=ctx, build=build)
        try:
            rt.validate()
            out[build] = (rt.render_frame(want_ids=True, **kw), rt.scene_info())
        finally:
            rt.close()
    return out


def assert_same(out, what):
    (h, hi), (d, di) = out["host"], out["device"]
    assert hi["build"] == N.TRR_BUILD_HOST and di["build"] == N.TRR_BUILD_DEVICE
    assert hi["n_triangles"] == di["n_triangles"] and hi["n_triangles_live"] == di["n_triangles_live"], what
    assert np.array_equal(h["hit_obj"], d["hit_obj"]), what + ": object ids differ between the builders"
    assert np.array_equal(h["hit_tri"], d["hit_tri"]), what + ": triangle ids differ between the builders"
    assert np.array_equal(h["rgb"], d["rgb"]), what + ": pixels differ between the builders"


@pytest.mark.parametrize("sky_image", [True, False])
def test_parity_scene_is_identical_with_both_builders(ctx, oracle_mod, sky_image):
    sc = standin.parity_scene(200, 150, detail=14, sky_image=sky_image)
    out = render_both(ctx, sc, max_depth=5)
    assert_same(out, "parity scene")
    ref = oracle_mod.OracleScene(sc).render(shade=oracle_mod.SHADE_HOISTED, max_depth=5)
    assert_parity(out["device"][0], ref, what="device-built parity scene against the oracle")


def test_standin_scenes_are_identical_with_both_builders(ctx):
    for name, sc, md in [("test scene", standin.standin_test(400, 300, n_car=120_000, n_floor=30_000), 2),
                         ("mirrors", standin.standin_mirrors(240, 160, n_spheres=9, tris_per_sphere=600, sky=(256, 128)), 9),
                         ("terrain", standin.standin_terrain(240, 160, nx=200, nz=100), 5)]:
        out = render_both(ctx, sc, max_depth=md)
        assert_same(out, name)
        info = out["device"][1]
        assert info["n_nodes"] > 0 and 1 <= info["bvh_depth"] < 64, info


def test_small_dead_and_duplicate_triangles(ctx, oracle_mod):
    """Root-leaf objects (1..4 triangles), an empty object, triangles the reference can never hit (left out of the tree),
    NaN vertices, and many identical triangles (equal Morton keys: the radix tree falls back to the positions)."""
    z = -3.0
    one = tri_rows(((-1, -1, z), (1, -1, z), (0, 1, z)))
    three = tri_rows(((-2, -1, z - 1), (0, -1, z - 1), (-1, 1, z - 1)), ((0, -1, z - 1), (2, -1, z - 1), (1, 1, z - 1)),
                     ((-1, 1, z - 1), (1, 1, z - 1), (0, 2, z - 1)))
    dup = np.repeat(tri_rows(((-0.5, -0.5, z + 1), (0.5, -0.5, z + 1), (0, 0.5, z + 1))), 37, axis=0)
    rows = [((-1, -1, z - 2), (1, -1, z - 2), (1, 1, z - 2)),
            ((0, 0, z), (0, 0, z), (0, 0, z)),                               # zero area: never in the tree
            ((float("nan"), 0, z), (1, 0, z), (0, 1, z)),                    # NaN vertex
            ((-1, -1, z - 2), (1, 1, z - 2), (-1, 1, z - 2))]
    for k in range(9):                                                       # a strip of thin triangles on a line
        x = -1.0 + 0.2 * k
        rows.append(((x, -1.5, z - 0.5), (x + 0.2, -1.5, z - 0.5), (x + 0.1, -1.2, z - 0.5)))
    mixed = tri_rows(*rows)
    m = [mat(kd=(0.4, 0.5, 0.6), illum=2, ks=(1, 1, 1), ns=5.0, dissolve=0.0)]
    objs = [Object(triangles=one, material_index=np.array([-1], np.int32)), Object(),
            Object(triangles=three, material_index=np.zeros(3, np.int32), materials=m),
            Object(triangles=dup, material_index=np.zeros(len(dup), np.int32), materials=m),
            Object(triangles=mixed, material_index=np.zeros(len(mixed), np.int32), materials=m)]
    sc = Scene(objects=objs, lights=[Light(np.array([0.5, 2.0, 0], F32), np.array([1, 1, 1], F32), 9.0)],
               camera=front_camera(96, 72), settings=Settings())
    out = render_both(ctx, sc)
    assert_same(out, "small / dead / duplicate triangles")
    assert out["device"][1]["n_triangles_live"] == out["device"][1]["n_triangles"] - 1
    ref = oracle_mod.OracleScene(sc).render()
    assert_parity(out["device"][0], ref, what="device build, edge cases")
    assert sorted(set(np.unique(out["device"][0]["hit_obj"]).tolist()) - {0xFFFFFFFF}) == [0, 2, 3, 4]


@pytest.mark.parametrize("knobs", [{"TRR_TREELET_MAX": "0"}, {"TRR_TREELET_MAX": "64", "TRR_TREELET_SMALL": "64"},
                                   {"TRR_TREELET_MAX": "100000", "TRR_TREELET_SMALL": "3"}, {"TRR_TREELET_MAX": "300", "TRR_TREELET_SMALL": "16"}])
def test_treelet_settings_all_give_a_valid_tree_and_the_same_image(ctx, monkeypatch, knobs):
    """The device build rebuilds small subtrees of its radix tree with SAH splits (lbvh_core.h): with the pass off, with
    threads only (every treelet at most `small`), with wavefronts doing nearly everything (one treelet for a whole object,
    ranges down to 3), and in between -- the tree validates and the image is the host-built tree's."""
    scenes = [standin.standin_test(320, 240, n_car=60_000, n_floor=9_000), standin.standin_terrain(240, 160, nx=150, nz=90)]
    for sc in scenes:
        rt = Raytracer(sc, max_depth=3, context=ctx, build="host")
        try:
            want = rt.render_frame(want_ids=True)
        finally:
            rt.close()
        for k, v in knobs.items():
            monkeypatch.setenv(k, v)
        rt = Raytracer(sc, max_depth=3, context=ctx, build="device")
        try:
            rt.validate()
            got = rt.render_frame(want_ids=True)
        finally:
            rt.close()
        for k in knobs:
            monkeypatch.delenv(k)
        assert np.array_equal(want["hit_tri"], got["hit_tri"]) and np.array_equal(want["rgb"], got["rgb"]), knobs


def test_treelets_of_identical_and_collinear_triangles(ctx):
    """What the SAH splits of the treelet pass cannot separate: thousands of identical triangles (no centroid extent on any
    axis: the ranges are halved by count), centroids on one line, and a big triangle over many small ones (one bin takes
    nearly everything) -- each well above the wavefront stage's range size, in one object and in objects of their own."""
    rng = np.random.default_rng(5)
    z = -4.0
    same = np.repeat(tri_rows(((-1.5, -1.0, z), (-0.5, -1.0, z), (-1.0, 0.0, z))), 5000, axis=0)
    xs = np.linspace(-1.9, 1.9, 6000)
    line = np.concatenate([tri_rows(((x, 0.6, z + 0.5), (x + 0.0005, 0.6, z + 0.5), (x + 0.00025, 0.9, z + 0.5))) for x in xs])
    small = []
    for _ in range(4000):
        cx, cy = rng.uniform(0.2, 1.8), rng.uniform(-1.2, 0.2)
        small.append(((cx, cy, z + 1.0), (cx + 0.01, cy, z + 1.0), (cx, cy + 0.01, z + 1.0)))
    big = tri_rows(((0.0, -1.5, z + 0.2), (2.0, -1.5, z + 0.2), (1.0, 0.5, z + 0.2)), *small)
    m = [mat(kd=(0.7, 0.5, 0.3), illum=2, ks=(1, 1, 1), ns=9.0, dissolve=0.0)]

    def obj(t):
        return Object(triangles=t, material_index=np.zeros(len(t), np.int32), materials=m)
    everything = np.concatenate([same, line, big])
    for name, objs in [("one object", [obj(everything)]), ("three objects", [obj(same), obj(line), obj(big)])]:
        sc = Scene(objects=objs, lights=[Light(np.array([0.5, 2.0, 0], F32), np.array([1, 1, 1], F32), 9.0)],
                   camera=front_camera(160, 120), settings=Settings())
        out = render_both(ctx, sc, max_depth=2)
        assert_same(out, name)
        assert out["device"][1]["bvh_depth"] < 64, out["device"][1]


def test_million_triangle_build_validates_and_renders_the_same(ctx):
    """The benchmark's mesh (BASELINE.json configs[1], 1.03 M triangles): device-built tree checked on the host, and a
    quarter-size frame identical to the host-built one."""
    sc = standin.standin_car(960, 540, n_tris=1_027_464)
    out = render_both(ctx, sc, max_depth=5)
    assert_same(out, "standin_car 1 M")
    hi, di = out["host"][1], out["device"][1]
    print(f"\nbuild ms: host {hi['ms_build']:.1f} (depth {hi['bvh_depth']}, {hi['n_nodes']} nodes)  "
          f"device {di['ms_build']:.1f} (depth {di['bvh_depth']}, {di['n_nodes']} nodes)")


def test_stack_need_is_reported_and_bounded(ctx, oracle_mod):
    """A one-sided tree: triangle centres on a geometric series along the diagonal (every Morton bit splits off one
    triangle) plus a cluster of identical triangles at its small end.  Either the upload reports a traversal-stack need
    within the kernel's 64 entries and renders what the oracle renders, or it refuses the tree -- never a silent overflow."""
    rows = []
    for k in range(40):
        c = 4.0 * 0.5 ** k
        rows.append(((c - 1.5, c - 1.5 - 0.3, -4 - c), (c - 1.5 + 0.3, c - 1.5 - 0.3, -4 - c), (c - 1.5, c - 1.5 + 0.3, -4 - c)))
    rows += [((-1.5, -1.8, -4.0), (-1.2, -1.8, -4.0), (-1.5, -1.2, -4.0))] * 150
    tri = tri_rows(*rows)
    obj = Object(triangles=tri, material_index=np.zeros(len(tri), np.int32), materials=[mat(kd=(0.7, 0.6, 0.5), illum=2, ks=(1, 1, 1), ns=8.0, dissolve=0.0)])
    sc = Scene(objects=[obj], lights=[Light(np.array([0.0, 3.0, 0.0], F32), np.array([1, 1, 1], F32), 20.0)],
               camera=front_camera(96, 72), settings=Settings())
    for build in ("host", "device"):
        try:
            rt = Raytracer(sc, context=ctx, build=build)
        except N.NativeError as e:
            assert "traversal stack" in str(e)
            continue
        try:
            info = rt.scene_info()
            assert 1 <= info["stack_need"] <= 64 and info["stack_need"] <= 3 * info["bvh_depth"], info
            rt.validate()
            out = rt.render_frame(want_ids=True)
        finally:
            rt.close()
        assert_parity(out, oracle_mod.OracleScene(sc).render(), what=f"one-sided tree, {build} build")
        print(f"\n{build}: depth {info['bvh_depth']}, stack need {info['stack_need']}")


def test_unknown_build_mode_is_rejected(ctx):
    import ctypes as C
    h = C.c_void_p()
    rc = N.lib().trr_scene_create_ex(ctx.handle, None, 0, None, 0, 7, C.byref(h))
    assert rc == N.TRR_ERR_INVALID and not h
