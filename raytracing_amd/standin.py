"""Seeded procedural stand-in scenes.

Every mesh/skybox asset that BASELINE.json's configs name (res/car_objs/car.obj, res/test/*.obj,
res/testSzene.zip, res/skyboxes/*.exr) is a git-LFS pointer stub in the reference snapshot
(SURVEY.md F1), so the benchmark and parity scenes are generated here.  Camera and lights are taken
verbatim from the one real asset, res/test/config.yaml:28-80.  Results on these scenes are labelled
"stand-in", never "car_objs".

All geometry is computed in float64 from integer-hash noise and rounded once to float32, so the arrays
are reproducible across machines (no float32 SIMD transcendental paths are involved).
"""
from __future__ import annotations

import numpy as np

from .scene import Camera, IlluminationModel, Light, Material, Object, Scene, Settings, Skybox

F32 = np.float32


# --------------------------------------------------------------------------------------------
# res/test/config.yaml:28-80 (verbatim values)
# --------------------------------------------------------------------------------------------
def config_yaml_lights():
    raw = [
        ((-2.0, 3.5, -0.8), (0.5773503, 0.5773503, 0.5773503), 40.0),
        ((5.5, 0.3, -4.5), (0.672333, 0.24602187, 0.6981702), 22.0),
        ((2.0, 4.0, 1.6), (0.57888657, 0.5770311, 0.57612985), 23.0),
        ((-0.7, 1.0, 5.2), (1.0, 1.0, 1.0), 6.0),
    ]
    return [Light.from_ke(p, ke, i) for p, ke, i in raw]


def config_yaml_camera(width=1920, height=1080) -> Camera:
    return Camera(
        position=np.array([3.7080245, 2.1147687, 7.0919194], F32),
        look_at=np.array([3.3485944, 1.793123, 6.215932], F32),
        up=np.array([0.0, 1.0, 0.0], F32),
        fov=float(np.float32(np.float32(63.02536) * np.float32(np.pi / 180.0))),   # f32::to_radians
        resolution=(int(width), int(height)),
    )


CONFIG_YAML_TRANSLATION = (0.7, -0.1, -0.5)   # res/test/config.yaml:3-6, 16-19


# --------------------------------------------------------------------------------------------
# deterministic value noise (integer hash -> [0,1))
# --------------------------------------------------------------------------------------------
def _hash01(ix, iy, seed):
    h = (ix.astype(np.uint64) * np.uint64(0x9E3779B185EBCA87)) ^ (iy.astype(np.uint64) * np.uint64(0xC2B2AE3D27D4EB4F))
    h ^= np.uint64(seed) * np.uint64(0x165667B19E3779F9)
    h ^= h >> np.uint64(29)
    h *= np.uint64(0xBF58476D1CE4E5B9)
    h ^= h >> np.uint64(32)
    return (h >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def value_noise(x, y, seed, octaves=4):
    """Smooth 2-D value noise in [0,1), float64 in/out."""
    out = np.zeros_like(x, dtype=np.float64)
    amp, tot = 1.0, 0.0
    with np.errstate(over="ignore"):
        for o in range(octaves):
            fx, fy = np.floor(x), np.floor(y)
            tx, ty = x - fx, y - fy
            sx, sy = tx * tx * (3 - 2 * tx), ty * ty * (3 - 2 * ty)
            ix, iy = fx.astype(np.int64), fy.astype(np.int64)
            a = _hash01(ix, iy, seed + o)
            b = _hash01(ix + 1, iy, seed + o)
            c = _hash01(ix, iy + 1, seed + o)
            d = _hash01(ix + 1, iy + 1, seed + o)
            out += amp * ((a * (1 - sx) + b * sx) * (1 - sy) + (c * (1 - sx) + d * sx) * sy)
            tot += amp
            amp *= 0.5
            x, y = x * 2.0, y * 2.0
    return out / tot


# --------------------------------------------------------------------------------------------
# grid -> Triangle rows (a, b, c, na, nb, nc, uva, uvb, uvc)
# --------------------------------------------------------------------------------------------
def grid_to_triangles(P, N, UV, flip=False) -> np.ndarray:
    """P, N: [nu+1, nv+1, 3]; UV: [nu+1, nv+1, 2] (float64) -> float32 [2*nu*nv, 24].
    Quad (i,j) gives triangles (00,10,11) and (00,11,01), emitted quad by quad."""
    p00, p10, p11, p01 = P[:-1, :-1], P[1:, :-1], P[1:, 1:], P[:-1, 1:]
    n00, n10, n11, n01 = N[:-1, :-1], N[1:, :-1], N[1:, 1:], N[:-1, 1:]
    t00, t10, t11, t01 = UV[:-1, :-1], UV[1:, :-1], UV[1:, 1:], UV[:-1, 1:]
    if flip:
        t1 = np.concatenate([p00, p11, p10, n00, n11, n10, t00, t11, t10], axis=-1)
        t2 = np.concatenate([p00, p01, p11, n00, n01, n11, t00, t01, t11], axis=-1)
    else:
        t1 = np.concatenate([p00, p10, p11, n00, n10, n11, t00, t10, t11], axis=-1)
        t2 = np.concatenate([p00, p11, p01, n00, n11, n01, t00, t11, t01], axis=-1)
    tri = np.stack([t1, t2], axis=2)   # [nu, nv, 2, 24]
    return np.ascontiguousarray(tri.reshape(-1, 24).astype(F32))


def _grid_normals(P):
    du = np.gradient(P, axis=0)
    dv = np.gradient(P, axis=1)
    n = np.cross(du, dv)
    ln = np.linalg.norm(n, axis=-1, keepdims=True)
    return np.where(ln > 1e-20, n / np.maximum(ln, 1e-20), np.array([0.0, 1.0, 0.0]))


def _spow(c, e):
    return np.sign(c) * np.abs(c) ** e


def superellipsoid(nu, nv, radii, e1=1.0, e2=1.0, center=(0, 0, 0), bump=0.0, seed=0, bump_freq=6.0):
    """Closed shell: u = longitude [0,2pi], v = latitude [-pi/2,pi/2].  Poles give zero-area triangles, which the
    reference's Triangle::intersect culls (triangle.rs:56) -- they stay in on purpose."""
    u = np.linspace(0.0, 2.0 * np.pi, nu + 1)
    v = np.linspace(-0.5 * np.pi, 0.5 * np.pi, nv + 1)
    U, V = np.meshgrid(u, v, indexing="ij")
    cu, su, cv, sv = np.cos(U), np.sin(U), np.cos(V), np.sin(V)
    x = _spow(cv, e1) * _spow(cu, e2)
    z = _spow(cv, e1) * _spow(su, e2)
    y = _spow(sv, e1)
    r = 1.0
    if bump > 0.0:
        # periodic in u so the seam closes
        nz = value_noise((np.cos(U) + 1.5) * bump_freq, (np.sin(U) + 1.5) * bump_freq + (V + 2.0) * bump_freq, seed)
        r = 1.0 + bump * (nz - 0.5)
    P = np.stack([x * r * radii[0] + center[0], y * r * radii[1] + center[1], z * r * radii[2] + center[2]], axis=-1)
    P[0] = P[-1]   # close the seam exactly
    N = -_grid_normals(P)
    UV = np.stack([U / (2.0 * np.pi), (V + 0.5 * np.pi) / np.pi], axis=-1)
    return grid_to_triangles(P, N, UV)


def torus(nu, nv, R, r, center=(0, 0, 0), axis=0):
    u = np.linspace(0.0, 2.0 * np.pi, nu + 1)
    v = np.linspace(0.0, 2.0 * np.pi, nv + 1)
    U, V = np.meshgrid(u, v, indexing="ij")
    a = (R + r * np.cos(V)) * np.cos(U)
    b = (R + r * np.cos(V)) * np.sin(U)
    c = r * np.sin(V)
    na, nb, nc = np.cos(V) * np.cos(U), np.cos(V) * np.sin(U), np.sin(V)
    if axis == 0:     # wheel axle along x
        P = np.stack([c, a, b], axis=-1)
        N = np.stack([nc, na, nb], axis=-1)
    elif axis == 1:
        P = np.stack([a, c, b], axis=-1)
        N = np.stack([na, nc, nb], axis=-1)
    else:
        P = np.stack([a, b, c], axis=-1)
        N = np.stack([na, nb, nc], axis=-1)
    P = P + np.asarray(center, np.float64)
    UV = np.stack([U / (2.0 * np.pi), V / (2.0 * np.pi)], axis=-1)
    return grid_to_triangles(P, N, UV)


def disc(nr, nt, radius, y=0.0, center=(0.0, 0.0)):
    """Subdivided disc floor in the y = const plane, normal +Y."""
    rr = np.linspace(0.0, radius, nr + 1)
    tt = np.linspace(0.0, 2.0 * np.pi, nt + 1)
    Rg, Tg = np.meshgrid(rr, tt, indexing="ij")
    P = np.stack([center[0] + Rg * np.cos(Tg), np.full_like(Rg, y), center[1] + Rg * np.sin(Tg)], axis=-1)
    P[:, -1] = P[:, 0]
    N = np.zeros_like(P)
    N[..., 1] = 1.0
    UV = np.stack([0.5 + 0.5 * Rg / radius * np.cos(Tg), 0.5 + 0.5 * Rg / radius * np.sin(Tg)], axis=-1)
    return grid_to_triangles(P, N, UV, flip=True)


def heightfield(nx, nz, size_x, size_z, height, seed, freq=8.0):
    x = np.linspace(-0.5 * size_x, 0.5 * size_x, nx + 1)
    z = np.linspace(-0.5 * size_z, 0.5 * size_z, nz + 1)
    X, Z = np.meshgrid(x, z, indexing="ij")
    Y = height * (value_noise((X / size_x + 0.5) * freq, (Z / size_z + 0.5) * freq, seed, octaves=6) - 0.5)
    P = np.stack([X, Y, Z], axis=-1)
    N = -_grid_normals(P)
    UV = np.stack([X / size_x + 0.5, Z / size_z + 0.5], axis=-1)
    return grid_to_triangles(P, N, UV, flip=True)


def checker_texture(w, h, cells=8, c0=(230, 60, 40), c1=(240, 240, 230)) -> np.ndarray:
    yy, xx = np.mgrid[0:h, 0:w]
    m = (((xx * cells) // w + (yy * cells) // h) & 1).astype(bool)
    img = np.where(m[..., None], np.array(c0, np.uint8), np.array(c1, np.uint8)).astype(np.uint8)
    return np.ascontiguousarray(img)


def procedural_sky(w=4096, h=2048, seed=3) -> np.ndarray:
    """Gradient + sun + horizon checker, RGB8 equirect (stand-in for res/skyboxes/*.exr after into_rgb8())."""
    yy, xx = np.mgrid[0:h, 0:w]
    t = yy.astype(np.float64) / (h - 1)
    top = np.array([60, 110, 220], np.float64)
    hor = np.array([235, 225, 200], np.float64)
    bot = np.array([70, 60, 50], np.float64)
    up = np.clip(1.0 - 2.0 * t, 0, 1)[..., None]
    dn = np.clip(2.0 * t - 1.0, 0, 1)[..., None]
    img = hor * (1 - up) * (1 - dn) + top * up + bot * dn
    sx, sy = int(0.3 * w), int(0.28 * h)
    d2 = ((xx - sx) / (0.02 * w)) ** 2 + ((yy - sy) / (0.04 * h)) ** 2
    img += 255.0 * np.exp(-d2)[..., None]
    chk = ((((xx * 64) // w) + ((yy * 32) // h)) & 1).astype(np.float64)[..., None]
    img *= 0.85 + 0.15 * chk
    img += 12.0 * (value_noise(xx / 37.0, yy / 37.0, seed, octaves=3)[..., None] - 0.5)
    return np.ascontiguousarray(np.clip(np.round(img), 0, 255).astype(np.uint8))


# --------------------------------------------------------------------------------------------
# materials
# --------------------------------------------------------------------------------------------
def _mat(name, kd=None, ks=None, ns=None, illum=2, d=None, tex=None):
    return Material(
        name=name,
        diffuse_color=None if kd is None else np.array(kd, F32),
        specular_color=None if ks is None else np.array(ks, F32),
        specular_exponent=ns,
        diffuse_texture=tex,
        illumination_model=IlluminationModel(illum),
        dissolve=None if d is None else float(np.float32(1.0) - np.float32(d)),   # object.rs:106
    )


def _scale_counts(target, parts):
    """parts: list of (weight, aspect nu/nv) -> list of (nu, nv) with sum 2*nu*nv ~= target."""
    wsum = float(sum(p[0] for p in parts))
    out = []
    for wgt, aspect in parts:
        quads = max(8.0, 0.5 * target * wgt / wsum)
        nv = max(2, int(round((quads / aspect) ** 0.5)))
        nu = max(3, int(round(quads / nv)))
        out.append((nu, nv))
    return out


TABLE_CENTER = (0.65, 1.84)   # object-space (x, z) under the centre of the config.yaml view
TABLE_SCALE = 0.8


def _place(s, d):
    """Camera-aligned disc coordinates (s = screen-right, d = away from the camera, both through the disc centre)
    -> object-space (x, z) for the res/test/config.yaml camera."""
    return (0.925 * s - 0.379 * d, -0.379 * s - 0.925 * d)


def car_object(n_tris=1_000_000, seed=1) -> Object:
    """Stand-in for res/car_objs/car.obj (== res/test/car.obj, 87 MB): like the real file (see
    res/test/render_example.png) ONE object holding a whole table-top arrangement -- a car shell with wheels, a
    bin, a bust-like blob, a mirror sphere, a ring knot, an apple and a pencil -- ~n_tris triangles, mixed
    illum 2/3/4, two materials without `d` (tinted shadows), two textured materials."""
    w_car, w_bust, w_bin, w_sph, w_knot, w_apple, w_pen = 0.40, 0.15, 0.08, 0.08, 0.12, 0.12, 0.05
    ((bu, bv), (cu, cv), (wu, wv), (hu, hv), (su, sv), (nu_, nv_), (pu, pv), (ku, kv), (au, av), (eu, ev)) = _scale_counts(
        n_tris, [(w_car * 0.55, 2.0), (w_car * 0.20, 2.0), (w_car * 0.20, 4.0), (w_car * 0.05, 2.0),
                 (w_bust, 2.0), (w_bin, 3.0), (w_sph, 2.0), (w_knot, 5.0), (w_apple, 2.0), (w_pen, 0.25)])
    wu = max(3, wu // 2)   # four wheels share their budget
    wv = max(3, wv // 2)
    parts, mats_idx = [], []

    def add(tri, mat):
        parts.append(tri)
        mats_idx.append(np.full(len(tri), mat) if np.isscalar(mat) else mat)

    cx, cz = _place(-3.4, -1.2)
    body = superellipsoid(bu, bv, (1.05, 0.42, 2.3), e1=0.55, e2=0.7, center=(cx, 0.62, cz), bump=0.05, seed=seed)
    add(body, np.where(np.arange(len(body)) % 7 == 0, 1, 0))
    add(superellipsoid(cu, cv, (0.82, 0.38, 1.15), e1=0.8, e2=0.8, center=(cx, 1.12, cz - 0.25), bump=0.02, seed=seed + 7), 2)
    for sx, sz in [(-1.0, 1.45), (1.0, 1.45), (-1.0, -1.45), (1.0, -1.45)]:
        add(torus(wu, wv, 0.33, 0.13, center=(cx + sx * 0.98, 0.46, cz + sz), axis=0), 3)
    for sx in (-0.55, 0.55):
        add(superellipsoid(max(3, hu // 2), max(2, hv // 2), (0.22, 0.13, 0.08), center=(cx + sx, 0.68, cz + 2.28)), 4)
    x, z = _place(0.0, 2.6)     # bust: bumpy head on a narrower neck
    add(superellipsoid(su, sv, (0.55, 0.75, 0.6), e1=0.9, e2=0.9, center=(x, 1.75, z), bump=0.22, seed=seed + 11, bump_freq=9.0), 5)
    add(superellipsoid(max(3, su // 2), max(2, sv // 2), (0.62, 0.5, 0.5), e1=0.6, e2=0.9, center=(x, 0.5, z), bump=0.1, seed=seed + 12), 5)
    x, z = _place(-1.7, 2.4)     # bin: squarish barrel, rust texture
    add(superellipsoid(nu_, nv_, (0.6, 1.15, 0.6), e1=0.25, e2=0.9, center=(x, 1.15, z), bump=0.03, seed=seed + 13), 6)
    x, z = _place(2.0, 2.9)      # mirror sphere
    add(superellipsoid(pu, pv, (1.0, 1.0, 1.0), center=(x, 1.0, z)), 7)
    x, z = _place(4.3, 1.8)      # knot: two interlocked rings
    add(torus(ku, max(3, kv // 2), 0.75, 0.24, center=(x, 0.55, z), axis=1), 8)
    add(torus(ku, max(3, kv // 2), 0.75, 0.24, center=(x + 0.55, 0.85, z), axis=2), 8)
    x, z = _place(3.1, -2.2)     # apple
    add(superellipsoid(au, av, (1.15, 1.05, 1.15), e1=0.85, e2=1.0, center=(x, 1.05, z), bump=0.06, seed=seed + 17, bump_freq=3.0), 9)
    x, z = _place(0.3, -1.8)     # pencil along screen-right
    pen = superellipsoid(eu, ev, (1.9, 0.07, 0.07), e1=0.3, e2=1.0, center=(0.0, 0.0, 0.0))
    ang = np.arctan2(-0.379, 0.925) + 0.12   # direction of screen-right in the xz plane, slightly skewed
    ca, sa = np.cos(ang), np.sin(ang)
    pen64 = pen.astype(np.float64)
    for k in (0, 3, 6, 9, 12, 15):   # rotate positions and normals about +Y, then translate the positions
        px, pz = pen64[:, k].copy(), pen64[:, k + 2].copy()
        pen64[:, k] = ca * px - sa * pz
        pen64[:, k + 2] = sa * px + ca * pz
    for k in (0, 3, 6):
        pen64[:, k] += x
        pen64[:, k + 1] += 0.09
        pen64[:, k + 2] += z
    add(np.ascontiguousarray(pen64.astype(F32)), 10)
    tri = np.concatenate(parts, axis=0)
    # shrink the arrangement about the point where the config.yaml camera's axis meets the floor
    tri64 = tri.astype(np.float64)
    for k in (0, 3, 6):
        tri64[:, k] = TABLE_CENTER[0] + TABLE_SCALE * tri64[:, k]
        tri64[:, k + 1] = TABLE_SCALE * tri64[:, k + 1]
        tri64[:, k + 2] = TABLE_CENTER[1] + TABLE_SCALE * tri64[:, k + 2]
    tri = np.ascontiguousarray(tri64.astype(F32))
    mi = np.concatenate(mats_idx).astype(np.int32)
    rust = checker_texture(388, 284, 24, (120, 70, 40), (80, 45, 30))      # size of res/test/textures' 388x284 image
    skin = checker_texture(1024, 1024, 64, (235, 120, 40), (220, 60, 30))  # size of the two 1024^2 JPGs
    mats = [
        _mat("paint", kd=(0.03, 0.22, 0.06), ks=(0.5, 0.5, 0.5), ns=180.0, illum=3, d=1.0),
        _mat("paint_stripe", kd=(0.3, 0.28, 0.26), ks=(0.3, 0.3, 0.3), ns=60.0, illum=2, d=1.0),
        _mat("glass", kd=(0.03, 0.04, 0.05), ks=(1.0, 1.0, 1.0), ns=400.0, illum=4),          # no d -> dissolve None
        _mat("rubber", kd=(0.02, 0.02, 0.02), ks=(0.1, 0.1, 0.1), ns=12.0, illum=2, d=1.0),
        _mat("chrome", kd=(0.18, 0.18, 0.18), ks=(1.0, 1.0, 1.0), ns=800.0, illum=3),           # no d
        _mat("marble", kd=(0.75, 0.68, 0.6), ks=(0.2, 0.2, 0.2), ns=30.0, illum=2, d=1.0),
        _mat("rust", kd=(0.4, 0.25, 0.15), ks=(0.1, 0.1, 0.1), ns=8.0, illum=2, d=1.0, tex=rust),
        _mat("mirror", kd=(0.02, 0.02, 0.02), ks=(1.0, 1.0, 1.0), ns=900.0, illum=3, d=1.0),
        _mat("pink", kd=(0.8, 0.25, 0.55), ks=(0.4, 0.4, 0.4), ns=50.0, illum=2, d=1.0),
        _mat("apple", kd=(0.8, 0.3, 0.1), ks=(0.3, 0.3, 0.3), ns=70.0, illum=2, d=1.0, tex=skin),
        _mat("pencil", kd=(0.08, 0.06, 0.25), ks=(0.5, 0.5, 0.5), ns=100.0, illum=2, d=1.0),
    ]
    return Object(name="Car", triangles=tri, material_index=mi, materials=mats,
                  translation=np.array(CONFIG_YAML_TRANSLATION, F32))


def floor_object(n_tris=300_000) -> Object:
    nr = max(2, int(round((n_tris / 8.0) ** 0.5)))
    nt = max(3, int(round(n_tris / (2.0 * nr))))
    tri = disc(nr, nt, 4.8, y=0.0, center=TABLE_CENTER)
    mats = [_mat("floor", kd=(0.035, 0.045, 0.04), ks=(0.6, 0.6, 0.6), ns=90.0, illum=3, d=1.0)]
    return Object(name="Floor", triangles=tri, material_index=np.zeros(len(tri), np.int32), materials=mats,
                  translation=np.array(CONFIG_YAML_TRANSLATION, F32))


def standin_car(width=1920, height=1080, n_tris=1_000_000, seed=1, **settings) -> Scene:
    """Configs 2 / 3: car shell only, camera + lights from res/test/config.yaml, Settings::default()."""
    return Scene(objects=[car_object(n_tris, seed)], lights=config_yaml_lights(),
                 camera=config_yaml_camera(width, height), settings=Settings(**settings))


def standin_test(width=800, height=600, n_car=1_000_000, n_floor=300_000, seed=1, **settings) -> Scene:
    """Config 1: floor first, car second (model order of res/test/config.yaml:1-27)."""
    return Scene(objects=[floor_object(n_floor), car_object(n_car, seed)], lights=config_yaml_lights(),
                 camera=config_yaml_camera(width, height), settings=Settings(**settings))


def standin_terrain(width=3840, height=2160, nx=3163, nz=1581, seed=2) -> Scene:
    """Config 4: nx*nz*2 triangles (3163 x 1581 -> 10.0 M), single illum-2 material, one light, fly-over camera."""
    tri = heightfield(nx, nz, 20.0, 10.0, 1.6, seed)
    mats = [_mat("terrain", kd=(0.2, 0.25, 0.15), ks=(0.08, 0.08, 0.08), ns=20.0, illum=2, d=1.0)]
    obj = Object(name="Terrain", triangles=tri, material_index=np.zeros(len(tri), np.int32), materials=mats)
    cam = Camera(position=np.array([-9.0, 3.2, 5.5], F32), look_at=np.array([0.5, -0.4, -0.5], F32),
                 up=np.array([0, 1, 0], F32), fov=float(np.float32(np.deg2rad(55.0))), resolution=(width, height))
    lights = [Light.from_ke((4.0, 9.0, 3.0), (1.0, 0.95, 0.9), 260.0)]
    return Scene(objects=[obj], lights=lights, camera=cam, settings=Settings())


def standin_mirrors(width=3840, height=2160, n_spheres=64, tris_per_sphere=80_000, seed=3, sky=(4096, 2048)) -> Scene:
    """Config 5: mirror spheres (illum 3) over a mirror plane under a procedural RGB8 sky, 2 lights."""
    g = int(round(n_spheres ** 0.5))
    nv = max(2, int(round((tris_per_sphere / 4.0) ** 0.5)))
    nu = 2 * nv
    parts = []
    for k in range(n_spheres):
        i, j = k % g, k // g
        hx = float(_hash01(np.array([i]), np.array([j]), seed)[0])
        r = 0.35 + 0.2 * hx
        parts.append(superellipsoid(nu, nv, (r, r, r), center=((i - 0.5 * (g - 1)) * 1.3, r + 0.02 + 0.6 * hx, (j - 0.5 * (g - 1)) * 1.3)))
    tri = np.concatenate(parts, axis=0)
    spheres = Object(name="Spheres", triangles=tri, material_index=(np.arange(len(tri)) // (len(tri) // n_spheres) % 3).astype(np.int32),
                     materials=[_mat("mirror", kd=(0.12, 0.12, 0.13), ks=(1, 1, 1), ns=500.0, illum=3, d=1.0),
                                _mat("gold", kd=(0.25, 0.18, 0.05), ks=(1, 0.9, 0.6), ns=300.0, illum=3, d=1.0),
                                _mat("blue", kd=(0.05, 0.08, 0.25), ks=(1, 1, 1), ns=200.0, illum=4, d=1.0)])
    ftri = disc(64, 256, 14.0, y=0.0)
    floor = Object(name="Plane", triangles=ftri, material_index=np.zeros(len(ftri), np.int32),
                   materials=[_mat("plane", kd=(0.05, 0.05, 0.055), ks=(0.8, 0.8, 0.8), ns=100.0, illum=3, d=1.0)])
    cam = Camera(position=np.array([6.5, 3.4, 8.0], F32), look_at=np.array([0.0, 0.5, 0.0], F32),
                 up=np.array([0, 1, 0], F32), fov=float(np.float32(np.deg2rad(50.0))), resolution=(width, height))
    lights = [Light.from_ke((-3.0, 7.0, 4.0), (1, 1, 1), 60.0), Light.from_ke((6.0, 5.0, -5.0), (0.9, 0.8, 1.0), 45.0)]
    st = Settings(skybox=Skybox(image=procedural_sky(sky[0], sky[1], seed)))
    return Scene(objects=[floor, spheres], lights=lights, camera=cam, settings=st)


def parity_scene(width=160, height=120, detail=24, seed=5, sky_image=True) -> Scene:
    """Small scene that exercises every branch of the path: three objects with translation / rotation /
    non-uniform scale, textured, untextured, material-less, no-`d`, transparent (illum 6) and mirror
    materials, image skybox, four lights.  ~10*detail^2 triangles."""
    d = max(3, int(detail))
    floor_tri = disc(d, 4 * d, 7.0, y=0.0)
    floor = Object(name="Floor", triangles=floor_tri, material_index=np.zeros(len(floor_tri), np.int32),
                   materials=[_mat("floor", kd=(0.035, 0.045, 0.04), ks=(0.6, 0.6, 0.6), ns=90.0, illum=3, d=1.0)],
                   translation=np.array(CONFIG_YAML_TRANSLATION, F32))
    blob = superellipsoid(2 * d, d, (1.0, 0.55, 2.0), e1=0.6, e2=0.75, center=(0, 0.75, 0), bump=0.08, seed=seed)
    ball = superellipsoid(2 * d, d, (0.45, 0.45, 0.45), center=(-1.7, 0.5, 1.3))
    glass = superellipsoid(2 * d, d, (0.5, 0.7, 0.08), e1=0.3, e2=0.3, center=(1.9, 0.75, 2.2))
    ring = torus(2 * d, d, 0.5, 0.16, center=(1.6, 0.7, -1.2), axis=1)
    tri = np.concatenate([blob, ball, glass, ring], axis=0)
    mi = np.concatenate([
        np.where(np.arange(len(blob)) % 5 == 0, 1, 0),
        np.full(len(ball), 2), np.full(len(glass), 3),
        np.where(np.arange(len(ring)) % 3 == 0, -1, 4)]).astype(np.int32)
    mats = [
        _mat("paint", kd=(0.03, 0.22, 0.06), ks=(0.5, 0.5, 0.5), ns=180.0, illum=3, d=1.0),
        _mat("tex", kd=(0.3, 0.3, 0.3), ks=(0.3, 0.3, 0.3), ns=40.0, illum=2, d=1.0, tex=checker_texture(44, 28, 6, (90, 25, 15), (95, 95, 90))),
        _mat("chrome", kd=(0.15, 0.15, 0.16), ks=None, ns=None, illum=4),                      # no d, no Ks, no Ns
        _mat("glass", kd=(0.7, 0.9, 0.8), ks=(1, 1, 1), ns=300.0, illum=6, d=0.35),         # transparent
        _mat("matte", kd=None, ks=(0.1, 0.1, 0.1), ns=None, illum=2, d=1.0),                # no Kd, no Ns
    ]
    things = Object(name="Things", triangles=tri, material_index=mi, materials=mats,
                    translation=np.array(CONFIG_YAML_TRANSLATION, F32),
                    rotation=Object.quaternion_from_euler_degrees(0.0, 25.0, 0.0),
                    scale=np.array([1.0, 1.0, 1.0], F32))
    knot = torus(2 * d, d, 0.6, 0.2, center=(0, 0, 0), axis=2)
    spin = Object(name="Spin", triangles=knot, material_index=np.zeros(len(knot), np.int32),
                  materials=[_mat("blue", kd=(0.05, 0.08, 0.3), ks=(0.6, 0.6, 0.6), ns=120.0, illum=2, d=1.0)],
                  translation=np.array([-1.2, 1.6, -2.0], F32),
                  rotation=Object.quaternion_from_euler_degrees(35.0, 10.0, 60.0),
                  scale=np.array([1.3, 0.7, 1.1], F32))
    st = Settings(skybox=Skybox(image=procedural_sky(256, 128, seed)) if sky_image else Skybox.default())
    return Scene(objects=[floor, things, spin], lights=config_yaml_lights(),
                 camera=config_yaml_camera(width, height), settings=st)
