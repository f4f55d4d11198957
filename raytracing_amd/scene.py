"""Host-side mirror of the reference's ``src/scene`` data model (plain data, no compute).

The reference keeps ``Scene{objects, lights, camera, settings}`` (src/scene/mod.rs:32-42) and hands a deep
clone of it to ``Raytracer::new`` (src/raytracer/render.rs:60,105).  This module keeps the same names and
field meanings so the parity tests read like reference code; the arithmetic lives in the HIP library.

Value conventions that the reference applies at LOAD time and that callers of this module must honour:
  * ``Light.color`` is the *normalised* ``Ke`` (src/scene/light.rs:33),
  * ``Settings.ambient_color`` is normalised and ``ambient_intensity`` is the norm of the YAML colour
    (src/scene/settings.rs:51-55),
  * ``Material.dissolve`` is ``1 - d`` (src/scene/object.rs:106),
  * ``Camera.fov`` is in radians (src/scene/camera.rs:75).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np

F32 = np.float32


def _v3(v) -> np.ndarray:
    a = np.asarray(v, dtype=F32).reshape(3)
    return a


class IlluminationModel:
    """src/scene/material.rs:31-53."""

    def __init__(self, value: int = 0):
        if not 0 <= int(value) <= 10:
            raise ValueError(f"Invalid illumination model: {value}")
        self.value = int(value)

    @classmethod
    def from_i32(cls, i: int) -> Optional["IlluminationModel"]:
        return cls(i) if 0 <= int(i) <= 10 else None

    def specular(self) -> bool:
        return self.value == 2

    def reflection(self) -> bool:
        return self.value in (3, 4)

    def transparency(self) -> bool:
        return self.value in (6, 7)

    def __eq__(self, other):
        return isinstance(other, IlluminationModel) and other.value == self.value

    def __repr__(self):
        return f"IlluminationModel({self.value})"


@dataclass
class Material:
    """src/scene/material.rs:5-15.  ``None`` fields are the reference's ``Option::None``."""

    name: str = ""
    diffuse_color: Optional[np.ndarray] = None      # Kd
    specular_color: Optional[np.ndarray] = None     # Ks
    specular_exponent: Optional[float] = None       # Ns
    diffuse_texture: Optional[np.ndarray] = None    # map_Kd as uint8 [h, w, 3]
    illumination_model: IlluminationModel = field(default_factory=IlluminationModel)
    dissolve: Optional[float] = None                # 1 - d
    refraction_index: Optional[float] = None        # Ni (unused by the reference too)


@dataclass
class Light:
    """src/scene/light.rs:5-9."""

    position: np.ndarray
    color: np.ndarray
    intensity: float

    @classmethod
    def from_ke(cls, position, ke, intensity) -> "Light":
        """YAML constructor: colour = Ke.try_normalize(0.0).unwrap_or_default() (light.rs:33)."""
        ke = _v3(ke)
        n = _norm3(ke)
        color = (ke / n).astype(F32) if n > 0 else np.zeros(3, F32)
        return cls(_v3(position), color, float(intensity))


def _norm3(v: np.ndarray) -> np.float32:
    v = v.astype(F32)
    s = F32(F32(v[0] * v[0]) + F32(v[1] * v[1]))
    s = F32(s + F32(v[2] * v[2]))
    return np.sqrt(s, dtype=F32)


@dataclass
class Skybox:
    """src/scene/skybox.rs:6-9: either a constant colour or an 8-bit RGB equirect image."""

    color: Optional[np.ndarray] = None
    image: Optional[np.ndarray] = None   # uint8 [h, w, 3]
    path: str = ""                       # where ``image`` came from (skybox.rs:7), for serialisation only

    @classmethod
    def default(cls) -> "Skybox":
        return cls(color=np.array([0.16, 0.16, 0.16], F32))   # skybox.rs:13


@dataclass
class Settings:
    """src/scene/settings.rs:4-25.  ``seed`` is ours (counter RNG replacing rand::random)."""

    max_bounces: int = 4     # unused by the reference as well (settings.rs:5)
    samples: int = 1
    ambient_color: np.ndarray = field(
        default_factory=lambda: (np.array([0.34, 0.14, 0.04], F32) / _norm3(np.array([0.34, 0.14, 0.04], F32))).astype(F32))
    ambient_intensity: float = 0.2
    skybox: Skybox = field(default_factory=Skybox.default)
    anti_aliasing: bool = False
    seed: int = 1

    @classmethod
    def from_yaml_color(cls, ambient_color, **kw) -> "Settings":
        """settings.rs:51-55: colour normalised, intensity = norm of the YAML colour."""
        c = _v3(ambient_color)
        n = _norm3(c)
        col = (c / n).astype(F32) if n > 0 else np.zeros(3, F32)
        return cls(ambient_color=col, ambient_intensity=float(n), **kw)


@dataclass
class Camera:
    """src/scene/camera.rs:5-23."""

    position: np.ndarray = field(default_factory=lambda: np.array([1, 1, 1], F32))
    look_at: np.ndarray = field(default_factory=lambda: np.zeros(3, F32))
    up: np.ndarray = field(default_factory=lambda: np.array([0, 1, 0], F32))
    fov: float = float(np.float32(np.deg2rad(np.float32(60.0))))
    resolution: Tuple[int, int] = (1920, 1080)


@dataclass
class Object:
    """src/scene/object.rs:19-29.

    ``triangles`` is float32 [n, 24] laid out exactly like ``Triangle`` (src/scene/triangle.rs:9-21):
    a, b, c, a_normal, b_normal, c_normal (3 each) then a_uv, b_uv, c_uv (2 each);
    ``material_index`` is int32 [n] into ``materials`` with -1 for ``None``.  A triangle's ID is
    (index of the object in ``Scene.objects``, row in ``triangles``) -- the reference's ``Vec`` order.
    ``rotation`` is the unit quaternion (i, j, k, w).
    """

    name: str = ""
    triangles: np.ndarray = field(default_factory=lambda: np.zeros((0, 24), F32))
    material_index: np.ndarray = field(default_factory=lambda: np.zeros((0,), np.int32))
    materials: List[Material] = field(default_factory=list)
    translation: np.ndarray = field(default_factory=lambda: np.zeros(3, F32))
    rotation: np.ndarray = field(default_factory=lambda: np.array([0, 0, 0, 1], F32))
    scale: np.ndarray = field(default_factory=lambda: np.ones(3, F32))
    material_name: str = ""   # names of the OBJ's `o` objects joined by ", " (object.rs:160-166); Hit.name, never shaded
    path: str = ""            # file path as written in the scene YAML (object.rs:352-355)

    def __post_init__(self):
        self.triangles = np.ascontiguousarray(self.triangles, dtype=F32).reshape(-1, 24)
        self.material_index = np.ascontiguousarray(self.material_index, dtype=np.int32).reshape(-1)
        if len(self.material_index) != len(self.triangles):
            raise ValueError("material_index and triangles differ in length")
        self.translation = _v3(self.translation)
        self.scale = _v3(self.scale)
        self.rotation = np.asarray(self.rotation, dtype=F32).reshape(4)

    @staticmethod
    def quaternion_from_euler_degrees(x: float, y: float, z: float) -> np.ndarray:
        """object.rs:336-340: degrees * PI * 2 / 360, then UnitQuaternion::from_euler_angles(roll, pitch, yaw)
        (nalgebra half-angle product formula, recalled: [crate], fp32 throughout)."""
        pi = F32(np.pi)

        def rad(d):
            return F32(F32(F32(F32(d) * pi) * F32(2.0)) / F32(360.0))

        def sc(a):
            h = F32(a * F32(0.5))
            # sinf/cosf: round the double result (== glibc's correctly rounded sinf/cosf in all but
            # vanishingly rare double-rounding cases; numpy's float32 SIMD loops are only ~1-2 ulp)
            return F32(math.sin(float(h))), F32(math.cos(float(h)))

        sr, cr = sc(rad(x))
        sp, cp = sc(rad(y))
        sy, cy = sc(rad(z))
        w = F32(F32(F32(cr * cp) * cy) + F32(F32(sr * sp) * sy))
        i = F32(F32(F32(sr * cp) * cy) - F32(F32(cr * sp) * sy))
        j = F32(F32(F32(cr * sp) * cy) + F32(F32(sr * cp) * sy))
        k = F32(F32(F32(cr * cp) * sy) - F32(F32(sr * sp) * cy))
        return np.array([i, j, k, w], F32)


@dataclass
class Scene:
    """src/scene/mod.rs:32-42."""

    objects: List[Object] = field(default_factory=list)
    lights: List[Light] = field(default_factory=list)
    camera: Camera = field(default_factory=Camera)
    settings: Settings = field(default_factory=Settings)
    path: str = ""

    def triangle_count(self) -> int:
        return int(sum(len(o.triangles) for o in self.objects))
