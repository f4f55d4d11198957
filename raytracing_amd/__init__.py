"""raytracing_amd -- MI355X (gfx950) render backend for Trayracer's src/raytracer hot path.

Only what the path needs lives here: csrc/ (HIP kernels + the C ABI of include/trayracer_hip.h), the host
mirror of the reference's Scene / Raytracer / Render interface, the procedural stand-in scenes and the
tile-sharded multi-GPU driver.  The HIP library is required; there is no CPU fallback.
"""
from .scene import Camera, IlluminationModel, Light, Material, Object, Scene, Settings, Skybox  # noqa: F401

__version__ = "0.1.0"
