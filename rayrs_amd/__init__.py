"""rayrs_amd -- MI355X (gfx950) build of rayrs-lib's per-pixel radiance integrator.

The product is librayrs_hip.so (hand-written HIP kernels behind the C ABI of
include/rayrs_hip.h); this package is the thin host-side mirror of the
rayrs-lib Scene / Camera / Object / Material interface on top of it.  Nothing
here computes radiance on the CPU and nothing falls back to a CPU path.
"""
from .api import (Axis, BvhHeuristic, Camera, Emission, Fresnel, Material, Object, Scene, frame_sample_chunk,
                  make_params, render, render_finish, render_launch, render_multi)

__all__ = ["Axis", "BvhHeuristic", "Camera", "Emission", "Fresnel", "Material", "Object", "Scene",
           "frame_sample_chunk", "make_params", "render", "render_finish", "render_launch", "render_multi"]
