"""gscream_amd -- MI355X-native differentiable Gaussian rasterizer (the hot path of W-Ted/GScream).

Public surface = the reference's `diff_gaussian_rasterization` module:
`GaussianRasterizationSettings`, `GaussianRasterizer` (+ `rasterize_gaussians`).
The top-level package `diff_gaussian_rasterization/` re-exports them under the reference's module
name so that GScream's `gaussian_renderer/__init__.py:15` imports this build unchanged.
"""
from .rasterizer import (GaussianRasterizationSettings, GaussianRasterizer, rasterize_gaussians,  # noqa: F401
                         set_tuning)

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians", "set_tuning"]
