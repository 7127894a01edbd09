"""Drop-in module name of the reference's rasterizer (GScream imports it at
gaussian_renderer/__init__.py:15: `from diff_gaussian_rasterization import
GaussianRasterizationSettings, GaussianRasterizer`).  Everything lives in gscream_amd."""
from gscream_amd.rasterizer import (GaussianRasterizationSettings, GaussianRasterizer,  # noqa: F401
                                    rasterize_gaussians)

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians"]
