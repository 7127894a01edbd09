"""GScream's `gaussian_renderer` entry points on the HIP rows: same names, arguments and return
dictionaries as gaussian_renderer/__init__.py:104-179 (`render`), :182-235 (`prefilter_voxel`), :240-296
(`prefilter_position2D`), built from this package's pieces:

    generate_neural_gaussians  ->  gscream_amd.neural_gaussians   (fused decode + compaction, gsr_decode_*)
    GaussianRasterizer         ->  gscream_amd.rasterizer         (gsr_forward / gsr_backward / gsr_filter)

The reference module keeps working unmodified on top of `diff_gaussian_rasterization`; importing
`render` from here additionally swaps the decode.  `pc` is the reference's GaussianModel (or anything exposing the
same attributes), `viewpoint_camera` its Camera, `pipe` its PipelineParams (`debug`, `compute_cov3D_python`)."""
import math

import torch

from .neural_gaussians import generate_neural_gaussians
from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer

__all__ = ["generate_neural_gaussians", "render", "prefilter_voxel", "prefilter_position2D"]


def _rasterizer(viewpoint_camera, pipe, bg_color, scaling_modifier):
    settings = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height), image_width=int(viewpoint_camera.image_width),
        tanfovx=math.tan(viewpoint_camera.FoVx * 0.5), tanfovy=math.tan(viewpoint_camera.FoVy * 0.5), bg=bg_color,
        scale_modifier=scaling_modifier, viewmatrix=viewpoint_camera.world_view_transform,
        projmatrix=viewpoint_camera.full_proj_transform, sh_degree=1, campos=viewpoint_camera.camera_center,
        prefiltered=False, debug=bool(getattr(pipe, "debug", False)))
    return GaussianRasterizer(raster_settings=settings)


def render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, visible_mask=None, retain_grad=False):
    """Decode the visible anchors into Gaussians and rasterize them.  Background tensor (bg_color) must be on the GPU."""
    is_training = pc.get_color_mlp.training
    out = generate_neural_gaussians(viewpoint_camera, pc, visible_mask, is_training=is_training)
    xyz, color, opacity, uncertainty, scaling, rot = out[:6]
    # gradient carrier of the 2-D (screen-space) means, as in the reference (:120-125)
    screenspace_points = torch.zeros_like(xyz, dtype=pc.get_anchor.dtype, requires_grad=True, device=xyz.device)
    if retain_grad:
        # a leaf keeps its .grad by itself: the reference's "+ 0" (a non-leaf copy, one more model-sized elementwise kernel per
        # iteration) followed by retain_grad() gives the caller the same thing
        try:
            screenspace_points.retain_grad()  # no-op on a leaf
        except Exception:  # noqa: BLE001  (the reference swallows this too)
            pass
    else:
        screenspace_points = screenspace_points + 0  # the reference's non-leaf, whose gradient is not kept
    rendered_image, rendered_depth, uncer, radii = _rasterizer(viewpoint_camera, pipe, bg_color, scaling_modifier)(
        means3D=xyz, means2D=screenspace_points, shs=None, colors_precomp=color, opacities=opacity,
        uncertainties=uncertainty, scales=scaling, rotations=rot, cov3D_precomp=None)
    result = {"render": rendered_image, "render_depth": rendered_depth, "uncertainty": uncer,
              "viewspace_points": screenspace_points, "visibility_filter": radii > 0, "radii": radii}
    if is_training:
        result.update(selection_mask=out[7], neural_opacity=out[6], scaling=scaling)
    return result


def _anchor_cov_inputs(pc, pipe, scaling_modifier):
    if getattr(pipe, "compute_cov3D_python", False):
        return None, None, pc.get_covariance(scaling_modifier)
    return pc.get_scaling[:, :3], pc.get_rotation, None


def prefilter_voxel(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None):
    """Which anchors project to a non-empty footprint (train.py:433): radii > 0 of the anchor cloud."""
    scales, rotations, cov3D_precomp = _anchor_cov_inputs(pc, pipe, scaling_modifier)
    radii = _rasterizer(viewpoint_camera, pipe, bg_color, scaling_modifier).visible_filter(
        means3D=pc.get_anchor, scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp)
    return radii > 0


def prefilter_position2D(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None):
    scales, rotations, cov3D_precomp = _anchor_cov_inputs(pc, pipe, scaling_modifier)
    radii, x, y = _rasterizer(viewpoint_camera, pipe, bg_color, scaling_modifier).position2D_filter(
        means3D=pc.get_anchor, scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp)
    return radii > 0, x, y
