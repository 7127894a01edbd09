"""CPU oracle for the neural-Gaussian decode.  TEST INFRASTRUCTURE ONLY.

`generate_neural_gaussians` below restates gaussian_renderer/__init__.py:18-102 line by line in plain torch (including
the feature-bank branch :39-49, off in every GScream config); the stand-in for the parts of scene/gaussian_model.py the
function touches (MLPs as :118-144, activations :43-54) lives in gscream_amd/standin_model.py.

PINNED against the reference's own code: tests/golden/make_reference_vectors2.py imports the reference's
gaussian_renderer.generate_neural_gaussians and GaussianModel.training_statis in the build container (stubs that raise
on use for the modules those functions never touch) and runs them on the same stand-in; tests/golden/ref_decode.npz /
ref_stats.npz hold what they computed (outputs, mask, all parameter gradients), tests/test_reference_vectors2.py compares."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gscream_amd.standin_model import Camera, Model  # noqa: E402,F401  (seeded parameter container, shared with bench.py)


def generate_neural_gaussians(viewpoint_camera, pc, visible_mask=None, is_training=False):
    if visible_mask is None:
        visible_mask = torch.ones(pc.get_anchor.shape[0], dtype=torch.bool, device=pc.get_anchor.device)
    feat = pc._anchor_feat[visible_mask]
    anchor = pc.get_anchor[visible_mask]
    grid_offsets = pc._offset[visible_mask]
    grid_scaling = pc.get_scaling[visible_mask]
    ob_view = anchor - viewpoint_camera.camera_center                     # :31
    ob_dist = ob_view.norm(dim=1, keepdim=True)                           # :33
    ob_view = ob_view / ob_dist                                           # :35
    if getattr(pc, "use_feat_bank", False):                               # :39-49 view-adaptive feature bank
        cat_view = torch.cat([ob_view, ob_dist], dim=1)
        bank_weight = pc.get_featurebank_mlp(cat_view).unsqueeze(dim=1)   # [n, 1, 3]
        feat = feat.unsqueeze(dim=-1)
        feat = (feat[:, ::4, :1].repeat([1, 4, 1]) * bank_weight[:, :, :1] + feat[:, ::2, :1].repeat([1, 2, 1]) * bank_weight[:, :, 1:2]
                + feat[:, ::1, :1] * bank_weight[:, :, 2:])
        feat = feat.squeeze(dim=-1)
    cat_local_view = torch.cat([feat, ob_view, ob_dist], dim=1)           # :52
    neural_opacity = pc.get_opacity_mlp(cat_local_view)                   # :55
    neural_opacity = neural_opacity.reshape([-1, 1])                      # :58
    mask = (neural_opacity > 0.0).view(-1)                                # :59-60
    opacity = neural_opacity[mask]                                        # :63
    K = pc.n_offsets
    uncertainty = pc.get_uncertainty_mlp(cat_local_view).reshape([anchor.shape[0] * K, 1])   # :66-67
    color = pc.get_color_mlp(cat_local_view).reshape([anchor.shape[0] * K, 3])               # :70-71
    scale_rot = pc.get_cov_mlp(cat_local_view).reshape([anchor.shape[0] * K, 7])             # :74-75
    offsets = grid_offsets.view([-1, 3])                                  # :78
    concatenated = torch.cat([grid_scaling, anchor], dim=-1)              # :81
    concatenated_repeated = concatenated.repeat_interleave(K, dim=0)      # :82 einops 'n (c) -> (n k) (c)'
    concatenated_all = torch.cat([concatenated_repeated, uncertainty, color, scale_rot, offsets], dim=-1)  # :84
    masked = concatenated_all[mask]                                       # :85
    scaling_repeat, repeat_anchor, uncertainty, color, scale_rot, offsets = masked.split([6, 3, 1, 3, 7, 3], dim=-1)  # :87
    scaling = scaling_repeat[:, 3:] * torch.sigmoid(scale_rot[:, :3])     # :90
    rot = pc.rotation_activation(scale_rot[:, 3:7])                       # :91
    offsets = offsets * scaling_repeat[:, :3]                             # :94
    xyz = repeat_anchor + offsets                                         # :95
    if is_training:
        return xyz, color, opacity, uncertainty, scaling, rot, neural_opacity, mask
    return xyz, color, opacity, uncertainty, scaling, rot


def training_statis(self, viewspace_point_tensor, opacity, update_filter, offset_selection_mask, anchor_visible_mask):
    """scene/gaussian_model.py:730-757, restated (`self` needs n_offsets and the four accumulators)."""
    temp_opacity = opacity.clone().view(-1).detach()
    temp_opacity[temp_opacity < 0] = 0
    temp_opacity = temp_opacity.view([-1, self.n_offsets])
    self.opacity_accum[anchor_visible_mask] += temp_opacity.sum(dim=1, keepdim=True)
    self.anchor_demon[anchor_visible_mask] += 1
    anchor_visible_mask = anchor_visible_mask.unsqueeze(dim=1).repeat([1, self.n_offsets]).view(-1)
    combined_mask = torch.zeros_like(self.offset_gradient_accum, dtype=torch.bool).squeeze(dim=1)
    combined_mask[anchor_visible_mask] = offset_selection_mask
    temp_mask = combined_mask.clone()
    combined_mask[temp_mask] = update_filter
    grad_norm = torch.norm(viewspace_point_tensor.grad[update_filter, :2], dim=-1, keepdim=True)
    self.offset_gradient_accum[combined_mask] += grad_norm
    self.offset_denom[combined_mask] += 1
