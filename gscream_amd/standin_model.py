"""Stand-in for the parts of GScream's `GaussianModel` / `Camera` that `generate_neural_gaussians` and `render()` read
(scene/gaussian_model.py:118-144 MLPs, :43-54 activations, :241-242 get_scaling, :265-266 get_rotation, the
densification accumulators; scene/cameras.py camera_center and the fields render() / prefilter_*() read).
A seeded parameter container for tests and bench.py -- no arithmetic of the path lives here."""
import numpy as np
import torch
from torch import nn


def voxelize(points, voxel_size=0.001):
    """GaussianModel.voxelize_sample (scene/gaussian_model.py:295-299): snap to a voxel_size grid and keep one point per voxel
    (np.unique sorts, so the reference's shuffle in front of it has no effect on the result).  0.001 is the reference's default
    (arguments/__init__.py:52); `voxel_size <= 0` there means "the median 3-NN distance" (:306-313), which the caller computes."""
    points = np.asarray(points)
    return np.unique(np.round(points / voxel_size), axis=0) * voxel_size


class Camera:
    """What the decode reads (camera_center) and, when the optional fields are given, what render() / prefilter_*() read
    from the reference's Camera (scene/cameras.py:17-69)."""

    def __init__(self, center, image_height=None, image_width=None, FoVx=None, FoVy=None, world_view_transform=None,
                 full_proj_transform=None):
        self.camera_center = center
        self.image_height, self.image_width, self.FoVx, self.FoVy = image_height, image_width, FoVx, FoVy
        self.world_view_transform, self.full_proj_transform = world_view_transform, full_proj_transform


class Model(nn.Module):
    """What generate_neural_gaussians reads from GaussianModel."""

    def __init__(self, N, K=10, feat_dim=32, seed=0, dtype=torch.float64, spread=3.0, use_feat_bank=False):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        r = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)
        self.n_offsets, self.use_feat_bank = K, bool(use_feat_bank)
        self._anchor = nn.Parameter((r(N, 3) * spread).to(dtype))
        self._anchor_feat = nn.Parameter((r(N, feat_dim) * 0.5).to(dtype))
        self._offset = nn.Parameter((r(N, K, 3) * 0.3).to(dtype))
        self._scaling = nn.Parameter((r(N, 6) * 0.3 - 2.0).to(dtype))
        mk = lambda out, act: nn.Sequential(nn.Linear(feat_dim + 3 + 1, feat_dim), nn.ReLU(True), nn.Linear(feat_dim, out),
                                            *([act] if act is not None else [])).to(dtype)
        torch.manual_seed(seed + 1)
        self.mlp_opacity = mk(K, nn.Tanh())             # scene/gaussian_model.py:118-123
        self.mlp_uncertainty = mk(K, nn.Sigmoid())       # :125-131
        self.mlp_cov = mk(7 * K, None)                   # :133-137
        self.mlp_color = mk(3 * K, nn.Sigmoid())         # :139-144
        self.rotation_activation = torch.nn.functional.normalize  # :54
        rot = torch.zeros((N, 4), dtype=dtype)
        rot[:, 0] = 1.0                                   # anchors carry the identity quaternion (:439-440)
        self.register_buffer("_rotation", rot, persistent=False)  # a buffer: the decode's parameter list stays as it was
        # densification statistics (:211-215 / training_setup :332-336), updated in place by training_statis
        for name, shape in (("opacity_accum", (N, 1)), ("anchor_demon", (N, 1)), ("offset_gradient_accum", (N * K, 1)),
                            ("offset_denom", (N * K, 1))):
            self.register_buffer(name, torch.zeros(shape, dtype=torch.float32), persistent=False)
        if use_feat_bank:                                # :107-113 (view-adaptive feature bank, off in every shipped config)
            self.mlp_feature_bank = nn.Sequential(nn.Linear(3 + 1, feat_dim), nn.ReLU(True), nn.Linear(feat_dim, 3),
                                                  nn.Softmax(dim=1)).to(dtype)

    @classmethod
    def from_pcd(cls, anchors, dist2, K=10, feat_dim=32, seed=0):
        """The state GaussianModel.create_from_pcd leaves (scene/gaussian_model.py:301-345) for the voxelised points `anchors` [N,3]
        and their clamped mean 3-NN squared distances `dist2` [N] (simple_knn.distCUDA2, clamp_min 1e-7): `_scaling` = log(sqrt(dist2))
        in all six columns, zero offsets and anchor features, identity rotations; the MLPs keep nn.Linear's default initialisation
        (the reference builds them with plain nn.Sequential(nn.Linear, ...), :118-144)."""
        N = int(anchors.shape[0])
        m = cls(N, K, feat_dim=feat_dim, seed=seed, dtype=torch.float32)
        with torch.no_grad():
            m._anchor.copy_(anchors.detach().float().cpu())
            m._anchor_feat.zero_()
            m._offset.zero_()
            m._scaling.copy_(torch.log(torch.sqrt(dist2.detach().float().cpu()))[:, None].repeat(1, 6))
        return m

    get_anchor = property(lambda self: self._anchor)
    get_scaling = property(lambda self: 1.0 * torch.exp(self._scaling))  # :241-242
    get_rotation = property(lambda self: self.rotation_activation(self._rotation))  # :262-263
    get_opacity_mlp = property(lambda self: self.mlp_opacity)
    get_uncertainty_mlp = property(lambda self: self.mlp_uncertainty)
    get_cov_mlp = property(lambda self: self.mlp_cov)
    get_color_mlp = property(lambda self: self.mlp_color)
    get_featurebank_mlp = property(lambda self: self.mlp_feature_bank)  # :245-246
