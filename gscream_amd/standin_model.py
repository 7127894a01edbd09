"""Stand-in for the parts of GScream's `GaussianModel` / `Camera` that `generate_neural_gaussians` and `render()` read
(scene/gaussian_model.py:118-144 MLPs, :43-54 activations, :241-242 get_scaling; scene/cameras.py camera_center).
A seeded parameter container for tests and bench.py -- no arithmetic of the path lives here."""
import torch
from torch import nn


class Camera:
    def __init__(self, center):
        self.camera_center = center


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
        if use_feat_bank:                                # :107-113 (view-adaptive feature bank, off in every shipped config)
            self.mlp_feature_bank = nn.Sequential(nn.Linear(3 + 1, feat_dim), nn.ReLU(True), nn.Linear(feat_dim, 3),
                                                  nn.Softmax(dim=1)).to(dtype)

    get_anchor = property(lambda self: self._anchor)
    get_scaling = property(lambda self: 1.0 * torch.exp(self._scaling))  # :241-242
    get_opacity_mlp = property(lambda self: self.mlp_opacity)
    get_uncertainty_mlp = property(lambda self: self.mlp_uncertainty)
    get_cov_mlp = property(lambda self: self.mlp_cov)
    get_color_mlp = property(lambda self: self.mlp_color)
    get_featurebank_mlp = property(lambda self: self.mlp_feature_bank)  # :245-246
