"""End-to-end: gscream_amd.gaussian_renderer.render / prefilter_voxel (decode -> rasterize on the HIP rows) against the
CPU oracles chained the same way (oracle decode in float64 -> C oracle rasterizer)."""
import math
import os
import sys
import types

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import helpers as Hh  # noqa: E402
from gscream_amd import synthetic as S  # noqa: E402
from oracle import decode_oracle as DO  # noqa: E402
from oracle import oracle as O  # noqa: E402

pytestmark = pytest.mark.gpu


def _camera(W, H, tanfovx, device, dtype):
    w2c = np.eye(4, dtype=np.float32)
    w2c[2, 3] = 6.0
    tanfovy = tanfovx * H / W
    view, proj, campos = S.camera_matrices(tanfovx, tanfovy, w2c)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device=device, dtype=dtype)
    cam = types.SimpleNamespace(image_height=H, image_width=W, FoVx=2 * math.atan(tanfovx), FoVy=2 * math.atan(tanfovy),
                                world_view_transform=t(view), full_proj_transform=t(proj), camera_center=t(campos))
    return cam, view, proj, campos, tanfovy


class _Model(DO.Model):
    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        g = torch.Generator().manual_seed(99)
        self._rotation = torch.nn.Parameter(torch.randn(self._anchor.shape[0], 4, generator=g, dtype=torch.float64).to(self._anchor.dtype))

    get_rotation = property(lambda self: torch.nn.functional.normalize(self._rotation))


def test_render_matches_chained_oracles():
    import copy
    from gscream_amd.gaussian_renderer import prefilter_voxel, render
    N, K, W, H, tanfovx = 1500, 10, 176, 112, 0.55
    ref = _Model(N, K, seed=21, dtype=torch.float64, spread=1.2)
    dut = copy.deepcopy(ref).float().cuda()
    dut.train()
    pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=False)
    cam_d, view, proj, campos, tanfovy = _camera(W, H, tanfovx, "cuda", torch.float32)
    bg = torch.tensor([0.1, 0.2, 0.3], device="cuda")
    vis = prefilter_voxel(cam_d, dut, pipe, bg)
    assert vis.dtype == torch.bool and vis.shape == (N,) and vis.any()
    pkg = render(cam_d, dut, pipe, bg, visible_mask=vis, retain_grad=True)
    assert set(pkg) >= {"render", "render_depth", "uncertainty", "viewspace_points", "visibility_filter", "radii",
                        "selection_mask", "neural_opacity", "scaling"}
    # oracle chain: float64 decode (same visibility mask) -> C rasterizer oracle
    cam_r = DO.Camera(torch.from_numpy(campos).double())
    xyz, color, opacity, unc, scaling, rot, nop, mask = DO.generate_neural_gaussians(cam_r, ref, vis.cpu(), True)
    assert torch.equal(mask, pkg["selection_mask"].cpu())
    f = lambda t: t.detach().float().numpy()
    scene = dict(means3D=f(xyz), colors=f(color), opacities=f(opacity), uncertainties=f(unc), scales=f(scaling), rotations=f(rot),
                 W=W, H=H, tanfovx=tanfovx, tanfovy=tanfovy, viewmatrix=view, projmatrix=proj, campos=campos,
                 bg=np.array([0.1, 0.2, 0.3], np.float32), scale_modifier=1.0)
    st = Hh.oracle_forward(scene)
    assert (pkg["radii"].cpu().numpy() == st["radii"]).mean() > 0.999  # fp32 decode vs fp64: a radius may differ by one
    for k, o in (("render", "out_color"), ("render_depth", "out_depth"), ("uncertainty", "out_unc")):
        diff = np.abs(pkg[k].detach().cpu().numpy() - st[o])
        assert np.percentile(diff, 99.9) < 2e-4 and diff.max() < 5e-2, (k, float(diff.max()))
    # gradients reach every parameter group through both rows
    (pkg["render"].mean() + 0.1 * pkg["render_depth"].mean()).backward()
    for name in ("_anchor", "_anchor_feat", "_offset", "_scaling"):
        gr = getattr(dut, name).grad
        assert gr is not None and torch.isfinite(gr).all() and gr.abs().sum() > 0, name
    assert pkg["viewspace_points"].grad is not None and pkg["viewspace_points"].grad.abs().sum() > 0
    dut.eval()
    with torch.no_grad():
        ev = render(cam_d, dut, pipe, bg, visible_mask=vis)
    assert "selection_mask" not in ev and torch.allclose(ev["render"], pkg["render"].detach(), atol=1e-6)
