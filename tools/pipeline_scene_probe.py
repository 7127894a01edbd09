#!/usr/bin/env python
"""Diagnostic: binning statistics of the scene bench.pipeline_row rasterizes (Gaussians decoded from clustered anchors)."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from gscream_amd import GaussianRasterizationSettings, _native, _layout, rasterizer
from gscream_amd import synthetic as S
from gscream_amd.neural_gaussians import generate_neural_gaussians
from gscream_amd import standin_model as DO

W, H, N, K = 1008, 567, 200_000, 10
dev = torch.device("cuda", 0)
_native.load()
model = DO.Model(N, K, seed=11, dtype=torch.float32, spread=1.5).to(dev)
w2c = np.eye(4, dtype=np.float32); w2c[2, 3] = 6.0
view, proj, campos = S.camera_matrices(0.6, 0.6 * H / W, w2c)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
cam = DO.Camera(t(campos))
rs = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=0.6, tanfovy=0.6 * H / W, bg=torch.zeros(3, device=dev),
                                   scale_modifier=1.0, viewmatrix=t(view), projmatrix=t(proj), sh_degree=1, campos=t(campos),
                                   prefiltered=False, debug=False)
with torch.no_grad():
    xyz, color, opacity, unc, scaling, rot, nop, mask = generate_neural_gaussians(cam, model, None, True)
P = xyz.shape[0]
e = torch.empty(0, device=dev)
out = rasterizer._forward_native(xyz.contiguous(), e, color.contiguous(), opacity.contiguous(), unc.contiguous(), scaling.contiguous(), rot.contiguous(), e, rs)
R, radii, geom, img = out[0], out[4], out[5], out[7]
iv = _layout.image_views(img, P, W, H)
gv = _layout.geom_views(geom, P)
rg = iv["ranges"].cpu().numpy().astype(np.int64)
n = rg[:, 1] - rg[:, 0]
work = iv["tile_work"].cpu().numpy().astype(np.int64)
q = [0, 10, 25, 50, 75, 90, 99, 100]
r = radii.cpu().numpy()
tiles = gv["tiles"].cpu().numpy()
print("P", P, "R", R, "visible", int((r > 0).sum()), "tiles/Gaussian mean", tiles[r > 0].mean(), "max", tiles.max())
print("radii pct", q, np.percentile(r[r > 0], q))
print("opacity pct", q, np.round(np.percentile(opacity.detach().cpu().numpy(), q), 3))
print("scaling pct", q, np.round(np.percentile(scaling.detach().cpu().numpy(), q), 4))
print("list length pct", q, np.percentile(n, q).astype(int), "mean", n.mean())
print("traversed   pct", q, np.percentile(work, q).astype(int), "mean", work.mean(), "sum/sumlist", work.sum() / max(n.sum(), 1))
per_wave = np.add.reduceat(tiles.astype(np.int64), np.arange(0, P, 64))
print("tiles per Gaussian pct", q, np.percentile(tiles[r > 0], q).astype(int))
print("slots per 64-Gaussian wave pct", q, np.percentile(per_wave, q).astype(int), "mean", per_wave.mean(), "top5", np.sort(per_wave)[-5:])
