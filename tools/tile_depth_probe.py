#!/usr/bin/env python
"""Diagnostic: distribution of per-tile list lengths and traversal depths (tile_work = deepest n_contrib of the tile)
on a bench workload -- what bounds the blend kernels' critical path.  usage: python tools/tile_depth_probe.py [config2]"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import bench
from gscream_amd import GaussianRasterizationSettings, _native, _layout, rasterizer
from gscream_amd import synthetic as S

wl = sys.argv[1] if len(sys.argv) > 1 else "config2"
P, W, H, seed, gsel, desc = bench.WORKLOADS[wl]
dev = torch.device("cuda", 0)
_native.load()
s = S.scene_slab(seed, P, W, H)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
rs = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=s["tanfovx"], tanfovy=s["tanfovy"], bg=t(s["bg"]),
                                   scale_modifier=1.0, viewmatrix=t(s["viewmatrix"]), projmatrix=t(s["projmatrix"]),
                                   sh_degree=1, campos=t(s["campos"]), prefiltered=False, debug=False)
empty = torch.empty(0, device=dev)
out = rasterizer._forward_native(t(s["means3D"]), empty, t(s["colors"]), t(s["opacities"]), t(s["uncertainties"]),
                                 t(s["scales"]), t(s["rotations"]), empty, rs)
R, img = out[0], out[7]
v = _layout.image_views(img, P, W, H)
rg = v["ranges"].cpu().numpy().astype(np.int64)
n = rg[:, 1] - rg[:, 0]
work = v["tile_work"].cpu().numpy().astype(np.int64)
q = [0, 10, 25, 50, 75, 90, 99, 100]
print("R", R, "tiles", len(n))
print("list length   pct", q, np.percentile(n, q).astype(int), "mean", n.mean())
print("traversed     pct", q, np.percentile(work, q).astype(int), "mean", work.mean())
print("sum traversed / sum list", work.sum() / n.sum())
# how well do quantities known BEFORE the blend predict a tile's traversal depth?
nc = v["n_contrib"].cpu().numpy().astype(np.int64)
gx = (W + 15) // 16
print("corr(list length, traversed) =", round(float(np.corrcoef(n, work)[0, 1]), 3))
# per 8x8 quadrant: deepest pixel of the quadrant vs the tile's list length
Hq, Wq = (H + 7) // 8, (W + 7) // 8
pad = np.zeros((Hq * 8, Wq * 8), np.int64); pad[:H, :W] = nc
qd = pad.reshape(Hq, 8, Wq, 8).max(axis=(1, 3))
print("quadrant depth pct", q, np.percentile(qd, q).astype(int), "mean", qd.mean())
