#!/usr/bin/env python
"""Diagnostic: cost of the partial sort's fall-back.  The large-splat scene of bench.pipeline_row with its opacities
scaled down so that (almost) no pixel saturates: every tile with a long list runs off its sorted prefix and is redone
after a full sort.  Prints forward times with partial sorting on / off for a few opacity scales."""
import sys
import time
import numpy as np
import torch
sys.path.insert(0, ".")
from gscream_amd import GaussianRasterizationSettings, _native, _layout, rasterizer, set_tuning
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
    xyz, color, opacity, unc, scaling, rot, nop, mask = [x.contiguous() if torch.is_tensor(x) else x for x in generate_neural_gaussians(cam, model, None, True)]
e = torch.empty(0, device=dev)
for osc in (1.0, 0.3, 0.05):
    op = (opacity * osc).contiguous()
    for partial in (True, False):
        set_tuning(partial_sort=partial)
        for _ in range(3):
            out = rasterizer._forward_native(xyz, e, color, op, unc, scaling, rot, e, rs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            out = rasterizer._forward_native(xyz, e, color, op, unc, scaling, rot, e, rs)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 10 * 1e3
        iv = _layout.image_views(out[7], xyz.shape[0], W, H)
        print(f"opacity x{osc}: partial_sort={partial}: forward {ms:.3f} ms, flagged tiles {int(iv['need_full'].sum())} of {iv['need_full'].numel()}, "
              f"mean traversed {float(iv['tile_work'].float().mean()):.0f}")
set_tuning()
