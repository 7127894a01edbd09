"""Latency of ONE wavefront's walk (GPU box): a 16x16 image (one tile, four quadrant waves on four SIMDs, nothing else on the
chip) under a stack of N large faint splats that every pixel blends and that never saturates.  The forward / backward blend
time divided by N is what a lone wave needs per instance -- the speed the tail of a full-size launch runs at.
usage: python tools/lone_wave_probe.py [N]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gscream_amd import GaussianRasterizationSettings, GaussianRasterizer, _native  # noqa: E402
from gscream_amd import synthetic as S  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
W = H = 16
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
view, proj, campos = S.camera_matrices(0.5, 0.5)
means = np.stack([rng.uniform(-0.01, 0.01, N), rng.uniform(-0.01, 0.01, N), np.linspace(2.0, 6.0, N)], 1).astype(np.float32)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
leaves = [t(means).requires_grad_(True),
          t(np.full((N, 1), 0.0043, np.float32)).requires_grad_(True),      # alpha ~ 1.1 / 255 everywhere, T stays > 1e-4 for N <= 2100
          t(rng.uniform(0, 1, (N, 1)).astype(np.float32)).requires_grad_(True),
          t(rng.uniform(0, 1, (N, 3)).astype(np.float32)).requires_grad_(True),
          t(np.full((N, 3), 3.0, np.float32)).requires_grad_(True),         # huge splats: G ~ 1 over the tile
          t(S._quats(rng, N)).requires_grad_(True)]
means3D, opac, unc, colors, scales, rots = leaves
means2D = torch.zeros_like(means3D, requires_grad=True)
rs = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=0.5, tanfovy=0.5, bg=t(np.zeros(3, np.float32)),
                                   scale_modifier=1.0, viewmatrix=t(view), projmatrix=t(proj), campos=t(campos), sh_degree=1,
                                   prefiltered=False, debug=False)
rast = GaussianRasterizer(raster_settings=rs)
g = torch.ones(3, H, W, device=dev)


def step():
    color, depth, feat, radii = rast(means3D, means2D, opac, unc, colors_precomp=colors, scales=scales, rotations=rots)
    torch.autograd.grad([color], leaves, [g])
    return color, radii


for _ in range(5):
    color, radii = step()
torch.cuda.synchronize()
from gscream_amd import rasterizer as RZ  # noqa: E402
print("N", N, "visible", int((radii > 0).sum()), "R", RZ._last_stage1["num_rendered"], "colour min / mean / max",
      float(color.detach().min()), float(color.detach().mean()), float(color.detach().max()))
_native.profile_begin()
for _ in range(20):
    step()
torch.cuda.synchronize()
prof = _native.profile_end()
for k, v in prof.items():
    if v[1]:
        us = v[0] / v[1] * 1e3
        print(f"  {k:16s} {us:8.1f} us" + (f"   = {us * 1e3 / N:6.1f} ns per instance" if "blend" in k else ""))
