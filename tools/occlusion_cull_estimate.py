"""What would a conservative per-tile occlusion cut-off remove?  (GPU box; analysis only.)
For the large-splat scene of the train-iteration row (and the bench scene): every (Gaussian, tile) instance gets the SMALLEST alpha
it has anywhere in its tile (the quadratic form is convex: its maximum over the tile box is at a corner); along a tile's depth-sorted
list the product of (1 - alpha_min) bounds every pixel's transmittance from above, so all instances behind the position where that
bound falls below 1e-4 are blended by no pixel.  Prints how many instances lie in front of that position, next to how many the
forward really walks (tile_work) and how many are binned today.
usage: python tools/occlusion_cull_estimate.py [large|bench]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gscream_amd import GaussianRasterizationSettings, _layout, synthetic as S  # noqa: E402
from gscream_amd import rasterizer as RZ  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "large"
dev = torch.device("cuda", 0)
W, H = 1008, 567
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
if which == "large":
    from gscream_amd import standin_model as SM
    from gscream_amd.neural_gaussians import generate_neural_gaussians
    model = SM.Model(200_000, 10, seed=11, dtype=torch.float32, spread=1.5).to(dev)
    w2c = np.eye(4, dtype=np.float32)
    w2c[2, 3] = 6.0
    view, proj, campos = S.camera_matrices(0.6, 0.6 * H / W, w2c)
    cam = SM.Camera(t(campos))
    with torch.no_grad():
        xyz, color, opacity, unc, scaling, rot, nop, mask = generate_neural_gaussians(cam, model, None, True)
    tfx, tfy = 0.6, 0.6 * H / W
else:
    s = S.scene_slab(1, 1_000_000, W, H)
    xyz, color, opacity, unc, scaling, rot = (t(s[k]) for k in ("means3D", "colors", "opacities", "uncertainties", "scales", "rotations"))
    view, proj, campos, tfx, tfy = s["viewmatrix"], s["projmatrix"], s["campos"], s["tanfovx"], s["tanfovy"]
rs = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=tfx, tanfovy=tfy, bg=torch.zeros(3, device=dev),
                                   scale_modifier=1.0, viewmatrix=t(view), projmatrix=t(proj), sh_degree=1, campos=t(campos),
                                   prefiltered=False, debug=False)
e = torch.Tensor([])
P = xyz.shape[0]
R, col, dep, un, radii, geom, binning, img, cap = RZ._forward_native(xyz.detach().contiguous(), e, color.detach().contiguous(), opacity.detach().contiguous(),
                                                                      unc.detach().contiguous(), scaling.detach().contiguous(), rot.detach().contiguous(), e, rs)
torch.cuda.synchronize()
gv, iv, bv = _layout.geom_views(geom, P), _layout.image_views(img, P, W, H), _layout.binning_views(binning, R, cap)
ranges = iv["ranges"].long()
T = ranges.shape[0]
gx = (W + 15) // 16
keys = bv["seg_keys"].clone()                       # (depth bits << 32 | id), tile segments, unsorted inside a segment
n = (ranges[:, 1] - ranges[:, 0])
tile_of = torch.repeat_interleave(torch.arange(T, device=dev), n)
# sort by (tile, key): keys are < 2^63, tiles < 2^12 -> two stable sorts
o1 = torch.argsort(keys, stable=True)
o2 = torch.argsort(tile_of[o1], stable=True)
order = o1[o2]
keys, tile_of = keys[order], tile_of[order]
gid = (keys & 0xffffffff).long()
rec = gv["rec_f32"]
px, py, cA, cB, cC, op = (rec[gid, i] for i in (0, 1, 2, 3, 4, 5))   # the records hold the raw conic (round 4)
L2E = 1.4426950408889634
hA, hB, hC = cA * (-0.5 * L2E), cB * (-L2E), cC * (-0.5 * L2E)          # the blend loops' pre-scaled form
tx, ty = (tile_of % gx).float() * 16, (tile_of // gx).float() * 16
x0, x1 = tx, torch.clamp(tx + 15, max=W - 1)
y0, y1 = ty, torch.clamp(ty + 15, max=H - 1)
pmin = None
for cx in (x0, x1):
    for cy in (y0, y1):
        dx, dy = px - cx, py - cy
        pw = dy * (hC * dy) + dx * (hA * dx + hB * dy)   # log2 of the falloff
        pmin = pw if pmin is None else torch.minimum(pmin, pw)
amin = torch.clamp(op * torch.exp2(pmin), max=0.99)
amin = torch.where((amin >= 1.0 / 255.0) & (pmin <= 0), amin, torch.zeros_like(amin))
lg = torch.log2(1.0 - amin).double()
cs = torch.cumsum(lg, 0)
seg_start = ranges[:, 0]
base = torch.where(seg_start > 0, cs[torch.clamp(seg_start - 1, min=0)], torch.zeros_like(cs[:1]).expand(T))
rel = cs - base[tile_of] - lg                        # log2 of the bound IN FRONT of each instance
alive = rel >= np.log2(1e-4)                         # an instance whose bound in front of it is still >= 1e-4 may blend
keep = torch.zeros(T, device=dev, dtype=torch.long).scatter_add_(0, tile_of, alive.long())
work = iv["tile_work"].long()
print(f"scene {which}: P {P}  instances binned today R = {R}  (longest list {int(n.max())})")
print(f"  walked by the forward (sum of tile_work)      {int(work.sum()):10d}  = {100.0 * int(work.sum()) / R:5.1f} % of R")
print(f"  in front of the conservative cut-off          {int(keep.sum()):10d}  = {100.0 * int(keep.sum()) / R:5.1f} % of R")
print(f"  instances with a non-zero whole-tile alpha    {int((amin > 0).sum()):10d}  = {100.0 * int((amin > 0).sum()) / R:5.1f} % of R")
for nb in (16, 32, 64):                               # bucketed cut-off: depth buckets of equal population per tile are not available on
    pos = torch.arange(R, device=dev) - seg_start[tile_of]    # the fly; equal-width buckets in list position are an optimistic stand-in
    width = torch.clamp((n[tile_of] + nb - 1) // nb, min=1)
    bstart = (pos // width) * width                   # the cut-off may only fall on a bucket boundary: an instance survives if the
    first = torch.clamp(seg_start[tile_of] + bstart, max=R - 1)   # bound in front of its BUCKET is still alive
    alive_b = (cs[first] - lg[first] - base[tile_of]) >= np.log2(1e-4)
    print(f"  ... with {nb:3d} buckets per tile                  {int(alive_b.sum()):10d}  = {100.0 * int(alive_b.sum()) / R:5.1f} % of R")

# ---- round 4: what a BUILDABLE version keeps.  Depth buckets of equal WIDTH in view depth between the frame's nearest and farthest
# binned instance (the same for every tile: any monotone map is conservative), and the per-instance mass quantised to a few fixed
# levels (an instance whose whole-tile alpha is >= a level counts with that level: conservative).  An instance survives if the
# bound accumulated over all buckets STRICTLY in front of its bucket is still >= 1e-4.
depth = (keys >> 32).int().view(torch.float32)        # depth bits are the key's high word
zmin, zmax = float(depth.min()), float(depth.max())
print(f"  depth of the binned instances: {zmin:.3f} .. {zmax:.3f}")
for levels in ((0.05,), (0.02, 0.1, 0.4), None):
    if levels is None:
        q = amin.double()
        name = "exact alpha_min"
    else:
        q = torch.zeros_like(amin, dtype=torch.float64)
        for lv in sorted(levels):
            q = torch.where(amin >= lv, torch.full_like(q, lv), q)
        name = "levels " + "/".join(str(v) for v in levels)
    lgq = torch.log2(1.0 - q)
    for nb in (4, 8, 16, 32):
        b = torch.clamp(((depth - zmin) / max(zmax - zmin, 1e-20) * nb).long(), 0, nb - 1)
        mass = torch.zeros(T * nb, device=dev, dtype=torch.float64).scatter_add_(0, tile_of * nb + b, lgq).view(T, nb)
        front = torch.cumsum(mass, 1) - mass          # log2 of the bound in front of each bucket
        alive_q = front[tile_of, b] >= np.log2(1e-4) + 0.5      # (half a binade of safety margin, like the kernel would keep)
        print(f"  {name:22s} {nb:3d} equal-width depth buckets: keep {int(alive_q.sum()):10d} = {100.0 * int(alive_q.sum()) / R:5.1f} % of R")
