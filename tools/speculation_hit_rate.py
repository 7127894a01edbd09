"""How often does the speculative forward (workspace + sort provision guessed from recent frames) hold when the camera
changes every iteration, as in training?  Diagnostic; run on the GPU box."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gscream_amd import GaussianRasterizationSettings, GaussianRasterizer, set_tuning, synthetic as S
from gscream_amd import rasterizer as RZ

P, W, H = 300_000, 756, 425
s = S.scene_slab(5, P, W, H)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
m3, op, un, col, sc, ro = (t(s[k]) for k in ("means3D", "opacities", "uncertainties", "colors", "scales", "rotations"))
rng = np.random.default_rng(0)
set_tuning()
hits = total = 0
Rs = []
for it in range(60):
    view, proj, campos = S.camera_matrices(s["tanfovx"], s["tanfovy"], S.random_w2c(rng, max_angle=0.25, max_shift=0.5))
    rs = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=s["tanfovx"], tanfovy=s["tanfovy"], bg=t(s["bg"]),
                                       scale_modifier=1.0, viewmatrix=t(view), projmatrix=t(proj), sh_degree=1, campos=t(campos),
                                       prefiltered=False, debug=False)
    GaussianRasterizer(rs)(m3, torch.zeros_like(m3), op, un, colors_precomp=col, scales=sc, rotations=ro)
    if it >= 1:
        total += 1
        hits += int(RZ._last_stage1["speculative"])
    Rs.append(RZ._last_stage1["num_rendered"])
print(f"speculative forward held in {hits}/{total} iterations; num_rendered min/max {min(Rs)}/{max(Rs)}")
