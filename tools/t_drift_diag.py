#!/usr/bin/env python
"""Which pixel's fast-walk final T drifts from the oracle's chain, and why?  (fuzz case c of seed 1000; GPU box)
usage: python tools/t_drift_diag.py [case]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as Hh
from gscream_amd import synthetic as S, set_tuning

c = int(sys.argv[1]) if len(sys.argv) > 1 else 56
seed0 = 1000
rng = np.random.default_rng(seed0 + c)
P = int(rng.choice([1, 7, 64, 65, 300, 1500, 4000, 12000]))
W, H = int(rng.integers(17, 700)), int(rng.integers(17, 500))
s = S.scene_config1(seed=seed0 + c, P=P, W=W, H=H)
mode = c % 4
if mode == 1: s["scales"] = (s["scales"] * np.float32(6.0)).astype(np.float32)
elif mode == 2: s["scales"] = (s["scales"] * np.float32(0.05)).astype(np.float32)
elif mode == 3: s["means3D"][:, 2] = np.round(s["means3D"][:, 2] * 2) / 2
view, proj, campos = S.camera_matrices(s["tanfovx"], s["tanfovy"], S.random_w2c(rng))
s["viewmatrix"], s["projmatrix"], s["campos"] = view, proj, campos
use = (True, bool(rng.integers(0, 2)), bool(rng.integers(0, 2)))
grads = S.upstream_grads(seed0 + c, W, H, *use)
st = Hh.oracle_forward(s, nthreads=16)
set_tuning(tile_cull=bool(c % 2))
got = Hh.hip_run(s, grads)
ft, rt = got["final_T"].astype(np.float64).reshape(-1), st["final_T"].astype(np.float64).reshape(-1)
ref_last = Hh.last_gaussian(st["ranges"], st["point_list"], st["n_contrib"], W, H).reshape(-1)
same = (got["last_gid"].reshape(-1) == ref_last) & (rt > 0)
rel = np.where(same, np.abs(ft - rt) / np.maximum(rt, 1e-300), 0)
order = np.argsort(-rel)[:5]
print("P", P, W, H, "mode", mode)
for p in order:
    y, x = divmod(int(p), W)
    print("pixel", x, y, "ours T", ft[p], "oracle T", rt[p], "rel", rel[p], "n_contrib oracle", int(np.asarray(st["n_contrib"]).reshape(-1)[p]), "color diff", float(np.abs(got["out_color"][:, y, x] - st["out_color"][:, y, x]).max()))
# the oracle's walk of the worst pixel: alpha of every blending instance
p = int(order[0]); y, x = divmod(p, W)
gx = (W + 15) // 16
tile = (y // 16) * gx + x // 16
r0, r1 = [int(v) for v in np.asarray(st["ranges"])[tile]]
ids = np.asarray(st["point_list"])[r0:r1]
m2 = np.asarray(st["means2D"]) if "means2D" in st else None
print("tile list length", r1 - r0, "keys in st:", sorted(st.keys())[:40])
