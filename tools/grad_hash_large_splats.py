#!/usr/bin/env python
"""sha256 of the rasterizer's gradients on a scene with large splats (heavy 64-Gaussian groups in the per-Gaussian backward):
two builds (GSR_LIB) that sum every Gaussian's slots in the same order must print the same hash."""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as Hh  # noqa: E402
from gscream_amd import synthetic as S  # noqa: E402

rng = np.random.default_rng(5)
s = S.scene_config1(seed=85, P=6000, W=640, H=400)
big = rng.choice(6000, size=600, replace=False)
s["scales"][big] = rng.uniform(0.5, 2.0, size=(600, 3)).astype(np.float32)
s["opacities"][big] = rng.uniform(0.02, 0.5, size=(600, 1)).astype(np.float32)
got = Hh.hip_run(s, S.upstream_grads(3, s["W"], s["H"]))
h = hashlib.sha256()
for k in Hh.GRAD_KEYS:
    if k in got:
        h.update(np.ascontiguousarray(got[k]).tobytes())
st = Hh.hip_run(s, keep_state=True)
print("R", st["num_rendered"], "gradients", h.hexdigest()[:24])
