#!/usr/bin/env python
"""sha256 of the forward's outputs and kept state (images, radii, final T, n_contrib, checkpoints reproduce through the
backward's gradients) on a few scenes -- to show that two builds (GSR_LIB) give the same BITS.
usage: GSR_LIB=... python tools/fwd_hash.py"""
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as Hh  # noqa: E402
from gscream_amd import _layout, synthetic as S  # noqa: E402

scenes = {"config2": S.scene_slab(1, 1_000_000, 1008, 567), "slab60k": S.scene_slab(21, 60_000, 504, 284),
          "stack": S.scene_stack(), "cfg1": S.scene_config1(), "config4": S.scene_slab(3, 2_000_000, 1920, 1080)}
for name, s in scenes.items():
    grads = S.upstream_grads(7, s["W"], s["H"])
    got = Hh.hip_run(s, grads)
    st = Hh.hip_run(s, keep_state=True)
    iv = _layout.image_views(st["img"], s["means3D"].shape[0], s["W"], s["H"])
    h = hashlib.sha256()
    for k in ("out_color", "out_depth", "out_unc", "radii"):
        h.update(np.ascontiguousarray(got[k]).tobytes())
    h.update(iv["final_T"].cpu().numpy().tobytes())
    h.update(iv["n_contrib"].cpu().numpy().tobytes())
    h.update(iv["tile_work"].cpu().numpy().tobytes())
    g = hashlib.sha256()
    for k in Hh.GRAD_KEYS:
        if k in got:
            g.update(np.ascontiguousarray(got[k]).tobytes())
    print(f"{name:8s} forward {h.hexdigest()[:16]}  gradients {g.hexdigest()[:16]}  max n_contrib {int(iv['n_contrib'].max())}")
