#!/usr/bin/env python
"""Gradient-slot counts of the per-Gaussian backward's 64-Gaussian groups on the bench workloads: how many groups exceed the
heavy threshold (GSR_K7_HEAVY_SLOTS), and how many slots they hold.  Used to choose when the heavy kernel is worth its launch.
   python tools/heavy_groups.py config4 train_iteration"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench as B  # noqa: E402
from gscream_amd import _layout, rasterizer  # noqa: E402

last = {}
_orig = rasterizer._forward_native


def _spy(*a, **k):
    out = _orig(*a, **k)
    last.update(R=int(out[0]), geom=out[5], P=int(a[0].shape[0]))
    return out


rasterizer._forward_native = _spy


def report(tag):
    P, R = last["P"], last["R"]
    off = _layout.geom_views(last["geom"], P)["offsets"][:P].cpu().numpy().astype(np.int64) & 0xffffffff
    starts = np.minimum(off[::64], R)
    ends = np.minimum(np.append(starts[1:], R), R)
    n = ends - starts
    line = [f"{tag}: P {P} R {R} groups {n.size} mean {n.mean():.0f} max {n.max()}"]
    for thr in (1024, 2048, 4096, 8192):
        sel = n > thr
        line.append(f">{thr}: {int(sel.sum())} groups, {int(n[sel].sum())} slots ({n[sel].sum() / max(R, 1):.3f} of R)")
    print("  ".join(line), flush=True)


dev = torch.device("cuda", 0)
for wl in sys.argv[1:] or ["config2", "config4", "train_iteration"]:
    if wl == "train_iteration":
        for occ in (False, True):
            rasterizer.set_tuning(occlusion_cut=occ)
            B.train_iteration_row(dev)
            report(f"train_iteration occlusion={occ}")
        rasterizer.set_tuning(occlusion_cut=None)
    else:
        P, W, H, seed, gsel, desc = B.WORKLOADS[wl]
        sb = B.SceneBench(dev, P, W, H, seed, seed, gsel)
        sb.step()
        torch.cuda.synchronize()
        report(wl)
