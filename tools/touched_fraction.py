#!/usr/bin/env python
"""How many Gaussians own a written gradient slot after a backward?  (per-Gaussian backward, compact path: gauss_bwd.hip; GPU box)
Prints, for a bench workload: P, visible, Gaussians with >= 1 written slot, written slots, the distribution of written slots per touched
Gaussian and of slot counts, and per 64-entry wave of the compact list the run length it gathers.
usage: python tools/touched_fraction.py [config2 | config3 | config4 | surfaces | init_state]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench as B  # noqa: E402
from gscream_amd import _layout  # noqa: E402


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "config2"
    P, W, H, seed, gsel, desc = B.WORKLOADS[wl]
    dev = torch.device("cuda", 0)
    sb = B.SceneBench(dev, P, W, H, seed, seed, gsel, wl)
    P = sb.P
    out = sb.rast(*sb.leaves[:1], sb.means2D, *sb.leaves[1:3], colors_precomp=sb.leaves[3], scales=sb.leaves[4], rotations=sb.leaves[5])
    color, depth, feat, radii = out
    fn = color.grad_fn
    geom, binning, img = fn.saved_tensors[-3:]
    R, cap = int(fn.num_rendered), int(fn.binning_capacity)
    loss = (color * sb.g[0]).sum()
    if gsel[1]:
        loss = loss + (depth * sb.g[1]).sum()
    if gsel[2]:
        loss = loss + (feat * sb.g[2]).sum()
    torch.autograd.grad(loss, sb.leaves)
    torch.cuda.synchronize()
    gv = _layout.geom_views(geom, P)
    bv = _layout.binning_views(binning, R, capacity=cap)
    tiles = gv["tiles"].long().cpu().numpy()
    offs = gv["offsets"].long().cpu().numpy()
    flags = bv["slot_written"].cpu().numpy().astype(np.int64)
    cs = np.concatenate([[0], np.cumsum(flags)])
    end = np.minimum(offs + tiles, R)
    offs = np.minimum(offs, R)
    c = cs[end] - cs[offs]
    touched = c > 0
    vis = (radii.cpu().numpy() > 0)
    q = lambda a: [int(x) for x in np.percentile(a, [50, 90, 99, 100])] if a.size else None
    # waves of the compact list: 64 touched Gaussians of one 1024-segment
    seg = np.arange(P) // 1024
    runs, cmaxs = [], []
    for s in range(0, P // 1024 + 1, max(1, (P // 1024) // 64)):
        ids = np.nonzero(touched & (seg == s))[0]
        for w in range(0, len(ids), 64):
            cc = c[ids[w:w + 64]]
            runs.append(int(cc.sum())); cmaxs.append(int(cc.max()))
    print(json.dumps({"workload": wl, "P": P, "visible": int(vis.sum()), "touched": int(touched.sum()), "touched_frac_of_P": round(float(touched.mean()), 4),
                      "R": R, "written_slots": int(flags.sum()), "slots_per_gaussian_p50_p90_p99_max": q(tiles[vis]),
                      "written_per_touched_p50_p90_p99_max": q(c[touched]), "wave_run_p50_p90_p99_max": q(np.array(runs)),
                      "wave_cmax_p50_p90_p99_max": q(np.array(cmaxs)), "waves_sampled": len(runs)}))


if __name__ == "__main__":
    main()
