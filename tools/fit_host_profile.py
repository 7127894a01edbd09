#!/usr/bin/env python
"""Where does the host time of an optimisation iteration go?  (gscream_amd/fit.py; GPU box)  cProfile over `iters` iterations + GPU kernel time.
usage: python tools/fit_host_profile.py [iters] [adam: foreach | fused | single]"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gscream_amd import fit as F  # noqa: E402
from gscream_amd import simple_knn as KN, standin_model as SM, synthetic as S  # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    W, H, dev = 1008, 567, "cuda"
    ts = F.teacher_scene(101, 600_000, W, H, 0.6, dev)
    cams = F.orbit_cameras(16, W, H, 0.6, ts["means3D"].astype(np.float64).mean(0), device=dev)
    gts, gd = F.render_teacher(ts, cams, dev)
    pts = SM.voxelize(S.surface_point_cloud(1, 200_000, 0.6, H / W), 0.001)
    anchors = torch.from_numpy(pts).float().to(dev)
    model = SM.Model.from_pcd(anchors, torch.clamp_min(KN.distCUDA2(anchors), 1e-7), K=10, seed=1).to(dev)
    F.fit(model, cams, gts, gd, 50)  # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    info = F.fit(model, cams, gts, gd, iters)
    torch.cuda.synchronize()
    print("wall ms per iteration", (time.perf_counter() - t0) * 1e3 / iters, "event ms", info["ms_per_iteration"])
    pr = cProfile.Profile()
    pr.enable()
    F.fit(model, cams, gts, gd, iters)
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(30)


if __name__ == "__main__":
    main()
