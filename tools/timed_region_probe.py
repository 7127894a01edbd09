#!/usr/bin/env python
"""What moves bench.py's timed region?  K-step blocks of the bench step loop (config 2), timed like the official region, under
different conditions: no stage timing, HIP-event brackets around the dominant kernel in every step / every 4th step / all stages,
and after an idle gap.  usage (GPU box): python tools/timed_region_probe.py [K]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import bench as B  # noqa: E402
from gscream_amd import _native  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 50
P, W, H, seed, gsel, _ = B.WORKLOADS["config2"]
dev = torch.device("cuda", 0)
_native.load()
sb = B.SceneBench(dev, P, W, H, seed, seed, gsel)
for _ in range(20):
    sb.step()
torch.cuda.synchronize()


def block(label, before=None, after=None):
    if before:
        before()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        sb.step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / K * 1e3
    if after:
        after()
    print(f"{label:48s} {ms:.4f} ms/step", flush=True)


for rep in range(2):
    block("plain")
    block("brackets, dominant kernel, every step", lambda: _native.profile_begin(["blend_backward"]), _native.profile_end)
    block("brackets, dominant kernel, every 4th step", lambda: _native.profile_begin(["blend_backward"], every=4), _native.profile_end)
    block("brackets, all stages, every step", lambda: _native.profile_begin(), _native.profile_end)
    block("plain")
    block("plain after 0.2 s idle", lambda: time.sleep(0.2))
    block("plain after 3 profiled steps + sync (bench order)", lambda: (_native.profile_begin(), [sb.step() for _ in range(3)], torch.cuda.synchronize(), _native.profile_end()))
    block("plain")
