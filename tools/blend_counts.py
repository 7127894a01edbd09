#!/usr/bin/env python
"""Iteration statistics of the blend kernels on a bench workload (diagnostic build with -DGSR_COUNT):
   make -C gscream_amd/csrc variant SRC=blend NAME=count FLAGS=-DGSR_COUNT ; GSR_LIB=gscream_amd/libgsraster_count.so python tools/blend_counts.py"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench as B  # noqa: E402
from gscream_amd import _native  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "config2"
P, W, H, seed, gsel, desc = B.WORKLOADS[wl]
dev = torch.device("cuda", 0)
lib = _native.load()
sb = B.SceneBench(dev, P, W, H, seed, seed, gsel, wl)
for _ in range(3):
    sb.step()
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 8)()
lib.gsr_debug_counters(buf, 1)
sb.step()
torch.cuda.synchronize()
lib.gsr_debug_counters(buf, 0)
it, bl, px, half, fit, fbl, fpx, band = [int(buf[i]) for i in range(8)]
print(f"{wl}: backward (instance, 16x8 strip) iterations {it}, blending {bl} ({bl / max(it, 1):.3f}), blending pixels per blending iteration "
      f"{px / max(bl, 1):.1f} of 128 ({px / max(bl, 1) / 128:.3f}), single-half iterations {half} ({half / max(bl, 1):.3f})")
print(f"{wl}: forward pair iterations {fit} (= {2 * fit} instance slots), blending instances {fbl} ({fbl / max(2 * fit, 1):.3f}), "
      f"blending pixels per blending instance {fpx / max(fbl, 1):.1f} of 64")
print(f"{wl}: guard-band re-checks (wave-level, forward + backward of one iteration) {band} = {band / max(2 * fit + it, 1):.2e} of the evaluated (wave, instance) slots")
