#!/usr/bin/env python
"""Runs bench.train_iteration_row alone (prefilter -> decode -> rasterize -> RGB + depth loss -> backward -> training_statis),
for rocprofv3:  rocprofv3 --kernel-trace --stats -d OUT -o ti -- python tools/train_iteration_probe.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from gscream_amd import _native  # noqa: E402

_native.load()
row = bench.train_iteration_row(torch.device("cuda", 0))
from gscream_amd import rasterizer as RZ  # noqa: E402
row["raster_counts"] = dict(RZ._last_stage1)
print(json.dumps(row))
