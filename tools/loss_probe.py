#!/usr/bin/env python
"""bench.loss_row + bench.depth_loss_row alone (the 8(f) loss rows), for A/B and rocprofv3."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from gscream_amd import _native  # noqa: E402

_native.load()
dev = torch.device("cuda", 0)
for _ in range(2):
    r = bench.loss_row(dev, 567, 1008, False)
    d = bench.depth_loss_row(dev, 567, 1008, False)
    print(json.dumps({"rgb_loss_ms": r["ms"], "rgb_autograd_ms": r["ms_through_autograd_api"], "depth_loss_ms": d["ms"]}))
