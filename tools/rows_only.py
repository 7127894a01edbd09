#!/usr/bin/env python
"""Side rows of bench.py alone (for A/B of library variants through GSR_LIB): python tools/rows_only.py decode loss depth"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench as B  # noqa: E402

dev = torch.device("cuda", 0)
tag = os.path.basename(os.environ.get("GSR_LIB", "libgsraster.so"))
for what in sys.argv[1:]:
    if what == "decode":
        r = B.decode_row(dev, False)
        print(tag, "decode native", r["native"]["forward_ms"], r["native"]["forward_backward_ms"], "autograd", r["forward_ms"], r["forward_backward_ms"], flush=True)
    elif what == "loss":
        r = B.loss_row(dev, 567, 1008, False)
        print(tag, "rgb_loss", r["ms"], flush=True)
    elif what == "depth":
        r = B.depth_loss_row(dev, 567, 1008, False)
        print(tag, "depth_loss", r["ms"], flush=True)
