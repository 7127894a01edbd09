"""Runs bench.pipeline_row alone (decode -> rasterize -> loss -> backward) for a kernel-level profile (diagnostic)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
print(bench.pipeline_row(torch.device("cuda:0")))
