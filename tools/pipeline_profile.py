#!/usr/bin/env python
"""Runs bench.pipeline_row alone (decode -> rasterize -> loss -> backward), for rocprofv3:
   rocprofv3 --kernel-trace --stats -d OUT -o pl -- python tools/pipeline_profile.py ; python tools/rocprof_summary.py OUT/.../pl_results.db"""
import sys
import torch
sys.path.insert(0, ".")
import bench
from gscream_amd import _native
_native.load()
print(bench.pipeline_row(torch.device("cuda", 0)))
