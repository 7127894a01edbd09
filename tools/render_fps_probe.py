#!/usr/bin/env python
"""Why is the render_fps row 3x slower on the full bench line (CPU baselines on) than with --no-cpu-baseline?  Times the inference
forward loop (GPU events + host clock per call) before and after each CPU-baseline row."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench as B  # noqa: E402

dev = torch.device("cuda", 0)
P, W, H, seed, gsel, desc = B.WORKLOADS["config2"]
sb = B.SceneBench(dev, P, W, H, seed, seed, gsel)
means3D, opac, unc, colors, scales, rots = [t.detach() for t in sb.leaves]
m2d = torch.zeros_like(means3D)


def raster_eval():
    with torch.no_grad():
        return sb.rast(means3D, m2d, opac, unc, colors_precomp=colors, scales=scales, rotations=rots)


import gc
_gc = {"t0": 0.0, "total": 0.0, "n": 0, "gen2": 0}


def _gc_cb(phase, info):
    if phase == "start":
        _gc["t0"] = time.perf_counter()
    else:
        _gc["total"] += time.perf_counter() - _gc["t0"]; _gc["n"] += 1; _gc["gen2"] += info["generation"] == 2


gc.callbacks.append(_gc_cb)


def probe(tag, n=200, collect=False):
    if collect:
        gc.collect()
    _gc.update(total=0.0, n=0, gen2=0)
    for _ in range(20):
        raster_eval()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    host = []
    e0.record()
    for _ in range(n):
        t0 = time.perf_counter()
        raster_eval()
        host.append(time.perf_counter() - t0)
    e1.record()
    torch.cuda.synchronize()
    host.sort()
    print(f"{tag}: gpu {e0.elapsed_time(e1) / n:.4f} ms/frame; host per call median {host[n // 2] * 1e3:.4f} p90 {host[int(n * .9)] * 1e3:.4f} max {host[-1] * 1e3:.4f} ms; "
          f"gc inside the loop: {_gc['n']} collections ({_gc['gen2']} full) {_gc['total'] * 1e3:.2f} ms; "
          f"torch threads {torch.get_num_threads()} alloc {torch.cuda.memory_allocated() >> 20} MiB reserved {torch.cuda.memory_reserved() >> 20} MiB", flush=True)


probe("fresh")
for name, fn in (("loss_row cpu", lambda: B.loss_row(dev, H, W, True)), ("depth_loss_row cpu", lambda: B.depth_loss_row(dev, H, W, True)),
                 ("decode_row cpu", lambda: B.decode_row(dev, True)), ("pipeline_row", lambda: B.pipeline_row(dev)),
                 ("train_iteration_row cpu", lambda: B.train_iteration_row(dev, with_cpu=True))):
    fn()
    probe("after " + name)
    fn()
    probe("after " + name + " + gc.collect()", collect=True)
gc.collect(); torch.cuda.empty_cache()
probe("after gc + empty_cache")
torch.set_num_threads(1)
probe("torch threads = 1")
