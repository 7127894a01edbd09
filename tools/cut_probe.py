#!/usr/bin/env python
"""What is "not binning what the previous visit of a view did not walk" worth?  (GPU box, EXPERIMENT build:
   GSR_LIB=gscream_amd/libgsraster_cut.so GSR_SKIP_ABI_CHECK=1 python tools/cut_probe.py [workload] -- built with -DGSR_CUT_EXPERIMENT, which
   lets the caller supply the per-tile cut-off table that the occlusion cut-off machinery acts on.)

Visit 1 of a view (normal forward): per tile the depth BUCKET (16 per octave of view depth, gsr_occ_bucket) of the deepest instance any of
its pixels blended; tiles whose walk reached the end of their list get no cut.  Visit 2 bins only instances up to that bucket + `margin`
buckets; everything downstream sees the smaller footprint.  Reported per margin: num_rendered, the front-end stage times, whether images /
radii / gradients came out bit-identical (they must whenever no walk runs past its cut -- a walk that does would need the redo this
experiment does not have: counted as a MISS by comparing the outputs), with the Gaussians static and perturbed by an optimiser-sized step
between the visits (bench.py SceneBench.make_rotation's perturbation)."""
import ctypes
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench as B  # noqa: E402
from gscream_amd import _layout, _native, set_tuning  # noqa: E402
from gscream_amd import rasterizer as RZ  # noqa: E402

NB = 160  # GSR_OCC_BUCKETS: "no cut"


def buckets(depth_bits):
    b = (depth_bits.long() >> 19) - ((127 - 3) << 4)
    return b.clamp(0, NB - 1)


def visit_state(sb):
    """forward state of the current scene: per-tile cut bucket (no margin), num_rendered, walked fraction"""
    means3D, opac, unc, colors, scales, rots = [x.detach() for x in sb.leaves]
    e = torch.Tensor([])
    R, color, depth, feat, radii, geom, binning, img, ns = RZ._forward_native(means3D, e, colors, opac, unc, scales, rots, e, sb.rs)
    iv, gv, bv = _layout.image_views(img, sb.P, sb.W, sb.H), _layout.geom_views(geom, sb.P), _layout.binning_views(binning, R, ns)
    rng = iv["ranges"].long()
    n = rng[:, 1] - rng[:, 0]
    work = iv["tile_work"].long().clamp(min=0)
    walked = torch.minimum(work, n)
    last = (rng[:, 0] + walked - 1).clamp(min=0)
    dk = gv["depthkey"][bv["point_list"].long()[last]]
    cut = buckets(dk)
    cut = torch.where((walked >= n) | (walked == 0), torch.full_like(cut, NB), cut)  # walked to the end (or empty): no cut
    return cut, int(R), float(walked.sum()) / max(int(n.sum()), 1)


def run_steps(sb, n=30):
    for _ in range(5):
        sb.step()
    _native.profile_begin()
    for _ in range(n):
        sb.step()
    torch.cuda.synchronize()
    pr = _native.profile_end()
    return {k: round(v[0] / max(v[1], 1) * 1e3, 1) for k, v in pr.items() if v[1]}


def outputs(sb):
    means3D, opac, unc, colors, scales, rots = sb.leaves
    color, depth, feat, radii = sb.rast(means3D, sb.means2D, opac, unc, colors_precomp=colors, scales=scales, rotations=rots)
    outs = [o for o, use in zip((color, depth, feat), sb.gsel) if use]
    gos = [g for g, use in zip(sb.g, sb.gsel) if use]
    grads = torch.autograd.grad(outs, sb.inputs, gos)
    return [color.detach().clone(), depth.detach().clone(), radii.clone()] + [g.clone() for g in grads]


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "config2"
    lib = _native.load()
    assert hasattr(lib, "gsr_debug_set_cut"), "needs the -DGSR_CUT_EXPERIMENT build (GSR_LIB=gscream_amd/libgsraster_cut.so)"
    lib.gsr_debug_set_cut.argtypes = [ctypes.c_void_p]
    lib.gsr_debug_set_cut.restype = None
    P, W, H, seed, gsel, desc = B.WORKLOADS[wl]
    dev = torch.device("cuda", 0)
    set_tuning(view_cache=False, occlusion_cut=False)
    sb = B.SceneBench(dev, P, W, H, seed, seed, gsel, wl)
    res = {"workload": wl}
    lib.gsr_debug_set_cut(None)
    cut0, R0, walked = visit_state(sb)
    res["first_visit"] = {"num_rendered": R0, "walked_fraction": round(walked, 4), "tiles_with_a_cut": int((cut0 < NB).sum()), "tiles": int(cut0.numel()),
                          "stages_us": run_steps(sb)}
    ref = outputs(sb)
    keep = [x.detach().clone() for x in sb.leaves[:2]]
    g = torch.Generator(device=dev).manual_seed(7)
    noise = (torch.randn(sb.leaves[0].shape, device=dev, generator=g), torch.randn(sb.leaves[1].shape, device=dev, generator=g))
    res["revisits"] = []
    for margin in (0, 1, 2, 4):
        cutm = torch.where(cut0 < NB, (cut0 + margin).clamp(max=NB), cut0).to(torch.int32).contiguous()
        row = {"margin_buckets": margin, "margin_depth_rel": round(2 ** (margin / 16) - 1, 3)}
        # (a) static Gaussians
        lib.gsr_debug_set_cut(ctypes.c_void_p(cutm.data_ptr()))
        got = outputs(sb)
        row["num_rendered"] = int(RZ._last_stage1["num_rendered"])
        row["identical_static"] = all(torch.equal(a, b) for a, b in zip(ref, got))
        row["stages_us"] = run_steps(sb)
        # (b) Gaussians moved by k optimiser-sized steps between the two visits (k = 64: one epoch of a 64-view rotation)
        for k in (1, 64):
            with torch.no_grad():
                sb.leaves[0].copy_(keep[0]).add_(noise[0], alpha=2e-4 * k ** 0.5)
                sb.leaves[1].copy_(keep[1]).add_(noise[1], alpha=1e-3 * k ** 0.5).clamp_(0.0, 1.0)
            lib.gsr_debug_set_cut(None)
            want = outputs(sb)
            lib.gsr_debug_set_cut(ctypes.c_void_p(cutm.data_ptr()))
            got = outputs(sb)
            same = all(torch.equal(a, b) for a, b in zip(want, got))
            bad_px = int(((want[0] != got[0]).any(0)).sum())
            row[f"moved_{k}_steps"] = {"identical": same, "pixels_that_differ": bad_px, "num_rendered": int(RZ._last_stage1["num_rendered"])}
        with torch.no_grad():
            sb.leaves[0].copy_(keep[0]); sb.leaves[1].copy_(keep[1])
        res["revisits"].append(row)
    lib.gsr_debug_set_cut(None)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
