#!/usr/bin/env python
"""Which path do the per-tile sorts take?  Replays gsr_sort_buckets' decisions (binning.hip) on the keys the scatter left in seg_keys:
per tile the list length, the keys in buckets beyond GSR_BUCKET_MAX, and whether the tile falls back to the compare-exchange network.
   python tools/sort_buckets_probe.py config2 config4 train_iteration"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench as B  # noqa: E402
from gscream_amd import _layout, rasterizer  # noqa: E402

NEAR_CAP, BUCKET_MAX, HEAVY_MAX = 2048, 16, 16
last = {}
_orig = rasterizer._forward_native


def _spy(*a, **k):
    out = _orig(*a, **k)
    rs = a[8]
    last.update(R=int(out[0]), binning=out[6], img=out[7], cap=int(out[8]), P=int(a[0].shape[0]), W=int(rs.image_width), H=int(rs.image_height))
    return out


rasterizer._forward_native = _spy


def bucket_stats(keys, nb_log=10):
    n = keys.size
    kmin, kmax = keys.min(), keys.max()
    span = int(kmax - kmin)
    shift = 0 if span < (1 << nb_log) else span.bit_length() - nb_log
    b = ((keys - kmin) >> np.uint64(shift)).astype(np.int64)
    cnt = np.bincount(b, minlength=1 << nb_log)
    heavy = cnt > BUCKET_MAX
    hk, hn = int(cnt[heavy].sum()), int(heavy.sum())
    fallback = 2 * hk > n or hn > HEAVY_MAX
    # insertion-sort work of the light buckets ~ sum c (c - 1) / 4; network work of the heavy slices ~ c log2(c)^2 / 4
    return hk, hn, fallback, int((cnt[~heavy] * (cnt[~heavy] - 1)).sum() // 4), cnt


def report(tag):
    R, W, H, P = last["R"], last["W"], last["H"], last["P"]
    torch.cuda.synchronize()
    iv = _layout.image_views(last["img"], P, W, H)
    ranges = iv["ranges"].cpu().numpy().astype(np.int64).reshape(-1, 2)
    keys = _layout.binning_views(last["binning"], R, last["cap"])["seg_keys"].cpu().numpy().view(np.uint64)
    T = ranges.shape[0]
    rows = dict(short=0, short_fallback=0, short_heavy_keys=0, short_keys=0, long=0, long_keys=0, near_keys=0, near_fallback=0, near_heavy_keys=0)
    maxb = []
    for t in range(T):
        lo, hi = ranges[t]
        n = hi - lo
        if n <= 1:
            continue
        k = keys[lo:hi]
        if n <= NEAR_CAP:
            hk, hn, fb, _, cnt = bucket_stats(k)
            rows["short"] += 1; rows["short_keys"] += n; rows["short_fallback"] += fb; rows["short_heavy_keys"] += hk
            maxb.append(cnt.max())
        else:
            rows["long"] += 1; rows["long_keys"] += n
            kmin, kmax = k.min(), k.max()
            span = int(kmax - kmin)
            shift = 0 if span < 1024 else span.bit_length() - 10
            b = ((k - kmin) >> np.uint64(shift)).astype(np.int64)
            cum = np.cumsum(np.bincount(b, minlength=1024))
            ok = np.nonzero(cum <= NEAR_CAP)[0]
            m = int(cum[ok[-1]]) if ok.size else 0
            rows["near_keys"] += m
            if m > 1:
                near = k[b <= ok[-1]]
                hk, hn, fb, _, cnt = bucket_stats(near)
                rows["near_fallback"] += fb; rows["near_heavy_keys"] += hk
    mb = np.array(maxb) if maxb else np.zeros(1)
    print(f"{tag}: R {R} tiles {T}  short lists {rows['short']} ({rows['short_keys']} keys; {rows['short_fallback']} fall back to the network, "
          f"{rows['short_heavy_keys']} keys in heavy buckets; largest bucket median {np.median(mb):.0f} p90 {np.percentile(mb, 90):.0f} max {mb.max()})  "
          f"long lists {rows['long']} ({rows['long_keys']} keys, near sets {rows['near_keys']} keys; {rows['near_fallback']} near sets fall back, "
          f"{rows['near_heavy_keys']} keys in heavy buckets)", flush=True)


dev = torch.device("cuda", 0)
for wl in sys.argv[1:] or ["config2", "config4", "train_iteration"]:
    if wl == "train_iteration":
        for occ in (False, True):
            rasterizer.set_tuning(occlusion_cut=occ)
            B.train_iteration_row(dev)
            report(f"train_iteration occlusion={occ}")
        rasterizer.set_tuning(occlusion_cut=None)
    else:
        P, W, H, seed, gsel, desc = B.WORKLOADS[wl]
        sb = B.SceneBench(dev, P, W, H, seed, seed, gsel, wl)
        sb.step()
        report(wl)
