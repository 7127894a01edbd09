#!/usr/bin/env python
"""Where do the backward blend's bytes go?  (run on the GPU box.)

Counts, from the forward state of a bench workload (tile ranges, tile_work, n_contrib, checkpoints passed), what every task of
gsr_blend_bwd_kernel requests per launch, by category -- requests, i.e. before any L2 hit:
  records        64 B per staged list entry (one line per instance; wave 0 takes {a, b}, wave 1 {c, d})
  list + offsets the 4-byte list entry (read by both waves) and offsets[id] (a 4-byte gather: one 32-byte sector per instance)
  pixel planes   final_T, n_contrib, dL/dcolor x 3 (+ dL/ddepth, dL/dfeature): 20 (28) B per pixel and TASK -- every depth segment
                 of a tile re-reads its 256 pixels' planes
  checkpoints    what behind() reads per pixel that blends behind the task's segment:
                   shipped   : {last slot, slot seg, every slot behind seg} = (np - seg + 1) x 16 B (+ 8 B with the aux maps)
                   suffix-sum experiment (tools/experiments/r06_suffix_sum_checkpoints.patch: the forward pays more than the backward
                   gains): ONE slot {T_e, suffix sums} = 16 B (+ 8 B)
  gradient slots 48 B + 1 flag byte per staged entry (written)
and prints them next to the PMC figure when profiles/pmc_latest.json has one for the workload.
usage: python tools/bwd_traffic_model.py [config2 | config3 | config4 | surfaces | init_state]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench as B  # noqa: E402
from gscream_amd import _layout  # noqa: E402
from gscream_amd import rasterizer as RZ  # noqa: E402


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "config2"
    P, W, H, seed, gsel, desc = B.WORKLOADS[wl]
    dev = torch.device("cuda", 0)
    sb = B.SceneBench(dev, P, W, H, seed, seed, gsel, wl)
    P = sb.P
    means3D, opac, unc, colors, scales, rots = [x.detach() for x in sb.leaves]
    e = torch.Tensor([])
    R, color, depth, feat, radii, geom, binning, img, ns = RZ._forward_native(means3D, e, colors, opac, unc, scales, rots, e, sb.rs)
    iv = _layout.image_views(img, P, W, H)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    T = gx * gy
    L = 64 if T <= 4096 else 128
    aux = bool(gsel[1] or gsel[2])
    S = _layout.SEG_MAX
    rng = iv["ranges"].long().cpu().numpy()
    n_list = rng[:, 1] - rng[:, 0]
    work = iv["tile_work"].long().cpu().numpy()
    nproc = np.minimum(n_list, work)
    pos = np.array([_layout.ckpt_pos(k, L, L) for k in range(S - 1)], np.int64)
    ncon = (iv["n_contrib"].long() & 0x3fffffff).cpu().numpy().reshape(H, W)
    N, Np = W * H, (W * H + 3) & ~3
    npass = iv["ckpt"][S - 1, :4 * Np].reshape(Np, 4)[:N, 0].contiguous().view(torch.int32).long().cpu().numpy().reshape(H, W)
    ys, xs = np.mgrid[0:H, 0:W]
    tile = (ys // 16) * gx + xs // 16
    staged = tasks = 0
    ck_old = ck_new = 0
    ck_unit = 16 + (8 if aux else 0)
    for seg in range(S):
        lo = 0 if seg == 0 else pos[seg - 1]
        hi = np.where(seg == S - 1, nproc, np.minimum(nproc, pos[seg] if seg < S - 1 else nproc))
        live = hi > lo                      # tiles that have this task
        tasks += int(live.sum())
        staged += int((hi - lo)[live].sum())
        # pixels of those tiles whose last contributor lies behind the segment's end: they start from checkpoints
        beh = live[tile] & (ncon > hi[tile]) & (seg < S - 1)
        n_beh = int(beh.sum())
        ck_new += n_beh * ck_unit
        ck_old += int(((npass[beh] - seg - 1).clip(min=0)).sum()) * ck_unit + n_beh * (ck_unit + 16)   # slots behind + last slot + slot seg (.x only: one 16-B piece)
    planes = 20 + (8 if aux else 0)
    cat = {"records": staged * 64, "list_entries_and_offsets": staged * (2 * 4 + 32), "pixel_planes": tasks * 256 * planes,
           "checkpoints": ck_old, "checkpoints_suffix_sum_experiment": ck_new, "gradient_slots_written": staged * 49}
    tot_old = sum(v for k, v in cat.items() if k != "checkpoints_suffix_sum_experiment")
    tot_new = sum(v for k, v in cat.items() if k != "checkpoints")
    out = {"workload": wl, "tiles": int(T), "tasks": int(tasks), "staged_list_entries": int(staged), "num_rendered": int(R),
           "requested_MB_per_launch": {k: round(v / 1e6, 1) for k, v in cat.items()},
           "total_MB": round(tot_old / 1e6, 1), "total_MB_suffix_sum_experiment": round(tot_new / 1e6, 1),
           "note": "bytes REQUESTED by the launch's tasks (every gather counted at its sector / line size), not HBM traffic: neighbouring tasks "
                   "share records and the four tasks of a tile share its pixel planes through the XCD's L2"}
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json"))).get(wl, {})
        if "blend_backward" in pmc:
            out["pmc_moved_MB_blend_backward"] = round(pmc["blend_backward"] / 1e6, 1)
            out["pmc_provenance"] = pmc.get("_provenance", {}).get("kernel_source_sha256", "")[:12]
    except Exception:  # noqa: BLE001
        pass
    print(json.dumps(out))


if __name__ == "__main__":
    main()
