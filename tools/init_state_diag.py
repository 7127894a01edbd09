#!/usr/bin/env python
"""Where do the init-state frame's parity outliers come from?  Runs helpers.parity_report on bench.py's `init_state` scene under
tuning-knob combinations (tile culling, partial sort) and prints, per combination, the outlier counts, the outlier Gaussians with
their footprint (radius, rectangle tiles, opacity, depth) and the outlier pixels with their tile's list length / walk depth.
(Test infrastructure: uses the oracle as the checker.)  usage (GPU box): python tools/init_state_diag.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as Hh  # noqa: E402
from gscream_amd import set_tuning, synthetic as S  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    W, H = 1008, 567
    s = S.scene_init_state(1, W, H)
    grads = S.upstream_grads(1, W, H, True, True, False)
    nt = max(1, min(O.max_threads(), os.cpu_count() or 1, 64))
    st = Hh.oracle_forward(s, nthreads=nt)
    ref = Hh.oracle_backward(s, st, grads, nthreads=nt)
    P = s["means3D"].shape[0]
    gx = (W + 15) // 16
    ll = st["ranges"][:, 1].astype(np.int64) - st["ranges"][:, 0]
    for name, kw in (("default", {}), ("no partial sort", dict(partial_sort=False)), ("no tile cull", dict(tile_cull=False)),
                     ("neither", dict(partial_sort=False, tile_cull=False))):
        set_tuning(**kw)
        got = Hh.hip_run(s, grads)
        rep = Hh.parity_report(got, st, ref, nthreads=nt)
        print(f"== {name}: px>1e-4 {rep['px_gt_1e-4']} (max {rep['max_abs']:.2e}), grad elems>1e-3 {rep['grad_elems_gt_1e-3']} worst {rep['worst_rel']:.3e} "
              f"{rep['grad_elems_by_cause']} stops {rep.get('last_contributor_differs')} radii equal {bool((got['radii'] == st['radii']).all())}")
        rows = np.zeros(P, bool)
        for k in Hh.GRAD_KEYS:
            if k in ref and k in got and np.asarray(ref[k]).size:
                g64, r64 = np.asarray(got[k], np.float64), np.asarray(ref[k], np.float64).reshape(np.asarray(got[k]).shape)
                r = np.abs(g64 - r64) / (np.abs(r64) + 1e-3 * max(np.abs(r64).max(), 1e-30))
                rows |= (r > 1e-3).reshape(P, -1).any(axis=1)
        ids = np.nonzero(rows)[0]
        print("   outlier Gaussians:", len(ids))
        for g in ids[:40]:
            print(f"     id {g}: radius {st['radii'][g]}, rect tiles {st['tiles_touched'][g]}, opacity {float(s['opacities'][g]):.4f}, depth {float(st['depths'][g]):.3f}, "
                  f"mean2D ({st['means2D'][g][0]:.1f}, {st['means2D'][g][1]:.1f}), scales {s['scales'][g]}")
        for k in ("out_color", "out_depth", "out_unc"):
            d = np.abs(got[k].astype(np.float64) - st[k].astype(np.float64)).reshape(-1, H, W).max(axis=0)
            for y, x in zip(*np.nonzero(d > 1e-4)):
                t = (y // 16) * gx + x // 16
                print(f"     {k} pixel ({x}, {y}) diff {d[y, x]:.3e}: tile {t} list {ll[t]}, oracle n_contrib {st['n_contrib'][y, x]}, final_T {st['final_T'][y, x]:.4e} ours {got['final_T'][y, x]:.4e}, "
                      f"last gid oracle {Hh.last_gaussian(st['ranges'], st['point_list'], st['n_contrib'], W, H)[y, x]} ours {got['last_gid'][y, x]}")
    set_tuning()


if __name__ == "__main__":
    main()
