import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as Hh
from gscream_amd import synthetic as S, _layout
s = S.scene_slab(1, 1_000_000, 1008, 567)
got = Hh.hip_run(s, keep_state=True)
iv = _layout.image_views(got["img"], 1_000_000, 1008, 567)
r = iv["ranges"].cpu().numpy().astype(np.int64)
np.save(os.path.join(ROOT, "gpurun_out", "tile_counts_config2.npy"), r[:, 1] - r[:, 0])
nc = iv["n_contrib"].cpu().numpy()
np.save(os.path.join(ROOT, "gpurun_out", "n_contrib_config2.npy"), nc.astype(np.int32))
print("saved", (r[:,1]-r[:,0]).sum())
gv = _layout.geom_views(got["geom"], 1_000_000)
t = gv["tiles"].cpu().numpy().astype(np.int64)
np.save(os.path.join(ROOT, "gpurun_out", "gauss_tiles_config2.npy"), t.astype(np.int32))
w = t.reshape(-1, 64).sum(1)
print("per-Gaussian slots: max", t.max(), "p99.9", np.percentile(t, 99.9), "; per-wave range: mean", w.mean(), "p50", np.median(w), "p99", np.percentile(w, 99), "max", w.max())
