"""Diagnostic: distribution of gradient errors (HIP vs oracle) on a slab scene.  GPU box only."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as Hh
from gscream_amd import synthetic as S
P, W, H, seed = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (60000, 504, 284, 21)
s = S.scene_slab(seed, P, W, H)
grads = S.upstream_grads(seed, W, H)
nt = min(32, os.cpu_count())
st = Hh.oracle_forward(s, nthreads=nt); ref = Hh.oracle_backward(s, st, grads, nthreads=nt)
got = Hh.hip_run(s, grads)
print("lib", os.environ.get("GSR_LIB", "default"), "R", st["num_rendered"])
for k in ("out_color", "out_depth", "out_unc"):
    print(k, Hh.image_report(got[k], st[k]))
for k in Hh.GRAD_KEYS:
    r = np.asarray(ref[k], np.float64).reshape(got[k].shape); g = np.asarray(got[k], np.float64)
    rel = np.abs(g - r) / (np.abs(r) + 1e-3 * np.abs(r).max())
    print(f"{k:16s} max {rel.max():.2e}  p99.99 {np.quantile(rel, 0.9999):.2e}  p99.9 {np.quantile(rel, 0.999):.2e}  p99 {np.quantile(rel, 0.99):.2e}  n>1e-3: {(rel > 1e-3).sum()}")
