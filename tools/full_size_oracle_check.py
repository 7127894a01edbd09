import os, sys, time, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import helpers as Hh
from gscream_amd import synthetic as S
from oracle import oracle as O
s = S.scene_slab(1, 1_000_000, 1008, 567)
grads = S.upstream_grads(1, 1008, 567, True, False, False)
nt = min(O.max_threads(), os.cpu_count() or 1, 64)
t = time.time(); st = Hh.oracle_forward(s, nthreads=nt); t1 = time.time() - t
t = time.time(); ref = Hh.oracle_backward(s, st, grads, nthreads=nt); t2 = time.time() - t
print("oracle fwd %.1fs bwd %.1fs threads %d R %d" % (t1, t2, nt, st["num_rendered"]))
t = time.time(); got = Hh.hip_run(s, grads); print("hip %.2fs" % (time.time() - t))
print("radii equal", (got["radii"] == st["radii"]).all())
for k in ("out_color", "out_depth", "out_unc"):
    d = np.abs(got[k] - st[k]); print(k, d.max(), (d > 1e-4).mean())
rep = {k: Hh.grad_report(got[k], ref[k], 1e-3) for k in Hh.GRAD_KEYS if k in ref and k in got}
for k, v in rep.items(): print(k, v)
