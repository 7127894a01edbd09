#!/usr/bin/env python
"""Element-wise parity of the HIP path (whatever library GSR_LIB points to) against the OpenMP oracle at BASELINE's full
sizes.  One JSON object on stdout: radii equality, per image the number of pixels beyond 1e-4 and the largest difference,
per gradient family the number of elements beyond 1e-3 relative (denominator |ref| + 1e-3 max|ref|), the 99.9th
percentile and the worst element.  tests/test_gpu_fullsize.py runs it once per library (shipped and parity build).
(Test infrastructure: uses the oracle as the checker.)
usage: python tools/full_size_oracle_check.py <seed> <P> <W> <H> <color 0/1> <depth 0/1> <feature 0/1>"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as Hh  # noqa: E402
from gscream_amd import _native, synthetic as S  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    seed, P, W, H = (int(a) for a in sys.argv[1:5])
    use = tuple(bool(int(a)) for a in sys.argv[5:8]) if len(sys.argv) >= 8 else (True, True, True)
    s = S.scene_slab(seed, P, W, H)
    grads = S.upstream_grads(seed, W, H, *use)
    nt = max(1, min(O.max_threads(), os.cpu_count() or 1, 64))
    st = Hh.oracle_forward(s, nthreads=nt)
    ref = Hh.oracle_backward(s, st, grads, nthreads=nt)
    got = Hh.hip_run(s, grads)
    out = {"lib": os.path.basename(_native.LIB_PATH), "P": P, "W": W, "H": H, "num_rendered_oracle": int(st["num_rendered"]),
           "radii_equal": bool((got["radii"] == st["radii"]).all()), "images": {}, "grads": {}}
    for k in ("out_color", "out_depth", "out_unc"):
        d = np.abs(got[k].astype(np.float64) - st[k].astype(np.float64))
        out["images"][k] = {"n": int(d.size), "gt_1e-4": int((d > 1e-4).sum()), "max": float(d.max())}
    for k in Hh.GRAD_KEYS:
        if k in ref and k in got:
            out["grads"][k] = Hh.grad_report(got[k], ref[k], 1e-3)
    out["parity_check"] = Hh.parity_report(got, st, ref, nthreads=nt, s=s, grads=grads)  # + every outlier classified (alpha = 1/255 vs T = 1e-4 flips)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
