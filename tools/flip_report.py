#!/usr/bin/env python
"""Counts the elements by which the HIP path (whatever library GSR_LIB points to) misses the oracle: pixels beyond 1e-4,
gradient elements beyond 1e-3 relative (denominator |ref| + 1e-3 max|ref|), per case, on the committed goldens + the
60k-Gaussian slab scene + config 1.  One JSON object on stdout.  tests/test_gpu_precise.py runs it once with the shipped
library and once with the parity build (libgsraster_precise.so) to attribute the shipped build's outliers to
alpha = 1/255 / T = 1e-4 threshold flips.  (Test infrastructure: uses the oracle as the checker.)"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as Hh  # noqa: E402
from golden import make_golden as MG  # noqa: E402
from gscream_amd import synthetic as S  # noqa: E402


def report(got, img_ref, grad_ref):
    out = {"pixels_gt_1e-4": 0, "pixels": 0, "img_max": 0.0, "grad_gt_1e-3": 0, "grad_elems": 0, "grad_max_rel": 0.0}
    for k in ("out_color", "out_depth", "out_unc"):
        r = Hh.image_report(got[k], img_ref[k])
        out["pixels_gt_1e-4"] += r["outliers"]; out["pixels"] += r["n"]; out["img_max"] = max(out["img_max"], r["max_abs"])
    for k in list(Hh.GRAD_KEYS) + ["dL_dsh", "dL_dcov3D"]:
        if k in got and k in grad_ref:
            r = Hh.grad_report(got[k], grad_ref[k])
            out["grad_gt_1e-3"] += r["n_bad"]; out["grad_elems"] += r["n"]; out["grad_max_rel"] = max(out["grad_max_rel"], r["max"])
    out["radii_equal"] = bool((got["radii"] == img_ref["radii"]).all())
    return out


def main():
    res = {}
    for name in sorted(MG.cases()):
        s, grads, exp = MG.load(name)
        got = Hh.hip_run(s, grads)
        res[name] = report(got, exp, {k[5:]: exp[k] for k in exp if k.startswith("grad_")})
    extra = {"config1": (S.scene_config1(), None), "slab60k": (S.scene_slab(21, 60_000, 504, 284), 21)}
    for name, (s, seed) in extra.items():
        grads = S.upstream_grads(seed if seed is not None else 1, s["W"], s["H"])
        nt = max(1, min(16, os.cpu_count() or 1))
        st = Hh.oracle_forward(s, nthreads=nt)
        ref = Hh.oracle_backward(s, st, grads, nthreads=nt)
        res[name] = report(Hh.hip_run(s, grads), st, ref)
    from gscream_amd import _native
    print(json.dumps({"lib": os.path.basename(_native.LIB_PATH), "cases": res}))


if __name__ == "__main__":
    main()
