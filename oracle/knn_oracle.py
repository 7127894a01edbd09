"""CPU oracle for simple_knn.distCUDA2.  TEST INFRASTRUCTURE ONLY.

The reference (submodules/simple-knn/simple_knn.cu:134-183) returns, for every point, the mean of the three smallest
squared distances to the other points; its Morton boxes only prune, the search is exact.  An exact 3-nearest-
neighbour query pins that result independently of any implementation detail: scipy.spatial.cKDTree (float64), self
match removed.  Coincident points count as neighbours at distance 0 (the reference skips only the point's own index).
Fewer than four points: the reference leaves FLT_MAX terms in the average -- restated in `mean_dist2` below.

PARITY: pinned by the definition (exact k-NN), not by reference-run vectors: the CUDA extension cannot be built here
and the reference ships no fixtures for it."""
import numpy as np
from scipy.spatial import cKDTree

FLT_MAX = float(np.finfo(np.float32).max)


def mean_dist2(points):
    pts = np.asarray(points, dtype=np.float64)
    P = pts.shape[0]
    if P == 0:
        return np.zeros((0,), np.float64)
    k = min(4, P)
    d, _ = cKDTree(pts).query(pts, k=k)
    d = np.asarray(d).reshape(P, k)
    d2 = np.sort(d, axis=1)[:, 1:] ** 2          # drop one zero (the point itself)
    if k < 4:                                    # simple_knn.cu:146,156-158: unfilled slots stay FLT_MAX
        d2 = np.concatenate([d2, np.full((P, 4 - k), FLT_MAX)], axis=1)
    return d2.sum(axis=1) / 3.0
