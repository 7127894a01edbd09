"""GPU parity of gscream_amd.simple_knn.distCUDA2 (-> gsr_knn_mean_dist2) against the exact 3-NN oracle.
Tolerance: 2e-5 relative (fp32 distances, contracted FMAs, vs float64)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import knn_oracle as KO  # noqa: E402

pytestmark = pytest.mark.gpu


def _check(pts, rtol=2e-5):
    from simple_knn._C import distCUDA2  # the import path GScream uses
    got = distCUDA2(torch.from_numpy(pts).cuda()).cpu().numpy().astype(np.float64)
    ref = KO.mean_dist2(pts.astype(np.float32))
    assert got.shape == ref.shape
    assert np.all(np.abs(got - ref) <= rtol * np.abs(ref) + 1e-12), float(np.max(np.abs(got - ref) / (np.abs(ref) + 1e-30)))


@pytest.mark.parametrize("P,kind", [(5000, "uniform"), (20000, "clustered"), (257, "uniform"), (4, "uniform"), (100000, "surface")])
def test_against_exact_knn(P, kind):
    rng = np.random.default_rng(P)
    if kind == "uniform":
        pts = rng.random((P, 3)) * 4 - 2
    elif kind == "clustered":  # dense blobs + far outliers: the pruning must stay exact
        c = rng.random((20, 3)) * 10
        pts = c[rng.integers(0, 20, P)] + 0.05 * rng.standard_normal((P, 3))
        pts[:50] = rng.random((50, 3)) * 1000 - 500
    else:                      # points on a surface, like an SfM cloud
        uv = rng.random((P, 2))
        pts = np.stack([uv[:, 0] * 6, uv[:, 1] * 4, np.sin(uv[:, 0] * 7) + 0.01 * rng.standard_normal(P)], 1)
    _check(pts.astype(np.float32))


def test_degenerate_inputs():
    from simple_knn._C import distCUDA2
    rng = np.random.default_rng(0)
    pts = (rng.random((3000, 3)) * 2).astype(np.float32)
    pts[100:200] = pts[0:100]            # coincident pairs: neighbours at distance 0
    pts[500:600, 2] = 0.25               # coplanar
    _check(pts)
    line = np.zeros((1000, 3), np.float32)
    line[:, 0] = np.linspace(0, 1, 1000)  # zero extent in y and z (Morton grid degenerates, result must not)
    _check(line)
    assert distCUDA2(torch.zeros((0, 3), device="cuda")).shape == (0,)
    three = distCUDA2(torch.from_numpy(pts[:3]).cuda()).cpu().numpy()
    assert np.all(three > 1e37)           # fewer than 4 points: FLT_MAX terms, as in the reference
    with pytest.raises(RuntimeError):
        distCUDA2(torch.zeros((4, 3)))
    with pytest.raises(ValueError):
        distCUDA2(torch.zeros((4, 2), device="cuda"))


def test_one_million_points_properties():
    """Size-independent properties at 1M points: permutation invariance and a brute-force spot check."""
    from simple_knn._C import distCUDA2
    g = torch.Generator(device="cuda").manual_seed(1)
    pts = torch.rand((1_000_000, 3), device="cuda", generator=g) * 10
    d = distCUDA2(pts)
    perm = torch.randperm(pts.shape[0], device="cuda", generator=g)
    d2 = distCUDA2(pts[perm])
    assert torch.equal(d[perm], d2), "result must not depend on the input order (bitwise: same fp32 distances)"
    idx = torch.randint(0, pts.shape[0], (64,), device="cuda", generator=g)
    for i in idx.tolist():
        dist = ((pts - pts[i]) ** 2).sum(1)
        dist[i] = float("inf")
        ref = dist.topk(3, largest=False).values.double().mean().item()
        assert abs(d[i].item() - ref) <= 2e-5 * ref
