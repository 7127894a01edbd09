"""CPU tests of the loss oracle (oracle/loss_oracle.py) and of the committed loss fixtures."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from golden import make_loss_golden as MLG  # noqa: E402
from oracle import loss_oracle as LO  # noqa: E402


def test_window_matches_the_published_constants():
    g = LO.gaussian(11, 1.5)
    assert abs(float(g.sum()) - 1.0) < 1e-6 and g.argmax() == 5 and torch.allclose(g, g.flip(0))
    assert abs(float(g[5]) - 0.26601171) < 1e-7 and abs(float(g[0]) - 0.00102838) < 1e-8
    # the 2-D window is the outer product, so it is separable -- the property the HIP kernel relies on
    w = LO.create_window(11, 3, torch.float64)[0, 0]
    assert torch.allclose(w, torch.outer(g.double(), g.double()), atol=1e-9)


def test_identities():
    rng = np.random.default_rng(0)
    x = torch.from_numpy(rng.random((3, 24, 31)))
    assert abs(float(LO.ssim(x, x)) - 1.0) < 1e-9 and float(LO.l1_loss(x, x)) == 0.0
    m = torch.from_numpy((rng.random((1, 24, 31)) > 0.5).astype(np.float64))
    assert abs(float(LO.ssim_masked(x, x, m)) - float(m.mean())) < 1e-9
    assert abs(float(LO.rgb_loss(x, x)) - 0.0) < 1e-9
    y = torch.from_numpy(rng.random((3, 24, 31)))
    assert abs(float(LO.ssim(x, y)) - float(LO.ssim(y, x))) < 1e-12  # symmetric in its arguments


def test_gradient_against_finite_differences():
    rng = np.random.default_rng(1)
    x, y = rng.random((2, 9, 12)), rng.random((2, 9, 12))
    w = rng.random((1, 9, 12))
    _, _, _, g = LO.value_and_grad(x, y, w, 0.3, 1.7)
    eps = 1e-6
    for idx in [(0, 0, 0), (1, 4, 7), (0, 8, 11), (1, 3, 0)]:
        xp, xm = x.copy(), x.copy()
        xp[idx] += eps
        xm[idx] -= eps
        fd = (LO.value_and_grad(xp, y, w, 0.3, 1.7)[0] - LO.value_and_grad(xm, y, w, 0.3, 1.7)[0]) / (2 * eps)
        assert abs(fd - g[idx]) < 1e-6 * max(1.0, abs(fd)), (idx, fd, g[idx])


def test_fixtures_reproduce():
    for name, c in MLG.cases().items():
        img, gt, w = MLG.make_inputs(c)
        exp = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
        loss, l1, ss, grad = LO.value_and_grad(img, gt, w, c["lam"], c["scale"])
        assert abs(loss - float(exp["loss"])) < 1e-12 and abs(ss - float(exp["ssim"])) < 1e-12
        assert np.allclose(grad, exp["grad"], rtol=0, atol=1e-14)
        # fp32 evaluation of the same restatement stays within the GPU test's tolerance of the fp64 fixture
        l32 = LO.value_and_grad(img, gt, w, c["lam"], c["scale"], dtype=torch.float32)[0]
        assert abs(l32 - loss) < 2e-6


def test_knn_oracle_against_brute_force():
    """oracle/knn_oracle.py (exact 3-NN through a k-d tree) against the O(P^2) definition."""
    from oracle import knn_oracle as KO
    rng = np.random.default_rng(3)
    pts = rng.random((400, 3))
    pts[10:20] = pts[0:10]  # coincident points are neighbours at distance 0
    d2 = ((pts[:, None, :] - pts[None, :, :]) ** 2).sum(-1)
    np.fill_diagonal(d2, np.inf)
    ref = np.sort(d2, axis=1)[:, :3].mean(1)
    assert np.allclose(KO.mean_dist2(pts), ref, rtol=1e-12, atol=1e-15)
    assert KO.mean_dist2(pts[:0]).shape == (0,) and np.all(KO.mean_dist2(pts[:3]) > 1e37)


def test_depth_oracle_gradient_and_fixtures():
    """The depth terms: autograd gradient (through the least-squares fit) against finite differences, the closed-form
    fit against numpy's lstsq, and the committed fixtures."""
    rng = np.random.default_rng(5)
    y = rng.random((7, 9)) * 4 + 1
    d = 0.8 * y - 0.3 + 0.1 * rng.standard_normal(y.shape)
    m = (rng.random(y.shape) > 0.25).astype(np.float64)
    loss, s, t, g = LO.depth_value_and_grad(d, y, m, m, m, 0.9, 0.6)
    A = np.stack([d[m > 0], np.ones(int(m.sum()))], 1)
    sol = np.linalg.lstsq(A, y[m > 0], rcond=None)[0]
    assert abs(abs(sol[0]) - s) < 1e-9 and abs(sol[1] - t) < 1e-9
    eps = 1e-6
    for idx in [(0, 0), (3, 4), (6, 8), (2, 7)]:
        dp, dm = d.copy(), d.copy()
        dp[idx] += eps
        dm[idx] -= eps
        fd = (LO.depth_value_and_grad(dp, y, m, m, m, 0.9, 0.6)[0] - LO.depth_value_and_grad(dm, y, m, m, m, 0.9, 0.6)[0]) / (2 * eps)
        assert abs(fd - g[idx]) < 1e-6 * max(1.0, abs(fd)), (idx, fd, g[idx])
    for name, c in MLG.depth_cases().items():
        dd, yy, mm, wg = MLG.make_depth_inputs(c)
        exp = np.load(os.path.join(ROOT, "tests", "golden", "loss_" + name + ".npz"))
        l2, s2, t2, g2 = LO.depth_value_and_grad(dd, yy, mm, wg, wg, c["l1"], c["sm"])
        assert abs(l2 - float(exp["loss"])) < 1e-12 and np.allclose(g2, exp["grad"], rtol=0, atol=1e-14)

