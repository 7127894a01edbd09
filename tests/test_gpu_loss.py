"""GPU parity tests of the fused RGB loss (gscream_amd.loss_utils -> ctypes -> gsr_rgb_loss_*) against the CPU
oracle (float64 restatement of GScream's utils/loss_utils.py) and the committed fixtures.

Tolerances (floating point; fp32 kernel vs fp64 oracle): loss / terms 2e-6 abs; gradient 1e-4 of its largest entry
(the separable window reorders the 121-term sums; measured differences are ~1e-6)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from golden import make_loss_golden as MLG  # noqa: E402
from oracle import loss_oracle as LO  # noqa: E402

pytestmark = pytest.mark.gpu


def _run(img, gt, w, lam, scale):
    from gscream_amd import loss_utils as L
    x = torch.from_numpy(img).cuda().requires_grad_(True)
    y = torch.from_numpy(gt).cuda()
    m = None if w is None else torch.from_numpy(w).cuda()
    loss, l1, ss = L.rgb_loss(x, y, m, lam, scale, return_parts=True)
    loss.backward()
    return float(loss.detach()), float(l1), float(ss), x.grad.cpu().numpy()


@pytest.mark.parametrize("name", sorted(MLG.cases().keys()))
def test_golden(name):
    c = MLG.cases()[name]
    img, gt, w = MLG.make_inputs(c)
    exp = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    loss, l1, ss, grad = _run(img, gt, w, c["lam"], c["scale"])
    assert abs(loss - float(exp["loss"])) < 2e-6 and abs(l1 - float(exp["l1"])) < 2e-6 and abs(ss - float(exp["ssim"])) < 2e-6
    assert np.abs(grad - exp["grad"]).max() <= 1e-4 * np.abs(exp["grad"]).max()


@pytest.mark.parametrize("C,H,W,weighted", [(3, 71, 112, False), (3, 142, 252, True), (1, 33, 65, True), (3, 16, 32, False)])
def test_against_live_oracle(C, H, W, weighted):
    rng = np.random.default_rng(C * 1000 + H)
    gt = rng.random((C, H, W)).astype(np.float32)
    img = np.clip(gt + 0.2 * rng.standard_normal(gt.shape), 0, 1).astype(np.float32)
    w = rng.random((1, H, W)).astype(np.float32) if weighted else None
    ref = LO.value_and_grad(img, gt, w, 0.2, 1.0)
    got = _run(img, gt, w, 0.2, 1.0)
    for a, b in zip(got[:3], ref[:3]):
        assert abs(a - b) < 2e-6
    assert np.abs(got[3] - ref[3]).max() <= 1e-4 * np.abs(ref[3]).max()


def test_mirrored_functions_and_upstream_gradient():
    """The four names the trainer imports, with a non-unit upstream gradient (train.py:538-545 mixes them)."""
    from gscream_amd import loss_utils as L
    rng = np.random.default_rng(5)
    gt = rng.random((3, 48, 80)).astype(np.float32)
    img = np.clip(gt + 0.1 * rng.standard_normal(gt.shape), 0, 1).astype(np.float32)
    mask = (rng.random((1, 48, 80)) > 0.5).astype(np.float32)
    x = torch.from_numpy(img).cuda().requires_grad_(True)
    y, m = torch.from_numpy(gt).cuda(), torch.from_numpy(mask).cuda()
    total = 0.8 * L.l1_loss(x, y) + 0.2 * (1.0 - L.ssim(x, y)) + 0.5 * (0.8 * L.l1_loss_masked(x, y, m) + 0.2 * (1.0 - L.ssim_masked(x, y, m)))
    total.backward()
    xr = torch.from_numpy(img).double().requires_grad_(True)
    yr, mr = torch.from_numpy(gt).double(), torch.from_numpy(mask).double()
    ref = 0.8 * LO.l1_loss(xr, yr) + 0.2 * (1.0 - LO.ssim(xr, yr)) + 0.5 * (0.8 * LO.l1_loss_masked(xr, yr, mr) + 0.2 * (1.0 - LO.ssim_masked(xr, yr, mr)))
    ref.backward()
    assert abs(float(total.detach()) - float(ref.detach())) < 3e-6
    assert (x.grad.cpu().double() - xr.grad).abs().max() <= 1e-4 * xr.grad.abs().max()
    # other window sizes (loss_utils.py:131 `window_size`): odd, up to 11 -- the same kernels with zero taps at both ends
    for ws in (1, 3, 7):
        x2 = torch.from_numpy(img).cuda().requires_grad_(True)
        v = L.ssim(x2, y, window_size=ws)
        vm = L.ssim_masked(x2, y, m, window_size=ws)
        (v + 0.5 * vm).backward()
        x3 = torch.from_numpy(img).double().requires_grad_(True)
        r = LO.ssim_map(x3, yr, ws).mean()
        rm = (LO.ssim_map(x3, yr, ws) * mr).mean()
        (r + 0.5 * rm).backward()
        assert abs(float(v.detach()) - float(r.detach())) < 3e-6 and abs(float(vm.detach()) - float(rm.detach())) < 3e-6, ws
        assert (x2.grad.cpu().double() - x3.grad).abs().max() <= 1e-4 * x3.grad.abs().max(), ws
    with pytest.raises(NotImplementedError):
        L.ssim(x, y, window_size=8)   # even: the reference's map changes size
    with pytest.raises(NotImplementedError):
        L.ssim(x, y, window_size=13)  # beyond the 11-tap frame
    with pytest.raises(RuntimeError):
        L.ssim(x.detach().cpu(), y.cpu())
    # size_average=False (loss_utils.py:157-160): one mean per image of a [B,C,H,W] batch; B = 1 here
    per = L.ssim(x.detach()[None], y[None], size_average=False)
    assert per.shape == (1,) and abs(float(per[0]) - float(LO.ssim(xr.detach(), yr))) < 3e-6
    with pytest.raises(NotImplementedError):
        L.ssim(x, y, size_average=False)  # the reference's .mean(1).mean(1).mean(1) needs the batch dimension too


def test_full_size_properties_and_reproducibility():
    """BASELINE's image size (3 x 567 x 1008): identities that need no oracle, and bit-reproducibility."""
    from gscream_amd import loss_utils as L
    g = torch.Generator(device="cuda").manual_seed(0)
    y = torch.rand((3, 567, 1008), device="cuda", generator=g)
    assert abs(float(L.ssim(y, y)) - 1.0) < 1e-5 and float(L.l1_loss(y, y)) == 0.0
    x = (y + 0.1 * torch.randn(y.shape, device="cuda", generator=g)).clamp(0, 1).requires_grad_(True)
    a = L.rgb_loss(x, y)
    a.backward()
    g1 = x.grad.clone()
    x.grad = None
    b = L.rgb_loss(x, y)
    b.backward()
    assert float(a) == float(b) and torch.equal(g1, x.grad), "fixed-order reductions: bit-reproducible"
    assert abs(float(L.ssim(x, y)) - float(L.ssim(y, x.detach()))) < 1e-6  # symmetric
    # directional derivative check at full size: L(x + eps d) - L(x - eps d) ~= 2 eps <grad, d>
    d = torch.sign(g1)  # steepest-ascent direction: the derivative along it is sum |grad|, far above fp32 noise
    eps = 1e-3
    with torch.no_grad():
        fd = (float(L.rgb_loss(x + eps * d, y)) - float(L.rgb_loss(x - eps * d, y))) / (2 * eps)
    an = float((g1 * d).sum())
    assert an > 0 and abs(fd - an) < 5e-2 * an, (fd, an)


# ---- depth terms --------------------------------------------------------------------------------------------------
def _depth_case(seed, H, W, masked):
    rng = np.random.default_rng(seed)
    y = (rng.random((H, W)) * 5 + 1).astype(np.float32)                      # "midas" target
    d = (0.6 * y + 0.4 + 0.08 * rng.standard_normal((H, W))).astype(np.float32)  # rendered depth: affine + noise
    m = (rng.random((H, W)) > 0.3).astype(np.float32)
    return d, y, m, (m if masked else None)


@pytest.mark.parametrize("H,W,masked,seed", [(71, 112, False, 1), (142, 252, True, 2), (9, 17, True, 3), (1, 33, False, 4), (64, 64, True, 5)])
def test_depth_loss_against_oracle(H, W, masked, seed):
    """compute_scale_and_shift + |scale| + L1 + four-scale gradient loss, value and gradient through the fit."""
    from gscream_amd import loss_utils as L
    d, y, m, wg = _depth_case(seed, H, W, masked)
    ref_loss, ref_s, ref_t, ref_g = LO.depth_value_and_grad(d, y, m, wg, wg, 0.7, 0.4)
    t = lambda a: None if a is None else torch.from_numpy(a).cuda().reshape(1, H, W)
    x = t(d).requires_grad_(True)
    loss, parts = L.depth_loss(x, t(y), t(m), t(wg), t(wg), 0.7, 0.4, return_parts=True)
    (1.5 * loss).backward()
    assert abs(float(loss.detach()) - ref_loss) < 2e-6 * max(1.0, abs(ref_loss))
    assert abs(float(parts[3]) - ref_s) < 2e-6 * max(1.0, abs(ref_s)) and abs(float(parts[4]) - ref_t) < 5e-6 * max(1.0, abs(ref_t))
    got = x.grad.cpu().numpy().reshape(H, W) / 1.5
    # |.| kinks: a pixel whose residual or edge difference sits within fp32 rounding of zero may take the other sign
    bad = np.abs(got - ref_g) > 1e-4 * np.abs(ref_g).max()
    assert bad.mean() <= 2e-3, float(bad.mean())


def test_depth_loss_reference_view_with_foreground_term():
    """The shipped run config (scripts/run.py: refer_depth_lr_fg = 100 > refer_depth_lr = 1) adds
    (fg - lr) * l1_loss_masked(aligned, midas, fg_mask) at train.py:555-557 -- the dominant depth term."""
    from gscream_amd import loss_utils as L
    H, W = 96, 160
    d, y, m, _ = _depth_case(11, H, W, False)
    fg = np.zeros((H, W), np.float32)
    fg[20:70, 40:120] = 1.0
    ref_loss, ref_s, ref_t, ref_g = LO.depth_value_and_grad(d, y, m, None, None, 1.0, 1.0, fg_mask=fg, lambda_fg=99.0)
    plain, *_ = LO.depth_value_and_grad(d, y, m, None, None, 1.0, 1.0)
    assert ref_loss > 3 * plain, "the foreground term dominates in the shipped configuration"
    t = lambda a: torch.from_numpy(a).cuda().reshape(1, H, W)
    x = t(d).requires_grad_(True)
    loss = L.depth_loss(x, t(y), t(m), None, None, 1.0, 1.0, fg_mask=t(fg), lambda_fg=99.0)
    loss.backward()
    assert abs(float(loss.detach()) - ref_loss) < 5e-6 * max(1.0, abs(ref_loss))
    bad = np.abs(x.grad.cpu().numpy().reshape(H, W) - ref_g) > 1e-4 * np.abs(ref_g).max()
    assert bad.mean() <= 2e-3, float(bad.mean())


def test_depth_loss_properties_full_size():
    from gscream_amd import loss_utils as L
    g = torch.Generator(device="cuda").manual_seed(3)
    y = torch.rand((1, 567, 1008), device="cuda", generator=g) * 4 + 1
    exact = (0.5 * y + 2.0)                       # exactly affine: the fit recovers it, every term vanishes
    loss, parts = L.depth_loss(exact, y, None, None, None, 1.0, 1.0, return_parts=True)
    assert float(loss) < 1e-5 and abs(float(parts[3]) - 2.0) < 1e-4 and abs(float(parts[4]) + 4.0) < 1e-3
    d = (exact + 0.05 * torch.randn(y.shape, device="cuda", generator=g)).requires_grad_(True)
    a = L.depth_loss(d, y, None, None, None, 1.0, 0.5)
    a.backward()
    g1 = d.grad.clone()
    d.grad = None
    b = L.depth_loss(d, y, None, None, None, 1.0, 0.5)
    b.backward()
    assert float(a) == float(b) and torch.equal(g1, d.grad), "bit-reproducible"
    # the loss is invariant to an affine change of the input depth (the fit absorbs it): gradient is orthogonal to 1 and d
    assert abs(float(g1.sum())) < 1e-4 * float(g1.abs().sum()) and abs(float((g1 * d.detach()).sum())) < 1e-4 * float((g1.abs() * d.detach().abs()).sum())


@pytest.mark.parametrize("name", sorted(MLG.depth_cases().keys()))
def test_depth_golden(name):
    from gscream_amd import loss_utils as L
    c = MLG.depth_cases()[name]
    d, y, m, wg = MLG.make_depth_inputs(c)
    exp = np.load(os.path.join(ROOT, "tests", "golden", "loss_" + name + ".npz"))
    H, W = d.shape
    t = lambda a: None if a is None else torch.from_numpy(a).cuda().reshape(1, H, W)
    x = t(d).requires_grad_(True)
    loss, parts = L.depth_loss(x, t(y), t(m), t(wg), t(wg), c["l1"], c["sm"], return_parts=True)
    loss.backward()
    assert abs(float(loss.detach()) - float(exp["loss"])) < 2e-6 and abs(float(parts[3]) - float(exp["scale"])) < 2e-6
    bad = np.abs(x.grad.cpu().numpy().reshape(H, W) - exp["grad"]) > 1e-4 * np.abs(exp["grad"]).max()
    assert bad.mean() <= 2e-3

