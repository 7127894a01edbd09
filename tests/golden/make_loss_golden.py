"""Generates tests/golden/loss_*.npz from the CPU oracle (oracle/loss_oracle.py, float64).

    python tests/golden/make_loss_golden.py

These vectors come from the restatement; the restatement itself is pinned to the reference's utils/loss_utils.py by
the reference-run vectors of make_reference_vectors2.py (ref_loss.npz), see the oracle's header."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import loss_oracle as LO  # noqa: E402


def cases():
    return {
        "loss_plain": dict(seed=1, C=3, H=40, W=56, weight=False, lam=0.2, scale=1.0),
        "loss_masked": dict(seed=2, C=3, H=37, W=61, weight=True, lam=0.2, scale=1.0),       # odd sizes, weight map
        "loss_mono": dict(seed=3, C=1, H=16, W=33, weight=True, lam=0.5, scale=0.7),          # one channel, tile edge
        "loss_tiny": dict(seed=4, C=3, H=5, W=7, weight=False, lam=0.2, scale=1.0),           # smaller than the window
    }


def make_inputs(c):
    rng = np.random.default_rng(c["seed"])
    gt = rng.random((c["C"], c["H"], c["W"])).astype(np.float32)
    # a rendering close to the ground truth, as in training (SSIM well away from 0)
    img = np.clip(gt + 0.15 * rng.standard_normal(gt.shape), 0.0, 1.0).astype(np.float32)
    w = None
    if c["weight"]:
        w = (rng.random((1, c["H"], c["W"])) > 0.4).astype(np.float32) * np.float32(0.75) + np.float32(0.25)
    return img, gt, w


def depth_cases():
    return {"depth_ref_view": dict(seed=11, H=33, W=47, masked=False, l1=1.0, sm=0.5),
            "depth_other_view": dict(seed=12, H=40, W=29, masked=True, l1=0.7, sm=0.4)}


def make_depth_inputs(c):
    rng = np.random.default_rng(c["seed"])
    y = (rng.random((c["H"], c["W"])) * 5 + 1).astype(np.float32)
    d = (0.6 * y + 0.4 + 0.08 * rng.standard_normal(y.shape)).astype(np.float32)
    m = (rng.random(y.shape) > 0.3).astype(np.float32)
    return d, y, m, (m if c["masked"] else None)


def main():
    for name, c in depth_cases().items():
        d, y, m, wg = make_depth_inputs(c)
        loss, s, t, grad = LO.depth_value_and_grad(d, y, m, wg, wg, c["l1"], c["sm"])
        np.savez_compressed(os.path.join(HERE, "loss_" + name + ".npz"), loss=np.float64(loss), scale=np.float64(s), shift=np.float64(t),
                            grad=grad.astype(np.float64))
        print(name, loss, s, t)
    for name, c in cases().items():
        img, gt, w = make_inputs(c)
        loss, l1, ss, grad = LO.value_and_grad(img, gt, w, c["lam"], c["scale"])
        np.savez_compressed(os.path.join(HERE, name + ".npz"), loss=np.float64(loss), l1=np.float64(l1), ssim=np.float64(ss),
                            grad=grad.astype(np.float64))
        print(name, loss, l1, ss, grad.shape)


if __name__ == "__main__":
    main()
