"""End to end under a real optimiser: train.py's iteration (prefilter -> decode -> rasterize -> RGB + depth losses -> backward ->
training_statis -> Adam step with the reference's parameter groups, gscream_amd/fit.py) against images of a synthetic teacher scene.
If any gradient of the chain pointed the wrong way, or the rows disagreed about layouts, the loss would not fall.
Reference: train.py:414-416, 433, 527, 535-575, 597-602, 627; scene/gaussian_model.py:350-395; arguments/__init__.py:93-133."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gscream_amd import fit as F  # noqa: E402
from gscream_amd import set_tuning  # noqa: E402

pytestmark = pytest.mark.gpu


def test_the_loss_falls_and_the_psnr_rises():
    set_tuning()
    W, H = 208, 117
    s, info = F.scene_fitted(3, W, H, iters=200, n_student=20_000, n_teacher=80_000, V=8, return_info=True)
    assert info["iterations"] == 200 and info["gaussians"] > 10_000
    assert np.isfinite(info["loss"]).all()
    assert info["loss"][-1] < 0.7 * info["loss"][0], info["loss"]
    assert info["psnr_last"] > info["psnr_first"] + 3.0, (info["psnr_first"], info["psnr_last"])
    # the fitted frame is a valid rasterizer-level scene
    for k in ("means3D", "scales", "rotations", "opacities", "colors"):
        assert np.isfinite(s[k]).all(), k
    assert s["means3D"].shape[0] == info["gaussians"]


def test_a_second_run_gives_the_same_model():
    """Fixed seeds + bit-reproducible kernels (no float atomics on the path): two runs end in the same place."""
    a, ia = F.scene_fitted(5, 160, 90, iters=40, n_student=8_000, n_teacher=30_000, V=4, return_info=True)
    b, ib = F.scene_fitted(5, 160, 90, iters=40, n_student=8_000, n_teacher=30_000, V=4, return_info=True)
    assert ia["gaussians"] == ib["gaussians"]
    for k in ("means3D", "scales", "opacities", "colors"):
        assert np.array_equal(a[k], b[k]), k
