"""Vectors computed by the reference's OWN Python helpers (tests/golden/make_reference_vectors.py, run in the build
container against /root/reference): the camera-matrix convention and the SH colour expansion.  These are the pinned
parts of the parity story; everything CUDA-only in the reference remains unpinned."""
import math
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import helpers as Hh  # noqa: E402
from gscream_amd import synthetic as S  # noqa: E402

CAM = np.load(os.path.join(ROOT, "tests", "golden", "ref_camera.npz"))
SHV = np.load(os.path.join(ROOT, "tests", "golden", "ref_sh.npz"))


@pytest.mark.parametrize("i", range(6))
def test_camera_matrices_match_the_reference_helpers(i):
    g = lambda k: CAM[f"{k}_{i}"]
    tanx, tany = math.tan(float(g("fovx")) / 2), math.tan(float(g("fovy")) / 2)
    view, proj, campos = S.camera_matrices(tanx, tany, g("w2v"), float(g("cx")), float(g("cy")), float(g("znear")), float(g("zfar")))
    assert np.allclose(view, g("world_view"), atol=1e-7)
    assert np.allclose(S.projection_matrix(float(g("znear")), float(g("zfar")), tanx, tany, float(g("cx")), float(g("cy"))).T,
                       g("projection"), atol=1e-6)
    assert np.allclose(proj, g("full_proj"), rtol=1e-5, atol=1e-6)
    assert np.allclose(campos, g("camera_center"), rtol=1e-5, atol=1e-6)
    assert abs(float(g("focal_roundtrip")) - float(g("fovx"))) < 1e-12


def _sh_scene(deg):
    """A scene whose view directions are exactly the fixture's: camera at the origin, Gaussian i at distance 3 along dirs[i]."""
    P = SHV["dirs"].shape[0]
    s = S.scene_config1(seed=5, P=P, W=64, H=64)
    s["means3D"] = (SHV["dirs"] * np.float32(3.0)).astype(np.float32)
    s["campos"] = np.zeros(3, np.float32)
    s["shs"] = np.ascontiguousarray(SHV["sh"][:, :(deg + 1) ** 2, :])
    s["sh_degree"] = deg
    s.pop("colors", None)
    return s


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_oracle_sh_colours_match_reference_eval_sh(deg):
    """forward.cu:22-73 = eval_sh + 0.5, clamped at 0; `clamped` flags where the clamp acted."""
    st = Hh.oracle_forward(_sh_scene(deg))
    expect = SHV[f"eval_sh_deg{deg}"] + np.float32(0.5)
    vis = st["radii"] > 0
    assert vis.sum() > 20                                    # the ones in front of the camera
    assert np.allclose(st["rgb"][vis], np.maximum(expect[vis], 0.0), atol=2e-6)
    assert np.array_equal(st["clamped"].reshape(-1, 3)[vis].astype(bool), expect[vis] < 0)


@pytest.mark.gpu
@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_hip_sh_colours_match_reference_eval_sh(deg):
    from gscream_amd import _layout
    s = _sh_scene(deg)
    got = Hh.hip_run(s, keep_state=True)
    rec = _layout.geom_views(got["geom"], s["means3D"].shape[0])["rec_f32"].cpu().numpy()
    vis = got["radii"] > 0
    expect = np.maximum(SHV[f"eval_sh_deg{deg}"] + np.float32(0.5), 0.0)
    assert vis.sum() > 20 and np.allclose(rec[vis, 8:11], expect[vis], atol=2e-6)   # record.c = {r, g, b, ...}
