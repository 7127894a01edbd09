"""SURVEY row a18 + the filter call sites, pinned by a RECORDED run of the reference's own callers
(tests/golden/make_reference_vectors3.py -> ref_render_call.npz: /root/reference/gaussian_renderer/__init__.py:104-179,190-302
executed in the build container with recording rasterizer classes).  This package's mirror (gscream_amd/gaussian_renderer.py) is
driven through the SAME recorder and must hand the rasterizer the same thing; then the real rasterizer must give train.py back
what the reference's dict holds."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

FIX = dict(np.load(os.path.join(ROOT, "tests", "golden", "ref_render_call.npz")))


def _parse(d):
    if not d.startswith("tensor|"):
        return d
    _, dtype, shape, cont, rg, leaf = d.split("|")
    return dict(dtype=dtype, shape=tuple(int(x) for x in shape.split("x")) if shape else (), contiguous=int(cont[-1]),
                requires_grad=int(rg[-1]), is_leaf=int(leaf[-1]))


def _same_argument(ref, got, what, per_gaussian_rows=None, leaf_may_differ=False, grad_may_be_absent=False):
    """`got` (the mirror) against `ref` (the reference's recorded argument).  Allowed to differ, and only these:
    * rows of a per-Gaussian tensor in the render scenarios: how many decoded Gaussians pass `opacity > 0` depends on the fp32
      rounding of the opacity MLP (the reference ran it on the CPU, the mirror on the GPU): compared as "the same P everywhere";
    * contiguity may be BETTER (the fused decode returns contiguous tensors where the reference hands over column slices; the
      rasterizer calls .contiguous() on every pointer like rasterize_points.cu:98-118 either way);
    * `means2D` with retain_grad=True: the mirror passes a leaf (a leaf keeps .grad by itself) where the reference passes
      `zeros + 0` with retain_grad() -- the recorded `viewspace_points_grad_after_backward` is what the caller depends on;
    * the returned `neural_opacity` (grad_may_be_absent): in the reference it is a live MLP output, in the mirror the fused decode marks it
      non-differentiable -- its only consumer, train.py:599 -> GaussianModel.training_statis, detaches it first
      (scene/gaussian_model.py:733), and the opacities that DO carry gradient reach the rasterizer as `opacities`."""
    r, g = _parse(ref), _parse(got)
    if not isinstance(r, dict):
        assert g == r, (what, ref, got)
        return
    assert isinstance(g, dict), (what, ref, got)
    assert g["dtype"] == r["dtype"] and (g["requires_grad"] == r["requires_grad"] or grad_may_be_absent), (what, ref, got)
    assert g["contiguous"] >= r["contiguous"], (what, ref, got)
    if not leaf_may_differ and not grad_may_be_absent:
        assert g["is_leaf"] == r["is_leaf"], (what, ref, got)
    if per_gaussian_rows is not None and len(r["shape"]) >= 1 and r["shape"][0] == per_gaussian_rows[0]:
        assert g["shape"][1:] == r["shape"][1:] and g["shape"][0] == per_gaussian_rows[1], (what, ref, got)
    else:
        assert g["shape"] == r["shape"], (what, ref, got)


def _compare(got, scenarios):
    for sc in scenarios:
        for key in ("ctor", "settings_fields", "settings_values", "method", "call_kwargs"):
            assert list(got[f"{sc}/{key}"]) == list(FIX[f"{sc}/{key}"]), (sc, key, list(got[f"{sc}/{key}"]), list(FIX[f"{sc}/{key}"]))
        rows = None
        if sc.startswith("render"):
            rows = (_parse(str(FIX[f"{sc}/call_values"][0]))["shape"][0], _parse(str(got[f"{sc}/call_values"][0]))["shape"][0])
            assert abs(rows[0] - rows[1]) <= 3, ("decoded Gaussian count", rows)
        for name, r, g in zip(FIX[f"{sc}/call_kwargs"], FIX[f"{sc}/call_values"], got[f"{sc}/call_values"]):
            _same_argument(str(r), str(g), (sc, str(name)), rows, leaf_may_differ=(sc == "render_train_retain" and name == "means2D"))
        if f"{sc}/return_keys" in FIX:
            assert list(got[f"{sc}/return_keys"]) == list(FIX[f"{sc}/return_keys"]), sc
            for name, r, g in zip(FIX[f"{sc}/return_keys"], FIX[f"{sc}/return_values"], got[f"{sc}/return_values"]):
                if name in ("selection_mask", "neural_opacity"):   # rows = visible anchors x K: identical (same mask)
                    _same_argument(str(r), str(g), (sc, "->", str(name)), grad_may_be_absent=(name == "neural_opacity"))
                else:
                    _same_argument(str(r), str(g), (sc, "->", str(name)), rows, leaf_may_differ=(sc == "render_train_retain" and name == "viewspace_points"))
        else:
            for i, (r, g) in enumerate(zip(FIX[f"{sc}/return_values"], got[f"{sc}/return_values"])):
                _same_argument(str(r), str(g), (sc, "->", i))
        k = f"{sc}/viewspace_points_grad_after_backward"
        if k in FIX:
            assert int(got[k][0]) == int(FIX[k][0]), (k, "the densification statistics read viewspace_points.grad (scene/gaussian_model.py:755)")


def test_fixture_holds_what_survey_a18_says():
    """The recorded facts themselves (a reader's summary of gaussian_renderer/__init__.py:131-158 in SURVEY 8 a18, now data)."""
    assert list(FIX["render_train_retain/ctor"]) == ["positional=0", "kw:raster_settings"]
    assert list(FIX["render_train_retain/settings_fields"]) == ["image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix",
                                                                "projmatrix", "sh_degree", "campos", "prefiltered", "debug"]
    vals = dict(zip(FIX["render_train_retain/settings_fields"], FIX["render_train_retain/settings_values"]))
    assert vals["sh_degree"] == "int|1" and vals["prefiltered"] == "bool|False" and vals["viewmatrix"] == "is:viewpoint_camera.world_view_transform"
    assert vals["projmatrix"] == "is:viewpoint_camera.full_proj_transform" and vals["campos"] == "is:viewpoint_camera.camera_center" and vals["bg"] == "is:bg_color"
    assert list(FIX["render_train_retain/call_kwargs"]) == ["means3D", "means2D", "shs", "colors_precomp", "opacities", "uncertainties", "scales",
                                                            "rotations", "cov3D_precomp"]
    call = dict(zip(FIX["render_train_retain/call_kwargs"], FIX["render_train_retain/call_values"]))
    assert call["shs"] == "None" and call["cov3D_precomp"] == "None"
    assert _parse(str(call["means2D"]))["is_leaf"] == 0 and _parse(str(call["means2D"]))["requires_grad"] == 1      # zeros_like(...) + 0
    assert list(FIX["render_eval/return_keys"]) == ["render", "render_depth", "uncertainty", "viewspace_points", "visibility_filter", "radii"]
    assert list(FIX["render_train_retain/return_keys"])[6:] == ["selection_mask", "neural_opacity", "scaling"]
    assert int(FIX["render_train_retain/viewspace_points_grad_after_backward"][0]) == 1 and int(FIX["render_train_noretain/viewspace_points_grad_after_backward"][0]) == 0
    # the anchor filters: keyword call, scales = the NON-CONTIGUOUS column slice get_scaling[:, :3]
    for sc, method in (("prefilter_voxel", "visible_filter"), ("prefilter_position2D", "position2D_filter")):
        assert list(FIX[f"{sc}/method"]) == [method, "positional=0"]
        assert list(FIX[f"{sc}/call_kwargs"]) == ["means3D", "scales", "rotations", "cov3D_precomp"]
        sc_arg = _parse(str(FIX[f"{sc}/call_values"][1]))
        assert sc_arg["shape"][1] == 3 and sc_arg["contiguous"] == 0


def test_mirror_filters_make_the_recorded_calls():
    """prefilter_voxel / prefilter_position2D of the mirror through the recorder (no decode involved: runs without a GPU)."""
    import make_reference_vectors3 as V3
    from gscream_amd import gaussian_renderer as GRM
    saved = (GRM.GaussianRasterizationSettings, GRM.GaussianRasterizer)
    try:
        V3.SCENARIOS, keep = ("prefilter_voxel", "prefilter_position2D"), V3.SCENARIOS
        got = V3.run(GRM)
    finally:
        V3.SCENARIOS = keep
        GRM.GaussianRasterizationSettings, GRM.GaussianRasterizer = saved
    _compare(got, ("prefilter_voxel", "prefilter_position2D"))


@pytest.mark.gpu
def test_mirror_render_makes_the_recorded_calls_and_returns_the_recorded_dict():
    """All five scenarios through the mirror on the GPU (the fused HIP decode feeds the recorder), then the REAL rasterizer:
    the dict train.py receives has the reference's keys, dtypes and shapes, and viewspace_points.grad behaves as recorded."""
    import make_reference_vectors3 as V3
    from gscream_amd import gaussian_renderer as GRM
    saved = (GRM.GaussianRasterizationSettings, GRM.GaussianRasterizer)
    try:
        got = V3.run(GRM, device="cuda")
    finally:
        GRM.GaussianRasterizationSettings, GRM.GaussianRasterizer = saved
    _compare(got, V3.SCENARIOS)
    # the real thing
    for sc in ("render_train_retain", "render_train_noretain", "render_eval"):
        m, cam, pipe, bg, vis = V3.standin("cuda")
        m.train(sc.startswith("render_train"))
        res = GRM.render(cam, m, pipe, bg, visible_mask=vis, retain_grad=(sc == "render_train_retain"))
        assert list(res) == list(FIX[f"{sc}/return_keys"])
        P = res["radii"].shape[0]
        for name, r in zip(FIX[f"{sc}/return_keys"], FIX[f"{sc}/return_values"]):
            ref, v = _parse(str(r)), res[str(name)]
            assert str(v.dtype).replace("torch.", "") == ref["dtype"] and v.is_cuda, (sc, name)
            assert tuple(v.shape[1:]) == ref["shape"][1:], (sc, name)
            if str(name) in ("render", "render_depth", "uncertainty"):
                assert tuple(v.shape) == ref["shape"]
            elif str(name) not in ("selection_mask", "neural_opacity"):
                assert v.shape[0] == P
        if sc.startswith("render_train"):
            (res["render"].sum() + res["render_depth"].sum()).backward()
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                has = res["viewspace_points"].grad is not None
            assert int(has) == int(FIX[f"{sc}/viewspace_points_grad_after_backward"][0])
            if has:
                assert res["viewspace_points"].grad[res["visibility_filter"], :2].abs().sum() > 0
    m, cam, pipe, bg, vis = V3.standin("cuda")
    v = GRM.prefilter_voxel(cam, m, pipe, bg)
    assert v.dtype == torch.bool and tuple(v.shape) == _parse(str(FIX["prefilter_voxel/return_values"][0]))["shape"]
    v, x, y = GRM.prefilter_position2D(cam, m, pipe, bg)
    for t_, r in zip((v, x, y), FIX["prefilter_position2D/return_values"]):
        assert str(t_.dtype).replace("torch.", "") == _parse(str(r))["dtype"] and tuple(t_.shape) == _parse(str(r))["shape"]
