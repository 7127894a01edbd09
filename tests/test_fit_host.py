"""CPU checks of gscream_amd/fit.py against values the reference's own argument classes produced (tests/golden/ref_optim.json, made by
tests/golden/make_reference_vectors4.py): the Adam groups' names and learning rates (scene/gaussian_model.py:376-390 reads them from
OptimizationParams by these names), lambda_dssim (train.py:545), and the model sizes.  No compute: there is no GPU in the build container."""
import inspect
import json
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def _ref():
    with open(os.path.join(HERE, "golden", "ref_optim.json")) as f:
        return json.load(f)


def test_adam_groups_are_the_references():
    from gscream_amd import fit as F
    from gscream_amd import standin_model as SM
    ref = _ref()["OptimizationParams"]
    m = SM.Model(64, 10, seed=0, dtype=torch.float32)
    groups = {g["name"]: g for g in F.adam_groups(m)}
    # group name -> the OptimizationParams field training_setup reads for it (x spatial_lr_scale for the offsets)
    field = {"offset": "offset_lr_init", "anchor_feat": "feature_lr", "scaling": "scaling_lr", "mlp_opacity": "mlp_opacity_lr_init",
             "mlp_uncertainty": "mlp_uncertainty_lr_init", "mlp_cov": "mlp_cov_lr_init", "mlp_color": "mlp_color_lr_init"}
    assert set(groups) == set(field)
    for name, f in field.items():
        assert groups[name]["lr"] == ref[f], (name, groups[name]["lr"], ref[f])
    # the groups fit.py leaves out: the anchors (rate 0 in the reference) and the per-anchor opacity / uncertainty / rotation tensors the
    # neural-Gaussian decode never reads (gaussian_renderer/__init__.py:18-102)
    assert ref["position_lr_init"] == 0.0
    # every parameter the decode differentiates is in exactly one group
    in_groups = [id(p) for g in groups.values() for p in g["params"]]
    assert len(in_groups) == len(set(in_groups))
    decoded = [m._offset, m._anchor_feat, m._scaling, *m.mlp_opacity.parameters(), *m.mlp_uncertainty.parameters(), *m.mlp_cov.parameters(),
               *m.mlp_color.parameters()]
    assert {id(p) for p in decoded} == set(in_groups)


def test_fit_defaults_are_the_references():
    from gscream_amd import fit as F
    from gscream_amd import standin_model as SM
    ref = _ref()
    sig = inspect.signature(F.fit)
    assert sig.parameters["lambda_dssim"].default == ref["OptimizationParams"]["lambda_dssim"]
    sf = inspect.signature(F.scene_fitted)
    assert sf.parameters["K"].default == ref["ModelParams"]["n_offsets"]
    assert inspect.signature(SM.Model.__init__).parameters["feat_dim"].default == ref["ModelParams"]["feat_dim"]
    assert inspect.signature(SM.voxelize).parameters["voxel_size"].default == ref["ModelParams"]["voxel_size"]
    assert ref["ModelParams"]["use_feat_bank"] is False


def test_learning_rate_schedule_is_the_references():
    """fit.expon_lr / LR_SCHEDULES against the reference's get_expon_lr_func configured as training_setup configures it
    (scene/gaussian_model.py:412-439), evaluated by the reference itself at a few iterations (ref_optim.json)."""
    from gscream_amd import fit as F
    ref = _ref()
    op = ref["OptimizationParams"]
    assert set(F.LR_SCHEDULES) == set(ref["schedules"])
    for name, (init, final, steps) in F.LR_SCHEDULES.items():
        assert (init, final, steps) == (op[name + "_lr_init"], op[name + "_lr_final"], op[name + "_lr_max_steps"]), name
        assert op[name + "_lr_delay_mult"] == 0.01  # (no effect: lr_delay_steps stays at its default 0)
        for st, want in zip(ref["schedule_steps"], ref["schedules"][name]):
            got = F.expon_lr(st, init, final, steps)
            assert abs(got - want) <= 1e-12 * max(abs(want), 1e-30) + 1e-18, (name, st, got, want)
    # the optimiser's groups follow it
    from gscream_amd import standin_model as SM
    m = SM.Model(8, 10, seed=0, dtype=torch.float32)
    opt = torch.optim.Adam(F.adam_groups(m), lr=0.0, eps=1e-15)
    F.update_learning_rate(opt, 1600)
    lrs = {g["name"]: g["lr"] for g in opt.param_groups}
    i = ref["schedule_steps"].index(1600)
    for name in F.LR_SCHEDULES:
        assert abs(lrs[name] - ref["schedules"][name][i]) < 1e-15
    assert lrs["anchor_feat"] == op["feature_lr"] and lrs["scaling"] == op["scaling_lr"]
