"""Golden values for gscream_amd/fit.py's optimiser groups, produced by RUNNING the reference's own argument classes.

    python tests/golden/make_reference_vectors4.py        # needs /root/reference

/root/reference/arguments/__init__.py is imported in place (it needs argparse only); `OptimizationParams(parser)` and
`ModelParams(parser)` are instantiated exactly as train.py does (:994-996) and the DEFAULTS they register are stored:
the learning rates fit.adam_groups() uses (scene/gaussian_model.py:376-390 reads them by these names), the loss weights of
train.py:535-573 that fit.fit() uses (lambda_dssim), the per-iteration learning rates of the scheduled groups (the reference's
get_expon_lr_func configured as training_setup does, evaluated at a few iterations), and the model sizes the stand-in model is built with (feat_dim, n_offsets,
voxel_size).  Nothing of the reference is copied: names and numbers only (tests/golden/ref_optim.json)."""
import json
import os
import sys
from argparse import ArgumentParser

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def main():
    sys.path.insert(0, REF)
    import arguments as A  # noqa: E402
    parser = ArgumentParser()
    op = A.OptimizationParams(parser)
    lp = A.ModelParams(parser)
    keep_op = [k for k in vars(op) if k.endswith(("_lr", "_lr_init", "_lr_final", "_lr_max_steps", "_lr_delay_mult")) or k in ("lambda_dssim", "iterations")]
    keep_lp = [k for k in ("feat_dim", "n_offsets", "voxel_size", "update_depth", "use_feat_bank", "sh_degree") if hasattr(lp, k)]
    # the learning-rate schedules update_learning_rate() applies per iteration (scene/gaussian_model.py:412-439,460-483): the reference's
    # own get_expon_lr_func, configured as training_setup configures it, evaluated at a few iterations
    import utils.general_utils as GU  # noqa: E402
    steps = [1, 2, 100, 400, 1600, 7000, 30000, 40000]
    sched = {}
    for name, pre in (("offset", "offset"), ("mlp_opacity", "mlp_opacity"), ("mlp_uncertainty", "mlp_uncertainty"), ("mlp_cov", "mlp_cov"),
                      ("mlp_color", "mlp_color")):
        fn = GU.get_expon_lr_func(lr_init=getattr(op, pre + "_lr_init"), lr_final=getattr(op, pre + "_lr_final"),
                                  lr_delay_mult=getattr(op, pre + "_lr_delay_mult"), max_steps=getattr(op, pre + "_lr_max_steps"))
        sched[name] = [float(fn(st)) for st in steps]
    out = {"schedule_steps": steps, "schedules": sched, "OptimizationParams": {k: getattr(op, k) for k in sorted(keep_op)}, "ModelParams": {k: getattr(lp, k) for k in keep_lp},
           "made_by": "tests/golden/make_reference_vectors4.py (the reference's arguments/__init__.py, imported in place)"}
    with open(os.path.join(HERE, "ref_optim.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
