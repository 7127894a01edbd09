"""Generates tests/golden/*.npz: seeded inputs + the CPU oracle's outputs for them.

Run in the build container:  python tests/golden/make_golden.py
The oracle (oracle/gs_oracle.c) is a restatement of the reference, not the reference itself, and
the reference ships no golden vectors, so these fixtures pin the ORACLE (regression
pin for tests/test_oracle.py) and give the GPU tests vectors that need no oracle at run time.
They do not upgrade the oracle's status from "parity unpinned".  Cases follow SURVEY 8(c)'s list.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from gscream_amd import synthetic as S  # noqa: E402
import helpers as Hh  # noqa: E402

INPUT_KEYS = ("means3D", "scales", "rotations", "opacities", "uncertainties", "colors", "viewmatrix", "projmatrix",
              "campos", "bg", "shs", "cov3D_precomp")
SCALARS = ("W", "H", "tanfovx", "tanfovy", "scale_modifier", "sh_degree")


def cases():
    rng = np.random.default_rng(42)
    c = {}
    c["cfg1"] = (S.scene_config1(seed=0, P=600, W=96, H=80), (True, True, True))
    c["clamp"] = (S.scene_config1(seed=1, P=600, W=96, H=80, lateral=0.85), (True, True, True))
    c["odd_size"] = (S.scene_config1(seed=2, P=500, W=112, H=71), (True, True, True))
    c["rgb_only"] = (S.scene_config1(seed=3, P=600, W=96, H=80), (True, False, False))
    c["stack"] = (S.scene_stack(seed=5, P=900, n_stack=500, W=112, H=71), (True, True, True))
    c["ties"] = (S.scene_ties(seed=6, P=600, W=64, H=48), (True, True, True))
    c["moved_cam"] = (S.scene_config1(seed=7, P=600, W=96, H=80, w2c=S.random_w2c(rng), cx=0.06, cy=-0.04), (True, True, True))
    c["white_bg"] = (S.scene_config1(seed=8, P=600, W=96, H=80, bg=(1.0, 1.0, 1.0)), (True, True, True))
    culled = S.scene_config1(seed=9, P=300, W=64, H=48)
    culled["means3D"][:, 2] = np.random.default_rng(9).uniform(-1.0, 0.2, size=300).astype(np.float32)
    c["all_culled"] = (culled, (True, True, True))
    sh = S.scene_config1(seed=10, P=500, W=96, H=80, w2c=S.random_w2c(rng))
    sh["shs"] = np.random.default_rng(10).normal(0, 0.4, size=(500, 16, 3)).astype(np.float32)
    sh["sh_degree"] = 3
    del sh["colors"]
    c["sh_deg3"] = (sh, (True, True, True))
    cov = S.scene_config1(seed=11, P=500, W=96, H=80)
    A = np.random.default_rng(11).normal(0, 0.08, size=(500, 3, 3))
    Sg = A @ A.transpose(0, 2, 1) + 1e-4 * np.eye(3)
    cov["cov3D_precomp"] = np.stack([Sg[:, 0, 0], Sg[:, 0, 1], Sg[:, 0, 2], Sg[:, 1, 1], Sg[:, 1, 2], Sg[:, 2, 2]], 1).astype(np.float32)
    c["cov_precomp"] = (cov, (True, True, True))
    return c


def main():
    for name, (s, gsel) in cases().items():
        st = Hh.oracle_forward(s)
        grads = S.upstream_grads(100 + len(name), s["W"], s["H"], *gsel)
        g = Hh.oracle_backward(s, st, grads)
        out = {"in_" + k: s[k] for k in INPUT_KEYS if k in s}
        out.update({"sc_" + k: np.asarray(s[k]) for k in SCALARS if k in s})
        out.update(g_color=grads[0], g_depth=grads[1], g_unc=grads[2])
        out.update(out_color=st["out_color"], out_depth=st["out_depth"], out_unc=st["out_unc"], radii=st["radii"],
                   num_rendered=np.int64(st["num_rendered"]), point_list=st["point_list"],
                   tile_counts=(st["ranges"][:, 1] - st["ranges"][:, 0]).astype(np.uint32), n_contrib=st["n_contrib"])
        for k, v in g.items():
            out["grad_" + k] = v
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print(f"{name:12s} P={s['means3D'].shape[0]:5d} {s['W']}x{s['H']} R={st['num_rendered']:6d} "
              f"visible={(st['radii'] > 0).sum():5d} -> {os.path.getsize(path) / 1024:.0f} KiB")
    # filter known answers recorded in SURVEY.md Appendix B-6 (from a run of the reference's own kernels)
    np.savez(os.path.join(HERE, "filter_known_answers.npz"),
             points=np.array([[0, 0, 3], [0, 0, 0.1], [30, 0, 3], [1.4, 0, 3], [0, 0, 0.2]], np.float32),
             radii=np.array([7, 0, 0, 7, 0], np.int32), x=np.array([55.5, 0, 0, 107.767, 0], np.float32),
             y=np.array([35.0, 0, 0, 35.0, 0], np.float32), visible=np.array([1, 0, 1, 1, 0], np.uint8),
             W=112, H=71, tanfovx=0.5, scale=0.05)


def load(name):
    """-> (scene dict, (g_color, g_depth, g_unc), expected dict)"""
    z = np.load(os.path.join(HERE, name + ".npz"))
    s = {k[3:]: z[k] for k in z.files if k.startswith("in_")}
    for k in z.files:
        if k.startswith("sc_"):
            v = z[k]
            s[k[3:]] = int(v) if k[3:] in ("W", "H", "sh_degree") else float(v)
    exp = {k: z[k] for k in z.files if not (k.startswith("in_") or k.startswith("sc_"))}
    return s, (z["g_color"], z["g_depth"], z["g_unc"]), exp


if __name__ == "__main__":
    main()
