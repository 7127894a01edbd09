"""Golden vectors produced by RUNNING the reference's own Python helpers in the build container
(`/root/reference/utils/graphics_utils.py`, `utils/sh_utils.py` import cleanly there; nothing from the reference is
copied -- only inputs and the outputs it computed are stored).

    python tests/golden/make_reference_vectors.py          # needs /root/reference; writes tests/golden/ref_*.npz

ref_camera.npz : getWorld2View2 / getProjectionMatrix (utils/graphics_utils.py:38-76) composed the way scene/cameras.py
                 builds a Camera (world_view_transform = W2V^T, projection_matrix = P^T, full_proj_transform = their
                 product, camera_center = inverse(world_view_transform)[3, :3]) -> pins the matrix convention every entry
                 point of the rasterizer receives and gscream_amd.synthetic.camera_matrices.
ref_sh.npz     : eval_sh (utils/sh_utils.py:57-115) for degrees 0..3 on random coefficients and unit directions -> pins
                 the SH polynomial and constants of the colour path (forward.cu:22-73 uses the same expansion).
These are the only parts of the path for which the reference can be executed here (its rasterizer is CUDA-only)."""
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def main():
    sys.path.insert(0, REF)
    from utils.graphics_utils import focal2fov, fov2focal, getProjectionMatrix, getWorld2View2
    from utils.sh_utils import eval_sh

    rng = np.random.default_rng(2024)
    cams = []
    for i in range(6):
        ax = rng.normal(size=3)
        ax /= np.linalg.norm(ax)
        ang = rng.uniform(-0.8, 0.8)
        K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
        R = np.eye(3) + math.sin(ang) * K + (1 - math.cos(ang)) * K @ K        # the dataset readers store R transposed
        T = rng.uniform(-2, 2, size=3)
        fovx, fovy = rng.uniform(0.5, 1.4), rng.uniform(0.4, 1.2)
        cx, cy = (0.0, 0.0) if i % 2 == 0 else (rng.uniform(-0.05, 0.05), rng.uniform(-0.05, 0.05))
        znear, zfar = 0.01, 100.0
        w2v = getWorld2View2(R, T, np.array([0.0, 0.0, 0.0]), 1.0)
        wv = torch.tensor(w2v).transpose(0, 1)
        proj = getProjectionMatrix(znear, zfar, fovx, fovy, cx, cy).transpose(0, 1)
        full = (wv.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0)
        center = wv.inverse()[3, :3]
        cams.append(dict(R=R, T=T, fovx=fovx, fovy=fovy, cx=cx, cy=cy, znear=znear, zfar=zfar, w2v=w2v, world_view=wv.numpy(),
                         projection=proj.numpy(), full_proj=full.numpy(), camera_center=center.numpy(),
                         focal_roundtrip=focal2fov(fov2focal(fovx, 1008), 1008)))
    np.savez_compressed(os.path.join(HERE, "ref_camera.npz"), **{f"{k}_{i}": np.asarray(c[k]) for i, c in enumerate(cams) for k in c})

    P = 200
    sh = rng.normal(0, 0.4, size=(P, 16, 3)).astype(np.float32)             # rasterizer layout [P, M, 3]
    dirs = rng.normal(size=(P, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    out = {"sh": sh, "dirs": dirs}
    for deg in range(4):
        M = (deg + 1) ** 2
        res = eval_sh(deg, torch.from_numpy(sh[:, :M, :]).transpose(1, 2), torch.from_numpy(dirs))  # sh as [..., C, M]
        out[f"eval_sh_deg{deg}"] = res.numpy()
    np.savez_compressed(os.path.join(HERE, "ref_sh.npz"), **out)
    print("wrote ref_camera.npz, ref_sh.npz")


if __name__ == "__main__":
    main()
