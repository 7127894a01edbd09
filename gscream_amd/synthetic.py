"""Seeded synthetic scenes and cameras for tests and bench.py.

The camera helpers restate the reference's matrix conventions so that the kernels see exactly
what GScream's `Camera` hands to the rasterizer:

* `projection_matrix`  -- utils/graphics_utils.py:51-74 getProjectionMatrix (principal point in
  P[0,2], P[1,2]; P[2,2] = (zn+zf)/(zf-zn)),
* `camera_matrices`    -- scene/cameras.py:64-69: world_view_transform = W2C^T,
  full_proj_transform = W2C^T @ P^T (row-vector convention).

The scene distributions are the ones SURVEY.md section 8(d) pins for BASELINE.json's configs
(config 1: 2k Gaussians @128x128; configs 2-4: the 1008x567 / 1920x1080 slab generator).
Everything is numpy + an explicit seed; nothing here touches the GPU.
"""
import math

import numpy as np


def projection_matrix(znear, zfar, tanfovx, tanfovy, cx=0.0, cy=0.0):
    P = np.zeros((4, 4), np.float32)
    top, right = tanfovy * znear, tanfovx * znear
    P[0, 0] = 2.0 * znear / (2.0 * right)
    P[1, 1] = 2.0 * znear / (2.0 * top)
    P[0, 2] = cx
    P[1, 2] = cy
    P[3, 2] = 1.0
    P[2, 2] = (znear + zfar) / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def camera_matrices(tanfovx, tanfovy, w2c=None, cx=0.0, cy=0.0, znear=0.01, zfar=100.0):
    """Returns (viewmatrix, projmatrix, campos) in the transposed form the rasterizer expects."""
    w2c = np.eye(4, dtype=np.float32) if w2c is None else np.asarray(w2c, np.float32)
    view = np.ascontiguousarray(w2c.T)
    proj = np.ascontiguousarray((view @ projection_matrix(znear, zfar, tanfovx, tanfovy, cx, cy).T).astype(np.float32))
    campos = np.linalg.inv(view)[3, :3].astype(np.float32)
    return view, proj, np.ascontiguousarray(campos)


def random_w2c(rng, max_angle=0.35, max_shift=0.4):
    """A small random rigid transform (exercises every entry of the view matrix)."""
    ax = rng.normal(size=3)
    ax /= np.linalg.norm(ax)
    ang = rng.uniform(-max_angle, max_angle)
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    Rm = np.eye(3) + math.sin(ang) * K + (1 - math.cos(ang)) * (K @ K)
    M = np.eye(4)
    M[:3, :3] = Rm
    M[:3, 3] = rng.uniform(-max_shift, max_shift, size=3)
    return M.astype(np.float32)


def _quats(rng, P):
    q = rng.uniform(-0.5, 0.5, size=(P, 4))
    q /= np.maximum(np.linalg.norm(q, axis=1, keepdims=True), 1e-8)
    return q.astype(np.float32)


def scene_config1(seed=0, P=2000, W=128, H=128, lateral=0.55, w2c=None, cx=0.0, cy=0.0, bg=(0.1, 0.2, 0.3),
                  tanfovx=0.5, tanfovy=None):
    """SURVEY 8(d) config 1 distribution (oracle / plumbing case)."""
    rng = np.random.default_rng(seed)
    tanfovy = tanfovx * H / W if tanfovy is None else tanfovy
    z = rng.uniform(2.0, 6.0, size=P)
    x = rng.uniform(-lateral, lateral, size=P) * z
    y = rng.uniform(-lateral, lateral, size=P) * z
    cam = np.stack([x, y, z], 1)
    view, proj, campos = camera_matrices(tanfovx, tanfovy, w2c, cx, cy)
    if w2c is not None:  # place the cloud in front of the moved camera
        c2w = np.linalg.inv(np.asarray(w2c, np.float64))
        cam = cam @ c2w[:3, :3].T + c2w[:3, 3]
    return dict(
        means3D=cam.astype(np.float32),
        scales=rng.uniform(0.02, 0.17, size=(P, 3)).astype(np.float32),
        rotations=_quats(rng, P),
        opacities=rng.uniform(0.05, 0.95, size=(P, 1)).astype(np.float32),
        uncertainties=rng.uniform(0, 1, size=(P, 1)).astype(np.float32),
        colors=rng.uniform(0, 1, size=(P, 3)).astype(np.float32),
        W=W, H=H, tanfovx=float(tanfovx), tanfovy=float(tanfovy), viewmatrix=view, projmatrix=proj, campos=campos,
        bg=np.asarray(bg, np.float32), scale_modifier=1.0)


def scene_slab(seed, P, W, H, tanfovx=0.6, bg=(0.0, 0.0, 0.0), scale_mu=0.01, scale_sigma=0.6):
    """the synthetic stand-in for a SPIn-NeRF scene.  Slab z~U(1.5,8),
    15% lateral overshoot (culling, edge tiles, a few +-1.3*tan clamps), log-normal scales,
    opacity U(0,1)^2."""
    rng = np.random.default_rng(seed)
    tanfovy = tanfovx * H / W
    z = rng.uniform(1.5, 8.0, size=P)
    x = rng.uniform(-1.15, 1.15, size=P) * tanfovx * z
    y = rng.uniform(-1.15, 1.15, size=P) * tanfovy * z
    view, proj, campos = camera_matrices(tanfovx, tanfovy)
    return dict(
        means3D=np.stack([x, y, z], 1).astype(np.float32),
        scales=np.exp(rng.normal(math.log(scale_mu), scale_sigma, size=(P, 3))).astype(np.float32),
        rotations=_quats(rng, P),
        opacities=(rng.uniform(0, 1, size=(P, 1)) ** 2).astype(np.float32),
        uncertainties=rng.uniform(0, 1, size=(P, 1)).astype(np.float32),
        colors=rng.uniform(0, 1, size=(P, 3)).astype(np.float32),
        W=W, H=H, tanfovx=float(tanfovx), tanfovy=float(tanfovy), viewmatrix=view, projmatrix=proj, campos=campos,
        bg=np.asarray(bg, np.float32), scale_modifier=1.0)


def scene_surfaces(seed, P, W, H, tanfovx=0.6, bg=(0.0, 0.0, 0.0), scale_mu=0.01, scale_sigma=0.6, n_surfaces=6, floaters=0.15):
    """A cloud shaped like a TRAINED scene rather than a uniform slab: 85 % of the Gaussians lie on a few tilted, slightly wavy surfaces
    (depth jitter 0.5 % of the surface depth, opacity U(0.3, 1)), the rest are slab-like floaters.  Per-tile depth keys cluster around
    the surface depths and the front surface saturates most pixels early: the regime the per-tile sort and the depth-ordered walks see
    in training, which the uniform slab of configs 2 - 4 does not exercise.  Same camera, scales and lateral overshoot as scene_slab."""
    rng = np.random.default_rng(seed)
    tanfovy = tanfovx * H / W
    u = rng.uniform(-1.15, 1.15, size=P)
    v = rng.uniform(-1.15, 1.15, size=P)
    z0 = np.sort(rng.uniform(2.0, 7.0, size=n_surfaces))
    tilt = rng.uniform(-0.6, 0.6, size=(n_surfaces, 2))
    wav = rng.uniform(0.0, 0.15, size=n_surfaces)
    which = rng.integers(0, n_surfaces, size=P)
    z = z0[which] + tilt[which, 0] * u + tilt[which, 1] * v + wav[which] * np.sin(5.0 * u + 3.0 * v)
    z = z * (1.0 + 0.005 * rng.normal(size=P))
    fl = rng.random(P) < floaters
    z = np.where(fl, rng.uniform(1.5, 8.0, size=P), np.clip(z, 1.2, 9.0))
    x, y = u * tanfovx * z, v * tanfovy * z
    opac = np.where(fl, rng.uniform(0, 1, size=P) ** 2, rng.uniform(0.3, 1.0, size=P))
    view, proj, campos = camera_matrices(tanfovx, tanfovy)
    return dict(
        means3D=np.stack([x, y, z], 1).astype(np.float32),
        scales=np.exp(rng.normal(math.log(scale_mu), scale_sigma, size=(P, 3))).astype(np.float32),
        rotations=_quats(rng, P),
        opacities=opac.reshape(P, 1).astype(np.float32),
        uncertainties=rng.uniform(0, 1, size=(P, 1)).astype(np.float32),
        colors=rng.uniform(0, 1, size=(P, 3)).astype(np.float32),
        W=W, H=H, tanfovx=float(tanfovx), tanfovy=float(tanfovy), viewmatrix=view, projmatrix=proj, campos=campos,
        bg=np.asarray(bg, np.float32), scale_modifier=1.0)


def surface_point_cloud(seed, n_points=200_000, tanfovx=0.6, aspect=567.0 / 1008.0):
    """A synthetic stand-in for the SfM point cloud GScream initialises a SPIn-NeRF scene from (scene/dataset_readers.py ->
    GaussianModel.create_from_pcd): a forward-facing capture (camera at the origin looking down +z, COLMAP axes: y down) of a room
    corner -- floor, back wall, side wall, a table top and two round objects on it, a few percent stray points -- sampled with the
    uneven density SfM gives (denser where textured / near).  [n, 3] float64, metres; every point inside a 1.2x frustum, z in 2..10."""
    rng = np.random.default_rng(seed)
    tanfovy = tanfovx * aspect
    share = np.array([0.30, 0.25, 0.10, 0.17, 0.08, 0.06, 0.04])
    cnt = np.floor(share * n_points).astype(int)
    cnt[0] += n_points - cnt.sum()
    parts = []
    u = lambda n, a, b: rng.uniform(a, b, size=n)
    # floor: y = 1.3 (below the camera), slightly tilted; denser near the camera
    z = 2.0 + 8.0 * rng.beta(1.2, 2.0, size=cnt[0]); x = u(cnt[0], -1.2, 1.2) * tanfovx * z
    parts.append(np.stack([x, 1.3 + 0.02 * x + 0.01 * np.sin(3.0 * z), z], 1))
    # back wall at z ~ 9.5 with a shallow relief
    x = u(cnt[1], -1.2, 1.2) * tanfovx * 9.5; y = u(cnt[1], -1.2, 1.0) * tanfovy * 9.5
    parts.append(np.stack([x, y, 9.5 + 0.05 * np.sin(2.0 * x) * np.cos(3.0 * y)], 1))
    # left side wall: x = -0.9 tan z ... a plane receding from the camera
    z = u(cnt[2], 3.0, 9.5); y = u(cnt[2], -1.0, 1.0) * tanfovy * z
    parts.append(np.stack([-0.95 * tanfovx * z + 0.3, y, z], 1))
    # table top (the object the reference removes sits on it): y = 0.45, z in 3.5..5.5
    z = u(cnt[3], 3.5, 5.5); x = u(cnt[3], -1.4, 1.4)
    parts.append(np.stack([x, np.full(cnt[3], 0.45), z], 1))
    # two round objects on the table
    for k, (c, r) in zip((4, 5), (((-0.4, 0.05, 4.3), 0.40), ((0.55, 0.20, 4.8), 0.25))):
        d = rng.normal(size=(cnt[k], 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
        d[:, 2] = -np.abs(d[:, 2])          # the half facing the camera
        parts.append(np.asarray(c)[None, :] + r * d)
    # stray points (SfM outliers) anywhere in the frustum
    z = u(cnt[6], 2.0, 10.0)
    parts.append(np.stack([u(cnt[6], -1.1, 1.1) * tanfovx * z, u(cnt[6], -1.1, 1.1) * tanfovy * z, z], 1))
    pts = np.concatenate(parts, 0)
    pts += 0.002 * rng.normal(size=pts.shape)       # triangulation noise, 2 mm
    return pts


def scene_init_state(seed, W, H, n_points=200_000, K=10, tanfovx=0.6, bg=(0.0, 0.0, 0.0), device="cuda", return_model=False):
    """The frame GScream renders at ITERATION 0 of a scene: `surface_point_cloud` -> the reference's initialisation restated
    (standin_model.Model.from_pcd = scene/gaussian_model.py:295-345: voxelise at voxel_size 0.001, anchor scales
    log(sqrt(mean 3-NN dist^2)) from this package's distCUDA2, zero offsets / features, identity rotations, default-init MLPs) -> the
    fused decode at a camera at the origin -> the rasterizer-level scene dict the other generators return.  Needs the GPU rows
    (gsr_knn_mean_dist2, gsr_decode_*): there is no CPU path in this package."""
    import torch
    from . import neural_gaussians as NG
    from . import simple_knn as KN
    from . import standin_model as SM
    pts = SM.voxelize(surface_point_cloud(seed, n_points, tanfovx, H / W), 0.001)
    anchors = torch.from_numpy(pts).float().to(device)
    dist2 = torch.clamp_min(KN.distCUDA2(anchors), 0.0000001)
    model = SM.Model.from_pcd(anchors, dist2, K=K, seed=seed).to(device)
    tanfovy = tanfovx * H / W
    view, proj, campos = camera_matrices(tanfovx, tanfovy)
    cam = SM.Camera(torch.from_numpy(campos).to(device))
    model.eval()
    with torch.no_grad():
        xyz, color, opacity, unc, scaling, rot = NG.generate_neural_gaussians(cam, model, None, is_training=False)
    n = lambda t: np.ascontiguousarray(t.detach().float().cpu().numpy())
    s = dict(means3D=n(xyz), scales=n(scaling), rotations=n(rot), opacities=n(opacity).reshape(-1, 1), uncertainties=n(unc).reshape(-1, 1),
             colors=n(color), W=W, H=H, tanfovx=float(tanfovx), tanfovy=float(tanfovy), viewmatrix=view, projmatrix=proj, campos=campos,
             bg=np.asarray(bg, np.float32), scale_modifier=1.0, anchors=int(anchors.shape[0]))
    return (s, model) if return_model else s


def scene_stack(seed=5, P=1500, n_stack=700, W=112, H=71):
    """SURVEY 8(c) fixture 5: many Gaussians stacked on the centre pixel (multi-batch tile lists,
    T < 1e-4 early stop) on a non-multiple-of-16 image."""
    s = scene_config1(seed=seed, P=P, W=W, H=H)
    rng = np.random.default_rng(seed + 1000)
    z = rng.uniform(2.0, 6.0, size=n_stack)
    s["means3D"][:n_stack, 0] = (rng.normal(0, 0.002, size=n_stack) * z).astype(np.float32)
    s["means3D"][:n_stack, 1] = (rng.normal(0, 0.002, size=n_stack) * z).astype(np.float32)
    s["means3D"][:n_stack, 2] = z.astype(np.float32)
    return s


def scene_ties(seed=6, P=600, W=64, H=48):
    """SURVEY 8(c) fixture 6: duplicated depths -> tie order must be ascending Gaussian index."""
    s = scene_config1(seed=seed, P=P, W=W, H=H)
    levels = np.asarray([2.5, 3.0, 3.5, 4.0], np.float32)
    rng = np.random.default_rng(seed + 7)
    s["means3D"][:, 2] = levels[rng.integers(0, len(levels), size=P)]
    return s


def upstream_grads(seed, W, H, color=True, depth=True, unc=True):
    """Random upstream gradients dL/d{color, depth, uncertainty} (N(0,1)/N)."""
    rng = np.random.default_rng(seed)
    n = W * H
    gc = rng.normal(size=(3, H, W)).astype(np.float32) / n if color else np.zeros((3, H, W), np.float32)
    gd = rng.normal(size=(1, H, W)).astype(np.float32) / n if depth else np.zeros((1, H, W), np.float32)
    gu = rng.normal(size=(1, H, W)).astype(np.float32) / n if unc else np.zeros((1, H, W), np.float32)
    return gc, gd, gu
