"""A short optimisation run of the neural-Gaussian model THROUGH the HIP rows -- the iteration train.py runs
(train.py:414-416 pop a camera, :433 anchor prefilter, :527 render with the visible mask, :535-573 RGB + depth losses, :575 backward,
:597-602 training_statis, :627 optimizer.step) with the reference's Adam groups and learning rates (scene/gaussian_model.py:350-395,
arguments/__init__.py:93-133), against images of a synthetic TEACHER scene (there is no dataset offline).

Two uses:
  * `bench.py --workload fitted`: the frame a scene renders after a few hundred optimiser steps -- Gaussians whose sizes, opacities and
    colours were shaped by gradients that went through these kernels, between the untrained iteration-0 frame (`init_state`: nothing
    saturates) and the saturated synthetic slabs.  NOT a BASELINE config; says so on the line.
  * tests: the loss falls and the PSNR rises, i.e. the gradients of the whole chain point the right way under a real optimiser.

Nothing of GScream's training logic beyond that iteration is rebuilt here: no densification decisions, no inpainting module, no
discriminator (out of scope).  Teacher: the dense surface cloud of synthetic.surface_point_cloud as small opaque splats with a
procedural texture, rendered by the rasterizer itself under no_grad; student: the state GaussianModel.create_from_pcd leaves for a
SPARSER, independently sampled cloud of the same surfaces (what SfM gives), standin_model.Model.from_pcd.  No CPU path."""
import math

import numpy as np
import torch

from . import densify_stats as DS
from . import gaussian_renderer as GR
from . import loss_utils as L
from . import simple_knn as KN
from . import standin_model as SM
from . import synthetic as S
from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer

__all__ = ["teacher_scene", "orbit_cameras", "render_teacher", "adam_groups", "expon_lr", "update_learning_rate", "fit", "scene_fitted"]


class _Pipe:
    debug, compute_cov3D_python = False, False


def _texture(p):
    """Procedural albedo in [0.05, 0.95] from world position: smooth colour fields + a fine checker, so that both the low and the high
    frequencies of the image carry signal."""
    x, y, z = p[:, 0], p[:, 1], p[:, 2]
    chk = (np.floor(x * 4.0) + np.floor(y * 4.0) + np.floor(z * 2.0)) % 2
    r = 0.5 + 0.35 * np.sin(1.7 * x + 0.3 * z) + 0.1 * (chk - 0.5)
    g = 0.5 + 0.35 * np.sin(2.3 * y + 0.9 * x + 1.0) - 0.1 * (chk - 0.5)
    b = 0.5 + 0.35 * np.cos(0.8 * z + 1.3 * y) + 0.1 * (chk - 0.5)
    return np.clip(np.stack([r, g, b], 1), 0.05, 0.95).astype(np.float32)


def teacher_scene(seed, n_points, W, H, tanfovx=0.6, device="cuda"):
    """Rasterizer-level scene dict of the teacher: every point of a dense surface cloud as an opaque isotropic splat of the local point
    spacing (mean 3-NN distance from this package's distCUDA2), coloured by `_texture`."""
    pts = SM.voxelize(S.surface_point_cloud(seed, n_points, tanfovx, H / W), 0.001)
    anchors = torch.from_numpy(pts).float().to(device)
    sp = torch.sqrt(torch.clamp_min(KN.distCUDA2(anchors), 1e-7)).clamp(max=0.05)
    P = int(anchors.shape[0])
    tanfovy = tanfovx * H / W
    view, proj, campos = S.camera_matrices(tanfovx, tanfovy)
    q = np.zeros((P, 4), np.float32)
    q[:, 0] = 1.0
    return dict(means3D=pts.astype(np.float32), scales=(sp * 0.9)[:, None].repeat(1, 3).cpu().numpy().astype(np.float32), rotations=q,
                opacities=np.full((P, 1), 0.95, np.float32), uncertainties=np.zeros((P, 1), np.float32), colors=_texture(pts),
                W=W, H=H, tanfovx=float(tanfovx), tanfovy=float(tanfovy), viewmatrix=view, projmatrix=proj, campos=campos,
                bg=np.zeros(3, np.float32), scale_modifier=1.0)


def orbit_cameras(V, W, H, tanfovx, centre, angle=0.04, device="cuda"):
    """V cameras on a small orbit about `centre` (yaw / pitch of `angle` rad around the reference camera at the origin): the forward-facing
    capture of a SPIn-NeRF scene in miniature.  Camera 0 of the list is the reference camera itself."""
    tanfovy = tanfovx * H / W
    cams = []
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    for i in range(V):
        a = 0.0 if i == 0 else angle
        yaw, pitch = a * math.cos(2 * math.pi * i / max(V - 1, 1)), a * math.sin(2 * math.pi * i / max(V - 1, 1))
        cy, sy, cp, sp = math.cos(yaw), math.sin(yaw), math.cos(pitch), math.sin(pitch)
        R = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]]) @ np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
        M = np.eye(4)
        M[:3, :3] = R
        M[:3, 3] = centre - R @ centre
        view, proj, campos = S.camera_matrices(tanfovx, tanfovy, np.linalg.inv(M).astype(np.float32))
        cams.append(SM.Camera(t(campos), image_height=H, image_width=W, FoVx=2 * math.atan(tanfovx), FoVy=2 * math.atan(tanfovy),
                              world_view_transform=t(view), full_proj_transform=t(proj)))
    return cams


def render_teacher(ts, cams, device="cuda"):
    """-> (images [V,3,H,W], depths [V,1,H,W]) of the teacher scene from every camera, rendered by the rasterizer under no_grad."""
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    m, sc, rot, op, unc, col = (t(ts[k]) for k in ("means3D", "scales", "rotations", "opacities", "uncertainties", "colors"))
    bg = t(ts["bg"])
    imgs, deps = [], []
    with torch.no_grad():
        for cam in cams:
            rs = GaussianRasterizationSettings(image_height=ts["H"], image_width=ts["W"], tanfovx=ts["tanfovx"], tanfovy=ts["tanfovy"], bg=bg,
                                               scale_modifier=1.0, viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform,
                                               sh_degree=1, campos=cam.camera_center, prefiltered=False, debug=False)
            color, depth, _u, _r = GaussianRasterizer(raster_settings=rs)(means3D=m, means2D=torch.zeros_like(m), opacities=op, uncertainties=unc,
                                                                          colors_precomp=col, scales=sc, rotations=rot)
            imgs.append(color.clamp(0, 1))
            deps.append(depth)
    return torch.stack(imgs), torch.stack(deps)


def adam_groups(model, spatial_lr_scale=1.0):
    """The reference's parameter groups and learning rates (scene/gaussian_model.py:376-390, arguments/__init__.py:96-133; the
    schedules' initial values; fit() applies the reference's per-iteration schedule, update_learning_rate below).  The anchors' rate is 0
    there, so they are left out."""
    return [{"params": [model._offset], "lr": 0.01 * spatial_lr_scale, "name": "offset"},
            {"params": [model._anchor_feat], "lr": 0.0075, "name": "anchor_feat"},
            {"params": [model._scaling], "lr": 0.007, "name": "scaling"},
            {"params": model.mlp_opacity.parameters(), "lr": 0.002, "name": "mlp_opacity"},
            {"params": model.mlp_uncertainty.parameters(), "lr": 0.002, "name": "mlp_uncertainty"},
            {"params": model.mlp_cov.parameters(), "lr": 0.004, "name": "mlp_cov"},
            {"params": model.mlp_color.parameters(), "lr": 0.008, "name": "mlp_color"}]


# (init, final, max_steps) of the groups update_learning_rate() reschedules every iteration (scene/gaussian_model.py:412-439, 460-483;
# arguments/__init__.py:101-133; lr_delay_steps is 0 there, so the delay factor is 1)
LR_SCHEDULES = {"offset": (0.01, 0.0001, 30_000), "mlp_opacity": (0.002, 0.00002, 30_000), "mlp_uncertainty": (0.002, 0.00002, 30_000),
                "mlp_cov": (0.004, 0.004, 30_000), "mlp_color": (0.008, 0.00005, 30_000)}


def expon_lr(step, lr_init, lr_final, max_steps):
    """utils/general_utils.py:104-136 get_expon_lr_func without a delay: log-linear interpolation from lr_init (step 0) to lr_final
    (step >= max_steps)."""
    if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
        return 0.0
    t = min(max(step / max_steps, 0.0), 1.0)
    return math.exp(math.log(lr_init) * (1.0 - t) + math.log(lr_final) * t)


def update_learning_rate(opt, iteration, spatial_lr_scale=1.0):
    """GaussianModel.update_learning_rate (scene/gaussian_model.py:460-483) for the groups of adam_groups()."""
    for g in opt.param_groups:
        sch = LR_SCHEDULES.get(g.get("name"))
        if sch is not None:
            scale = spatial_lr_scale if g["name"] == "offset" else 1.0
            g["lr"] = expon_lr(iteration, sch[0] * scale, sch[1] * scale, sch[2])


def psnr(a, b):
    return float(-10.0 * torch.log10(torch.mean((a - b) ** 2).clamp_min(1e-12)))


def fit(model, cams, gts, gdepths, iters, seed=0, lambda_dssim=0.2, depth_weight=1.0, bg=None, log_every=0):
    """`iters` iterations of train.py's loop on the HIP rows.  Returns {"loss": [...], "psnr_first", "psnr_last", "ms_per_iteration"}
    (PSNR of camera 0, rendered in eval mode before and after)."""
    dev = gts.device
    bg = torch.zeros(3, device=dev) if bg is None else bg
    # (eps: scene/gaussian_model.py:392.  fused: one kernel per parameter group instead of eight foreach kernels -- the same update rule;
    # seven groups x eight launches were 0.5 ms of host time per iteration of a loop that is host-paced)
    try:
        opt = torch.optim.Adam(adam_groups(model), lr=0.0, eps=1e-15, fused=True)
    except (RuntimeError, TypeError):
        opt = torch.optim.Adam(adam_groups(model), lr=0.0, eps=1e-15)
    rng = np.random.default_rng(seed)
    stack = []
    ones = torch.ones_like(gdepths[0])

    def eval_psnr():
        model.eval()
        with torch.no_grad():
            vis, _x, _y = GR.prefilter_position2D(cams[0], model, _Pipe, bg)
            out = GR.render(cams[0], model, _Pipe, bg, visible_mask=vis)["render"]
        model.train()
        return psnr(out.clamp(0, 1), gts[0])

    model.train()
    p0 = eval_psnr()
    losses = []
    t0 = torch.cuda.Event(enable_timing=True)
    t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for it in range(iters):
        update_learning_rate(opt, it + 1)  # train.py:408 (iterations count from 1)
        if not stack:
            stack = list(rng.permutation(len(cams)))   # train.py:414-416
        v = int(stack.pop())
        vis, _x, _y = GR.prefilter_position2D(cams[v], model, _Pipe, bg)
        pkg = GR.render(cams[v], model, _Pipe, bg, visible_mask=vis, retain_grad=True)
        loss = L.rgb_loss(pkg["render"], gts[v], None, lambda_dssim, 1.0)
        if depth_weight:
            loss = loss + depth_weight * L.depth_loss(pkg["render_depth"], gdepths[v], lsq_mask=ones, lambda_l1=1.0, lambda_smooth=1.0)
        loss.backward()
        with torch.no_grad():
            DS.training_statis(model, pkg["viewspace_points"], pkg["neural_opacity"], pkg["visibility_filter"], pkg["selection_mask"], vis)
        opt.step()
        opt.zero_grad(set_to_none=True)
        if log_every and (it % log_every == 0 or it == iters - 1):
            losses.append(float(loss.detach()))
    t1.record()
    torch.cuda.synchronize()
    return {"loss": losses, "psnr_first": p0, "psnr_last": eval_psnr(), "iterations": iters,
            "ms_per_iteration": t0.elapsed_time(t1) / max(iters, 1)}


def scene_fitted(seed, W, H, iters=400, n_student=200_000, n_teacher=600_000, V=16, tanfovx=0.6, K=10, device="cuda", return_info=False):
    """The rasterizer-level scene dict of the student's frame at the reference camera after `iters` optimiser steps (decoded in eval
    mode, like scene_init_state), + what the run did."""
    from . import neural_gaussians as NG
    ts = teacher_scene(seed + 100, n_teacher, W, H, tanfovx, device)
    centre = ts["means3D"].astype(np.float64).mean(0)
    cams = orbit_cameras(V, W, H, tanfovx, centre, device=device)
    gts, gdepths = render_teacher(ts, cams, device)
    pts = SM.voxelize(S.surface_point_cloud(seed, n_student, tanfovx, H / W), 0.001)
    anchors = torch.from_numpy(pts).float().to(device)
    model = SM.Model.from_pcd(anchors, torch.clamp_min(KN.distCUDA2(anchors), 0.0000001), K=K, seed=seed).to(device)
    info = fit(model, cams, gts, gdepths, iters, seed=seed, log_every=max(1, iters // 8))
    model.eval()
    with torch.no_grad():
        xyz, color, opacity, unc, scaling, rot = NG.generate_neural_gaussians(cams[0], model, None, is_training=False)
    n = lambda t: np.ascontiguousarray(t.detach().float().cpu().numpy())
    view, proj, campos = S.camera_matrices(tanfovx, tanfovx * H / W)
    s = dict(means3D=n(xyz), scales=n(scaling), rotations=n(rot), opacities=n(opacity).reshape(-1, 1), uncertainties=n(unc).reshape(-1, 1),
             colors=n(color), W=W, H=H, tanfovx=float(tanfovx), tanfovy=float(tanfovx * H / W), viewmatrix=view, projmatrix=proj, campos=campos,
             bg=np.zeros(3, np.float32), scale_modifier=1.0, anchors=int(anchors.shape[0]))
    info.update(anchors=int(anchors.shape[0]), gaussians=int(xyz.shape[0]), teacher_gaussians=int(ts["means3D"].shape[0]), views=V)
    return (s, info) if return_info else s
