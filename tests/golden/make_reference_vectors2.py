"""More golden vectors produced by RUNNING the reference's own Python in the build container.
Nothing of the reference is copied: its modules are imported in place from /root/reference, executed on seeded inputs,
and only inputs + the outputs it computed are stored (tests/golden/ref_*.npz).

    python tests/golden/make_reference_vectors2.py        # needs /root/reference

Modules the exercised functions never touch are replaced by stubs that RAISE on any use (kornia.metrics, plyfile,
simple_knn._C, torch_scatter, bidirectional_cross_attention, scene.stylegan2, diff_gaussian_rasterization for the
renderer import); package `scene` is entered without running its __init__ (dataset readers need PIL / plyfile).

ref_loss.npz     utils/loss_utils.py: l1_loss, l1_loss_masked, ssim, ssim_masked (:26-30,131-190), compute_scale_and_shift
                 (:77-104), gradient_loss (:58-74) -- values and autograd gradients, float64, at 71x112, 142x252 and a
                 567x1008 case (values + gradient samples), and their train.py:535-573 compositions.
ref_decode.npz   gaussian_renderer.generate_neural_gaussians (:18-102) run on the stand-in parameter container
                 (gscream_amd/standin_model.py) in float64: outputs, mask, and the gradients of all 16 MLP tensors +
                 anchor / feature / offset / scaling through autograd; K = 10 and 4, with and without a visibility mask,
                 and the use_feat_bank branch (:39-49).
ref_stats.npz    GaussianModel.training_statis (scene/gaussian_model.py:730-757) as an unbound function on a stand-in.
ref_rotation.npz utils/general_utils.py build_rotation / build_scaling_rotation / strip_symmetric (:125-160) ->
                 Sigma_3D = L L^T, the covariance convention computeCov3D (forward.cu:120-154) must reproduce.
ref_wrapper.npz  the reference's diff_gaussian_rasterization/__init__.py imported with a RECORDING `_C`: the 20
                 forward / 23 backward argument slots and the routing of the 9 returned gradients to the 10 inputs
                 (:63-84,125-147,174-185), plus the filter entry points' argument slots (:213,265,294).
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)


class _RaisingStub(types.ModuleType):
    """A module whose every attribute is a callable that raises when CALLED (so `from x import y` works at import
    time, but nothing of the stub can take part in a computation)."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        mod = self.__name__

        class _Bomb:
            def __init__(self, *a, **k):
                raise RuntimeError(f"stub {mod}.{name} was used: the fixture would not be the reference's arithmetic")
        _Bomb.__name__ = name
        return _Bomb


def _stub(*names):
    for n in names:
        parts = n.split(".")
        for i in range(1, len(parts) + 1):
            full = ".".join(parts[:i])
            if full not in sys.modules:
                m = _RaisingStub(full)
                m.__path__ = []
                sys.modules[full] = m


def import_reference():
    sys.path.insert(0, REF)
    _stub("kornia", "kornia.metrics", "plyfile", "simple_knn", "simple_knn._C", "torch_scatter",
          "bidirectional_cross_attention", "torch_utils", "torch_utils.ops")
    sys.modules["kornia"].metrics = sys.modules["kornia.metrics"]
    # package `scene` without its __init__ (which pulls the dataset readers): submodules load from the directory
    scene = types.ModuleType("scene")
    scene.__path__ = [os.path.join(REF, "scene")]
    sys.modules["scene"] = scene
    _stub("scene.stylegan2")
    dgr = _RaisingStub("diff_gaussian_rasterization")   # the renderer module imports the two class names; never called here
    sys.modules["diff_gaussian_rasterization"] = dgr
    import utils.loss_utils as LU
    import utils.general_utils as GU
    import gaussian_renderer as GR
    from scene.gaussian_model import GaussianModel
    return LU, GU, GR, GaussianModel


# ---------------------------------------------------------------------------------------------------------------------
def loss_inputs(seed, C, H, W):
    """Shared with the tests: seeded inputs (not stored for the large case)."""
    rng = np.random.default_rng(seed)
    gt = rng.random((C, H, W)).astype(np.float32)
    img = np.clip(gt + 0.15 * rng.standard_normal(gt.shape), 0.0, 1.0).astype(np.float32)
    w = ((rng.random((1, H, W)) > 0.4).astype(np.float32) * np.float32(0.75) + np.float32(0.25))
    return img, gt, w


def depth_inputs(seed, H, W):
    rng = np.random.default_rng(seed)
    y = (rng.random((1, H, W)) * 5 + 1).astype(np.float32)
    d = (0.6 * y + 0.4 + 0.08 * rng.standard_normal(y.shape)).astype(np.float32)
    m = (rng.random(y.shape) > 0.3).astype(np.float32)
    fg = np.zeros_like(m)
    fg[:, H // 4: 3 * H // 4, W // 3: 2 * W // 3] = 1.0
    return d, y, m, fg


LOSS_SIZES = {"s": (71, 112), "m": (142, 252), "l": (567, 1008)}
GRAD_SAMPLE_STRIDE = {"s": 1, "m": 7, "l": 37}  # stored gradient elements (flattened, every n-th, as float32) per size


def loss_vectors(LU):
    out = {}
    t64 = lambda a: torch.from_numpy(a).double()
    for tag, (H, W) in LOSS_SIZES.items():
        img, gt, w = loss_inputs(100 + H, 3, H, W)
        for name, fn, args in (("l1", LU.l1_loss, ()), ("l1m", LU.l1_loss_masked, (t64(w),)),
                               ("ssim", LU.ssim, ()), ("ssimm", LU.ssim_masked, (t64(w),))):
            x = t64(img).requires_grad_(True)
            v = fn(x, t64(gt), *args)
            v.backward()
            out[f"{tag}_{name}"] = np.float64(v.item())
            g = x.grad.numpy()
            out[f"{tag}_{name}_g"] = g.reshape(-1)[::GRAD_SAMPLE_STRIDE[tag]].astype(np.float32)
            out[f"{tag}_{name}_gsum"], out[f"{tag}_{name}_gnorm"] = np.float64(g.sum()), np.float64(np.sqrt((g * g).sum()))
        # the RGB composition of train.py:538-545 (reference view, fg term included: lr 1.0, lr_fg 20.0, lambda_dssim 0.2)
        x = t64(img).requires_grad_(True)
        lam, lr, lr_fg, m = 0.2, 1.0, 20.0, t64(w)
        Ll1 = LU.l1_loss(x, t64(gt))
        loss = lr * ((1.0 - lam) * Ll1 + lam * (1.0 - LU.ssim(x, t64(gt))))
        loss = loss + (lr_fg - lr) * ((1.0 - lam) * LU.l1_loss_masked(x, t64(gt), m) + lam * (1.0 - LU.ssim_masked(x, t64(gt), m)))
        loss.backward()
        out[f"{tag}_rgb_refview"] = np.float64(loss.item())
        g = x.grad.numpy()
        out[f"{tag}_rgb_refview_g"] = g.reshape(-1)[::GRAD_SAMPLE_STRIDE[tag]].astype(np.float32)
        # depth terms, train.py:548-561 (reference view with the foreground term) and :563-573 (other view)
        d, y, mk, fg = depth_inputs(200 + H, H, W)
        for view in ("ref", "other"):
            dd = t64(d).requires_grad_(True)
            scale, shift = LU.compute_scale_and_shift(dd, t64(y), t64(mk))
            out[f"{tag}_depth_{view}_scale_signed"], out[f"{tag}_depth_{view}_shift"] = np.float64(scale.item()), np.float64(shift.item())
            scale = torch.abs(scale)
            aligned = scale.view(-1, 1, 1) * dd + shift.view(-1, 1, 1)
            if view == "ref":
                lr, lr_fg, sm = 1.0, 100.0, 1.0
                loss = lr * LU.l1_loss(aligned, t64(y)) + (lr_fg - lr) * LU.l1_loss_masked(aligned, t64(y), t64(fg))
                gm = torch.ones_like(t64(mk))
            else:
                lr, sm = 0.1, 0.1
                loss = lr * LU.l1_loss_masked(aligned, t64(y), t64(mk))
                gm = t64(mk)
            parts = []
            for k in range(4):
                step = pow(2, k)
                gl = LU.gradient_loss(aligned[:, ::step, ::step], t64(y)[:, ::step, ::step], gm[:, ::step, ::step])
                parts.append(float(gl))
                loss = loss + 0.5 * sm * gl
            loss.backward()
            out[f"{tag}_depth_{view}"] = np.float64(loss.item())
            out[f"{tag}_depth_{view}_gradient_losses"] = np.asarray(parts)
            g = dd.grad.numpy()
            out[f"{tag}_depth_{view}_g"] = g.reshape(-1)[::GRAD_SAMPLE_STRIDE[tag]].astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "ref_loss.npz"), **out)
    print("ref_loss.npz", len(out), "arrays")


# ---------------------------------------------------------------------------------------------------------------------
DECODE_CASES = {"k10": dict(N=160, K=10, seed=3, vis=False, bank=False), "k4_vis": dict(N=220, K=4, seed=4, vis=True, bank=False),
                "k10_bank": dict(N=120, K=10, seed=5, vis=True, bank=True)}
DECODE_OUT = ("xyz", "color", "opacity", "uncertainty", "scaling", "rot")


def decode_case_inputs(c):
    """Stand-in model (float64, values rounded to float32 so the fp32 kernels see exactly the same parameters), camera
    centre, visibility mask and the fixed weights of the scalar loss.  Shared with the tests."""
    from gscream_amd import standin_model as SM
    torch.manual_seed(c["seed"])
    m = SM.Model(c["N"], c["K"], seed=c["seed"], dtype=torch.float64, spread=1.5, use_feat_bank=c["bank"])
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(p.float().double())
    cam = SM.Camera(torch.tensor([0.3, -0.2, -5.0], dtype=torch.float64))
    g = torch.Generator().manual_seed(c["seed"] + 50)
    vis = (torch.rand(c["N"], generator=g) > 0.35) if c["vis"] else None
    return m, cam, vis


def decode_loss(outs, seed):
    """Scalar with fixed pseudo-random weights on every output element (so every gradient path is exercised)."""
    g = torch.Generator().manual_seed(seed + 99)
    return sum((o * torch.randn(o.shape, generator=g, dtype=torch.float64).to(o.device, o.dtype)).sum() for o in outs)


def decode_vectors(GR):
    out = {}
    for name, c in DECODE_CASES.items():
        m, cam, vis = decode_case_inputs(c)
        res = GR.generate_neural_gaussians(cam, m, vis, is_training=True)
        xyz, color, opacity, unc, scaling, rot, nop, mask = res
        loss = decode_loss(res[:6], c["seed"])
        params = dict(m.named_parameters())
        grads = torch.autograd.grad(loss, list(params.values()), allow_unused=True)
        for k, v in zip(DECODE_OUT, (xyz, color, opacity, unc, scaling, rot)):
            out[f"{name}_{k}"] = v.detach().numpy()
        out[f"{name}_neural_opacity"], out[f"{name}_mask"] = nop.detach().numpy(), mask.numpy()
        for k, g in zip(params, grads):
            out[f"{name}_grad_{k}"] = np.zeros(tuple(params[k].shape)) if g is None else g.numpy()
        ev = GR.generate_neural_gaussians(cam, m, vis, is_training=False)
        assert len(ev) == 6
        print(name, "Gaussians", xyz.shape[0], "of", (int(vis.sum()) if vis is not None else c["N"]) * c["K"])
    np.savez_compressed(os.path.join(HERE, "ref_decode.npz"), **out)


# ---------------------------------------------------------------------------------------------------------------------
def stats_inputs(seed=7, N=150, K=10):
    rng = np.random.default_rng(seed)
    anchor_vis = rng.random(N) > 0.3
    Nv = int(anchor_vis.sum())
    opacity = rng.normal(0.1, 0.5, size=(Nv * K, 1)).astype(np.float32)
    sel = (opacity > 0).reshape(-1)
    M = int(sel.sum())
    update_filter = rng.random(M) > 0.25
    grad = rng.normal(size=(M, 3)).astype(np.float32)
    acc0 = dict(opacity_accum=rng.random((N, 1)).astype(np.float32), anchor_demon=rng.integers(0, 5, (N, 1)).astype(np.float32),
                offset_gradient_accum=rng.random((N * K, 1)).astype(np.float32), offset_denom=rng.integers(0, 5, (N * K, 1)).astype(np.float32))
    return anchor_vis, opacity, sel, update_filter, grad, acc0


def stats_vectors(GaussianModel):
    anchor_vis, opacity, sel, uf, grad, acc0 = stats_inputs()
    self = types.SimpleNamespace(n_offsets=10, **{k: torch.from_numpy(v.copy()) for k, v in acc0.items()})
    vpt = types.SimpleNamespace(grad=torch.from_numpy(grad))
    for _ in range(2):  # two accumulating iterations
        GaussianModel.training_statis(self, vpt, torch.from_numpy(opacity), torch.from_numpy(uf), torch.from_numpy(sel), torch.from_numpy(anchor_vis))
    np.savez_compressed(os.path.join(HERE, "ref_stats.npz"), **{k: getattr(self, k).numpy() for k in acc0})
    print("ref_stats.npz")


# ---------------------------------------------------------------------------------------------------------------------
def rotation_vectors(GU):
    rng = np.random.default_rng(31)
    P = 64
    q = rng.normal(size=(P, 4)).astype(np.float32)
    q[:8] *= 3.0                      # not normalised: build_rotation normalises, computeCov3D does NOT (callers do, gaussian_renderer :91)
    s = np.exp(rng.normal(-2, 0.7, size=(P, 3))).astype(np.float32)
    qt, st = torch.from_numpy(q), torch.from_numpy(s)
    # build_rotation / build_scaling_rotation allocate their result with device='cuda' (general_utils.py:137,158): there is
    # no GPU in the build container, so that one allocation is redirected to the CPU; every arithmetic line is the reference's
    real_zeros = torch.zeros
    GU.torch.zeros = lambda *a, **k: real_zeros(*a, **{kk: v for kk, v in k.items() if kk != "device"})
    try:
        R = GU.build_rotation(qt)
        L = GU.build_scaling_rotation(1.0 * st, qt)
        cov = GU.strip_symmetric(L @ L.transpose(1, 2))      # scene/gaussian_model.py:44-48 covariance from scaling + rotation
    finally:
        GU.torch.zeros = real_zeros
    np.savez_compressed(os.path.join(HERE, "ref_rotation.npz"), quat=q, scale=s, R=R.numpy(), cov6=cov.numpy(),
                        quat_normalised=(qt / qt.norm(dim=1, keepdim=True)).numpy())
    print("ref_rotation.npz")


# ---------------------------------------------------------------------------------------------------------------------
def wrapper_vectors():
    """Import the reference wrapper under a private name with a recording `_C`."""
    d = os.path.join(REF, "submodules", "diff-gaussian-rasterization", "diff_gaussian_rasterization")
    rec = types.ModuleType("ref_dgr._C")
    calls = {}
    P, H, W = 5, 33, 47
    names = {}

    def ident(a):
        if isinstance(a, torch.Tensor):
            for k, v in names.items():
                if isinstance(v, torch.Tensor) and (a is v or (a.shape == v.shape and a.numel() > 0 and torch.equal(a.detach(), v.detach()))):
                    return k
            return f"<tensor{tuple(a.shape)}>" if a.numel() else "<empty>"
        for k, v in names.items():
            if not isinstance(v, torch.Tensor) and type(v) is type(a) and v == a:
                return k
        return repr(a)

    def rasterize_gaussians(*args):
        calls["forward"] = [ident(a) for a in args]
        names.update(geomBuffer=torch.full((7,), 1, dtype=torch.uint8), binningBuffer=torch.full((8,), 2, dtype=torch.uint8),
                     imgBuffer=torch.full((9,), 3, dtype=torch.uint8), radii=torch.arange(P, dtype=torch.int32) + 1, num_rendered=1234)
        return (names["num_rendered"], torch.zeros(3, H, W), torch.zeros(1, H, W), torch.zeros(1, H, W), names["radii"],
                names["geomBuffer"], names["binningBuffer"], names["imgBuffer"])

    ret_names = ("dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_duncertainty", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations")
    ret_shapes = ((P, 3), (P, 3), (P, 1), (P, 1), (P, 3), (P, 6), (P, 4, 3), (P, 3), (P, 4))

    def rasterize_gaussians_backward(*args):
        calls["backward"] = [ident(a) for a in args]
        return tuple(torch.full(s, float(i + 1)) for i, s in enumerate(ret_shapes))   # rasterize_points.cu:210 return order

    def filt(name, n_out):
        def f(*args):
            calls[name] = [ident(a) for a in args]
            return torch.zeros(P, dtype=torch.int32) if n_out == 1 else tuple(torch.zeros(P) for _ in range(n_out))
        return f
    rec.rasterize_gaussians, rec.rasterize_gaussians_backward = rasterize_gaussians, rasterize_gaussians_backward
    rec.rasterize_aussians_filter, rec.rasterize_aussians_filter_position2D = filt("visible_filter", 1), filt("position2D_filter", 3)
    rec.mark_visible = filt("mark_visible", 1)
    sys.modules["ref_dgr._C"] = rec
    spec = importlib.util.spec_from_file_location("ref_dgr", os.path.join(d, "__init__.py"), submodule_search_locations=[d])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["ref_dgr"] = mod
    spec.loader.exec_module(mod)

    g = torch.Generator().manual_seed(1)
    r = lambda *s: torch.rand(*s, generator=g)
    names.update(bg=r(3), viewmatrix=r(4, 4), projmatrix=r(4, 4) + 1, campos=r(3) + 2, scale_modifier=0.77, tanfovx=0.51, tanfovy=0.37,
                 image_height=H, image_width=W, sh_degree=2, prefiltered=True, debug=False)
    rs = mod.GaussianRasterizationSettings(**{k: names[k] for k in mod.GaussianRasterizationSettings._fields})
    out = {"settings_fields": np.array(mod.GaussianRasterizationSettings._fields)}
    # the way gaussian_renderer.render() calls it (colors + scales/rotations), and the SH / cov3D_precomp way
    for variant in ("colors_scales", "sh_cov"):
        inp = dict(means3D=r(P, 3).requires_grad_(True), means2D=torch.zeros(P, 3, requires_grad=True),
                   opacities=r(P, 1).requires_grad_(True), uncertainties=(r(P, 1) + 3).requires_grad_(True))
        if variant == "colors_scales":
            inp.update(colors_precomp=(r(P, 3) + 4).requires_grad_(True), scales=(r(P, 3) + 5).requires_grad_(True), rotations=r(P, 4).requires_grad_(True))
        else:
            inp.update(shs=r(P, 4, 3).requires_grad_(True), cov3D_precomp=r(P, 6).requires_grad_(True))
        for k in ("sh", "colors_precomp", "scales", "rotations", "cov3Ds_precomp"):
            names.pop(k, None)
        names.update({("sh" if k == "shs" else "cov3Ds_precomp" if k == "cov3D_precomp" else k): v for k, v in inp.items()})
        rast = mod.GaussianRasterizer(raster_settings=rs)
        color, depth, unc, radii = rast(**inp)
        names.update(grad_out_color=torch.full((3, H, W), 0.25), grad_out_depth=torch.full((1, H, W), 0.5), grad_out_uncertainty=torch.full((1, H, W), 0.75))
        torch.autograd.backward([color, depth, unc], [names["grad_out_color"], names["grad_out_depth"], names["grad_out_uncertainty"]])
        out[f"{variant}_forward_slots"] = np.array(calls["forward"])
        out[f"{variant}_backward_slots"] = np.array(calls["backward"])
        out[f"{variant}_returns"] = np.array(["color", "depth", "uncertainty", "radii"])
        # which of the 9 native gradients landed in each input's .grad (0 = none)
        route = {k: (0 if v.grad is None else int(v.grad.reshape(-1)[0].item())) for k, v in inp.items()}
        out[f"{variant}_grad_route_inputs"] = np.array(list(route.keys()))
        out[f"{variant}_grad_route_native_index"] = np.array([route[k] for k in route])
        assert all(v.grad is None or bool((v.grad == v.grad.reshape(-1)[0]).all()) for v in inp.values())
    out["native_backward_returns"] = np.array(ret_names)
    names.update(means3D=r(P, 3), scales=r(P, 3) + 5, rotations=r(P, 4))
    for k in ("cov3Ds_precomp", "sh", "colors_precomp"):
        names.pop(k, None)
    rast.visible_filter(names["means3D"], names["scales"], names["rotations"])
    rast.position2D_filter(names["means3D"], names["scales"], names["rotations"])
    names["positions"] = names.pop("means3D")
    rast.markVisible(names["positions"])
    for k in ("visible_filter", "position2D_filter", "mark_visible"):
        out[f"{k}_slots"] = np.array(calls[k])
    # the two argument-combination errors of GaussianRasterizer.forward (:224-228)
    errs = []
    for kw in (dict(), dict(shs=r(P, 4, 3), colors_precomp=r(P, 3)), dict(colors_precomp=r(P, 3)), dict(colors_precomp=r(P, 3), scales=r(P, 3), rotations=r(P, 4), cov3D_precomp=r(P, 6))):
        try:
            rast(r(P, 3), r(P, 3), r(P, 1), r(P, 1), **kw)
            errs.append("")
        except Exception as e:  # noqa: BLE001
            errs.append(str(e))
    out["forward_errors"] = np.array(errs)
    np.savez_compressed(os.path.join(HERE, "ref_wrapper.npz"), **out)
    for k in ("colors_scales_forward_slots", "colors_scales_backward_slots", "colors_scales_grad_route_inputs", "colors_scales_grad_route_native_index",
              "sh_cov_grad_route_native_index", "visible_filter_slots", "mark_visible_slots", "forward_errors"):
        print(k, list(out[k]))


def main():
    wrapper_vectors()
    LU, GU, GR, GaussianModel = import_reference()
    rotation_vectors(GU)
    stats_vectors(GaussianModel)
    decode_vectors(GR)
    loss_vectors(LU)


if __name__ == "__main__":
    main()
