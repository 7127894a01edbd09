"""Golden vectors for SURVEY row a18 / the filter call sites, produced by RUNNING the reference's own callers.

    python tests/golden/make_reference_vectors3.py        # needs /root/reference

/root/reference/gaussian_renderer/__init__.py is imported in place (same stub discipline as make_reference_vectors2.py) and its
`GaussianRasterizationSettings` / `GaussianRasterizer` names are bound to RECORDING classes (class level, not `_C` level), so what
is stored is exactly what `render()` (:104-179), `prefilter_voxel()` (:190-243) and `prefilter_position2D()` (:248-302) hand to
the rasterizer and give back to train.py (:433,527,759-760): constructor keywords, the twelve settings fields (scalars by value,
tensors by WHICH camera / caller object they are), the method called, its keyword names in call order, and per argument dtype,
shape, contiguity, requires_grad, is_leaf; the keys, dtypes and shapes of the returned dict; and whether `viewspace_points.grad`
exists after a backward (the densification statistics read it, scene/gaussian_model.py:755).  Nothing of the reference is copied:
only these facts are stored (tests/golden/ref_render_call.npz).

The model is the stand-in parameter container (gscream_amd/standin_model.py, float32) and a camera built like
scene/cameras.py:58-69; the reference's three `device="cuda"` allocations (:120,190,248) are redirected to the CPU for the run --
there is no GPU in the build container -- every other line is the reference's.
"""
import math
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import make_reference_vectors2 as V2  # noqa: E402

N_ANCHORS, K, W, H, TANFOVX = 300, 10, 112, 71, 0.55
SCENARIOS = ("render_train_retain", "render_train_noretain", "render_eval", "prefilter_voxel", "prefilter_position2D")


def standin(device="cpu"):
    """The model / camera / pipe / background every scenario runs on (shared with the GPU test that replays the fixture)."""
    from gscream_amd import standin_model as SM
    from gscream_amd import synthetic as S
    m = SM.Model(N_ANCHORS, K, seed=11, dtype=torch.float32, spread=1.2).to(device)
    w2c = np.eye(4, dtype=np.float32)
    w2c[2, 3] = 6.0
    tanfovy = TANFOVX * H / W
    view, proj, campos = S.camera_matrices(TANFOVX, tanfovy, w2c)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float().to(device)
    cam = SM.Camera(t(campos), image_height=H, image_width=W, FoVx=2 * math.atan(TANFOVX), FoVy=2 * math.atan(tanfovy),
                    world_view_transform=t(view), full_proj_transform=t(proj))
    pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=False)
    bg = torch.tensor([0.1, 0.2, 0.3], device=device)
    g = torch.Generator().manual_seed(5)
    vis = (torch.rand(N_ANCHORS, generator=g) > 0.3).to(device)
    return m, cam, pipe, bg, vis


class Recorder:
    """Recording stand-ins for the two names gaussian_renderer imports from diff_gaussian_rasterization."""

    def __init__(self, named_objects):
        self.named = named_objects   # {name: object}: settings tensors are identified by identity against these
        self.log = {}
        rec = self

        class Settings:
            def __init__(self, *args, **kw):
                assert not args, "the reference builds the settings by keyword"
                rec.log["settings_kwargs"] = list(kw)
                rec.log["settings"] = kw
                self.__dict__.update(kw)

        class Rasterizer(torch.nn.Module):
            def __init__(self, *args, **kw):
                super().__init__()
                rec.log["ctor_args"], rec.log["ctor_kwargs"] = len(args), list(kw)
                self.raster_settings = kw.get("raster_settings", args[0] if args else None)

            def _note(self, method, args, kw):
                rec.log["method"], rec.log["call_positional"], rec.log["call_kwargs"] = method, len(args), list(kw)
                rec.log["call"] = kw

            def forward(self, *args, **kw):
                self._note("forward", args, kw)
                P, rs = kw["means3D"].shape[0], self.raster_settings
                # outputs connected to every differentiable input, so that a backward reaches means2D like the real node's
                link = sum(v.sum() for v in kw.values() if isinstance(v, torch.Tensor) and v.requires_grad) * 0.0
                dev = kw["means3D"].device
                f = lambda c: torch.zeros(c, rs.image_height, rs.image_width, device=dev) + link
                return f(3), f(1), f(1), torch.ones(P, dtype=torch.int32, device=dev)

            def visible_filter(self, *args, **kw):
                self._note("visible_filter", args, kw)
                return torch.ones(kw["means3D"].shape[0], dtype=torch.int32, device=kw["means3D"].device)

            def position2D_filter(self, *args, **kw):
                self._note("position2D_filter", args, kw)
                P, dev = kw["means3D"].shape[0], kw["means3D"].device
                return torch.ones(P, dtype=torch.int32, device=dev), torch.zeros(P, device=dev), torch.zeros(P, device=dev)

        self.Settings, self.Rasterizer = Settings, Rasterizer

    def identify(self, v):
        for k, o in self.named.items():
            if v is o:
                return k
        return ""


def describe(v):
    """dtype | shape | contiguous | requires_grad | is_leaf of a tensor; 'None'; or the Python type and value of a scalar."""
    if v is None:
        return "None"
    if isinstance(v, torch.Tensor):
        return f"tensor|{str(v.dtype).replace('torch.', '')}|{'x'.join(str(d) for d in v.shape)}|contiguous={int(v.is_contiguous())}|" \
               f"requires_grad={int(v.requires_grad)}|is_leaf={int(v.is_leaf)}"
    return f"{type(v).__name__}|{v!r}"


def run(GR, device="cpu"):
    """`GR` = the reference's gaussian_renderer module (or, in the GPU test, this package's mirror driven the same way)."""
    out = {"scenarios": np.array(SCENARIOS)}
    for sc in SCENARIOS:
        m, cam, pipe, bg, vis = standin(device)
        named = {"bg_color": bg, "viewpoint_camera.world_view_transform": cam.world_view_transform,
                 "viewpoint_camera.full_proj_transform": cam.full_proj_transform, "viewpoint_camera.camera_center": cam.camera_center}
        rec = Recorder(named)
        GR.GaussianRasterizationSettings, GR.GaussianRasterizer = rec.Settings, rec.Rasterizer
        m.train(sc.startswith("render_train"))
        if sc.startswith("render"):
            res = GR.render(cam, m, pipe, bg, visible_mask=vis, retain_grad=(sc == "render_train_retain"))
        elif sc == "prefilter_voxel":
            res = GR.prefilter_voxel(cam, m, pipe, bg)
        else:
            res = GR.prefilter_position2D(cam, m, pipe, bg)
        L = rec.log
        out[f"{sc}/ctor"] = np.array([f"positional={L['ctor_args']}"] + [f"kw:{k}" for k in L["ctor_kwargs"]])
        out[f"{sc}/settings_fields"] = np.array(L["settings_kwargs"])
        out[f"{sc}/settings_values"] = np.array([("is:" + rec.identify(v)) if rec.identify(v) else describe(v) for v in L["settings"].values()])
        out[f"{sc}/method"] = np.array([L["method"], f"positional={L['call_positional']}"])
        out[f"{sc}/call_kwargs"] = np.array(L["call_kwargs"])
        out[f"{sc}/call_values"] = np.array([describe(v) for v in L["call"].values()])
        if isinstance(res, dict):
            out[f"{sc}/return_keys"] = np.array(list(res))
            out[f"{sc}/return_values"] = np.array([describe(v) for v in res.values()])
            if sc.startswith("render_train"):
                (res["render"].sum() + res["render_depth"].sum()).backward()
                vp = res["viewspace_points"]
                import warnings
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")   # (.grad of a non-leaf that did not retain it: None + a warning)
                    out[f"{sc}/viewspace_points_grad_after_backward"] = np.array([int(vp.grad is not None)])
        else:
            res = res if isinstance(res, tuple) else (res,)
            out[f"{sc}/return_values"] = np.array([describe(v) for v in res])
    return out


def main():
    _LU, _GU, GR, _GM = V2.import_reference()
    # the three `device="cuda"` allocations of the reference's callers (:120,190,248) -> CPU for this run (no GPU here)
    real = torch.zeros_like
    torch.zeros_like = lambda *a, **k: real(*a, **{kk: v for kk, v in k.items() if kk != "device"})
    try:
        out = run(GR)
    finally:
        torch.zeros_like = real
    np.savez_compressed(os.path.join(HERE, "ref_render_call.npz"), **out)
    for k in sorted(out):
        print(k, list(out[k]))


if __name__ == "__main__":
    main()
