"""Host-side mirror of the reference's rasterizer module for the MI355X build.

Drop-in for `diff_gaussian_rasterization` (reference: submodules/diff-gaussian-rasterization/
diff_gaussian_rasterization/__init__.py, "DGR/__init__.py"): the same three public names with the
same fields, argument order, defaults, return arity, dtypes and error messages, so GScream's
`gaussian_renderer.render()` / `prefilter_*()` and `train.py` run untouched:

    GaussianRasterizationSettings   DGR/__init__.py:189-201
    GaussianRasterizer              DGR/__init__.py:203-312  (.forward, .markVisible,
                                    .visible_filter, .position2D_filter)
    rasterize_gaussians             DGR/__init__.py:21-44

Underneath, instead of the pybind module `_C` (DGR/ext.cpp:15-21) the calls go to libgsraster.so
through the C ABI of include/gsraster.h (gscream_amd/_native.py).  PyTorch only provides device
memory (caching allocator), the current HIP stream and autograd plumbing.
"""
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _native

_tuning = _native.Tuning()
_capacity_hint = {}  # per-device: (binning capacity, longest-list provision) for the next speculative forward
_recent = {}         # per-device: (num_rendered, max_tile_count) of the last few forwards (training hops between views)
_RECENT_FRAMES = 8
_pinned = {}  # per-device pinned int32[4] that receives gsr_stage1_result (truly asynchronous D2H copy)
_last_stage1 = {}  # debugging aid: counts reported by the most recent forward


def set_tuning(tile_cull=True, speculative=True, partial_sort=True):
    """Performance knobs.  Images, radii and gradients do not depend on them.
    partial_sort=False sorts every per-tile list completely (the reference's lists); by default lists longer than 2048
    entries are depth-sorted only as far as the blend is expected to walk, with a complete sort as fall-back.
    tile_cull=False bins every tile of every rectangle: the internal per-tile lists and num_rendered become
    bit-identical to the reference's.  speculative=False always uses the two-stage forward (host reads
    num_rendered, then sizes the binning workspace exactly), like the reference's blocking read-back."""
    _tuning.disable_tile_cull = 0 if tile_cull else 1
    _tuning.disable_speculation = 0 if speculative else 1
    _tuning.disable_partial_sort = 0 if partial_sort else 1
    _capacity_hint.clear()
    _recent.clear()


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


def _require_gpu(t, name):
    if not t.is_cuda:
        raise RuntimeError(
            f"gscream_amd: `{name}` must live on the GPU (got device {t.device}); the rasterizer has no CPU path")


def _f32c(t, device=None):
    """contiguous fp32 view/copy; mirrors the `.contiguous().data<float>()` of DGR rasterize_points.cu:98-118"""
    if t is None:
        return None
    if device is not None and t.device != device:
        t = t.to(device)
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _stream():
    return _native.ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _snapshot(args):
    return tuple(a.detach().cpu().clone() if isinstance(a, torch.Tensor) else a for a in args)


def _cam(rs, device):
    return (_f32c(rs.viewmatrix, device), _f32c(rs.projmatrix, device), _f32c(rs.campos, device))


def _forward_native(means3D, sh, colors_precomp, opacities, uncertainties, scales, rotations, cov3Ds_precomp, rs):
    """The work of `_C.rasterize_gaussians` (DGR rasterize_points.cu:35-122)."""
    lib = _native.load()
    if means3D.dim() != 2 or means3D.shape[1] != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")  # rasterize_points.cu:58-60
    _require_gpu(means3D, "means3D")
    dev = means3D.device
    P, H, W = means3D.shape[0], int(rs.image_height), int(rs.image_width)
    f32 = dict(dtype=torch.float32, device=dev)
    # Outputs are zero-filled like the reference's torch::full (rasterize_points.cu:69-72): with
    # P == 0 the kernels are skipped and the zeros are what the caller gets.
    color = torch.zeros((3, H, W), **f32) if P == 0 else torch.empty((3, H, W), **f32)
    depth = torch.zeros((1, H, W), **f32) if P == 0 else torch.empty((1, H, W), **f32)
    unc = torch.zeros((1, H, W), **f32) if P == 0 else torch.empty((1, H, W), **f32)
    radii = torch.zeros((P,), dtype=torch.int32, device=dev) if P == 0 else torch.empty((P,), dtype=torch.int32, device=dev)
    u8 = dict(dtype=torch.uint8, device=dev)
    if P == 0:
        e = torch.empty((0,), **u8)
        return 0, color, depth, unc, radii, e, e.clone(), e.clone(), 0

    means3D_c, opac_c, unc_c = _f32c(means3D), _f32c(opacities), _f32c(uncertainties)
    scales_c, rot_c, cov_c = _f32c(scales, dev), _f32c(rotations, dev), _f32c(cov3Ds_precomp, dev)
    colors_c, sh_c = _f32c(colors_precomp, dev), _f32c(sh, dev)
    M = sh_c.shape[1] if (sh_c is not None and sh_c.numel() != 0) else 0
    view, proj, campos = _cam(rs, dev)
    bg = _f32c(rs.bg, dev)

    geom = torch.empty((lib.gsr_geom_bytes(P),), **u8)
    img = torch.empty((lib.gsr_image_bytes(P, W, H),), **u8)
    pin = _pinned.get(dev.index)
    if pin is None:
        pin = _pinned[dev.index] = torch.zeros(4, dtype=torch.int32).pin_memory()
    res = _native.ctypes.cast(pin.data_ptr(), _native.ctypes.POINTER(_native.Stage1Result))
    debug = int(bool(rs.debug))
    common = (P, int(rs.sh_degree), M, W, H, _native.ptr(means3D_c), _native.ptr(scales_c), float(rs.scale_modifier),
              _native.ptr(rot_c), _native.ptr(opac_c), _native.ptr(unc_c), _native.ptr(sh_c), _native.ptr(cov_c),
              _native.ptr(colors_c), _native.ptr(view), _native.ptr(proj), _native.ptr(campos), float(rs.tanfovx),
              float(rs.tanfovy), int(bool(rs.prefiltered)))
    outs = (_native.ptr(color), _native.ptr(depth), _native.ptr(unc))
    with torch.cuda.device(dev):
        stream = _stream()
        cap, tile_hint = _capacity_hint.get(dev.index, (0, 0))
        done = False
        if cap > 0 and not _tuning.disable_speculation:
            # Speculative single call: stage 2 is enqueued before the host learns num_rendered, against a workspace
            # sized from the previous frames.  No GPU-idle window; redone below only if the guess was too small.
            binning = torch.empty((lib.gsr_binning_bytes(cap),), **u8)
            rc = lib.gsr_forward(*common, _native.ptr(bg), _native.ptr(geom), _native.ptr(img), _native.ptr(binning), cap,
                                 tile_hint, _native.ptr(radii), *outs, res, _native.ctypes.byref(_tuning), debug, stream)
            if rc not in (0, _native.NEED_CAPACITY):
                _native.check(rc, "gsr_forward")
            done = rc == 0
        else:
            rc = lib.gsr_forward_stage1(*common, _native.ptr(geom), _native.ptr(img), _native.ptr(radii), res,
                                        _native.ctypes.byref(_tuning), debug, stream)
            _native.check(rc, "gsr_forward_stage1")
        res = res.contents
        R = int(res.num_rendered)
        if not done:
            cap = R
            binning = torch.empty((lib.gsr_binning_bytes(cap),), **u8)
            rc = lib.gsr_forward_stage2(P, W, H, R, int(res.max_tile_count), _native.ptr(bg), _native.ptr(geom),
                                        _native.ptr(img), _native.ptr(binning), *outs, _native.ctypes.byref(_tuning),
                                        debug, stream)
            _native.check(rc, "gsr_forward_stage2")
        # Provision for the next call from the largest of the last few frames (a trainer hops between views, so the
        # previous frame alone is a poor predictor): binning capacity, and the longest tile list (sizes the LDS of the
        # per-tile sort; a tight value lets more sort workgroups be resident).  Exceeded -> GSR_NEED_CAPACITY -> stage 2
        # is redone above.
        hist = _recent.setdefault(dev.index, [])
        hist.append((R, int(res.max_tile_count)))
        del hist[:-_RECENT_FRAMES]
        _capacity_hint[dev.index] = (int(1.25 * max(h[0] for h in hist)) + 65536,
                                     max(1024, int(1.25 * max(h[1] for h in hist)) + 64))
    _last_stage1.update(num_rendered=R, max_tile_count=int(res.max_tile_count), num_slots=int(res.num_slots),
                        binning_capacity=cap, speculative=done)
    return R, color, depth, unc, radii, geom, binning, img, cap


def _backward_native(rs, num_rendered, binning_capacity, means3D, radii, colors_precomp, sh, scales, rotations, cov3Ds_precomp,
                     geom, binning, img, g_color, g_depth, g_unc):
    """The work of `_C.rasterize_gaussians_backward` (DGR rasterize_points.cu:124-211)."""
    lib = _native.load()
    dev = means3D.device
    P, H, W = means3D.shape[0], int(rs.image_height), int(rs.image_width)
    f32 = dict(dtype=torch.float32, device=dev)
    have_sh = sh is not None and sh.numel() != 0
    have_cov = cov3Ds_precomp is not None and cov3Ds_precomp.numel() != 0
    M = sh.shape[1] if have_sh else 0
    mk = torch.zeros if P == 0 else torch.empty  # the kernels write every row when P > 0
    g_means2D, g_colors = mk((P, 3), **f32), mk((P, 3), **f32)
    g_opac, g_feat = mk((P, 1), **f32), mk((P, 1), **f32)
    g_means3D = mk((P, 3), **f32)
    # gradients of inputs that were not provided (empty tensors) are None: autograd ignores them, and the
    # reference's zero tensors for them would cost a fill kernel per iteration
    g_cov = mk((P, 6), **f32) if have_cov else None
    g_sh = mk((P, M, 3), **f32) if have_sh else None
    g_scales = None if have_cov else mk((P, 3), **f32)
    g_rot = None if have_cov else mk((P, 4), **f32)
    if P == 0:
        return g_means2D, g_colors, g_opac, g_feat, g_means3D, g_cov, g_sh, g_scales, g_rot
    view, proj, campos = _cam(rs, dev)
    bg = _f32c(rs.bg, dev)
    # set_materialize_grads(False): an output the loss never touched arrives as None.  GScream's loss never uses
    # the uncertainty map (train.py:532) and early iterations use no depth either -> cheaper kernel variant.
    H_, W_ = int(rs.image_height), int(rs.image_width)
    gc = _f32c(g_color, dev) if g_color is not None else torch.zeros((3, H_, W_), **f32)
    if g_depth is None and g_unc is None:
        gd = gu = None
    else:
        gd = _f32c(g_depth, dev) if g_depth is not None else torch.zeros((1, H_, W_), **f32)
        gu = _f32c(g_unc, dev) if g_unc is not None else torch.zeros((1, H_, W_), **f32)
    # keep every converted tensor referenced until the launches are enqueued: a temporary freed early
    # could hand its block to the next temporary
    means3D_c, colors_c, sh_c = _f32c(means3D), _f32c(colors_precomp, dev), _f32c(sh, dev)
    scales_c, rot_c, cov_c = _f32c(scales, dev), _f32c(rotations, dev), _f32c(cov3Ds_precomp, dev)
    scratch = torch.empty((lib.gsr_backward_scratch_bytes(P, num_rendered),), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = lib.gsr_backward(
            P, int(rs.sh_degree), M, W, H, int(num_rendered), int(binning_capacity), _native.ptr(bg), _native.ptr(means3D_c),
            _native.ptr(radii), _native.ptr(colors_c), _native.ptr(sh_c),
            _native.ptr(scales_c), float(rs.scale_modifier), _native.ptr(rot_c),
            _native.ptr(cov_c), _native.ptr(view), _native.ptr(proj), _native.ptr(campos),
            float(rs.tanfovx), float(rs.tanfovy), _native.ptr(gc), _native.ptr(gd), _native.ptr(gu),
            _native.ptr(geom), _native.ptr(img), _native.ptr(binning), _native.ptr(scratch),
            _native.ptr(g_means2D), _native.ptr(g_colors), _native.ptr(g_opac), _native.ptr(g_feat),
            _native.ptr(g_means3D), _native.ptr(g_cov), _native.ptr(g_sh), _native.ptr(g_scales), _native.ptr(g_rot),
            _native.ctypes.byref(_tuning), int(bool(rs.debug)), _stream())
        _native.check(rc, "gsr_backward")
    return g_means2D, g_colors, g_opac, g_feat, g_means3D, g_cov, g_sh, g_scales, g_rot


class _RasterizeGaussians(torch.autograd.Function):
    """Autograd node; saved state and gradient order follow DGR/__init__.py:46-187."""

    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, uncertainties, scales, rotations,
                cov3Ds_precomp, raster_settings):
        args = (means3D, sh, colors_precomp, opacities, uncertainties, scales, rotations, cov3Ds_precomp,
                raster_settings)
        if raster_settings.debug:
            saved = _snapshot(args)  # copy before anything can be corrupted (DGR/__init__.py:87-95)
            try:
                out = _forward_native(*args)
            except Exception:
                torch.save(saved, "snapshot_fw.dump")
                print("\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.")
                raise
        else:
            out = _forward_native(*args)
        num_rendered, color, depth, uncertainty, radii, geom, binning, img, capacity = out
        ctx.raster_settings = raster_settings
        ctx.num_rendered = num_rendered
        ctx.binning_capacity = capacity
        ctx.opacity_shape, ctx.uncertainty_shape = opacities.shape, uncertainties.shape
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geom, binning, img)
        ctx.mark_non_differentiable(radii)
        return color, depth, uncertainty, radii

    @staticmethod
    def backward(ctx, grad_out_color, grad_out_depth, grad_out_uncertainty, _grad_radii):
        rs = ctx.raster_settings
        if grad_out_color is None and grad_out_depth is None and grad_out_uncertainty is None:
            return (None,) * 10
        colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geom, binning, img = ctx.saved_tensors
        args = (rs, ctx.num_rendered, ctx.binning_capacity, means3D, radii, colors_precomp, sh, scales, rotations, cov3Ds_precomp,
                geom, binning, img, grad_out_color, grad_out_depth, grad_out_uncertainty)
        if rs.debug:
            saved = _snapshot(args)
            try:
                grads = _backward_native(*args)
            except Exception:
                torch.save(saved, "snapshot_bw.dump")
                print("\nAn error occured in backward. Writing snapshot_bw.dump for debugging.\n")
                raise
        else:
            grads = _backward_native(*args)
        (g_means2D, g_colors, g_opac, g_unc, g_means3D, g_cov, g_sh, g_scales, g_rot) = grads
        # input order of forward(): means3D, means2D, sh, colors_precomp, opacities, uncertainties,
        # scales, rotations, cov3Ds_precomp, raster_settings        (DGR/__init__.py:174-185)
        return (g_means3D, g_means2D, g_sh, g_colors, g_opac.reshape(ctx.opacity_shape),
                g_unc.reshape(ctx.uncertainty_shape), g_scales, g_rot, g_cov, None)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, uncertainties, scales, rotations,
                        cov3Ds_precomp, raster_settings):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, uncertainties, scales,
                                     rotations, cov3Ds_precomp, raster_settings)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        """bool[P]: view-space z > 0.2 (DGR/__init__.py:208-217, rasterizer_impl.cu:54-66)."""
        lib = _native.load()
        rs = self.raster_settings
        with torch.no_grad():
            _require_gpu(positions, "positions")
            dev = positions.device
            P = positions.shape[0]
            present = torch.zeros((P,), dtype=torch.bool, device=dev)
            if P:
                view, proj, _ = _cam(rs, dev)
                pos_c = _f32c(positions)
                with torch.cuda.device(dev):
                    rc = lib.gsr_mark_visible(P, _native.ptr(pos_c), _native.ptr(view), _native.ptr(proj),
                                              _native.ptr(present), _stream())
                _native.check(rc, "gsr_mark_visible")
        return present

    def forward(self, means3D, means2D, opacities, uncertainties, shs=None, colors_precomp=None, scales=None,
                rotations=None, cov3D_precomp=None):
        raster_settings = self.raster_settings
        if (shs is None) == (colors_precomp is None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        pair_missing = scales is None or rotations is None
        pair_any = scales is not None or rotations is not None
        if (pair_missing and cov3D_precomp is None) or (pair_any and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        empty = torch.Tensor([])  # absent inputs travel as empty tensors (DGR/__init__.py:230-240)
        return rasterize_gaussians(
            means3D, means2D,
            empty if shs is None else shs,
            empty if colors_precomp is None else colors_precomp,
            opacities, uncertainties,
            empty if scales is None else scales,
            empty if rotations is None else rotations,
            empty if cov3D_precomp is None else cov3D_precomp,
            raster_settings)

    def _filter(self, means3D, scales, rotations, cov3D_precomp, want_xy):
        lib = _native.load()
        rs = self.raster_settings
        with torch.no_grad():
            if means3D.dim() != 2 or means3D.shape[1] != 3:
                raise RuntimeError("means3D must have dimensions (num_points, 3)")  # rasterize_points.cu:251-253
            _require_gpu(means3D, "means3D")
            dev = means3D.device
            P = means3D.shape[0]
            radii = torch.zeros((P,), dtype=torch.int32, device=dev)
            px = torch.zeros((P,), dtype=torch.float32, device=dev) if want_xy else None
            py = torch.zeros((P,), dtype=torch.float32, device=dev) if want_xy else None
            if P:
                view, proj, _ = _cam(rs, dev)
                means_c, scales_c = _f32c(means3D), _f32c(scales, dev)  # e.g. get_scaling[:, :3] is a strided slice
                rot_c, cov_c = _f32c(rotations, dev), _f32c(cov3D_precomp, dev)
                with torch.cuda.device(dev):
                    rc = lib.gsr_filter(
                        P, int(rs.image_width), int(rs.image_height), _native.ptr(means_c),
                        _native.ptr(scales_c), float(rs.scale_modifier), _native.ptr(rot_c),
                        _native.ptr(cov_c), _native.ptr(view), _native.ptr(proj),
                        float(rs.tanfovx), float(rs.tanfovy), int(bool(rs.prefiltered)), _native.ptr(radii),
                        _native.ptr(px), _native.ptr(py), int(bool(rs.debug)), _stream())
                _native.check(rc, "gsr_filter")
        return radii, px, py

    def visible_filter(self, means3D, scales=None, rotations=None, cov3D_precomp=None):
        """radii[P] of a point cloud, no rendering (DGR/__init__.py:256-282)."""
        return self._filter(means3D, scales, rotations, cov3D_precomp, False)[0]

    def position2D_filter(self, means3D, scales=None, rotations=None, cov3D_precomp=None):
        """(radii, x, y): radii plus pixel-space centres, 0 where culled (DGR/__init__.py:285-312)."""
        return self._filter(means3D, scales, rotations, cov3D_precomp, True)
