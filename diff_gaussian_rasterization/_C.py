"""`diff_gaussian_rasterization._C` for a maintainer who keeps the REFERENCE's Python wrapper
(submodules/diff-gaussian-rasterization/diff_gaussian_rasterization/__init__.py) and swaps only the native module:
the five entry points the wrapper calls (DGR/ext.cpp:15-21), with the reference's positional argument order and return
tuples (DGR/rasterize_points.h:18-103), bound to libgsraster.so through the C ABI (include/gsraster.h).  ctypes only,
no compilation step.  The argument order is pinned by tests/golden/ref_wrapper.npz, recorded from the reference wrapper
itself (tests/test_reference_vectors2.py).

Two-stage forward (capacity == num_rendered), because the reference's backward receives only `R`."""
import ctypes

import torch

from gscream_amd import _native

_TUNING = _native.Tuning()


def _s():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f(t, dev=None):
    """contiguous fp32 on the compute device (rasterize_points.cu:98-118 `.contiguous().data<float>()`); empty stays empty"""
    if dev is not None and t.device != dev:
        t = t.to(dev)
    return t.float().contiguous()


def rasterize_gaussians(background, means3D, colors, opacity, uncertainty, scales, rotations, scale_modifier, cov3D_precomp,
                        viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos,
                        prefiltered, debug):
    """RasterizeGaussiansCUDA (rasterize_points.cu:35-122) -> (num_rendered, color, depth, uncertainty, radii, geomBuffer,
    binningBuffer, imgBuffer)."""
    L, P_ = _native.load(), _native.ptr
    if means3D.dim() != 2 or means3D.shape[1] != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    if not means3D.is_cuda:
        raise RuntimeError("diff_gaussian_rasterization._C: tensors must be on a HIP device (no CPU path)")
    Pn, H, W, dev = means3D.shape[0], int(image_height), int(image_width), means3D.device
    f = dict(dtype=torch.float32, device=dev)
    color, depth, unc = torch.zeros(3, H, W, **f), torch.zeros(1, H, W, **f), torch.zeros(1, H, W, **f)
    radii = torch.zeros(Pn, dtype=torch.int32, device=dev)
    e = torch.empty(0, dtype=torch.uint8, device=dev)
    if Pn == 0:
        return 0, color, depth, unc, radii, e, e.clone(), e.clone()
    m, c, o, u, s, r, cov, shc = (_f(t, dev) for t in (means3D, colors, opacity, uncertainty, scales, rotations, cov3D_precomp, sh))
    view, proj, cam, bg = _f(viewmatrix, dev), _f(projmatrix, dev), _f(campos, dev), _f(background, dev)
    M = shc.shape[1] if shc.numel() else 0
    with torch.cuda.device(dev):
        geom = torch.empty(L.gsr_geom_bytes(Pn), dtype=torch.uint8, device=dev)
        img = torch.empty(L.gsr_image_bytes(Pn, W, H), dtype=torch.uint8, device=dev)
        res = _native.Stage1Result()
        _native.check(L.gsr_forward_stage1(Pn, int(degree), M, W, H, P_(m), P_(s), float(scale_modifier), P_(r), P_(o), P_(u), P_(shc),
                                           P_(cov), P_(c), P_(view), P_(proj), P_(cam), float(tan_fovx), float(tan_fovy),
                                           int(bool(prefiltered)), P_(geom), P_(img), P_(radii), ctypes.byref(res),
                                           ctypes.byref(_TUNING), int(bool(debug)), _s()), "gsr_forward_stage1")
        binning = torch.empty(L.gsr_binning_bytes(res.num_rendered), dtype=torch.uint8, device=dev)
        _native.check(L.gsr_forward_stage2(Pn, W, H, res.num_rendered, res.max_tile_count, P_(bg), P_(geom), P_(img), P_(binning),
                                           P_(color), P_(depth), P_(unc), ctypes.byref(_TUNING), int(bool(debug)), _s()),
                      "gsr_forward_stage2")
    return int(res.num_rendered), color, depth, unc, radii, geom, binning, img


def rasterize_gaussians_backward(background, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
                                 projmatrix, tan_fovx, tan_fovy, dL_dout_color, dL_dout_depth, dL_dout_uncertainty, sh, degree,
                                 campos, geomBuffer, R, binningBuffer, imageBuffer, debug):
    """RasterizeGaussiansBackwardCUDA (rasterize_points.cu:124-211) -> (dL_dmeans2D, dL_dcolors, dL_dopacity,
    dL_duncertainty, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations), all zero-filled like the reference's."""
    L, P_ = _native.load(), _native.ptr
    Pn, dev = means3D.shape[0], means3D.device
    H, W = int(dL_dout_color.shape[1]), int(dL_dout_color.shape[2])          # rasterize_points.cu:150-151
    shc = _f(sh, dev)
    M = shc.shape[1] if shc.numel() else 0
    z = lambda *shape: torch.zeros(*shape, dtype=torch.float32, device=dev)
    g_m2, g_col, g_op, g_unc, g_m3 = z(Pn, 3), z(Pn, 3), z(Pn, 1), z(Pn, 1), z(Pn, 3)
    g_cov, g_sh, g_sc, g_rot = z(Pn, 6), z(Pn, M, 3), z(Pn, 3), z(Pn, 4)
    if Pn == 0:
        return g_m2, g_col, g_op, g_unc, g_m3, g_cov, g_sh, g_sc, g_rot
    m, c, s, r, cov = (_f(t, dev) for t in (means3D, colors, scales, rotations, cov3D_precomp))
    view, proj, cam, bg = _f(viewmatrix, dev), _f(projmatrix, dev), _f(campos, dev), _f(background, dev)
    gc, gd, gu = _f(dL_dout_color, dev), _f(dL_dout_depth, dev), _f(dL_dout_uncertainty, dev)
    have_cov = cov.numel() != 0
    with torch.cuda.device(dev):
        scratch = torch.empty(L.gsr_backward_scratch_bytes(Pn, int(R)), dtype=torch.uint8, device=dev)
        _native.check(L.gsr_backward(Pn, int(degree), M, W, H, int(R), int(R), -1, P_(bg),  # (-1: the longest list is not kept by this entry -> full task grid)
                                     P_(m), P_(radii), P_(c), P_(shc), P_(s),
                                     float(scale_modifier), P_(r), P_(cov), P_(view), P_(proj), P_(cam), float(tan_fovx),
                                     float(tan_fovy), P_(gc), P_(gd), P_(gu), P_(geomBuffer), P_(imageBuffer), P_(binningBuffer),
                                     P_(scratch), P_(g_m2), P_(g_col), P_(g_op), P_(g_unc), P_(g_m3),
                                     P_(g_cov) if have_cov else None, P_(g_sh) if M else None,
                                     None if have_cov else P_(g_sc), None if have_cov else P_(g_rot),
                                     ctypes.byref(_TUNING), int(bool(debug)), _s()), "gsr_backward")
    return g_m2, g_col, g_op, g_unc, g_m3, g_cov, g_sh, g_sc, g_rot


def _filter(want_xy, means3D, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy,
            image_height, image_width, prefiltered, debug):
    L, P_ = _native.load(), _native.ptr
    if means3D.dim() != 2 or means3D.shape[1] != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    Pn, dev = means3D.shape[0], means3D.device
    radii = torch.zeros(Pn, dtype=torch.int32, device=dev)
    px = torch.zeros(Pn, dtype=torch.float32, device=dev) if want_xy else None
    py = torch.zeros(Pn, dtype=torch.float32, device=dev) if want_xy else None
    if Pn:
        m, s, r, cov = (_f(t, dev) for t in (means3D, scales, rotations, cov3D_precomp))
        view, proj = _f(viewmatrix, dev), _f(projmatrix, dev)
        with torch.cuda.device(dev):
            _native.check(L.gsr_filter(Pn, int(image_width), int(image_height), P_(m), P_(s), float(scale_modifier), P_(r), P_(cov),
                                       P_(view), P_(proj), float(tan_fovx), float(tan_fovy), int(bool(prefiltered)), P_(radii),
                                       P_(px), P_(py), int(bool(debug)), _s()), "gsr_filter")
    return radii, px, py


def rasterize_aussians_filter(means3D, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy,
                              image_height, image_width, prefiltered, debug):
    """RasterizeGaussiansfilterCUDA (rasterize_points.cu:235-296; the reference's spelling) -> radii."""
    return _filter(False, means3D, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy,
                   image_height, image_width, prefiltered, debug)[0]


def rasterize_aussians_filter_position2D(means3D, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix, projmatrix, tan_fovx,
                                         tan_fovy, image_height, image_width, prefiltered, debug):
    """RasterizeGaussiansfilterposition2DCUDA (rasterize_points.cu:299-373) -> (radii, position2D_x, position2D_y)."""
    return _filter(True, means3D, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy,
                   image_height, image_width, prefiltered, debug)


def mark_visible(means3D, viewmatrix, projmatrix):
    """markVisible (rasterize_points.cu:213-232) -> bool[P]."""
    L, P_ = _native.load(), _native.ptr
    Pn, dev = means3D.shape[0], means3D.device
    present = torch.zeros(Pn, dtype=torch.bool, device=dev)
    if Pn:
        m, view, proj = _f(means3D, dev), _f(viewmatrix, dev), _f(projmatrix, dev)
        with torch.cuda.device(dev):
            _native.check(L.gsr_mark_visible(Pn, P_(m), P_(view), P_(proj), P_(present), _s()), "gsr_mark_visible")
    return present
