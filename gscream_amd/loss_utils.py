"""Image-space RGB losses on the HIP path -- SURVEY 8(f) rank 2, the step right after the rasterizer.

Mirrors the names and call signatures GScream's trainer imports from ``utils/loss_utils.py``:

    l1_loss(network_output, gt)                    loss_utils.py:26-27
    l1_loss_masked(network_output, gt, mask)       loss_utils.py:29-30
    ssim(img1, img2, window_size=11, size_average=True)                 loss_utils.py:131-160
    ssim_masked(img1, img2, mask, window_size=11, size_average=True)    loss_utils.py:165-190

and adds ``rgb_loss`` = the composition train.py:538-545 builds from them, in ONE forward and ONE backward kernel:

    rgb_loss(image, gt, weight=None, lambda_dssim=0.2, scale=1.0)
        = scale * ((1 - lambda) * mean(|image - gt| * weight) + lambda * (1 - mean(ssim_map * weight)))

All of them run `gsr_rgb_loss_forward/backward` (include/gsraster.h) through ctypes; gradients flow to the first
image argument only (the ground truth is a constant in the trainer).  There is no CPU fallback: tensors must live on
a HIP device.  `window_size`: odd, up to 11 (11 everywhere in GScream).  Differences from the reference: the window is applied separably (two 11-tap passes) instead of
as one 121-tap depthwise conv2d, so values agree to fp32 rounding (~1e-6), not bit for bit.
"""
import ctypes

import torch

from . import _native

__all__ = ["l1_loss", "l1_loss_masked", "ssim", "ssim_masked", "rgb_loss", "depth_loss"]


def _prep(img, gt, weight):
    if not img.is_cuda:
        raise RuntimeError("gscream_amd.loss_utils: tensors must be on a HIP device (there is no CPU fallback)")
    if img.shape != gt.shape:
        raise ValueError(f"image {tuple(img.shape)} and ground truth {tuple(gt.shape)} differ in shape")
    if img.dim() == 4 and img.shape[0] == 1:
        img, gt = img[0], gt[0]
    if img.dim() != 3:
        raise ValueError("expected a [C,H,W] (or [1,C,H,W]) image")
    C, H, W = img.shape
    x = img.contiguous().float()
    y = gt.detach().contiguous().float()
    w = None
    if weight is not None:
        w = weight.detach().float()
        if w.numel() != H * W:
            raise ValueError(f"weight must broadcast as [1,H,W]; got {tuple(weight.shape)} for a {H}x{W} image")
        w = w.reshape(H, W).contiguous()
    return x, y, w, C, H, W


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class _FusedLoss(torch.autograd.Function):
    """L = a_l1 * mean(|img - gt| * m) + a_ssim * mean(ssim_map(img, gt) * m)."""

    @staticmethod
    def forward(ctx, img, gt, weight, a_l1, a_ssim, window_size=11):
        lib = _native.load()
        x, y, w, C, H, W = _prep(img, gt, weight)
        need_grad = img.requires_grad
        with torch.cuda.device(x.device):
            ws = torch.empty((lib.gsr_loss_workspace_bytes(C, H, W),), dtype=torch.uint8, device=x.device)
            out = torch.empty((3,), dtype=torch.float32, device=x.device)
            _native.check(lib.gsr_rgb_loss_forward_window(C, H, W, _native.ptr(x), _native.ptr(y), _native.ptr(w), float(a_l1),
                                                          float(a_ssim), int(window_size), _native.ptr(ws), _native.ptr(out),
                                                          int(need_grad), _stream()), "gsr_rgb_loss_forward")
        ctx.save_for_backward(x, y, w if w is not None else torch.empty(0, device=x.device), ws)
        ctx.coef = (float(a_l1), float(a_ssim))
        ctx.window_size = int(window_size)
        ctx.in_shape = tuple(img.shape)
        ctx.mark_non_differentiable(out)
        return out[0], out

    @staticmethod
    def backward(ctx, g_loss, _g_parts):
        lib = _native.load()
        x, y, w, ws = ctx.saved_tensors
        C, H, W = x.shape
        up = g_loss.detach().reshape(1).float().contiguous()
        with torch.cuda.device(x.device):
            grad = torch.empty_like(x)
            _native.check(lib.gsr_rgb_loss_backward_window(C, H, W, _native.ptr(x), _native.ptr(y), _native.ptr(w), ctx.coef[0],
                                                           ctx.coef[1], ctx.window_size, _native.ptr(ws), _native.ptr(up),
                                                           _native.ptr(grad), _stream()), "gsr_rgb_loss_backward")
        return grad.reshape(ctx.in_shape), None, None, None, None, None


def _check_window(window_size, size_average, img):
    if not (1 <= int(window_size) <= 11) or int(window_size) % 2 == 0:
        # even windows change the map's size in the reference (padding = window_size // 2) and fail against the mask; larger
        # ones are beyond the kernels' 11-tap frame.  GScream uses 11 everywhere.
        raise NotImplementedError("window_size must be odd and at most 11")
    if not size_average and not (img.dim() == 4 and img.shape[0] == 1):
        # loss_utils.py:160 `ssim_map.mean(1).mean(1).mean(1)`: a per-image mean, defined for [B,C,H,W] input only
        # (the reference itself fails on the [C,H,W] images the trainer passes); one image per call here
        raise NotImplementedError("size_average=False needs a [1,C,H,W] input (one image per call)")


def _per_image(v, size_average):
    return v if size_average else v.reshape(1)  # loss_utils.py:157-160: mean over the whole map, or one mean per image


def l1_loss(network_output, gt):
    return _FusedLoss.apply(network_output, gt, None, 1.0, 0.0)[0]


def l1_loss_masked(network_output, gt, mask):
    return _FusedLoss.apply(network_output, gt, mask, 1.0, 0.0)[0]


def ssim(img1, img2, window_size=11, size_average=True):
    _check_window(window_size, size_average, img1)
    return _per_image(_FusedLoss.apply(img1, img2, None, 0.0, 1.0, window_size)[0], size_average)


def ssim_masked(img1, img2, mask, window_size=11, size_average=True):
    _check_window(window_size, size_average, img1)
    return _per_image(_FusedLoss.apply(img1, img2, mask, 0.0, 1.0, window_size)[0], size_average)


def rgb_loss(image, gt, weight=None, lambda_dssim=0.2, scale=1.0, return_parts=False):
    """train.py:538-545 in one pass.  return_parts -> (loss, L1 term, SSIM term) (the last two detached)."""
    loss, parts = _FusedLoss.apply(image, gt, weight, scale * (1.0 - lambda_dssim), -scale * lambda_dssim)
    loss = loss + scale * lambda_dssim
    return (loss, parts[1], parts[2]) if return_parts else loss


class _DepthLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, depth, target, lsq_mask, l1_weight, grad_mask, lambda_l1, lambda_smooth):
        lib = _native.load()
        if not depth.is_cuda:
            raise RuntimeError("gscream_amd.loss_utils: tensors must be on a HIP device (there is no CPU fallback)")
        H, W = depth.shape[-2:]
        if depth.numel() != H * W or target.numel() != H * W:
            raise ValueError("depth and target must be [1,H,W] / [H,W] maps of the same size")
        f = lambda t: None if t is None else t.detach().reshape(H, W).contiguous().float()
        d, y, m, w, g = f(depth), f(target), f(lsq_mask), f(l1_weight), f(grad_mask)
        with torch.cuda.device(d.device):
            ws = torch.empty((lib.gsr_depth_loss_workspace_bytes(H, W),), dtype=torch.uint8, device=d.device)
            out = torch.empty((5,), dtype=torch.float32, device=d.device)
            _native.check(lib.gsr_depth_loss_forward(H, W, _native.ptr(d), _native.ptr(y), _native.ptr(m), _native.ptr(w),
                                                     _native.ptr(g), float(lambda_l1), float(lambda_smooth), _native.ptr(ws),
                                                     _native.ptr(out), _stream()), "gsr_depth_loss_forward")
        ctx.save_for_backward(d, y, m if m is not None else torch.empty(0, device=d.device), ws)
        ctx.in_shape = tuple(depth.shape)
        ctx.mark_non_differentiable(out)
        return out[0], out

    @staticmethod
    def backward(ctx, g_loss, _g_parts):
        lib = _native.load()
        d, y, m, ws = ctx.saved_tensors
        H, W = d.shape
        up = g_loss.detach().reshape(1).float().contiguous()
        with torch.cuda.device(d.device):
            grad = torch.empty_like(d)
            _native.check(lib.gsr_depth_loss_backward(H, W, _native.ptr(d), _native.ptr(y), _native.ptr(m), _native.ptr(ws),
                                                      _native.ptr(up), _native.ptr(grad), _stream()), "gsr_depth_loss_backward")
        return grad.reshape(ctx.in_shape), None, None, None, None, None, None


def depth_loss(depth, target, lsq_mask=None, l1_weight=None, grad_mask=None, lambda_l1=1.0, lambda_smooth=1.0,
               return_parts=False, fg_mask=None, lambda_fg=0.0):
    """The depth terms of train.py:548-573 in one forward and one backward:
        scale, shift = compute_scale_and_shift(depth, target, lsq_mask); aligned = |scale| * depth + shift
        lambda_l1 * l1_loss[_masked](aligned, target[, l1_weight])
          + lambda_fg * l1_loss_masked(aligned, target, fg_mask)                                  (train.py:555-557)
          + sum_{k<4} 0.5 * lambda_smooth * gradient_loss(aligned[:, ::2^k, ::2^k], target[...], grad_mask[...])
    Reference view (train.py:548-561): lsq_mask = 1 - gt_mask, l1_weight = grad_mask = None, lambda_l1 =
    opt.refer_depth_lr, lambda_smooth = opt.refer_depth_lr_smooth and -- in the shipped run config (scripts/run.py:
    refer_depth_lr_fg = 100 > refer_depth_lr = 1) -- the foreground term `fg_mask = get_random_mask(...)`, `lambda_fg =
    opt.refer_depth_lr_fg - opt.refer_depth_lr`, the DOMINANT depth term there: do not drop it.  Other views (:563-573):
    l1_weight = grad_mask = lsq_mask = valid_mask, lambda_l1 = opt.other_depth_lr, lambda_smooth =
    opt.other_depth_lr_smooth.  Both L1 terms are means over all pixels of |aligned - target| times a weight
    (utils/loss_utils.py:26-30), so they are folded into ONE per-pixel weight map `lambda_l1 * l1_weight + lambda_fg *
    fg_mask` for the kernel.  Gradients flow to `depth`, through the alignment as well.
    return_parts -> (loss, (loss, weighted L1 mean, smooth part, scale, shift))."""
    if fg_mask is not None and float(lambda_fg) != 0.0:
        base = float(lambda_l1) if l1_weight is None else float(lambda_l1) * l1_weight.detach().float()
        l1_weight = base + float(lambda_fg) * fg_mask.detach().float().reshape(
            fg_mask.shape[-2:] if l1_weight is None else l1_weight.shape)
        lambda_l1 = 1.0
    loss, parts = _DepthLoss.apply(depth, target, lsq_mask, l1_weight, grad_mask, lambda_l1, lambda_smooth)
    return (loss, parts) if return_parts else loss
