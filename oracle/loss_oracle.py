"""CPU oracle for the image-space RGB loss.  TEST INFRASTRUCTURE ONLY: imported by tests/,
__graft_entry__.smoke() and bench.py's cpu leg, never by the product path.

Restates GScream's utils/loss_utils.py in plain torch (float64 by default), function by function:
    gaussian / create_window   :113-121   (sigma 1.5, fp32 1-D window, outer product, one [C,1,11,11] depthwise kernel)
    _ssim                      :140-160   (five F.conv2d with padding 5, C1 = 0.01^2, C2 = 0.03^2, mean)
    _ssim_masked               :174-190   (same map, times the mask, mean)
    l1_loss / l1_loss_masked   :26-30
and the composition of train.py:538-545.

PINNED against the reference's own code: tests/golden/make_reference_vectors2.py imports the reference's
utils/loss_utils.py in the build container (its `kornia` import, which the exercised functions never touch, is
satisfied by a stub that raises on any use) and stores the values and autograd gradients of the functions below and of
their train.py compositions in tests/golden/ref_loss.npz; tests/test_reference_vectors2.py holds this oracle to those
to 1e-12.  The restatement follows the cited lines one to one (the 2-D window is applied as ONE conv2d, as the
reference does -- the HIP kernel applies it separably).
"""
from math import exp

import torch
import torch.nn.functional as F


def gaussian(window_size=11, sigma=1.5):  # loss_utils.py:113-115 (fp32 like torch.Tensor)
    g = torch.Tensor([exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(window_size)])
    return g / g.sum()


def create_window(window_size, channel, dtype):  # loss_utils.py:117-121
    w1 = gaussian(window_size, 1.5).unsqueeze(1)
    w2 = w1.mm(w1.t()).float().unsqueeze(0).unsqueeze(0)
    return w2.expand(channel, 1, window_size, window_size).contiguous().to(dtype)


def ssim_map(img1, img2, window_size=11):  # loss_utils.py:140-157
    C = img1.size(-3)
    win = create_window(window_size, C, img1.dtype).to(img1.device)  # loss_utils.py:135-137
    pad = window_size // 2
    x, y = img1.reshape(1, C, *img1.shape[-2:]), img2.reshape(1, C, *img2.shape[-2:])
    mu1, mu2 = F.conv2d(x, win, padding=pad, groups=C), F.conv2d(y, win, padding=pad, groups=C)
    mu1_sq, mu2_sq, mu1_mu2 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    sigma1_sq = F.conv2d(x * x, win, padding=pad, groups=C) - mu1_sq
    sigma2_sq = F.conv2d(y * y, win, padding=pad, groups=C) - mu2_sq
    sigma12 = F.conv2d(x * y, win, padding=pad, groups=C) - mu1_mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    return (((2 * mu1_mu2 + C1) * (2 * sigma12 + C2)) / ((mu1_sq + mu2_sq + C1) * (sigma1_sq + sigma2_sq + C2)))[0]


def ssim(img1, img2):  # :159
    return ssim_map(img1, img2).mean()


def ssim_masked(img1, img2, mask):  # :186-189
    return (ssim_map(img1, img2) * mask).mean()


def l1_loss(a, b):  # :26-27
    return torch.abs(a - b).mean()


def l1_loss_masked(a, b, mask):  # :29-30
    return (torch.abs(a - b) * mask).mean()


def rgb_loss(image, gt, weight=None, lambda_dssim=0.2, scale=1.0):  # train.py:538-545
    if weight is None:
        return scale * ((1.0 - lambda_dssim) * l1_loss(image, gt) + lambda_dssim * (1.0 - ssim(image, gt)))
    return scale * ((1.0 - lambda_dssim) * l1_loss_masked(image, gt, weight)
                    + lambda_dssim * (1.0 - ssim_masked(image, gt, weight)))


def value_and_grad(image, gt, weight=None, lambda_dssim=0.2, scale=1.0, dtype=torch.float64):
    """-> (loss, l1 term, ssim term, dL/dimage) as numpy arrays, computed on the CPU in `dtype`."""
    x = torch.as_tensor(image).to(dtype).clone().requires_grad_(True)
    y = torch.as_tensor(gt).to(dtype)
    w = None if weight is None else torch.as_tensor(weight).to(dtype).reshape(1, *x.shape[-2:])
    loss = rgb_loss(x, y, w, lambda_dssim, scale)
    loss.backward()
    with torch.no_grad():
        l1 = l1_loss(x, y) if w is None else l1_loss_masked(x, y, w)
        ss = ssim(x, y) if w is None else ssim_masked(x, y, w)
    return float(loss.detach()), float(l1), float(ss), x.grad.numpy()


# ---- depth terms (train.py:548-573) -------------------------------------------------------------------------------
def compute_scale_and_shift(prediction, target, mask):  # loss_utils.py:77-104
    a_00 = torch.sum(mask * prediction * prediction, (1, 2))
    a_01 = torch.sum(mask * prediction, (1, 2))
    a_11 = torch.sum(mask, (1, 2))
    b_0 = torch.sum(mask * prediction * target, (1, 2))
    b_1 = torch.sum(mask * target, (1, 2))
    x_0, x_1 = torch.zeros_like(b_0), torch.zeros_like(b_1)
    det = a_00 * a_11 - a_01 * a_01
    valid = det.nonzero()
    x_0[valid] = (a_11[valid] * b_0[valid] - a_01[valid] * b_1[valid]) / det[valid]
    x_1[valid] = (-a_01[valid] * b_0[valid] + a_00[valid] * b_1[valid]) / det[valid]
    return x_0, x_1


def reduction_batch_based(image_loss, M):  # :40-49
    divisor = torch.sum(M)
    return 0 if divisor == 0 else torch.sum(image_loss) / divisor


def gradient_loss(prediction, target, mask):  # :58-74
    M = torch.sum(mask, (1, 2))
    diff = torch.mul(mask, prediction - target)
    grad_x = torch.abs(diff[:, :, 1:] - diff[:, :, :-1]) * torch.mul(mask[:, :, 1:], mask[:, :, :-1])
    grad_y = torch.abs(diff[:, 1:, :] - diff[:, :-1, :]) * torch.mul(mask[:, 1:, :], mask[:, :-1, :])
    return reduction_batch_based(torch.sum(grad_x, (1, 2)) + torch.sum(grad_y, (1, 2)), M)


def depth_loss(depth, target, lsq_mask=None, l1_weight=None, grad_mask=None, lambda_l1=1.0, lambda_smooth=1.0,
               fg_mask=None, lambda_fg=0.0):
    """train.py:548-561 (reference view: weights None; fg_mask / lambda_fg = the foreground term :555-557) / :563-573
    (other views: l1_weight = grad_mask = valid_mask)."""
    ones = torch.ones_like(depth)
    m = ones if lsq_mask is None else lsq_mask
    scale, shift = compute_scale_and_shift(depth, target, m)          # train.py:551
    scale = torch.abs(scale)                                          # :552
    aligned = scale.view(-1, 1, 1) * depth + shift.view(-1, 1, 1)     # :553
    loss = lambda_l1 * (l1_loss(aligned, target) if l1_weight is None else l1_loss_masked(aligned, target, l1_weight))
    if fg_mask is not None:
        loss = loss + lambda_fg * l1_loss_masked(aligned, target, fg_mask)   # train.py:555-557
    g = ones if grad_mask is None else grad_mask
    for k in range(4):                                                # :558-561
        step = pow(2, k)
        loss = loss + 0.5 * lambda_smooth * gradient_loss(aligned[:, ::step, ::step], target[:, ::step, ::step], g[:, ::step, ::step])
    return loss, scale, shift


def depth_value_and_grad(depth, target, lsq_mask=None, l1_weight=None, grad_mask=None, lambda_l1=1.0, lambda_smooth=1.0,
                         dtype=torch.float64, fg_mask=None, lambda_fg=0.0):
    t = lambda a: None if a is None else torch.as_tensor(a).to(dtype).reshape(1, *torch.as_tensor(a).shape[-2:])
    d = t(depth).clone().requires_grad_(True)
    loss, scale, shift = depth_loss(d, t(target), t(lsq_mask), t(l1_weight), t(grad_mask), lambda_l1, lambda_smooth, t(fg_mask), lambda_fg)
    loss.backward()
    return float(loss.detach()), float(scale.detach()), float(shift.detach()), d.grad[0].numpy()

