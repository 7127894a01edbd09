"""Naive PyTorch-CPU per-pixel alpha blend: the secondary oracle and the north-star CPU baseline.

TEST INFRASTRUCTURE ONLY (see oracle/gs_oracle.c).  A dense restatement of SURVEY Appendix A as
differentiable torch code: one [pixels x Gaussians] alpha matrix, exclusive cumprod transmittance,
rect-membership masking (A-10), straight-through 0.99 clamp (A-12, backward.cu:529,585,601).

What it is good for
  * forward images: valid everywhere (compare with the C oracle / HIP at ~1e-6 in float64),
  * gradients via autograd: valid EXCEPT for Gaussians whose view-space x/z or y/z is clamped to
    +-1.3*tan(fov/2), where the reference's hand-written backward deliberately differs from
    calculus (backward.cu:175-176,262-264; SURVEY finding 0-4).  `clamped_mask` reports those.
  * BASELINE.json config 1: "2k random Gaussians @128x128, forward-only, naive PyTorch per-pixel
    alpha-blend on CPU" -- `render(...)` with dtype=float32 is exactly that and bench.py times it.

Discontinuous decisions that must not flip between precisions (radius, tile rect, depth order) are
taken from float32 values computed the same way the reference does (passed in as `radii` etc. when
available), everything smooth is recomputed in `dtype`.
"""
import math

import numpy as np
import torch


def _rect(px, py, radii, W, H):
    gx, gy = (W + 15) // 16, (H + 15) // 16
    r = radii.to(torch.float32)
    px, py = px.to(torch.float32), py.to(torch.float32)
    x0 = torch.clamp(((px - r) / 16).to(torch.int64), 0, gx)
    y0 = torch.clamp(((py - r) / 16).to(torch.int64), 0, gy)
    x1 = torch.clamp(((px + r + 15) / 16).to(torch.int64), 0, gx)
    y1 = torch.clamp(((py + r + 15) / 16).to(torch.int64), 0, gy)
    return x0, y0, x1, y1


def render(means3D, scales, rotations, opacities, uncertainties, colors, *, W, H, tanfovx, tanfovy, viewmatrix,
           projmatrix, bg, scale_modifier=1.0, dtype=torch.float64, radii=None):
    """Returns (color[3,H,W], depth[1,H,W], unc[1,H,W], radii[P], clamped_mask[P]).  All tensor inputs
    are torch CPU tensors; gradients flow to means3D, scales, rotations, opacities, uncertainties, colors."""
    f32 = torch.float32
    P = means3D.shape[0]
    vm32, pm32 = viewmatrix.to(f32), projmatrix.to(f32)
    vm, pm = viewmatrix.to(dtype), projmatrix.to(dtype)
    m = means3D.to(dtype)
    ones = torch.ones(P, 1, dtype=dtype)
    hom = torch.cat([m, ones], 1) @ pm
    p_w = 1.0 / (hom[:, 3] + 1e-7)
    proj = hom[:, :3] * p_w[:, None]
    t = torch.cat([m, ones], 1) @ vm
    tz = t[:, 2]
    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    txtz, tytz = t[:, 0] / tz, t[:, 1] / tz
    clamped_mask = (txtz.abs() > limx) | (tytz.abs() > limy)
    tx = torch.clamp(txtz, -limx, limx) * tz
    ty = torch.clamp(tytz, -limy, limy) * tz
    fx, fy = W / (2.0 * tanfovx), H / (2.0 * tanfovy)
    # Sigma3D = R diag(s)^2 R^T with the un-normalised quaternion (forward.cu:129-145)
    q = rotations.to(dtype)
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    Rm = torch.stack([
        torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)], -1),
        torch.stack([2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)], -1),
        torch.stack([2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1)], 1)
    s = scales.to(dtype) * scale_modifier
    Mm = Rm * s[:, None, :]
    Sigma = Mm @ Mm.transpose(1, 2)
    zeros = torch.zeros_like(tz)
    J = torch.stack([torch.stack([fx / tz, zeros, -fx * tx / (tz * tz)], -1),
                     torch.stack([zeros, fy / tz, -fy * ty / (tz * tz)], -1)], 1)  # [P,2,3]
    Wr = vm[:3, :3].T  # world->camera rotation (standard form)
    A = J @ Wr
    cov = A @ Sigma @ A.transpose(1, 2)
    a, b, c = cov[:, 0, 0] + 0.3, cov[:, 0, 1], cov[:, 1, 1] + 0.3
    det = a * c - b * b
    conA, conB, conC = c / det, -b / det, a / det
    px = ((proj[:, 0] + 1.0) * W - 1.0) * 0.5
    py = ((proj[:, 1] + 1.0) * H - 1.0) * 0.5

    with torch.no_grad():
        viewz32 = (torch.cat([means3D.to(f32), torch.ones(P, 1)], 1) @ vm32)[:, 2]
        if radii is None:
            mid = 0.5 * (a + c)
            lam = mid + torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
            radii = torch.ceil(3.0 * torch.sqrt(lam)).to(torch.int32)
            radii = torch.where((viewz32 > 0.2) & (det != 0), radii, torch.zeros_like(radii))
        x0, y0, x1, y1 = _rect(px.detach(), py.detach(), radii, W, H)
        vis = (radii > 0) & ((x1 - x0) * (y1 - y0) > 0)
        radii = torch.where(vis, radii, torch.zeros_like(radii))
        # global (depth bits, index) order == every tile's order (A-9)
        dbits = viewz32.contiguous().view(torch.int32).to(torch.int64)
        order = torch.argsort(dbits * (P + 1) + torch.arange(P), stable=True)
        order = order[vis[order]]
    G = order.shape[0]
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    pixx, pixy = xs.reshape(-1).to(dtype), ys.reshape(-1).to(dtype)
    tilex, tiley = (xs.reshape(-1) // 16), (ys.reshape(-1) // 16)
    member = ((tilex[:, None] >= x0[order][None]) & (tilex[:, None] < x1[order][None])
              & (tiley[:, None] >= y0[order][None]) & (tiley[:, None] < y1[order][None]))
    dx = px[order][None, :] - pixx[:, None]
    dy = py[order][None, :] - pixy[:, None]
    power = -0.5 * (conA[order][None] * dx * dx + conC[order][None] * dy * dy) - conB[order][None] * dx * dy
    raw = opacities.to(dtype).reshape(-1)[order][None] * torch.exp(power)
    alpha = raw + (torch.clamp(raw, max=0.99) - raw).detach()  # straight-through clamp
    keep = member & (power <= 0) & (alpha.detach() >= 1.0 / 255.0)
    alpha = torch.where(keep, alpha, torch.zeros_like(alpha))
    one_minus = 1.0 - alpha
    T_incl = torch.cumprod(one_minus, 1)
    T_before = torch.cat([torch.ones(T_incl.shape[0], 1, dtype=dtype), T_incl[:, :-1]], 1)
    with torch.no_grad():
        stop = (keep & (T_incl < 1e-4)).to(torch.int8)
        stopped = torch.cummax(stop, 1).values.bool()
    w = torch.where(stopped, torch.zeros_like(alpha), alpha * T_before)
    alive = torch.where(stopped, torch.ones_like(one_minus), one_minus)
    T_final = torch.prod(alive, 1)
    col = colors.to(dtype)[order]
    Cimg = w @ col + T_final[:, None] * bg.to(dtype)[None]
    depth = t[:, 2][order]
    Dimg = w @ depth
    Uimg = w @ uncertainties.to(dtype).reshape(-1)[order]
    return (Cimg.T.reshape(3, H, W), Dimg.reshape(1, H, W), Uimg.reshape(1, H, W), radii, clamped_mask)


def render_numpy_scene(s, dtype=torch.float32):
    """Forward-only convenience for a gscream_amd.synthetic scene dict."""
    tt = lambda k: torch.from_numpy(np.ascontiguousarray(s[k]))
    with torch.no_grad():
        return render(tt("means3D"), tt("scales"), tt("rotations"), tt("opacities"), tt("uncertainties"), tt("colors"),
                      W=s["W"], H=s["H"], tanfovx=s["tanfovx"], tanfovy=s["tanfovy"], viewmatrix=tt("viewmatrix"),
                      projmatrix=tt("projmatrix"), bg=tt("bg"), scale_modifier=s["scale_modifier"], dtype=dtype)
