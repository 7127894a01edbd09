"""GPU tests at BASELINE.json's full sizes (1M Gaussians @1008x567, 2M @1920x1080): size-independent properties of
the rasterizer, and (last test) direct element-wise parity against the OpenMP oracle at 1M Gaussians.

  * partition of unity: colours == 1 and background == 1  =>  image == 1 (sum_i alpha_i T_i + T_final = 1);
    features == 1 => feature map == 1 - T_final; depth map bounded by [0, z_max];
  * binning: per-tile lists sorted by (depth, id), ranges partition [0, R), per-tile counts equal an independent
    torch recomputation from the tile rectangles, sum == num_rendered == sum(tiles_touched);
  * backward: bit-reproducible; linear in the upstream gradient; sum_g dL/dcolor[g] == sum_pix g(pix)(1 - T_final);
    the RGB-only kernel variant agrees with the full one; culled Gaussians get exact zeros.
"""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import helpers as Hh  # noqa: E402
from gscream_amd import GaussianRasterizer, _layout, set_tuning  # noqa: E402
from gscream_amd import rasterizer as RZ  # noqa: E402
from gscream_amd import synthetic as S  # noqa: E402

pytestmark = pytest.mark.gpu

# What the parity build leaves at full size.  Its arithmetic is the reference's expression evaluated with a correctly rounded
# expf on BOTH sides (libm in the oracle, ocml on the GPU), but the two expf differ in the last ulp on a small fraction of
# arguments, so among the ~1e8 (pixel, Gaussian) pairs of a 1M-Gaussian frame a few still land on opposite sides of
# alpha = 1/255.  Bounds = measured counts on the GPU box with head-room (DESIGN 6 has the measured numbers).
PRECISE_MAX_GRAD_ELEMS = 12

CONFIGS = {"config2_1M_1008x567": (1, 1_000_000, 1008, 567), "config4_2M_1920x1080": (3, 2_000_000, 1920, 1080)}


@pytest.fixture(scope="module", params=sorted(CONFIGS))
def scene(request):
    seed, P, W, H = CONFIGS[request.param]
    s = S.scene_slab(seed, P, W, H)
    dev = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in s.items() if isinstance(v, np.ndarray)}
    return s, dev, Hh.hip_settings(s)


def _forward_state(s, dev, rs, colors=None, unc=None):
    e = torch.Tensor([])
    return RZ._forward_native(dev["means3D"], e, dev["colors"] if colors is None else colors, dev["opacities"],
                              dev["uncertainties"] if unc is None else unc, dev["scales"], dev["rotations"], e, rs)


def test_partition_of_unity(scene):
    s, dev, rs = scene
    rs1 = rs._replace(bg=torch.ones(3, device="cuda"))
    R, color, depth, feat, radii, geom, binning, img, _ns = _forward_state(
        s, dev, rs1, colors=torch.ones_like(dev["colors"]), unc=torch.ones_like(dev["uncertainties"]))
    assert R > s["means3D"].shape[0]
    assert (color - 1.0).abs().max().item() < 2e-5
    fT = _layout.image_views(img, s["means3D"].shape[0], s["W"], s["H"])["final_T"]
    assert (feat[0] - (1.0 - fT)).abs().max().item() < 2e-5
    assert fT.min().item() >= 1e-4 * 0.01 - 1e-9 and fT.max().item() <= 1.0
    zmax = dev["means3D"][:, 2].max().item()
    assert depth.min().item() >= 0 and depth.max().item() <= zmax * (1 + 1e-5)


def test_binning_invariants(scene):
    s, dev, rs = scene
    P, W, H = s["means3D"].shape[0], s["W"], s["H"]
    R, color, depth, feat, radii, geom, binning, img, _ns = _forward_state(s, dev, rs)
    gv, iv, bv = _layout.geom_views(geom, P), _layout.image_views(img, P, W, H), _layout.binning_views(binning, R, _ns)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    tiles = gv["tiles"].long()   # per-Gaussian gradient-slot count = surviving tiles of its rectangle
    assert R == int(iv["info"][0]) and _ns >= R and int(tiles.sum()) == R
    ranges = iv["ranges"].long()
    counts = ranges[:, 1] - ranges[:, 0]
    assert ranges[0, 0] == 0 and bool((ranges[1:, 0] == ranges[:-1, 1]).all()) and ranges[-1, 1] == R
    assert int(counts.max()) == int(iv["info"][1])
    # independent recomputation of the tile rectangles from (pixel centre, radius) with torch float32 ops
    rec = gv["rec_f32"]
    px, py, r = rec[:, 0], rec[:, 1], radii.float()
    vis = radii > 0
    x0 = torch.clamp(((px - r) / 16).int(), 0, gx); x1 = torch.clamp(((px + r + 15) / 16).int(), 0, gx)
    y0 = torch.clamp(((py - r) / 16).int(), 0, gy); y1 = torch.clamp(((py + r + 15) / 16).int(), 0, gy)
    x0, x1, y0, y1 = [torch.where(vis, v, torch.zeros_like(v)) for v in (x0, x1, y0, y1)]
    rect = gv["rect"]
    assert bool(((rect[:, 0] & 0xffff) == x0).all() and ((rect[:, 0] >> 16) == x1).all())
    assert bool(((rect[:, 1] & 0xffff) == y0).all() and ((rect[:, 1] >> 16) == y1).all())
    # expand every (Gaussian, tile-of-rectangle) pair, keep the survivors of the tile-cull mask, histogram them
    area = ((x1 - x0) * (y1 - y0)).long()
    offs_exp = torch.cumsum(area, 0) - area
    nexp = int(area.sum())
    g_of = torch.repeat_interleave(torch.arange(P, device="cuda"), area)
    pos = torch.arange(nexp, device="cuda") - offs_exp[g_of]
    wdt = (x1 - x0).long()[g_of]
    etx, ety = x0.long()[g_of] + pos % wdt, y0.long()[g_of] + pos // wdt
    alive = (pos >= 64) | (((gv["tmask"][g_of] >> pos.clamp(max=63)) & 1) == 1)
    hist = torch.bincount((ety * gx + etx)[alive], minlength=gx * gy)
    assert bool((hist == counts).all()), "per-tile counts == surviving (Gaussian, tile) pairs"
    assert int(alive.sum()) == R and R < nexp, "tile culling must drop something on this workload"
    surv = torch.zeros(P, dtype=torch.int64, device="cuda").index_add_(0, g_of, alive.long())
    assert bool((surv == tiles).all()), "slot count per Gaussian == surviving tiles"
    offs = gv["offsets"].long()
    assert bool((offs == torch.cumsum(tiles, 0) - tiles).all()), "offsets = exclusive scan of the slot counts"
    # sortedness of every tile list by (depth bits, id), and membership of every entry in its tile's rectangle
    pl = bv["point_list"].long()
    dk = gv["depthkey"].long()[pl] & 0xffffffff
    key = dk * (P + 1) + pl
    tile_of = torch.repeat_interleave(torch.arange(gx * gy, device="cuda"), counts)
    same = tile_of[1:] == tile_of[:-1]
    assert bool((key[1:][same] > key[:-1][same]).all()), "per-tile lists strictly ascending in (depth, id)"
    tx, ty = tile_of % gx, tile_of // gx
    assert bool(((tx >= x0[pl]) & (tx < x1[pl]) & (ty >= y0[pl]) & (ty < y1[pl])).all())
    # gradient-slot map of the backward blend: offset + rank of the tile among the survivors of its rectangle must
    # be a bijection from the binned instances onto [0, R)
    csum = torch.cumsum(alive.long(), 0) - alive.long()
    e = offs_exp[pl] + (ty - y0[pl]) * (x1[pl] - x0[pl]).long() + (tx - x0[pl])
    slot = offs[pl] + (csum[e] - csum[offs_exp[pl]])
    assert bool(alive[e].all())
    assert bool((torch.sort(slot).values == torch.arange(R, device="cuda")).all())


def _rel_stats(got, ref):
    """(max, 99.99th percentile) of |got-ref| / (|ref| + 1e-3 max|ref|).  At 1M Gaussians the max over millions
    of elements is dominated by a few ill-conditioned ones (dL/dscale, dL/drot go through the covariance
    chain with heavy cancellation), so properties are asserted on the 99.99th percentile plus a loose max."""
    ref = ref.double().reshape(-1)
    rel = (got.double().reshape(-1) - ref).abs() / (ref.abs() + 1e-3 * ref.abs().max().clamp_min(1e-30))
    k = max(1, int(rel.numel() * 0.9999))
    return rel.max().item(), rel.kthvalue(k).values.item()


def _run_bwd(s, dev, rs, grads):
    leaves = {k: dev[k].clone().requires_grad_(True) for k in ("means3D", "opacities", "uncertainties", "colors", "scales", "rotations")}
    m2 = torch.zeros_like(leaves["means3D"], requires_grad=True)
    c, d, u, radii = GaussianRasterizer(raster_settings=rs)(
        leaves["means3D"], m2, leaves["opacities"], leaves["uncertainties"], colors_precomp=leaves["colors"],
        scales=leaves["scales"], rotations=leaves["rotations"])
    torch.autograd.backward([c, d, u], list(grads))
    out = {k: v.grad for k, v in leaves.items()}
    out["means2D"] = m2.grad
    return out, radii


def test_backward_properties(scene):
    s, dev, rs = scene
    W, H = s["W"], s["H"]
    g1 = [torch.from_numpy(g).cuda() for g in S.upstream_grads(5, W, H)]
    g2 = [torch.from_numpy(g).cuda() for g in S.upstream_grads(6, W, H)]
    a, radii = _run_bwd(s, dev, rs, g1)
    b, _ = _run_bwd(s, dev, rs, g1)
    for k in a:
        # no float atomics; per-wavefront LDS accumulators summed in a fixed order: two runs give the same bits, at 1M Gaussians too
        assert torch.equal(a[k], b[k]), f"{k}: the backward must be bit-reproducible (max diff {(a[k] - b[k]).abs().max().item()})"
    culled = radii <= 0
    assert int(culled.sum()) > 0
    for k in ("means3D", "opacities", "scales", "rotations", "colors", "means2D"):
        assert not a[k][culled].any(), k
    assert not a["means2D"][:, 2].any() and all(torch.isfinite(v).all() for v in a.values())
    # linearity in the upstream gradient
    c2, _ = _run_bwd(s, dev, rs, g2)
    c12, _ = _run_bwd(s, dev, rs, [x + 0.5 * y for x, y in zip(g1, g2)])
    for k in a:
        mx, p9999 = _rel_stats(c12[k], a[k] + 0.5 * c2[k])
        assert p9999 < 1e-3 and mx < 5e-2, (k, mx, p9999)
    # sum_g dL/dcolor[g, ch] = sum_pix g_ch (1 - T_final)  (each pixel's blend weights sum to 1 - T_final)
    R, color, depth, feat, radii2, geom, binning, img, _ns = _forward_state(s, dev, rs)
    fT = _layout.image_views(img, s["means3D"].shape[0], W, H)["final_T"].double()
    for ch in range(3):
        lhs = a["colors"][:, ch].double().sum().item()
        rhs = (g1[0][ch].double() * (1.0 - fT)).sum().item()
        scale = (g1[0][ch].double().abs() * (1.0 - fT)).sum().item()
        assert abs(lhs - rhs) <= 1e-4 * scale, (ch, lhs, rhs)
    lhs = a["uncertainties"].double().sum().item()
    rhs = (g1[2][0].double() * (1.0 - fT)).sum().item()
    assert abs(lhs - rhs) <= 1e-4 * (g1[2][0].double().abs() * (1.0 - fT)).sum().item()


def test_tile_culling_changes_no_pixel(scene):
    s, dev, rs = scene
    try:
        set_tuning(tile_cull=False)
        full = _forward_state(s, dev, rs)
        set_tuning(tile_cull=True)
        cull = _forward_state(s, dev, rs)
    finally:
        set_tuning()
    assert cull[0] < 0.7 * full[0], (cull[0], full[0])
    P, W, H = s["means3D"].shape[0], s["W"], s["H"]
    assert torch.equal(full[4], cull[4]), "radii must be identical with and without tile culling"
    # the per-pixel transmittance is one sequential product over the blended instances: culled instances are exactly the
    # ones that multiply it by nothing, so it must not change by a bit
    fT = [_layout.image_views(x[7], P, W, H)["final_T"] for x in (full, cull)]
    assert torch.equal(fT[0], fT[1]), "final transmittance must be bit-identical with and without tile culling"
    # the image sums are put together from per-depth-segment sums whose boundaries are list positions, and those move
    # when culled instances leave the lists: same terms, other association -> rounding of a float sum, nothing more
    for i in (1, 2, 3):
        scale = max(1.0, full[i].abs().max().item())
        assert (full[i] - cull[i]).abs().max().item() <= 2e-6 * scale, "images must agree to summation rounding"


def test_depth_checkpoints_are_consistent(scene):
    """The forward's per-pixel depth checkpoints (gsr_common.h: GSR_SEG_MAX slots of {float4, float2}; slot k = list position
    gsr_ckpt_pos(k): {T in front of it, r, g, b} {depth, feature} summed over the segment that ends there; last slot:
    {checkpoints passed, sums behind the last one}) must reproduce the images and be monotone in T."""
    s, dev, rs = scene
    P, W, H = s["means3D"].shape[0], s["W"], s["H"]
    R, color, depth, feat, radii, geom, binning, img, _ns = _forward_state(s, dev, rs)
    iv = _layout.image_views(img, P, W, H)
    N, Np = W * H, (W * H + 3) & ~3
    S = _layout.SEG_MAX
    slots = iv["ckpt"]                                   # [S][6 * Np] floats
    a = slots[:, :4 * Np].reshape(S, Np, 4)[:, :N]        # float4 part
    b = slots[:, 4 * Np:].reshape(S, Np, 2)[:, :N]        # float2 part
    npass = a[S - 1, :, 0].view(torch.int32).long()
    assert int(npass.min()) >= 0 and int(npass.max()) <= S - 1
    T = (W + 15) // 16 * ((H + 15) // 16)
    seg = 64 if T <= 4096 else 128
    ncon = iv["n_contrib"].reshape(-1).long()
    # tier-1 checkpoints sit at multiples of the segment length (tier 2 is exercised by test_second_tier_of_depth_segments)
    assert bool((npass * seg >= torch.minimum(ncon - 1, torch.full_like(ncon, _layout.SEG1 * seg)).clamp(min=0) // seg * seg).all()), \
        "a pixel passes every checkpoint in front of its last contributor"
    k = torch.arange(S - 1, device="cuda")[:, None]
    live = k < npass[None, :]                             # slots the pixel really wrote
    fT = iv["final_T"].reshape(-1)
    Tk = a[:S - 1, :, 0]
    assert bool((~live[1:] | (Tk[1:] <= Tk[:-1] * (1 + 1e-6))).all()), "T decreases from checkpoint to checkpoint"
    assert bool((~live | ((Tk <= 1.0) & (Tk >= fT[None, :] * (1 - 1e-6)))).all()), "1 >= T_k >= final T"
    sums = torch.where(live[:, :, None], a[:S - 1, :, 1:], torch.zeros_like(a[:S - 1, :, 1:])).sum(0) + a[S - 1, :, 1:]
    bg = rs.bg.to(sums.device)
    img_from_sums = sums.t().reshape(3, H, W) + fT.reshape(1, H, W) * bg[:, None, None]
    assert (img_from_sums - color).abs().max().item() < 1e-5
    dsum = torch.where(live[:, :, None], b[:S - 1], torch.zeros_like(b[:S - 1])).sum(0) + b[S - 1]
    assert (dsum[:, 0].reshape(H, W) - depth[0]).abs().max().item() < 1e-4 * max(1.0, depth.abs().max().item())
    assert (dsum[:, 1].reshape(H, W) - feat[0]).abs().max().item() < 1e-5


def test_rgb_only_variant_matches_full_kernel(scene):
    """No gradient into the depth / feature maps (None from autograd) selects the 9-reduction kernel variant; it
    must agree with the full variant fed explicit zeros."""
    s, dev, rs = scene
    g = [torch.from_numpy(x).cuda() for x in S.upstream_grads(8, s["W"], s["H"], True, False, False)]
    full, _ = _run_bwd(s, dev, rs, g)  # explicit zero tensors -> AUX kernel
    leaves = {k: dev[k].clone().requires_grad_(True) for k in ("means3D", "opacities", "uncertainties", "colors", "scales", "rotations")}
    m2 = torch.zeros_like(leaves["means3D"], requires_grad=True)
    c, d, u, radii = GaussianRasterizer(raster_settings=rs)(
        leaves["means3D"], m2, leaves["opacities"], leaves["uncertainties"], colors_precomp=leaves["colors"],
        scales=leaves["scales"], rotations=leaves["rotations"])
    (c * g[0]).sum().backward()  # depth / feature maps untouched -> None grads -> RGB-only variant
    lite = {k: v.grad for k, v in leaves.items()}
    lite["means2D"] = m2.grad
    assert not lite["uncertainties"].any()
    for k in full:
        mx, p9999 = _rel_stats(lite[k], full[k])
        assert p9999 < 1e-4 and mx < 1e-2, (k, mx, p9999)


def _full_size_report(lib, seed, P, W, H, use):
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.pop("GSR_LIB", None)
    if lib:
        env["GSR_LIB"] = os.path.join(root, "gscream_amd", lib)
    cmd = [sys.executable, os.path.join(root, "tools", "full_size_oracle_check.py"), str(seed), str(P), str(W), str(H)] + [str(int(u)) for u in use]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    return json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])


@pytest.mark.parametrize("build", ["shipped", "precise"])
@pytest.mark.parametrize("seed,P,W,H,use", [(1, 1_000_000, 1008, 567, (True, False, False)),   # config 2: RGB-only gradients
                                            (2, 1_000_000, 1008, 567, (True, True, True)),     # config 3: all three maps
                                            (3, 2_000_000, 1920, 1080, (True, True, True))])   # config 4: 2M @1080p
def test_full_size_element_wise_parity(seed, P, W, H, use, build):
    """Direct element-wise parity at BASELINE's size: the OpenMP oracle does 1M Gaussians @1008x567 in about a second
    per pass on the GPU box's host cores.  Radii bit-exact for every Gaussian.
    shipped build: v_exp_f32 / v_rcp_f32 / pre-scaled quadratic form, with both discontinuous decisions of the walk settled by the
      reference's own expressions -- alpha = 1/255 inside a guard band (round 4), the T = 1e-4 stop by an exact replay of the pixels
      that end near it (round 5; blend.hip).  Measured: 0 pixels beyond 1e-4 on all three configs, 0 pixels whose walk ends at
      another Gaussian than the oracle's (4 / 1 / 6 without the replay), 1 - 4 gradient elements beyond 1e-3, every one of them inside
      the reference algorithm's own summation-order range.
    parity build (libgsraster_precise.so: the reference's own expression, libm expf, IEEE division, no contraction): the same gate."""
    r = _full_size_report("libgsraster_precise.so" if build == "precise" else None, seed, P, W, H, use)
    assert r["lib"] == ("libgsraster_precise.so" if build == "precise" else "libgsraster.so")
    assert r["radii_equal"]
    print(f"\n[{build}] P={P} {W}x{H}: " + ", ".join(f"{k}: {v['gt_1e-4']} px > 1e-4 (max {v['max']:.2e})" for k, v in r["images"].items()))
    print(f"[{build}] gradients: " + ", ".join(f"{k}: {v['n_bad']} > 1e-3 (max {v['max']:.2e})" for k, v in r["grads"].items()))
    pc = r["parity_check"]
    print(f"[{build}] by cause: pixels {pc['px_by_cause']}, gradient elements {pc['grad_elems_by_cause']}; pixels at risk {pc['pixels_at_risk']}")
    # Round 5: the gate is the BAR, with the two things no implementation can pin against an fp32 / unordered-atomic reference named
    # and measured instead of being given a blanket allowance:
    #  * a pixel may differ by more than 1e-4 only if the ORACLE's own walk of it holds a pair within 1e-6 (relative) of alpha = 1/255:
    #    an expf tie, decided by the libm (glibc in the oracle, ocml on the GPU, CUDA's in the reference);
    #  * a gradient element may differ by more than 1e-3 only if our value lies INSIDE the range the reference algorithm's own
    #    unordered fp32 atomicAdd sums span (oracle.backward_envelope, 256 random orders), or its Gaussian is blended by such a tie pixel;
    #  * no pixel's walk may end at another Gaussian than the oracle's (the T = 1e-4 replay of the shipped build; the parity build's
    #    chain is the reference's by construction, up to one expf tie).
    ties = sum(1 for p_ in pc["outlier_pixels"] if p_["expf_tie"])
    assert pc["px_gt_1e-4"] <= ties <= 2, (pc["px_gt_1e-4"], pc["outlier_pixels"])
    for k, v in r["images"].items():
        assert v["max"] < 5e-3, (k, v)
    for k, v in r["grads"].items():
        assert v["n_bad"] <= PRECISE_MAX_GRAD_ELEMS and v["p999"] <= 1e-4, (k, v)
    assert pc["last_contributor_differs"]["pixels"] <= (0 if build == "shipped" else 1) + ties + pc["last_contributor_differs"].get("expf_tie_at_the_T_stop", 0), pc["last_contributor_differs"]
    env = pc.get("order_noise_envelope")
    if env:
        print(f"[{build}] order-noise envelope: {env['elements_inside']} of {env['elements_inside'] + env['elements_outside']} elements beyond 1e-3 lie inside the "
              f"reference algorithm's own fp32 summation-order range")
        assert env["rows_examined"] == env["rows_total"], "more outlier rows than the envelope examines"
        assert env["elements_outside_not_in_an_expf_tie_walk"] == 0, [e for e in env["elements"] if not e["inside_envelope"] and not e["in_expf_tie_walk"]]
