"""GPU parity of the fused neural-Gaussian decode (gscream_amd.neural_gaussians -> gsr_decode_*) against the CPU
oracle (float64 restatement of gaussian_renderer/__init__.py:18-102).

Tolerances (fp32 kernels vs fp64 oracle): forward values 2e-5 abs/rel, gradients 2e-4 of the largest entry of each
tensor.  The compaction mask is `tanh(z) > 0`: an fp32 pre-activation within rounding of zero may legitimately flip,
so the mask is required to agree wherever |neural_opacity| > 1e-5 and the row-wise comparison uses seeds where it
agrees everywhere (checked)."""
import copy
import ctypes
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import decode_oracle as DO  # noqa: E402

pytestmark = pytest.mark.gpu
CAM = [0.3, -0.2, -6.0]


def _pair(N, K, seed):
    ref = DO.Model(N, K, seed=seed, dtype=torch.float64)
    dut = copy.deepcopy(ref).float().cuda()
    return ref, dut


def _close(a, b, tol, what):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    if a.numel():
        err = (a - b).abs().max().item()
        assert err <= tol * max(1.0, b.abs().max().item()), (what, err)


@pytest.mark.parametrize("N,K,seed,vis", [(3000, 10, 1, False), (2571, 10, 2, True), (700, 4, 3, False), (1, 10, 4, False)])
def test_forward_and_backward_against_oracle(N, K, seed, vis):
    from gscream_amd.neural_gaussians import generate_neural_gaussians
    ref, dut = _pair(N, K, seed)
    g = torch.Generator().manual_seed(seed)
    vm = (torch.rand(N, generator=g) > 0.35) if vis else None
    cam_r = DO.Camera(torch.tensor(CAM, dtype=torch.float64))
    cam_d = DO.Camera(torch.tensor(CAM, dtype=torch.float32, device="cuda"))
    out_r = DO.generate_neural_gaussians(cam_r, ref, vm, True)
    out_d = generate_neural_gaussians(cam_d, dut, None if vm is None else vm.cuda(), True)
    names = ("xyz", "color", "opacity", "uncertainty", "scaling", "rot", "neural_opacity", "mask")
    nop_r, mask_r, mask_d = out_r[6].view(-1), out_r[7], out_d[7].cpu()
    sure = nop_r.abs() > 1e-5
    assert torch.equal(mask_r[sure], mask_d[sure]), "mask differs away from zero"
    assert torch.equal(mask_r, mask_d), "seed chosen so that no pre-activation sits within fp32 rounding of zero"
    for nm, a, b in zip(names[:7], out_d[:7], out_r[:7]):
        _close(a, b, 2e-5, nm)
    assert out_d[7].dtype == torch.bool and out_d[2].shape[1] == 1 and out_d[6].shape == (out_r[6].shape[0], 1)
    # backward: random upstream gradients on the six differentiable outputs
    ups = [torch.randn(o.shape, generator=g, dtype=torch.float64) for o in out_r[:6]]
    (sum((o * u).sum() for o, u in zip(out_r[:6], ups))).backward()
    (sum((o * u.float().cuda()).sum() for o, u in zip(out_d[:6], ups))).backward()
    pr, pd = dict(ref.named_parameters()), dict(dut.named_parameters())
    for k in pr:
        assert pd[k].grad is not None, k
        _close(pd[k].grad, pr[k].grad, 2e-4, "grad " + k)


@pytest.mark.parametrize("N,K", [(15, 10), (16, 7), (17, 1), (257, 3), (1025, 10)])
def test_group_and_offset_edges_against_oracle(N, K):
    """Anchor counts around the 16-anchor step of the matrix-core kernels and offset counts that leave rows of the
    second-layer tiles unused (K = 1, 3, 7): forward rows and every parameter gradient against the oracle.  A seed whose
    mask differs from the oracle's only through an opacity within fp32 rounding of zero is skipped over."""
    from gscream_amd.neural_gaussians import generate_neural_gaussians
    for seed in range(40, 48):
        ref, dut = _pair(N, K, seed)
        cam_r = DO.Camera(torch.tensor(CAM, dtype=torch.float64))
        cam_d = DO.Camera(torch.tensor(CAM, dtype=torch.float32, device="cuda"))
        out_r = DO.generate_neural_gaussians(cam_r, ref, None, True)
        out_d = generate_neural_gaussians(cam_d, dut, None, True)
        sure = out_r[6].view(-1).abs() > 1e-5
        assert torch.equal(out_r[7][sure], out_d[7].cpu()[sure]), "mask differs away from zero"
        if torch.equal(out_r[7], out_d[7].cpu()):
            break
    else:
        pytest.fail("no seed with an unambiguous mask")
    for nm, a, b in zip(("xyz", "color", "opacity", "uncertainty", "scaling", "rot", "neural_opacity"), out_d[:7], out_r[:7]):
        _close(a, b, 2e-5, nm)
    g = torch.Generator().manual_seed(seed)
    ups = [torch.randn(o.shape, generator=g, dtype=torch.float64) for o in out_r[:6]]
    (sum((o * u).sum() for o, u in zip(out_r[:6], ups))).backward()
    (sum((o * u.float().cuda()).sum() for o, u in zip(out_d[:6], ups))).backward()
    pr, pd = dict(ref.named_parameters()), dict(dut.named_parameters())
    for k in pr:
        assert pd[k].grad is not None, k
        _close(pd[k].grad, pr[k].grad, 2e-4, "grad " + k)


def test_backward_is_bit_reproducible():
    """Two backward passes over the same forward give the same bits in every gradient (weight gradients included: wave
    partials are added in wave order, workgroup partials in workgroup order)."""
    from gscream_amd.neural_gaussians import generate_neural_gaussians
    _, dut = _pair(20_000, 10, 5)
    cam = DO.Camera(torch.tensor(CAM, dtype=torch.float32, device="cuda"))
    params = list(dut.parameters())
    grads = []
    for _ in range(2):
        out = generate_neural_gaussians(cam, dut, None, True)
        loss = sum((o * o).sum() for o in out[:6])
        grads.append(torch.autograd.grad(loss, params, allow_unused=True))
    for a, b in zip(*grads):
        assert (a is None) == (b is None)
        if a is not None:
            assert torch.equal(a, b)


def test_eval_path_empty_and_errors():
    from gscream_amd.neural_gaussians import generate_neural_gaussians
    ref, dut = _pair(500, 10, 7)
    cam_d = DO.Camera(torch.tensor(CAM, device="cuda"))
    with torch.no_grad():
        out = generate_neural_gaussians(cam_d, dut, None, False)
    assert len(out) == 6 and out[0].shape[1] == 3 and out[5].shape[1] == 4
    assert torch.allclose(out[5].norm(dim=1), torch.ones_like(out[5][:, 0]), atol=1e-5)
    none = generate_neural_gaussians(cam_d, dut, torch.zeros(500, dtype=torch.bool, device="cuda"), True)
    assert none[0].shape == (0, 3) and none[6].shape == (0, 1) and none[7].shape == (0,)
    # backward through an empty selection: every parameter gradient exists and is exactly zero
    grads = torch.autograd.grad(sum(o.sum() for o in none[:6]), list(dut.parameters()), allow_unused=True)
    assert all(g is not None and g.abs().sum().item() == 0.0 for g in grads)
    with pytest.raises(RuntimeError):
        generate_neural_gaussians(DO.Camera(torch.tensor(CAM)), copy.deepcopy(ref).float(), None, False)
    # a visibility mask of the wrong length: the reference's x[visible_mask] raises an IndexError; so does the mirror (the
    # device-side row compaction must never be handed a mask shorter than the anchor table)
    with pytest.raises(IndexError):
        generate_neural_gaussians(cam_d, dut, torch.ones(499, dtype=torch.bool, device="cuda"), True)


def test_feeds_the_rasterizer_at_scale():
    """200k anchors x 10 offsets -> ~1M Gaussians straight into the rasterizer (the two rows back to back);
    size-independent checks: row count = mask count, rows follow boolean-mask order, unit quaternions."""
    from gscream_amd import GaussianRasterizationSettings, GaussianRasterizer, synthetic as S
    from gscream_amd.neural_gaussians import generate_neural_gaussians
    dut = DO.Model(200_000, 10, seed=11, dtype=torch.float32, spread=1.5).cuda()
    cam = DO.Camera(torch.tensor([0.0, 0.0, -6.0], device="cuda"))
    xyz, color, opacity, unc, scaling, rot, nop, mask = generate_neural_gaussians(cam, dut, None, True)
    M = int(mask.sum())
    assert xyz.shape == (M, 3) and opacity.shape == (M, 1) and 0 < M < 2_000_000
    assert torch.equal(opacity.view(-1), nop.view(-1)[mask]), "rows are in boolean-mask order"
    assert torch.allclose(rot.norm(dim=1), torch.ones(M, device="cuda"), atol=1e-5) and (opacity > 0).all()
    W, H = 1008, 567
    w2c = np.eye(4, dtype=np.float32)
    w2c[2, 3] = 6.0  # camera at z = -6 looking down +z
    view, proj, campos = S.camera_matrices(0.6, 0.6 * H / W, w2c)
    assert np.allclose(campos, [0, 0, -6], atol=1e-5)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    rs = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=0.6, tanfovy=0.6 * H / W, bg=torch.zeros(3, device="cuda"),
                                       scale_modifier=1.0, viewmatrix=t(view), projmatrix=t(proj), sh_degree=1, campos=t(campos),
                                       prefiltered=False, debug=False)
    img, depth, feat, radii = GaussianRasterizer(rs)(means3D=xyz, means2D=torch.zeros_like(xyz, requires_grad=True), opacities=opacity,
                                                     uncertainties=unc, colors_precomp=color, scales=scaling, rotations=rot)
    assert img.shape == (3, H, W) and torch.isfinite(img).all() and (radii > 0).any()
    img.mean().backward()
    assert dut._anchor_feat.grad is not None and torch.isfinite(dut._anchor_feat.grad).all() and dut._anchor_feat.grad.abs().sum() > 0
    assert dut.mlp_color[2].weight.grad.abs().sum() > 0


def test_training_statistics_against_oracle():
    """GaussianModel.training_statis (densification statistics) after a real decode + render + backward, twice
    (the accumulators add up), against the restated reference on the CPU."""
    import types
    from gscream_amd.densify_stats import training_statis
    from gscream_amd.neural_gaussians import generate_neural_gaussians
    N, K = 1200, 10
    ref, dut = _pair(N, K, 31)
    g = torch.Generator().manual_seed(31)
    mk = lambda dev: types.SimpleNamespace(n_offsets=K, opacity_accum=torch.zeros(N, 1, device=dev), anchor_demon=torch.zeros(N, 1, device=dev),
                                           offset_gradient_accum=torch.zeros(N * K, 1, device=dev), offset_denom=torch.zeros(N * K, 1, device=dev))
    acc_d, acc_r = mk("cuda"), mk("cpu")
    cam_d = DO.Camera(torch.tensor(CAM, device="cuda"))
    for it in range(2):
        vm = torch.rand(N, generator=g) > 0.3
        vmd = vm.cuda()
        xyz, color, opacity, unc, scaling, rot, nop, mask = generate_neural_gaussians(cam_d, dut, vmd, True)
        M = xyz.shape[0]
        update_filter = torch.rand(M, generator=g) > 0.4                      # stands for radii > 0
        vsp = types.SimpleNamespace(grad=torch.randn(M, 3, generator=g))
        # iteration 0: the visibility mask arrives as ANOTHER tensor object -> everything is re-derived from the masks;
        # iteration 1: the very tensors of the decode, as train.py passes them -> the decode's bookkeeping is reused
        book = getattr(mask, "_gsr_decode", None)
        assert book is not None and book.matches(vmd, K) and not book.matches(vm.cuda(), K) and book.M == M
        training_statis(acc_d, types.SimpleNamespace(grad=vsp.grad.cuda()), nop, update_filter.cuda(), mask, vm.cuda() if it == 0 else vmd)
        DO.training_statis(acc_r, vsp, nop.detach().cpu(), update_filter, mask.cpu(), vm)
    for name in ("opacity_accum", "anchor_demon", "offset_gradient_accum", "offset_denom"):
        a, b = getattr(acc_d, name).cpu(), getattr(acc_r, name)
        assert a.shape == b.shape and torch.allclose(a, b, rtol=1e-6, atol=1e-7), name
    assert acc_d.anchor_demon.max() == 2 and acc_d.offset_denom.sum() > 0


@pytest.mark.parametrize("N", [0, 1, 63, 255, 256, 257, 1000, 200_001])
def test_visible_rows_on_the_device_match_nonzero(N):
    """gsr_decode_visible_rows (the row list of the visible anchors without torch.nonzero's host round trip): same rows, same
    order, same count as torch.nonzero, for sizes around the block edges."""
    from gscream_amd import _native
    lib = _native.load()
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(N + 5)
    for frac in (0.0, 0.1, 0.9, 1.0):
        mask = torch.rand((N,), device=dev, generator=g) < frac
        rows = torch.full((max(N, 1),), -7, dtype=torch.int32, device=dev)
        cnt = torch.full((1,), -1, dtype=torch.int32, device=dev)
        scratch = torch.empty((N // 256 + 2,), dtype=torch.int32, device=dev)
        m8 = mask.view(torch.uint8) if N else torch.empty((0,), dtype=torch.uint8, device=dev)
        _native.check(lib.gsr_decode_visible_rows(N, _native.ptr(m8) if N else None, _native.ptr(rows), _native.ptr(cnt), _native.ptr(scratch),
                                                  ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "gsr_decode_visible_rows")
        want = torch.nonzero(mask).view(-1).int()
        n = int(cnt.item())
        assert n == want.numel()
        assert torch.equal(rows[:n], want)
        assert bool((rows[n:] == -7).all()), "nothing is written behind the count"


def test_absent_upstream_gradients_travel_as_null():
    """A loss that touches only some decode outputs: the others' upstream gradients are None (set_materialize_grads(False)) and
    reach the kernel as NULL pointers instead of zero tensors -- same parameter gradients, bit for bit, as explicit zeros."""
    from gscream_amd.neural_gaussians import generate_neural_gaussians
    _ref, dut = _pair(700, 10, 13)
    cam_d = DO.Camera(torch.tensor(CAM, device="cuda"))
    params = list(dut.parameters())
    out = generate_neural_gaussians(cam_d, dut, None, True)
    w = torch.rand(out[0].shape, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    sparse = torch.autograd.grad((out[0] * w).sum() + out[4].sum(), params, allow_unused=True)               # xyz + scaling only
    out = generate_neural_gaussians(cam_d, dut, None, True)
    dense = torch.autograd.grad((out[0] * w).sum() + out[4].sum() + sum((o * 0.0).sum() for o in (out[1], out[2], out[3], out[5])),
                                params, allow_unused=True)                                                      # explicit zero gradients
    for a, b in zip(sparse, dense):
        assert (a is None) == (b is None)
        if a is not None:
            assert torch.equal(a, b)
