"""CPU tests of the host side: the C-ABI library loads and exports everything include/gsraster.h declares,
workspace size queries behave, and the Python mirror keeps the reference's interface and error behaviour.
No compute call is made here (there is no GPU in the build container)."""
import inspect
import os
import re
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(native_lib):
    from gscream_amd import _native
    hdr = open(os.path.join(ROOT, "include", "gsraster.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(gsr_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_native.EXPORTED_SYMBOLS), declared ^ set(_native.EXPORTED_SYMBOLS)
    for sym in declared:
        assert hasattr(native_lib, sym), sym
    assert native_lib.gsr_version().startswith(b"gsraster")
    # the integer ABI version of the header == the library's == the binding's (a stale .so fails at load time, not in a kernel)
    m = re.search(r"#define\s+GSR_ABI_VERSION\s+(\d+)", open(os.path.join(ROOT, "include", "gsraster.h")).read())
    assert m and int(m.group(1)) == native_lib.gsr_abi_version() == _native.ABI_VERSION


def test_tuning_struct_layout_and_the_inference_twin():
    """gsr_tuning: seven knobs + the per-view walk-depth array (ABI 6: `walk_depths_valid` took the reserved word, the 64-bit address
    follows at offset 32); set_tuning keeps the inference twin of the knobs in step; a forward whose inputs need no gradient is routed
    outside the autograd node."""
    import ctypes
    from gscream_amd import _native, rasterizer as RZ
    assert ctypes.sizeof(_native.Tuning) == 40 and _native.Tuning.inference.offset == 12
    assert _native.Tuning.walk_depths_valid.offset == 28 and _native.Tuning.walk_depths.offset == 32 and _native.ABI_VERSION == 8
    RZ.set_tuning(tile_cull=False, partial_sort=False)
    try:
        assert RZ._tuning_variants[(1, 0)].inference == 1 and RZ._tuning.inference == 0 and RZ._tuning_variants[(0, 1)].occlusion_cut == 1
        assert all(v.disable_tile_cull == 1 and v.disable_partial_sort == 1 for v in RZ._tuning_variants.values())
    finally:
        RZ.set_tuning()
    assert all(v.disable_tile_cull == 0 for v in RZ._tuning_variants.values())
    a, b = torch.zeros(3, 3), torch.zeros(3, 3, requires_grad=True)
    assert RZ._no_grad_needed(a, a) and not RZ._no_grad_needed(a, b)
    with torch.no_grad():
        assert RZ._no_grad_needed(a, b)


def test_workspace_sizes(native_lib):
    P, W, H, R = 1_000_000, 1008, 567, 6_000_000
    g, i, b = native_lib.gsr_geom_bytes(P), native_lib.gsr_image_bytes(P, W, H), native_lib.gsr_binning_bytes(R)
    assert g % 256 == 0 and i % 256 == 0 and b % 256 == 0
    assert 84 * P <= g <= 100 * P          # 64-B record + rect 8 + depthkey 4 + tiles 4 + offsets 4 (+ SH flags)
    assert 13 * R <= b <= 13 * R + 4096    # 8-B sort key + 4-B sorted id + 1 "slot written" byte per instance
    assert native_lib.gsr_backward_scratch_bytes(P, R) >= 48 * R
    assert native_lib.gsr_geom_bytes(0) > 0 and native_lib.gsr_binning_bytes(0) > 0
    # layout mirror used by the tests agrees with the C carve
    from gscream_amd import _layout
    buf = torch.zeros(native_lib.gsr_image_bytes(5000, 112, 71), dtype=torch.uint8)
    v = _layout.image_views(buf, 5000, 112, 71)
    assert v["ranges"].shape == (35, 2) and v["table"].shape == (5, 35)  # 5 = ceil(5000 / 1024) binning chunks (gsr_num_chunks)
    # the last field of the mirror ends inside the buffer the library asked for (the C carve and the mirror agree on every size)
    last = v["tile_group"]
    assert last.data_ptr() + last.numel() * 4 <= buf.data_ptr() + buf.numel()
    assert _layout.num_chunks(1_000_000) == 256 and _layout.num_chunks(100_000) == 98 and _layout.num_chunks(1) == 1


def test_no_device_is_reported_not_crashed(native_lib):
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    assert native_lib.gsr_device_count() < 0
    assert b"hipGetDeviceCount" in native_lib.gsr_last_error()


def test_argument_validation_without_gpu(native_lib):
    from gscream_amd import _native
    res = _native.Stage1Result()
    rc = native_lib.gsr_forward_stage1(-1, 0, 0, 64, 64, None, None, 1.0, None, None, None, None, None, None, None, None,
                                       None, 0.5, 0.5, 0, None, None, None, _native.ctypes.byref(res), None, 0, None)
    assert rc == -1 and b"P must be" in native_lib.gsr_last_error()
    rc = native_lib.gsr_forward_stage1(10, 0, 0, 0, 64, None, None, 1.0, None, None, None, None, None, None, None, None,
                                       None, 0.5, 0.5, 0, None, None, None, _native.ctypes.byref(res), None, 0, None)
    assert rc == -1
    rc = native_lib.gsr_forward_stage1(10, 0, 0, 16 * 5000, 16 * 5000, None, None, 1.0, None, None, None, None, None, None,
                                       None, None, None, 0.5, 0.5, 0, None, None, None, _native.ctypes.byref(res), None, 0, None)
    assert rc == -3  # unsupported: 25M tiles (beyond 2^24; up to there the global-counter binning fallback applies)
    rc = native_lib.gsr_forward_stage1(0, 0, 0, 64, 64, None, None, 1.0, None, None, None, None, None, None, None, None,
                                       None, 0.5, 0.5, 0, None, None, None, _native.ctypes.byref(res), None, 0, None)
    assert rc == 0 and res.num_rendered == 0  # P == 0 short-circuits (DGR rasterize_points.cu:85)
    with pytest.raises(RuntimeError, match="native call"):
        _native.check(-1, "x")


def _settings():
    from diff_gaussian_rasterization import GaussianRasterizationSettings
    return GaussianRasterizationSettings(image_height=32, image_width=48, tanfovx=0.5, tanfovy=0.4, bg=torch.zeros(3),
                                         scale_modifier=1.0, viewmatrix=torch.eye(4), projmatrix=torch.eye(4), sh_degree=1,
                                         campos=torch.zeros(3), prefiltered=False, debug=False)


def test_public_surface_matches_reference_module():
    """Names, field order, argument order and defaults of DGR/diff_gaussian_rasterization/__init__.py:189-312."""
    import diff_gaussian_rasterization as dgr
    assert dgr.GaussianRasterizationSettings._fields == (
        "image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix", "projmatrix",
        "sh_degree", "campos", "prefiltered", "debug")
    sig = inspect.signature(dgr.GaussianRasterizer.forward)
    assert list(sig.parameters) == ["self", "means3D", "means2D", "opacities", "uncertainties", "shs", "colors_precomp",
                                    "scales", "rotations", "cov3D_precomp"]
    assert all(sig.parameters[k].default is None for k in ("shs", "colors_precomp", "scales", "rotations", "cov3D_precomp"))
    for name in ("visible_filter", "position2D_filter"):
        sig = inspect.signature(getattr(dgr.GaussianRasterizer, name))
        assert list(sig.parameters) == ["self", "means3D", "scales", "rotations", "cov3D_precomp"]
    assert list(inspect.signature(dgr.GaussianRasterizer.markVisible).parameters) == ["self", "positions"]
    assert list(inspect.signature(dgr.rasterize_gaussians).parameters) == [
        "means3D", "means2D", "sh", "colors_precomp", "opacities", "uncertainties", "scales", "rotations",
        "cov3Ds_precomp", "raster_settings"]
    r = dgr.GaussianRasterizer(raster_settings=_settings())
    assert isinstance(r, torch.nn.Module) and r.raster_settings.image_width == 48


def test_exclusivity_errors_match_reference_messages():
    from diff_gaussian_rasterization import GaussianRasterizer
    r = GaussianRasterizer(raster_settings=_settings())
    m = torch.zeros(4, 3)
    o = torch.zeros(4, 1)
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        r(m, m, o, o, scales=m, rotations=torch.zeros(4, 4))
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        r(m, m, o, o, shs=torch.zeros(4, 4, 3), colors_precomp=m, scales=m, rotations=torch.zeros(4, 4))
    with pytest.raises(Exception, match="exactly one of either scale/rotation pair or precomputed 3D covariance"):
        r(m, m, o, o, colors_precomp=m)
    with pytest.raises(Exception, match="exactly one of either scale/rotation pair or precomputed 3D covariance"):
        r(m, m, o, o, colors_precomp=m, scales=m, rotations=torch.zeros(4, 4), cov3D_precomp=torch.zeros(4, 6))


def test_product_path_has_no_cpu_fallback():
    """CPU tensors must fail loudly; nothing under gscream_amd/ may import the oracle."""
    from diff_gaussian_rasterization import GaussianRasterizer
    r = GaussianRasterizer(raster_settings=_settings())
    m = torch.rand(4, 3)
    with pytest.raises(RuntimeError, match="no CPU path"):
        r(m, torch.zeros(4, 3), torch.rand(4, 1), torch.rand(4, 1), colors_precomp=m, scales=m, rotations=torch.rand(4, 4))
    with pytest.raises(RuntimeError, match="no CPU path"):
        r.visible_filter(m, scales=m, rotations=torch.rand(4, 4))
    with pytest.raises(RuntimeError, match=r"\(num_points, 3\)"):
        r(torch.rand(4, 2), torch.zeros(4, 3), torch.rand(4, 1), torch.rand(4, 1), colors_precomp=m, scales=m,
          rotations=torch.rand(4, 4))
    pkg = os.path.join(ROOT, "gscream_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f
                assert "libgsoracle" not in src, f


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from gscream_amd import _native
    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _native.load()


def test_occlusion_cutoff_switches_itself_from_what_the_frames_look_like():
    """The host side of gsr_tuning.occlusion_cut in automatic mode (rasterizer._occlusion_next): large-splat frames turn it on, it
    stays on while it removes at least a quarter of the instances, an ineffective probe turns it off for 64 frames."""
    from gscream_amd import rasterizer as RZ
    st = {"on": False, "hold": 0}
    RZ._occlusion_next(st, P=1_000_000, R=2_700_000, occluded=0)          # bench-like frame: 2.7 instances per Gaussian
    assert st == {"on": False, "hold": 0}
    RZ._occlusion_next(st, P=950_000, R=13_500_000, occluded=0)           # large splats: on for the next frame
    assert st["on"]
    RZ._occlusion_next(st, P=950_000, R=3_500_000, occluded=10_000_000)   # removes 74 %: stays on
    assert st["on"] and st["hold"] == 0
    RZ._occlusion_next(st, P=2_000_000, R=12_900_000, occluded=100)       # config-4-like: many small splats, nothing to remove
    assert not st["on"] and st["hold"] == 64
    for _ in range(64):
        RZ._occlusion_next(st, P=2_000_000, R=12_900_000, occluded=0)
        assert not st["on"]
    RZ._occlusion_next(st, P=2_000_000, R=12_900_000, occluded=0)         # probed again after the hold
    assert st["on"]
    # forcing it through set_tuning overrides the automatic switch and clears its state
    RZ._occlusion_state[0] = {"on": True, "hold": 0}
    RZ.set_tuning(occlusion_cut=False)
    assert RZ._occlusion_mode[0] is False and not RZ._occlusion_state
    RZ.set_tuning()
    assert RZ._occlusion_mode[0] is None


def test_init_state_generator_follows_create_from_pcd():
    """synthetic.surface_point_cloud -> standin_model.voxelize / Model.from_pcd: the state scene/gaussian_model.py:301-345 leaves
    (voxelised anchors, _scaling = log(sqrt(mean 3-NN dist^2)) in all six columns, zero offsets / features, identity rotations,
    default-init MLPs), here with the exact 3-NN distances of the kNN oracle in the place of distCUDA2 (the GPU path uses this
    package's gsr_knn_mean_dist2, tested against the same oracle)."""
    import numpy as np
    from gscream_amd import standin_model as SM
    from gscream_amd import synthetic as S
    from oracle import decode_oracle as DO
    from oracle import knn_oracle as KO
    pts = S.surface_point_cloud(3, 6000)
    assert pts.shape == (6000, 3) and pts[:, 2].min() > 1.9 and pts[:, 2].max() < 10.1
    assert np.array_equal(pts, S.surface_point_cloud(3, 6000)) and not np.array_equal(pts, S.surface_point_cloud(4, 6000))
    v = SM.voxelize(pts, 0.001)
    assert v.shape[0] <= 6000 and np.abs(v / 0.001 - np.round(v / 0.001)).max() < 1e-6            # on the grid
    assert np.unique(np.round(v / 0.001), axis=0).shape[0] == v.shape[0]                         # one point per voxel
    coarse = SM.voxelize(pts, 0.5)
    assert coarse.shape[0] < 2000                                                                 # a coarse grid merges points
    d2 = np.maximum(KO.mean_dist2(v), 1e-7)
    m = SM.Model.from_pcd(torch.from_numpy(v).float(), torch.from_numpy(d2).float(), K=10, seed=1)
    assert torch.equal(m._anchor.detach(), torch.from_numpy(v).float())
    assert not m._offset.any() and not m._anchor_feat.any() and bool((m.get_rotation[:, 0] == 1).all())
    want = np.log(np.sqrt(d2)).astype(np.float32)
    assert np.allclose(m._scaling.detach().numpy(), np.repeat(want[:, None], 6, axis=1), rtol=0, atol=1e-6)
    # decoded at the origin: every offset of an anchor sits ON the anchor (zero offsets), about half pass opacity > 0, faint
    m.eval()
    with torch.no_grad():
        xyz, color, opacity, unc, scaling, rot = DO.generate_neural_gaussians(SM.Camera(torch.zeros(3)), m, None, False)
    assert 0.2 * 10 * v.shape[0] < xyz.shape[0] < 0.8 * 10 * v.shape[0]
    d = torch.cdist(xyz[:50].double(), m._anchor.detach().double()).min(dim=1).values
    assert float(d.max()) < 1e-6
    assert 0.0 < float(opacity.mean()) < 0.4 and float(opacity.max()) < 1.0


def test_segment_position_helpers_mirror_the_header():
    """gscream_amd/_layout.py seg2_len / ckpt_pos = gsr_common.h gsr_seg2_len / gsr_ckpt_pos: seven segments of L, six of 3 L, then
    segments of 8 L (round 6), all at FIXED list positions (they must not follow the list length: the occlusion cut-off shortens lists
    behind everything that blends, and moving boundaries would re-associate the forward's sums), then the rest.  (The GPU kernels and
    this mirror are compared through the checkpoints in test_second_tier_of_depth_segments.)"""
    from gscream_amd import _layout as LY
    assert LY.SEG_MAX == LY.SEG1 + LY.SEG2 == 27
    for L in (64, 128):
        for n in (10, 7 * L, 7 * L + 1, 3238, 14000):
            assert LY.seg2_len(n, L) == L
        pos = [LY.ckpt_pos(k, L, L) for k in range(LY.SEG_MAX - 1)]
        assert pos[:7] == [(k + 1) * L for k in range(7)]
        assert [b - a for a, b in zip(pos[6:-1], pos[7:])] == [m * L for m in (3,) * 6 + (8,) * 13]
        assert pos[12] == 25 * L and pos[-1] == 129 * L and all(p % 64 == 0 for p in pos)


def test_C_stub_module_imports_and_exports_the_five_entry_points():
    """diff_gaussian_rasterization/_C.py (the reference's pybind module name, DGR/ext.cpp:15-21) must at least IMPORT without a GPU:
    a syntax slip in it once reached the GPU box unnoticed because only `-m gpu` tests touched the module."""
    import importlib
    C = importlib.import_module("diff_gaussian_rasterization._C")
    for name in ("rasterize_gaussians", "rasterize_gaussians_backward", "mark_visible", "rasterize_aussians_filter", "rasterize_aussians_filter_position2D"):
        assert callable(getattr(C, name)), name


def test_tile_to_xcd_map_and_view_cache():
    """Round 5: (1) chunks of four consecutive tiles dealt round-robin to the XCDs (gsr_common.h gsr_xcd_tile, mirrored in _layout): every
    tile is some XCD's slot exactly once, for image sizes with and without a remainder.  (2) the per-view walk-depth cache of the mirror:
    keyed by the view matrix's address + image size, first visit records (valid 0), later visits order (valid 1), least recently used
    out first, off on request."""
    import torch
    from gscream_amd import _layout, rasterizer as RZ
    for T in (1, 3, 4, 31, 32, 33, 2268, 8160, 36864):
        xt = _layout.xcd_tiles(T)
        assert xt % _layout.XCD_CHUNK == 0 and 8 * xt >= T and 8 * xt < T + 8 * _layout.XCD_CHUNK + 8
        seen = [t for x in range(8) for t in (_layout.xcd_tile(x, i, T) for i in range(xt)) if t >= 0]
        assert sorted(seen) == list(range(T)), T
    Rs = lambda vm: RZ.GaussianRasterizationSettings(image_height=48, image_width=64, tanfovx=1.0, tanfovy=1.0, bg=None, scale_modifier=1.0,
                                                     viewmatrix=vm, projmatrix=None, sh_degree=0, campos=None, prefiltered=False, debug=False)
    cpu = torch.device("cpu")
    saved, RZ._VIEW_CACHE_MAX = RZ._VIEW_CACHE_MAX, 3
    try:
        RZ.set_tuning()
        a, b = torch.eye(4), torch.eye(4)
        wa, va = RZ._walk_depths(Rs(a), cpu, 64, 48)
        assert va == 0 and wa.shape == (4 * 4 * 3,) and wa.dtype == torch.int32
        wa2, va2 = RZ._walk_depths(Rs(a), cpu, 64, 48)
        assert va2 == 1 and wa2 is wa, "second visit of the view: the same array, now valid"
        wb, vb = RZ._walk_depths(Rs(b), cpu, 64, 48)
        assert vb == 0 and wb is not wa, "another view matrix: another array"
        assert RZ._walk_depths(Rs(a), cpu, 128, 48)[1] == 0, "another image size: another array"
        others = [torch.eye(4) for _ in range(3)]
        for o in others:
            RZ._walk_depths(Rs(o), cpu, 64, 48)
        assert len(RZ._view_cache_tls.cache) == 3 and RZ._walk_depths(Rs(a), cpu, 64, 48)[1] == 0, "evicted: recorded afresh"
        assert RZ._walk_depths(Rs(None), cpu, 64, 48) == (None, 0)
        # forwards under no_grad use an entry that training made, and never make one (evaluation views are not revisited)
        e = torch.eye(4)
        n = len(RZ._view_cache_tls.cache)
        assert RZ._walk_depths(Rs(e), cpu, 64, 48, True) == (None, 0) and len(RZ._view_cache_tls.cache) == n
        we, _ = RZ._walk_depths(Rs(e), cpu, 64, 48)
        assert RZ._walk_depths(Rs(e), cpu, 64, 48, True) == (we, 1)
        # an entry whose view matrix is gone does not order whatever tensor lands on its address: it starts over (same array)
        key = next(k for k, v in RZ._view_cache_tls.cache.items() if v[1] is we)
        stranger = torch.eye(4)
        RZ._view_cache_tls.cache[(stranger.data_ptr(),) + key[1:]] = RZ._view_cache_tls.cache.pop(key)
        del e
        ws, vs = RZ._walk_depths(Rs(stranger), cpu, 64, 48)
        assert ws is we and vs == 0 and RZ._walk_depths(Rs(stranger), cpu, 64, 48) == (we, 1)
        # every host thread's cache is cleared by set_tuning
        import threading
        other = {}
        th = threading.Thread(target=lambda: other.update(c=(RZ._walk_depths(Rs(a), cpu, 64, 48), RZ._view_cache_tls.cache)[1]))
        th.start(); th.join()
        assert len(other["c"]) == 1
        RZ.set_tuning()
        assert len(other["c"]) == 0 and len(RZ._view_cache_tls.cache) == 0
        RZ.set_tuning(view_cache=False)
        assert RZ._walk_depths(Rs(a), cpu, 64, 48) == (None, 0) and len(RZ._view_cache_tls.cache) == 0
    finally:
        RZ._VIEW_CACHE_MAX = saved
        RZ.set_tuning()
