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
import collections
import threading
import weakref
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _native

_tuning = _native.Tuning()                    # the knobs as set_tuning left them (occlusion_cut = 0, inference = 0)
_tuning_ref = _native.ctypes.byref(_tuning)  # built once: the struct is mutated in place by set_tuning
# The per-call variants -- `inference` (forwards that no backward will follow) x `occlusion_cut` (decided per device and frame) -- are four
# IMMUTABLE-between-set_tuning structs: a forward picks one and never writes a shared struct, so two host threads (one per device) cannot
# flip each other's knob between the choice and the native call.
_tuning_variants = {(inf, occ): _native.Tuning(inference=inf, occlusion_cut=occ) for inf in (0, 1) for occ in (0, 1)}
_tuning_variant_refs = {k: _native.ctypes.byref(v) for k, v in _tuning_variants.items()}


def _sync_tuning_variants():
    for (inf, occ), t in _tuning_variants.items():
        for name, _ in _native.Tuning._fields_:
            setattr(t, name, getattr(_tuning, name))
        t.inference, t.occlusion_cut = inf, occ
_capacity_hint = {}  # per-device: (binning capacity, longest-list provision) for the next speculative forward
_recent = {}         # per-device: (num_rendered, max_tile_count) of the last few forwards (training hops between views)
_RECENT_FRAMES = 8
_OCC_OFF = {"on": False, "hold": 0}
_pinned_tls = threading.local()  # per host thread and device: (pinned int32[4] that receives gsr_stage1_result, its ctypes pointer)
_last_stage1 = {}  # debugging aid: counts reported by the most recent forward


# Per-view walk depths (gsr_tuning.walk_depths): one int32[4 T] per (view, image size, device), keyed by the address of the view matrix --
# GScream's cameras keep theirs for the whole run (scene/cameras.py:64: world_view_transform is made once per Camera; train.py:414-416 pops a stack of
# those objects every epoch).  Per host thread (one per device), least recently used out first.  An entry remembers WHICH tensor it was
# recorded for (weak reference): once that tensor is gone its address may belong to another camera, and the entry starts over instead of
# ordering the new view by the old one's depths (that would only have cost the frame its dispatch order, never a result).  Forwards
# under no_grad use a view's entry when training made one but never create one: evaluation renders are not revisited.
_view_cache_on = [True]
_view_cache_tls = threading.local()
_VIEW_CACHE_MAX = 1024


class _ViewCache(collections.OrderedDict):  # (a subclass: plain dicts cannot be weakly referenced; compared by identity in the registry)
    __hash__ = object.__hash__

    def __eq__(self, other):
        return self is other


_view_caches = weakref.WeakSet()  # every host thread's cache, so that set_tuning() clears them all
_view_caches_lock = threading.Lock()


def _walk_depths(rs, dev, W, H, inference=False):
    """-> (tensor or None, valid): the array this view's forward records its quadrants' walk depths in, and whether it holds a previous
    visit's already."""
    vm = rs.viewmatrix
    if not _view_cache_on[0] or not isinstance(vm, torch.Tensor):
        return None, 0
    cache = getattr(_view_cache_tls, "cache", None)
    if cache is None:
        cache = _view_cache_tls.cache = _ViewCache()
        with _view_caches_lock:
            _view_caches.add(cache)
    key = (vm.data_ptr(), W, H, dev.index)
    hit = cache.get(key)
    if hit is not None:
        ref, depths = hit
        owner = ref()
        cache.move_to_end(key)
        if owner is vm or (owner is not None and owner.data_ptr() == key[0]):
            return depths, 1
        if inference:
            return None, 0
        cache[key] = (weakref.ref(vm), depths)  # the address changed hands: same array, recorded afresh by this forward
        return depths, 0
    if inference:
        return None, 0
    T = ((W + 15) // 16) * ((H + 15) // 16)
    t = torch.empty((4 * max(T, 1),), dtype=torch.int32, device=dev)
    cache[key] = (weakref.ref(vm), t)
    if len(cache) > _VIEW_CACHE_MAX:
        cache.popitem(last=False)
    return t, 0


_occlusion_mode = [None]  # None = automatic (per device, from the previous frames), True / False = forced
_occlusion_state = {}     # per device: {"on": bool, "hold": frames left before the next probe}


def set_tuning(tile_cull=True, speculative=True, partial_sort=True, scatter_bands=0, occlusion_cut=None, heavy_groups=None, view_cache=True):
    """Performance knobs.  Images, radii and gradients do not depend on them.
    partial_sort=False sorts every per-tile list completely (the reference's lists); by default lists longer than 2048
    entries are depth-sorted only as far as the blend is expected to walk, with a complete sort as fall-back.
    tile_cull=False bins every tile of every rectangle: the internal per-tile lists and num_rendered become
    bit-identical to the reference's.  speculative=False always uses the two-stage forward (host reads
    num_rendered, then sizes the binning workspace exactly), like the reference's blocking read-back."""
    _tuning.disable_tile_cull = 0 if tile_cull else 1
    _tuning.disable_speculation = 0 if speculative else 1
    _tuning.disable_partial_sort = 0 if partial_sort else 1
    _tuning.scatter_bands = int(scatter_bands)  # 0 = automatic (bands of tile rows per chunk in the scatter)
    # occlusion_cut: conservative per-tile occlusion cut-off in front of the binning (gsr_tuning.occlusion_cut).  None = automatic:
    # switched on for frames of large splats (>= 4 binned instances per Gaussian in the previous frame), kept while it removes at least
    # a quarter of the instances, probed again every 64 frames otherwise.  Results do not depend on it; num_rendered does.
    # heavy_groups: None = automatic, True / False = always / never launch the per-Gaussian backward's cooperative kernel for groups of
    # large splats (gsr_tuning.heavy_groups)
    _tuning.heavy_groups = 0 if heavy_groups is None else 1 if heavy_groups else 2
    # view_cache: keep every view's walk depths between its visits so that the forward dispatches its deepest walks first
    # (gsr_tuning.walk_depths; ~36 KB per 1008x567 view, at most _VIEW_CACHE_MAX views per host thread)
    _view_cache_on[0] = bool(view_cache)
    with _view_caches_lock:
        for c in list(_view_caches):
            c.clear()
    _occlusion_mode[0] = occlusion_cut
    _occlusion_state.clear()
    lib = _native._lib  # (the library's own feedback heuristics start over too: the partial sort's bet, gsraster.h gsr_adaptive_reset)
    if lib is not None and hasattr(lib, "gsr_adaptive_reset"):
        lib.gsr_adaptive_reset()
    _sync_tuning_variants()
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


_F32 = torch.float32
_EMPTY = torch.Tensor([])
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)  # the handle without building a torch.cuda.Stream


def _f32c(t, device=None):
    """contiguous fp32 view/copy; mirrors the `.contiguous().data<float>()` of DGR rasterize_points.cu:98-118"""
    if t is None:
        return None
    if t.dtype is _F32 and t.is_contiguous() and (device is None or t.device == device):
        return t  # the per-iteration case: nothing to do
    if device is not None and t.device != device:
        t = t.to(device)
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _p(t):
    """device address for the C ABI: None (NULL) for absent / empty tensors (the reference's 'not provided')"""
    return t.data_ptr() if (t is not None and t.numel() != 0) else None


def _stream_handle(index):
    """HIP stream handle of torch's current stream on device `index` (what the kernels are enqueued on)."""
    if _raw_stream is not None:
        return _raw_stream(index)
    return torch.cuda.current_stream(index).cuda_stream


def _stream():
    return _native.ctypes.c_void_p(_stream_handle(torch.cuda.current_device()))


class _on_device:
    """`with torch.cuda.device(dev)` that costs nothing when `dev` already is the current device (the training case)."""
    __slots__ = ("index", "prev")

    def __init__(self, index):
        self.index, self.prev = index, -1

    def __enter__(self):
        cur = torch.cuda.current_device()
        if cur != self.index:
            self.prev = cur
            torch.cuda.set_device(self.index)
        return self.index

    def __exit__(self, *exc):
        if self.prev >= 0:
            torch.cuda.set_device(self.prev)
        return False


def _snapshot(args):
    return tuple(a.detach().cpu().clone() if isinstance(a, torch.Tensor) else a for a in args)


def _cam(rs, device):
    return (_f32c(rs.viewmatrix, device), _f32c(rs.projmatrix, device), _f32c(rs.campos, device))


def _forward_native(means3D, sh, colors_precomp, opacities, uncertainties, scales, rotations, cov3Ds_precomp, rs, inference=False):
    """The work of `_C.rasterize_gaussians` (DGR rasterize_points.cu:35-122).  inference=True: no backward will follow
    (gsr_tuning.inference) -- same images and radii, none of the state a backward would read."""
    lib = _native.load()
    if means3D.dim() != 2 or means3D.shape[1] != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")  # rasterize_points.cu:58-60
    _require_gpu(means3D, "means3D")
    dev = means3D.device
    idx = dev.index
    P, H, W = means3D.shape[0], int(rs.image_height), int(rs.image_width)
    if P == 0:
        # Outputs are zero-filled like the reference's torch::full (rasterize_points.cu:69-72): with
        # P == 0 the kernels are skipped and the zeros are what the caller gets.
        f32 = dict(dtype=torch.float32, device=dev)
        e = torch.empty((0,), dtype=torch.uint8, device=dev)
        return (0, torch.zeros((3, H, W), **f32), torch.zeros((1, H, W), **f32), torch.zeros((1, H, W), **f32),
                torch.zeros((0,), dtype=torch.int32, device=dev), e, e.clone(), e.clone(), 0)
    empty = torch.empty
    color, depth = empty((3, H, W), dtype=_F32, device=dev), empty((1, H, W), dtype=_F32, device=dev)
    unc, radii = empty((1, H, W), dtype=_F32, device=dev), empty((P,), dtype=torch.int32, device=dev)

    # keep every converted tensor referenced until the launches are enqueued
    means3D_c, opac_c, unc_c = _f32c(means3D), _f32c(opacities), _f32c(uncertainties)
    scales_c, rot_c, cov_c = _f32c(scales, dev), _f32c(rotations, dev), _f32c(cov3Ds_precomp, dev)
    colors_c, sh_c = _f32c(colors_precomp, dev), _f32c(sh, dev)
    M = sh_c.shape[1] if (sh_c is not None and sh_c.numel() != 0) else 0
    view, proj, campos = _cam(rs, dev)
    bg = _f32c(rs.bg, dev)

    geom = empty((lib.gsr_geom_bytes(P),), dtype=torch.uint8, device=dev)
    img = empty((lib.gsr_image_bytes(P, W, H),), dtype=torch.uint8, device=dev)
    pins = getattr(_pinned_tls, "pins", None)
    if pins is None:
        pins = _pinned_tls.pins = {}
    pin = pins.get(idx)
    if pin is None:
        t = torch.zeros(4, dtype=torch.int32).pin_memory()
        pin = pins[idx] = (t, _native.ctypes.cast(t.data_ptr(), _native.ctypes.POINTER(_native.Stage1Result)))
    res = pin[1]
    debug = 1 if rs.debug else 0
    occ = _occlusion_mode[0]
    if occ is None:
        occ = _occlusion_state.get(idx, _OCC_OFF)["on"]
    variant = (1 if inference else 0, 1 if occ else 0)
    tuning = _tuning_variant_refs[variant]  # nothing shared is written per call
    walk, walk_valid = _walk_depths(rs, dev, W, H, inference)
    if walk is not None:  # this call's own copy of the knobs + the view's array
        tun = _native.Tuning.from_buffer_copy(_tuning_variants[variant])
        tun.walk_depths, tun.walk_depths_valid = walk.data_ptr(), walk_valid
        tuning = _native.ctypes.byref(tun)
    common = (P, int(rs.sh_degree), M, W, H, means3D_c.data_ptr(), _p(scales_c), float(rs.scale_modifier),
              _p(rot_c), _p(opac_c), _p(unc_c), _p(sh_c), _p(cov_c), _p(colors_c), _p(view), _p(proj), _p(campos),
              float(rs.tanfovx), float(rs.tanfovy), 1 if rs.prefiltered else 0)
    gp, ip, bgp = geom.data_ptr(), img.data_ptr(), _p(bg)
    cp, dp, up, rp = color.data_ptr(), depth.data_ptr(), unc.data_ptr(), radii.data_ptr()
    with _on_device(idx):
        stream = _stream_handle(idx)
        cap, tile_hint = _capacity_hint.get(idx, (0, 0))
        done = False
        if cap > 0 and not _tuning.disable_speculation:
            # Speculative single call: stage 2 is enqueued before the host learns num_rendered, against a workspace
            # sized from the previous frames.  No GPU-idle window; redone below only if the guess was too small.
            binning = empty((lib.gsr_binning_bytes(cap),), dtype=torch.uint8, device=dev)
            rc = lib.gsr_forward(*common, bgp, gp, ip, binning.data_ptr(), cap, tile_hint, rp, cp, dp, up, res, tuning,
                                 debug, stream)
            if rc != 0 and rc != _native.NEED_CAPACITY:
                _native.check(rc, "gsr_forward")
            done = rc == 0
        else:
            rc = lib.gsr_forward_stage1(*common, gp, ip, rp, res, tuning, debug, stream)
            if rc != 0:
                _native.check(rc, "gsr_forward_stage1")
        r = res.contents
        R, longest, nslots, occluded = int(r.num_rendered), int(r.max_tile_count), int(r.num_slots), int(r.num_occluded)
        if not done:
            cap = R
            binning = empty((lib.gsr_binning_bytes(cap),), dtype=torch.uint8, device=dev)
            rc = lib.gsr_forward_stage2(P, W, H, R, longest, bgp, gp, ip, _p(binning), cp, dp, up, tuning, debug, stream)
            if rc != 0:
                _native.check(rc, "gsr_forward_stage2")
    # Provision for the next call from the largest of the last few frames (a trainer hops between views, so the
    # previous frame alone is a poor predictor): binning capacity, and the longest tile list (sizes the LDS of the
    # per-tile sort; a tight value lets more sort workgroups be resident).  Exceeded -> GSR_NEED_CAPACITY -> stage 2
    # is redone above.
    hist = _recent.get(idx)
    if hist is None:
        hist = _recent[idx] = []
    hist.append((R, longest))
    if len(hist) > _RECENT_FRAMES:
        del hist[0]
    maxR, maxL = R, longest
    for h in hist:
        if h[0] > maxR:
            maxR = h[0]
        if h[1] > maxL:
            maxL = h[1]
    # (the pad on top of the 25 % is relative for small scenes: the library derives the scatter's band count from the capacity, and a
    # flat +65536 made it split tiny frames into bands for nothing)
    _capacity_hint[idx] = (int(1.25 * maxR) + min(65536, maxR // 2 + 1024), max(1024, int(1.25 * maxL) + 64))
    if _occlusion_mode[0] is None:  # automatic occlusion cut-off: decide for the NEXT frame on this device
        st = _occlusion_state.get(idx)
        if st is None:
            st = _occlusion_state[idx] = {"on": False, "hold": 0}
        _occlusion_next(st, P, R, occluded, was_on=bool(occ))
    _pinned_tls.last_longest = longest  # this thread's most recent forward: the autograd node keeps it for its backward
    ls = _last_stage1
    ls["num_rendered"], ls["max_tile_count"], ls["num_slots"], ls["binning_capacity"], ls["speculative"] = R, longest, nslots, cap, done
    ls["num_occluded"] = occluded
    return R, color, depth, unc, radii, geom, binning, img, cap


def _backward_native(rs, num_rendered, binning_capacity, means3D, radii, colors_precomp, sh, scales, rotations, cov3Ds_precomp,
                     geom, binning, img, g_color, g_depth, g_unc, max_tile_count=-1):
    """The work of `_C.rasterize_gaussians_backward` (DGR rasterize_points.cu:124-211)."""
    lib = _native.load()
    dev = means3D.device
    idx = dev.index
    P, H, W = means3D.shape[0], int(rs.image_height), int(rs.image_width)
    have_sh = sh is not None and sh.numel() != 0
    have_cov = cov3Ds_precomp is not None and cov3Ds_precomp.numel() != 0
    M = sh.shape[1] if have_sh else 0
    if P == 0:
        z = lambda *shape: torch.zeros(shape, dtype=_F32, device=dev)  # noqa: E731
        return (z(0, 3), z(0, 3), z(0, 1), z(0, 1), z(0, 3), z(0, 6) if have_cov else None, z(0, M, 3) if have_sh else None,
                None if have_cov else z(0, 3), None if have_cov else z(0, 4))
    mk = torch.empty  # the kernels write every row when P > 0
    g_means2D, g_colors = mk((P, 3), dtype=_F32, device=dev), mk((P, 3), dtype=_F32, device=dev)
    g_opac, g_feat = mk((P, 1), dtype=_F32, device=dev), mk((P, 1), dtype=_F32, device=dev)
    g_means3D = mk((P, 3), dtype=_F32, device=dev)
    # gradients of inputs that were not provided (empty tensors) are None: autograd ignores them, and the
    # reference's zero tensors for them would cost a fill kernel per iteration
    g_cov = mk((P, 6), dtype=_F32, device=dev) if have_cov else None
    g_sh = mk((P, M, 3), dtype=_F32, device=dev) if have_sh else None
    g_scales = None if have_cov else mk((P, 3), dtype=_F32, device=dev)
    g_rot = None if have_cov else mk((P, 4), dtype=_F32, device=dev)
    view, proj, campos = _cam(rs, dev)
    bg = _f32c(rs.bg, dev)
    # set_materialize_grads(False): an output the loss never touched arrives as None.  GScream's loss never uses
    # the uncertainty map (train.py:532) and early iterations use no depth either -> cheaper kernel variant.
    gc = _f32c(g_color, dev) if g_color is not None else torch.zeros((3, H, W), dtype=_F32, device=dev)
    # (either of the two auxiliary maps may be absent on its own: the kernel reads zeros for a NULL one, no fill kernel)
    gd = _f32c(g_depth, dev) if g_depth is not None else None
    gu = _f32c(g_unc, dev) if g_unc is not None else None
    # keep every converted tensor referenced until the launches are enqueued: a temporary freed early
    # could hand its block to the next temporary
    means3D_c, colors_c, sh_c = _f32c(means3D), _f32c(colors_precomp, dev), _f32c(sh, dev)
    scales_c, rot_c, cov_c = _f32c(scales, dev), _f32c(rotations, dev), _f32c(cov3Ds_precomp, dev)
    scratch = mk((lib.gsr_backward_scratch_bytes(P, num_rendered),), dtype=torch.uint8, device=dev)
    with _on_device(idx):
        rc = lib.gsr_backward(
            P, int(rs.sh_degree), M, W, H, int(num_rendered), int(binning_capacity), int(max_tile_count), _p(bg), means3D_c.data_ptr(),
            radii.data_ptr(), _p(colors_c), _p(sh_c), _p(scales_c), float(rs.scale_modifier), _p(rot_c),
            _p(cov_c), _p(view), _p(proj), _p(campos), float(rs.tanfovx), float(rs.tanfovy), gc.data_ptr(), _p(gd), _p(gu),
            _p(geom), _p(img), _p(binning), scratch.data_ptr(),
            g_means2D.data_ptr(), g_colors.data_ptr(), g_opac.data_ptr(), g_feat.data_ptr(),
            g_means3D.data_ptr(), _p(g_cov), _p(g_sh), _p(g_scales), _p(g_rot),
            _tuning_ref, 1 if rs.debug else 0, _stream_handle(idx))
        if rc != 0:
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
        ctx.max_tile_count = int(getattr(_pinned_tls, "last_longest", -1)) if num_rendered > 0 else -1  # this forward's longest list (sizes the backward's task grid)
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
                geom, binning, img, grad_out_color, grad_out_depth, grad_out_uncertainty, ctx.max_tile_count)
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


def _occlusion_next(st, P, R, occluded, was_on=None):
    """Automatic switch of the occlusion cut-off (gsr_tuning.occlusion_cut), per device, from the frame that was just rendered:
    P Gaussians, R binned instances, `occluded` instances the pass removed (0 when it was off).  Off -> on when the frame had at
    least 4 instances per Gaussian (large splats: long lists of which the blend walks a fraction); on -> off when the pass removed
    less than a quarter of the instances, and then not probed again for 64 frames (it costs 3-4 % where nothing covers a tile)."""
    if was_on is None:
        was_on = st["on"]
    if st["on"]:
        # judged only on a frame that really ran with the pass (was_on: what the native call was given for THIS frame)
        if was_on and occluded * 4 < R + occluded:
            st["on"], st["hold"] = False, 64
    elif st["hold"] > 0:
        st["hold"] -= 1
    elif R >= 4 * P:
        st["on"] = True
    return st


def _no_grad_needed(*tensors):
    if not torch.is_grad_enabled():
        return True
    for t in tensors:
        if t.requires_grad:
            return False
    return True


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, uncertainties, scales, rotations,
                        cov3Ds_precomp, raster_settings):
    if _no_grad_needed(means3D, means2D, sh, colors_precomp, opacities, uncertainties, scales, rotations, cov3Ds_precomp):
        # Evaluation (the reference renders under torch.no_grad() in its test / FPS loops, train.py:756-763,861-878): the
        # inference forward, outside the autograd graph.  Debug snapshots as in the training path (DGR/__init__.py:87-95).
        args = (means3D, sh, colors_precomp, opacities, uncertainties, scales, rotations, cov3Ds_precomp, raster_settings)
        if raster_settings.debug:
            saved = _snapshot(args)
            try:
                out = _forward_native(*args, inference=True)
            except Exception:
                torch.save(saved, "snapshot_fw.dump")
                print("\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.")
                raise
        else:
            out = _forward_native(*args, inference=True)
        return out[1], out[2], out[3], out[4]
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
        empty = _EMPTY  # absent inputs travel as empty tensors (DGR/__init__.py:230-240); one shared instance, never written
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
            # every element is written by the kernel (0 for culled points): no fill kernels in front of it
            radii = torch.empty((P,), dtype=torch.int32, device=dev)
            px = torch.empty((P,), dtype=torch.float32, device=dev) if want_xy else None
            py = torch.empty((P,), dtype=torch.float32, device=dev) if want_xy else None
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
