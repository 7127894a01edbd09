"""Fused neural-Gaussian decode on the HIP path -- SURVEY 8(f) rank 1, the step right before the rasterizer.

`generate_neural_gaussians(viewpoint_camera, pc, visible_mask=None, is_training=False)` mirrors GScream's
gaussian_renderer/__init__.py:18-102 (same arguments, same return tuple, same row order) for a `pc` that exposes what
the reference's GaussianModel does: `_anchor_feat`, `get_anchor`, `_offset`, `get_scaling`, `n_offsets`,
`use_feat_bank`, `get_opacity_mlp`, `get_uncertainty_mlp`, `get_color_mlp`, `get_cov_mlp`
(nn.Sequential(Linear(36,32), ReLU, Linear(32,out)[, act]) as in scene/gaussian_model.py:118-144).

The visible-anchor gather (:25-28) is folded into the kernels (the mask becomes a row list once); view vector, four
MLPs, opacity mask, boolean-mask compaction, post-processing run in `gsr_decode_count` / `gsr_decode_emit`
(include/gsraster.h), the backward in `gsr_decode_backward` (input gradients and all 16 weight / bias
gradients in one native call on the f32 matrix cores; no library GEMM on the path).
`use_feat_bank=True` (off in every GScream config, arguments/__init__.py:57) runs the bank MLP + blend as a torch pre-step.
No CPU fallback."""
import ctypes
import threading

import torch

from . import _native

__all__ = ["generate_neural_gaussians", "decode"]


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _weight_array(ws):
    arr = (ctypes.c_void_p * 16)(*[w.data_ptr() for w in ws])
    return arr


class _Decode(torch.autograd.Function):
    """inputs: feat[N,32], anchor[N,3], offsets[N,K,3], grid_scaling[N,6], campos[3], then 16 weight tensors in the order
    {w1[4], b1[4], w2[4], b2[4]} for the MLPs {opacity, uncertainty, color, cov}."""

    @staticmethod
    def forward(ctx, feat, anchor, offsets, gscale, campos, vis_idx, vis_mask, *weights):
        lib = _native.load()
        if not feat.is_cuda:
            raise RuntimeError("gscream_amd.neural_gaussians: tensors must be on a HIP device (no CPU fallback)")
        K = int(offsets.shape[1])
        # Three ways to say which anchors: every row; an explicit row list (its length known on the host); or the boolean mask
        # alone -- then the row list is compacted ON THE DEVICE and its length read back together with the output row count,
        # after the emit pass is enqueued: no host round trip in front of the decode (torch.nonzero, like the reference's
        # x[visible_mask], drains the stream first, and that was the one point of a training iteration where the GPU ran dry).
        device_rows = vis_idx is None and vis_mask is not None
        if vis_mask is not None and vis_mask.numel() != anchor.shape[0]:
            # the reference's x[visible_mask] raises an indexing error here; the row-compaction kernel would read past the mask
            raise IndexError(f"The shape of the mask {list(vis_mask.shape)} at index 0 does not match the shape of the indexed "
                             f"tensor {list(anchor.shape)} at index 0")
        N = int(anchor.shape[0]) if vis_idx is None else int(vis_idx.shape[0])  # anchors decoded (device_rows: the upper bound)
        vis = None if vis_idx is None else vis_idx.detach().contiguous().int()
        if feat.shape[1] != 32:
            raise NotImplementedError("feat_dim must be 32 (arguments/__init__.py:50)")
        dev = feat.device
        f32 = lambda t: t.detach().contiguous().float()
        feat_c, anchor_c, off_c, gs_c, cam_c = f32(feat), f32(anchor), f32(offsets), f32(gscale), f32(campos)
        ws = [f32(w) for w in weights]
        warr = _weight_array(ws)
        with torch.cuda.device(dev):
            nop = torch.empty((N * K, 1), dtype=torch.float32, device=dev)
            mask = torch.empty((N * K,), dtype=torch.uint8, device=dev)
            count = torch.empty((max(N, 1),), dtype=torch.uint8, device=dev)
            first = torch.empty((max(N, 1),), dtype=torch.int32, device=dev)
            total = torch.empty((2,), dtype=torch.int32, device=dev)  # [0] output rows (count pass), [1] visible anchors (row compaction)
            scratch = torch.empty((N // 256 + 2,), dtype=torch.int32, device=dev)
            vis_count = None
            if device_rows:
                vmask8 = vis_mask.detach().contiguous().view(torch.uint8)
                vis = torch.empty((max(N, 1),), dtype=torch.int32, device=dev)
                vis_count = total[1:]
                _native.check(lib.gsr_decode_visible_rows(N, _native.ptr(vmask8), _native.ptr(vis), _native.ptr(vis_count), _native.ptr(scratch),
                                                          _stream()), "gsr_decode_visible_rows")
            _native.check(lib.gsr_decode_count(N, K, warr, _native.ptr(vis), _native.ptr(vis_count), _native.ptr(feat_c), _native.ptr(anchor_c), _native.ptr(cam_c),
                                               _native.ptr(nop), _native.ptr(mask), _native.ptr(count), _native.ptr(first),
                                               _native.ptr(total), _native.ptr(scratch), _stream()), "gsr_decode_count")
            # The row count M has to reach the host (the outputs' shapes), as it does in the reference's boolean-mask indexing --
            # but the GPU need not wait for that round trip: the emit pass only needs the per-anchor first rows, which are on
            # the device, so it is enqueued BEFORE the read against buffers provisioned for every offset (N*K rows, 60 B each),
            # and the outputs are the first M rows of those.  (Before: count -> host reads M -> allocate -> emit, ~36 us of GPU
            # idle per iteration on either side of the read.)
            pin, ev = _readback(dev)
            pin.copy_(total, non_blocking=True)  # asynchronous D2H into pinned memory, ordered behind the count pass
            ev.record()
            cap = N * K
            e = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
            xyz, color, opacity, unc, scaling, rot = e(cap, 3), e(cap, 3), e(cap, 1), e(cap, 1), e(cap, 3), e(cap, 4)
            _native.check(lib.gsr_decode_emit(N, K, warr, _native.ptr(vis), _native.ptr(vis_count), _native.ptr(feat_c), _native.ptr(anchor_c), _native.ptr(off_c),
                                              _native.ptr(gs_c), _native.ptr(cam_c), _native.ptr(nop), _native.ptr(mask), _native.ptr(first),
                                              _native.ptr(xyz), _native.ptr(color), _native.ptr(opacity), _native.ptr(unc),
                                              _native.ptr(scaling), _native.ptr(rot), _stream()), "gsr_decode_emit")
            ev.synchronize()   # waits for the count pass only: the copy was enqueued in front of the emit pass
            M = int(pin[0])
            xyz, color, opacity, unc, scaling, rot = xyz[:M], color[:M], opacity[:M], unc[:M], scaling[:M], rot[:M]
            if device_rows:  # the buffers were provisioned for every model row: cut them to the visible anchors
                N = int(pin[1])
                vis, nop, mask, first = vis[:N], nop[:N * K], mask[:N * K], first[:N]
        # the boolean mask the row list came from (one byte per model row), if the caller has it: lets the backward zero the hidden
        # rows only instead of zero-filling the model-sized gradients.  Saved through autograd, whose version check turns an in-place
        # change of the mask between forward and backward into an error instead of wrong zeros.
        vmask_saved = None if (vis is None or vis_mask is None) else vis_mask.detach().contiguous().view(torch.uint8)
        ctx.has_vis_mask = vmask_saved is not None
        ctx.save_for_backward(feat_c, anchor_c, off_c, gs_c, cam_c, mask, first, *ws, *([vmask_saved] if vmask_saved is not None else []))
        ctx.vis = vis
        ctx.dims = (N, K, M)
        ctx.in_shapes = [tuple(t.shape) for t in (feat, anchor, offsets, gscale)] + [tuple(w.shape) for w in weights]
        bmask = mask.bool()
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(nop, bmask)
        _last_decode.update(vis=vis, first=first, N=N, K=K, M=M)  # handed to densify_stats through the selection mask (see decode)
        return xyz, color, opacity, unc, scaling, rot, nop, bmask

    @staticmethod
    def backward(ctx, g_xyz, g_color, g_opacity, g_unc, g_scaling, g_rot, _g_nop, _g_mask):
        lib = _native.load()
        feat_c, anchor_c, off_c, gs_c, cam_c, mask, first, *ws = ctx.saved_tensors
        vis_mask8 = ws.pop() if ctx.has_vis_mask else None
        N, K, M = ctx.dims
        dev = feat_c.device
        # an output the loss never touched arrives as None (set_materialize_grads(False)) and travels as NULL: the kernel reads zeros
        z = lambda g, c: (None if g is None else g.detach().contiguous().float())  # noqa: E731
        g_xyz, g_color, g_opacity, g_unc, g_scaling, g_rot = z(g_xyz, 3), z(g_color, 3), z(g_opacity, 1), z(g_unc, 1), z(g_scaling, 3), z(g_rot, 4)
        warr = _weight_array(ws)
        with torch.cuda.device(dev):
            e = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
            vis = ctx.vis
            full = feat_c.shape[0]  # model-sized gradients; rows outside `vis` stay zero
            # model-sized gradients; with a row list the rows outside it must be zero: ONE fill for the four tensors
            per = 32 + 3 + 3 * K + 6
            hidden_by_kernel = vis is not None and vis_mask8 is not None and vis_mask8.numel() == full
            flatg = torch.zeros((full * per,), dtype=torch.float32, device=dev) if (vis is not None and not hidden_by_kernel) else e(full * per)
            d_feat = flatg[:full * 32].view(full, 32)
            d_anchor = flatg[full * 32:full * 35].view(full, 3)
            d_off = flatg[full * 35:full * (35 + 3 * K)].view(full, K, 3)
            d_gs = flatg[full * (35 + 3 * K):].view(full, 6)
            # one native call: input / geometry gradients and all 16 weight / bias gradients (accumulated in registers on the
            # f32 matrix cores, workgroup partials added in a fixed order) straight into the gradient tensors
            outs = (K, K, 3 * K, 7 * K)
            shapes = [(32, 36)] * 4 + [(32,)] * 4 + [(outs[m], 32) for m in range(4)] + [(outs[m],) for m in range(4)]
            sizes = [int(torch.Size(sh).numel()) for sh in shapes]
            flat = e(sum(sizes))  # one allocation for the 16 gradients (every element is written by the kernel)
            views, at = [], 0
            for sh, n in zip(shapes, sizes):
                views.append(flat[at:at + n].view(sh))
                at += n
            gw1, gb1, gw2, gb2 = views[0:4], views[4:8], views[8:12], views[12:16]
            garr = (ctypes.c_void_p * 16)(*[g.data_ptr() for g in gw1 + gb1 + gw2 + gb2])
            wsp = _workspace(dev, lib.gsr_decode_weight_grad_workspace_bytes())
            _native.check(lib.gsr_decode_backward(
                N, K, warr, _native.ptr(vis), _native.ptr(feat_c), _native.ptr(anchor_c), _native.ptr(off_c), _native.ptr(gs_c),
                _native.ptr(cam_c), _native.ptr(mask), _native.ptr(first), _native.ptr(g_xyz), _native.ptr(g_color), _native.ptr(g_opacity),
                _native.ptr(g_unc), _native.ptr(g_scaling), _native.ptr(g_rot), _native.ptr(d_feat), _native.ptr(d_anchor),
                _native.ptr(d_off), _native.ptr(d_gs), _native.ptr(wsp), garr, _stream()), "gsr_decode_backward")
            if hidden_by_kernel:
                _native.check(lib.gsr_decode_zero_hidden_rows(full, K, _native.ptr(vis_mask8), _native.ptr(d_feat), _native.ptr(d_anchor),
                                                              _native.ptr(d_off), _native.ptr(d_gs), _stream()), "gsr_decode_zero_hidden_rows")
        grads_w = gw1 + gb1 + gw2 + gb2
        sh = ctx.in_shapes
        return (d_feat.reshape(sh[0]), d_anchor.reshape(sh[1]), d_off.reshape(sh[2]), d_gs.reshape(sh[3]), None, None, None,
                *[g.reshape(s) for g, s in zip(grads_w, sh[4:])])


_tls = threading.local()  # per host thread: the last decode's bookkeeping and the read-back word (two threads may decode at once)


class _LastDecode:
    """dict-like view of this thread's slot"""

    def _d(self):
        d = getattr(_tls, "last", None)
        if d is None:
            d = _tls.last = {}
        return d

    def update(self, **kw):
        self._d().update(**kw)

    def clear(self):
        self._d().clear()

    def __bool__(self):
        return bool(self._d())

    def __getitem__(self, k):
        return self._d()[k]


_last_decode = _LastDecode()


def _readback(dev):
    """(pinned int32[2], event) per host thread and device: the row counts travel through them without a stream-wide
    synchronisation."""
    cache = getattr(_tls, "readback", None)
    if cache is None:
        cache = _tls.readback = {}
    r = cache.get(dev.index)
    if r is None:
        r = cache[dev.index] = (torch.zeros(2, dtype=torch.int32).pin_memory(), torch.cuda.Event())
    return r



class DecodeBookkeeping:
    """What the decode knows about the rows it produced, attached to the selection mask it returns (attribute `_gsr_decode`):
    `training_statis` (densify_stats.py) receives that very tensor from train.py:599 and finds the row list of the visible
    anchors, each anchor's first output row and the row count here instead of re-deriving them with a dozen small torch
    kernels and two host syncs per iteration.  Plain data; absent or stale -> the consumer recomputes everything."""
    __slots__ = ("vis", "first", "N", "K", "M", "visible_mask_ref", "visible_mask_version")

    def __init__(self, vis, first, N, K, M, visible_mask):
        import weakref
        self.vis, self.first, self.N, self.K, self.M = vis, first, N, K, M
        self.visible_mask_ref = None if visible_mask is None else weakref.ref(visible_mask)
        self.visible_mask_version = None if visible_mask is None else visible_mask._version

    def matches(self, anchor_visible_mask, n_offsets):
        if self.K != int(n_offsets) or self.visible_mask_ref is None:
            return False
        m = self.visible_mask_ref()
        return m is anchor_visible_mask and m._version == self.visible_mask_version


def _workspace(dev, nbytes):
    """Scratch for the weight-gradient partials (~20 MB; every byte read is written first).  Allocated per call through
    torch's caching allocator: it is stream-aware, so two backwards in flight on different streams of one device never
    share the buffer, and in a training loop the same block comes back every iteration at the cost of one torch.empty."""
    return torch.empty((nbytes,), dtype=torch.uint8, device=dev)


def _mlp_tensors(mlp):
    lin = [m for m in mlp if isinstance(m, torch.nn.Linear)]
    if len(lin) != 2 or lin[0].in_features != 36 or lin[0].out_features != 32 or lin[1].in_features != 32:
        raise NotImplementedError("expected Sequential(Linear(36,32), ReLU, Linear(32,out)[, act]) (scene/gaussian_model.py:118-144)")
    return lin[0].weight, lin[0].bias, lin[1].weight, lin[1].bias


def decode(feat, anchor, offsets, grid_scaling, campos, opacity_mlp, uncertainty_mlp, color_mlp, cov_mlp, visible_idx=None, visible_mask=None):
    """-> xyz, color, opacity, uncertainty, scaling, rot, neural_opacity, mask.  `visible_idx` (int tensor of anchor rows,
    ascending) folds the reference's visible-anchor gather into the kernels; None decodes every row.  `visible_mask` (the boolean
    mask that row list was made from, optional) spares the backward a model-sized zero-fill."""
    t = [_mlp_tensors(m) for m in (opacity_mlp, uncertainty_mlp, color_mlp, cov_mlp)]
    weights = [t[m][i] for i in range(4) for m in range(4)]  # {w1[4], b1[4], w2[4], b2[4]}
    return _Decode.apply(feat, anchor, offsets, grid_scaling, campos, visible_idx, visible_mask, *weights)


def _generate_with_feature_bank(viewpoint_camera, pc, visible_mask, is_training):
    """gaussian_renderer/__init__.py:39-49: the view-adaptive feature bank (use_feat_bank=True; off in every shipped
    config, arguments/__init__.py:57).  The bank MLP (4 -> 32 -> 3, softmax) and the three-resolution blend of the anchor
    feature are a per-anchor pre-step in front of the fused decode: done with torch ops on the gathered rows (autograd
    carries their backward), then the same kernels run on the blended feature."""
    if visible_mask is None:
        visible_mask = torch.ones(pc.get_anchor.shape[0], dtype=torch.bool, device=pc.get_anchor.device)
    feat, anchor = pc._anchor_feat[visible_mask], pc.get_anchor[visible_mask]
    grid_offsets, grid_scaling = pc._offset[visible_mask], pc.get_scaling[visible_mask]
    ob_view = anchor - viewpoint_camera.camera_center
    ob_dist = ob_view.norm(dim=1, keepdim=True)
    ob_view = ob_view / ob_dist
    bank_weight = pc.get_featurebank_mlp(torch.cat([ob_view, ob_dist], dim=1)).unsqueeze(dim=1)  # [n, 1, 3]
    f = feat.unsqueeze(dim=-1)
    f = (f[:, ::4, :1].repeat([1, 4, 1]) * bank_weight[:, :, :1] + f[:, ::2, :1].repeat([1, 2, 1]) * bank_weight[:, :, 1:2]
         + f[:, ::1, :1] * bank_weight[:, :, 2:])
    out = decode(f.squeeze(dim=-1), anchor, grid_offsets, grid_scaling, viewpoint_camera.camera_center, pc.get_opacity_mlp,
                 pc.get_uncertainty_mlp, pc.get_color_mlp, pc.get_cov_mlp, None)
    return out if is_training else out[:6]


def generate_neural_gaussians(viewpoint_camera, pc, visible_mask=None, is_training=False):
    if getattr(pc, "use_feat_bank", False):
        return _generate_with_feature_bank(viewpoint_camera, pc, visible_mask, is_training)
    # gaussian_renderer/__init__.py:20-28: `x[visible_mask]` for four tensors.  Here the mask becomes a row list once
    # and the kernels read (and, in the backward, write) the model-sized tensors through it; no mask = every row.
    on_device = visible_mask is not None and visible_mask.dtype == torch.bool and visible_mask.is_cuda and visible_mask.dim() == 1
    # (a boolean mask on the device is compacted there; anything else -- index tensors, CPU masks -- takes the host's nonzero)
    vis_idx = None if (visible_mask is None or on_device) else torch.nonzero(visible_mask, as_tuple=False).view(-1).int()
    _last_decode.clear()
    xyz, color, opacity, uncertainty, scaling, rot, neural_opacity, mask = decode(
        pc._anchor_feat, pc.get_anchor, pc._offset, pc.get_scaling, viewpoint_camera.camera_center, pc.get_opacity_mlp,
        pc.get_uncertainty_mlp, pc.get_color_mlp, pc.get_cov_mlp, vis_idx,
        visible_mask if (visible_mask is not None and visible_mask.dtype == torch.bool) else None)
    if _last_decode and visible_mask is not None:
        try:
            mask._gsr_decode = DecodeBookkeeping(_last_decode["vis"], _last_decode["first"], _last_decode["N"], _last_decode["K"],
                                                 _last_decode["M"], visible_mask)
        except Exception:  # noqa: BLE001  (a tensor subclass that refuses attributes: the consumer recomputes)
            pass
    if is_training:  # :98-102
        return xyz, color, opacity, uncertainty, scaling, rot, neural_opacity, mask
    return xyz, color, opacity, uncertainty, scaling, rot
