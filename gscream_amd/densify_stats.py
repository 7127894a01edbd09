"""Densification statistics on the HIP path (second half).

`training_statis(model, viewspace_point_tensor, opacity, update_filter, offset_selection_mask, anchor_visible_mask)`
mirrors GaussianModel.training_statis (scene/gaussian_model.py:730-757; `model` takes the place of `self`): it
accumulates, in place, `model.opacity_accum[N,1]`, `model.anchor_demon[N,1]`, `model.offset_gradient_accum[N*K,1]`
and `model.offset_denom[N*K,1]` from one rendered view -- one kernel (gsr_training_stats) instead of ~15 boolean-mask
indexing ops.  Arguments as in the reference: `opacity` = the decode's neural_opacity [Nv*K,1], `offset_selection_mask`
its mask [Nv*K], `update_filter` = radii > 0 over the decoded Gaussians [M], `viewspace_point_tensor.grad` [M,3]."""
import ctypes

import torch

from . import _native

__all__ = ["training_statis"]


def training_statis(model, viewspace_point_tensor, opacity, update_filter, offset_selection_mask, anchor_visible_mask):
    lib = _native.load()
    K = int(model.n_offsets)
    acc = model.opacity_accum
    if not acc.is_cuda:
        raise RuntimeError("gscream_amd.densify_stats: tensors must be on a HIP device (no CPU fallback)")
    for t in (model.opacity_accum, model.anchor_demon, model.offset_gradient_accum, model.offset_denom):
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise ValueError("the accumulators must be contiguous float32 tensors (they are updated in place)")
    dev = acc.device
    uf = update_filter.detach().reshape(-1)
    uf = (uf if uf.dtype == torch.uint8 else uf.view(torch.uint8) if uf.dtype == torch.bool else uf.to(torch.uint8)).contiguous()
    grad = viewspace_point_tensor.grad.detach().contiguous().float()
    M = int(uf.shape[0])
    if M != int(grad.shape[0]):
        raise ValueError("update_filter and viewspace_point_tensor.grad disagree on the number of Gaussians")
    sel = offset_selection_mask.detach().reshape(-1)
    sel = (sel if sel.dtype == torch.uint8 else sel.view(torch.uint8) if sel.dtype == torch.bool else sel.to(torch.uint8)).contiguous()
    nop = opacity.detach().reshape(-1).contiguous().float()
    book = getattr(offset_selection_mask, "_gsr_decode", None)
    if book is not None and book.matches(anchor_visible_mask, K) and sel.numel() == book.N * K:
        # the selection mask is the one this package's decode returned for this very visibility mask: its bookkeeping (row
        # list of the visible anchors, first output row per anchor, rows kept) is reused -- no torch kernels, no host sync
        vis, first, Nv, kept = book.vis, book.first, book.N, book.M
    else:
        vis = torch.nonzero(anchor_visible_mask, as_tuple=False).view(-1).int()
        Nv = int(vis.shape[0])
        if sel.numel() != Nv * K:
            raise ValueError(f"offset_selection_mask has {sel.numel()} entries, expected visible anchors * n_offsets = {Nv * K}")
        # first output row of each visible anchor = exclusive scan of its kept offsets (the decode's `first`)
        counts = sel.view(Nv, K).sum(1, dtype=torch.int32) if Nv else torch.zeros(0, dtype=torch.int32, device=dev)
        first = (torch.cumsum(counts, 0, dtype=torch.int32) - counts).contiguous()
        kept = int(counts.sum()) if Nv else 0
    # the reference's `combined_mask[temp_mask] = update_filter` raises a shape error when the selection mask comes from
    # another render than update_filter; same here (the kernel additionally never reads beyond row M)
    if kept != M:
        raise ValueError(f"offset_selection_mask keeps {kept} offsets but update_filter has {M} entries (masks of different renders?)")
    with torch.cuda.device(dev):
        _native.check(lib.gsr_training_stats(Nv, K, M, _native.ptr(vis), _native.ptr(nop), _native.ptr(sel), _native.ptr(first),
                                             _native.ptr(uf), _native.ptr(grad), _native.ptr(model.opacity_accum),
                                             _native.ptr(model.anchor_demon), _native.ptr(model.offset_gradient_accum),
                                             _native.ptr(model.offset_denom),
                                             ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "gsr_training_stats")
