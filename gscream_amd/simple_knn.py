"""simple_knn.distCUDA2 on the HIP path (GScream calls it once, at initialisation:
scene/gaussian_model.py `dist2 = torch.clamp_min(distCUDA2(points), 0.0000001)`).

Mirrors `simple_knn._C.distCUDA2(points[P,3] float32 cuda) -> float32[P]` (submodules/simple-knn/spatial.cu):
the mean of the three smallest squared distances from each point to the other points."""
import ctypes

import torch

from . import _native

__all__ = ["distCUDA2"]


def distCUDA2(points):
    if not points.is_cuda:
        raise RuntimeError("gscream_amd.simple_knn.distCUDA2: points must be on a HIP device (no CPU fallback)")
    if points.dim() != 2 or points.shape[1] != 3:
        raise ValueError(f"expected [P,3] points, got {tuple(points.shape)}")
    lib = _native.load()
    pts = points.detach().contiguous().float()
    P = pts.shape[0]
    out = torch.zeros((P,), dtype=torch.float32, device=pts.device)  # the reference allocates torch.full(.., 0.0)
    if P == 0:
        return out
    with torch.cuda.device(pts.device):
        ws = torch.empty((lib.gsr_knn_workspace_bytes(P),), dtype=torch.uint8, device=pts.device)
        _native.check(lib.gsr_knn_mean_dist2(P, _native.ptr(pts), _native.ptr(out), _native.ptr(ws),
                                             ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "gsr_knn_mean_dist2")
    return out
