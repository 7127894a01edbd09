"""ctypes binding of libgsraster.so (the C ABI declared in include/gsraster.h).

This is the only place that touches the native library.  There is NO fallback: if the HIP
library is missing or a call fails, a RuntimeError is raised -- the product path never routes
through the CPU oracle or through eager PyTorch.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GSR_LIB") or os.path.join(_HERE, "libgsraster.so")  # GSR_LIB: diagnostic builds

#: every symbol include/gsraster.h declares (tests check the .so exports all of them)
EXPORTED_SYMBOLS = (
    "gsr_version", "gsr_abi_version", "gsr_last_error", "gsr_device_count", "gsr_geom_bytes", "gsr_image_bytes",
    "gsr_binning_bytes", "gsr_backward_scratch_bytes", "gsr_forward_stage1", "gsr_forward_stage2", "gsr_forward",
    "gsr_backward", "gsr_filter", "gsr_mark_visible", "gsr_profile_begin", "gsr_profile_begin_sampled", "gsr_profile_end", "gsr_stage_name",
    "gsr_loss_workspace_bytes", "gsr_rgb_loss_forward", "gsr_rgb_loss_backward", "gsr_rgb_loss_forward_window",
    "gsr_rgb_loss_backward_window",
    "gsr_knn_workspace_bytes", "gsr_knn_mean_dist2", "gsr_decode_count", "gsr_decode_emit", "gsr_decode_backward",
    "gsr_depth_loss_workspace_bytes", "gsr_depth_loss_forward", "gsr_depth_loss_backward", "gsr_training_stats",
    "gsr_decode_weight_grad_workspace_bytes", "gsr_decode_zero_hidden_rows", "gsr_decode_visible_rows", "gsr_adaptive_reset",
)
NUM_STAGES = 7
ABI_VERSION = 8  # include/gsraster.h GSR_ABI_VERSION this binding was written against


class Stage1Result(ctypes.Structure):
    _fields_ = [("num_rendered", ctypes.c_int32), ("max_tile_count", ctypes.c_int32),
                ("num_slots", ctypes.c_int32), ("num_occluded", ctypes.c_int32)]


class Tuning(ctypes.Structure):
    _fields_ = [("disable_tile_cull", ctypes.c_int32), ("disable_speculation", ctypes.c_int32),
                ("disable_partial_sort", ctypes.c_int32), ("inference", ctypes.c_int32), ("scatter_bands", ctypes.c_int32),
                ("occlusion_cut", ctypes.c_int32), ("heavy_groups", ctypes.c_int32), ("walk_depths_valid", ctypes.c_int32),
                ("walk_depths", ctypes.c_uint64)]


class Profile(ctypes.Structure):
    _fields_ = [("total_ms", ctypes.c_double * NUM_STAGES), ("launches", ctypes.c_int64 * NUM_STAGES)]


_lib = None
_c_int, _c_float, _vp = ctypes.c_int, ctypes.c_float, ctypes.c_void_p


def load():
    """Load libgsraster.so once.  Raises RuntimeError (never falls back) if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"gscream_amd: native library {LIB_PATH} is missing. Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C gscream_amd/csrc`. "
            "There is no CPU fallback for the rasterizer.")
    lib = ctypes.CDLL(LIB_PATH)
    try:
        lib.gsr_abi_version.restype = _c_int
        abi = int(lib.gsr_abi_version())
    except AttributeError:
        abi = None
    if abi != ABI_VERSION and not os.environ.get("GSR_SKIP_ABI_CHECK"):  # (the knob: A/B runs against an older build, tools/)
        raise RuntimeError(f"gscream_amd: {LIB_PATH} has C-ABI version {abi}, this binding needs {ABI_VERSION} "
                           "(rebuild with `make -C gscream_amd/csrc`)")
    lib.gsr_version.restype = ctypes.c_char_p
    lib.gsr_last_error.restype = ctypes.c_char_p
    lib.gsr_device_count.restype = _c_int
    for name in ("gsr_geom_bytes", "gsr_binning_bytes"):
        getattr(lib, name).restype = ctypes.c_size_t
        getattr(lib, name).argtypes = [_c_int]
    lib.gsr_image_bytes.restype = ctypes.c_size_t
    lib.gsr_image_bytes.argtypes = [_c_int, _c_int, _c_int]
    lib.gsr_backward_scratch_bytes.restype = ctypes.c_size_t
    lib.gsr_backward_scratch_bytes.argtypes = [_c_int, _c_int]
    lib.gsr_forward_stage1.restype = _c_int
    lib.gsr_forward_stage1.argtypes = (
        [_c_int] * 5 + [_vp, _vp, _c_float, _vp] + [_vp] * 5 + [_vp, _vp, _vp, _c_float, _c_float, _c_int]
        + [_vp, _vp, _vp, ctypes.POINTER(Stage1Result), ctypes.POINTER(Tuning), _c_int, _vp])
    lib.gsr_forward_stage2.restype = _c_int
    lib.gsr_forward_stage2.argtypes = [_c_int] * 5 + [_vp] * 7 + [ctypes.POINTER(Tuning), _c_int, _vp]
    lib.gsr_forward.restype = _c_int
    lib.gsr_forward.argtypes = (
        [_c_int] * 5 + [_vp, _vp, _c_float, _vp] + [_vp] * 5 + [_vp, _vp, _vp, _c_float, _c_float, _c_int]
        + [_vp, _vp, _vp, _vp, _c_int, _c_int, _vp, _vp, _vp, _vp, ctypes.POINTER(Stage1Result), ctypes.POINTER(Tuning), _c_int, _vp])
    lib.gsr_backward.restype = _c_int
    lib.gsr_backward.argtypes = (
        [_c_int] * 8 + [_vp] * 6 + [_c_float] + [_vp] * 5 + [_c_float, _c_float] + [_vp] * 3 + [_vp] * 4
        + [_vp] * 9 + [ctypes.POINTER(Tuning), _c_int, _vp])
    lib.gsr_filter.restype = _c_int
    lib.gsr_filter.argtypes = [_c_int] * 3 + [_vp, _vp, _c_float] + [_vp] * 4 + [_c_float, _c_float, _c_int] + [_vp] * 3 + [_c_int, _vp]
    lib.gsr_mark_visible.restype = _c_int
    lib.gsr_mark_visible.argtypes = [_c_int] + [_vp] * 5
    lib.gsr_profile_begin.restype = _c_int
    lib.gsr_profile_begin.argtypes = [ctypes.c_uint]
    if hasattr(lib, "gsr_profile_begin_sampled") or not os.environ.get("GSR_SKIP_ABI_CHECK"):  # (absent from an older build in an A/B run)
        lib.gsr_profile_begin_sampled.restype = _c_int
        lib.gsr_profile_begin_sampled.argtypes = [ctypes.c_uint, ctypes.c_uint]
    lib.gsr_profile_end.restype = _c_int
    lib.gsr_profile_end.argtypes = [ctypes.POINTER(Profile)]
    if hasattr(lib, "gsr_adaptive_reset") or not os.environ.get("GSR_SKIP_ABI_CHECK"):
        lib.gsr_adaptive_reset.restype = None
        lib.gsr_adaptive_reset.argtypes = []
    lib.gsr_stage_name.restype = ctypes.c_char_p
    lib.gsr_stage_name.argtypes = [_c_int]
    lib.gsr_loss_workspace_bytes.restype = ctypes.c_size_t
    lib.gsr_loss_workspace_bytes.argtypes = [_c_int] * 3
    lib.gsr_rgb_loss_forward.restype = _c_int
    lib.gsr_rgb_loss_forward.argtypes = [_c_int] * 3 + [_vp] * 3 + [_c_float, _c_float, _vp, _vp, _c_int, _vp]
    lib.gsr_rgb_loss_backward.restype = _c_int
    lib.gsr_rgb_loss_backward.argtypes = [_c_int] * 3 + [_vp] * 3 + [_c_float, _c_float, _vp, _vp, _vp, _vp]
    lib.gsr_rgb_loss_forward_window.restype = _c_int
    lib.gsr_rgb_loss_forward_window.argtypes = [_c_int] * 3 + [_vp] * 3 + [_c_float, _c_float, _c_int, _vp, _vp, _c_int, _vp]
    lib.gsr_rgb_loss_backward_window.restype = _c_int
    lib.gsr_rgb_loss_backward_window.argtypes = [_c_int] * 3 + [_vp] * 3 + [_c_float, _c_float, _c_int, _vp, _vp, _vp, _vp]
    lib.gsr_knn_workspace_bytes.restype = ctypes.c_size_t
    lib.gsr_knn_workspace_bytes.argtypes = [_c_int]
    lib.gsr_knn_mean_dist2.restype = _c_int
    lib.gsr_knn_mean_dist2.argtypes = [_c_int, _vp, _vp, _vp, _vp]
    lib.gsr_decode_count.restype = _c_int
    lib.gsr_decode_count.argtypes = [_c_int, _c_int] + [_vp] * 13
    lib.gsr_decode_emit.restype = _c_int
    lib.gsr_decode_emit.argtypes = [_c_int, _c_int] + [_vp] * 18
    lib.gsr_decode_visible_rows.restype = _c_int
    lib.gsr_decode_visible_rows.argtypes = [_c_int] + [_vp] * 5
    lib.gsr_decode_backward.restype = _c_int
    lib.gsr_decode_backward.argtypes = [_c_int, _c_int] + [_vp] * 22
    lib.gsr_decode_weight_grad_workspace_bytes.restype = ctypes.c_size_t
    lib.gsr_decode_weight_grad_workspace_bytes.argtypes = []
    lib.gsr_decode_zero_hidden_rows.restype = _c_int
    lib.gsr_decode_zero_hidden_rows.argtypes = [_c_int, _c_int] + [_vp] * 6
    lib.gsr_depth_loss_workspace_bytes.restype = ctypes.c_size_t
    lib.gsr_depth_loss_workspace_bytes.argtypes = [_c_int, _c_int]
    lib.gsr_depth_loss_forward.restype = _c_int
    lib.gsr_depth_loss_forward.argtypes = [_c_int, _c_int] + [_vp] * 5 + [_c_float, _c_float, _vp, _vp, _vp]
    lib.gsr_depth_loss_backward.restype = _c_int
    lib.gsr_depth_loss_backward.argtypes = [_c_int, _c_int] + [_vp] * 7
    lib.gsr_training_stats.restype = _c_int
    lib.gsr_training_stats.argtypes = [_c_int, _c_int, _c_int] + [_vp] * 11
    _lib = lib
    return lib


STAGE_NAMES = ("preprocess", "count_scan", "scatter", "tile_sort", "blend_forward", "blend_backward", "gauss_backward")
NEED_CAPACITY = 1


def profile_begin(stages=None, every=1):
    """Start timing stages with HIP events (all stages, or only the named ones to keep the stream undisturbed; every = n
    times only every n-th invocation of a stage)."""
    mask = 0 if not stages else sum(1 << STAGE_NAMES.index(s) for s in stages)
    lib = load()
    if os.environ.get("GSR_SKIP_ABI_CHECK") and not hasattr(lib, "gsr_profile_begin_sampled"):  # an older library in an A/B run (tools/gpu_ab_r3.sh)
        check(lib.gsr_profile_begin(mask), "gsr_profile_begin")
        return
    check(lib.gsr_profile_begin_sampled(mask, int(every)), "gsr_profile_begin_sampled")


def profile_end():
    """-> {stage name: (total ms, launches)} measured with HIP events on the launch stream."""
    lib = load()
    p = Profile()
    check(lib.gsr_profile_end(ctypes.byref(p)), "gsr_profile_end")
    return {lib.gsr_stage_name(i).decode(): (float(p.total_ms[i]), int(p.launches[i])) for i in range(NUM_STAGES)}


def check(rc, what):
    if rc != 0:
        msg = load().gsr_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"gscream_amd native call {what} failed (code {rc}): {msg}")


def ptr(t):
    """Device pointer of a tensor, or NULL for None / empty tensors (the reference's 'not provided')."""
    if t is None or t.numel() == 0:
        return None
    return ctypes.c_void_p(t.data_ptr())
