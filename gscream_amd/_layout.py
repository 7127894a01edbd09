"""Python mirror of the workspace layout in csrc/gsr_common.h (debugging / tests only):
decodes the opaque geom / image / binning byte tensors the forward returns."""
import torch


import os

# GSR_SEG1, GSR_T2_LEN, GSR_T2_N, GSR_SEG3_LEN, GSR_SEG2 (gsr_common.h): three tiers of depth segments at fixed list positions
# (the env knobs: A/B against a library built with other -D values)
SEG1 = 7
T2_LEN, T2_N = int(os.environ.get("GSR_T2_LEN", "3")), int(os.environ.get("GSR_T2_N", "6"))
SEG3_LEN = int(os.environ.get("GSR_SEG3_LEN", "8"))
SEG2 = int(os.environ.get("GSR_SEG2", "20"))
SEG_MAX = SEG1 + SEG2    # GSR_SEG_MAX: segments per tile = checkpoint slots (SEG_MAX - 1 checkpoints + the "last" slot)
CKPT_PLANES = SEG_MAX * 6
# ends of the segments behind the first tier in units of L
SEG2_ENDS = tuple(SEG1 + T2_LEN * (j + 1) for j in range(T2_N)) + tuple(SEG1 + T2_LEN * T2_N + SEG3_LEN * (j + 1) for j in range(128))


def seg2_len(n, L):
    """gsr_seg2_len: the unit of the second tier's boundaries = the launch's segment length L (64 up to 4096 tiles, 128 beyond),
    whatever the list length n."""
    return L


def ckpt_pos(k, L, unit):
    """gsr_ckpt_pos: list position of checkpoint k = 0 .. SEG_MAX-2 (fixed positions: they do not depend on the list)."""
    return (k + 1) * L if k < SEG1 else unit * SEG2_ENDS[k - SEG1]


def _align(x):
    return (x + 255) & ~255


MAX_CHUNKS, CHUNK_GAUSS = 256, 1024  # GSR_MAX_CHUNKS, GSR_CHUNK_GAUSS (gsr_common.h)


def num_chunks(P):
    return max(1, min(MAX_CHUNKS, (P + CHUNK_GAUSS - 1) // CHUNK_GAUSS))


def _take(buf, off, nbytes, dtype, shape):
    return buf[off:off + nbytes].view(dtype).reshape(shape)


def geom_views(buf, P):
    p = max(P, 1)
    off = 0
    out = {}
    out["rec_f32"] = _take(buf, off, p * 64, torch.float32, (p, 16)); out["rec_i32"] = _take(buf, off, p * 64, torch.int32, (p, 16)); off += _align(p * 64)
    out["rect"] = _take(buf, off, p * 8, torch.int32, (p, 2)); off += _align(p * 8)
    out["depthkey"] = _take(buf, off, p * 4, torch.int32, (p,)); off += _align(p * 4)
    out["tiles"] = _take(buf, off, p * 4, torch.int32, (p,)); off += _align(p * 4)
    out["offsets"] = _take(buf, off, p * 4, torch.int32, (p,)); off += _align(p * 4)
    out["tmask"] = _take(buf, off, p * 8, torch.int64, (p,)); off += _align(p * 8)
    return {k: v[:P] for k, v in out.items()}


OCC_BUCKETS = 160


XCD_CHUNK = 4   # gsr_common.h GSR_XCD_CHUNK: consecutive tiles an XCD gets at a time


def xcd_tiles(T):
    """Tile slots per XCD (gsr_common.h gsr_xcd_tiles)."""
    return (((T + XCD_CHUNK - 1) // XCD_CHUNK + 7) // 8) * XCD_CHUNK


def xcd_tile(xcd, i, T):
    """The i-th tile of XCD `xcd`, -1 = none (gsr_common.h gsr_xcd_tile)."""
    t = (xcd + 8 * (i // XCD_CHUNK)) * XCD_CHUNK + i % XCD_CHUNK
    return t if t < T else -1


def image_views(buf, P, W, H):
    gx, gy = (W + 15) // 16, (H + 15) // 16
    T, N = max(gx * gy, 1), max(W * H, 1)
    off = 0
    out = {}
    out["ranges"] = _take(buf, off, T * 8, torch.int32, (T, 2)); off += _align(T * 8)
    out["final_T"] = _take(buf, off, N * 4, torch.float32, (H, W)); off += _align(N * 4)
    out["n_contrib"] = _take(buf, off, N * 4, torch.int32, (H, W)); off += _align(N * 4)
    nb = num_chunks(P) if T <= 36864 else 1  # beyond the LDS tile limit only T cursor words (gsr_common.h)
    out["table"] = _take(buf, off, nb * T * 4, torch.int32, (nb, T)); off += _align(nb * T * 4)
    out["tile_count"] = _take(buf, off, T * 4, torch.int32, (T,)); off += _align(T * 4)
    out["tile_work"] = _take(buf, off, T * 4, torch.int32, (T,)); off += _align(T * 4)
    out["sorted_len"] = _take(buf, off, T * 4, torch.int32, (T,)); off += _align(T * 4)
    out["need_full"] = _take(buf, off, T * 4, torch.int32, (T,)); off += _align(T * 4)
    Np = (N + 3) & ~3
    nocc = 8 * T * OCC_BUCKETS if T <= 8192 else 1   # occlusion cut-off: occ_mass x 8 XCD copies ALIASES the checkpoint area (gsr_common.h)
    out["ckpt"] = _take(buf, off, CKPT_PLANES * Np * 4, torch.float32, (SEG_MAX, 6 * Np))
    out["occ_mass"] = _take(buf, off, nocc * 4, torch.int32, (nocc,))
    off += _align(max(CKPT_PLANES * Np * 4, nocc * 4))
    out["info"] = _take(buf, off, 16, torch.int32, (4,)); off += _align(16)
    out["qresume"] = _take(buf, off, 4 * T * 4, torch.int32, (4 * T,)); off += _align(4 * T * 4)
    nq = 32 * xcd_tiles(T)   # dispatch order of the forward's quadrant tasks (gsr_tuning.walk_depths)
    out["qorder"] = _take(buf, off, nq * 4, torch.int32, (8, nq // 8)); off += _align(nq * 4)
    out["occ_cut"] = _take(buf, off, T * 4, torch.int32, (T,)); off += _align(T * 4)
    out["occ_drop"] = _take(buf, off, 256 * 4, torch.int32, (256,)); off += _align(256 * 4)
    out["tile_group"] = _take(buf, off, (T // 64 + 1) * 4, torch.int32, (T // 64 + 1,)); off += _align((T // 64 + 1) * 4)
    return out


def binning_views(buf, R, capacity=None):
    """`capacity` = what the workspace was sized for (>= R after a speculative forward)."""
    r = max(R if capacity is None else capacity, 1)
    off = 0
    out = {}
    out["seg_keys"] = _take(buf, off, r * 8, torch.int64, (r,))[:R]; off += _align(r * 8)
    raw = _take(buf, off, r * 4, torch.int32, (r,))[:R]; off += _align(r * 4)
    out["point_list_raw"] = raw
    out["point_list"] = raw & 0x7fffffff   # Gaussian ids; bit 31 of an entry = the forward met it inside the alpha = 1/255 guard band
    out["band_flag"] = raw < 0
    out["slot_written"] = _take(buf, off, r, torch.uint8, (r,))[:R]; off += _align(r)
    return out
