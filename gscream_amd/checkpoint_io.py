"""GScream's model file formats (second half): the anchor PLY and the MLP checkpoint.

Host-side I/O only (no kernels).  The reference writes both with third-party packages -- `plyfile` for the PLY
(scene/gaussian_model.py:623-642 save_ply, :644-686 load_ply_sparse_gaussian, attribute order :502-514) and
`torch.save` for the MLPs (:975-1000).  `plyfile` is not available here; the PLY below is written and parsed
directly: the standard header `plyfile` emits for one `vertex` element of float32 properties
(`format binary_little_endian 1.0`), followed by the packed records.

PARITY OF THE LAYOUT: **unpinned** -- `plyfile` is not installed in the build container, so no file written by the reference's own
`PlyData.write` exists to compare with; the header below is the one plyfile's documentation specifies for a structured array of
'f4' fields (`ply` / `format binary_little_endian 1.0` / `element vertex N` / one `property float <name>` line per field in dtype
order / `end_header`, newline-terminated ASCII, records packed without padding).  tests/test_checkpoint_io.py writes such a file
byte by byte (independently of `save_ply`), with the optional `comment` / `obj_info` lines plyfile preserves, and checks that the
reader recovers every column and that `save_ply` emits exactly those header bytes.

    attribute order:  x y z  nx ny nz  f_offset_0..(3K-1)  f_anchor_feat_0..(F-1)  opacity  uncertainty  scale_0..5  rot_0..3
    f_offset_i       : `_offset.transpose(1, 2).flatten(1)`  (component-major: index c*K + k), undone on load
"""
import os

import numpy as np
import torch

__all__ = ["construct_list_of_attributes", "save_ply", "load_ply_sparse_gaussian", "save_mlp_checkpoints", "load_mlp_checkpoints"]

_PLY_TYPES = {"float": "<f4", "float32": "<f4", "double": "<f8", "float64": "<f8", "uchar": "u1", "uint8": "u1", "char": "i1",
              "int8": "i1", "short": "<i2", "int16": "<i2", "ushort": "<u2", "uint16": "<u2", "int": "<i4", "int32": "<i4",
              "uint": "<u4", "uint32": "<u4"}


def construct_list_of_attributes(n_offset_floats, feat_dim, n_scale=6, n_rot=4):  # scene/gaussian_model.py:502-514
    names = ["x", "y", "z", "nx", "ny", "nz"]
    names += [f"f_offset_{i}" for i in range(n_offset_floats)]
    names += [f"f_anchor_feat_{i}" for i in range(feat_dim)]
    names += ["opacity", "uncertainty"]
    names += [f"scale_{i}" for i in range(n_scale)]
    names += [f"rot_{i}" for i in range(n_rot)]
    return names


def _np(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def save_ply(path, anchor, offset, anchor_feat, opacity, uncertainty, scaling, rotation):
    """anchor[N,3], offset[N,K,3], anchor_feat[N,F], opacity[N,1], uncertainty[N,1], scaling[N,6], rotation[N,4]."""
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    anchor = _np(anchor).astype(np.float32)
    off = _np(offset).astype(np.float32)
    off_flat = np.ascontiguousarray(np.transpose(off, (0, 2, 1))).reshape(off.shape[0], -1)  # :629 transpose(1,2).flatten(1)
    cols = [anchor, np.zeros_like(anchor), off_flat, _np(anchor_feat).astype(np.float32), _np(opacity).astype(np.float32).reshape(-1, 1),
            _np(uncertainty).astype(np.float32).reshape(-1, 1), _np(scaling).astype(np.float32), _np(rotation).astype(np.float32)]
    attributes = np.ascontiguousarray(np.concatenate(cols, axis=1), dtype="<f4")               # :639
    names = construct_list_of_attributes(off_flat.shape[1], cols[3].shape[1], cols[6].shape[1], cols[7].shape[1])
    assert attributes.shape[1] == len(names)
    header = ["ply", "format binary_little_endian 1.0", f"element vertex {attributes.shape[0]}"]
    header += [f"property float {n}" for n in names] + ["end_header"]
    with open(path, "wb") as f:
        f.write(("\n".join(header) + "\n").encode("ascii"))
        f.write(attributes.tobytes())


def _read_vertex_table(path):
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, count, props, in_vertex = None, None, [], False
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: truncated PLY header")
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] in ("comment", "obj_info"):
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                in_vertex = tok[1] == "vertex"
                if in_vertex:
                    count = int(tok[2])
                elif count is None:
                    raise ValueError(f"{path}: an element precedes 'vertex'; not a GScream anchor file")
            elif tok[0] == "property" and in_vertex:
                if tok[1] == "list":
                    raise ValueError(f"{path}: list properties in the vertex element are not supported")
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if fmt != "binary_little_endian" or count is None:
            raise ValueError(f"{path}: expected a binary_little_endian PLY with a vertex element (got format {fmt})")
        table = np.frombuffer(f.read(count * np.dtype(props).itemsize), dtype=np.dtype(props), count=count)
    return table


def load_ply_sparse_gaussian(path, device="cpu"):
    """-> dict of float32 tensors {anchor, offset[N,K,3], anchor_feat, opacity[N,1], uncertainty[N,1], scaling, rotation}
    (what scene/gaussian_model.py:644-686 assigns to the model's parameters; columns are collected by name prefix and
    sorted by their numeric suffix, like the reference)."""
    v = _read_vertex_table(path)
    names = v.dtype.names

    def cols(prefix):
        picked = sorted([n for n in names if n.startswith(prefix)], key=lambda n: int(n.split("_")[-1]))
        return np.stack([np.asarray(v[n], dtype=np.float32) for n in picked], axis=1) if picked else np.zeros((len(v), 0), np.float32)

    anchor = np.stack([v["x"], v["y"], v["z"]], axis=1).astype(np.float32)
    offsets = cols("f_offset")
    offsets = offsets.reshape(offsets.shape[0], 3, -1)                       # :678
    out = {
        "anchor": anchor,
        "offset": np.ascontiguousarray(np.transpose(offsets, (0, 2, 1))),     # :682 transpose(1, 2)
        "anchor_feat": cols("f_anchor_feat"),
        "opacity": np.asarray(v["opacity"], dtype=np.float32)[:, None],
        "uncertainty": np.asarray(v["uncertainty"], dtype=np.float32)[:, None],
        "scaling": cols("scale_"),
        "rotation": cols("rot"),
    }
    return {k: torch.tensor(a, dtype=torch.float32, device=device) for k, a in out.items()}


_MLP_KEYS = (("opacity_mlp", "mlp_opacity"), ("uncertainty_mlp", "mlp_uncertainty"), ("cov_mlp", "mlp_cov"), ("color_mlp", "mlp_color"))


def save_mlp_checkpoints(model, path):  # scene/gaussian_model.py:975-992
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    ckpt = {k: getattr(model, attr).state_dict() for k, attr in _MLP_KEYS}
    if getattr(model, "use_feat_bank", False):
        ckpt["mlp_feature_bank"] = model.mlp_feature_bank.state_dict()
    torch.save(ckpt, path)


def load_mlp_checkpoints(model, path, map_location=None):  # :995-1002
    ckpt = torch.load(path, map_location=map_location)
    for k, attr in _MLP_KEYS:
        getattr(model, attr).load_state_dict(ckpt[k])
    if getattr(model, "use_feat_bank", False):
        model.mlp_feature_bank.load_state_dict(ckpt["mlp_feature_bank"])
