"""CPU tests of gscream_amd.checkpoint_io (GScream's anchor PLY + MLP checkpoint formats)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gscream_amd import checkpoint_io as IO  # noqa: E402
from oracle import decode_oracle as DO  # noqa: E402


def test_ply_layout_and_round_trip(tmp_path):
    N, K, F = 57, 10, 32
    m = DO.Model(N, K, F, seed=3, dtype=torch.float32)
    opacity, unc = torch.rand(N, 1), torch.rand(N, 1)
    rotation = torch.randn(N, 4)
    path = str(tmp_path / "point_cloud" / "iteration_1" / "point_cloud.ply")
    IO.save_ply(path, m._anchor, m._offset, m._anchor_feat, opacity, unc, m._scaling, rotation)
    raw = open(path, "rb").read()
    header, body = raw.split(b"end_header\n", 1)
    lines = header.decode().split("\n")
    assert lines[:3] == ["ply", "format binary_little_endian 1.0", f"element vertex {N}"]
    names = [ln.split()[2] for ln in lines if ln.startswith("property")]
    assert all(ln.split()[1] == "float" for ln in lines if ln.startswith("property"))
    assert names == IO.construct_list_of_attributes(3 * K, F) and len(names) == 6 + 30 + 32 + 2 + 6 + 4
    table = np.frombuffer(body, dtype="<f4").reshape(N, len(names))
    assert np.array_equal(table[:, 0:3], m._anchor.detach().numpy()) and not table[:, 3:6].any()
    # f_offset_i is component-major: column c*K + k holds offset[n, k, c]   (gaussian_model.py:629)
    off = m._offset.detach().numpy()
    assert np.array_equal(table[:, 6 + 1 * K + 4], off[:, 4, 1]) and np.array_equal(table[:, 6 + 2 * K + 9], off[:, 9, 2])
    assert np.array_equal(table[:, 6 + 30:6 + 30 + F], m._anchor_feat.detach().numpy())
    assert np.array_equal(table[:, 68], opacity[:, 0].numpy()) and np.array_equal(table[:, 69], unc[:, 0].numpy())
    back = IO.load_ply_sparse_gaussian(path)
    for key, ref in (("anchor", m._anchor), ("offset", m._offset), ("anchor_feat", m._anchor_feat), ("opacity", opacity),
                     ("uncertainty", unc), ("scaling", m._scaling), ("rotation", rotation)):
        assert torch.equal(back[key], ref.detach()), key
    assert back["offset"].shape == (N, K, 3) and back["offset"].is_contiguous()


def test_ply_reader_accepts_reordered_and_commented_headers(tmp_path):
    """Columns are found by name, in numeric-suffix order, whatever their position (as the reference's loader does)."""
    N = 5
    names = ["rot_1", "x", "comment_dummy", "y", "z", "opacity", "uncertainty", "rot_0", "scale_0", "f_offset_2", "f_offset_0",
             "f_offset_1", "f_anchor_feat_0", "rot_3", "rot_2"]
    data = np.arange(N * len(names), dtype="<f4").reshape(N, len(names))
    path = str(tmp_path / "odd.ply")
    with open(path, "wb") as f:
        f.write(("ply\nformat binary_little_endian 1.0\ncomment made by a test\nelement vertex %d\n" % N).encode())
        f.write("".join(f"property float {n}\n" for n in names).encode())
        f.write(b"end_header\n")
        f.write(data.tobytes())
    back = IO.load_ply_sparse_gaussian(path)
    col = lambda n: torch.from_numpy(data[:, names.index(n)].copy())
    assert torch.equal(back["rotation"], torch.stack([col("rot_0"), col("rot_1"), col("rot_2"), col("rot_3")], 1))
    assert torch.equal(back["offset"][:, 0, :], torch.stack([col("f_offset_0"), col("f_offset_1"), col("f_offset_2")], 1))
    assert torch.equal(back["anchor"], torch.stack([col("x"), col("y"), col("z")], 1))


def test_mlp_checkpoint_round_trip(tmp_path):
    a, b = DO.Model(4, seed=1, dtype=torch.float32), DO.Model(4, seed=2, dtype=torch.float32)
    path = str(tmp_path / "ckpt" / "checkpoint.pth")
    IO.save_mlp_checkpoints(a, path)
    ckpt = torch.load(path)
    assert set(ckpt) == {"opacity_mlp", "uncertainty_mlp", "cov_mlp", "color_mlp"}     # gaussian_model.py:985-990
    assert set(ckpt["cov_mlp"]) == {"0.weight", "0.bias", "2.weight", "2.bias"}
    IO.load_mlp_checkpoints(b, path)
    for (ka, pa), (kb, pb) in zip(a.state_dict().items(), b.state_dict().items()):
        if ka.startswith("mlp_"):
            assert ka == kb and torch.equal(pa, pb), ka


def test_reader_parses_a_file_laid_out_as_plyfile_documents_it(tmp_path):
    """`plyfile` is absent here, so the PLY layout is UNPINNED against the reference's own writer (checkpoint_io.py says so); this is
    the next best thing: a file built byte by byte in the test -- the header plyfile documents for `PlyElement.describe(structured
    array of 'f4' fields, 'vertex')` + `PlyData([el]).write()` on a little-endian host, attribute order of
    scene/gaussian_model.py:502-514 / :623-642, with the `comment` and `obj_info` lines plyfile carries through -- is read back
    column by column, and `save_ply` produces the same header bytes (minus the optional lines) and the same record bytes."""
    from gscream_amd import checkpoint_io as IO
    N, K, F = 7, 10, 32
    rng = np.random.default_rng(5)
    names = (["x", "y", "z", "nx", "ny", "nz"] + [f"f_offset_{i}" for i in range(3 * K)] + [f"f_anchor_feat_{i}" for i in range(F)] +
             ["opacity", "uncertainty"] + [f"scale_{i}" for i in range(6)] + [f"rot_{i}" for i in range(4)])
    table = rng.standard_normal((N, len(names))).astype("<f4")
    table[:, 3:6] = 0.0                                              # normals are zeros (:628)
    plain = "ply\nformat binary_little_endian 1.0\n" + f"element vertex {N}\n" + "".join(f"property float {n}\n" for n in names) + "end_header\n"
    annotated = plain.replace("element vertex", "comment written by a test, not by plyfile\nobj_info none\nelement vertex")
    path = tmp_path / "point_cloud.ply"
    path.write_bytes(annotated.encode("ascii") + table.tobytes())
    got = IO.load_ply_sparse_gaussian(str(path))
    col = {n: table[:, i] for i, n in enumerate(names)}
    assert np.array_equal(got["anchor"].numpy(), table[:, 0:3])
    assert np.array_equal(got["opacity"].numpy()[:, 0], col["opacity"]) and np.array_equal(got["uncertainty"].numpy()[:, 0], col["uncertainty"])
    assert np.array_equal(got["scaling"].numpy(), np.stack([col[f"scale_{i}"] for i in range(6)], 1))
    assert np.array_equal(got["rotation"].numpy(), np.stack([col[f"rot_{i}"] for i in range(4)], 1))
    assert np.array_equal(got["anchor_feat"].numpy(), np.stack([col[f"f_anchor_feat_{i}"] for i in range(F)], 1))
    # f_offset_{c*K + k} = _offset[n, k, c]  (:629 `_offset.transpose(1, 2).flatten(start_dim=1)`, undone at :678-682)
    off = got["offset"].numpy()
    assert off.shape == (N, K, 3)
    for c in range(3):
        for k in range(K):
            assert np.array_equal(off[:, k, c], col[f"f_offset_{c * K + k}"])
    # and the writer: same header text, same record bytes
    out = tmp_path / "written.ply"
    IO.save_ply(str(out), got["anchor"], got["offset"], got["anchor_feat"], got["opacity"], got["uncertainty"], got["scaling"], got["rotation"])
    assert out.read_bytes() == plain.encode("ascii") + table.tobytes()
