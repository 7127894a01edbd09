"""numpy/ctypes front-end of the CPU oracle (oracle/gs_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of gs_oracle.c.  Only tests/, the smoke check in
__graft_entry__.py and bench.py's cpu_baseline leg import this module.  Parity status of the
oracle itself: *parity unpinned* (the reference ships no golden vectors and cannot be built
here), except the camera convention and the SH colours, which tests/test_reference_vectors.py
pins with vectors the reference's own Python helpers computed; see DESIGN.md section "Oracle".

The call structure mirrors the reference's C++ entry points
(DGR/rasterize_points.cu:35-122 forward, :124-211 backward, :213-373 filters), with numpy
arrays instead of torch tensors.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_f32p = ctypes.POINTER(ctypes.c_float)
_f64p = ctypes.POINTER(ctypes.c_double)
_i32p = ctypes.POINTER(ctypes.c_int32)
_u32p = ctypes.POINTER(ctypes.c_uint32)
_u64p = ctypes.POINTER(ctypes.c_uint64)
_u8p = ctypes.POINTER(ctypes.c_uint8)


def build(force=False):
    """Compile libgsoracle.so with gcc (recipe: oracle/Makefile)."""
    so = os.path.join(_HERE, "libgsoracle.so")
    src = os.path.join(_HERE, "gs_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libgsoracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        _LIB.gso_bin.restype = ctypes.c_int
        _LIB.gso_sort_bits.restype = ctypes.c_int
        _LIB.gso_max_threads.restype = ctypes.c_int
    return _LIB


def max_threads():
    return int(lib().gso_max_threads())


def _f(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def _p(a, t):
    return None if a is None else a.ctypes.data_as(t)


def _prep(mode, means3D, scales, rotations, opacities, uncertainties, shs, sh_degree, cov3D_precomp,
          colors_precomp, viewmatrix, projmatrix, campos, W, H, tanfovx, tanfovy, scale_modifier):
    L = lib()
    means3D = _f(means3D)
    P = means3D.shape[0]
    scales, rotations, opacities, uncertainties = _f(scales), _f(rotations), _f(opacities), _f(uncertainties)
    shs, cov3D_precomp, colors_precomp = _f(shs), _f(cov3D_precomp), _f(colors_precomp)
    viewmatrix, projmatrix, campos = _f(viewmatrix), _f(projmatrix), _f(campos)
    M = 0 if shs is None else shs.shape[1]
    st = dict(
        radii=np.zeros(P, np.int32),
        means2D=np.zeros((P, 2), np.float32) if mode != 2 else np.zeros((2, P), np.float32),
        depths=np.zeros(P, np.float32),
        cov3D=np.zeros((P, 6), np.float32),
        conic_opacity=np.zeros((P, 4), np.float32),
        unc=np.zeros(P, np.float32),
        rgb=np.zeros((P, 3), np.float32),
        clamped=np.zeros((P, 3), np.uint8),
        tiles_touched=np.zeros(P, np.uint32),
    )
    L.gso_preprocess(
        ctypes.c_int(mode), ctypes.c_int(P), ctypes.c_int(sh_degree), ctypes.c_int(M),
        _p(means3D, _f32p), _p(scales, _f32p), ctypes.c_float(scale_modifier), _p(rotations, _f32p),
        _p(opacities, _f32p), _p(uncertainties, _f32p), _p(shs, _f32p), _p(cov3D_precomp, _f32p),
        _p(colors_precomp, _f32p), _p(viewmatrix, _f32p), _p(projmatrix, _f32p), _p(campos, _f32p),
        ctypes.c_int(W), ctypes.c_int(H), ctypes.c_float(tanfovx), ctypes.c_float(tanfovy),
        _p(st["radii"], _i32p), _p(st["means2D"], _f32p), _p(st["depths"], _f32p), _p(st["cov3D"], _f32p),
        _p(st["conic_opacity"], _f32p), _p(st["unc"], _f32p), _p(st["rgb"], _f32p), _p(st["clamped"], _u8p),
        _p(st["tiles_touched"], _u32p))
    return st


def forward(means3D, scales, rotations, opacities, uncertainties, *, W, H, tanfovx, tanfovy, viewmatrix,
            projmatrix, campos=None, bg=(0, 0, 0), scale_modifier=1.0, colors_precomp=None, shs=None,
            sh_degree=0, cov3D_precomp=None, nthreads=1):
    """Full forward (DGR/cuda_rasterizer/rasterizer_impl.cu:199-347).  Returns a dict with the
    three images, radii, num_rendered and every intermediate the backward needs."""
    L = lib()
    campos = np.zeros(3, np.float32) if campos is None else campos
    st = _prep(0, means3D, scales, rotations, opacities, uncertainties, shs, sh_degree, cov3D_precomp,
               colors_precomp, viewmatrix, projmatrix, campos, W, H, tanfovx, tanfovy, scale_modifier)
    P = st["radii"].shape[0]
    gx, gy = (W + 15) // 16, (H + 15) // 16
    R = int(st["tiles_touched"].astype(np.int64).sum())
    st["point_list"] = np.zeros(max(R, 1), np.uint32)
    st["point_keys"] = np.zeros(max(R, 1), np.uint64)
    st["ranges"] = np.zeros((gx * gy, 2), np.uint32)
    if P > 0:
        R2 = L.gso_bin(ctypes.c_int(P), _p(st["radii"], _i32p), _p(st["means2D"], _f32p), _p(st["depths"], _f32p),
                       _p(st["tiles_touched"], _u32p), ctypes.c_int(W), ctypes.c_int(H),
                       _p(st["point_list"], _u32p), _p(st["point_keys"], _u64p), _p(st["ranges"], _u32p))
        assert R2 == R
    st["point_list"] = st["point_list"][:R]
    st["point_keys"] = st["point_keys"][:R]
    colors = _f(colors_precomp) if colors_precomp is not None else st["rgb"]
    if cov3D_precomp is not None:
        st["cov3D"] = _f(cov3D_precomp)
    bg = _f(np.asarray(bg, np.float32))
    st.update(out_color=np.zeros((3, H, W), np.float32), out_depth=np.zeros((1, H, W), np.float32),
              out_unc=np.zeros((1, H, W), np.float32), final_T=np.zeros((H, W), np.float32),
              n_contrib=np.zeros((H, W), np.uint32), num_rendered=R, colors=colors, bg=bg, W=W, H=H)
    if P > 0:
        L.gso_render_forward(ctypes.c_int(W), ctypes.c_int(H), _p(st["ranges"], _u32p), _p(st["point_list"], _u32p),
                             _p(st["means2D"], _f32p), _p(colors, _f32p), _p(st["depths"], _f32p), _p(st["unc"], _f32p),
                             _p(st["conic_opacity"], _f32p), _p(bg, _f32p), _p(st["out_color"], _f32p),
                             _p(st["out_depth"], _f32p), _p(st["out_unc"], _f32p), _p(st["final_T"], _f32p),
                             _p(st["n_contrib"], _u32p), ctypes.c_int(nthreads))
    else:
        st["final_T"][:] = 0  # reference leaves the outputs at their zero fill when P == 0 (rasterize_points.cu:85)
    return st


def margins(st, nthreads=1):
    """Per pixel, how close the forward walk came to flipping a decision (gso_render_margins): dict of m_alpha, g_alpha,
    m_T, g_T, m_pow, each [H, W].  Diagnostic: classifies a build's outlier pixels into alpha = 1/255 and T = 1e-4 flips."""
    L = lib()
    W, H = st["W"], st["H"]
    out = dict(m_alpha=np.zeros((H, W), np.float32), g_alpha=np.zeros((H, W), np.uint32), m_T=np.zeros((H, W), np.float32),
               g_T=np.zeros((H, W), np.uint32), m_pow=np.zeros((H, W), np.float32))
    L.gso_render_margins(ctypes.c_int(W), ctypes.c_int(H), _p(st["ranges"], _u32p), _p(st["point_list"], _u32p),
                         _p(st["means2D"], _f32p), _p(st["conic_opacity"], _f32p), _p(out["m_alpha"], _f32p),
                         _p(out["g_alpha"], _u32p), _p(out["m_T"], _f32p), _p(out["g_T"], _u32p), _p(out["m_pow"], _f32p),
                         ctypes.c_int(nthreads))
    return out


def backward(st, means3D, scales, rotations, dL_dcolor, dL_ddepth, dL_dunc, *, tanfovx, tanfovy, viewmatrix,
             projmatrix, campos=None, scale_modifier=1.0, shs=None, sh_degree=0, nthreads=1):
    """Full backward (rasterizer_impl.cu:536-643) given the forward state `st`.  Returns the nine
    gradients of rasterize_points.cu:210 plus the two internal ones (dL_ddepths, dL_dconic)."""
    L = lib()
    W, H = st["W"], st["H"]
    means3D, scales, rotations, shs = _f(means3D), _f(scales), _f(rotations), _f(shs)
    viewmatrix, projmatrix = _f(viewmatrix), _f(projmatrix)
    campos = _f(np.zeros(3, np.float32) if campos is None else campos)
    P = means3D.shape[0]
    M = 0 if shs is None else shs.shape[1]
    dL_dcolor = _f(dL_dcolor).reshape(3, H, W)
    dL_ddepth = _f(dL_ddepth).reshape(H, W)
    dL_dunc = _f(dL_dunc).reshape(H, W)
    g = dict(mean2D=np.zeros((P, 2), np.float64), conic=np.zeros((P, 3), np.float64), opacity=np.zeros(P, np.float64),
             colors=np.zeros((P, 3), np.float64), depth=np.zeros(P, np.float64), unc=np.zeros(P, np.float64))
    if P > 0:
        L.gso_render_backward(ctypes.c_int(W), ctypes.c_int(H), _p(st["ranges"], _u32p), _p(st["point_list"], _u32p),
                              _p(st["bg"], _f32p), _p(st["means2D"], _f32p), _p(st["conic_opacity"], _f32p),
                              _p(st["colors"], _f32p), _p(st["depths"], _f32p), _p(st["unc"], _f32p),
                              _p(st["final_T"], _f32p), _p(st["n_contrib"], _u32p), _p(dL_dcolor, _f32p),
                              _p(dL_ddepth, _f32p), _p(dL_dunc, _f32p), _p(g["mean2D"], _f64p), _p(g["conic"], _f64p),
                              _p(g["opacity"], _f64p), _p(g["colors"], _f64p), _p(g["depth"], _f64p), _p(g["unc"], _f64p),
                              ctypes.c_int(nthreads))
    g32 = {k: v.astype(np.float32) for k, v in g.items()}
    out = dict(dL_dmeans3D=np.zeros((P, 3), np.float32), dL_dcov3D=np.zeros((P, 6), np.float32),
               dL_dsh=np.zeros((P, M, 3), np.float32), dL_dscales=np.zeros((P, 3), np.float32),
               dL_drotations=np.zeros((P, 4), np.float32))
    if P > 0:
        L.gso_preprocess_backward(
            ctypes.c_int(P), ctypes.c_int(sh_degree), ctypes.c_int(M), _p(means3D, _f32p), _p(st["radii"], _i32p),
            _p(shs, _f32p), _p(st["clamped"], _u8p), _p(scales, _f32p), _p(rotations, _f32p),
            ctypes.c_float(scale_modifier), _p(_f(st["cov3D"]), _f32p), _p(viewmatrix, _f32p), _p(projmatrix, _f32p),
            ctypes.c_int(W), ctypes.c_int(H), ctypes.c_float(tanfovx), ctypes.c_float(tanfovy), _p(campos, _f32p),
            _p(g32["mean2D"], _f32p), _p(g32["conic"], _f32p), _p(g32["colors"], _f32p), _p(g32["depth"], _f32p),
            _p(out["dL_dmeans3D"], _f32p), _p(out["dL_dcov3D"], _f32p), _p(out["dL_dsh"], _f32p),
            _p(out["dL_dscales"], _f32p), _p(out["dL_drotations"], _f32p))
    m2 = np.zeros((P, 3), np.float32)
    m2[:, :2] = g32["mean2D"]
    out.update(dL_dmeans2D=m2, dL_dcolors=g32["colors"], dL_dopacity=g32["opacity"].reshape(P, 1),
               dL_duncertainty=g32["unc"].reshape(P, 1), dL_ddepths=g32["depth"], dL_dconic=g32["conic"])
    return out


def backward_envelope(st, gids, means3D, scales, rotations, dL_dcolor, dL_ddepth, dL_dunc, *, tanfovx, tanfovy, viewmatrix,
                      projmatrix, campos=None, scale_modifier=1.0, shs=None, sh_degree=0, K=64, seed=1):
    """Order-noise envelope of the reference algorithm for the Gaussians `gids` (gso_backward_envelope): the 11 per-contribution
    sums of backward.cu:554-601 accumulated in fp32 in K random contribution orders (what unordered atomicAdd yields), each pushed
    through the per-Gaussian backward (backward.cu:144-406) like the real thing.  Returns {family: [len(gids), K, ...]} for the
    same gradient families as backward(); compare with backward()'s double-accumulated rows."""
    L = lib()
    L.gso_backward_envelope.restype = ctypes.c_int
    W, H = st["W"], st["H"]
    gids = np.ascontiguousarray(gids, dtype=np.int32)
    ng = int(gids.shape[0])
    means3D, scales, rotations, shs = _f(means3D), _f(scales), _f(rotations), _f(shs)
    viewmatrix, projmatrix = _f(viewmatrix), _f(projmatrix)
    campos = _f(np.zeros(3, np.float32) if campos is None else campos)
    M = 0 if shs is None else shs.shape[1]
    dL_dcolor = _f(dL_dcolor).reshape(3, H, W)
    dL_ddepth = _f(dL_ddepth).reshape(H, W)
    dL_dunc = _f(dL_dunc).reshape(H, W)
    sums = np.zeros((max(ng, 1), K, 11), np.float32)
    fam = dict(dL_dmeans2D=np.zeros((ng, K, 3), np.float32), dL_dcolors=np.zeros((ng, K, 3), np.float32),
               dL_dopacity=np.zeros((ng, K, 1), np.float32), dL_duncertainty=np.zeros((ng, K, 1), np.float32),
               dL_dmeans3D=np.zeros((ng, K, 3), np.float32), dL_dcov3D=np.zeros((ng, K, 6), np.float32),
               dL_dscales=np.zeros((ng, K, 3), np.float32), dL_drotations=np.zeros((ng, K, 4), np.float32),
               dL_dsh=np.zeros((ng, K, M, 3), np.float32))
    if ng == 0:
        return fam
    most = L.gso_backward_envelope(
        ctypes.c_int(W), ctypes.c_int(H), _p(st["ranges"], _u32p), _p(st["point_list"], _u32p), _p(st["bg"], _f32p),
        _p(st["means2D"], _f32p), _p(st["radii"], _i32p), _p(st["conic_opacity"], _f32p), _p(st["colors"], _f32p),
        _p(st["depths"], _f32p), _p(st["unc"], _f32p), _p(st["final_T"], _f32p), _p(st["n_contrib"], _u32p),
        _p(dL_dcolor, _f32p), _p(dL_ddepth, _f32p), _p(dL_dunc, _f32p), ctypes.c_int(ng), _p(gids, _i32p), ctypes.c_int(K),
        ctypes.c_uint64(seed), _p(sums, _f32p))
    fam["contributions_max"] = int(most)
    # the per-Gaussian backward on the ng rows, once per order (every per-Gaussian array gathered to ng rows)
    gat = lambda a: None if a is None else np.ascontiguousarray(a[gids])
    m3, sc, ro, sh_g = gat(means3D), gat(scales), gat(rotations), gat(shs)
    radii, clamped, cov3D = gat(st["radii"]), gat(st["clamped"]), gat(_f(st["cov3D"]))
    for k in range(K):
        g_mean2D = np.ascontiguousarray(sums[:ng, k, 0:2]); g_conic = np.ascontiguousarray(sums[:ng, k, 2:5])
        g_colors = np.ascontiguousarray(sums[:ng, k, 6:9]); g_depth = np.ascontiguousarray(sums[:ng, k, 9])
        o = dict(m3=np.zeros((ng, 3), np.float32), cov=np.zeros((ng, 6), np.float32), sh=np.zeros((ng, M, 3), np.float32),
                 sc=np.zeros((ng, 3), np.float32), ro=np.zeros((ng, 4), np.float32))
        L.gso_preprocess_backward(
            ctypes.c_int(ng), ctypes.c_int(sh_degree), ctypes.c_int(M), _p(m3, _f32p), _p(radii, _i32p), _p(sh_g, _f32p),
            _p(clamped, _u8p), _p(sc, _f32p), _p(ro, _f32p), ctypes.c_float(scale_modifier), _p(cov3D, _f32p),
            _p(viewmatrix, _f32p), _p(projmatrix, _f32p), ctypes.c_int(W), ctypes.c_int(H), ctypes.c_float(tanfovx),
            ctypes.c_float(tanfovy), _p(campos, _f32p), _p(g_mean2D, _f32p), _p(g_conic, _f32p), _p(g_colors, _f32p),
            _p(g_depth, _f32p), _p(o["m3"], _f32p), _p(o["cov"], _f32p), _p(o["sh"], _f32p), _p(o["sc"], _f32p), _p(o["ro"], _f32p))
        fam["dL_dmeans2D"][:, k, :2] = sums[:ng, k, 0:2]
        fam["dL_dcolors"][:, k] = sums[:ng, k, 6:9]
        fam["dL_dopacity"][:, k, 0] = sums[:ng, k, 5]
        fam["dL_duncertainty"][:, k, 0] = sums[:ng, k, 10]
        fam["dL_dmeans3D"][:, k] = o["m3"]; fam["dL_dcov3D"][:, k] = o["cov"]; fam["dL_dscales"][:, k] = o["sc"]
        fam["dL_drotations"][:, k] = o["ro"]; fam["dL_dsh"][:, k] = o["sh"]
    return fam


def visible_filter(means3D, scales, rotations, *, W, H, tanfovx, tanfovy, viewmatrix, projmatrix, scale_modifier=1.0,
                   cov3D_precomp=None):
    """rasterizer_impl.cu:350-406 visible_filter -> radii."""
    st = _prep(1, means3D, scales, rotations, None, None, None, 0, cov3D_precomp, np.zeros((1, 3), np.float32),
               viewmatrix, projmatrix, np.zeros(3, np.float32), W, H, tanfovx, tanfovy, scale_modifier)
    return st["radii"]


def position2D_filter(means3D, scales, rotations, *, W, H, tanfovx, tanfovy, viewmatrix, projmatrix, scale_modifier=1.0,
                      cov3D_precomp=None):
    """rasterizer_impl.cu:470-530 position2D_filter -> (radii, x, y)."""
    st = _prep(2, means3D, scales, rotations, None, None, None, 0, cov3D_precomp, np.zeros((1, 3), np.float32),
               viewmatrix, projmatrix, np.zeros(3, np.float32), W, H, tanfovx, tanfovy, scale_modifier)
    return st["radii"], st["means2D"][0].copy(), st["means2D"][1].copy()


def mark_visible(means3D, viewmatrix):
    """rasterizer_impl.cu:141-153 markVisible -> bool[P]."""
    means3D, viewmatrix = _f(means3D), _f(viewmatrix)
    P = means3D.shape[0]
    out = np.zeros(P, np.uint8)
    lib().gso_mark_visible(ctypes.c_int(P), _p(means3D, _f32p), _p(viewmatrix, _f32p), _p(out, _u8p))
    return out.astype(bool)


def sort_bits(W, H):
    return int(lib().gso_sort_bits(ctypes.c_int(W), ctypes.c_int(H)))
