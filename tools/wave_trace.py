#!/usr/bin/env python
"""Diagnostic: per-wavefront schedule of the backward blend (needs the `make trace` build):
   GSR_LIB=$PWD/gscream_amd/libgsraster_trace.so [GSR_BWD1=1] python tools/wave_trace.py [config2]
Prints the launch span, the distribution of wave lifetimes and start times, and how the waves spread over
XCDs / CUs / SIMDs."""
import collections
import ctypes
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import bench
from gscream_amd import GaussianRasterizationSettings, GaussianRasterizer, _native
from gscream_amd import synthetic as S

wl = sys.argv[1] if len(sys.argv) > 1 else "config2"
P, W, H, seed, gsel, desc = bench.WORKLOADS[wl]
dev = torch.device("cuda", 0)
lib = _native.load()
s = S.scene_slab(seed, P, W, H)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
leaves = [t(s[k]).requires_grad_(True) for k in ("means3D", "opacities", "uncertainties", "colors", "scales", "rotations")]
means3D, opac, unc, colors, scales, rots = leaves
means2D = torch.zeros_like(means3D, requires_grad=True)
rs = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=s["tanfovx"], tanfovy=s["tanfovy"], bg=t(s["bg"]),
                                   scale_modifier=1.0, viewmatrix=t(s["viewmatrix"]), projmatrix=t(s["projmatrix"]),
                                   sh_degree=1, campos=t(s["campos"]), prefiltered=False, debug=False)
rast = GaussianRasterizer(raster_settings=rs)
gc, gd, gu = (t(g) for g in S.upstream_grads(seed, W, H, *gsel))
for _ in range(4):
    color, depth, feat, radii = rast(means3D, means2D, opac, unc, colors_precomp=colors, scales=scales, rotations=rots)
    outs = [o for o, use in zip((color, depth, feat), gsel) if use]
    gos = [g for g, use in zip((gc, gd, gu), gsel) if use]
    torch.autograd.grad(outs, leaves + [means2D], gos)
torch.cuda.synchronize()
T = ((W + 15) // 16) * ((H + 15) // 16)
buf = np.zeros(2 * 4 * 4 * 36864, dtype=np.uint64)
lib.gsr_debug_trace.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
rc = lib.gsr_debug_trace(buf.ctypes.data, buf.nbytes)
assert rc == 0, rc
which = sys.argv[2] if len(sys.argv) > 2 else "bwd"
half = buf[: 4 * 4 * 36864] if which == "bwd" else buf[4 * 4 * 36864:]
print("kernel:", which)
rows = half.reshape(-1, 4)
rows = rows[rows[:, 1] > 0]
st, en, hw, xcc = (rows[:, i].astype(np.int64) for i in range(4))
t0 = st.min()
st, en = (st - t0) / 100.0, (en - t0) / 100.0  # microseconds
life = en - st
q = [0, 5, 25, 50, 75, 95, 100]
print("waves", len(rows), "tiles", T, "span us", en.max())
print("start us  pct", q, np.round(np.percentile(st, q), 1))
print("end us    pct", q, np.round(np.percentile(en, q), 1))
print("life us   pct", q, np.round(np.percentile(life, q), 1), "mean", round(life.mean(), 1))
simd = (hw >> 4) & 3
cu = (hw >> 8) & 15
sh = (hw >> 12) & 1
se = (hw >> 13) & 7
x = xcc & 15
slot = x * 10000 + se * 1000 + sh * 100 + cu
print("distinct XCC", len(set(x.tolist())), "distinct (xcc,se,sh,cu)", len(set(slot.tolist())))
print("waves per SIMD id", sorted(collections.Counter(simd.tolist()).items()))
per_cu = collections.Counter(slot.tolist())
print("waves per CU: pct", q, np.percentile(list(per_cu.values()), q))
per_simd = collections.Counter((slot * 4 + simd).tolist())
print("waves per (CU,SIMD): pct", q, np.percentile(list(per_simd.values()), q), "slots used", len(per_simd))
for tt in np.linspace(0, en.max(), 9)[:-1]:
    print("  t=%6.1f us  resident waves %5d" % (tt, int(((st <= tt) & (en > tt)).sum())))
