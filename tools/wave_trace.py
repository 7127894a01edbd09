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
s = bench.scene_for(wl, seed, P, W, H)
P = s["means3D"].shape[0]
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
live = rows[:, 1] > 0
# (the buffer keeps the rows of earlier, larger launches -- another grid size, the fix-up pass: only the last launch's waves count)
live &= rows[:, 0].astype(np.int64) >= rows[live, 0].astype(np.int64).max() - 100 * 2000   # started within 2 ms of the last wave to start
rows = rows[live]
st, en, hw, xcc = (rows[:, i].astype(np.int64) for i in range(4))
t0 = st.min()
st, en = (st - t0) / 100.0, (en - t0) / 100.0  # microseconds
life = en - st
q = [0, 5, 25, 50, 75, 95, 100]
print("waves", len(rows), "tiles", T, "span us", en.max(), " sum of wave lifetimes / (span x 1024 SIMDs): %.2f waves per SIMD on average" % (life.sum() / (en.max() * 1024)))
print("start us  pct", q, np.round(np.percentile(st, q), 1))
print("end us    pct", q, np.round(np.percentile(en, q), 1))
print("life us   pct", q, np.round(np.percentile(life, q), 1), "mean", round(life.mean(), 1))
simd = (hw >> 4) & 3
cu = (hw >> 8) & 15
sh = (hw >> 12) & 1
se = (hw >> 13) & 7
x = xcc & 15
tr_it, tr_bl, tr_tile, tr_seg = (xcc >> 4) & 0xffff, (xcc >> 20) & 0xffff, (xcc >> 36) & 0xffff, (xcc >> 52) & 0x3f  # (backward, trace build: iterations evaluated / blending, tile, segment)
slot = x * 10000 + se * 1000 + sh * 100 + cu
print("distinct XCC", len(set(x.tolist())), "distinct (xcc,se,sh,cu)", len(set(slot.tolist())))
print("waves per SIMD id", sorted(collections.Counter(simd.tolist()).items()))
per_cu = collections.Counter(slot.tolist())
print("waves per CU: pct", q, np.percentile(list(per_cu.values()), q))
per_simd = collections.Counter((slot * 4 + simd).tolist())
print("waves per (CU,SIMD): pct", q, np.percentile(list(per_simd.values()), q), "slots used", len(per_simd))
for tt in np.linspace(0, en.max(), 9)[:-1]:
    print("  t=%6.1f us  resident waves %5d" % (tt, int(((st <= tt) & (en > tt)).sum())))
# finer timeline + when each SIMD runs dry (the launch lasts until the last one does)
key = slot * 4 + simd
last_end = collections.defaultdict(float)
busy = collections.defaultdict(float)
for k, e, l in zip(key.tolist(), en.tolist(), life.tolist()):
    last_end[k] = max(last_end[k], e)
    busy[k] += l
le = np.array(list(last_end.values()))
print("per-SIMD time of its last wave's end: pct", q, np.round(np.percentile(le, q), 1))
for tt in np.linspace(0, en.max(), 21)[:-1]:
    res = (st <= tt) & (en > tt)
    act = len(set(key[res].tolist()))
    print("  t=%6.1f us  resident waves %5d  SIMDs with >=1 wave %4d  mean waves on those %.2f" % (tt, int(res.sum()), act, res.sum() / max(act, 1)))
if which == "bwd":
    # which tasks make the tail: every row is one wavefront of workgroup b = rows' index // 2 (slot b >> 3 of XCD b & 7)
    idx = np.nonzero(live)[0]
    blk = idx // 2
    from gscream_amd import _layout
    xt = _layout.xcd_tiles(T)
    rank = (blk >> 3) // xt
    order = np.argsort(-en)
    print("the 30 waves that end last: start, end, life, segment rank, XCD | iterations evaluated, blending, tile, segment, ns per evaluated iteration")
    for i in order[:30]:
        print("   %6.1f %6.1f %6.1f %3d %d | %5d %5d %5d %3d %7.0f" % (st[i], en[i], life[i], rank[i], blk[i] & 7, tr_it[i], tr_bl[i], tr_tile[i], tr_seg[i], 1e3 * life[i] / max(int(tr_it[i]), 1)))
    if tr_it.sum():
        print("all waves: iterations evaluated %d, blending %d; ns of wave life per evaluated iteration: pct(5,50,95) %s" % (int(tr_it.sum()), int(tr_bl.sum()), np.round(np.percentile(1e3 * life[tr_it > 0] / tr_it[tr_it > 0], [5, 50, 95]), 0)))
    for r in range(int(rank.max()) + 1):
        sel = rank == r
        if sel.sum():
            print("  rank %2d: waves %5d  start pct(5,50,95) %s  life pct(5,50,95) %s" % (r, int(sel.sum()), np.round(np.percentile(st[sel], [5, 50, 95]), 1), np.round(np.percentile(life[sel], [5, 50, 95]), 1)))
    for xcd in range(8):
        sel = (blk & 7) == xcd
        print("  XCD %d: waves %5d  sum of lifetimes %.0f us  last end %.1f" % (xcd, int(sel.sum()), life[sel].sum(), en[sel].max()))
if which == "fwd":
    nlist = (xcc >> 8) & 0xffffff
    deep = (rows[:, 3] >> np.uint64(32)).astype(np.int64)
    walked = (rows[:, 2] >> np.uint64(32)).astype(np.int64)
    print("list length pct", q, np.percentile(nlist, q), " deepest contributor pct", np.percentile(deep, q), " walked pct", np.percentile(walked, q))
    print("corr(life, walked) %.3f  corr(life, list length) %.3f  corr(walked, list length) %.3f" % (
        np.corrcoef(life, walked)[0, 1], np.corrcoef(life, nlist)[0, 1], np.corrcoef(walked, nlist)[0, 1]))
    order = np.argsort(-en)
    print("the 25 waves that end last: start, end, life, list length, walked, deepest contributor")
    for i in order[:25]:
        print("   %6.1f %6.1f %6.1f %6d %6d %6d" % (st[i], en[i], life[i], nlist[i], walked[i], deep[i]))
    early = st < 5
    late = st > 25
    for name, sel in (("started < 5 us", early), ("started > 25 us", late)):
        print(name, "n", int(sel.sum()), "life mean %.1f" % life[sel].mean(), "walked mean %.0f" % walked[sel].mean(), "ns per walked position %.0f" % (1e3 * life[sel].sum() / max(walked[sel].sum(), 1)))
