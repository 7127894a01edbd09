#!/usr/bin/env python
"""How well do 'lanes = pixels' blend kernels use their lanes on the bench scene, and what would finer work units buy?
CPU analysis on the oracle's forward state (sampled tiles): for every traversed (instance, pixel) pair that really blends
(alpha >= 1/255, position < the pixel's n_contrib) count, per granularity G in {16x8 strip, 8x8 quadrant, 8x4, 4x4}:
  pairs_G  = (instance, block) pairs with at least one blending pixel        (what a block-level compaction would walk)
  util_G   = blending (instance, pixel) pairs / (pairs_G * pixels per block)  (lane utilisation of those iterations)
  iters_4  = sum over groups of 4 sibling blocks of max(block list length)    (4 blocks per wavefront, one per DPP row)
Round 5 ("opposite-half pairing"): the backward's lane holds one pixel of the strip's LEFT 8x8 half and one of
the RIGHT half; an instance that blends only left pixels and one that blends only right pixels commute (no pixel sees both), so the
two could share ONE packed iteration; instances that blend in both halves are barriers.  Per (tile, strip, 64-position depth
segment) -- the backward's work unit -- the walk's sequence of blending instances is classified L / R / both and the iterations
that vanish are counted for pairing windows of 1, 4, 8 and unbounded (greedy zip of the L and R sub-sequences between barriers).
usage: python tools/lane_utilisation.py [ntiles] [workload: config2 | config3 | config4 | surfaces | init_state]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as Hh  # noqa: E402
from gscream_amd import synthetic as S  # noqa: E402


def paired_iterations(seq, window):
    """seq: array of 1 (left only), 2 (right only), 3 (both), in walk order.  Iterations when a left-only and a right-only instance
    at most `window` sequence positions apart (None = any) may share one; both-half instances are barriers."""
    it, pend_type, pend = 0, 0, []
    for k, t in enumerate(seq):
        if t == 3:
            it += 1
            pend, pend_type = [], 0
            continue
        if window is not None:
            while pend and pend[0] < k - window:
                pend.pop(0)
        if pend and pend_type != t:
            pend.pop(0)          # rides along with the pending opposite-half instance
        else:
            if not pend:
                pend_type = t
            pend.append(k)
            it += 1
    return it


def main():
    ntiles = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    wl = sys.argv[2] if len(sys.argv) > 2 else "config2"
    seed, P, W, H = {"config2": (1, 1_000_000, 1008, 567), "config3": (2, 1_000_000, 1008, 567), "config4": (3, 2_000_000, 1920, 1080),
                     "surfaces": (1, 1_000_000, 1008, 567), "init_state": (1, 0, 1008, 567)}[wl]
    if wl == "init_state":  # GScream's iteration-0 frame, decoded with the float64 decode ORACLE on the CPU (the product decode needs a GPU)
        import torch
        from gscream_amd import standin_model as SM
        from oracle import decode_oracle as DO
        from oracle import knn_oracle as KO
        pts = SM.voxelize(S.surface_point_cloud(1, 200_000, 0.6, H / W), 0.001)
        m = SM.Model.from_pcd(torch.from_numpy(pts).float(), torch.from_numpy(np.maximum(KO.mean_dist2(pts), 1e-7)).float(), K=10, seed=1)
        m.eval()
        view, proj, campos = S.camera_matrices(0.6, 0.6 * H / W)
        with torch.no_grad():
            xyz, color, opacity, unc, scaling, rot = DO.generate_neural_gaussians(SM.Camera(torch.from_numpy(campos)), m, None, False)
        s = dict(means3D=xyz.numpy(), scales=scaling.numpy(), rotations=rot.numpy(), opacities=opacity.numpy().reshape(-1, 1), uncertainties=unc.numpy().reshape(-1, 1),
                 colors=color.numpy(), W=W, H=H, tanfovx=0.6, tanfovy=0.6 * H / W, viewmatrix=view, projmatrix=proj, campos=campos, bg=np.zeros(3, np.float32), scale_modifier=1.0)
    else:
        s = (S.scene_surfaces if wl == "surfaces" else S.scene_slab)(seed, P, W, H)
    st = Hh.oracle_forward(s, nthreads=os.cpu_count())
    gx, gy = (W + 15) // 16, (H + 15) // 16
    rng = np.random.default_rng(0)
    tiles = rng.choice(gx * (gy - 1), size=ntiles, replace=False)  # full tiles only
    m2, con, ncon = st["means2D"].reshape(-1, 2), st["conic_opacity"].reshape(-1, 4), st["n_contrib"]
    grans = {"16x8": (16, 8), "8x8": (8, 8), "8x4": (8, 4), "4x4": (4, 4)}
    tot_pairs = 0
    acc = {k: dict(pairs=0, iters4=0) for k in grans}
    trav = 0
    windows = (1, 4, 8, None)
    pair_now, pair_single, pair_after = 0, 0, {w: 0 for w in windows}
    box_iters = box_pixels = box_instances = 0
    quad_pair_iters = quad_single_iters = half_iters = strip_iters = 0
    seg_len = 64 if gx * gy <= 4096 else 128
    for t in tiles:
        tx, ty = t % gx, t // gx
        a, b = st["ranges"][t]
        ids = st["point_list"][a:b].astype(np.int64)
        nc = ncon[ty * 16:ty * 16 + 16, tx * 16:tx * 16 + 16].astype(np.int64)   # [16,16]
        depth = int(nc.max())
        ids = ids[:depth]
        trav += depth
        xs, ys = np.arange(tx * 16, tx * 16 + 16, dtype=np.float32), np.arange(ty * 16, ty * 16 + 16, dtype=np.float32)
        dx = m2[ids, 0][:, None, None] - xs[None, None, :]
        dy = m2[ids, 1][:, None, None] - ys[None, :, None]
        power = -0.5 * (con[ids, 0][:, None, None] * dx * dx + con[ids, 2][:, None, None] * dy * dy) - con[ids, 1][:, None, None] * dx * dy
        alpha = np.minimum(0.99, con[ids, 3][:, None, None] * np.exp(power))
        blend = (power <= 0) & (alpha >= 1 / 255) & (np.arange(depth)[:, None, None] < nc[None])   # [n,16,16]
        tot_pairs += int(blend.sum())
        # Box-mapped lanes (round 5, modelled only): one wave per (tile, segment) whose 64 lanes are laid over the bounding box of the
        # instance's BLENDING pixels in the tile (width rounded up to a power of two so that lane -> pixel is shift / mask; pixel
        # state lives in LDS because a lane's pixel changes with every instance): iterations = ceil(box pixels / 64) per instance
        any_px = blend.any(axis=0)
        for i in range(depth):
            b = blend[i]
            if not b.any():
                continue
            ys_, xs_ = np.nonzero(b)
            w = int(xs_.max() - xs_.min() + 1); h = int(ys_.max() - ys_.min() + 1)
            w2 = 1 << (w - 1).bit_length()
            box_iters += -(-(w2 * h) // 64)
            box_pixels += int(b.sum())
            box_instances += 1
        # opposite-half pairing, per (strip, depth segment): the backward walks back to front, the count does not depend on direction
        for strip in range(2):
            left = blend[:, strip * 8:strip * 8 + 8, 0:8].any(axis=(1, 2))
            right = blend[:, strip * 8:strip * 8 + 8, 8:16].any(axis=(1, 2))
            pat = left.astype(np.int8) + 2 * right.astype(np.int8)
            for lo in range(0, depth, seg_len):
                hi = depth if lo // seg_len >= 7 else min(depth, lo + seg_len)   # the last of 8 segments takes the rest
                seq = pat[lo:hi]
                seq = seq[seq > 0]
                pair_now += int(seq.size)
                pair_single += int((seq < 3).sum())
                for w in windows:
                    pair_after[w] += paired_iterations(seq, w)
                if lo // seg_len >= 7:
                    break
        # Round 6: the backward on 8x8 QUADRANTS with TWO INSTANCES per packed iteration -- per (quadrant, depth
        # segment) the blending instances pair up in list order: ceil(n / 2) iterations of 64 pixels x 2 instances.  And its cousin that
        # keeps today's lane map: each 8x8 half of a strip walks ITS OWN list (.x = an instance of the left half, .y = one of the right
        # half, no pixel sees both): max(n_left, n_right) iterations per (strip, segment).
        for lo in range(0, depth, seg_len):
            hi = depth if lo // seg_len >= 7 else min(depth, lo + seg_len)
            qb = blend[lo:hi].reshape(hi - lo, 2, 8, 2, 8).any(axis=(2, 4))           # [n, qy, qx]: blends in that quadrant
            nq = qb.sum(0)                                                            # [2, 2]
            quad_pair_iters += int(((nq + 1) // 2).sum())
            quad_single_iters += int(nq.sum())
            half_iters += int(nq.max(axis=1).sum())                                   # per strip (qy): max over its two halves
            strip_iters += int(qb.any(axis=2).sum())                                  # today: instances that blend anywhere in the strip
            if lo // seg_len >= 7:
                break
        for k, (bw, bh) in grans.items():
            blk = blend.reshape(depth, 16 // bh, bh, 16 // bw, bw).any(axis=(2, 4))   # [n, by, bx]
            cnt = blk.sum(0)                                                        # list length per block
            acc[k]["pairs"] += int(cnt.sum())
            flat = cnt.reshape(-1)
            # groups of 4 sibling blocks (as laid out row-major): one wavefront
            acc[k]["iters4"] += int(flat.reshape(-1, 4).max(1).sum()) if flat.size >= 4 else int(flat.max())
    print(f"{ntiles} tiles, traversed instances per tile {trav / ntiles:.0f}, blending (instance, pixel) pairs per tile {tot_pairs / ntiles:.0f}")
    for k, (bw, bh) in grans.items():
        p = acc[k]["pairs"]
        print(f"  {k:5s}: (instance, block) pairs per tile {p / ntiles:8.0f}   lane utilisation {tot_pairs / (p * bw * bh):.3f}   "
              f"4-block wave iterations per tile {acc[k]['iters4'] / ntiles:8.0f}  (balance {p / 4 / max(acc[k]['iters4'], 1):.2f})")
    print(f"box-mapped lanes ({wl}): {box_instances / ntiles:.0f} blending (instance, tile) pairs per tile, {box_iters / ntiles:.0f} 64-lane iterations per tile "
          f"({box_iters / max(box_instances, 1):.2f} per pair), lane utilisation {box_pixels / max(box_iters * 64, 1):.3f}")
    # issue cycles per iteration (HISTORY 8's measured cost table applied to the instruction lists in DESIGN 11, round 6):
    #   today, (instance, 16x8 strip), two pixels per lane:                 head 72 + blend 118 + reduction of 6 partials 106 = 296
    #   two instances per lane (either variant): per-half dy / operands     head 88 + blend 106 (no cross-pixel adds) + reduction of
    #   12 partials through one transposing butterfly (12 + 10 masked DPP adds, 4 dy products, 3 + 2 permlane swaps + adds, 4 quad steps, 2 LDS adds) 174 = 368
    C_NOW, C_PAIR = 296, 368
    print(f"two instances per packed iteration ({wl}, segments of {seg_len}), blending iterations per tile: today (instance, strip) {strip_iters / ntiles:.0f}; "
          f"8x8 quadrant pairs {quad_pair_iters / ntiles:.0f} (from {quad_single_iters / ntiles:.0f} (instance, quadrant) units: {quad_single_iters / max(2 * strip_iters, 1):.3f} of today's lanes); "
          f"own list per 8x8 half {half_iters / ntiles:.0f}")
    print(f"  modelled issue cycles per tile: today {strip_iters / ntiles * C_NOW:.0f}; quadrant pairs {quad_pair_iters / ntiles * C_PAIR:.0f} "
          f"({quad_pair_iters * C_PAIR / max(strip_iters * C_NOW, 1):.3f} x); own list per half {half_iters / ntiles * C_PAIR:.0f} "
          f"({half_iters * C_PAIR / max(strip_iters * C_NOW, 1):.3f} x)   [gate: <= 0.88 x]")
    print(f"opposite-half pairing ({wl}, segments of {seg_len}): blending (instance, strip) iterations per tile {pair_now / ntiles:.0f}, "
          f"{pair_single / max(pair_now, 1):.1%} of them touch one 8x8 half only")
    for w in windows:
        print(f"  window {str(w):>4s}: {pair_after[w] / ntiles:8.0f} iterations per tile, {1 - pair_after[w] / max(pair_now, 1):.1%} vanish")


    # (printed by main, kept separate for readability)
if __name__ == "__main__":
    main()
