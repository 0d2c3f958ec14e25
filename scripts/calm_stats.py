"""Diagnostic (not a judged number): how many targets of a late dam-break frame could take their pair sums from recorded
neighbour lists under different calmness criteria.  Runs F frames on the GPU, downloads the positions, steps ONE more frame,
downloads again: delta = displacement over that frame (the test of sub-step k sees (k-1)/5 of it).

    python scripts/calm_stats.py [particles] [frames] [skin/h ...]

Criteria, per 4x4x4-bin brick with its one-cell neighbourhood (what k_brick_activity covers):
  A  range of delta over ALL slots of the neighbourhood <= 0.9 skin          (the engine's test)
  B  range over the VISIBLE slots only (first 32 ids of a cell = the candidates) + per target: farthest corner of that box
     from the target's own delta <= 0.9 skin
"""
import os, sys, time
import numpy as np
from scipy import ndimage
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sphfluidsimulation_amd import presets
from sphfluidsimulation_amd.simulation import SphFluidSimulation

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16777216
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 280
skins = [float(a) for a in sys.argv[3:]] or [0.04, 0.08, 0.16]
R = presets.scene_bucket_resolution(n)
h = np.float32(1.0) / np.float32(R - 1)
pos0, vel0 = presets.dam_break(n, 0.8, 1)
sim = SphFluidSimulation(particleNumber=n, bucketResolution=R, damFillRate=0.8).Start(pos0, vel0)
dt = presets.bench_dt_frame(R)
sim.step_n(dt, frames)
pa, va, _ = sim.download()
sim.step_n(dt, 1)
pb, vb, _ = sim.download()
sim.OnDestroy()
pa = pa[:, :3]; pb = pb[:, :3]
delta_frame = (pb - pa).astype(np.float32)
cell = np.clip((pa * np.float32(R - 1)).astype(np.int32), 0, R - 1)
key = cell[:, 0].astype(np.int64) + R * (cell[:, 1].astype(np.int64) + R * cell[:, 2].astype(np.int64))
order = np.argsort(key, kind="stable")            # stable: ascending id inside a cell
ks = key[order]
start = np.searchsorted(ks, np.arange(R ** 3 + 1))
rank = np.arange(n) - start[ks]
visible = np.zeros(n, bool); visible[order] = rank < 32
cnt = np.diff(start)
print(f"N {n} R {R} frame {frames}: max occupancy {cnt.max()}, cells over 32: {(cnt > 32).sum()}, invisible particles {(~visible).sum()} "
      f"({(~visible).mean():.3f}), |v| mean {np.linalg.norm(va[:, :3], axis=1).mean():.3f} max {np.abs(va[:, :3]).max():.2f}, "
      f"|delta_frame|/h mean {np.linalg.norm(delta_frame, axis=1).mean() / h:.4f} p99 {np.quantile(np.linalg.norm(delta_frame, axis=1), 0.99) / h:.4f}")

def cell_minmax(sel, dl):
    lo = np.full((3, R ** 3), np.inf, np.float32); hi = np.full((3, R ** 3), -np.inf, np.float32)
    for a in range(3):
        np.minimum.at(lo[a], key[sel], dl[sel, a]); np.maximum.at(hi[a], key[sel], dl[sel, a])
    return lo.reshape(3, R, R, R), hi.reshape(3, R, R, R)      # index [a, z, y, x]

def brick_of(c):
    return c // 4

nb = (R + 3) // 4
def brick_boxes(lo, hi):
    # neighbourhood of a brick = its cells +- 1: 3x3x3 min filter per cell, then the min over the brick's cells
    out_lo = np.empty((3, nb, nb, nb), np.float32); out_hi = np.empty((3, nb, nb, nb), np.float32)
    pad = nb * 4 - R
    for a in range(3):
        l = ndimage.minimum_filter(lo[a], size=3, mode="constant", cval=np.inf)
        u = ndimage.maximum_filter(hi[a], size=3, mode="constant", cval=-np.inf)
        l = np.pad(l, ((0, pad),) * 3, constant_values=np.inf); u = np.pad(u, ((0, pad),) * 3, constant_values=-np.inf)
        out_lo[a] = l.reshape(nb, 4, nb, 4, nb, 4).min(axis=(1, 3, 5)); out_hi[a] = u.reshape(nb, 4, nb, 4, nb, 4).max(axis=(1, 3, 5))
    return out_lo, out_hi

bz, by, bx = cell[:, 2] // 4, cell[:, 1] // 4, cell[:, 0] // 4
crowded = cnt[key] > 32
for k in (2, 3, 4, 5):                      # sub-step k tests the displacement of k - 1 sub-steps
    dl = delta_frame * np.float32((k - 1) / 5.0)
    lo_all, hi_all = brick_boxes(*cell_minmax(np.ones(n, bool), dl))
    lo_vis, hi_vis = brick_boxes(*cell_minmax(visible, dl))
    rng_all = np.sqrt(sum(np.maximum(hi_all[a] - lo_all[a], 0) ** 2 for a in range(3)))
    rng_vis = np.sqrt(sum(np.maximum(hi_vis[a] - lo_vis[a], 0) ** 2 for a in range(3)))
    far = np.sqrt(sum(np.maximum(np.abs(dl[:, a] - lo_vis[a][bz, by, bx]), np.abs(dl[:, a] - hi_vis[a][bz, by, bx])) ** 2 for a in range(3)))
    far = np.where(np.isfinite(far), far, 0.0)          # no visible candidate around: nothing to miss
    for skin in skins:
        ok = np.float32(0.9 * skin) * h
        A = rng_all[bz, by, bx] <= ok
        Bb = rng_vis[bz, by, bx] <= ok                   # brick-level part of B (candidates calm among themselves is implied by the per-target test only if the target is visible; keep it as information)
        B = far <= ok
        print(f"sub-step {k} skin {skin:.2f}: A {A.mean():.3f} (crowded targets {A[crowded].mean():.3f})   "
              f"B {B.mean():.3f} (crowded {B[crowded].mean():.3f}; bricks with calm candidates hold {Bb.mean():.3f} of the targets)", flush=True)
