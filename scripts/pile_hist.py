"""Diagnostic: how the targets of a late dam-break frame are distributed over cell occupancies (who is worth a per-cell kernel)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sphfluidsimulation_amd import presets
from sphfluidsimulation_amd.simulation import SphFluidSimulation
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16777216
R = presets.scene_bucket_resolution(n)
pos0, vel0 = presets.dam_break(n, 0.8, 1)
sim = SphFluidSimulation(particleNumber=n, bucketResolution=R, damFillRate=0.8).Start(pos0, vel0)
dt = presets.bench_dt_frame(R)
done = 0
for frames in [int(a) for a in sys.argv[2:]] or [100, 150, 200, 280]:
    sim.step_n(dt, frames - done); done = frames
    p, v, _ = sim.download()
    cell = np.clip((p[:, :3] * np.float32(R - 1)).astype(np.int32), 0, R - 1)
    key = cell[:, 0].astype(np.int64) + R * (cell[:, 1].astype(np.int64) + R * cell[:, 2].astype(np.int64))
    cnt = np.bincount(key, minlength=R ** 3)
    edges = [0, 8, 16, 32, 48, 64, 96, 128, 192, 256, 384, 100000]
    line = []
    for a, b in zip(edges[:-1], edges[1:]):
        sel = (cnt > a) & (cnt <= b)
        line.append(f"({a},{b}]: {sel.sum()} cells / {cnt[sel].sum() / n:.3f}")
    # neighbourhood load of the members of crowded cells: listed candidates in their 27 cells
    listed = np.minimum(cnt, 32).reshape(R, R, R).astype(np.float32)
    from scipy import ndimage
    nb = ndimage.uniform_filter(listed, size=3, mode="constant") * 27.0
    big = cnt.reshape(R, R, R) > 32
    print(f"frame {frames}: " + "  ".join(line))
    print(f"   cells > 32: mean candidates in the 27-cell stencil {nb[big].mean():.0f} (max {nb.max():.0f}); y-layer of the cells > 32: "
          + str(np.bincount(np.nonzero(big)[1])[:6]) + " x-layer: " + str(np.bincount(np.nonzero(big)[2])[:4]) + " ... " + str(np.bincount(np.nonzero(big)[2], minlength=R)[-4:]), flush=True)
