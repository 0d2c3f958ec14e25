"""Diagnostic: per-brick phase durations of k_velpos_brick (build with SPH_HIPCC_EXTRA=-DSPH_STAMPS=1) or of k_velpos_tile
(-DSPH_STAMPS=2; phases: set-up + staging, entry rows + target, walk of thread 0, integrate + store of thread 0).
  usage: python scripts/stamps.py [exact|tolerance]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sphfluidsimulation_amd import presets, _native as N
from sphfluidsimulation_amd.simulation import SphFluidSimulation
n = 16777216; R = presets.scene_bucket_resolution(n)
pos0, vel0 = presets.dam_break(n, 0.8, 1)
sim = SphFluidSimulation(particleNumber=n, bucketResolution=R, damFillRate=0.8, exact_order=(sys.argv[1:] or ['exact'])[0] == 'exact').Start(pos0, vel0)
sim.step_n(presets.bench_dt_frame(R), 4)
nb = ((R + 3) // 4) ** 3
buf = np.zeros((nb, 8), np.uint64)
L = N.lib(); L.sph_debug_read_order.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
assert L.sph_debug_read_order(sim._ctx, buf.ctypes.data_as(C.c_void_p), buf.nbytes) == 0
b = buf[buf[:, 4] > 0].astype(np.int64)
if b[:, 5].max() > 0:      # (k_velpos_tile: bricks that returned early hold the sort's leftovers -- keep the rows that look like stamps)
    o = [0, 5, 6, 1, 2, 3, 4]
    ok = np.ones(len(b), bool)
    for a, c in zip(o[:-1], o[1:]):
        ok &= b[:, a] <= b[:, c]
    ok &= (b[:, 4] - b[:, 0]) < 4_000_000
    b = b[ok]
print("bricks with targets:", len(b), "of", nb)
phases = (("setup", 0, 1), ("scan", 1, 2), ("phase2", 2, 3), ("integrate+store", 3, 4), ("total", 0, 4))
if b[:, 5].max() > 0:      # k_velpos_tile (SPH_STAMPS=2): start -> rows known -> cell table + offsets -> tile staged -> target ready -> walked -> stored
    # (with brick records: start -> state known -> record in LDS -> tile staged -> ...)
    phases = (("rows (+cell loads) | state", 0, 5), ("cell table + offsets | record", 5, 6), ("tile staged", 6, 1), ("entry rows + target", 1, 2), ("walk", 2, 3),
              ("integrate + store", 3, 4), ("total", 0, 4))
for name, a, c in phases:
    dt = (b[:, c] - b[:, a])
    print(f"{name:16s} mean {dt.mean():9.0f} cycles  median {np.median(dt):9.0f}  p90 {np.percentile(dt, 90):9.0f}  ({dt.mean()/100:.1f} us at 100 MHz ticks?)")
if b[:, 7].max() > 0:
    ok = b[:, 7] > 0
    print("s_memtime ticks per s_memrealtime tick (100 MHz): median %.2f -> clock of s_memtime %.0f MHz" % (np.median((b[ok, 4] - b[ok, 0]) / b[ok, 7]), 100 * np.median((b[ok, 4] - b[ok, 0]) / b[ok, 7])))
span = b[:, 4].max() - b[:, 0].min()
print("kernel span (ticks):", span)
