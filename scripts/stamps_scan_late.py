"""Diagnostic (build with SPH_HIPCC_EXTRA=-DSPH_STAMPS=3): the longest work item of every block of the scanning kernel
k_velpos_brick in the last sub-step of a late dam-break frame.   usage: python scripts/stamps_scan_late.py [frames]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sphfluidsimulation_amd import presets, _native as N
from sphfluidsimulation_amd.simulation import SphFluidSimulation
n = 16777216; R = presets.scene_bucket_resolution(n)
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 280
pos0, vel0 = presets.dam_break(n, 0.8, 1)
sim = SphFluidSimulation(particleNumber=n, bucketResolution=R, damFillRate=0.8).Start(pos0, vel0)
sim.step_n(presets.bench_dt_frame(R), frames)
buf = np.zeros((3072, 8), np.uint64)
L = N.lib(); L.sph_debug_read_order.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
assert L.sph_debug_read_order(sim._ctx, buf.ctypes.data_as(C.c_void_p), buf.nbytes) == 0
b = buf[buf[:, 7] > 0].astype(np.int64)
nb = (R + 3) // 4
print("blocks with work:", len(b), " work items:", b[0, 6] if len(b) else 0, " kernel span (cycles):", b[:, 1].max() - b[:, 0].min())
tot = b[:, 1] - b[:, 0]
print("block busy time: mean %.0f median %.0f p90 %.0f max %.0f cycles" % (tot.mean(), np.median(tot), np.percentile(tot, 90), tot.max()))
order = np.argsort(-b[:, 2])
for i in order[:12]:
    br = b[i, 3]; bx, by, bz = br % nb, (br // nb) % nb, br // (nb * nb)
    print(f"  longest item {b[i, 2]:9d} cycles  brick ({bx},{by},{bz})  targets swept {b[i, 4]}  handed over {b[i, 5]}  items of the block {b[i, 7]}")
for lo, hi in ((0, 1), (1, 2), (2, 1000)):
    sel = ((b[:, 3] // nb) % nb >= lo) & ((b[:, 3] // nb) % nb < hi)
    if sel.any(): print(f"  bricks of layer y in [{lo},{hi}): {sel.sum()} longest items, mean {b[sel, 2].mean():.0f} max {b[sel, 2].max()} cycles, swept {b[sel, 4].mean():.0f} handed over {b[sel, 5].mean():.0f}")
