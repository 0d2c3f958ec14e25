"""N z-slabs of the 16.7 M dam break on ONE GPU (multi-device context with repeated ordinals: peer-copy transport): the
whole-frame time divided by N bounds the per-rank cost of an N-GPU run from above (all fixed per-slab overheads included,
the slabs' kernels share one GPU), WITHOUT the xGMI transfer time."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sphfluidsimulation_amd import presets
from sphfluidsimulation_amd.simulation import SphFluidSimulation
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16777216
R = presets.scene_bucket_resolution(n)
pos0, vel0 = presets.dam_break(n, 0.8, 1)
dt = presets.bench_dt_frame(R)
overlap = not (len(sys.argv) > 2 and sys.argv[2] == "nooverlap")
for slabs in (1, 2, 4, 8):
    kw = dict(device_ids=[0] * slabs, overlap=overlap) if slabs > 1 else {}
    sim = SphFluidSimulation(particleNumber=n, bucketResolution=R, damFillRate=0.8, **kw).Start(pos0, vel0)
    sim.step_n(dt, 5); sim.synchronize()
    sim.reset_stats()
    t = time.perf_counter(); sim.step_n(dt, 15); sim.synchronize(); el = (time.perf_counter() - t) / 15 * 1e3
    st = sim.stats()
    print(f"{slabs} slab(s): {el:7.2f} ms/frame = {el / slabs:6.2f} ms per slab; host syncs/frame {st['host_syncs'] / 15:.2f}, violations {st['halo_violations']}, "
          f"listed {st['listed_target_substeps'] / (n * 75.0):.3f}", flush=True)
    sim.OnDestroy()
