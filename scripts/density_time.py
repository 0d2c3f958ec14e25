"""Diagnostic: per-pass times at 16.7 M particles with the neighbour lists on and off (the density kernel with and without RECORD)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sphfluidsimulation_amd import presets
from sphfluidsimulation_amd.simulation import SphFluidSimulation
n = 16777216; R = presets.scene_bucket_resolution(n)
pos0, vel0 = presets.dam_break(n, 0.8, 1)
for lists in (True, False):
    sim = SphFluidSimulation(particleNumber=n, bucketResolution=R, damFillRate=0.8, profile=True, lists=lists).Start(pos0, vel0)
    dt = presets.bench_dt_frame(R)
    sim.step_n(dt, 5); sim.synchronize(); sim.reset_stats()
    sim.step_n(dt, 10); sim.synchronize()
    st = sim.stats()
    print(f"lists {lists}: grid {st['ms_grid']/max(st['launches_grid'],1):.3f} density {st['ms_density']/max(st['launches_density'],1):.3f} "
          f"velpos {st['ms_velpos']/max(st['launches_velpos'],1):.3f} ms", flush=True)
    sim.OnDestroy()
