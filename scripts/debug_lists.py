import numpy as np, sys
sys.path.insert(0, ".")
from sphfluidsimulation_amd import presets
from sphfluidsimulation_amd.simulation import SphFluidSimulation
for n in (4096, 16384, 65536):
    fill = 0.8
    R = presets.scene_bucket_resolution(n)
    pos0, vel0 = presets.dam_break(n, fill, seed=31)
    print("n", n, "R", R, "pos range", pos0[:, :3].min(), pos0[:, :3].max())
    sim = SphFluidSimulation(particleNumber=n, bucketResolution=R, damFillRate=fill).Start(pos0, vel0)
    dt = presets.bench_dt_frame(R)
    for f in range(4):
        sim.reset_stats()
        sim.Update(dt)
        st = sim.stats()
        print(" frame", f, {k: st[k] for k in ("listed_target_substeps", "scanned_brick_substeps", "list_entries", "queued_targets", "max_cell_occupancy", "dropped_particles")})
    sim.OnDestroy()
