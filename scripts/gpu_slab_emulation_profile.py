import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from sphfluidsimulation_amd import presets
from sphfluidsimulation_amd.simulation import SphFluidSimulation
n = 16777216
slabs = int(sys.argv[1])
R = presets.scene_bucket_resolution(n)
pos0, vel0 = presets.dam_break(n, 0.8, 1)
dt = presets.bench_dt_frame(R)
overlap = not (len(sys.argv) > 2 and sys.argv[2] == 'nooverlap')
kw = dict(device_ids=[0] * slabs, overlap=overlap) if slabs > 1 else {}
sim = SphFluidSimulation(particleNumber=n, bucketResolution=R, damFillRate=0.8, **kw).Start(pos0, vel0)
sim.step_n(dt, 15); sim.synchronize()
sim.OnDestroy()
