"""frame / density time of the list-free engine (SPH_FLAG_NO_LISTS) at 16.7 M: the path small scenes and late phases take"""
import sys, time
sys.path.insert(0, ".")
from sphfluidsimulation_amd import presets
from sphfluidsimulation_amd.simulation import SphFluidSimulation
n = 16777216
R = presets.scene_bucket_resolution(n)
pos0, vel0 = presets.dam_break(n, 0.8, 1)
dt = presets.bench_dt_frame(R)
sim = SphFluidSimulation(particleNumber=n, bucketResolution=R, damFillRate=0.8, lists=False, profile=True).Start(pos0, vel0)
sim.step_n(dt, 5); sim.synchronize(); sim.reset_stats()
t = time.perf_counter(); sim.step_n(dt, 10); sim.synchronize(); el = (time.perf_counter() - t) / 10
st = sim.stats()
print(f"nolists: {el*1e3:.2f} ms/frame density {st['ms_density']/st['launches_density']:.3f} velpos {st['ms_velpos']/st['launches_velpos']:.3f}")
