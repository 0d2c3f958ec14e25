"""frame time and HOST time per sph_step with and without HIP-graph replay (no profiling events: they disable graphs)"""
import sys, time
sys.path.insert(0, ".")
from sphfluidsimulation_amd import presets
from sphfluidsimulation_amd.simulation import SphFluidSimulation
for n in (262144, 1048576, 16777216):
    R = presets.scene_bucket_resolution(n)
    pos0, vel0 = presets.dam_break(n, 0.8, 1)
    dt = presets.bench_dt_frame(R)
    for graphs in (True, False):
        sim = SphFluidSimulation(particleNumber=n, bucketResolution=R, damFillRate=0.8, graphs=graphs).Start(pos0, vel0)
        sim.step_n(dt, 10); sim.synchronize()
        t = time.perf_counter(); sim.step_n(dt, 20); host = (time.perf_counter() - t) / 20
        sim.synchronize(); el = (time.perf_counter() - t) / 20
        # host cost alone: one frame issued into an idle queue
        h1 = []
        for _ in range(5):
            sim.synchronize(); t = time.perf_counter(); sim.Update(dt); h1.append(time.perf_counter() - t)
        sim.synchronize()
        print(f"n={n} graphs={graphs}: {el*1e3:.3f} ms/frame, host thread {min(h1)*1e6:.0f} us per sph_step (idle queue), "
              f"{host*1e3:.3f} ms/frame when 20 frames are queued", flush=True)
        sim.OnDestroy()
