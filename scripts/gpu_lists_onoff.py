"""frame time with neighbour lists on / off at several sizes and phases of the dam break (no profiling events)"""
import sys, time
sys.path.insert(0, ".")
from sphfluidsimulation_amd import presets
from sphfluidsimulation_amd.simulation import SphFluidSimulation
for n in (262144, 1048576, 4194304):
    R = presets.scene_bucket_resolution(n)
    pos0, vel0 = presets.dam_break(n, 0.8, 1)
    dt = presets.bench_dt_frame(R)
    for lists in (True, False):
        sim = SphFluidSimulation(particleNumber=n, bucketResolution=R, damFillRate=0.8, lists=lists).Start(pos0, vel0)
        out = []
        for phase in range(6):
            sim.synchronize(); t = time.perf_counter(); sim.step_n(dt, 20); sim.synchronize()
            out.append((time.perf_counter() - t) / 20 * 1e3)
        st = sim.stats()
        print(f"n={n} lists={lists}: ms/frame per 20-frame phase: " + " ".join(f"{x:.2f}" for x in out) +
              f"   listed {st['listed_target_substeps'] / (n * 600.0):.2f}", flush=True)
        sim.OnDestroy()
