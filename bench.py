#!/usr/bin/env python3
"""Headline benchmark: particle-steps/s of the SPH frame update on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N=1 default; N>1 via torch.distributed.run)

One "step" = one reference frame, SphFluidSimulation.cs::Update (:96-102): grid build + density
pass + 5 force/integrate sub-steps, on the synthetic dam-break of BASELINE.json
(16 777 216 particles, R = 185, scene parameters, dt_frame = 0.7667 h).  Inputs are resident in
HBM before the timed region.  Prints ONE JSON line (rank 0).

Extra objects on that line:
  roofline      dominant kernel (k_velpos_list, the VelPos replacement walking the frame's neighbour
                lists; k_velpos_brick when lists are off): algorithmic bytes per launch
                (52 B/particle) / its mean launch duration from HIP events recorded
                around that kernel on the engine's stream inside the timed region, against the
                8 TB/s HBM peak.  The sweeps are FP32-VALU bound: valu_frac is next to it.
  cpu_baseline  oracle/ (CPU restatement of the reference, "port") timed on this host for ONE frame
                of the same workload, all OpenMP threads; also used to check the GPU result.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
VALU_PEAK_TFLOPS = 157.3       # fp32 vector peak
BYTES_PER_SUBSTEP = 52         # (12 pos + 12 vel + 4 rho) R + (12 + 12) W
BYTES_PER_STEP = 32 + 52 * 5   # 292 B per particle-step at S = 5
FLOPS_DENSITY, FLOPS_FORCE = 16, 55  # per candidate, SURVEY.md 8d


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--particles", type=int, default=16777216)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-slab", action="store_true", help="diagnostic: run the z-slab driver even with one rank")
    return ap.parse_args()


CPU_FRAMES = 2   # frame 1 scans (the jittered lattice starts partly outside the unit box), frame 2 walks neighbour lists


def cpu_baseline(n, R, fill, pos0, vel0, dt_frame):
    """Oracle timed on the host cores for CPU_FRAMES frames of the same workload (checker + baseline)."""
    from oracle import sph_oracle as O
    O.build()
    threads = O.max_threads()
    p = O.derive(n, R, fill, viscosity=0.01, rest_density=1.5, gas_constant=150.0, stiffness=5000.0,
                 damping=10.0, num_threads=threads)
    sim = O.OracleSimulation(p)
    sim.upload(pos0, vel0)
    t0 = time.perf_counter()
    for _ in range(CPU_FRAMES):
        sim.Update(dt_frame)
    dt = time.perf_counter() - t0
    out = sim.download()
    sim.close()
    return dict(value=n * CPU_FRAMES / dt, unit="particle-steps/s", cores=threads, kind="port",
                sample=f"{CPU_FRAMES} frames (grid + density + 5 sub-steps each) of the same {n}-particle dam-break, "
                       f"{dt:.2f} s wall, OpenMP {threads} threads, gcc -O2 -ffp-contract=off"), out


def pmc_traffic(kernel_substr):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC run
    (profiles/*_hbm_traffic_pmc.csv: FETCH_SIZE and WRITE_SIZE in separate passes, KiB).  FETCH_SIZE is
    doubled as MI355X_MICROARCH.md prescribes for wide streaming reads (calibrated on k_upload_split in
    profiles/README.md), so this is an upper bound for the 4-byte-per-lane part of the kernel's reads."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_hbm_traffic_pmc.csv")))
    if not files:
        return None, None
    for row in csv.DictReader(open(files[-1])):
        if kernel_substr in row["kernel"]:
            return (2.0 * float(row["avg_FETCH_SIZE_raw"]) + float(row["avg_WRITE_SIZE_raw"])) * 1024.0, \
                os.path.relpath(files[-1], ROOT) + " (16 777 216 particles)"
    return None, None


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    import torch
    import torch.distributed as dist

    from sphfluidsimulation_amd import presets

    n, fill = presets.next_power_of_two(a.particles), 0.8
    R = presets.scene_bucket_resolution(n)
    dt_frame = presets.bench_dt_frame(R)
    workload = f"dam-break Preset1 lattice, {n} particles, R={R}, fill 0.8, scene parameters, dt_frame=0.7667h, seed {a.seed}"

    if world > 1 or a.force_slab:
        from sphfluidsimulation_amd import distributed as D
        result = D.bench_main(a, n, R, fill, dt_frame, rank, world, local_rank)
        if rank == 0:
            result["config"]["workload"] = workload
            print(json.dumps(result), flush=True)
        return

    from sphfluidsimulation_amd.simulation import SphFluidSimulation
    torch.cuda.set_device(0)
    pos0, vel0 = presets.dam_break(n, fill, a.seed)
    sim = SphFluidSimulation(particleNumber=n, bucketResolution=R, damFillRate=fill, viscosity=0.01,
                             restDensity=1.5, gasConstant=150.0, stiffnessCoefficient=5000.0,
                             dampingCoefficient=10.0, device_id=0, profile=True).Start(pos0, vel0)

    # GPU result after CPU_FRAMES frames, checked against the CPU baseline run below; then start over for the timing
    warm = a.warmup
    frame1 = None
    if not a.no_cpu_baseline:
        for _ in range(CPU_FRAMES):
            sim.Update(dt_frame)
        frame1 = sim.download()
        sim.upload(pos0, vel0)
    sim.step_n(dt_frame, warm)
    sim.synchronize()
    sim.reset_stats()

    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sim.step_n(dt_frame, a.steps)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    st = sim.stats()
    # SURVEY.md 8d asks for median and mean: a few more frames, each synchronised and timed on its own (never part of `value`)
    singles = []
    for _ in range(min(a.steps, 10)):
        sim.synchronize()
        ts = time.perf_counter()
        sim.Update(dt_frame)
        sim.synchronize()
        singles.append((time.perf_counter() - ts) * 1e3)
    # row f3, the render feed after the path (UpdateMeshProperties, never part of `value`): 112 B per particle
    sim.UpdateMeshProperties()
    sim.synchronize()
    tm = time.perf_counter()
    for _ in range(20):
        sim.UpdateMeshProperties()
    sim.synchronize()
    mesh_ms = (time.perf_counter() - tm) * 1e3 / 20
    # boundary cost when the caller hands over HOST arrays every frame (never part of `value`)
    t1 = time.perf_counter()
    sim.upload(pos0, vel0)
    t2 = time.perf_counter()
    sim.download()
    t3 = time.perf_counter()
    ms_per_step = elapsed * 1e3 / a.steps
    value = n * a.steps / elapsed
    pbar = st["candidates_density"] / n
    vel_ms = st["ms_velpos"] / max(st["launches_velpos"], 1)           # whole sub-step (all its kernels)
    listed = st["launches_velpos_list"] > 0
    kernel = "k_velpos_list" if listed else "k_velpos_brick"
    kernel_ms = st["ms_velpos_list"] / st["launches_velpos_list"] if listed else vel_ms
    achieved = BYTES_PER_SUBSTEP * n / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    valu_tflops = value * pbar * (FLOPS_DENSITY + FLOPS_FORCE * 5) / 1e12
    traffic, traffic_src = pmc_traffic(kernel) if n == 16777216 else (None, None)
    out = {
        "metric": "particle-steps/s", "value": value, "unit": "particle-steps/s", "n_gpus": 1,
        "steps": a.steps, "warmup": warm, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": workload, "particles": n, "bucket_resolution": R, "substeps": 5,
                   "mode": "exact_order"},
        "roofline": {
            "bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
            "algorithmic_bytes_per_launch": BYTES_PER_SUBSTEP * n,
            "kernel_ms": kernel_ms, "launches": st["launches_velpos_list"] if listed else st["launches_velpos"],
            "step_hbm_frac": value * BYTES_PER_STEP / (HBM_PEAK_GBS * 1e9),
            "valu_frac": valu_tflops / VALU_PEAK_TFLOPS,
            "candidates_per_particle": pbar,
        },
        "passes_ms": {"grid": st["ms_grid"] / max(st["launches_grid"], 1),
                      "density": st["ms_density"] / max(st["launches_density"], 1),
                      "velpos": vel_ms},
        "particle_substeps_per_s": value * 5,
        "single_frames_ms": {"n": len(singles), "median": float(np.median(singles)) if singles else None,
                             "mean": float(np.mean(singles)) if singles else None},
        "host_boundary": {"upload_ms": (t2 - t1) * 1e3, "download_ms": (t3 - t2) * 1e3,
                          "pcie_inclusive_particle_steps_per_s": n / (elapsed / a.steps + (t3 - t1))},
        "render_feed": {"kernel": "k_mesh_properties", "ms": mesh_ms, "algorithmic_bytes": 112 * n,
                        "achieved_GBs": 112 * n / (mesh_ms * 1e-3) / 1e9,
                        "hbm_frac": 112 * n / (mesh_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
        "neighbour_lists": {
            "listed_fraction_of_target_substeps": st["listed_target_substeps"] / max(n * st["launches_velpos"], 1),
            "entries_per_listed_target": st["list_entries"] / max(st["listed_target_substeps"], 1),
            "scanned_brick_substeps": st["scanned_brick_substeps"]},
        "max_cell_occupancy": st["max_cell_occupancy"], "nan_resets": st["nan_resets"],
        "queued_targets_last_sweep": st["queued_targets"],
    }
    if not a.no_cpu_baseline:
        base, (po, vo, ro) = cpu_baseline(n, R, fill, pos0, vel0, dt_frame)
        out["cpu_baseline"] = base
        out["parity_frames"] = CPU_FRAMES
        out["parity_frame1_max_abs_diff"] = {   # (key kept from round 1; the state compared is the one after parity_frames)
            "pos": float(np.abs(frame1[0] - po).max()), "vel": float(np.abs(frame1[1] - vo).max()),
            "rho": float(np.abs(frame1[2] - ro).max())}
    sim.OnDestroy()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
