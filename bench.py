#!/usr/bin/env python3
"""Headline benchmark: particle-steps/s of the SPH frame update on MI355X.

    python bench.py --gpus N --steps K --warmup W

N = 1 (default): one GPU.  N > 1: one rank per GPU -- either launched by torch.distributed.run (RANK / LOCAL_RANK /
WORLD_SIZE in the environment) or, when started plainly, bench.py starts its own N rank processes BEFORE anything touches
a GPU and relays rank 0's line.

One "step" = one reference frame, SphFluidSimulation.cs::Update (:96-102): grid build + density
pass + 5 force/integrate sub-steps, on the synthetic dam-break of BASELINE.json
(16 777 216 particles, R = 185, scene parameters, dt_frame = 0.7667 h).  Inputs are resident in
HBM before the timed region.  Prints ONE JSON line (rank 0).

Extra objects on that line:
  roofline      dominant kernel (k_velpos_list, the VelPos replacement walking the frame's neighbour
                lists; k_velpos_brick when lists are off): algorithmic bytes per launch
                (52 B/particle) / its mean launch duration from HIP events recorded
                around that kernel on the engine's stream inside the timed region, against the
                8 TB/s HBM peak.  The sweeps are FP32-VALU bound: `valu_issue_frac` is the kernel's
                EXECUTED vector instructions per second (SQ_INSTS_VALU of the committed rocprofv3 --pmc run of this
                command, profiles/*_sq_pmc.csv) against the chip's issue peak (1024 SIMDs x 2.4 GHz / 2 cycles), and
                `bound` names the larger of the two fractions.
  cpu_baseline  oracle/ (CPU restatement of the reference, "port") timed on this host: 3 frames of the same workload on
                all OpenMP threads (also the parity check of the GPU result), plus the reference's own scene size
                (262 144, all threads and 1 thread) and the 4 096-particle case.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
VALU_PEAK_TFLOPS = 157.3       # fp32 vector peak
VALU_ISSUE_PEAK = 1024 * 2.4e9 / 2.0   # wave64 vector instructions per second: 256 CUs x 4 SIMDs, 2 cycles each at 2.4 GHz
BYTES_PER_SUBSTEP = 52         # (12 pos + 12 vel + 4 rho) R + (12 + 12) W
FLOPS_DENSITY, FLOPS_FORCE = 16, 55  # per candidate, SURVEY.md 8d


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--particles", type=int, default=16777216)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--mode", choices=["exact", "tolerance"], default="exact",
                    help="exact: bit-identical to the CPU oracle (SphParams.exact_order = 1); tolerance: hardware reciprocal / sqrt, parity bar 1e-4")
    ap.add_argument("--list-skin", type=float, default=0.0, help="diagnostic: neighbour-list skin / h (0 = the default 0.04)")
    ap.add_argument("--global-lists", action="store_true", help="diagnostic: lists of global slots gathered through L1 by k_velpos_list (SPH_FLAG_GLOBAL_LISTS)")
    ap.add_argument("--precision", type=int, default=0, help="1: positions stored as fp16 cell offsets (BASELINE config 5), arithmetic fp32")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the S = 1 variant and the late-phase (frames 281-300) figure")
    ap.add_argument("--force-slab", action="store_true", help="diagnostic: run the z-slab driver even with one rank")
    return ap.parse_args()


CPU_FRAMES = 3   # frame 1 scans (the jittered lattice starts partly outside the unit box), frames 2-3 walk neighbour lists


def oracle_run(O, n, R, fill, pos0, vel0, dt_frame, frames, threads):
    p = O.derive(n, R, fill, viscosity=0.01, rest_density=1.5, gas_constant=150.0, stiffness=5000.0,
                 damping=10.0, num_threads=threads)
    sim = O.OracleSimulation(p)
    sim.upload(pos0, vel0)
    t0 = time.perf_counter()
    for _ in range(frames):
        sim.Update(dt_frame)
    dt = time.perf_counter() - t0
    out = sim.download()
    sim.close()
    return n * frames / dt, dt, out


def cpu_baseline(n, R, fill, pos0, vel0, dt_frame):
    """Oracle timed on the host cores (checker + baseline): CPU_FRAMES frames of the bench workload on all threads, and
    the smaller configurations SURVEY.md 8d lists."""
    from oracle import sph_oracle as O
    from sphfluidsimulation_amd import presets
    O.build()
    threads = O.max_threads()
    rate, dt, out = oracle_run(O, n, R, fill, pos0, vel0, dt_frame, CPU_FRAMES, threads)
    base = dict(value=rate, unit="particle-steps/s", cores=threads, kind="port",
                sample=f"{CPU_FRAMES} frames (grid + density + 5 sub-steps each) of the same {n}-particle dam-break, "
                       f"{dt:.2f} s wall, OpenMP {threads} threads, gcc -O2 -ffp-contract=off")
    other = {}
    # (the small boxes on as many threads as they can feed: 256 OpenMP threads on 4 096 particles only wait for each other)
    for name, nn, frames, thr in (("C0_4096", 4096, 20, min(threads, 4)), ("C1_262144", 262144, 5, min(threads, 32)),
                                  ("C1_262144_1_thread", 262144, 1, 1)):
        RR = presets.scene_bucket_resolution(nn)
        p0, v0 = presets.dam_break(nn, fill, 1)
        r, d, _ = oracle_run(O, nn, RR, fill, p0, v0, presets.bench_dt_frame(RR), frames, thr)
        other[name] = {"value": r, "unit": "particle-steps/s", "cores": thr, "frames": frames, "wall_s": d}
    base["other_configs"] = other
    return base, out


def committed_csv(suffix, kernel_substr, tolerance=False):
    """Row of the newest committed profiles/*<suffix> for the kernel (rocprofv3 --pmc runs of this command at 16 777 216).
    The list kernels are templates whose last argument says which arithmetic they were compiled for (true: tolerance mode)."""
    import csv
    import glob
    import re
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*" + suffix)), reverse=True):   # newest file that has the kernel
        for row in csv.DictReader(open(f)):
            name = row["kernel"]
            if not (kernel_substr + "<" in name or kernel_substr + "(" in name):  # (k_velpos_list, not k_velpos_list_changers)
                continue
            m = re.search(re.escape(kernel_substr) + r"<([^>]*)>", name)
            fast = bool(m) and "true" in [x.strip() for x in m.group(1).split(",")]   # (the only bool among the template arguments)
            if fast == tolerance:
                return row, os.path.relpath(f, ROOT) + " (16 777 216 particles)"
    return None, None


def pmc_traffic(kernel_substr, tolerance=False):
    """HBM bytes per launch of the dominant kernel (FETCH_SIZE and WRITE_SIZE in separate passes, KiB).  FETCH_SIZE is
    doubled as MI355X_MICROARCH.md prescribes for wide streaming reads (calibrated on k_upload_split in
    profiles/README.md), so this is an upper bound for the 4-byte-per-lane part of the kernel's reads."""
    row, src = committed_csv("_hbm_traffic_pmc.csv", kernel_substr, tolerance)
    if row is None:
        return None, None
    return (2.0 * float(row["avg_FETCH_SIZE_raw"]) + float(row["avg_WRITE_SIZE_raw"])) * 1024.0, src


def pmc_valu(kernel_substr, tolerance=False):
    """Executed wave-level vector instructions per launch of the dominant kernel (SQ_INSTS_VALU, per-launch median)."""
    row, src = committed_csv("_sq_pmc.csv", kernel_substr, tolerance)
    if row is None:
        return None, None, None
    return float(row["SQ_INSTS_VALU"]), float(row.get("SQ_INSTS_VALU_TRANS_F32", 0) or 0), src


def spawn_ranks(a):
    """python bench.py --gpus N started plainly: N rank processes (children of a parent that never touches a GPU)."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out0, _ = procs[0].communicate()
    rcs = [p.wait() for p in procs]
    sys.stdout.write(out0 or "")
    sys.stdout.flush()
    sys.exit(max(rcs) if all(rc >= 0 for rc in rcs) else 1)


def supervise(rank, world):
    """N > 1: the process the launcher started never touches a GPU.  It runs the measurement in a child process and keeps
    watch: the halo transport over RCCL has never run on more than one GPU (the development box has one), and a collective
    that hangs cannot be recovered from inside the process that hangs.  If any rank's child fails or exceeds its time limit,
    EVERY rank starts a second child with the library's host transport (halo messages staged through the host, moved by
    gloo) -- slower, but a number; the line says which transport produced it (config.decomposition)."""
    import socket
    # rank 0 owes the launcher ONE JSON line on stdout: gloo and RCCL print banners there ("[Gloo] Rank 0 is connected ...",
    # "RCCL version ..."), from this process and from the worker -- everything but the line goes to stderr
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)      # control plane of the supervisors (CPU only)
    forced = os.environ.get("SPH_DIST_TRANSPORT")
    attempts = [(forced, 1200)] if forced else [("auto", int(os.environ.get("SPH_BENCH_RCCL_LIMIT_S", "420"))), ("host", 1200)]
    rc, out = 1, ""
    for transport, limit in attempts:
        port = [0]
        if rank == 0:                                                   # a fresh rendezvous port for the workers
            sk = socket.socket()
            sk.bind(("127.0.0.1", 0))
            port[0] = sk.getsockname()[1]
            sk.close()
        dist.broadcast_object_list(port, src=0)
        # (the workers rendezvous among themselves on their own port: the launcher's agent store -- TORCHELASTIC_USE_AGENT_STORE
        # -- only listens on the port it gave the supervisors)
        env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_")}
        env.update(SPH_BENCH_WORKER="1", SPH_DIST_TRANSPORT=transport, MASTER_PORT=str(port[0]))
        child = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                 stdout=subprocess.PIPE if rank == 0 else subprocess.DEVNULL, text=True)
        try:
            out, _ = child.communicate(timeout=limit)
            rc = child.returncode
        except subprocess.TimeoutExpired:
            child.kill()                                                # (this child, by its handle)
            out, _ = child.communicate()
            rc = -9
            sys.stderr.write(f"bench.py: rank {rank}: the worker with transport '{transport}' exceeded {limit} s and was stopped\n")
        ok = rc == 0 and (rank != 0 or '"metric"' in (out or ""))
        flag = torch.tensor([1 if ok else 0])
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            if rank == 0:
                lines = [l for l in (out or "").splitlines() if l.startswith("{") and '"metric"' in l]
                sys.stderr.write("".join(l + "\n" for l in (out or "").splitlines() if l not in lines[-1:]))
                os.write(real_stdout, (lines[-1] + "\n").encode())
            dist.destroy_process_group()
            sys.exit(0)
        if rank == 0:
            sys.stderr.write(f"bench.py: the run with transport '{transport}' failed on some rank (rank 0: rc {rc})\n")
    dist.destroy_process_group()
    sys.exit(rc if rc > 0 else 1)


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(a)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        sys.exit(f"bench.py --gpus {a.gpus} inside a torch.distributed.run world of {world} ranks")
    if world > 1 and os.environ.get("SPH_BENCH_WORKER") != "1":
        supervise(rank, world)

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
    pos0, vel0 = presets.dam_break(n, fill, a.seed)
    sim = SphFluidSimulation(particleNumber=n, bucketResolution=R, damFillRate=fill, viscosity=0.01,
                             restDensity=1.5, gasConstant=150.0, stiffnessCoefficient=5000.0,
                             dampingCoefficient=10.0, device_id=0, profile=True, precision=a.precision,
                             exact_order=a.mode == "exact", tile_lists=not a.global_lists, list_skin=a.list_skin).Start(pos0, vel0)

    # GPU result after CPU_FRAMES frames, checked against the CPU baseline run below; then start over for the timing
    warm = a.warmup
    parity_state = None
    if not a.no_cpu_baseline:
        for _ in range(CPU_FRAMES):
            sim.Update(dt_frame)
        parity_state = sim.download()
        sim.upload(pos0, vel0)
    sim.step_n(dt_frame, warm)
    sim.synchronize()
    sim.reset_stats()

    t0 = time.perf_counter()
    sim.step_n(dt_frame, a.steps)
    sim.synchronize()
    elapsed = time.perf_counter() - t0
    frames_done = warm + a.steps

    st = sim.stats()
    # SURVEY.md 8d asks for median and mean: a few more frames, each synchronised and timed on its own (never part of `value`)
    singles = []
    for _ in range(min(a.steps, 10)):
        sim.synchronize()
        ts = time.perf_counter()
        sim.Update(dt_frame)
        sim.synchronize()
        singles.append((time.perf_counter() - ts) * 1e3)
    frames_done += len(singles)
    # row f3, the render feed after the path (UpdateMeshProperties, never part of `value`): 112 B per particle
    sim.UpdateMeshProperties()
    sim.synchronize()
    tm = time.perf_counter()
    for _ in range(20):
        sim.UpdateMeshProperties()
    sim.synchronize()
    mesh_ms = (time.perf_counter() - tm) * 1e3 / 20
    extras = {}
    tol_state = None
    if not a.no_extras:
        # the dam collapses: cells fill up, the calm boxes the neighbour lists live on shrink (never part of `value`)
        late_to = 300
        if frames_done < late_to - 20:
            sim.step_n(dt_frame, late_to - 20 - frames_done)
            sim.synchronize()
            tl = time.perf_counter()
            sim.step_n(dt_frame, 20)
            sim.synchronize()
            late = time.perf_counter() - tl
            ls = sim.stats()
            extras["late_phase"] = {"frames": f"{late_to - 19}-{late_to}", "ms_per_step": late * 1e3 / 20,
                                    "particle_steps_per_s": n * 20 / late, "max_cell_occupancy": ls["max_cell_occupancy"],
                                    "queued_targets_last_sweep": ls["queued_targets"]}
        # S = 1 (one sub-step per grid build, what conventional SPH codes do): from the initial state again
        sim.upload(pos0, vel0)
        sim.substeps = 1
        sim.set_params()
        sim.step_n(dt_frame, warm)
        sim.synchronize()
        t1 = time.perf_counter()
        sim.step_n(dt_frame, a.steps)
        sim.synchronize()
        e1 = time.perf_counter() - t1
        extras["s1_variant"] = {"substeps": 1, "ms_per_step": e1 * 1e3 / a.steps, "particle_steps_per_s": n * a.steps / e1}
        sim.substeps = 5
        sim.set_params()
        if a.mode == "exact" and a.precision == 0:
            # the tolerance mode (SphParams.exact_order = 0; parity bar 1e-4, BASELINE.json north_star), same workload, same
            # protocol: its own engine, warm-up, then the same number of frames (never part of `value`)
            sim.OnDestroy()
            tol = SphFluidSimulation(particleNumber=n, bucketResolution=R, damFillRate=fill, viscosity=0.01, restDensity=1.5,
                                     gasConstant=150.0, stiffnessCoefficient=5000.0, dampingCoefficient=10.0, device_id=0,
                                     profile=True, exact_order=False).Start(pos0, vel0)
            tol_state = None
            if not a.no_cpu_baseline:
                for _ in range(CPU_FRAMES):
                    tol.Update(dt_frame)
                tol_state = tol.download()
                tol.upload(pos0, vel0)
            tol.step_n(dt_frame, warm)
            tol.synchronize()
            tol.reset_stats()
            tt = time.perf_counter()
            tol.step_n(dt_frame, a.steps)
            tol.synchronize()
            et = time.perf_counter() - tt
            ts = tol.stats()
            extras["tolerance_mode"] = {
                "mode": "SphParams.exact_order = 0: v_rcp_f32 / v_sqrt_f32 instead of IEEE division / sqrt, lists of LDS tile indices",
                "ms_per_step": et * 1e3 / a.steps, "particle_steps_per_s": n * a.steps / et, "parity_bar": "1e-4",
                "kernel": "k_velpos_tile", "kernel_ms": ts["ms_velpos_list"] / max(ts["launches_velpos_list"], 1),
                "passes_ms": {"grid": ts["ms_grid"] / max(ts["launches_grid"], 1), "density": ts["ms_density"] / max(ts["launches_density"], 1),
                              "velpos": ts["ms_velpos"] / max(ts["launches_velpos"], 1)}}
            tol.OnDestroy()
            sim = SphFluidSimulation(particleNumber=n, bucketResolution=R, damFillRate=fill, viscosity=0.01, restDensity=1.5,
                                     gasConstant=150.0, stiffnessCoefficient=5000.0, dampingCoefficient=10.0, device_id=0,
                                     profile=True, precision=a.precision).Start(pos0, vel0)
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
    # lists of LDS tile indices (k_velpos_tile); --global-lists: lists of global slots (k_velpos_list)
    kernel = ("k_velpos_list" if a.global_lists else "k_velpos_tile") if listed else "k_velpos_brick"
    kernel_ms = st["ms_velpos_list"] / st["launches_velpos_list"] if listed else vel_ms
    bytes_substep = BYTES_PER_SUBSTEP if a.precision == 0 else 40     # (6 + 12 + 4) R + (6 + 12) W with fp16 positions
    bytes_step = (32 + 52 * 5) if a.precision == 0 else (20 + 40 * 5)
    achieved = bytes_substep * n / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    traffic, traffic_src = pmc_traffic(kernel, a.mode == "tolerance") if n == 16777216 else (None, None)
    insts, trans, insts_src = pmc_valu(kernel, a.mode == "tolerance") if n == 16777216 else (None, None, None)
    # transcendental instructions (v_rcp, v_sqrt) hold the issue port four times as long as the rest
    issue_frac = ((insts + 3.0 * trans) / (kernel_ms * 1e-3) / VALU_ISSUE_PEAK) if insts and kernel_ms > 0 else None
    hbm_frac = achieved / HBM_PEAK_GBS
    # executed pair work: list entries walked (5 sub-steps) + candidates scanned by the density pass
    entries_per_frame = st["list_entries"] / max(a.steps, 1)
    exec_flops = (entries_per_frame * FLOPS_FORCE + st["candidates_density"] * FLOPS_DENSITY) / (ms_per_step * 1e-3) if listed else \
        value * pbar * (FLOPS_DENSITY + FLOPS_FORCE * 5)
    out = {
        "metric": "particle-steps/s", "value": value, "unit": "particle-steps/s", "n_gpus": 1,
        "steps": a.steps, "warmup": warm, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": workload, "particles": n, "bucket_resolution": R, "substeps": 5,
                   "mode": "exact_order" if a.mode == "exact" else "tolerance 1e-4 (v_rcp_f32 / v_sqrt_f32)", "positions": "fp32" if a.precision == 0 else "fp16 offsets from the cell centre (stored), fp32 arithmetic"},
        "roofline": {
            "bound": "valu" if (issue_frac or 0.0) > hbm_frac else "hbm", "kernel": kernel, "achieved": achieved,
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_frac, "traffic": traffic, "traffic_source": traffic_src,
            "algorithmic_bytes_per_launch": bytes_substep * n,
            "kernel_ms": kernel_ms, "launches": st["launches_velpos_list"] if listed else st["launches_velpos"],
            "step_hbm_frac": value * bytes_step / (HBM_PEAK_GBS * 1e9),
            "valu_issue_frac": issue_frac, "valu_insts_per_launch": insts, "valu_source": insts_src,
            "valu_frac_executed_pair_flops": exec_flops / 1e12 / VALU_PEAK_TFLOPS,
            "valu_frac_nominal_candidates": value * pbar * (FLOPS_DENSITY + FLOPS_FORCE * 5) / 1e12 / VALU_PEAK_TFLOPS,
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
    out.update(extras)
    if not a.no_cpu_baseline:
        base, (po, vo, ro) = cpu_baseline(n, R, fill, pos0, vel0, dt_frame)
        out["cpu_baseline"] = base
        out["parity_frames"] = CPU_FRAMES
        out["parity_bar"] = "bit-identical (0.0)" if a.precision == 0 and a.mode == "exact" else "1e-4"
        out["parity_max_abs_diff"] = {"pos": float(np.abs(parity_state[0] - po).max()), "vel": float(np.abs(parity_state[1] - vo).max()),
                                      "rho": float(np.abs(parity_state[2] - ro).max())}
        if "tolerance_mode" in out and tol_state is not None:
            vs = max(1.0, float(np.abs(vo[:, :3]).max()))
            ev = np.abs(tol_state[1][:, :3].astype(np.float64) - vo[:, :3]).max(1) / vs
            out["tolerance_mode"]["parity_frames"] = CPU_FRAMES
            out["tolerance_mode"]["parity_vs_oracle"] = {
                "pos_max_abs": float(np.abs(tol_state[0][:, :3] - po[:, :3]).max()), "vel_max_rel": float(ev.max()),
                "vel_p99.99_rel": float(np.percentile(ev, 99.99)), "particles_beyond_1e-4": int((ev > 1e-4).sum()),
                "rho_max_rel": float((np.abs(tol_state[2].astype(np.float64) - ro) / np.maximum(np.abs(ro), 1e-6)).max())}
    sim.OnDestroy()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
