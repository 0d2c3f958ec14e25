"""Worker bodies for the multi-process tests (spawned with torch.multiprocessing, gloo rendezvous on
127.0.0.1).  Kept importable as a module so 'spawn' can pickle the entry points."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _init(rank, world, port):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    return dist


def transport_worker(rank, world, port, out_dir):
    """CPU only: the neighbour transport with ragged sizes (what the slab driver runs per frame)."""
    import torch
    from sphfluidsimulation_amd.distributed import NeighbourTransport
    dist = _init(rank, world, port)
    t = NeighbourTransport(rank, world)
    ok = True
    for rnd in range(3):
        to_l = (rank * 7 + rnd * 3) % 5 if rank > 0 else 0          # includes zero-length messages
        to_r = (rank * 5 + rnd) % 4 if rank < world - 1 else 0
        in_l, in_r = t.exchange_counts(to_l, to_r)
        exp_l = ((rank - 1) * 5 + rnd) % 4 if rank > 0 else 0
        exp_r = ((rank + 1) * 7 + rnd * 3) % 5 if rank < world - 1 else 0
        ok &= (in_l, in_r) == (exp_l, exp_r)
        sl = torch.full((to_l * 9,), float(rank * 100 + rnd), dtype=torch.float32)
        sr = torch.full((to_r * 9,), float(rank * 100 + rnd + 50), dtype=torch.float32)
        rl, rr = t.exchange(sl, sr, in_l * 9, in_r * 9, sl)
        ok &= rl.numel() == in_l * 9 and rr.numel() == in_r * 9
        if in_l:
            ok &= bool((rl == float((rank - 1) * 100 + rnd + 50)).all())
        if in_r:
            ok &= bool((rr == float((rank + 1) * 100 + rnd)).all())
    ok &= t.all_reduce(float(rank), "max") == float(world - 1)
    ok &= t.all_reduce(1.0, "sum") == float(world)
    t.barrier()
    open(os.path.join(out_dir, f"ok_{rank}"), "w").write("1" if ok else "0")
    dist.destroy_process_group()


def slab_worker(rank, world, port, out_dir, n, R, frames, speed, halo_w, seed):
    """GPU: `world` ranks share cuda:0 (gloo, host-staged halo exchange); rank 0 writes the gathered state."""
    from sphfluidsimulation_amd import presets
    from sphfluidsimulation_amd.distributed import SlabSimulation
    dist = _init(rank, world, port)
    fill = 0.8
    pos0, vel0 = presets.dam_break(n, fill, seed)
    if speed:
        vel0[:, :3] = np.random.default_rng(5).normal(0, speed, (vel0.shape[0], 3)).astype(np.float32)
    sim = SlabSimulation(rank, world, n, R, fill, 0.01, 1.5, 150.0, 5000.0, 10.0, device_id=0, halo_w=halo_w).Start(pos0, vel0)
    dt_frame = presets.bench_dt_frame(R)
    states = {}
    full0 = sim.gather(pos0.shape[0])                    # before any step: the upload round trip
    if rank == 0:
        states["pos_f0"], states["vel_f0"], states["rho_f0"] = full0
    viol = 0.0
    for f in range(1, frames + 1):
        sim.Update(dt_frame)
        full = sim.gather(pos0.shape[0])
        st = sim.stats()
        viol = sim.transport.all_reduce(float(st["halo_violations"]), "sum")
        if rank == 0:
            states[f"pos_f{f}"], states[f"vel_f{f}"], states[f"rho_f{f}"] = full
    st = sim.stats()
    migrated = sim.transport.all_reduce(float(st["owned"]), "sum")
    if rank == 0:
        np.savez(os.path.join(out_dir, "slab.npz"), violations=viol, owned_total=migrated,
                 ghosts=st["ghosts_left"] + st["ghosts_right"], **states)
    sim.OnDestroy()
    dist.barrier()
    dist.destroy_process_group()
