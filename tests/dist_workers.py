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
    """CPU only: the host transport the library calls back into (HostExchange), with ragged and empty messages, driven
    through the same C callback signatures the library uses."""
    import ctypes as C
    from sphfluidsimulation_amd.distributed import HostExchange, Control
    dist = _init(rank, world, port)
    hx = HostExchange(rank, world)
    ok = True

    def buf(nbytes, fill):
        a = np.full(max(nbytes, 1), fill, np.uint8)
        return a, a.ctypes.data

    for rnd in range(3):
        to_l = ((rank * 7 + rnd * 3) % 5) * 40 if rank > 0 else 0          # includes zero-length messages
        to_r = ((rank * 5 + rnd) % 4) * 40 if rank < world - 1 else 0
        in_l = (((rank - 1) * 5 + rnd) % 4) * 40 if rank > 0 else 0
        in_r = (((rank + 1) * 7 + rnd * 3) % 5) * 40 if rank < world - 1 else 0
        sl, psl = buf(to_l, (rank * 10 + rnd) % 251)
        sr, psr = buf(to_r, (rank * 10 + rnd + 5) % 251)
        rl, prl = buf(in_l, 255)
        rr, prr = buf(in_r, 255)
        rc = hx.exchange_cb(None, psl, to_l, psr, to_r, prl, in_l, prr, in_r)   # through the CFUNCTYPE thunk
        ok &= rc == 0
        if in_l:
            ok &= bool((rl[:in_l] == ((rank - 1) * 10 + rnd + 5) % 251).all())
        if in_r:
            ok &= bool((rr[:in_r] == ((rank + 1) * 10 + rnd) % 251).all())
    v = C.c_ulonglong(rank + 1)
    ok &= hx.allreduce_cb(None, C.byref(v)) == 0 and v.value == world * (world + 1) // 2
    ctl = Control(rank, world)
    ok &= ctl.all_reduce(float(rank), "max") == float(world - 1)
    ok &= ctl.all_reduce(1.0, "sum") == float(world)
    ident = bytes(range(128)) if rank == 0 else None
    ok &= ctl.broadcast_bytes(ident, 128) == bytes(range(128))
    ctl.barrier()
    open(os.path.join(out_dir, f"ok_{rank}"), "w").write("1" if ok else "0")
    dist.destroy_process_group()


def slab_worker(rank, world, port, out_dir, n, R, frames, speed, halo_w, seed, precision=0):
    """GPU: `world` ranks share cuda:0; the library stages its halo messages through the host and gloo moves them
    (sph_slab_set_host_transport); rank 0 writes the gathered state."""
    from sphfluidsimulation_amd import presets
    from sphfluidsimulation_amd.distributed import SlabSimulation
    dist = _init(rank, world, port)
    fill = 0.8
    pos0, vel0 = presets.dam_break(n, fill, seed)
    if speed:
        vel0[:, :3] = np.random.default_rng(5).normal(0, speed, (vel0.shape[0], 3)).astype(np.float32)
    sim = SlabSimulation(rank, world, n, R, fill, 0.01, 1.5, 150.0, 5000.0, 10.0, device_id=0, halo_w=halo_w,
                         transport="host", precision=precision).Start(pos0, vel0)
    dt_frame = presets.bench_dt_frame(R)
    states = {}
    full0 = sim.gather(pos0.shape[0])                    # before any step: the upload round trip
    if rank == 0:
        states["pos_f0"], states["vel_f0"], states["rho_f0"] = full0
    for f in range(1, frames + 1):
        sim.Update(dt_frame)
        full = sim.gather(pos0.shape[0])
        if rank == 0:
            states[f"pos_f{f}"], states[f"vel_f{f}"], states[f"rho_f{f}"] = full
    st = sim.stats()
    viol = sim.control.all_reduce(float(st["halo_violations"]), "sum")
    migrated = sim.control.all_reduce(float(st["owned"]), "sum")
    if rank == 0:
        np.savez(os.path.join(out_dir, "slab.npz"), violations=viol, owned_total=migrated, reruns=st["frames_rerun"],
                 halo=st["halo_planes"], ghosts=st["ghosts_left"] + st["ghosts_right"], **states)
    sim.OnDestroy()
    dist.barrier()
    dist.destroy_process_group()
