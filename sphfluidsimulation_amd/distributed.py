"""z-slab domain decomposition of the SPH frame update, one process per GPU.

New work -- the reference is single-GPU.  The particle set is split along z,
the slowest axis of the bucket row index ``cx + cy*R + cz*R^2`` (Assets/Resources/Bucket.compute:28):
rank r owns the grid planes ``cz in [z_lo, z_hi)`` BY BUCKET ROW and holds ghost copies of the halo
planes on either side.  The frame itself -- classify / pack, the exchange with the two z-neighbours,
BucketGeneration, DensityCalculation, the ghost refreshes between the sub-steps, the re-run with a
wider halo when a particle outran it -- happens INSIDE libsph_hip.so (``sph_slab_step``,
csrc/sph_slab_frame.hip).  This module only

  * decides the initial ownership (host side, numpy: the same arithmetic as Bucket.compute:26-29),
  * gives the library its transport: RCCL (``ncclSend/ncclRecv`` over xGMI from inside the library;
    the 128-byte communicator id travels through torch.distributed) or, for rehearsals with several
    ranks on ONE GPU, a host transport: the library stages its messages through pinned memory and
    calls back into ``HostExchange``, which moves them with torch.distributed (gloo),
  * gathers results by particle id.

``torch.distributed`` is the control plane (rendezvous, id broadcast, barriers, timing reductions).
"""
from __future__ import annotations

import ctypes as C
import time

import numpy as np

from . import _native as N
from . import presets


# ---- partition (host side, numpy) -----------------------------------------------------------------

def slab_planes(R: int, world: int):
    """[z_lo, z_hi) per rank: R grid planes split as evenly as possible (= sph_slab_planes)."""
    return [((r * R) // world, ((r + 1) * R) // world) for r in range(world)]


def bucket_plane(pos4: np.ndarray, R: int) -> np.ndarray:
    """cz of every particle's BUCKET ROW exactly as Bucket.compute:26-29 computes it (float index
    arithmetic, uint conversion, *32 wrap, bounds drop); -1 for rows outside the bucket."""
    f32 = np.float32
    p = np.asarray(pos4, np.float32)
    with np.errstate(invalid="ignore", over="ignore"):
        def to_int(v):
            t = np.trunc(v.astype(np.float64))
            t = np.where(np.isnan(t), 0.0, np.clip(t, -2147483648.0, 2147483647.0))
            return t.astype(np.float32)
        ix, iy, iz = (to_int(p[:, a] * f32(R - 1)) for a in range(3))
        fidx = (ix + iy * f32(R)) + iz * f32(R) * f32(R)
        u = np.where(fidx > 0, np.minimum(np.trunc(fidx.astype(np.float64)), 4294967295.0), 0.0).astype(np.uint64)
    idx = (u * np.uint64(32)) & np.uint64(0xFFFFFFFF)
    ncell = np.uint64(R) ** 3
    key = np.where(idx < ncell * np.uint64(32), idx // np.uint64(32), ncell)
    cz = (key // np.uint64(R * R)).astype(np.int64)
    return np.where(key < ncell, cz, -1)


def owner_of_plane(cz: np.ndarray, R: int, world: int) -> np.ndarray:
    """Rank owning each plane; particles outside the bucket (-1) stay with rank 0."""
    planes = slab_planes(R, world)
    bounds = np.array([hi for _, hi in planes])
    return np.where(cz < 0, 0, np.searchsorted(bounds, cz, side="right"))


# ---- host transport: what the library calls back into (rehearsal: several ranks on one GPU, gloo) -------------

class HostExchange:
    """Moves the library's HOST-staged messages between z-neighbours with torch.distributed
    (sph_slab_set_host_transport).  Every rank calls exchange / allreduce in the same order."""

    def __init__(self, rank: int, world: int, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank, self.world, self.group = rank, world, group
        self.left = rank - 1 if rank > 0 else None
        self.right = rank + 1 if rank < world - 1 else None
        self.calls = 0
        # the CFUNCTYPE objects must outlive the context that holds their addresses
        self.exchange_cb = N.HOST_EXCHANGE_FN(self._exchange)
        self.allreduce_cb = N.HOST_ALLREDUCE_FN(self._allreduce)

    def _tensor(self, ptr, nbytes):
        buf = (C.c_uint8 * nbytes).from_address(ptr)
        return self.torch.from_numpy(np.ctypeslib.as_array(buf))

    def _exchange(self, user, sl, nsl, sr, nsr, rl, nrl, rr, nrr):
        try:
            dist = self.dist
            ops = []
            for peer, sp, ns, rp, nr in ((self.left, sl, nsl, rl, nrl), (self.right, sr, nsr, rr, nrr)):
                if peer is None:
                    if ns or nr:
                        return 2
                    continue
                if ns:
                    ops.append(dist.P2POp(dist.isend, self._tensor(sp, ns), peer, group=self.group))
                if nr:
                    ops.append(dist.P2POp(dist.irecv, self._tensor(rp, nr), peer, group=self.group))
            if ops:
                for req in dist.batch_isend_irecv(ops):
                    req.wait()
            self.calls += 1
            return 0
        except Exception as e:  # never let an exception cross the C boundary
            print(f"HostExchange rank {self.rank}: {e!r}", flush=True)
            return 1

    def _allreduce(self, user, value):
        try:
            t = self.torch.tensor([int(value[0])], dtype=self.torch.int64)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
            value[0] = int(t[0])
            return 0
        except Exception as e:
            print(f"HostExchange rank {self.rank}: {e!r}", flush=True)
            return 1


class Control:
    """Control-plane reductions over torch.distributed (CPU tensors; world 1 needs no process group)."""

    def __init__(self, rank: int, world: int, group=None):
        self.rank, self.world, self.group = rank, world, group

    def all_reduce(self, value: float, op: str = "max") -> float:
        if self.world == 1:
            return value
        import torch
        import torch.distributed as dist
        t = torch.tensor([value], dtype=torch.float64)
        dist.all_reduce(t, op={"max": dist.ReduceOp.MAX, "sum": dist.ReduceOp.SUM, "min": dist.ReduceOp.MIN}[op], group=self.group)
        return float(t[0])

    def barrier(self):
        if self.world > 1:
            import torch.distributed as dist
            dist.barrier(group=self.group)

    def broadcast_bytes(self, data: bytes | None, nbytes: int, src: int = 0) -> bytes:
        if self.world == 1:
            return data
        import torch
        import torch.distributed as dist
        t = torch.zeros(nbytes, dtype=torch.uint8)
        if self.rank == src:
            t.copy_(torch.frombuffer(bytearray(data), dtype=torch.uint8))
        dist.broadcast(t, src=src, group=self.group)
        return bytes(t.numpy().tobytes())


# ---- the per-rank simulation ------------------------------------------------------------------------

class SlabSimulation:
    """Same surface as simulation.SphFluidSimulation (Start / Update / download), one z-slab per rank.

    transport = "rccl": the library talks to its neighbours itself (one GPU per rank);
                "host": HostExchange over the default torch.distributed group (gloo; ranks may share a GPU);
                "auto": rccl, falling back to host on EVERY rank if any rank could not set it up."""

    def __init__(self, rank: int, world: int, particleNumber: int, bucketResolution: int, damFillRate: float = 0.5,
                 viscosity: float = 0.01, restDensity: float = 1.5, gasConstant: float = 150.0,
                 stiffnessCoefficient: float = 5000.0, dampingCoefficient: float = 10.0, device_id: int = 0,
                 halo_w: int = 2, capacity_factor: float = 1.6, profile: bool = False, group=None,
                 transport: str = "auto", precision: int = 0, exact_order: bool = True):
        self.rank, self.world = rank, world
        self.halo_w = halo_w
        self.capacity_factor = capacity_factor
        self.control = Control(rank, world, group)
        self.group = group
        self.transport_request = transport
        self.transport = None
        self.host_exchange = None
        p = N.default_params()
        p.num_particles = int(particleNumber)
        p.bucket_resolution = int(bucketResolution)
        p.dam_fill_rate = damFillRate
        p.viscosity, p.rest_density, p.gas_constant = viscosity, restDensity, gasConstant
        p.stiffness_coeff, p.damping_coeff = stiffnessCoefficient, dampingCoefficient
        p.device_id = device_id
        p.halo_planes = int(halo_w)
        p.precision = int(precision)                  # 1: fp16 position records (BASELINE config 5), see include/sph_hip.h
        p.exact_order = 1 if exact_order else 0       # 0: tolerance mode (hardware reciprocal / sqrt, parity bar 1e-4)
        p.flags = (N.SPH_FLAG_PROFILE if profile else 0) | int(__import__('os').environ.get('SPH_SLAB_FLAGS', '0'))   # (diagnostic flags, e.g. 64 / 128)
        self._params = p
        self.R = int(bucketResolution)
        self.z_lo, self.z_hi = slab_planes(self.R, world)[rank]
        self._ctx = C.c_void_p()
        self.frames = 0

    def _check(self, rc):
        N.check(rc, self._ctx)

    def _setup_transport(self):
        L = N.lib()
        want = self.transport_request
        if self.world == 1:
            self.transport = "none"
            return
        if want in ("rccl", "auto"):
            # Three collective steps, each entered by EVERY rank whatever happened before (a rank that skipped one would
            # leave the others waiting): (1) can every rank load RCCL and make an id?  (2) rank 0's id to all;
            # (3) ncclCommInitRank, then: did every rank succeed?
            ident = C.create_string_buffer(N.SPH_COMM_ID_BYTES)
            have = 1.0 if L.sph_comm_unique_id(ident) == N.SPH_OK else 0.0
            if have == 0.0:
                print(f"rank {self.rank}: RCCL unavailable ({L.sph_last_error(None).decode()})", flush=True)
            ok = 0.0
            if self.control.all_reduce(have, "min") >= 1.0:
                raw = self.control.broadcast_bytes(ident.raw if self.rank == 0 else None, N.SPH_COMM_ID_BYTES)
                ident = C.create_string_buffer(raw, N.SPH_COMM_ID_BYTES)
                rc = L.sph_slab_comm_init(self._ctx, ident)
                ok = 1.0 if rc == N.SPH_OK else 0.0
                if ok == 0.0:
                    print(f"rank {self.rank}: ncclCommInitRank failed ({L.sph_last_error(self._ctx).decode()})", flush=True)
                if self.control.all_reduce(ok, "min") >= 1.0:
                    self.transport = "rccl"
                    return
            if want == "rccl":
                raise RuntimeError("RCCL transport could not be set up on every rank")
            if self.rank == 0:
                print("falling back to the host transport (gloo) on every rank", flush=True)
        self.host_exchange = HostExchange(self.rank, self.world, self.group)
        self._check(L.sph_slab_set_host_transport(self._ctx, self.host_exchange.exchange_cb, self.host_exchange.allreduce_cb, None))
        self.transport = "host"

    # -- Start(): distribute the initial particle set ---------------------------------------------------
    def Start(self, positions: np.ndarray, velocities: np.ndarray | None = None):
        """Every rank passes the same global float4 arrays (id order) and keeps the particles whose
        bucket row lies in its planes."""
        L = N.lib()
        pos = np.ascontiguousarray(positions, np.float32).reshape(-1, 4)
        cz = bucket_plane(pos, self.R)
        owner = owner_of_plane(cz, self.R, self.world)
        ids = np.nonzero(owner == self.rank)[0].astype(np.uint32)
        n_global = pos.shape[0]
        # capacity: owned + up to 8 ghost planes either side (the halo may grow), with headroom
        near = np.count_nonzero((cz >= self.z_lo - 8) & (cz < self.z_hi + 8) & (owner != self.rank))
        self.capacity = int(min(max((len(ids) + near) * self.capacity_factor, 4096), n_global + 64))
        self._check(L.sph_slab_create(C.byref(self._params), self.capacity, C.byref(self._ctx)))
        self._check(L.sph_slab_configure(self._ctx, self.rank, self.world))
        self._setup_transport()
        my_pos = np.ascontiguousarray(pos[ids])
        my_vel = None
        if velocities is not None:
            my_vel = np.ascontiguousarray(np.asarray(velocities, np.float32).reshape(-1, 4)[ids])
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else None
        self._check(L.sph_slab_upload(self._ctx, len(ids), fp(my_pos), fp(my_vel), ids.ctypes.data_as(C.POINTER(C.c_uint32))))
        self.substeps = int(self._params.substeps)
        return self

    # -- one Update(): SphFluidSimulation.cs:96-102, halo exchange included --------------------------------
    def Update(self, deltaTime: float):
        self._check(N.lib().sph_slab_step(self._ctx, float(np.float32(deltaTime))))
        self.frames += 1

    def step_n(self, deltaTime: float, frames: int):
        for _ in range(frames):
            self.Update(deltaTime)

    def finish(self):
        """Collective: validates the frame in flight (re-running it with a wider halo if needed)."""
        self._check(N.lib().sph_slab_finish(self._ctx))

    # -- results ------------------------------------------------------------------------------------------
    def download_local(self):
        cap = self.capacity
        ids = np.empty(cap, np.uint32)
        pos = np.empty((cap, 4), np.float32)
        vel = np.empty((cap, 4), np.float32)
        rho = np.empty(cap, np.float32)
        n = C.c_uint32()
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
        self._check(N.lib().sph_slab_download(self._ctx, cap, ids.ctypes.data_as(C.POINTER(C.c_uint32)), fp(pos),
                                              fp(vel), fp(rho), C.byref(n)))
        k = n.value
        return ids[:k].copy(), pos[:k].copy(), vel[:k].copy(), rho[:k].copy()

    def gather(self, n_global: int, dst: int = 0):
        """Full id-ordered arrays on rank dst (None elsewhere).  Collective."""
        local = self.download_local()
        if self.world == 1:
            parts = [local]
        else:
            import torch.distributed as dist
            parts = [None] * self.world if self.rank == dst else None
            dist.gather_object(local, parts, dst=dst, group=self.group)
            if self.rank != dst:
                return None
        pos = np.full((n_global, 4), np.nan, np.float32)
        vel = np.full((n_global, 4), np.nan, np.float32)
        rho = np.full(n_global, np.nan, np.float32)
        seen = np.zeros(n_global, np.int32)
        for ids, p, v, r in parts:
            pos[ids], vel[ids], rho[ids] = p, v, r
            np.add.at(seen, ids, 1)
        if not np.all(seen == 1):
            raise RuntimeError(f"particle ownership broken: {int((seen != 1).sum())} ids not held exactly once")
        return pos, vel, rho

    def stats(self) -> dict:
        """Collective (validates the frame in flight first)."""
        s = N.SphStats()
        self._check(N.lib().sph_get_stats(self._ctx, C.byref(s)))
        d = s.as_dict()
        c = (C.c_uint32 * 5)()
        self._check(N.lib().sph_slab_counts(self._ctx, c))
        d.update(owned=int(c[0]), ghosts_left=int(c[1]), ghosts_right=int(c[2]), exported_left=int(c[3]),
                 exported_right=int(c[4]), capacity=self.capacity, planes=(self.z_lo, self.z_hi), transport=self.transport)
        return d

    def reset_stats(self):
        self._check(N.lib().sph_reset_stats(self._ctx))

    def synchronize(self):
        self._check(N.lib().sph_synchronize(self._ctx))

    def OnDestroy(self):
        if self._ctx:
            N.lib().sph_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.OnDestroy()
        except Exception:
            pass


# ---- bench.py --gpus N (N > 1) ----------------------------------------------------------------------------

def bench_main(a, n, R, fill, dt_frame, rank, world, local_rank):
    """Strong scaling: the SAME n-particle dam-break split over `world` GPUs (one rank each).
    Control plane: torch.distributed gloo on 127.0.0.1; halo: RCCL from inside the library."""
    import os
    import torch.distributed as dist
    # rehearsal hooks (tests / one-GPU boxes): SPH_DIST_TRANSPORT=host stages the halo through the host (gloo),
    # SPH_DIST_SINGLE_DEVICE=1 puts every rank on device 0.  The driver's runs use neither.
    transport = os.environ.get("SPH_DIST_TRANSPORT", "auto")
    if os.environ.get("SPH_DIST_SINGLE_DEVICE") == "1":
        local_rank = 0
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")   # one node: RCCL's bootstrap needs no external interface
    if not dist.is_initialized() and world > 1:
        import datetime
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=300))
    pos0, vel0 = presets.dam_break(n, fill, a.seed)
    sim = SlabSimulation(rank, world, n, R, fill, 0.01, 1.5, 150.0, 5000.0, 10.0, device_id=local_rank,
                         halo_w=2, profile=True, transport=transport, precision=getattr(a, "precision", 0),
                         exact_order=getattr(a, "mode", "exact") == "exact").Start(pos0, vel0)
    del pos0, vel0
    ctl = sim.control
    warm = max(a.warmup, 1)
    sim.step_n(dt_frame, warm)
    sim.finish()
    sim.synchronize()
    sim.reset_stats()
    ctl.barrier()
    t0 = time.perf_counter()
    sim.step_n(dt_frame, a.steps)
    sim.finish()
    sim.synchronize()
    ctl.barrier()
    elapsed = ctl.all_reduce(time.perf_counter() - t0, "max")
    st = sim.stats()
    viol = ctl.all_reduce(float(st["halo_violations"]), "sum")
    owned_max = ctl.all_reduce(float(st["owned"]), "max")
    ghosts = ctl.all_reduce(float(st["ghosts_left"] + st["ghosts_right"]), "sum")
    substeps = max(a.steps * sim.substeps, 1)            # (a sub-step is issued in two passes when its ghost refresh is overlapped)
    vel_ms = st["ms_velpos"] / substeps
    listed = st.get("launches_velpos_list", 0) > 0
    kernel = "k_velpos_list" if listed else "k_velpos_brick"
    kernel_ms = st["ms_velpos_list"] / substeps if listed else vel_ms
    value = n * a.steps / elapsed
    achieved = 52.0 * st["owned"] / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    out = {
        "metric": "particle-steps/s", "value": value, "unit": "particle-steps/s", "n_gpus": world, "steps": a.steps,
        "warmup": warm, "ms_per_step": elapsed * 1e3 / a.steps, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "", "particles": n, "bucket_resolution": R, "substeps": 5,
                   "mode": "exact_order" if getattr(a, "mode", "exact") == "exact" else "tolerance 1e-4 (v_rcp_f32 / v_sqrt_f32)",
                   "decomposition": f"{world} z-slabs, halo {st['halo_planes']} planes, in-library neighbour exchange "
                                    f"({sim.transport}), ghost refreshes overlapped with the interior targets of each sub-step, "
                                    f"torch.distributed/gloo control plane"},
        "roofline": {"bound": "hbm", "kernel": kernel + " (rank 0)", "achieved": achieved, "peak": 8000.0,
                     "unit": "GB/s", "frac": achieved / 8000.0, "traffic": None, "kernel_ms": kernel_ms},
        "neighbour_lists_rank0": {"listed_fraction_of_target_substeps":
                                  st.get("listed_target_substeps", 0) / max(st["owned"] * substeps, 1)},
        "halo_violations": viol, "frames_rerun": st["frames_rerun"], "host_syncs_per_frame_rank0": st["host_syncs"] / max(a.steps, 1),
        "max_owned_per_rank": owned_max, "ghost_particles_total": ghosts,
        "passes_ms_rank0": {"grid": st["ms_grid"] / max(st["launches_grid"], 1),
                            "density": st["ms_density"] / max(st["launches_density"], 1), "velpos": vel_ms},
    }
    sim.OnDestroy()
    ctl.barrier()
    return out
