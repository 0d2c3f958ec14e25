"""z-slab domain decomposition of the SPH frame update: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on MI355X; "gloo" in tests) for the neighbour exchange.

New work -- the reference is single-GPU.  The particle set is split along z,
the slowest axis of the bucket row index ``cx + cy*R + cz*R^2`` (Assets/Resources/Bucket.compute:28):
rank r owns the grid planes ``cz in [z_lo, z_hi)`` BY BUCKET ROW and holds ghost copies of
``halo_w`` planes on either side.  The reference freezes bucket and density for the 5 sub-steps of
a frame (Assets/Scripts/SphFluidSimulation.cs:98-102), so per frame there is

    1 migration exchange  (particles whose bucket row left the slab; variable size)
    1 ghost export        (boundary planes, full state; variable size)
    1 ghost rho refresh   (after the density pass; fixed size)
    S-1 ghost pos/vel refreshes (after every sub-step but the last; fixed size)

all point-to-point with the two z-neighbours (no bulk collective: xGMI is point-to-point and a
rank only ever needs its two faces).  With ``halo_w = 2`` an owned particle may drift one plane
out of its slab inside a frame and still find all 27 cells; ``stats()['halo_violations']`` counts
targets that went further (the result is then no longer identical to the single-GPU run).

The device work is libsph_hip.so's ``sph_slab_*`` entry points (include/sph_hip.h); this module only
sequences them and moves their buffers.
"""
from __future__ import annotations

import ctypes as C
import time

import numpy as np

from . import _native as N
from . import presets

RECORD = N.SPH_SLAB_RECORD_WORDS


# ---- partition (host side, numpy) -----------------------------------------------------------------

def slab_planes(R: int, world: int):
    """[z_lo, z_hi) per rank: R grid planes split as evenly as possible."""
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


# ---- neighbour transport --------------------------------------------------------------------------

class NeighbourTransport:
    """Point-to-point exchange with the left (rank-1) and right (rank+1) z-neighbour.

    Tensors are handed to the process group directly when it can take them (nccl + device tensors,
    gloo + host tensors) and staged through host memory otherwise (gloo + device tensors: the
    single-GPU rehearsal used by the tests)."""

    def __init__(self, rank: int, world: int, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank, self.world, self.group = rank, world, group
        self.left = rank - 1 if rank > 0 else None
        self.right = rank + 1 if rank < world - 1 else None
        self.backend = dist.get_backend(group) if world > 1 else "none"

    def _direct(self, t) -> bool:
        return (self.backend == "nccl") == t.is_cuda

    def exchange(self, send_left, send_right, n_recv_left: int, n_recv_right: int, like):
        """Send 1-D tensors to the neighbours (None / empty = nothing) and receive n_recv_* elements
        from each; returns (recv_left, recv_right) on ``like``'s device."""
        torch, dist = self.torch, self.dist
        recv = [None, None]
        ops, staged = [], []
        for side, peer, snd, nrecv in ((0, self.left, send_left, n_recv_left), (1, self.right, send_right, n_recv_right)):
            if peer is None:
                continue
            if snd is not None and snd.numel() > 0:
                buf = snd.contiguous() if self._direct(snd) else snd.detach().cpu().contiguous()
                ops.append(dist.P2POp(dist.isend, buf, peer, group=self.group))
            if nrecv > 0:
                if self._direct(like):
                    r = torch.empty(nrecv, dtype=like.dtype, device=like.device)
                else:
                    r = torch.empty(nrecv, dtype=like.dtype)
                    staged.append(side)
                recv[side] = r
                ops.append(dist.P2POp(dist.irecv, r, peer, group=self.group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        for side in staged:
            recv[side] = recv[side].to(like.device)
        for side in (0, 1):
            if recv[side] is None:
                recv[side] = torch.empty(0, dtype=like.dtype, device=like.device)
        return recv[0], recv[1]

    def exchange_counts(self, to_left: int, to_right: int):
        """Tell each neighbour how many elements it is about to receive; returns (from_left, from_right)."""
        torch = self.torch
        dev = "cuda" if self.backend == "nccl" else "cpu"
        like = torch.empty(0, dtype=torch.int64, device=dev)
        sl = torch.tensor([to_left], dtype=torch.int64, device=dev) if self.left is not None else None
        sr = torch.tensor([to_right], dtype=torch.int64, device=dev) if self.right is not None else None
        rl, rr = self.exchange(sl, sr, 1 if self.left is not None else 0, 1 if self.right is not None else 0, like)
        return (int(rl[0]) if rl.numel() else 0), (int(rr[0]) if rr.numel() else 0)

    def all_reduce(self, value: float, op: str = "max") -> float:
        if self.world == 1:
            return value
        torch, dist = self.torch, self.dist
        dev = "cuda" if self.backend == "nccl" else "cpu"
        t = torch.tensor([value], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op={"max": dist.ReduceOp.MAX, "sum": dist.ReduceOp.SUM}[op], group=self.group)
        return float(t[0])

    def barrier(self):
        if self.world > 1:
            self.dist.barrier(group=self.group)


# ---- the per-rank simulation ------------------------------------------------------------------------

class SlabSimulation:
    """Same surface as simulation.SphFluidSimulation (Start / Update / download), one z-slab per rank."""

    def __init__(self, rank: int, world: int, particleNumber: int, bucketResolution: int, damFillRate: float = 0.5,
                 viscosity: float = 0.01, restDensity: float = 1.5, gasConstant: float = 150.0,
                 stiffnessCoefficient: float = 5000.0, dampingCoefficient: float = 10.0, device_id: int = 0,
                 halo_w: int = 2, capacity_factor: float = 1.6, profile: bool = False, group=None):
        import torch
        self.torch = torch
        self.rank, self.world = rank, world
        self.halo_w = halo_w
        self.capacity_factor = capacity_factor
        self.transport = NeighbourTransport(rank, world, group)
        p = N.default_params()
        p.num_particles = int(particleNumber)
        p.bucket_resolution = int(bucketResolution)
        p.dam_fill_rate = damFillRate
        p.viscosity, p.rest_density, p.gas_constant = viscosity, restDensity, gasConstant
        p.stiffness_coeff, p.damping_coeff = stiffnessCoefficient, dampingCoefficient
        p.device_id = device_id
        p.flags = N.SPH_FLAG_PROFILE if profile else 0
        self._params = p
        self.R = int(bucketResolution)
        self.z_lo, self.z_hi = slab_planes(self.R, world)[rank]
        self.device = torch.device("cuda", device_id)
        self._ctx = C.c_void_p()
        self.frames = 0

    def _check(self, rc):
        N.check(rc, self._ctx)

    # -- Start(): distribute the initial particle set ---------------------------------------------------
    def Start(self, positions: np.ndarray, velocities: np.ndarray | None = None):
        """Every rank passes the same global float4 arrays (id order) and keeps the particles whose
        bucket row lies in its planes."""
        torch = self.torch
        pos = np.ascontiguousarray(positions, np.float32).reshape(-1, 4)
        owner = owner_of_plane(bucket_plane(pos, self.R), self.R, self.world)
        ids = np.nonzero(owner == self.rank)[0].astype(np.uint32)
        n_global = pos.shape[0]
        planes = max(self.z_hi - self.z_lo, 1)
        share = n_global / self.world * (planes + 2 * self.halo_w) / planes
        self.capacity = int(min(max(share * self.capacity_factor, len(ids) * self.capacity_factor, 4096), n_global + 64))
        self._check(N.lib().sph_slab_create(C.byref(self._params), self.capacity, C.byref(self._ctx)))
        self._check(N.lib().sph_slab_configure(self._ctx, self.z_lo, self.z_hi, self.halo_w,
                                               int(self.rank > 0), int(self.rank < self.world - 1)))
        torch.cuda.set_device(self.device)
        self._check(N.lib().sph_set_stream(self._ctx, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        my_pos = np.ascontiguousarray(pos[ids])
        my_vel = None
        if velocities is not None:
            my_vel = np.ascontiguousarray(np.asarray(velocities, np.float32).reshape(-1, 4)[ids])
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else None
        self._check(N.lib().sph_slab_upload(self._ctx, len(ids), fp(my_pos), fp(my_vel),
                                            ids.ctypes.data_as(C.POINTER(C.c_uint32))))
        words = self.capacity * RECORD
        self._send = [torch.empty(words, dtype=torch.float32, device=self.device) for _ in range(2)]
        self.substeps = int(self._params.substeps)
        self.dt_divisor = float(self._params.dt_divisor)
        return self

    # -- one Update(): SphFluidSimulation.cs:96-102 with the halo exchange woven in ----------------------
    def Update(self, deltaTime: float):
        L, T = N.lib(), self.transport
        sl, sr = self._send
        p = lambda t: C.c_void_p(t.data_ptr())
        # 1. migration
        c3 = (C.c_uint32 * 3)()
        self._check(L.sph_slab_begin_frame(self._ctx, p(sl), p(sr), self.capacity, c3))
        to_l, to_r = int(c3[1]), int(c3[2])
        in_l, in_r = T.exchange_counts(to_l, to_r)
        rl, rr = T.exchange(sl[:to_l * RECORD], sr[:to_r * RECORD], in_l * RECORD, in_r * RECORD, sl)
        self._check(L.sph_slab_append(self._ctx, p(rl), in_l, p(rr), in_r, 0))
        # 2. ghost planes
        c2 = (C.c_uint32 * 2)()
        self._check(L.sph_slab_export_ghosts(self._ctx, p(sl), p(sr), self.capacity, c2))
        ex_l, ex_r = int(c2[0]), int(c2[1])
        gh_l, gh_r = T.exchange_counts(ex_l, ex_r)
        rl, rr = T.exchange(sl[:ex_l * RECORD], sr[:ex_r * RECORD], gh_l * RECORD, gh_r * RECORD, sl)
        self._check(L.sph_slab_append(self._ctx, p(rl), gh_l, p(rr), gh_r, 1))
        # 3. BucketGeneration + DensityCalculation (:98-99), then the neighbours' rho for the ghosts
        self._check(L.sph_bucket_generation(self._ctx))
        self._check(L.sph_density_calculation(self._ctx))
        self._refresh(N.SPH_SLAB_UPDATE_RHO, 1, ex_l, ex_r, gh_l, gh_r)
        # 4. sub-steps (:101-102); ghosts follow their owners after each but the last
        dt = float(np.float32(deltaTime) / np.float32(self.dt_divisor))
        for i in range(self.substeps):
            self._check(L.sph_update_velocity_and_position(self._ctx, dt))
            if i + 1 < self.substeps:
                self._refresh(N.SPH_SLAB_UPDATE_POSVEL, N.SPH_SLAB_POSVEL_WORDS, ex_l, ex_r, gh_l, gh_r)
        self.frames += 1

    def _refresh(self, what, words, ex_l, ex_r, gh_l, gh_r):
        L, T = N.lib(), self.transport
        sl, sr = self._send
        p = lambda t: C.c_void_p(t.data_ptr())
        self._check(L.sph_slab_pack_update(self._ctx, what, p(sl), p(sr)))
        rl, rr = T.exchange(sl[:ex_l * words], sr[:ex_r * words], gh_l * words, gh_r * words, sl)
        self._check(L.sph_slab_unpack_update(self._ctx, what, p(rl), p(rr)))

    def step_n(self, deltaTime: float, frames: int):
        for _ in range(frames):
            self.Update(deltaTime)

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
        """Full id-ordered arrays on rank dst (None elsewhere)."""
        local = self.download_local()
        if self.world == 1:
            parts = [local]
        else:
            parts = [None] * self.world if self.rank == dst else None
            self.transport.dist.gather_object(local, parts, dst=dst, group=self.transport.group)
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
        s = N.SphStats()
        self._check(N.lib().sph_get_stats(self._ctx, C.byref(s)))
        d = s.as_dict()
        c = (C.c_uint32 * 5)()
        self._check(N.lib().sph_slab_counts(self._ctx, c))
        d.update(owned=int(c[0]), ghosts_left=int(c[1]), ghosts_right=int(c[2]), exported_left=int(c[3]),
                 exported_right=int(c[4]), capacity=self.capacity, planes=(self.z_lo, self.z_hi))
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
    """Strong scaling: the SAME n-particle dam-break split over `world` GPUs (one rank each)."""
    import os
    import torch
    import torch.distributed as dist
    # rehearsal hooks (tests / one-GPU boxes): SPH_DIST_BACKEND=gloo stages the halo through the host,
    # SPH_DIST_SINGLE_DEVICE=1 puts every rank on cuda:0.  The driver's runs use neither.
    backend = os.environ.get("SPH_DIST_BACKEND", "nccl")
    if os.environ.get("SPH_DIST_SINGLE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if not dist.is_initialized() and world == 1:      # --force-slab diagnostic without a launcher
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        dist.init_process_group("gloo", rank=0, world_size=1)
    if not dist.is_initialized():
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    pos0, vel0 = presets.dam_break(n, fill, a.seed)
    sim = SlabSimulation(rank, world, n, R, fill, 0.01, 1.5, 150.0, 5000.0, 10.0, device_id=local_rank,
                         halo_w=2, profile=True).Start(pos0, vel0)
    del pos0, vel0
    warm = max(a.warmup, 1)
    sim.step_n(dt_frame, warm)
    sim.synchronize()
    sim.reset_stats()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sim.step_n(dt_frame, a.steps)
    torch.cuda.synchronize()
    dist.barrier()
    elapsed = sim.transport.all_reduce(time.perf_counter() - t0, "max")
    st = sim.stats()
    viol = sim.transport.all_reduce(float(st["halo_violations"]), "sum")
    owned_max = sim.transport.all_reduce(float(st["owned"]), "max")
    ghosts = sim.transport.all_reduce(float(st["ghosts_left"] + st["ghosts_right"]), "sum")
    vel_ms = st["ms_velpos"] / max(st["launches_velpos"], 1)
    listed = st.get("launches_velpos_list", 0) > 0
    kernel = "k_velpos_list" if listed else "k_velpos_brick"
    kernel_ms = st["ms_velpos_list"] / st["launches_velpos_list"] if listed else vel_ms
    value = n * a.steps / elapsed
    out = {
        "metric": "particle-steps/s", "value": value, "unit": "particle-steps/s", "n_gpus": world, "steps": a.steps,
        "warmup": warm, "ms_per_step": elapsed * 1e3 / a.steps, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "", "particles": n, "bucket_resolution": R, "substeps": 5, "mode": "exact_order",
                   "decomposition": f"{world} z-slabs, halo {sim.halo_w} planes, p2p ghost exchange over "
                                    f"torch.distributed/{dist.get_backend()}"},
        "roofline": {"bound": "hbm", "kernel": kernel + " (rank 0)",
                     "achieved": 52.0 * st["owned"] / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0, "peak": 8000.0,
                     "unit": "GB/s", "frac": (52.0 * st["owned"] / (kernel_ms * 1e-3) / 1e9 / 8000.0) if kernel_ms > 0 else 0.0,
                     "traffic": None, "kernel_ms": kernel_ms},
        "neighbour_lists_rank0": {"listed_fraction_of_target_substeps":
                                  st.get("listed_target_substeps", 0) / max(st["owned"] * st["launches_velpos"], 1)},
        "halo_violations": viol, "max_owned_per_rank": owned_max, "ghost_particles_total": ghosts,
        "passes_ms_rank0": {"grid": st["ms_grid"] / max(st["launches_grid"], 1),
                            "density": st["ms_density"] / max(st["launches_density"], 1), "velpos": vel_ms},
    }
    sim.OnDestroy()
    dist.barrier()
    return out
