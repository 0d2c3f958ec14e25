"""Host-side mirror of the reference driver, Assets/Scripts/SphFluidSimulation.cs, over the C ABI.

Same public field names (inspector fields :35-45), same method names and meaning
(Start :82, Update :96, OnDestroy :110, BucketGeneration :203, DensityCalculation :223,
UpdateVelocityAndPosition :243), so tests read like the reference driver.  Every method goes
through libsph_hip.so (include/sph_hip.h); there is no CPU path here.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native as N
from . import presets


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else None


class SphFluidSimulation:
    # constants, SphFluidSimulation.cs:8-9
    NumThreads = 32
    MaxParticlesPerVoxel = 32

    def __init__(self, preset: int = 1, particleNumber: int = 1024, bucketResolution: int = 256,
                 damFillRate: float = 0.5, viscosity: float = 0.01, restDensity: float = 1.5,
                 gasConstant: float = 150.0, stiffnessCoefficient: float = 5000.0,
                 dampingCoefficient: float = 10.0, device_id: int = 0, profile: bool = False,
                 tiling: bool = True, lists: bool = True, graphs: bool = True, particleRadius: float = 0.01, lowSpeed: float = 0.0,
                 highSpeed: float = 0.5, substeps: int = 5, device_ids=None, halo_planes: int = 0, list_skin: float = 0.0,
                 peer_copies: bool = False, precision: int = 0, overlap: bool = True, exact_order: bool = True, tile_lists: bool = True, tile_slots: int = 0, chunk_boxes: bool = False, side_stream: bool = True):
        # inspector fields and defaults, SphFluidSimulation.cs:35-45
        self.preset = preset
        self.particleNumber = particleNumber
        self.bucketResolution = bucketResolution
        self.damFillRate = damFillRate
        self.viscosity = viscosity
        self.restDensity = restDensity
        self.gasConstant = gasConstant
        self.stiffnessCoefficient = stiffnessCoefficient
        self.dampingCoefficient = dampingCoefficient
        # rendering fields :49-53 (values of SampleScene.unity:372-376), consumed by UpdateMeshProperties()
        self.particleRadius = particleRadius
        self.lowSpeed = lowSpeed
        self.highSpeed = highSpeed
        self.device_id = device_id
        self.profile = profile
        self.tiling = tiling            # False: diagnostic per-target kernels (SPH_FLAG_NO_TILING)
        self.lists = lists              # False: every sub-step re-scans its 27 bins (SPH_FLAG_NO_LISTS); "always":
                                        # record lists every frame instead of only while they pay (SPH_FLAG_LISTS_ALWAYS)
        self.graphs = graphs            # False: no HIP-graph replay of whole frames (SPH_FLAG_NO_GRAPHS)
        self.substeps = substeps        # 5, SphFluidSimulation.cs:101
        self.device_ids = list(device_ids) if device_ids else None   # 2..8 ordinals: z-slabs on several GPUs of this process
        self.halo_planes = halo_planes  # ghost planes per slab face to start with (0 = 2)
        self.list_skin = list_skin      # neighbour-list skin / h (0 = 0.04)
        self.precision = precision      # 1: positions stored as fp16 offsets from the cell centre (config C4)
        self.side_stream = side_stream  # False: the scanning kernel in stream order (SPH_FLAG_NO_SIDE_STREAM, diagnostic)
        self.chunk_boxes = chunk_boxes  # True: displacement ranges from k_chunk_box (SPH_FLAG_CHUNK_BOXES, diagnostic)
        self.tile_slots = tile_slots    # 1 / 2: force the k_velpos_tile instantiation (SPH_FLAG_TILE_SLOTS_*, diagnostic); 0: the engine picks
        self.tile_lists = tile_lists    # False: lists of global slots gathered through L1 by k_velpos_list (SPH_FLAG_GLOBAL_LISTS, diagnostic)
        self.exact_order = exact_order  # False: tolerance mode (SphParams.exact_order = 0): hardware reciprocal / sqrt, parity bar 1e-4
        self.overlap = overlap          # False: multi-device, refresh ghosts after the whole sub-step (SPH_FLAG_NO_OVERLAP)
        self.peer_copies = peer_copies  # multi-device: peer-to-peer copies instead of RCCL (SPH_FLAG_PEER_COPIES)
        self._ctx = C.c_void_p()
        self._derived = None

    # ---- parameter plumbing -------------------------------------------------------------------
    def _params(self) -> N.SphParams:
        p = N.default_params()
        p.num_particles = int(self.particleNumber)
        p.bucket_resolution = int(self.bucketResolution)
        p.dam_fill_rate = self.damFillRate
        p.viscosity = self.viscosity
        p.rest_density = self.restDensity
        p.gas_constant = self.gasConstant
        p.stiffness_coeff = self.stiffnessCoefficient
        p.damping_coeff = self.dampingCoefficient
        p.device_id = self.device_id
        p.substeps = int(self.substeps)
        p.precision = int(self.precision)
        p.exact_order = 1 if self.exact_order else 0
        p.halo_planes = int(self.halo_planes)
        p.list_skin = float(self.list_skin)
        if self.device_ids:
            p.num_devices = len(self.device_ids)
            for k, dev in enumerate(self.device_ids):
                p.device_ids[k] = int(dev)
        p.flags = ((N.SPH_FLAG_PROFILE if self.profile else 0) | (0 if self.tiling else N.SPH_FLAG_NO_TILING) |
                   (0 if self.lists else N.SPH_FLAG_NO_LISTS) | (0 if self.graphs else N.SPH_FLAG_NO_GRAPHS) |
                   (N.SPH_FLAG_LISTS_ALWAYS if self.lists == "always" else 0) | (N.SPH_FLAG_PEER_COPIES if self.peer_copies else 0) | (0 if self.overlap else N.SPH_FLAG_NO_OVERLAP) |
                   (0 if self.tile_lists else N.SPH_FLAG_GLOBAL_LISTS) |
                   {0: 0, 1: N.SPH_FLAG_TILE_SLOTS_ONE, 2: N.SPH_FLAG_TILE_SLOTS_TWO}[int(self.tile_slots)] |
                   (N.SPH_FLAG_CHUNK_BOXES if self.chunk_boxes else 0) | (0 if self.side_stream else N.SPH_FLAG_NO_SIDE_STREAM))
        return p

    def _check(self, rc):
        N.check(rc, self._ctx)

    # ---- Unity functions ----------------------------------------------------------------------
    def Start(self, positions: np.ndarray | None = None, velocities: np.ndarray | None = None,
              seed: int = 1):
        """SphFluidSimulation.cs:82-94.  The reference fills the position texture with
        InitParticles.compute (kernel index = preset) and clears the velocity texture; here the
        caller passes the float4 arrays, gets the synthetic dam-break of presets.dam_break (Preset1 lattice,
        hash jitter, `seed`), or -- with seed=None -- the reference's own generator on the device
        (sph_init_particles: kernel index = self.preset, simplex-noise jitter)."""
        p = self._params()
        self._check(N.lib().sph_create(C.byref(p), C.byref(self._ctx)))
        d = N.SphDerived()
        self._check(N.lib().sph_get_derived(self._ctx, C.byref(d)))
        self._derived = d
        self.particleNumber = d.num_particles                    # :84
        self._particleTextureResolution = d.particle_resolution  # :85
        self._effectiveRadius = d.effective_radius               # :159
        self._particleMass = d.particle_mass                     # :176
        if positions is None and seed is None:
            # exactly like the reference: InitParticles.compute kernel #preset + cleared velocities (:179-189)
            self._check(N.lib().sph_init_particles(self._ctx, int(self.preset)))
            return self
        if positions is None:
            positions, velocities = presets.dam_break(self.particleNumber, self.damFillRate, seed)
        self.upload(positions, velocities)
        return self

    def Update(self, deltaTime: float):
        """SphFluidSimulation.cs:96-102 (render hooks :104-107 are out of scope)."""
        self._check(N.lib().sph_step(self._ctx, float(np.float32(deltaTime))))

    def OnDestroy(self):
        """SphFluidSimulation.cs:110-120."""
        if self._ctx:
            N.lib().sph_destroy(self._ctx)
            self._ctx = C.c_void_p()

    close = OnDestroy

    def __del__(self):
        try:
            self.OnDestroy()
        except Exception:
            pass

    # ---- Update region ------------------------------------------------------------------------
    def BucketGeneration(self):
        self._check(N.lib().sph_bucket_generation(self._ctx))

    def DensityCalculation(self):
        self._check(N.lib().sph_density_calculation(self._ctx))

    def UpdateVelocityAndPosition(self, dt: float):
        self._check(N.lib().sph_update_velocity_and_position(self._ctx, float(np.float32(dt))))

    # ---- data in / out (the reference never reads back; the drop-in must) ---------------------
    @property
    def count(self) -> int:
        return int(self._derived.particle_count)

    @property
    def derived(self) -> N.SphDerived:
        return self._derived

    def upload(self, positions: np.ndarray, velocities: np.ndarray | None = None):
        pos = np.ascontiguousarray(positions, np.float32).reshape(-1, 4)
        if pos.shape[0] != self.count:
            raise ValueError(f"positions must hold {self.count} float4, got {pos.shape[0]}")
        vel = None
        if velocities is not None:
            vel = np.ascontiguousarray(velocities, np.float32).reshape(-1, 4)
            if vel.shape[0] != self.count:
                raise ValueError(f"velocities must hold {self.count} float4, got {vel.shape[0]}")
        self._check(N.lib().sph_upload(self._ctx, _fp(pos), _fp(vel)))

    def download(self):
        pos = np.empty((self.count, 4), np.float32)
        vel = np.empty((self.count, 4), np.float32)
        rho = np.empty((self.count,), np.float32)
        self._check(N.lib().sph_download(self._ctx, _fp(pos), _fp(vel), _fp(rho)))
        return pos, vel, rho

    def UpdateMeshProperties(self, simTRS=None):
        """SphFluidSimulation.cs:271-284 (row f3).  simTRS = transform.localToWorldMatrix as a 4x4 array in math
        (row, column) notation, default identity; uses self.particleRadius / lowSpeed / highSpeed (:49-53)."""
        m = N.SphMeshParams()
        m.struct_size = C.sizeof(N.SphMeshParams)
        m.low_speed = float(np.float32(self.lowSpeed))
        m.high_speed = float(np.float32(self.highSpeed))
        for k in range(3):
            m.particle_scale[k] = float(np.float32(self.particleRadius))      # :280
        trs = np.eye(4, dtype=np.float32) if simTRS is None else np.asarray(simTRS, np.float32).reshape(4, 4)
        for k, val in enumerate(trs.T.reshape(-1)):                            # column-major like Matrix4x4
            m.sim_trs[k] = float(val)
        self._check(N.lib().sph_update_mesh_properties(self._ctx, C.byref(m)))

    def download_mesh_properties(self) -> np.ndarray:
        """(count, 20) float32: the _particleMeshPropertiesBuffer (:174) contents, record = texel id."""
        props = np.empty((self.count, N.SPH_MESH_RECORD_FLOATS), np.float32)
        self._check(N.lib().sph_download_mesh_properties(self._ctx, _fp(props)))
        return props

    def step_n(self, deltaTime: float, frames: int):
        self._check(N.lib().sph_step_n(self._ctx, float(np.float32(deltaTime)), int(frames)))

    def set_params(self):
        """Push changed inspector scalars (viscosity .. dampingCoefficient) mid-run."""
        p = self._params()
        self._check(N.lib().sph_set_params(self._ctx, C.byref(p)))

    def synchronize(self):
        self._check(N.lib().sph_synchronize(self._ctx))

    def stats(self) -> dict:
        s = N.SphStats()
        self._check(N.lib().sph_get_stats(self._ctx, C.byref(s)))
        return s.as_dict()

    def reset_stats(self):
        self._check(N.lib().sph_reset_stats(self._ctx))

    def set_stream(self, hip_stream: int):
        self._check(N.lib().sph_set_stream(self._ctx, C.c_void_p(hip_stream)))
