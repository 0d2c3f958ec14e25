"""ctypes front-end of the CPU oracle (oracle/sph_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of sph_oracle.c.  Importable only from
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  PARITY UNPINNED by
the reference (it ships no tests or golden vectors and cannot be built here); pinned by
the analytic KATs in tests/test_oracle_kat.py and by oracle/sph_oracle_py.py.

The Python surface mirrors the reference driver's method names
(Assets/Scripts/SphFluidSimulation.cs): BucketGeneration :203, DensityCalculation :223,
UpdateVelocityAndPosition :243, Update :96.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "build", "libsph_oracle.so")


class OracleParams(C.Structure):
    _fields_ = [
        ("num_particles", C.c_uint32),
        ("res", C.c_uint32),
        ("bucket_resolution", C.c_uint32),
        ("particle_mass", C.c_float),
        ("h", C.c_float),
        ("h2", C.c_float),
        ("h6", C.c_float),
        ("h9", C.c_float),
        ("viscosity", C.c_float),
        ("gas_const", C.c_float),
        ("rest_density", C.c_float),
        ("stiffness", C.c_float),
        ("damping", C.c_float),
        ("substeps", C.c_uint32),
        ("dt_divisor", C.c_float),
        ("num_threads", C.c_int32),
    ]


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile). Returns the library path."""
    src = os.path.join(_HERE, "sph_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        fp = C.POINTER(C.c_float)
        up = C.POINTER(C.c_uint32)
        pp = C.POINTER(OracleParams)
        L.oracle_derive.argtypes = [C.c_uint32, C.c_uint32] + [C.c_float] * 6 + [pp]
        L.oracle_derive.restype = C.c_int
        L.oracle_bucket_len.argtypes = [pp]
        L.oracle_bucket_len.restype = C.c_uint64
        L.oracle_max_threads.restype = C.c_int
        L.oracle_clear_bucket.argtypes = [pp, up]
        L.oracle_bucket.argtypes = [pp, fp, up]
        L.oracle_clear_float.argtypes = [pp, fp]
        L.oracle_density.argtypes = [pp, fp, up, fp]
        L.oracle_velpos.argtypes = [pp, fp, fp, fp, up, fp, fp, C.c_float]
        L.oracle_sim_create.argtypes = [pp]
        L.oracle_sim_create.restype = C.c_void_p
        L.oracle_sim_destroy.argtypes = [C.c_void_p]
        L.oracle_sim_set_params.argtypes = [C.c_void_p, pp]
        L.oracle_sim_upload.argtypes = [C.c_void_p, fp, fp]
        L.oracle_sim_download.argtypes = [C.c_void_p, fp, fp, fp]
        L.oracle_sim_bucket_generation.argtypes = [C.c_void_p]
        L.oracle_sim_density_calculation.argtypes = [C.c_void_p]
        L.oracle_sim_update_velocity_and_position.argtypes = [C.c_void_p, C.c_float]
        L.oracle_sim_update.argtypes = [C.c_void_p, C.c_float]
        L.oracle_sim_bucket.argtypes = [C.c_void_p]
        L.oracle_sim_bucket.restype = up
        L.oracle_snoise4.argtypes = [C.c_float] * 4
        L.oracle_snoise4.restype = C.c_float
        L.oracle_init_particles.argtypes = [C.c_int, C.c_uint32, C.c_float, C.c_uint32, C.c_int, fp]
        L.oracle_init_particles.restype = C.c_int
        L.oracle_update_mesh_properties.argtypes = [C.c_uint32, fp, fp, C.c_float, C.c_float, fp, fp, fp]
        L.oracle_update_mesh_properties.restype = None
        L.oracle_particle_per_dim.argtypes = [C.c_int, C.c_uint32, C.c_float]
        L.oracle_particle_per_dim.restype = C.c_uint32
        _lib = L
    return _lib


def _fp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _up(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def max_threads() -> int:
    return int(lib().oracle_max_threads())


def derive(particle_number: int, bucket_resolution: int, dam_fill_rate: float = 0.5,
           viscosity: float = 0.01, rest_density: float = 1.5, gas_constant: float = 150.0,
           stiffness: float = 5000.0, damping: float = 10.0, num_threads: int = 0) -> OracleParams:
    """Derived scalars exactly as SphFluidSimulation.cs:84-85,159,176,236-237,255.
    Defaults are the inspector defaults SphFluidSimulation.cs:38-45."""
    p = OracleParams()
    rc = lib().oracle_derive(particle_number, bucket_resolution, dam_fill_rate, viscosity,
                             rest_density, gas_constant, stiffness, damping, C.byref(p))
    if rc != 0:
        raise ValueError("oracle_derive: invalid arguments")
    p.num_threads = num_threads
    return p


class OracleSimulation:
    """Headless CPU re-execution of SphFluidSimulation.Update() (reference :96-102)."""

    def __init__(self, params: OracleParams):
        self.p = params
        self._h = lib().oracle_sim_create(C.byref(self.p))
        if not self._h:
            raise MemoryError("oracle_sim_create failed")
        self.count = int(params.res) * int(params.res)

    def close(self):
        if self._h:
            lib().oracle_sim_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def set_params(self, params: OracleParams):
        self.p = params
        lib().oracle_sim_set_params(self._h, C.byref(self.p))

    def upload(self, pos4: np.ndarray, vel4: np.ndarray):
        pos4 = np.ascontiguousarray(pos4, dtype=np.float32).reshape(self.count, 4)
        vel4 = np.ascontiguousarray(vel4, dtype=np.float32).reshape(self.count, 4)
        lib().oracle_sim_upload(self._h, _fp(pos4), _fp(vel4))

    def download(self):
        pos = np.empty((self.count, 4), np.float32)
        vel = np.empty((self.count, 4), np.float32)
        rho = np.empty((self.count,), np.float32)
        lib().oracle_sim_download(self._h, _fp(pos), _fp(vel), _fp(rho))
        return pos, vel, rho

    def bucket(self) -> np.ndarray:
        n = int(lib().oracle_bucket_len(C.byref(self.p)))
        ptr = lib().oracle_sim_bucket(self._h)
        return np.ctypeslib.as_array(ptr, shape=(n,)).reshape(-1, 32).copy()

    # names follow the reference driver
    def BucketGeneration(self):
        lib().oracle_sim_bucket_generation(self._h)

    def DensityCalculation(self):
        lib().oracle_sim_density_calculation(self._h)

    def UpdateVelocityAndPosition(self, dt: float):
        lib().oracle_sim_update_velocity_and_position(self._h, float(np.float32(dt)))

    def Update(self, delta_time: float):
        lib().oracle_sim_update(self._h, float(np.float32(delta_time)))


# ---- stateless passes (for per-pass parity tests) --------------------------------------

def bucket_generation(p: OracleParams, pos4: np.ndarray) -> np.ndarray:
    n = int(lib().oracle_bucket_len(C.byref(p)))
    b = np.empty(n, np.uint32)
    pos4 = np.ascontiguousarray(pos4, np.float32)
    lib().oracle_clear_bucket(C.byref(p), _up(b))
    lib().oracle_bucket(C.byref(p), _fp(pos4), _up(b))
    return b


def density(p: OracleParams, pos4: np.ndarray, bucket: np.ndarray) -> np.ndarray:
    cnt = int(p.res) * int(p.res)
    rho = np.zeros(cnt, np.float32)
    pos4 = np.ascontiguousarray(pos4, np.float32)
    lib().oracle_density(C.byref(p), _fp(pos4), _up(bucket), _fp(rho))
    return rho


def velpos(p: OracleParams, pos4, vel4, rho, bucket, dt: float):
    cnt = int(p.res) * int(p.res)
    pos4 = np.ascontiguousarray(pos4, np.float32)
    vel4 = np.ascontiguousarray(vel4, np.float32)
    rho = np.ascontiguousarray(rho, np.float32)
    po = np.empty((cnt, 4), np.float32)
    vo = np.empty((cnt, 4), np.float32)
    lib().oracle_velpos(C.byref(p), _fp(pos4), _fp(vel4), _fp(rho), _up(bucket), _fp(po), _fp(vo),
                        float(np.float32(dt)))
    return po, vo


# ---- row f1: InitParticles.compute presets + noiseSimplex.cginc::snoise(float4) -------------------------

def particle_per_dim(preset: int, res: int, fill: float) -> int:
    """InitParticles.compute:20 (Preset1) / :53,:86 (Preset2/3): uint(ceil(pow(count / fill[^2], 1/3))), the one
    transcendental of the generator, evaluated once in fp32 here for oracle AND engine."""
    return int(lib().oracle_particle_per_dim(preset, res, float(np.float32(fill))))


def snoise4(x, y=None, z=None, w=None):
    """noiseSimplex.cginc::snoise(float4): four scalars -> float, or an (n, 4) array -> (n,) float32."""
    f = lib().oracle_snoise4
    if y is None:
        v = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, 4)
        return np.array([f(float(a), float(b), float(c), float(d)) for a, b, c, d in v], dtype=np.float32)
    return float(f(float(np.float32(x)), float(np.float32(y)), float(np.float32(z)), float(np.float32(w))))


def init_particles(preset: int, res: int, fill: float, noise: bool = True) -> np.ndarray:
    """preset = kernel index as in SphFluidSimulation.cs:184 (0 Preset1 dam break, 1 Preset2, 2 Preset3)."""
    pos = np.empty((res * res, 4), np.float32)
    rc = lib().oracle_init_particles(preset, res, float(np.float32(fill)), particle_per_dim(preset, res, fill),
                                     1 if noise else 0, _fp(pos))
    if rc != 0:
        raise ValueError("oracle_init_particles: invalid arguments")
    return pos


# ---- row f3: UpdateMeshProperties.compute ---------------------------------------------------------------

def update_mesh_properties(pos4: np.ndarray, vel4: np.ndarray, low_speed: float, high_speed: float,
                           particle_scale, sim_trs=None) -> np.ndarray:
    """UpdateMeshProperties.compute:18-64.  sim_trs: 4x4 in (row, column) notation (default identity).
    Returns (count, 20) float32: column-major mat, then color."""
    pos4 = np.ascontiguousarray(pos4, dtype=np.float32).reshape(-1, 4)
    vel4 = np.ascontiguousarray(vel4, dtype=np.float32).reshape(-1, 4)
    scale = np.ascontiguousarray(np.broadcast_to(np.asarray(particle_scale, np.float32), (3,)))
    trs = np.eye(4, dtype=np.float32) if sim_trs is None else np.asarray(sim_trs, np.float32).reshape(4, 4)
    trs_cm = np.ascontiguousarray(trs.T).reshape(-1)
    out = np.empty((pos4.shape[0], 20), np.float32)
    lib().oracle_update_mesh_properties(pos4.shape[0], _fp(pos4), _fp(vel4), float(np.float32(low_speed)),
                                        float(np.float32(high_speed)), _fp(scale), _fp(trs_cm), _fp(out))
    return out
