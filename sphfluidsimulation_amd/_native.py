"""ctypes binding of include/sph_hip.h (libsph_hip.so).  Fails loudly when the HIP library is
missing or a call fails -- there is no CPU fallback behind this module."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SPH_HIP_LIBRARY") or os.path.join(_HERE, "lib", "libsph_hip.so")   # (override: A/B runs of two builds)

SPH_OK = 0
SPH_E_INVALID_ARG = -1
SPH_E_BAD_STATE = -2
SPH_E_OOM = -3
SPH_E_HIP = -4
SPH_E_UNSUPPORTED = -5
SPH_E_COMM = -6
SPH_FLAG_PROFILE = 1
SPH_FLAG_NO_TILING = 2
SPH_FLAG_NO_LISTS = 4
SPH_FLAG_NO_GRAPHS = 8
SPH_FLAG_LISTS_ALWAYS = 16
SPH_FLAG_PEER_COPIES = 32
SPH_FLAG_NO_OVERLAP = 64
SPH_FLAG_FORCE_TWO_PASS = 128
SPH_FLAG_GLOBAL_LISTS = 256
SPH_FLAG_TILE_SLOTS_ONE = 512
SPH_FLAG_TILE_SLOTS_TWO = 1024
SPH_FLAG_CHUNK_BOXES = 2048
SPH_FLAG_NO_SIDE_STREAM = 4096
SPH_COMM_ID_BYTES = 128
SPH_SLAB_UPDATE_RHO = 0
SPH_SLAB_UPDATE_POSVEL = 1

_ERR_NAMES = {
    SPH_E_INVALID_ARG: "SPH_E_INVALID_ARG", SPH_E_BAD_STATE: "SPH_E_BAD_STATE", SPH_E_OOM: "SPH_E_OOM",
    SPH_E_HIP: "SPH_E_HIP", SPH_E_UNSUPPORTED: "SPH_E_UNSUPPORTED", SPH_E_COMM: "SPH_E_COMM",
}


class SphError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"{_ERR_NAMES.get(code, code)}: {message}")
        self.code = code


class SphParams(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("num_particles", C.c_uint32),
        ("bucket_resolution", C.c_uint32),
        ("dam_fill_rate", C.c_float),
        ("viscosity", C.c_float),
        ("rest_density", C.c_float),
        ("gas_constant", C.c_float),
        ("stiffness_coeff", C.c_float),
        ("damping_coeff", C.c_float),
        ("max_per_voxel", C.c_uint32),
        ("substeps", C.c_uint32),
        ("dt_divisor", C.c_float),
        ("gravity", C.c_float * 3),
        ("precision", C.c_uint32),
        ("exact_order", C.c_uint32),
        ("device_id", C.c_int32),
        ("flags", C.c_uint32),
        # ABI version 2
        ("num_devices", C.c_uint32),
        ("device_ids", C.c_int32 * 8),
        ("halo_planes", C.c_uint32),
        ("list_skin", C.c_float),
    ]


class SphDerived(C.Structure):
    _fields_ = [
        ("num_particles", C.c_uint32),
        ("particle_resolution", C.c_uint32),
        ("particle_count", C.c_uint32),
        ("bucket_resolution", C.c_uint32),
        ("particle_mass", C.c_float),
        ("effective_radius", C.c_float),
        ("effective_radius2", C.c_float),
        ("effective_radius6", C.c_float),
        ("effective_radius9", C.c_float),
    ]


class SphStats(C.Structure):
    _fields_ = [
        ("frames", C.c_uint64),
        ("substeps", C.c_uint64),
        ("ms_grid", C.c_double),
        ("ms_density", C.c_double),
        ("ms_velpos", C.c_double),
        ("launches_grid", C.c_uint64),
        ("launches_density", C.c_uint64),
        ("launches_velpos", C.c_uint64),
        ("max_cell_occupancy", C.c_uint32),
        ("overflowed_cells", C.c_uint32),
        ("dropped_particles", C.c_uint32),
        ("queued_targets", C.c_uint32),
        ("nan_resets", C.c_uint64),
        ("candidates_density", C.c_uint64),
        ("halo_violations", C.c_uint64),
        ("listed_target_substeps", C.c_uint64),
        ("scanned_brick_substeps", C.c_uint64),
        ("list_entries", C.c_uint64),
        ("ms_velpos_list", C.c_double),
        ("launches_velpos_list", C.c_uint64),
        ("graph_replays", C.c_uint64),
        ("frames_without_lists", C.c_uint64),
        ("host_syncs", C.c_uint64),
        ("frames_rerun", C.c_uint64),
        ("halo_planes", C.c_uint64),
        ("list_row_entries", C.c_uint64),
        ("lists_too_long", C.c_uint64),
    ]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class SphMeshParams(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("low_speed", C.c_float),
        ("high_speed", C.c_float),
        ("particle_scale", C.c_float * 3),
        ("sim_trs", C.c_float * 16),
    ]


SPH_MESH_RECORD_FLOATS = 20


class SphRuntimeInfo(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("hip_runtime_version", C.c_int32),
        ("hip_driver_version", C.c_int32),
        ("device_count", C.c_int32),
        ("rccl_version", C.c_int32),
        ("hip_library_path", C.c_char * 256),
        ("rccl_library_path", C.c_char * 256),
    ]


# host transport callbacks (sph_slab_set_host_transport)
HOST_EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                               C.c_void_p, C.c_size_t)
HOST_ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_ulonglong))

# every symbol include/sph_hip.h declares: name -> (restype, argtypes)
_FP = C.POINTER(C.c_float)
_CTX = C.c_void_p
SYMBOLS = {
    "sph_version": (C.c_int, []),
    "sph_params_default": (C.c_int, [C.POINTER(SphParams)]),
    "sph_create": (C.c_int, [C.POINTER(SphParams), C.POINTER(_CTX)]),
    "sph_destroy": (C.c_int, [_CTX]),
    "sph_upload": (C.c_int, [_CTX, _FP, _FP]),
    "sph_download": (C.c_int, [_CTX, _FP, _FP, _FP]),
    "sph_init_particles": (C.c_int, [_CTX, C.c_int]),
    "sph_update_mesh_properties": (C.c_int, [_CTX, C.POINTER(SphMeshParams)]),
    "sph_download_mesh_properties": (C.c_int, [_CTX, _FP]),
    "sph_mesh_properties_device_ptr": (C.c_int, [_CTX, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "sph_export_mesh_properties_fd": (C.c_int, [_CTX, C.POINTER(C.c_int), C.POINTER(C.c_size_t)]),
    "sph_update_state_textures": (C.c_int, [_CTX]),
    "sph_export_state_fd": (C.c_int, [_CTX, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_size_t)]),
    "sph_import_semaphore_fd": (C.c_int, [_CTX, C.c_int, C.c_int]),
    "sph_signal_semaphore": (C.c_int, [_CTX, C.c_uint64]),
    "sph_step": (C.c_int, [_CTX, C.c_float]),
    "sph_step_n": (C.c_int, [_CTX, C.c_float, C.c_int]),
    "sph_bucket_generation": (C.c_int, [_CTX]),
    "sph_density_calculation": (C.c_int, [_CTX]),
    "sph_update_velocity_and_position": (C.c_int, [_CTX, C.c_float]),
    "sph_set_params": (C.c_int, [_CTX, C.POINTER(SphParams)]),
    "sph_get_derived": (C.c_int, [_CTX, C.POINTER(SphDerived)]),
    "sph_get_stats": (C.c_int, [_CTX, C.POINTER(SphStats)]),
    "sph_reset_stats": (C.c_int, [_CTX]),
    "sph_synchronize": (C.c_int, [_CTX]),
    "sph_set_stream": (C.c_int, [_CTX, C.c_void_p]),
    "sph_runtime_info": (C.c_int, [C.POINTER(SphRuntimeInfo)]),
    "sph_selftest_arithmetic": (C.c_int, [C.c_int, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]),
    "sph_slab_create": (C.c_int, [C.POINTER(SphParams), C.c_uint32, C.POINTER(_CTX)]),
    "sph_slab_planes": (C.c_int, [C.c_uint32, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "sph_slab_configure": (C.c_int, [_CTX, C.c_int, C.c_int]),
    "sph_comm_unique_id": (C.c_int, [C.c_void_p]),
    "sph_slab_comm_init": (C.c_int, [_CTX, C.c_void_p]),
    "sph_slab_set_host_transport": (C.c_int, [_CTX, HOST_EXCHANGE_FN, HOST_ALLREDUCE_FN, C.c_void_p]),
    "sph_slab_upload": (C.c_int, [_CTX, C.c_uint32, _FP, _FP, C.POINTER(C.c_uint32)]),
    "sph_slab_step": (C.c_int, [_CTX, C.c_float]),
    "sph_slab_finish": (C.c_int, [_CTX]),
    "sph_slab_counts": (C.c_int, [_CTX, C.POINTER(C.c_uint32)]),
    "sph_slab_download": (C.c_int, [_CTX, C.c_uint32, C.POINTER(C.c_uint32), _FP, _FP, _FP, C.POINTER(C.c_uint32)]),
    "sph_last_error": (C.c_char_p, [_CTX]),
}

_lib = None


def lib() -> C.CDLL:
    """Load libsph_hip.so (built by sphfluidsimulation_amd.build / __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -m sphfluidsimulation_amd.build` "
                "(hipcc, gfx950). The engine has no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError if the library does not export it
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc: int, ctx=None):
    if rc != SPH_OK:
        msg = lib().sph_last_error(ctx)
        raise SphError(rc, msg.decode() if msg else "")


def runtime_info() -> dict:
    info = SphRuntimeInfo()
    info.struct_size = C.sizeof(SphRuntimeInfo)
    check(lib().sph_runtime_info(C.byref(info)))
    return {"hip_runtime_version": info.hip_runtime_version, "hip_driver_version": info.hip_driver_version,
            "device_count": info.device_count, "rccl_version": info.rccl_version,
            "hip_library_path": info.hip_library_path.decode(), "rccl_library_path": info.rccl_library_path.decode()}


def default_params() -> SphParams:
    p = SphParams()
    check(lib().sph_params_default(C.byref(p)))
    return p
