"""Exports the render-feed buffer (sph_export_mesh_properties_fd) and imports the dma-buf fd IN THE SAME PROCESS -- what a
Unity / Vulkan host does -- through the HIP runtime image libsph_hip.so itself is bound to (sph_runtime_info), with typed
ctypes arguments.  Prints CHECKSUM lines; run as a script so that a crash cannot take the test session down.

    import_fd_same_process.py same        the supported way
    import_fd_same_process.py two-images  round 1's layout: libsph_hip.so loaded first (system runtime), PyTorch imported
                                          afterwards (a SECOND runtime image), the import issued through that other image
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
mode = sys.argv[1] if len(sys.argv) > 1 else "same"

from sphfluidsimulation_amd import _native as N  # noqa: E402
from sphfluidsimulation_amd import presets  # noqa: E402
from sphfluidsimulation_amd.simulation import SphFluidSimulation  # noqa: E402

N.lib()
info = N.runtime_info()
print("ENGINE_RUNTIME", info["hip_library_path"], info["hip_runtime_version"], flush=True)
if mode == "two-images":
    import torch  # noqa: F401  (maps torch/lib/libamdhip64.so next to the image above)
    hip_path = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
else:
    hip_path = info["hip_library_path"]
hip = C.CDLL(hip_path)
images = sorted({l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l})
print("HIP_IMAGES", len(images), " ".join(images), flush=True)
print("IMPORT_THROUGH", hip_path, flush=True)


class Loc(C.Structure):
    _fields_ = [("type", C.c_int), ("id", C.c_int)]


class Acc(C.Structure):
    _fields_ = [("location", Loc), ("flags", C.c_int)]


hip.hipMemImportFromShareableHandle.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_int]
hip.hipMemAddressReserve.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_size_t, C.c_void_p, C.c_ulonglong]
hip.hipMemMap.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_ulonglong]
hip.hipMemSetAccess.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(Acc), C.c_size_t]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
hip.hipMemUnmap.argtypes = [C.c_void_p, C.c_size_t]
hip.hipMemAddressFree.argtypes = [C.c_void_p, C.c_size_t]
hip.hipMemRelease.argtypes = [C.c_void_p]
for f in ("hipMemImportFromShareableHandle", "hipMemAddressReserve", "hipMemMap", "hipMemSetAccess", "hipMemcpy", "hipMemUnmap",
          "hipMemAddressFree", "hipMemRelease", "hipSetDevice"):
    getattr(hip, f).restype = C.c_int


def ok(rc, what):
    print("STEP", what, rc, flush=True)
    if rc != 0:
        print(f"FAIL {what} rc={rc}", flush=True)
        sys.exit(3)


n = 16384
pos0, vel0 = presets.dam_break(n, 0.8, seed=8, jitter=0.3)
vel0[:, :3] = np.random.default_rng(2).standard_normal((n, 3)).astype(np.float32) * np.float32(0.2)
sim = SphFluidSimulation(particleNumber=n, bucketResolution=19, damFillRate=0.8).Start(pos0, vel0)
sim.UpdateMeshProperties()


def checksum(a):
    u = np.ascontiguousarray(a).view(np.uint32).reshape(-1)
    return int(u.astype(np.uint64).sum()), int(np.bitwise_xor.reduce(u))


fd, size = C.c_int(-1), C.c_size_t(0)
rc = N.lib().sph_export_mesh_properties_fd(sim._ctx, C.byref(fd), C.byref(size))
if rc == N.SPH_E_UNSUPPORTED:
    print("UNSUPPORTED", N.lib().sph_last_error(sim._ctx).decode(), flush=True)
    sys.exit(0)
ok(rc, "export")
if mode == "two-images":
    ok(hip.hipSetDevice(0), "hipSetDevice (second image)")
handle, va = C.c_void_p(), C.c_void_p()
ok(hip.hipMemImportFromShareableHandle(C.byref(handle), C.c_void_p(fd.value), 1), "import")     # PosixFileDescriptor
ok(hip.hipMemAddressReserve(C.byref(va), size.value, 0, None, 0), "reserve")
ok(hip.hipMemMap(va, size.value, 0, handle, 0), "map")
acc = Acc(Loc(1, 0), 3)                                                                            # device 0, read-write
ok(hip.hipMemSetAccess(va, size.value, C.byref(acc), 1), "access")


def through_import():
    out = np.empty(n * 20, np.float32)
    sim.synchronize()
    ok(hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), va, n * 80, 2), "memcpy")                     # DeviceToHost
    return out


print("CHECKSUM before", *checksum(sim.download_mesh_properties()), *checksum(through_import()), flush=True)
sim.Update(1 / 600)
sim.UpdateMeshProperties()                      # writes into the exported memory
print("CHECKSUM after", *checksum(sim.download_mesh_properties()), *checksum(through_import()), flush=True)
ok(hip.hipMemUnmap(va, size.value), "unmap")
hip.hipMemAddressFree(va, size.value)
hip.hipMemRelease(handle)
os.close(fd.value)
sim.OnDestroy()
print("DONE", flush=True)
