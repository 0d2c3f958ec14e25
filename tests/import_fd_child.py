"""Child process of test_gpu_mesh_properties_zero_copy_export: imports an inherited dma-buf file descriptor with the HIP
runtime (no torch, no libsph_hip in this process), maps it and prints a checksum of the first `nbytes` bytes -- the
same steps a Vulkan renderer performs with VK_KHR_external_memory_fd.  usage: import_fd_child.py FD ALLOC_BYTES NBYTES"""
import ctypes as C
import sys

import numpy as np

fd, size, nbytes = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
hip = C.CDLL("libamdhip64.so.7")


class Loc(C.Structure):
    _fields_ = [("type", C.c_int), ("id", C.c_int)]


class Acc(C.Structure):
    _fields_ = [("location", Loc), ("flags", C.c_int)]


def ok(rc, what):
    if rc != 0:
        print(f"FAIL {what} rc={rc}")
        sys.exit(3)


ok(hip.hipSetDevice(0), "hipSetDevice")
handle, va = C.c_void_p(), C.c_void_p()
ok(hip.hipMemImportFromShareableHandle(C.byref(handle), C.c_void_p(fd), 1), "import")          # PosixFileDescriptor
ok(hip.hipMemAddressReserve(C.byref(va), C.c_size_t(size), C.c_size_t(0), None, C.c_ulonglong(0)), "reserve")
ok(hip.hipMemMap(va, C.c_size_t(size), C.c_size_t(0), handle, C.c_ulonglong(0)), "map")
acc = Acc(Loc(1, 0), 3)                                                                           # device 0, read-write
ok(hip.hipMemSetAccess(va, C.c_size_t(size), C.byref(acc), C.c_size_t(1)), "access")
out = np.empty(nbytes // 4, np.uint32)
ok(hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), va, C.c_size_t(nbytes), 2), "memcpy")            # DeviceToHost
print("CHECKSUM", int(out.astype(np.uint64).sum()), int(np.bitwise_xor.reduce(out)))
hip.hipMemUnmap(va, C.c_size_t(size))
hip.hipMemAddressFree(va, C.c_size_t(size))
hip.hipMemRelease(handle)
