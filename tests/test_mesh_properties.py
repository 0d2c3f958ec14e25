"""Row f3: UpdateMeshProperties.compute:18-64, the render feed after the path.

CPU: oracle (oracle_update_mesh_properties) vs an independent numpy restatement + hand-computed records.
GPU: sph_update_mesh_properties / sph_download_mesh_properties == oracle on the engine's own state, bit-exact.
No reference fixture exists for this kernel ("parity unpinned", like the rest of the oracle).
"""
import numpy as np
import pytest

from oracle import sph_oracle as so
from sphfluidsimulation_amd import presets

f32 = np.float32


def _bits(a):
    return np.ascontiguousarray(a, dtype=f32).view(np.uint32)


def np_mesh_properties(pos, vel, low, high, scale, trs):
    """Array-at-a-time restatement for FINITE inputs: mul(translation, scale) written out as its closed form
    (diag(scale) with the translation in the last column), which is what the 4-term sums reduce to."""
    pos = pos.astype(f32)
    vel = vel.astype(f32)
    n = pos.shape[0]
    ss = np.array([trs[0][0], trs[1][1], trs[2][2]], f32)
    w = pos[:, :3] * ss - ss * f32(0.5)
    out = np.zeros((n, 20), f32)
    out[:, 0], out[:, 5], out[:, 10], out[:, 15] = scale[0], scale[1], scale[2], 1.0
    out[:, 12:15] = w + f32(0.0)                       # 0 + w: -0 becomes +0 as in the summed product
    speed = np.sqrt((vel[:, 0] * vel[:, 0] + vel[:, 1] * vel[:, 1]) + vel[:, 2] * vel[:, 2])
    with np.errstate(divide="ignore", invalid="ignore"):
        t = (speed - f32(low)) / (f32(high) - f32(low))
    t = np.where(np.isnan(t), f32(0), np.minimum(np.maximum(t, f32(0)), f32(1))).astype(f32)
    out[:, 16] = t
    out[:, 17] = 0
    out[:, 18] = f32(1) + t * f32(-1)
    out[:, 19] = 1
    return out


def test_mesh_record_by_hand():
    pos = np.array([[0.25, 0.5, 1.0, 1.0], [0.0, 0.0, 0.0, 1.0]], f32)
    vel = np.array([[0.3, 0.0, 0.4, 7.0], [0.0, 0.0, 0.0, 0.0]], f32)     # |v| = 0.5 (w ignored), 0
    trs = np.diag([5.0, 3.0, 5.0, 1.0]).astype(f32)
    trs[:3, 3] = [0.0, -1.5, 0.0]                                         # the scene's fluid-box transform; translation is unused (:38)
    out = so.update_mesh_properties(pos, vel, 0.0, 1.0, 0.01, trs)
    r = out[0]
    assert r[0] == f32(0.01) and r[5] == f32(0.01) and r[10] == f32(0.01) and r[15] == 1.0
    assert list(r[12:15]) == [f32(0.25) * f32(5) - f32(2.5), f32(0.5) * f32(3) - f32(1.5), f32(5) - f32(2.5)]
    assert [r[k] for k in (1, 2, 3, 4, 6, 7, 8, 9, 11)] == [0.0] * 9
    t = f32(np.sqrt(f32(f32(0.3) * f32(0.3) + f32(0.0)) + f32(0.4) * f32(0.4)))
    assert r[16] == t and r[17] == 0.0 and r[18] == f32(1) - t and r[19] == 1.0
    assert list(out[1][16:]) == [0.0, 0.0, 1.0, 1.0]                       # at rest: pure blue
    assert list(out[1][12:15]) == [-2.5, -1.5, -2.5]


@pytest.mark.parametrize("low,high", [(0.0, 0.5), (0.2, 0.2), (1.0, 0.25), (0.0, 1000.0)])
def test_mesh_properties_two_restatements_agree_bitwise(low, high):
    rng = np.random.default_rng(8)
    n = 5000
    pos = rng.uniform(0, 1, (n, 4)).astype(f32)
    pos[:7, :3] = [[0, 0, 0], [1, 1, 1], [0.5, 0.5, 0.5], [0.5, 0, 1], [1e-30, 0.5, 0.25], [0.5, 0.5, 0], [0, 0.5, 0.5]]
    vel = (rng.standard_normal((n, 4)) * rng.choice([0.0, 0.01, 0.3, 5.0], (n, 1))).astype(f32)
    trs = np.diag([5.0, 3.0, 2.0, 1.0]).astype(f32)
    scale = np.array([0.01, 0.02, 0.03], f32)
    got = so.update_mesh_properties(pos, vel, low, high, scale, trs)
    want = np_mesh_properties(pos, vel, low, high, scale, trs)
    assert np.array_equal(_bits(got), _bits(want))
    assert got[:, 16].min() >= 0.0 and got[:, 16].max() <= 1.0


def test_mesh_properties_non_finite_position_poisons_its_row():
    pos = np.array([[np.inf, 0.5, 0.5, 1.0], [0.5, np.nan, 0.5, 1.0]], f32)
    vel = np.array([[np.nan, 0, 0, 0], [np.inf, 0, 0, 0]], f32)
    out = so.update_mesh_properties(pos, vel, 0.0, 0.5, 0.01)
    m0 = out[0, :16].reshape(4, 4).T            # back to (row, column)
    m1 = out[1, :16].reshape(4, 4).T
    assert np.isnan(m0[0, :3]).all() and m0[0, 3] == np.inf and not np.isnan(m0[1:]).any()   # inf * 0 = NaN, inf * 1 = inf
    assert np.isnan(m1[1]).all() and not np.isnan(m1[[0, 2, 3]]).any()
    assert out[0, 16] == 0.0 and out[1, 16] == 1.0                   # saturate(NaN) = 0, saturate(inf) = 1


# ---- GPU parity ------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("frames", [0, 2])
def test_gpu_mesh_properties_bit_exact(frames):
    from sphfluidsimulation_amd.simulation import SphFluidSimulation
    n, fill = 16384, 0.8
    R = presets.scene_bucket_resolution(n)
    pos0, vel0 = presets.dam_break(n, fill, seed=5)
    vel0[:, :3] = np.random.default_rng(1).standard_normal((n, 3)).astype(f32) * f32(0.2)
    sim = SphFluidSimulation(particleNumber=n, bucketResolution=R, damFillRate=fill, particleRadius=0.013,
                             lowSpeed=0.05, highSpeed=0.45).Start(pos0, vel0)
    try:
        for _ in range(frames):
            sim.Update(1 / 60)
        trs = np.diag([5.0, 3.0, 5.0, 1.0]).astype(f32)
        trs[:3, 3] = [0.0, -1.5, 0.0]
        sim.UpdateMeshProperties(trs)
        got = sim.download_mesh_properties()
        pos, vel, _ = sim.download()
        want = so.update_mesh_properties(pos, vel, 0.05, 0.45, 0.013, trs)
        assert got.shape == (n, 20)
        assert np.array_equal(_bits(got), _bits(want))
        if frames:
            assert 0.0 < got[:, 16].mean() < 1.0
    finally:
        sim.OnDestroy()


@pytest.mark.gpu
def test_gpu_mesh_properties_weird_inputs_and_errors():
    from sphfluidsimulation_amd.simulation import SphFluidSimulation
    from sphfluidsimulation_amd import _native as N
    import ctypes as C
    n = 1024
    pos0, vel0 = presets.dam_break(n, 0.8, seed=6)
    pos0[0, :3] = [np.inf, 0.5, 0.5]
    pos0[1, :3] = [0.5, np.nan, 0.5]
    pos0[2, :3] = [-3.0, 7.0, 0.5]
    vel0[3, :3] = [np.nan, 0, 0]
    vel0[4, :3] = [np.inf, 0, 0]
    vel0[5, :3] = [1e30, 1e30, 0]                 # |v|^2 overflows -> inf -> t = 1
    sim = SphFluidSimulation(particleNumber=n, bucketResolution=9, damFillRate=0.8)
    sim.Start(pos0, vel0)
    try:
        props = np.empty((n, 20), f32)
        fp = props.ctypes.data_as(C.POINTER(C.c_float))
        assert N.lib().sph_download_mesh_properties(sim._ctx, fp) == N.SPH_E_BAD_STATE   # nothing computed yet
        sim.UpdateMeshProperties()
        got = sim.download_mesh_properties()
        want = so.update_mesh_properties(pos0, vel0, sim.lowSpeed, sim.highSpeed, sim.particleRadius)
        assert np.array_equal(_bits(got), _bits(want))
        bad = N.SphMeshParams()
        assert N.lib().sph_update_mesh_properties(sim._ctx, C.byref(bad)) == N.SPH_E_INVALID_ARG   # struct_size 0
        assert N.lib().sph_update_mesh_properties(sim._ctx, None) == N.SPH_E_INVALID_ARG
        ptr, nbytes = C.c_void_p(), C.c_size_t()
        assert N.lib().sph_mesh_properties_device_ptr(sim._ctx, C.byref(ptr), C.byref(nbytes)) == 0
        assert ptr.value and nbytes.value == n * 80
    finally:
        sim.OnDestroy()


@pytest.mark.gpu
def test_gpu_mesh_properties_zero_copy_export():
    """Row f4: the render-feed buffer as a dma-buf file descriptor.  A CHILD PROCESS that knows nothing of this library
    imports the inherited fd through the HIP runtime (hipMemImportFromShareableHandle + its own mapping -- what a
    Vulkan importer does with VK_KHR_external_memory_fd) and sees the records, including those of a LATER update,
    without any copy made by this library."""
    import ctypes as C
    import os
    import subprocess
    import sys
    from sphfluidsimulation_amd.simulation import SphFluidSimulation
    from sphfluidsimulation_amd import _native as N
    n = 16384
    pos0, vel0 = presets.dam_break(n, 0.8, seed=8, jitter=0.3)
    vel0[:, :3] = np.random.default_rng(2).standard_normal((n, 3)).astype(f32) * f32(0.2)
    sim = SphFluidSimulation(particleNumber=n, bucketResolution=19, damFillRate=0.8).Start(pos0, vel0)
    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), "import_fd_child.py")

    def checksum(a):
        u = _bits(a).reshape(-1)
        return int(u.astype(np.uint64).sum()), int(np.bitwise_xor.reduce(u))

    def seen_by_child(fd, size):
        sim.synchronize()
        env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
        r = subprocess.run([sys.executable, child, str(fd), str(size), str(n * 80)], pass_fds=(fd,), env=env,
                           capture_output=True, text=True, timeout=300)
        line = [l for l in r.stdout.splitlines() if l.startswith(("CHECKSUM", "FAIL"))]
        assert r.returncode == 0 and line and line[0].startswith("CHECKSUM"), (r.returncode, r.stdout[-500:], r.stderr[-500:])
        return tuple(int(x) for x in line[0].split()[1:])

    try:
        sim.UpdateMeshProperties()
        before = sim.download_mesh_properties()
        fd, size = C.c_int(-1), C.c_size_t(0)
        rc = N.lib().sph_export_mesh_properties_fd(sim._ctx, C.byref(fd), C.byref(size))
        if rc == N.SPH_E_UNSUPPORTED:
            pytest.skip("no exportable device memory on this system: " + N.lib().sph_last_error(sim._ctx).decode())
        assert rc == 0 and fd.value >= 0 and size.value >= n * 80
        assert np.array_equal(_bits(sim.download_mesh_properties()), _bits(before))      # survived the move
        assert seen_by_child(fd.value, size.value) == checksum(before)
        sim.Update(1 / 600)
        sim.UpdateMeshProperties()                   # writes into the exported memory
        after = sim.download_mesh_properties()
        assert checksum(after) != checksum(before)
        assert seen_by_child(fd.value, size.value) == checksum(after)
        os.close(fd.value)
    finally:
        sim.OnDestroy()


@pytest.mark.gpu
def test_gpu_mesh_properties_same_process_import():
    """Row f4 as a Unity host would use it: the fd is imported in the process that exported it, through the SAME HIP runtime
    image libsph_hip.so is bound to (sph_runtime_info names it) and with typed arguments.  Round 1 crashed here: its test
    asked ctypes for "libamdhip64.so" in a process that had mapped a second runtime image (PyTorch's bundled one) next to
    the system runtime the library used -- a handle exported by one image was imported through the other
    (tests/import_fd_same_process.py two-images reproduces that layout; profiles/ records what it does)."""
    import os
    import subprocess
    import sys
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "import_fd_same_process.py")
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    r = subprocess.run([sys.executable, script, "same"], capture_output=True, text=True, env=env, timeout=600)
    lines = r.stdout.splitlines()
    if any(l.startswith("UNSUPPORTED") for l in lines):
        pytest.skip("no exportable device memory on this system")
    assert r.returncode == 0 and "DONE" in lines, (r.returncode, r.stdout[-1500:], r.stderr[-1500:])
    images = [l for l in lines if l.startswith("HIP_IMAGES")][0].split()
    assert images[1] == "1", images                       # one runtime image in the process
    sums = {l.split()[1]: l.split()[2:] for l in lines if l.startswith("CHECKSUM")}
    for phase in ("before", "after"):
        assert sums[phase][:2] == sums[phase][2:], (phase, sums)   # the import sees what the library wrote, also after an update
    assert sums["before"] != sums["after"]


@pytest.mark.gpu
def test_gpu_state_textures_zero_copy_export():
    """Row f4, second step: the position / velocity / density textures (reference layout, SphFluidSimulation.cs:141-153) filled
    on the device and handed out as dma-buf fds; a child process imports them (as a renderer would) and sees what sph_download
    returns, also after a later update.  The semaphore entry points fail cleanly on something that is not a semaphore."""
    import ctypes as C
    import os
    import subprocess
    import sys
    from sphfluidsimulation_amd.simulation import SphFluidSimulation
    from sphfluidsimulation_amd import _native as N
    n = 16384
    pos0, vel0 = presets.dam_break(n, 0.8, seed=8, jitter=0.3)
    sim = SphFluidSimulation(particleNumber=n, bucketResolution=19, damFillRate=0.8).Start(pos0, vel0)
    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), "import_fd_child.py")
    L = N.lib()

    def checksum(a):
        u = _bits(np.ascontiguousarray(a)).reshape(-1)
        return int(u.astype(np.uint64).sum()), int(np.bitwise_xor.reduce(u))

    def seen_by_child(fd, size, nbytes):
        sim.synchronize()
        env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
        r = subprocess.run([sys.executable, child, str(fd), str(size), str(nbytes)], pass_fds=(fd,), env=env,
                           capture_output=True, text=True, timeout=300)
        line = [l for l in r.stdout.splitlines() if l.startswith(("CHECKSUM", "FAIL"))]
        assert r.returncode == 0 and line and line[0].startswith("CHECKSUM"), (r.returncode, r.stdout[-500:], r.stderr[-500:])
        return tuple(int(x) for x in line[0].split()[1:])

    try:
        sim.Update(1 / 600)
        fds = []
        for which, nbytes in ((0, n * 16), (1, n * 16), (2, n * 4)):
            fd, size = C.c_int(-1), C.c_size_t(0)
            rc = L.sph_export_state_fd(sim._ctx, which, C.byref(fd), C.byref(size))
            if rc == N.SPH_E_UNSUPPORTED:
                pytest.skip("no exportable device memory on this system: " + L.sph_last_error(sim._ctx).decode())
            assert rc == 0 and fd.value >= 0 and size.value >= nbytes
            fds.append((fd.value, size.value, nbytes))
        for frame in range(2):
            N.check(L.sph_update_state_textures(sim._ctx), sim._ctx)
            want = sim.download()
            for (fd, size, nbytes), arr in zip(fds, want):
                assert seen_by_child(fd, size, nbytes) == checksum(arr), f"texture differs from sph_download (frame {frame})"
            sim.Update(1 / 600)
        assert L.sph_signal_semaphore(sim._ctx, 1) == N.SPH_E_BAD_STATE
        r, w = os.pipe()
        assert L.sph_import_semaphore_fd(sim._ctx, r, 0) in (N.SPH_E_UNSUPPORTED, N.SPH_E_HIP)   # a pipe is no semaphore: refused, no crash
        os.close(r); os.close(w)
        for fd, _, _ in fds:
            os.close(fd)
    finally:
        sim.OnDestroy()
