"""Row f1: InitParticles.compute presets + noiseSimplex.cginc::snoise(float4).

CPU part: the C oracle (oracle/sph_oracle.c::oracle_init_particles / oracle_snoise4) against an independent,
array-at-a-time numpy-float32 restatement written from the same source lines, plus structural properties.
The reference holds no golden vectors for this path either, so like the rest of the oracle it is
"parity unpinned": two independent restatements agreeing bit for bit is the strongest check available here.
GPU part: sph_init_particles (HIP) == oracle, bit-exact, and frames started from it stay bit-exact.
"""
import numpy as np
import pytest

from oracle import sph_oracle as so
from sphfluidsimulation_amd import presets

f32 = np.float32


# ---- independent numpy restatement (noiseSimplex.cginc:62-130, 296-425) --------------------------------------------
def _mod289(x):
    return x - np.floor(x * f32(0.00346020761245674740484429065744)) * f32(289.0)


def _permute(x):
    return _mod289(x * x * f32(34.0) + x)


def _tis(r):
    return f32(1.79284291400159) - f32(0.85373472095314) * r


def _step(y, x):
    return (x >= y).astype(f32)


def _sat(x):
    return np.minimum(np.maximum(x, f32(0)), f32(1))


def _dot(a, b):
    acc = a[0] * b[0]
    for k in range(1, len(a)):
        acc = acc + a[k] * b[k]
    return acc


def _grad4(j, ip):
    p = [np.floor((j * ip[k] - np.floor(j * ip[k])) * f32(7.0)) * ip[2] - f32(1.0) for k in range(3)]
    pw = f32(1.5) - _dot([np.abs(q) for q in p], [f32(1), f32(1), f32(1)])
    s = [f32(1) - _step(f32(0), q) for q in p + [pw]]
    p = [p[k] + (s[k] * f32(2) - f32(1)) * s[3] for k in range(3)]
    return p + [pw]


def np_snoise4(v):
    """v: (n,4) float32 -> (n,) float32"""
    v = [np.ascontiguousarray(v[:, k], dtype=f32) for k in range(4)]
    C = [f32(0.138196601125011), f32(0.276393202250021), f32(0.414589803375032), f32(-0.447213595499958)]
    F4 = f32(0.309016994374947451)
    sk = _dot(v, [F4] * 4)
    i = [np.floor(c + sk) for c in v]
    un = _dot(i, [C[0]] * 4)
    x0 = [v[k] - i[k] + un for k in range(4)]
    isX = [_step(x0[1], x0[0]), _step(x0[2], x0[0]), _step(x0[3], x0[0])]
    isYZ = [_step(x0[2], x0[1]), _step(x0[3], x0[1]), _step(x0[3], x0[2])]
    i0 = [isX[0] + isX[1] + isX[2], f32(1) - isX[0], f32(1) - isX[1], f32(1) - isX[2]]
    i0[1] = i0[1] + (isYZ[0] + isYZ[1])
    i0[2] = i0[2] + (f32(1) - isYZ[0])
    i0[3] = i0[3] + (f32(1) - isYZ[1])
    i0[2] = i0[2] + isYZ[2]
    i0[3] = i0[3] + (f32(1) - isYZ[2])
    i3 = [_sat(c) for c in i0]
    i2 = [_sat(c - f32(1)) for c in i0]
    i1 = [_sat(c - f32(2)) for c in i0]
    x1 = [x0[k] - i1[k] + C[0] for k in range(4)]
    x2 = [x0[k] - i2[k] + C[1] for k in range(4)]
    x3 = [x0[k] - i3[k] + C[2] for k in range(4)]
    x4 = [x0[k] + C[3] for k in range(4)]
    i = [_mod289(c) for c in i]
    j0 = _permute(_permute(_permute(_permute(i[3]) + i[2]) + i[1]) + i[0])
    one = np.ones_like(i[0])
    j1 = []
    for a in (i1, i2, i3, [one] * 4):
        j1.append(_permute(_permute(_permute(_permute(i[3] + a[3]) + i[2] + a[2]) + i[1] + a[1]) + i[0] + a[0]))
    ip = [f32(0.003401360544217687075), f32(0.020408163265306122449), f32(0.142857142857142857143), f32(0)]
    P = [_grad4(j0, ip)] + [_grad4(j, ip) for j in j1]
    for k in range(5):
        n = _tis(_dot(P[k], P[k]))
        P[k] = [c * n for c in P[k]]
    X = [x0, x1, x2, x3, x4]
    m = [np.maximum(f32(0.6) - _dot(x, x), f32(0)) for x in X]
    m = [c * c for c in m]
    d = [_dot(P[k], X[k]) for k in range(5)]
    a = _dot([m[0] * m[0], m[1] * m[1], m[2] * m[2]], d[:3])
    b = _dot([m[3] * m[3], m[4] * m[4]], d[3:])
    return f32(49.0) * (a + b)


def np_init_particles(preset, res, fill, ppd, noise=True):
    """InitParticles.compute:10-107, array-at-a-time."""
    fill = f32(fill)
    n = res * res
    i = np.arange(n, dtype=np.uint32)
    xs = np.uint32(np.ceil(f32(ppd) * fill))
    ys = np.uint32(ppd)
    zs = np.uint32(ppd) if preset == 0 else np.uint32(np.ceil(f32(ppd) * fill))
    cs = f32(1.0) / f32(ppd)
    qx = (i // (zs * ys)).astype(f32)
    qy = ((i // zs) % ys).astype(f32)
    qz = (i % zs).astype(f32)
    half = cs / f32(2.0)
    if preset == 0:
        px = half + qx * fill / f32(xs)
        pz = half + qz * f32(1.0) / f32(zs)
    elif preset == 1:
        px = half + fill / f32(2.0) + qx * fill / f32(xs)
        pz = half + fill / f32(2.0) + qz * fill / f32(zs)
    else:
        px = half + qx * fill / f32(xs)
        pz = half + qz * fill / f32(zs)
    py = half + qy * f32(0.9) / f32(ys)
    pos = np.stack([px, py, pz, np.ones(n, f32)], axis=1).astype(f32)
    if noise:
        fi = i.astype(np.int32).astype(f32)
        nz = np_snoise4(pos + fi[:, None]) * cs
        pos = pos + nz[:, None]
    pos[:, 3] = f32(1.0)
    return pos


def _bits(a):
    return np.ascontiguousarray(a, dtype=f32).view(np.uint32)


# ---- CPU tests ------------------------------------------------------------------------------------------------------
def test_snoise4_two_restatements_agree_bitwise():
    rng = np.random.default_rng(5)
    v = np.concatenate([
        rng.uniform(-4, 4, (4000, 4)),
        rng.uniform(0, 300000, (4000, 4)),           # the magnitudes pos + i reaches at 262144 particles
        rng.uniform(-1e-3, 1e-3, (500, 4)),
        np.zeros((1, 4)), np.ones((1, 4)),
        np.array([[0.5, 0.5, 0.5, 0.5], [1.0, 2.0, 3.0, 4.0], [289.0, 578.0, -289.0, 0.25]]),
    ]).astype(f32)
    got = so.snoise4(v)
    want = np_snoise4(v)
    assert np.array_equal(_bits(got), _bits(want))
    assert np.all(np.isfinite(got)) and np.abs(got).max() <= 1.0 + 1e-5
    assert so.snoise4(np.zeros((1, 4), f32))[0] == 0.0     # every simplex corner has zero offset-dot-gradient at the origin


def test_snoise4_is_continuous_and_not_constant():
    rng = np.random.default_rng(6)
    v = rng.uniform(-3, 3, (2000, 4)).astype(f32)
    a = so.snoise4(v)
    b = so.snoise4(v + f32(1e-4))
    assert np.abs(a - b).max() < 5e-3
    assert a.std() > 0.1 and abs(a.mean()) < 0.05


@pytest.mark.parametrize("preset", [0, 1, 2])
@pytest.mark.parametrize("res,fill", [(32, 0.8), (64, 0.8), (128, 0.5), (4, 1.0), (64, 0.37)])
def test_init_particles_two_restatements_agree_bitwise(preset, res, fill):
    ppd = so.particle_per_dim(preset, res, fill)
    want = np_init_particles(preset, res, fill, ppd)
    got = so.init_particles(preset, res, fill)
    assert got.shape == (res * res, 4)
    assert np.array_equal(_bits(got), _bits(want))
    assert np.all(got[:, 3] == 1.0)


def test_particle_per_dim_scene_values():
    # SampleScene: 262144 particles, fill 0.8 -> (262144/0.8)^(1/3) = 68.9.. -> 69; Preset2 divides by fill^2
    assert so.particle_per_dim(0, 512, 0.8) == 69
    assert so.particle_per_dim(1, 512, 0.8) == 75
    assert so.particle_per_dim(2, 512, 0.8) == 75
    assert so.particle_per_dim(0, 64, 0.8) == 18
    for preset, res, fill in [(0, 512, 0.8), (1, 512, 0.8), (0, 4096, 0.8), (2, 100, 0.3)]:
        ppd = so.particle_per_dim(preset, res, fill)
        xs = int(np.ceil(f32(ppd) * f32(fill)))
        zs = ppd if preset == 0 else xs
        assert xs * ppd * zs >= res * res          # the lattice has room for every particle (InitParticles.compute:20-24)


def test_preset1_lattice_matches_host_preset_without_jitter():
    for n, fill in [(4096, 0.8), (16384, 0.5)]:
        res = int(np.sqrt(n))
        lattice = so.init_particles(0, res, fill, noise=False)
        host, _ = presets.dam_break(n, fill, seed=1, jitter=0.0)
        assert np.array_equal(_bits(lattice), _bits(host))


@pytest.mark.parametrize("preset", [0, 1, 2])
def test_init_particles_extent(preset):
    res, fill = 64, 0.8
    ppd = so.particle_per_dim(preset, res, fill)
    cs = 1.0 / ppd
    base = so.init_particles(preset, res, fill, noise=False)
    pos = so.init_particles(preset, res, fill)
    # one scalar per particle is added to x, y and z alike (InitParticles.compute:37), bounded by the cube size
    d = pos[:, :3] - base[:, :3]
    assert np.abs(d).max() <= cs * 1.0001
    assert np.abs(d[:, 0] - d[:, 1]).max() < 1e-6 and np.abs(d[:, 0] - d[:, 2]).max() < 1e-6
    lo, hi = base[:, :3].min(0), base[:, :3].max(0)
    if preset == 0:
        assert hi[0] < 0.8 + cs and hi[2] < 1.0
    elif preset == 1:
        # Preset2 starts at fill/2 and spans `fill`: with fill 0.8 it reaches 1.2, outside the unit box
        assert lo[0] > 0.4 and lo[2] > 0.4 and hi[0] > 1.0
    else:
        assert hi[0] < 0.8 + cs and hi[2] < 0.8 + cs
    assert hi[1] < 0.9 + cs


# ---- GPU parity ------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("preset", [0, 1, 2])
@pytest.mark.parametrize("n,fill", [(4096, 0.8), (262144, 0.8), (65536, 0.45)])
def test_gpu_init_particles_bit_exact(preset, n, fill):
    from sphfluidsimulation_amd.simulation import SphFluidSimulation
    sim = SphFluidSimulation(preset=preset, particleNumber=n, bucketResolution=32, damFillRate=fill).Start(seed=None)
    try:
        pos, vel, _ = sim.download()
        want = so.init_particles(preset, int(np.sqrt(n)), fill)
        assert np.array_equal(_bits(pos), _bits(want))
        assert not vel.any()
    finally:
        sim.OnDestroy()


@pytest.mark.gpu
@pytest.mark.parametrize("preset", [0, 1])
def test_gpu_frames_from_device_init_match_oracle(preset):
    """Start() like the reference (device-side InitParticles), then 3 frames: bit-exact against the oracle
    started from the oracle's own InitParticles.  Preset2 at fill 0.8 starts partly outside the unit box."""
    from sphfluidsimulation_amd.simulation import SphFluidSimulation
    from sphfluidsimulation_amd import presets as P
    n, fill = 16384, 0.8
    R = P.scene_bucket_resolution(n)
    sim = SphFluidSimulation(preset=preset, particleNumber=n, bucketResolution=R, damFillRate=fill).Start(seed=None)
    try:
        osim = so.OracleSimulation(so.derive(n, R, fill))
        osim.upload(so.init_particles(preset, 128, fill), np.zeros((n, 4), f32))
        for _ in range(3):
            sim.Update(1 / 60)
            osim.Update(1 / 60)
        pos, vel, rho = sim.download()
        opos, ovel, orho = osim.download()
        assert np.array_equal(_bits(pos), _bits(opos))
        assert np.array_equal(_bits(vel), _bits(ovel))
        assert np.array_equal(_bits(rho), _bits(orho))
    finally:
        sim.OnDestroy()


@pytest.mark.gpu
def test_gpu_init_particles_errors():
    from sphfluidsimulation_amd.simulation import SphFluidSimulation
    from sphfluidsimulation_amd import _native as N
    sim = SphFluidSimulation(particleNumber=1024, bucketResolution=8).Start(seed=None)
    try:
        assert N.lib().sph_init_particles(sim._ctx, 3) == N.SPH_E_INVALID_ARG
        assert N.lib().sph_init_particles(sim._ctx, -1) == N.SPH_E_INVALID_ARG
        assert N.lib().sph_init_particles(None, 0) == N.SPH_E_INVALID_ARG
    finally:
        sim.OnDestroy()
