"""Synthetic dam-break particle sets (host side, numpy).

Lattice of the reference's ``Preset1`` kernel (Assets/Resources/InitParticles.compute:10-41,
dispatched from Assets/Scripts/SphFluidSimulation.cs:179-184): ``ppd = ceil((cnt/fill)^(1/3))``,
``xSize = ceil(ppd*fill)``, ``ySize = zSize = ppd``, cell size ``cs = 1/ppd`` and for particle i

    x = cs/2 + (i / (zSize*ySize))      * fill / xSize          (:31)
    y = cs/2 + ((i / zSize) % ySize)    * 0.9  / ySize          (:32)
    z = cs/2 + (i % zSize)              * 1.0  / zSize          (:33)

followed by the SAME scalar jitter ``j_i * cs`` on x, y and z (:37).  The reference takes
``j_i`` from 4-D simplex noise (no seed); BASELINE.json asks for "the same random seed", so the
benchmark input uses a counter-based hash instead: ``j_i = 2*u - 1`` with ``u`` the top 24 bits of
``splitmix64(seed + i*0x9E3779B97F4A7C15)`` divided by 2**24 (exact in fp32).  Velocities start
at zero (SphFluidSimulation.cs:189) and ``pos.w = 1`` (:38).

This generates INPUT DATA for both the engine and the oracle; it is not on the hot path.
"""
from __future__ import annotations

import math

import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)


def next_power_of_two(v: int) -> int:
    """Mathf.NextPowerOfTwo (SphFluidSimulation.cs:84)."""
    p = 1
    while p < v:
        p <<= 1
    return p


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = x + _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def hash_uniform(seed: int, n: int, start: int = 0) -> np.ndarray:
    """u_i in [0,1), multiples of 2^-24, for i in [start, start+n)."""
    i = np.arange(start, start + n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = np.uint64(seed) + i * _GOLDEN
    bits = _splitmix64(x) >> np.uint64(40)
    return (bits.astype(np.float32) * np.float32(2.0 ** -24)).astype(np.float32)


def scene_bucket_resolution(num_particles: int) -> int:
    """R that keeps the shipped scene's particles-per-cell regime
    (Assets/Scenes/SampleScene.unity:363-364: N=262144, R=47), capped at the inspector
    maximum 256 (SphFluidSimulation.cs:37):  R - 1 = 46 * (N/262144)^(1/3)."""
    r = int(round(46.0 * (num_particles / 262144.0) ** (1.0 / 3.0))) + 1
    return max(2, min(256, r))


def dam_break(num_particles: int, fill: float = 0.8, seed: int = 1, jitter: float = 1.0):
    """Return (pos4, vel4): float32 arrays [res*res, 4] in particle-id order (id = y*res + x,
    Assets/Resources/Bucket.compute:24)."""
    n = next_power_of_two(int(num_particles))
    res = int(math.sqrt(np.float32(n)))
    cnt = res * res
    f32 = np.float32
    fill32 = f32(fill)
    # InitParticles.compute:19-26
    ppd = int(math.ceil(float(np.power(f32(cnt) / fill32, f32(1.0 / 3.0), dtype=np.float32))))
    x_size = int(math.ceil(float(f32(ppd) * fill32)))
    y_size = ppd
    z_size = ppd
    cs = f32(1.0) / f32(ppd)
    half = cs / f32(2.0)
    i = np.arange(cnt, dtype=np.int64)
    qx = (i // (z_size * y_size)).astype(np.float32)
    qy = ((i // z_size) % y_size).astype(np.float32)
    qz = (i % z_size).astype(np.float32)
    pos = np.empty((cnt, 4), np.float32)
    pos[:, 0] = half + qx * fill32 / f32(x_size)
    pos[:, 1] = half + qy * f32(0.9) / f32(y_size)
    pos[:, 2] = half + qz * f32(1.0) / f32(z_size)
    if jitter != 0.0:
        j = (f32(2.0) * hash_uniform(seed, cnt) - f32(1.0)) * f32(jitter)
        pos[:, :3] += (j * cs)[:, None]
    pos[:, 3] = 1.0
    vel = np.zeros((cnt, 4), np.float32)
    return pos, vel


def bench_dt_frame(bucket_resolution: int) -> float:
    """The reference leaves dt to Time.deltaTime (SphFluidSimulation.cs:102); the bench fixes
    dt_frame = 0.7666667*h, i.e. 1/60 s at the shipped scene's R = 47."""
    h = np.float32(1.0) / np.float32(bucket_resolution - 1)
    return float(np.float32(0.7666667) * h)
