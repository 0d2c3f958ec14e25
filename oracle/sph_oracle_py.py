"""Independent pure-Python restatement of the reference frame (small N only).

TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED by the reference (no tests / golden vectors
ship with it, and it cannot be run here).  This file exists to pin oracle/sph_oracle.c from a
second, structurally different direction: instead of the reference's 32-slot CAS table
(Assets/Resources/Bucket.compute:30-35) it builds cell lists by a STABLE sort of particle ids
by cell key and truncates each list to its 32 lowest ids -- the data structure the HIP engine
uses.  tests/test_oracle_kat.py requires the two to agree bit for bit.

All arithmetic is done on numpy float32 scalars, one IEEE operation per Python expression node,
in the source order of Density.compute:22-60 and VelPos.compute:33-158.
"""
from __future__ import annotations

import math

import numpy as np

f32 = np.float32
MAX_PER_VOXEL = 32
PI = f32(3.14159265358979323846264338327950288)
EPS = f32(1e-6)


def _int(f) -> int:
    f = float(f)
    if math.isnan(f):
        return 0
    return int(max(-2147483648.0, min(2147483647.0, math.trunc(f))))


def _uint(f) -> int:
    f = float(f)
    if not f > 0.0:
        return 0
    return int(min(4294967295.0, math.trunc(f)))


def _clamp01(x):
    x = float(x)
    if math.isnan(x):
        return f32(0.0)
    return f32(min(max(x, 0.0), 1.0))


def derive(particle_number: int, R: int, fill: float):
    """SphFluidSimulation.cs:84-85,159,176,236-237,255."""
    N = 1
    while N < particle_number:
        N <<= 1
    res = int(f32(math.sqrt(float(f32(N)))))
    h = f32(1.0) / f32(R - 1)
    return dict(N=N, res=res, R=R, h=h, h2=h * h, h6=f32(math.pow(float(h), 6.0)),
                h9=f32(math.pow(float(h), 9.0)), m=f32(fill) / f32(N))


def cell_lists(d, pos4):
    """Cell key per particle (Bucket.compute:27-28, float arithmetic) -> dict key -> ids
    ascending, truncated to 32 (== serial execution of Bucket.compute:30-35)."""
    R, cnt = d["R"], d["res"] * d["res"]
    Rm1, Rf = f32(R - 1), f32(R)
    keys = np.empty(cnt, np.int64)
    for i in range(cnt):
        ix = f32(_int(pos4[i, 0] * Rm1))
        iy = f32(_int(pos4[i, 1] * Rm1))
        iz = f32(_int(pos4[i, 2] * Rm1))
        k = _uint((ix + iy * Rf) + iz * Rf * Rf)
        k = (k * MAX_PER_VOXEL) & 0xFFFFFFFF
        keys[i] = k // MAX_PER_VOXEL if k < R * R * R * MAX_PER_VOXEL else -1
    order = np.argsort(keys, kind="stable")
    lists = {}
    for i in order:
        k = int(keys[i])
        if k < 0:
            continue
        lst = lists.setdefault(k, [])
        if len(lst) < MAX_PER_VOXEL:
            lst.append(int(i))
    return lists


def _neighbours(d, lists, pos_i):
    R = d["R"]
    Rm1 = f32(R - 1)
    ix, iy, iz = (f32(_int(pos_i[a] * Rm1)) for a in range(3))
    for xo in (-1, 0, 1):
        for yo in (-1, 0, 1):
            for zo in (-1, 0, 1):
                jx, jy, jz = _int(ix + f32(xo)), _int(iy + f32(yo)), _int(iz + f32(zo))
                if 0 <= jx < R and 0 <= jy < R and 0 <= jz < R:
                    for j in lists.get(jx + jy * R + jz * R * R, ()):
                        yield j


def density(d, pos4, lists):
    cnt = d["res"] * d["res"]
    c = f32(315.0) / (f32(64.0) * PI)
    rho = np.zeros(cnt, np.float32)
    m, h2, h9 = d["m"], d["h2"], d["h9"]
    for i in range(cnt):
        acc = f32(0.0)
        for j in _neighbours(d, lists, pos4[i]):
            rx, ry, rz = pos4[i, 0] - pos4[j, 0], pos4[i, 1] - pos4[j, 1], pos4[i, 2] - pos4[j, 2]
            r2 = (rx * rx + ry * ry) + rz * rz
            diff = h2 - r2
            w = c * ((diff * diff) * diff) / h9 if f32(0.0) < diff else f32(0.0)
            acc = acc + m * w
        rho[i] = acc
    return rho


def velpos(d, pos4, vel4, rho, lists, dt, mu, k, rho0, ks, kd):
    cnt = d["res"] * d["res"]
    h, h6, m = d["h"], d["h6"], d["m"]
    mu, k, rho0, ks, kd, dt = f32(mu), f32(k), f32(rho0), f32(ks), f32(kd), f32(dt)
    c = f32(45.0) / PI
    one, zero = f32(1.0), f32(0.0)
    pos_o = np.empty_like(pos4)
    vel_o = np.empty_like(vel4)
    with np.errstate(all="ignore"):
        for i in range(cnt):
            pi, vi = pos4[i], vel4[i]
            rho_i = rho[i]
            p_i = k * (rho_i - rho0)
            fp = [zero, zero, zero]
            fv = [zero, zero, zero]
            for j in _neighbours(d, lists, pi):
                if j == i:
                    continue
                rho_j = rho[j]
                p_j = k * (rho_j - rho0)
                r = [pi[a] - pos4[j, a] for a in range(3)]
                if rho_j > EPS:
                    abs_r = np.sqrt((r[0] * r[0] + r[1] * r[1]) + r[2] * r[2])
                    diff = h - abs_r
                    cp = (p_i + p_j) / (f32(2.0) * rho_j)
                    if EPS < diff and EPS < abs_r:
                        s = (c / h6) * ((diff * diff) * diff)
                        g = [(s * r[a]) / abs_r for a in range(3)]
                    else:
                        g = [zero, zero, zero]
                    gv = (c / h6) * (h - abs_r) if abs_r < h else zero
                    for a in range(3):
                        fp[a] = fp[a] + cp * g[a]
                        fv[a] = fv[a] + ((vel4[j, a] - vi[a]) / rho_j) * gv
            if rho_i > EPS:
                sp = m * m / rho_i
                sv = mu * m * m / rho_i
                fp = [x * sp for x in fp]
                fv = [x * sv for x in fv]
            r = [zero, zero, zero]
            for a in range(3):
                if pi[a] < h:
                    r[a] = r[a] + (h - pi[a])
                elif pi[a] > one - h:
                    r[a] = r[a] + (one - pi[a] - h)
            max_r = max(abs(float(r[0])), abs(float(r[1])), abs(float(r[2])))
            damp = [kd if r[a] != zero else zero for a in range(3)]
            fw = [zero, zero, zero]
            if max_r > 0.0:
                dd = (damp[0] * vi[0] + damp[1] * vi[1]) + damp[2] * vi[2]
                fw = [(r[a] * ks - dd) * m for a in range(3)]
            g = [zero, f32(-9.8), zero]
            acc = [g[a] + ((fp[a] + fv[a]) + fw[a]) / m for a in range(3)]
            if any(math.isnan(float(x)) for x in acc):
                vo = [zero, zero, zero, f32(0.003)]
            else:
                vo = [vi[a] + acc[a] * dt for a in range(3)] + [vi[3] + one]
            pw = [pi[0], pi[1], pi[2], one]
            for a in range(4):
                pos_o[i, a] = _clamp01(pw[a] + dt * vo[a])
                vel_o[i, a] = vo[a]
    return pos_o, vel_o


def update(d, pos4, vel4, delta_time, mu=0.01, k=150.0, rho0=1.5, ks=5000.0, kd=10.0,
           substeps=5, divisor=25.0):
    """SphFluidSimulation.cs:96-102: grid + density once, then 5 stale-grid sub-steps."""
    pos4 = np.array(pos4, np.float32)
    vel4 = np.array(vel4, np.float32)
    lists = cell_lists(d, pos4)
    rho = density(d, pos4, lists)
    dt = f32(delta_time) / f32(divisor)
    for _ in range(substeps):
        pos4, vel4 = velpos(d, pos4, vel4, rho, lists, dt, mu, k, rho0, ks, kd)
    return pos4, vel4, rho
