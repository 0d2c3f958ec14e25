"""Known-answer tests that pin the CPU oracle (oracle/sph_oracle.c).

The reference ships no tests or golden vectors, so every expectation here is
derived analytically from the shader text; file:line citations are relative to /root/reference.
K1..K11 follow SURVEY.md section 8c.
"""
import math

import numpy as np
import pytest

from oracle import sph_oracle_py as P
from sphfluidsimulation_amd import presets

PI = math.pi


def make_sim(oracle, n, R, fill=0.8, substeps=None, **kw):
    p = oracle.derive(n, R, fill, **kw)
    if substeps is not None:
        p.substeps = substeps
    return oracle.OracleSimulation(p), p


def parked(n):
    """n isolated interior particles: one per cell on a coarse lattice, > h from walls and each
    other when R = 9 (h = 1/8)."""
    pos = np.zeros((n, 4), np.float32)
    k = 0
    for a in range(3):
        for b in range(3):
            for c in range(3):
                if k < n:
                    pos[k, :3] = (0.2 + 0.3 * a, 0.2 + 0.3 * b, 0.2 + 0.3 * c)
                    k += 1
    pos[:, 3] = 1.0
    assert k == n
    return pos


def test_derived_parameters(oracle):
    # SphFluidSimulation.cs:84-85,159,176,236-237,255 ; values from SURVEY.md section 8d
    p = oracle.derive(262144, 47, 0.8)
    assert (p.num_particles, p.res) == (262144, 512)
    assert p.h == np.float32(1.0) / np.float32(46)
    assert p.particle_mass == np.float32(0.8) / np.float32(262144)
    assert p.h2 == np.float32(p.h) * np.float32(p.h)
    assert p.h6 == np.float32(float(np.float32(p.h)) ** 6)
    assert p.h9 == np.float32(float(np.float32(p.h)) ** 9)
    # NextPowerOfTwo + (int)sqrt: odd powers of two leave res*res < N
    p = oracle.derive(1500, 9, 0.5)
    assert (p.num_particles, p.res) == (2048, 45)
    p = oracle.derive(4096, 13, 0.8)
    assert abs(p.h - 0.083333336) < 1e-9 and abs(p.particle_mass - 1.953125e-4) < 1e-11


def test_shader_constants_are_fp32():
    # Density.compute:23 and VelPos.compute:34 evaluated in fp32
    c6 = np.float32(315.0) / (np.float32(64.0) * np.float32(PI))
    c45 = np.float32(45.0) / np.float32(PI)
    assert abs(float(c6) - 1.5666815) < 2e-7
    assert abs(float(c45) - 14.323945) < 2e-6


@pytest.mark.parametrize("n,R,expected", [(262144, 47, 0.46538), (4096, 13, 0.52875)])
def test_k1_isolated_density(oracle, n, R, expected):
    # Density.compute:22-27,54: rho = m*c*(h^2)^3/h^9 = m*c/h^3 for a lone particle (self term)
    p = oracle.derive(n, R, 0.8)
    cnt = p.res * p.res
    pos = np.full((cnt, 4), 2.5, np.float32)     # everything else out of the grid -> dropped
    pos[0] = (0.5, 0.5, 0.5, 1.0)
    b = oracle.bucket_generation(p, pos)
    rho = oracle.density(p, pos, b)
    m, h = 0.8 / n, 1.0 / (R - 1)
    analytic = m * 315.0 / (64.0 * PI) / h ** 3
    assert abs(rho[0] / analytic - 1) < 2e-6
    assert abs(rho[0] - expected) < 1e-5
    assert np.all(rho[1:] == 0.0)                # dropped particles see nobody


def test_k2_free_fall_semi_implicit(oracle):
    # VelPos.compute:139,150,153: v += g*dt ; pos += dt*v_new
    sim, p = make_sim(oracle, 16, 9, substeps=1)
    pos = parked(16)
    vel = np.zeros((16, 4), np.float32)
    sim.upload(pos, vel)
    dt_frame = 0.05
    sim.Update(dt_frame)
    po, vo, _ = sim.download()
    dt = np.float32(dt_frame) / np.float32(25)
    vy = np.float32(-9.8) * dt
    assert np.all(vo[:, 0] == 0) and np.all(vo[:, 2] == 0)
    assert np.all(vo[:, 1] == vy)
    assert np.all(po[:, 1] == pos[:, 1] + dt * vy)
    assert np.all(po[:, 0] == pos[:, 0]) and np.all(po[:, 2] == pos[:, 2])


def _pair(d, axis=0):
    pos = parked(16)
    pos[1, :3] = pos[0, :3]
    pos[1, axis] += d
    return pos


def test_k3_pressure_pair(oracle):
    # VelPos.compute:37,93,103: equal and opposite, magnitude
    # ((p_i+p_j)/(2 rho_j)) * (45/pi/h^6) (h-d)^3 * m^2/rho_i   (note: CUBED, positive sign)
    sim, p = make_sim(oracle, 16, 9, substeps=1, viscosity=0.0)
    h, m = float(p.h), float(p.particle_mass)
    d = 0.4 * h
    pos = _pair(d)
    sim.upload(pos, np.zeros((16, 4), np.float32))
    sim.Update(0.025)
    _, vo, rho = sim.download()
    dt = 0.025 / 25
    c6 = 315.0 / (64 * PI)
    rho_pair = m * c6 * (h ** 6 + (h * h - d * d) ** 3) / h ** 9
    assert abs(rho[0] / rho_pair - 1) < 1e-5 and rho[0] == rho[1]
    pr = 150.0 * (rho_pair - 1.5)
    f = ((pr + pr) / (2 * rho_pair)) * (45 / PI / h ** 6) * (h - d) ** 3 * m * m / rho_pair
    # r_ij = pos_i - pos_j: particle 0 sits at smaller x -> direction -1 for particle 0
    assert abs(vo[0, 0] / (-f / m * dt) - 1) < 1e-4
    assert vo[0, 0] == -vo[1, 0]
    assert vo[0, 1] == vo[1, 1] == np.float32(-9.8) * np.float32(np.float32(0.025) / np.float32(25))
    assert vo[0, 2] == 0 and vo[1, 2] == 0


def test_k4_viscosity_pair(oracle):
    # VelPos.compute:43,94,104: f_vis = mu*m^2/rho_i * (v_j - v_i)/rho_j * (45/pi/h^6)(h-d); 0 for d>=h
    out = {}
    for mu in (0.0, 0.05):
        sim, p = make_sim(oracle, 16, 9, substeps=1, viscosity=mu)
        h, m = float(p.h), float(p.particle_mass)
        d = 0.5 * h
        pos = _pair(d)
        vel = np.zeros((16, 4), np.float32)
        vel[1, 2] = 0.25                                     # relative velocity along z
        sim.upload(pos, vel)
        sim.Update(0.025)
        _, vo, rho = sim.download()
        out[mu] = (vo.copy(), rho.copy())
    dt = 0.025 / 25
    rho_p = float(out[0.05][1][0])
    f = 0.05 * m * m / rho_p * (0.25 / rho_p) * (45 / PI / h ** 6) * (h - d)
    dv = out[0.05][0] - out[0.0][0]
    assert abs(dv[0, 2] / (f / m * dt) - 1) < 1e-3          # particle 0 is dragged along +z
    assert abs(dv[1, 2] / (-f / m * dt) - 1) < 1e-3         # particle 1 is slowed down
    assert np.all(dv[2:] == 0)
    # beyond the support: no coupling at all (only gravity), d = 1.01 h, still adjacent cells
    sim, p = make_sim(oracle, 16, 9, substeps=1, viscosity=0.05)
    pos = _pair(1.01 * float(p.h))
    vel = np.zeros((16, 4), np.float32)
    vel[1, 2] = 0.25
    sim.upload(pos, vel)
    sim.Update(0.025)
    _, vo, _ = sim.download()
    assert vo[0, 0] == 0 and vo[0, 2] == 0 and vo[1, 2] == np.float32(0.25)


def test_k5_wall_force_and_scalar_damping(oracle):
    # VelPos.compute:110-137: r.x = h - x for x<h ; f_wall = (r*k_s - dot(damp,v))*m where the dot
    # product is a SCALAR subtracted from all three components
    sim, p = make_sim(oracle, 16, 9, substeps=1)
    h = float(p.h)
    pos = parked(16)
    pos[0, 0] = 0.25 * h
    pos[1, 0] = 0.25 * h
    vel = np.zeros((16, 4), np.float32)
    vel[1, :3] = (0.3, 0.1, -0.2)
    sim.upload(pos, vel)
    sim.Update(0.025)
    _, vo, _ = sim.download()
    dt = 0.025 / 25
    ax = 5000.0 * (h - 0.25 * h)
    assert abs(vo[0, 0] / (ax * dt) - 1) < 1e-5
    assert abs(vo[0, 1] / (-9.8 * dt) - 1) < 1e-6 and vo[0, 2] == 0
    dd = 10.0 * 0.3                                          # damp = (k_d, 0, 0)
    assert abs((vo[1, 0] - 0.3) / ((ax - dd) * dt) - 1) < 1e-4
    assert abs((vo[1, 1] - 0.1) / ((-9.8 - dd) * dt) - 1) < 1e-4
    assert abs((vo[1, 2] + 0.2) / ((-dd) * dt) - 1) < 1e-4  # z has no penetration, still damped
    # far wall: r.x = 1 - x - h (negative)
    sim, p = make_sim(oracle, 16, 9, substeps=1)
    pos = parked(16)
    pos[0, 0] = 1.0 - 0.5 * h
    sim.upload(pos, np.zeros((16, 4), np.float32))
    sim.Update(0.025)
    _, vo, _ = sim.download()
    assert abs(vo[0, 0] / (5000.0 * (-0.5 * h) * dt) - 1) < 1e-4


def test_k6_w_lanes(oracle):
    # VelPos.compute:150-154: vel.w counts sub-steps, pos.w == 1
    sim, p = make_sim(oracle, 4096, 13)
    pos, vel = presets.dam_break(4096, 0.8, seed=3)
    sim.upload(pos, vel)
    sim.Update(0.02)
    po, vo, _ = sim.download()
    assert np.all(po[:, 3] == 1.0) and np.all(vo[:, 3] == 5.0)
    sim.Update(0.02)
    po, vo, _ = sim.download()
    assert np.all(po[:, 3] == 1.0) and np.all(vo[:, 3] == 10.0)
    assert po[:, :3].min() >= 0.0 and po[:, :3].max() <= 1.0


def test_k7_nan_guard(oracle):
    # VelPos.compute:143-147: NaN acceleration -> a = 0, vel = (0,0,0,0.003); position unchanged
    sim, p = make_sim(oracle, 16, 9, substeps=1)
    h = float(p.h)
    pos = _pair(0.5 * h)
    vel = np.zeros((16, 4), np.float32)
    vel[1, 0] = np.nan
    vel[5, 1] = np.nan                                       # isolated particle with NaN velocity
    sim.upload(pos, vel)
    sim.Update(0.025)
    po, vo, _ = sim.download()
    for i in (0, 1):                                         # both see (v_j - v_i) = NaN
        assert np.array_equal(vo[i], np.array([0, 0, 0, 0.003], np.float32))
        assert np.array_equal(po[i], pos[i])
    # isolated: acceleration is finite (gravity), so the NaN velocity is NOT trapped;
    # pos.y = clamp(NaN) = 0 (D3D max/min semantics)
    assert np.isnan(vo[5, 1]) and vo[5, 3] == 1.0 and po[5, 1] == 0.0
    # a NaN candidate poisons every lister inside the 27 cells, even beyond the radius h
    sim, p = make_sim(oracle, 16, 9, substeps=1)
    pos = _pair(1.2 * h)
    vel = np.zeros((16, 4), np.float32)
    vel[1, 0] = np.nan
    sim.upload(pos, vel)
    sim.Update(0.025)
    _, vo, _ = sim.download()
    assert np.array_equal(vo[0], np.array([0, 0, 0, 0.003], np.float32))


def test_k8_clamp_to_unit_box(oracle):
    # VelPos.compute:154
    sim, p = make_sim(oracle, 16, 9, substeps=1)
    pos = parked(16)
    vel = np.zeros((16, 4), np.float32)
    vel[0, 0] = 5000.0
    vel[1, 2] = -5000.0
    sim.upload(pos, vel)
    sim.Update(0.025)
    po, _, _ = sim.download()
    assert po[0, 0] == 1.0 and po[1, 2] == 0.0


def test_k9_cell_overflow_keeps_32_lowest_ids(oracle):
    # Bucket.compute:30-35 + SURVEY.md 9.2: 40 particles in one cell; ids >= 32 of that cell are
    # invisible to everybody (including their own density) but still integrate
    n, R = 64, 4
    p = oracle.derive(n, R, 0.8)
    h, m = float(p.h), float(p.particle_mass)
    rng = np.random.default_rng(7)
    pos = np.full((n, 4), 2.5, np.float32)
    crowd = np.arange(10, 50)                                # 40 ids share cell (1,1,1)
    pos[crowd, :3] = (h * (1.2 + 0.6 * rng.random((40, 3)))).astype(np.float32)
    pos[:, 3] = 1.0
    b = oracle.bucket_generation(p, pos).reshape(-1, 32)
    key = 1 + 1 * R + 1 * R * R
    assert np.array_equal(b[key], np.arange(10, 42, dtype=np.uint32))
    rho = oracle.density(p, pos, b.reshape(-1))
    vis = np.arange(10, 42)
    c6 = 315.0 / (64 * PI)
    for i in (10, 25, 41, 42, 49):
        d2 = ((pos[i, :3].astype(np.float64) - pos[vis, :3].astype(np.float64)) ** 2).sum(1)
        w = np.where(h * h - d2 > 0, c6 * (h * h - d2) ** 3 / h ** 9, 0.0)
        assert abs(rho[i] / (m * w.sum()) - 1) < 1e-5
    sim = oracle.OracleSimulation(p)
    sim.upload(pos, np.zeros((n, 4), np.float32))
    sim.Update(0.01)
    _, vo, _ = sim.download()
    assert np.all(vo[crowd, 3] == 5.0)                       # hidden ones still run as "i"


def test_k10_truncation_not_floor(oracle):
    # Bucket.compute:27: int3() truncates toward zero: -0.3/(R-1) -> cell 0, not -1
    n, R = 16, 9
    p = oracle.derive(n, R, 0.8)
    pos = np.full((n, 4), 2.5, np.float32)
    pos[3, :3] = (-0.3 / (R - 1), 0.5, 0.5)
    pos[4, :3] = (0.2 / (R - 1), 0.5, 0.5)
    b = oracle.bucket_generation(p, pos).reshape(-1, 32)
    key = 0 + 4 * R + 4 * R * R
    assert list(b[key][:3]) == [3, 4, n]
    # a coordinate index == R aliases into the next row: x = 1.13 -> ix = 9 = R
    pos[5, :3] = (1.13, 0.5, 0.5)
    b = oracle.bucket_generation(p, pos).reshape(-1, 32)
    assert b[0 + 5 * R + 4 * R * R][0] == 5
    # and the linear index past the end of the buffer is dropped
    pos[6, :3] = (0.5, 0.5, 1.13)
    b = oracle.bucket_generation(p, pos).reshape(-1, 32)
    assert not np.any(b == 6)


def test_k11_stale_bucket_within_frame(oracle):
    # SphFluidSimulation.cs:98-102: the grid is built once per frame.  Two particles three cells
    # apart fly towards each other; after sub-step 1 they are within h, but each one's 27-cell
    # neighbourhood only lists FRAME-START members, so they do not interact in sub-step 2.
    sim, p = make_sim(oracle, 16, 9, substeps=2, viscosity=0.0)
    h = float(p.h)
    pos = parked(16)
    pos[0, :3] = (0.5 - 1.6 * h, 0.35, 0.35)                # clear of the parked lattice
    pos[1, :3] = (0.5 + 1.6 * h, 0.35, 0.35)
    dt_frame = 0.025
    dt = np.float32(dt_frame) / np.float32(25)
    v = np.float32(1.4 * h / float(dt))
    vel = np.zeros((16, 4), np.float32)
    vel[0, 0], vel[1, 0] = v, -v
    sim.upload(pos, vel)
    sim.Update(dt_frame)
    po, vo, _ = sim.download()
    assert vo[0, 0] == v and vo[1, 0] == -v                  # x velocity untouched: no pair force
    # control: rebuilding the grid between the sub-steps does couple them
    p1 = oracle.derive(16, 9, 0.8, viscosity=0.0)
    b = oracle.bucket_generation(p1, pos)
    rho = oracle.density(p1, pos, b)
    q, w = oracle.velpos(p1, pos, vel, rho, b, dt)
    assert abs(q[1, 0] - q[0, 0]) < h                        # within h when sub-step 2 starts
    b2 = oracle.bucket_generation(p1, q)
    q2, w2 = oracle.velpos(p1, q, w, rho, b2, dt)
    assert w2[0, 0] != v and w2[0, 0] == -w2[1, 0]


def test_thread_count_does_not_change_bits(oracle):
    pos, vel = presets.dam_break(4096, 0.8, seed=5)
    outs = []
    for nt in (1, 4):
        sim, _ = make_sim(oracle, 4096, 13, num_threads=nt)
        sim.upload(pos, vel)
        for _ in range(3):
            sim.Update(presets.bench_dt_frame(13))
        outs.append(sim.download())
    for a, b in zip(*outs):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("n,R,frames,seed", [(256, 6, 3, 1), (1024, 9, 2, 2), (256, 3, 2, 4)])
def test_c_oracle_matches_pure_python_restatement(oracle, n, R, frames, seed):
    """Slot-table/CAS formulation (C) == stable-sort cell lists truncated to 32 (Python),
    bit for bit, including cells that overflow (R=3 packs ~40 particles per cell)."""
    fill = 0.8
    p = oracle.derive(n, R, fill)
    d = P.derive(n, R, fill)
    pos, vel = presets.dam_break(n, fill, seed=seed)
    sim = oracle.OracleSimulation(p)
    sim.upload(pos, vel)
    dt = presets.bench_dt_frame(R) * 0.5
    pp, vv = pos, vel
    for _ in range(frames):
        sim.Update(dt)
        pp, vv, rr = P.update(d, pp, vv, dt)
        po, vo, ro = sim.download()
        assert np.array_equal(po.view(np.uint32), pp.view(np.uint32))
        assert np.array_equal(vo.view(np.uint32), vv.view(np.uint32))
        assert np.array_equal(ro.view(np.uint32), rr.view(np.uint32))
    if R == 3:
        occ = (sim.bucket() != p.num_particles).sum(1).max()
        assert occ == 32                                     # the cap really triggered
