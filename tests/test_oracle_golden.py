"""The CPU oracle must reproduce the committed golden fixtures bit for bit
(tests/golden/make_golden.py regenerates them; SURVEY.md section 8c explains why they come from
this repo's own restatement: the reference ships none)."""
import hashlib
import os

import numpy as np

from sphfluidsimulation_amd import presets

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SCENE = dict(viscosity=0.01, rest_density=1.5, gas_constant=150.0, stiffness=5000.0, damping=10.0)


def _eq(a, b):
    return np.array_equal(np.asarray(a).view(np.uint32), np.asarray(b).view(np.uint32))


def test_c0_golden(oracle):
    g = np.load(os.path.join(GOLD, "c0_dambreak_seed1.npz"))
    n, R = int(g["n"]), int(g["R"])
    pos, vel = presets.dam_break(n, float(g["fill"]), int(g["seed"]))
    assert _eq(pos, g["pos_in"]) and _eq(vel, g["vel_in"])          # the generator is pinned too
    assert np.float32(presets.bench_dt_frame(R)) == g["dt_frame"]
    p = oracle.derive(n, R, float(g["fill"]), **SCENE)
    b = oracle.bucket_generation(p, pos)
    rho = oracle.density(p, pos, b)
    assert _eq(rho, g["rho_pass"])
    po, vo = oracle.velpos(p, pos, vel, rho, b, np.float32(g["dt_frame"]) / np.float32(25))
    assert _eq(po, g["pos_sub1"]) and _eq(vo, g["vel_sub1"])
    sim = oracle.OracleSimulation(p)
    sim.upload(pos, vel)
    for f in range(1, 11):
        sim.Update(float(g["dt_frame"]))
        if f in (1, 10):
            po, vo, ro = sim.download()
            assert _eq(po, g[f"pos_f{f}"]) and _eq(vo, g[f"vel_f{f}"]) and _eq(ro, g[f"rho_f{f}"])


def test_c1_golden_subset(oracle):
    g = np.load(os.path.join(GOLD, "c1_dambreak_seed1_subset.npz"))
    n, R = int(g["n"]), int(g["R"])
    pos, vel = presets.dam_break(n, float(g["fill"]), int(g["seed"]))
    sim = oracle.OracleSimulation(oracle.derive(n, R, float(g["fill"]), **SCENE))
    sim.upload(pos, vel)
    sim.Update(float(g["dt_frame"]))
    po, vo, ro = sim.download()
    ids = g["ids"]
    assert _eq(po[ids], g["pos_f1"]) and _eq(vo[ids], g["vel_f1"]) and _eq(ro[ids], g["rho_f1"])
    sha = "".join(hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest() for a in (po, vo, ro))
    assert sha == str(g["sha256_f1"])
