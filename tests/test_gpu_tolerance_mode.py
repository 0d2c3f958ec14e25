"""SphParams.exact_order = 0, the TOLERANCE MODE (BASELINE.json north_star: "match ... within 1e-4 relative fp32"; its named
technique, lists of candidates walked against an LDS tile): the same pairs, guards and summation order as the reference, with
v_rcp_f32 / v_sqrt_f32 (1 ulp) instead of IEEE division / sqrt and re-associated products (csrc/sph_sweep_common.hpp pair_fast,
density_term_r2<true>).  Bar: within 1e-4 of the CPU oracle after one sub-step (every particle) and after one frame (the bulk:
the reference's grid makes the result discontinuous in the positions, see tests/test_gpu_precision.py); the engine's own routes
must still agree with each other BIT FOR BIT in this mode (they share the arithmetic and the order).

The kernel this mode is built around, k_velpos_tile (lists of LDS tile indices), also runs with the reference's IEEE arithmetic
(the default list format): there it must be bit-identical to the oracle like every other route."""
import numpy as np
import pytest

from sphfluidsimulation_amd import presets
from tests.test_gpu_parity import SCENE, assert_parity

pytestmark = pytest.mark.gpu


def engine(n, R, **kw):
    from sphfluidsimulation_amd.simulation import SphFluidSimulation
    return SphFluidSimulation(particleNumber=n, bucketResolution=R, damFillRate=0.8, **kw)


def rel_err(got, want):
    """max |got - want| / max(1, max |want|) per particle over x, y, z."""
    scale = max(1.0, float(np.abs(want[:, :3]).max()))
    return np.abs(got[:, :3].astype(np.float64) - want[:, :3]).max(1) / scale


@pytest.mark.parametrize("n,R", [(4096, 13), (262144, 47)])
def test_tolerance_mode_within_1e4_of_the_oracle(oracle, n, R):
    pos0, vel0 = presets.dam_break(n, 0.8, 1)
    dt = presets.bench_dt_frame(R)
    ora = oracle.OracleSimulation(oracle.derive(n, R, 0.8, **SCENE))
    ora.upload(pos0, vel0)
    sim = engine(n, R, exact_order=False, lists="always").Start(pos0, vel0)
    try:
        # one sub-step from identical inputs: EVERY particle within 1e-4 (in fact ~1e-6)
        ora.BucketGeneration(); ora.DensityCalculation(); ora.UpdateVelocityAndPosition(np.float32(dt) / np.float32(25))
        sim.BucketGeneration(); sim.DensityCalculation(); sim.UpdateVelocityAndPosition(np.float32(dt) / np.float32(25))
        po, vo, ro = ora.download()
        p, v, r = sim.download()
        ep, ev = rel_err(p, po), rel_err(v, vo)
        er = np.abs(r.astype(np.float64) - ro) / np.maximum(np.abs(ro), 1e-6)
        assert ep.max() <= 1e-4 and ev.max() <= 1e-4 and er.max() <= 1e-4, (ep.max(), ev.max(), er.max())
        assert np.array_equal(v[:, 3], vo[:, 3]) and np.array_equal(p[:, 3], po[:, 3])
        one = (ep.max(), ev.max(), er.max())
        # one frame (5 sub-steps on a frozen grid): the bulk within 1e-4, the grid's discontinuity outliers counted
        ora.upload(pos0, vel0)
        sim.upload(pos0, vel0)
        ora.Update(dt)
        sim.Update(dt)
        po, vo, ro = ora.download()
        p, v, r = sim.download()
        ep, ev = rel_err(p, po), rel_err(v, vo)
        out1 = int((np.maximum(ep, ev) > 1e-4).sum())
        assert np.percentile(ep, 99.99) <= 1e-4 and np.percentile(ev, 99.9) <= 1e-4, (np.percentile(ep, 99.99), np.percentile(ev, 99.9))
        assert out1 <= max(2, n // 2000), out1
        assert np.array_equal(v[:, 3], vo[:, 3]) and np.array_equal(p[:, 3], po[:, 3])
        print(f"n={n}: 1 sub-step max rel err pos {one[0]:.2e} vel {one[1]:.2e} rho {one[2]:.2e}; 1 frame: pos median {np.median(ep):.2e} "
              f"max {ep.max():.2e}, vel median {np.median(ev):.2e} 99.9 % {np.percentile(ev, 99.9):.2e} max {ev.max():.2e}; "
              f"{out1} of {n} particles beyond 1e-4")
    finally:
        sim.OnDestroy()


def test_tolerance_mode_routes_agree_bit_for_bit():
    """Lists of tile indices, lists off (scanning bricks), per-target kernels, issued launches, several slabs: in tolerance mode
    too, the route a target takes cannot change a bit -- an out-of-range candidate contributes fma(0, r, f) = f."""
    n, R = 262144, 47
    pos0, vel0 = presets.dam_break(n, 0.8, 5)
    vel0[:, :3] = np.random.default_rng(3).normal(0, 0.5, (n, 3)).astype(np.float32)    # (cell changes from the first sub-step on)
    dt = presets.bench_dt_frame(R)
    outs = []
    for kw in (dict(lists="always"), dict(lists=False), dict(tiling=False), dict(lists="always", graphs=False),
               dict(lists="always", device_ids=[0, 0, 0])):
        sim = engine(n, R, exact_order=False, **kw).Start(pos0, vel0)
        sim.step_n(dt, 6)
        outs.append(sim.download())
        if kw.get("lists") == "always" and "device_ids" not in kw:
            st = sim.stats()
            assert st["listed_target_substeps"] > n, st["listed_target_substeps"]
        sim.OnDestroy()
    for k, o in enumerate(outs[1:]):
        for a, b, what in zip(outs[0], o, ("pos", "vel", "rho")):
            assert np.array_equal(a, b, equal_nan=True), f"route {k + 1}: {what} differs"


@pytest.mark.parametrize("n,R,frames", [(4096, 13, 10), (262144, 47, 6)])
def test_tile_lists_with_exact_arithmetic_match_the_oracle(oracle, n, R, frames):
    """k_velpos_tile + the tile-index lists with exact_order = 1 (the default list format): bit-identical to the oracle, with
    targets that change cell, over-full cells and one process split into slabs."""
    pos0, vel0 = presets.dam_break(n, 0.8, 2)
    vel0[:, :3] = np.random.default_rng(4).normal(0, 0.7, (n, 3)).astype(np.float32)
    dt = presets.bench_dt_frame(R)
    ora = oracle.OracleSimulation(oracle.derive(n, R, 0.8, **SCENE))
    ora.upload(pos0, vel0)
    sims = [engine(n, R, tile_lists=True, lists="always").Start(pos0, vel0)]
    if n >= 262144:
        sims.append(engine(n, R, tile_lists=True, lists="always", device_ids=[0, 0]).Start(pos0, vel0))
    try:
        for f in range(1, frames + 1):
            ora.Update(dt)
            for sim in sims:
                sim.Update(dt)
            if f in (1, frames):
                want = ora.download()
                for k, sim in enumerate(sims):
                    for a, b, what in zip(sim.download(), want, ("pos", "vel", "rho")):
                        assert_parity(a, b, f"tile lists, engine {k}, frame {f}: {what}")
        st = sims[0].stats()
        if n >= 262144:
            assert st["listed_target_substeps"] > n, st["listed_target_substeps"]
    finally:
        for sim in sims:
            sim.OnDestroy()


def test_tile_lists_crowded_cells_and_precision_1():
    """Cells beyond the 32-cap (their surplus members are targets but never candidates, and never in the tile) and over-full
    tiles (their targets keep lists of global slots): tile lists == no lists, bit for bit; the same with fp16 position records."""
    n, R = 65536, 12                                     # ~38 per cell: every cell over the cap, every tile over-full or halved
    pos0, vel0 = presets.dam_break(n, 0.8, 7)
    dt = presets.bench_dt_frame(R)
    for prec in (0, 1):
        outs = []
        for kw in (dict(tile_lists=True, lists="always"), dict(lists=False)):
            sim = engine(n, R, precision=prec, **kw).Start(pos0, vel0)
            sim.step_n(dt, 4)
            outs.append(sim.download())
            sim.OnDestroy()
        for a, b in zip(*outs):
            assert np.array_equal(a, b, equal_nan=True), f"precision {prec}"
    n, R = 262144, 47
    pos0, vel0 = presets.dam_break(n, 0.8, 7)
    dt = presets.bench_dt_frame(R)
    outs = []
    for kw in (dict(tile_lists=True, lists="always"), dict(lists=False), dict(exact_order=False, lists="always"), dict(exact_order=False, lists=False)):
        sim = engine(n, R, precision=1, **kw).Start(pos0, vel0)
        sim.step_n(dt, 5)
        outs.append(sim.download())
        sim.OnDestroy()
    for a, b in zip(outs[0], outs[1]):
        assert np.array_equal(a, b, equal_nan=True), "precision 1, exact arithmetic"
    for a, b in zip(outs[2], outs[3]):
        assert np.array_equal(a, b, equal_nan=True), "precision 1, tolerance mode"


def test_tolerance_mode_full_size_invariants():
    """16 777 216 particles, R = 185 (the bench configuration) in tolerance mode: three frames, then what the reference's
    semantics guarantee -- nobody lost, unit box, vel.w counts the sub-steps, no NaN, densities positive."""
    n, R = 16777216, 185
    pos0, vel0 = presets.dam_break(n, 0.8, 1)
    sim = engine(n, R, exact_order=False).Start(pos0, vel0)
    try:
        sim.step_n(presets.bench_dt_frame(R), 3)
        p, v, r = sim.download()
        st = sim.stats()
        assert p.shape[0] == n and np.all(p[:, :3] >= 0) and np.all(p[:, :3] <= 1)
        assert not np.isnan(p).any() and not np.isnan(v).any() and np.all(r > 0)
        assert np.all(v[:, 3] == 15.0)                   # (vel.w: +1 per sub-step, VelPos.compute:150)
        assert st["nan_resets"] == 0 and st["listed_target_substeps"] > 0.5 * n * 15
    finally:
        sim.OnDestroy()


def test_global_slot_lists_still_match(oracle):
    """SPH_FLAG_GLOBAL_LISTS (the list format and kernel of rounds 1-2, kept as a diagnostic and for the targets of over-full
    tiles): bit-identical to the oracle with the reference's arithmetic, bit-identical to the tile lists in tolerance mode."""
    n, R, frames = 262144, 47, 4
    pos0, vel0 = presets.dam_break(n, 0.8, 9)
    vel0[:, :3] = np.random.default_rng(8).normal(0, 0.4, (n, 3)).astype(np.float32)
    dt = presets.bench_dt_frame(R)
    ora = oracle.OracleSimulation(oracle.derive(n, R, 0.8, **SCENE))
    ora.upload(pos0, vel0)
    sim = engine(n, R, tile_lists=False, lists="always").Start(pos0, vel0)
    try:
        for _ in range(frames):
            ora.Update(dt)
            sim.Update(dt)
        for a, b, what in zip(sim.download(), ora.download(), ("pos", "vel", "rho")):
            assert_parity(a, b, f"lists of global slots, frame {frames}: {what}")
        assert sim.stats()["listed_target_substeps"] > n
    finally:
        sim.OnDestroy()
    outs = []
    for tile in (True, False):
        sim = engine(n, R, exact_order=False, tile_lists=tile, lists="always").Start(pos0, vel0)
        sim.step_n(dt, frames)
        outs.append(sim.download())
        sim.OnDestroy()
    for a, b in zip(*outs):
        assert np.array_equal(a, b, equal_nan=True)


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("SPH_FUZZ_SEEDS", "10"))))
def test_random_configurations_tolerance_routes(seed):
    """Random sizes, grids, fill rates, speeds and scalars in tolerance mode: tile lists, lists of global slots and no lists at
    all give the same bits (whatever mixture of list, scanning, hand-over and generic kernels each ends up on)."""
    rng = np.random.default_rng(5000 + seed)
    n = int(rng.choice([1024, 4096, 16384, 65536]))
    R = int(rng.integers(4, 40)) if seed % 3 else int(rng.choice([2, 3, 64, 128]))
    fill = float(rng.choice([0.3, 0.5, 0.8, 1.0]))
    speed = float(rng.choice([0.0, 0.05, 0.5, 5.0]))
    scal = dict(viscosity=float(rng.uniform(0.0, 0.1)), restDensity=float(rng.uniform(0.5, 3.0)),
                gasConstant=float(rng.uniform(10.0, 500.0)), stiffnessCoefficient=float(rng.uniform(1000.0, 10000.0)),
                dampingCoefficient=float(rng.uniform(1.0, 50.0)))
    pos0, vel0 = presets.dam_break(n, fill, seed=seed, jitter=float(rng.choice([0.0, 0.3, 1.0])))
    vel0[:, :3] = rng.standard_normal((n, 3)).astype(np.float32) * np.float32(speed)
    dt = float(rng.uniform(0.05, 1.0)) / (R - 1 if R > 1 else 1) * 0.7666667
    frames = int(rng.integers(3, 8))
    outs = []
    for kw in (dict(lists="always"), dict(lists="always", tile_lists=False), dict(lists=False)):
        sim = engine(n, R, exact_order=False, **dict(kw, **scal))
        sim.damFillRate = fill
        sim.Start(pos0, vel0)
        sim.step_n(dt, frames)
        outs.append(sim.download())
        sim.OnDestroy()
    for k, o in enumerate(outs[1:]):
        for a, b, what in zip(outs[0], o, ("pos", "vel", "rho")):
            assert np.array_equal(a, b, equal_nan=True), f"route {k + 1}: {what}: n={n} R={R} fill={fill} speed={speed} frames={frames}"


def test_mode_and_list_format_switch_mid_run(oracle):
    """sph_set_params between the passes of a frame and between frames: switching the list format retires the lists of the
    frame in flight (every target is still integrated once); switching exact_order changes the arithmetic of the next pass."""
    from sphfluidsimulation_amd import _native as N
    n, R = 262144, 47
    pos0, vel0 = presets.dam_break(n, 0.8, 11)
    dt = presets.bench_dt_frame(R)
    ora = oracle.OracleSimulation(oracle.derive(n, R, 0.8, **SCENE))
    ora.upload(pos0, vel0)
    sim = engine(n, R, lists="always").Start(pos0, vel0)
    try:
        for _ in range(3):
            sim.Update(dt)
            ora.Update(dt)
        # a frame by hand, the list format flipped after the density pass has recorded tile lists
        sub = np.float32(dt) / np.float32(25)
        sim.BucketGeneration(); sim.DensityCalculation()
        sim.tile_lists = False
        sim.set_params()
        for _ in range(5):
            sim.UpdateVelocityAndPosition(sub)
        ora.Update(dt)
        for _ in range(2):                               # ... and two more frames on lists of global slots
            sim.Update(dt)
            ora.Update(dt)
        for a, b, what in zip(sim.download(), ora.download(), ("pos", "vel", "rho")):
            assert_parity(a, b, f"after the format switch: {what}")
        # tolerance mode from here on == a tolerance-mode engine started from this state
        state = sim.download()
        sim.exact_order = False
        sim.tile_lists = True
        sim.set_params()
        sim.step_n(dt, 2)
        ref = engine(n, R, exact_order=False, lists="always").Start(state[0], state[1])
        ref.step_n(dt, 2)
        for a, b in zip(sim.download(), ref.download()):
            assert np.array_equal(a, b, equal_nan=True)
        ref.OnDestroy()
    finally:
        sim.OnDestroy()
