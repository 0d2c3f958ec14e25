"""Frame-coherent neighbour lists (sphfluidsimulation_amd/csrc/sph_kernels_list.hip): the density pass records, per
target, the candidates within h + skin; the five VelPos sub-steps of the frame walk those lists where the displacement
bounds allow and fall back to the scanning kernels elsewhere.  Whatever path serves a target, the result must be the
oracle's, bit for bit -- these cases are built to make every path run (checked through SphStats)."""
import numpy as np
import pytest

from sphfluidsimulation_amd import presets

pytestmark = pytest.mark.gpu
f32 = np.float32


def _bits(a):
    return np.ascontiguousarray(a, dtype=f32).view(np.uint32)


def _run_pair(n, R, fill, pos0, vel0, frames, dt, **kw):
    from sphfluidsimulation_amd.simulation import SphFluidSimulation
    out = []
    for lists in ("always", False):     # "always": the policy that drops lists where they do not pay is not under test here
        sim = SphFluidSimulation(particleNumber=n, bucketResolution=R, damFillRate=fill, lists=lists, **kw).Start(pos0, vel0)
        try:
            sim.reset_stats()
            for _ in range(frames):
                sim.Update(dt)
            out.append((sim.download(), sim.stats()))
        finally:
            sim.OnDestroy()
    return out


@pytest.mark.parametrize("speed,dt_scale,expect", [(0.0, 0.02, "calm"), (0.3, 0.25, "mixed"), (4.0, 1.0, "fast")])
def test_lists_match_scanning_and_oracle(oracle, speed, dt_scale, expect):
    """Tiny time steps from rest: nearly every target sub-step is served from its list.  Random velocities and a
    larger step: some bricks re-scan, some targets change cell and are masked or handed over.  Fast and at the full
    step (a 65536-particle box is all wall layer: 5000 * h of wall acceleration): the relative displacement exceeds
    the skin after the first sub-step almost everywhere."""
    n, fill = 65536, 0.8
    R = presets.scene_bucket_resolution(n)
    pos0, vel0 = presets.dam_break(n, fill, seed=17, jitter=0.3)     # inside the unit box: frame 1 is not blocked
    rng = np.random.default_rng(3)
    vel0[:, :3] = rng.standard_normal((n, 3)).astype(f32) * f32(speed)
    dt = presets.bench_dt_frame(R) * dt_scale
    frames = 3
    (on, st_on), (off, st_off) = _run_pair(n, R, fill, pos0, vel0, frames, dt)
    for a, b in zip(on, off):
        assert np.array_equal(_bits(a), _bits(b))
    osim = oracle.OracleSimulation(oracle.derive(n, R, fill))
    osim.upload(pos0, vel0)
    for _ in range(frames):
        osim.Update(dt)
    for a, b in zip(on, osim.download()):
        assert np.array_equal(_bits(a), _bits(b))
    assert st_off["listed_target_substeps"] == 0
    frac = st_on["listed_target_substeps"] / (n * frames * 5)
    if expect == "calm":
        assert frac > 0.95
    elif expect == "mixed":
        assert 0.25 < frac < 0.98 and st_on["scanned_brick_substeps"] > 0
    else:
        assert 0.15 < frac < 0.5 and st_on["scanned_brick_substeps"] > 0     # the first sub-step is always listed


def test_lists_with_reference_dispatch_granularity(oracle):
    """The three driver methods called by hand: a second density pass mid-frame must not re-record (the particles have
    moved off the positions they were binned at), and sub-steps without a density pass since the grid build scan."""
    from sphfluidsimulation_amd.simulation import SphFluidSimulation
    n, fill = 16384, 0.8
    R = presets.scene_bucket_resolution(n)
    pos0, vel0 = presets.dam_break(n, fill, seed=23, jitter=0.3)
    vel0[:, :3] = np.random.default_rng(4).standard_normal((n, 3)).astype(f32) * f32(0.3)
    dt = presets.bench_dt_frame(R) / 25
    sim = SphFluidSimulation(particleNumber=n, bucketResolution=R, damFillRate=fill, lists="always").Start(pos0, vel0)
    osim = oracle.OracleSimulation(oracle.derive(n, R, fill))
    osim.upload(pos0, vel0)
    try:
        for s in (sim, osim):
            s.BucketGeneration()
            s.DensityCalculation()          # records
            s.UpdateVelocityAndPosition(dt)
            s.UpdateVelocityAndPosition(dt)
            s.DensityCalculation()          # must not record: positions moved
            s.UpdateVelocityAndPosition(dt)
            s.BucketGeneration()            # new grid, no density pass: nothing recorded, sub-steps scan
            s.UpdateVelocityAndPosition(dt)
            s.UpdateVelocityAndPosition(dt)
            s.DensityCalculation()
            s.UpdateVelocityAndPosition(dt)
        for a, b in zip(sim.download(), osim.download()):
            assert np.array_equal(_bits(a), _bits(b))
        st = sim.stats()
        assert n <= st["listed_target_substeps"] <= 3 * n          # only the three sub-steps behind a recording pass
    finally:
        sim.OnDestroy()


def test_lists_crowded_cells_and_long_lists(oracle):
    """R = 9 with 8192 particles: ~20 per cell, many cells over the 32-cap and lists longer than 64 entries -> those
    bricks are never listed; the rest still is."""
    n, R, fill = 8192, 9, 0.8
    pos0, vel0 = presets.dam_break(n, fill, seed=29)
    dt = presets.bench_dt_frame(R)
    (on, st_on), (off, _) = _run_pair(n, R, fill, pos0, vel0, 2, dt)
    for a, b in zip(on, off):
        assert np.array_equal(_bits(a), _bits(b))
    osim = oracle.OracleSimulation(oracle.derive(n, R, fill))
    osim.upload(pos0, vel0)
    for _ in range(2):
        osim.Update(dt)
    for a, b in zip(on, osim.download()):
        assert np.array_equal(_bits(a), _bits(b))
    assert st_on["scanned_brick_substeps"] > 0


def test_lists_blocked_by_out_of_box_upload(oracle):
    """A position outside [0,1] makes bins meaningless for one frame (wrapped rows): lists are not used in that frame
    and come back in the next (VelPos clamps)."""
    n, fill = 4096, 0.8
    R = presets.scene_bucket_resolution(n)
    pos0, vel0 = presets.dam_break(n, fill, seed=31, jitter=0.3)
    pos0[7, 0] = -1e-3      # (far outside, the wall force makes |v| > 1e15: the literal-loop path of the "wild" tests)
    pos0[99, 2] = 1.0 + 1e-3
    dt = presets.bench_dt_frame(R)
    from sphfluidsimulation_amd.simulation import SphFluidSimulation
    sim = SphFluidSimulation(particleNumber=n, bucketResolution=R, damFillRate=fill, lists="always").Start(pos0, vel0)
    osim = oracle.OracleSimulation(oracle.derive(n, R, fill))
    osim.upload(pos0, vel0)
    try:
        sim.reset_stats()
        sim.Update(dt)
        osim.Update(dt)
        assert sim.stats()["listed_target_substeps"] == 0
        sim.Update(dt)
        osim.Update(dt)
        assert sim.stats()["listed_target_substeps"] > 0
        for a, b in zip(sim.download(), osim.download()):
            assert np.array_equal(_bits(a), _bits(b))
    finally:
        sim.OnDestroy()


def test_frame_graph_replay_matches_issued_launches(oracle):
    """sph_step replays a captured HIP graph from the third frame on (SphStats.graph_replays).  Every frame gets its own
    deltaTime (read from device memory by the replayed kernels), the inspector scalars change mid-run (-> re-capture)
    and a hand-dispatched frame sits in between (-> the pointer parity flips): bit-identical to the oracle and to a
    context that issues all launches itself."""
    from sphfluidsimulation_amd.simulation import SphFluidSimulation
    n, fill = 16384, 0.8
    R = presets.scene_bucket_resolution(n)
    pos0, vel0 = presets.dam_break(n, fill, seed=41, jitter=0.3)
    base = presets.bench_dt_frame(R) * 0.2
    dts = [base * (1.0 + 0.37 * ((7 * k) % 5)) for k in range(12)]
    sims = [SphFluidSimulation(particleNumber=n, bucketResolution=R, damFillRate=fill, graphs=g, lists="always").Start(pos0, vel0)
            for g in (True, False)]
    osim = oracle.OracleSimulation(oracle.derive(n, R, fill))
    osim.upload(pos0, vel0)
    try:
        for k, dt in enumerate(dts):
            if k == 6:                               # inspector tweak mid-run
                for s in sims:
                    s.viscosity, s.gasConstant = 0.02, 120.0
                    s.set_params()
                osim.set_params(oracle.derive(n, R, fill, viscosity=0.02, gas_constant=120.0))
            if k == 8:                               # one frame dispatched pass by pass
                for s in sims + [osim]:
                    s.BucketGeneration()
                    s.DensityCalculation()
                    for _ in range(5):
                        s.UpdateVelocityAndPosition(float(np.float32(dt) / np.float32(25.0)))
                continue
            for s in sims:
                s.Update(dt)
            osim.Update(dt)
        want = osim.download()
        for s in sims:
            for a, b in zip(s.download(), want):
                assert np.array_equal(_bits(a), _bits(b))
        assert sims[0].stats()["graph_replays"] >= 6 and sims[1].stats()["graph_replays"] == 0
    finally:
        for s in sims:
            s.OnDestroy()


def test_adopted_stream_replays_graphs_too(oracle):
    """sph_set_stream: on a stream the caller owns (here a torch side stream) frames are captured and replayed exactly as on the
    context's own stream; on the legacy default stream (NULL, which cannot be captured) the launches are issued one by one.
    Both bit-identical to the oracle."""
    import torch
    from sphfluidsimulation_amd.simulation import SphFluidSimulation
    n, fill = 16384, 0.8
    R = presets.scene_bucket_resolution(n)
    pos0, vel0 = presets.dam_break(n, fill, seed=43, jitter=0.3)
    dt = presets.bench_dt_frame(R) * 0.5
    osim = oracle.OracleSimulation(oracle.derive(n, R, fill))
    osim.upload(pos0, vel0)
    for _ in range(7):
        osim.Update(dt)
    want = osim.download()
    side = torch.cuda.Stream(device=0)
    for handle, replays in ((side.cuda_stream, True), (0, False)):
        sim = SphFluidSimulation(particleNumber=n, bucketResolution=R, damFillRate=fill, lists="always").Start(pos0, vel0)
        try:
            sim.set_stream(handle)
            for _ in range(7):
                sim.Update(dt)
            for a, b in zip(sim.download(), want):
                assert np.array_equal(_bits(a), _bits(b))
            assert (sim.stats()["graph_replays"] >= 4) == replays
        finally:
            sim.OnDestroy()


def test_one_million_particles_six_frames(oracle):
    """A mid-size end-to-end run the way a Unity caller would drive it: the reference's own initial condition on the
    device (InitParticles.compute Preset1 + simplex noise), six sph_step frames (lists from frame 2, graph replay from
    frame 3), render feed at the end -- everything bit-identical to the oracle."""
    from sphfluidsimulation_amd.simulation import SphFluidSimulation
    n, fill = 1048576, 0.8
    R = presets.scene_bucket_resolution(n)
    dt = presets.bench_dt_frame(R)
    sim = SphFluidSimulation(preset=0, particleNumber=n, bucketResolution=R, damFillRate=fill, lists="always").Start(seed=None)
    osim = oracle.OracleSimulation(oracle.derive(n, R, fill))
    osim.upload(oracle.init_particles(0, 1024, fill), np.zeros((n, 4), f32))
    try:
        for _ in range(6):
            sim.Update(dt)
            osim.Update(dt)
        got, want = sim.download(), osim.download()
        for a, b in zip(got, want):
            assert np.array_equal(_bits(a), _bits(b))
        st = sim.stats()
        assert st["listed_target_substeps"] > 0.3 * n * 5 * 5 and st["graph_replays"] == 4   # (a 1M box is still mostly wall layer)
        sim.UpdateMeshProperties()
        props = sim.download_mesh_properties()
        assert np.array_equal(_bits(props), _bits(oracle.update_mesh_properties(want[0], want[1], sim.lowSpeed, sim.highSpeed,
                                                                               sim.particleRadius)))
    finally:
        sim.OnDestroy()


def test_list_policy_drops_lists_where_they_do_not_pay(oracle):
    """Default contexts record neighbour lists only while at least 90 % of the target sub-steps are served from them
    (SphStats.frames_without_lists).  A 65536-particle box is small: it starts without lists and probes at frame 32;
    at the full time step it is all wall layer, the probe fails and the back-off doubles; with tiny steps it is calm and
    the lists stay on.  The results are the oracle's either way, across every switch."""
    from sphfluidsimulation_amd.simulation import SphFluidSimulation
    n, fill = 65536, 0.8
    R = presets.scene_bucket_resolution(n)
    pos0, vel0 = presets.dam_break(n, fill, seed=43, jitter=0.3)
    for scale, violent in ((1.0, True), (0.02, False)):
        dt = presets.bench_dt_frame(R) * scale
        sim = SphFluidSimulation(particleNumber=n, bucketResolution=R, damFillRate=fill).Start(pos0, vel0)
        osim = oracle.OracleSimulation(oracle.derive(n, R, fill))
        osim.upload(pos0, vel0)
        try:
            frames = 48
            for _ in range(frames):
                sim.Update(dt)
                sim.synchronize()            # a caller that looks at every frame: the policy sees every report
                osim.Update(dt)
            for a, b in zip(sim.download(), osim.download()):
                assert np.array_equal(_bits(a), _bits(b))
            st = sim.stats()
            assert st["listed_target_substeps"] > 0              # the probe ran
            if violent:
                assert st["frames_without_lists"] >= 40          # 32 before the probe, and again after it failed
            else:
                assert 30 <= st["frames_without_lists"] <= 34    # the probe succeeded: lists from frame 33 on
        finally:
            sim.OnDestroy()


def test_scene_size_120_frames_default_context(oracle):
    """The reference's own scene (262 144 particles, R = 47) for 120 frames of 1/60 s in a default context: frame graphs
    from frame 3, the list policy probing at frame 32 and again after its back-off, lists recorded and dropped in
    between -- one bit pattern with the oracle at the end, through the beginning of the collapse."""
    from sphfluidsimulation_amd.simulation import SphFluidSimulation
    n, R, fill = 262144, 47, 0.8
    pos0, vel0 = presets.dam_break(n, fill, seed=51)
    dt = presets.bench_dt_frame(R)
    sim = SphFluidSimulation(particleNumber=n, bucketResolution=R, damFillRate=fill, viscosity=0.01, restDensity=1.5,
                             gasConstant=150.0, stiffnessCoefficient=5000.0, dampingCoefficient=10.0).Start(pos0, vel0)
    osim = oracle.OracleSimulation(oracle.derive(n, R, fill, viscosity=0.01, rest_density=1.5, gas_constant=150.0,
                                                 stiffness=5000.0, damping=10.0))
    osim.upload(pos0, vel0)
    try:
        for f in range(120):
            sim.Update(dt)
            osim.Update(dt)
            if f % 8 == 7:
                sim.synchronize()            # (lets the policy see its reports, as a caller reading back would)
        for a, b in zip(sim.download(), osim.download()):
            assert np.array_equal(_bits(a), _bits(b))
        st = sim.stats()
        assert st["graph_replays"] >= 100 and st["listed_target_substeps"] > 0 and st["frames_without_lists"] > 60
    finally:
        sim.OnDestroy()


def test_per_brick_and_per_chunk_displacement_ranges_agree(oracle):
    """Which bricks may use their lists in a sub-step is decided from displacement ranges: per brick, accumulated where the
    targets are integrated (the default), or per 64 slots from a pass over all positions (SPH_FLAG_CHUNK_BOXES; what z-slab
    contexts use).  Either is only a sufficient condition -- the results are the oracle's bit for bit both ways -- and both
    serve most target sub-steps of a flow that wakes some bricks."""
    from sphfluidsimulation_amd.simulation import SphFluidSimulation
    n, fill = 262144, 0.8
    R = presets.scene_bucket_resolution(n)
    pos0, vel0 = presets.dam_break(n, fill, seed=31, jitter=0.3)
    rng = np.random.default_rng(7)
    vel0[:, :3] = rng.standard_normal((n, 3)).astype(f32) * f32(0.3)
    dt = presets.bench_dt_frame(R) * 0.25
    frames = 3
    outs = []
    for chunk_boxes in (False, True):
        sim = SphFluidSimulation(particleNumber=n, bucketResolution=R, damFillRate=fill, lists="always", chunk_boxes=chunk_boxes).Start(pos0, vel0)
        try:
            sim.reset_stats()
            for _ in range(frames):
                sim.Update(dt)
            outs.append((sim.download(), sim.stats()))
        finally:
            sim.OnDestroy()
    for a, b in zip(outs[0][0], outs[1][0]):
        assert np.array_equal(_bits(a), _bits(b))
    osim = oracle.OracleSimulation(oracle.derive(n, R, fill))
    osim.upload(pos0, vel0)
    for _ in range(frames):
        osim.Update(dt)
    for a, b in zip(outs[0][0], osim.download()):
        assert np.array_equal(_bits(a), _bits(b))
    fr = [o[1]["listed_target_substeps"] / (n * frames * 5) for o in outs]
    assert 0.2 < fr[0] < 1.0 and 0.2 < fr[1] < 1.0, fr
    assert outs[0][1]["scanned_brick_substeps"] > 0 and outs[1][1]["scanned_brick_substeps"] > 0


@pytest.mark.parametrize("R", [47, 38])
def test_tile_record_slots_agree(oracle, R):
    """k_velpos_tile takes a brick's tile set-up from the records the density pass wrote (whole brick, or -- where ~6
    particles per cell make the halo-1 box overflow the tile, R = 38 here, config C4's regime -- its two halves).  Both
    instantiations (one record slot: halved bricks are set up from cell_start; two slots) agree bit for bit with each other,
    with the scanning kernels and with the oracle."""
    from sphfluidsimulation_amd.simulation import SphFluidSimulation
    n, fill = 262144, 0.8
    pos0, vel0 = presets.dam_break(n, fill, seed=29, jitter=0.3)
    dt = presets.bench_dt_frame(R) * 0.1
    frames = 2
    outs = []
    for slots, lists in ((1, "always"), (2, "always"), (0, False)):
        sim = SphFluidSimulation(particleNumber=n, bucketResolution=R, damFillRate=fill, lists=lists, tile_slots=slots).Start(pos0, vel0)
        try:
            sim.reset_stats()
            for _ in range(frames):
                sim.Update(dt)
            outs.append((sim.download(), sim.stats()))
        finally:
            sim.OnDestroy()
    for other, _ in outs[1:]:
        for a, b in zip(outs[0][0], other):
            assert np.array_equal(_bits(a), _bits(b))
    osim = oracle.OracleSimulation(oracle.derive(n, R, fill))
    osim.upload(pos0, vel0)
    for _ in range(frames):
        osim.Update(dt)
    for a, b in zip(outs[0][0], osim.download()):
        assert np.array_equal(_bits(a), _bits(b))
    for k in (0, 1):
        assert outs[k][1]["listed_target_substeps"] > 0.8 * n * frames * 5


@pytest.mark.parametrize("what", ["velocity", "position", "pressure"])
def test_tiny_operands_take_the_plain_divisions(oracle, what):
    """k_velpos_tile decides from per-particle operand tests whether its shared-reciprocal quotients are safe
    (sph_sweep_common.hpp, operand_tiny): a velocity component of 1e-30, a position component of 1e-20 or a pressure of
    ~1e-21 (gas constant 1e-20) puts numerators below the window, and the workgroups that stage
    such a particle must fall back to IEEE division -- bit-identical to the oracle and to the scanning kernels either way."""
    n, fill = 65536, 0.8
    R = presets.scene_bucket_resolution(n)
    pos0, vel0 = presets.dam_break(n, fill, seed=23, jitter=0.3)
    rng = np.random.default_rng(5)
    vel0[:, :3] = rng.standard_normal((n, 3)).astype(f32) * f32(0.01)
    pick = rng.choice(n, 4096, replace=False)
    kw = {}
    if what == "velocity":
        vel0[pick[:2048], 0] = f32(1e-30)
        vel0[pick[2048:], 2] = f32(-3e-33)
        vel0[pick[:64], 1] = f32(1e-42)       # denormal
    elif what == "position":
        # on the x = 0 and z = 0 walls: nonzero components below 2^-40 (the clamp keeps them in the box)
        pos0[pick[:2048], 0] = f32(1e-20)
        pos0[pick[2048:], 2] = f32(3e-25)
        vel0[pick, :3] = 0
    else:
        kw = dict(gasConstant=1e-20)          # every pressure k * (rho - rho0) ~ 1e-21 < 2^-60: every workgroup falls back
    dt = presets.bench_dt_frame(R) * 0.05
    frames = 2
    (on, st_on), (off, _) = _run_pair(n, R, fill, pos0, vel0, frames, dt, **kw)
    for a, b in zip(on, off):
        assert np.array_equal(_bits(a), _bits(b))
    d = oracle.derive(n, R, fill)
    if kw:
        d = oracle.derive(n, R, fill, gas_constant=kw["gasConstant"])
    osim = oracle.OracleSimulation(d)
    osim.upload(pos0, vel0)
    for _ in range(frames):
        osim.Update(dt)
    for a, b in zip(on, osim.download()):
        assert np.array_equal(_bits(a), _bits(b))
    assert st_on["listed_target_substeps"] > 0.5 * n * frames * 5


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("SPH_FUZZ_SEEDS", "10"))))
def test_random_configurations_all_paths(oracle, seed):
    """Random sizes, grids, fill rates, speeds, time steps, inspector scalars and list modes: whatever mixture of list,
    scanning, hand-over, generic, graph-replay and policy paths a configuration ends up on, the state after a few
    frames is the oracle's."""
    from sphfluidsimulation_amd.simulation import SphFluidSimulation
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([1024, 4096, 16384, 65536]))
    R = int(rng.integers(4, 40)) if seed % 3 else int(rng.choice([2, 3, 64, 128, 256]))
    fill = float(rng.choice([0.3, 0.5, 0.8, 1.0]))
    speed = float(rng.choice([0.0, 0.05, 0.5, 5.0]))
    lists = [True, "always", False][seed % 3]
    scal = dict(viscosity=float(rng.uniform(0.0, 0.1)), restDensity=float(rng.uniform(0.5, 3.0)),
                gasConstant=float(rng.uniform(10.0, 500.0)), stiffnessCoefficient=float(rng.uniform(1000.0, 10000.0)),
                dampingCoefficient=float(rng.uniform(1.0, 50.0)))
    pos0, vel0 = presets.dam_break(n, fill, seed=seed, jitter=float(rng.choice([0.0, 0.3, 1.0])))
    vel0[:, :3] = rng.standard_normal((n, 3)).astype(f32) * f32(speed)
    dt = float(rng.uniform(0.05, 1.0)) / (R - 1 if R > 1 else 1) * 0.7666667
    frames = int(rng.integers(3, 9))
    # (round 3: which k_velpos_tile instantiation walks the lists, where the displacement ranges come from, where the scanning
    # kernel runs -- none of it may change a bit)
    route = dict(tile_slots=int(rng.choice([0, 1, 2])), chunk_boxes=bool(rng.integers(0, 2)), side_stream=bool(rng.integers(0, 2)))
    sim = SphFluidSimulation(particleNumber=n, bucketResolution=R, damFillRate=fill, lists=lists, **route, **scal).Start(pos0, vel0)
    osim = oracle.OracleSimulation(oracle.derive(n, R, fill, viscosity=scal["viscosity"], rest_density=scal["restDensity"],
                                                 gas_constant=scal["gasConstant"], stiffness=scal["stiffnessCoefficient"],
                                                 damping=scal["dampingCoefficient"]))
    osim.upload(pos0, vel0)
    try:
        for _ in range(frames):
            sim.Update(dt)
            osim.Update(dt)
        for a, b, name in zip(sim.download(), osim.download(), ("pos", "vel", "rho")):
            same = (_bits(a) == _bits(b)) | (np.isnan(a) & np.isnan(b))
            assert same.all(), f"{name}: n={n} R={R} fill={fill} speed={speed} lists={lists} frames={frames} route={route}"
    finally:
        sim.OnDestroy()
