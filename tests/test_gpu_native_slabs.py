"""Native row-slab path (layer 3 of include/fluidsolver.h: planner + kernels + transport inside the
library) on the 1-GPU test box.  RCCL needs one GPU per rank, so K slabs run here as an in-process
group (fs_comm_create_local / fs_group_step): same planner, same kernels, same exchange segments,
device memcpy instead of ncclSend/ncclRecv.  The RCCL transport itself is exercised with a
single-rank communicator (send/recv to self + all-reduce).  Every result must equal the unsplit
run bit for bit (SURVEY.md 8(e) "Exactness")."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
F = np.float32

CASES = {
    # name: (W, H, params, ghost capacity, frames, force scale)
    "config2_physics": (600, 500, (1.5e-4, 0.0, 0.5, 10.0, 0.002, 40), None, 3, 0.02),
    "thin_ghost_rechunks": (300, 260, (1.5e-4, 0.00001, 0.5, 10.0, 0.002, 20), 12, 3, 0.02),
    "multi_hop_ghost": (200, 180, (1.5e-4, 0.0, 0.5, 10.0, 0.002, 8), 180, 3, 6.0),
    "dye_diffusion": (260, 240, (0.00005, 0.00001, 0.4, 10.0, 0.008, 5), None, 3, 0.05),
    "no_viscosity_no_vorticity": (192, 161, (0.0, 0.0, 0.5, 0.0, 0.02, 4), None, 3, 0.05),
}


# transport x plan: blocking copies / asynchronous copies ordered by the RCCL path's events; halo
# traffic exposed (exchange, wait, compute) / hidden (second stream, split advection, guessed halo)
MODES = {"blocking-plain": (False, False), "blocking-overlap": (False, True), "async-plain": (True, False),
         "async-overlap": (True, True)}


def _run_group(case, world, asynchronous=False, overlap=True):
    from fluidsolver_amd.native_slabs import LocalGroup
    from fluidsolver_amd.script import splat
    W, H, params, ghost, frames, fscale = case
    g = LocalGroup(W, H, params, world, device=0, ghost=ghost, asynchronous=asynchronous, overlap=overlap)
    try:
        halos = []
        for f in range(frames):
            for ptr in range(2):
                pos, force, col = splat(W, H, 5 * f + 1, ptr, 2)
                g.add_force(pos, (force[0] * F(fscale), force[1] * F(fscale)), col)
            g.update()
            halos.append(g.slabs[0].last_halo)
        g.sync()
        return g.field("vel"), g.field("dye"), g.field("p"), halos, [s.stats for s in g.slabs]
    finally:
        g.close()


_REF = {}


@pytest.mark.parametrize("mode", sorted(MODES))
@pytest.mark.parametrize("world", [2, 3, 8])
@pytest.mark.parametrize("name", sorted(CASES))
def test_local_group_equals_oracle(oracle_omp, world, name, mode):
    import test_slabs as T
    case = CASES[name]
    vel, dye, prs, halos, stats = _run_group(case, world, *MODES[mode])
    if name not in _REF:
        _REF[name] = T._reference(case)
    rvel, rdye, rprs = _REF[name]
    assert np.array_equal(vel, rvel), (name, "vel")
    assert np.array_equal(dye, rdye), (name, "dye")
    assert np.array_equal(prs, rprs), (name, "p")
    assert np.abs(rvel).max() > 0
    if name == "multi_hop_ghost" and world == 8:
        assert max(halos) > case[1] // world, "the case is meant to need rows from beyond the neighbour: %r" % halos
    if name == "config2_physics" and world <= 3:
        # two exchange rounds per frame (one in the first frame, whose all-zero fields need none at its start)
        assert all(s["exchanges"] == 2 * case[4] - 1 for s in stats), stats
    if not MODES[mode][1]:
        assert all(s["redo"] == 0 for s in stats)


def test_config4_eight_slabs_equal_single_gpu_context():
    """BASELINE config 4 at full size: the 16384^2 grid as EIGHT slabs of 2048 rows (the driver's 8-GPU strong-scaling
    case, here as an in-process group with asynchronous copies and the overlapped plan) against fs_step on the whole
    grid, N = 40, eight scripted pointers, 4 frames: bit-identical velocity and dye, halos of tens to hundreds of rows."""
    from fluidsolver_amd import FluidSolver, lib as L
    from fluidsolver_amd.native_slabs import LocalGroup
    from fluidsolver_amd.script import splat
    W = H = 16384
    params = (1.5e-4, 0.0, 0.5, 10.0, 0.002, 40)
    g = LocalGroup(W, H, params, 8, device=0, asynchronous=True, overlap=True)
    s = FluidSolver(W, H)
    (s.viscosity, s.diffusion, s.deltaTime, s.vorticityStrength, s.fadeSpeed, s.solverIterations) = params
    s.displayMode = None
    try:
        halos = []
        for f in range(4):
            for ptr in range(8):
                pos, force, col = splat(W, H, 5 * f + 1, ptr, 8)
                g.add_force(pos, force, col)
                s.applyForce(pos, force, col)
            g.update()
            s.update()
            halos.append(g.slabs[0].last_halo)
        g.sync()
        s.sync()
        assert max(halos) > 30, halos
        for name, fid in (("vel", L.FIELD_VELOCITY), ("dye", L.FIELD_DENSITY)):
            a = g.field(name)
            assert np.array_equal(a, s.read_field(fid)), name
            assert np.abs(a).max() > 0
            del a
    finally:
        g.close()
        s.close()


def test_config5_slabs_equal_single_gpu_context():
    """BASELINE config 5 at full slab size: 8192 x 8192 per slab, N = 100 (a 203-row stencil chain per frame), here
    two slabs stacked to 8192 x 16384 against fs_step on the whole grid, 3 frames."""
    from fluidsolver_amd import FluidSolver, lib as L
    from fluidsolver_amd.native_slabs import LocalGroup
    from fluidsolver_amd.script import splat
    W, H = 8192, 16384
    params = (1.5e-4, 0.0, 0.5, 10.0, 0.002, 100)
    g = LocalGroup(W, H, params, 2, device=0, asynchronous=True, overlap=True)
    s = FluidSolver(W, H)
    (s.viscosity, s.diffusion, s.deltaTime, s.vorticityStrength, s.fadeSpeed, s.solverIterations) = params
    s.displayMode = None
    try:
        for f in range(3):
            for ptr in range(2):
                pos, force, col = splat(W, H, 5 * f + 1, ptr, 2)
                g.add_force(pos, force, col)
                s.applyForce(pos, force, col)
            g.update()
            s.update()
        g.sync()
        s.sync()
        for name, fid in (("vel", L.FIELD_VELOCITY), ("dye", L.FIELD_DENSITY), ("p", L.FIELD_PRESSURE)):
            a = g.field(name)
            assert np.array_equal(a, s.read_field(fid)), name
            assert np.abs(a).max() > 0
            del a
    finally:
        g.close()
        s.close()


@pytest.mark.parametrize("ask", [True, False])
@pytest.mark.parametrize("mode", sorted(MODES))
def test_local_group_equals_single_gpu_context_at_size(mode, ask):
    """4 slabs of a 1536 x 1024 grid, N = 40, full-strength scripted splats (advection halo of tens of
    rows, changing from frame to frame) against fs_step on the whole grid (no oracle: too slow).  ask = the halo is
    read back after every frame, which runs the frame's halo check at once; without it the frames run one ahead of their
    checks (the velocity roles rotate through the spare buffer)."""
    from fluidsolver_amd import FluidSolver, lib as L
    from fluidsolver_amd.native_slabs import LocalGroup
    from fluidsolver_amd.script import splat
    W, H, params = 1536, 1024, (1.5e-4, 0.0, 0.5, 10.0, 0.002, 40)
    asynchronous, overlap = MODES[mode]
    g = LocalGroup(W, H, params, 4, device=0, asynchronous=asynchronous, overlap=overlap)
    s = FluidSolver(W, H)
    (s.viscosity, s.diffusion, s.deltaTime, s.vorticityStrength, s.fadeSpeed, s.solverIterations) = params
    s.displayMode = None
    try:
        halos = []
        for f in range(12):
            for ptr in range(4):
                pos, force, col = splat(W, H, 7 * f + 1, ptr, 4)
                g.add_force(pos, force, col)
                s.applyForce(pos, force, col)
            g.update()
            s.update()
            halos.append(g.slabs[0].last_halo if ask or f == 11 else 9)
        g.sync()
        s.sync()
        assert max(halos) > 8, halos
        # from the third frame on the exchange-free run of launches of a frame (vorticity .. projection) is a replayed hipGraph
        gs = [sl.graph_stats for sl in g.slabs]
        assert all(x["replays"] >= (6 if ask else 3) and 1 <= x["captures"] <= (4 if ask else 7) for x in gs), gs
        assert np.array_equal(g.field("vel"), s.read_field(L.FIELD_VELOCITY))
        assert np.array_equal(g.field("dye"), s.read_field(L.FIELD_DENSITY))
        assert np.array_equal(g.field("p"), s.read_field(L.FIELD_PRESSURE))
        assert np.abs(g.field("vel")).max() > 0
    finally:
        g.close()
        s.close()


@pytest.mark.parametrize("fuse", [0, 4])
@pytest.mark.parametrize("world", [2, 3])
def test_slab_plans_with_every_fusion_forced_or_off(oracle_omp, world, fuse):
    """The big-grid choices of the slab plan (both advections of the interior rows in one pass) and the no-fusion
    plan, forced at a test size with FS_OPT_FUSE 4 / 0: still the oracle's bits."""
    import test_slabs as T
    from fluidsolver_amd.native_slabs import LocalGroup
    from fluidsolver_amd.script import splat
    case = CASES["config2_physics"]
    W, H, params, ghost, frames, fscale = case
    g = LocalGroup(W, H, params, world, device=0, ghost=ghost, asynchronous=True, overlap=True)
    try:
        for sl in g.slabs:
            sl.set_fuse(fuse)
        for f in range(frames):
            for ptr in range(2):
                pos, force, col = splat(W, H, 5 * f + 1, ptr, 2)
                g.add_force(pos, (force[0] * F(fscale), force[1] * F(fscale)), col)
            g.update()
        g.sync()
        if "config2_physics" not in _REF:
            _REF["config2_physics"] = T._reference(case)
        rvel, rdye, rprs = _REF["config2_physics"]
        assert np.array_equal(g.field("vel"), rvel) and np.array_equal(g.field("dye"), rdye)
        assert np.array_equal(g.field("p"), rprs)
    finally:
        g.close()


def test_wrong_halo_guess_is_replanned():
    """Overlap mode guesses the advection halo from the previous frame.  A splat 40x stronger than the
    last one makes the guess too small: the advection must be planned again with the true depth (counted
    in the stats) and the fields must still equal the single-GPU run bit for bit."""
    from fluidsolver_amd import FluidSolver, lib as L
    from fluidsolver_amd.native_slabs import LocalGroup
    W, H, params = 512, 768, (1.5e-4, 0.0, 0.5, 10.0, 0.002, 10)
    g = LocalGroup(W, H, params, 3, device=0, asynchronous=True, overlap=True)
    s = FluidSolver(W, H)
    (s.viscosity, s.diffusion, s.deltaTime, s.vorticityStrength, s.fadeSpeed, s.solverIterations) = params
    s.displayMode = None
    try:
        halos = []
        for f, strength in enumerate([5.0, 5.0, 200.0, 200.0, 5.0, 900.0, 1.0]):
            pos, force, col = (200.0 + 30 * f, 250.0 + 3 * f), (0.3 * strength, strength), (50.0, 20.0, 5.0)
            g.add_force(pos, force, col)
            s.applyForce(pos, force, col)
            g.update()
            s.update()
            halos.append(g.slabs[0].last_halo)
        g.sync()
        s.sync()
        redo = [sl.stats["redo"] for sl in g.slabs]
        assert redo[0] >= 2 and len(set(redo)) == 1, (redo, halos)
        assert np.array_equal(g.field("vel"), s.read_field(L.FIELD_VELOCITY))
        assert np.array_equal(g.field("dye"), s.read_field(L.FIELD_DENSITY))
        assert np.array_equal(g.field("p"), s.read_field(L.FIELD_PRESSURE))
    finally:
        g.close()
        s.close()


@pytest.mark.parametrize("defer", [0, 1])
def test_deferred_halo_check_repairs_frame_and_splats(defer):
    """FS_OPT_SLAB_DEFER: fs_group_step returns without waiting for the CFL bound; the check runs at the next call.  Several
    splats land between a frame whose guess was too small and the call that finds out: the repair must repeat them.  Same
    bits as one GPU with the check deferred and not."""
    from fluidsolver_amd import FluidSolver, lib as L
    from fluidsolver_amd.native_slabs import LocalGroup
    W, H, params = 512, 768, (1.5e-4, 0.0, 0.5, 10.0, 0.002, 10)
    g = LocalGroup(W, H, params, 3, device=0, asynchronous=True, overlap=True)
    s = FluidSolver(W, H)
    (s.viscosity, s.diffusion, s.deltaTime, s.vorticityStrength, s.fadeSpeed, s.solverIterations) = params
    s.displayMode = None
    try:
        for sl in g.slabs:
            sl.set_option(L.OPT_SLAB_DEFER, defer)
        for f, strength in enumerate([5.0, 5.0, 300.0, 5.0, 5.0, 900.0, 1.0, 1.0]):
            for k in range(3):   # three splats per frame, two of them across slab boundaries
                pos = (150.0 + 40 * f + 60 * k, 250.0 * (k + 0.02) + 3 * f)
                force, col = (0.3 * strength, strength * (1 if k != 1 else -1)), (50.0, 20.0 + k, 5.0)
                g.add_force(pos, force, col)
                s.applyForce(pos, force, col)
            g.update()
            s.update()
        g.sync()
        s.sync()
        redo = [sl.stats["redo"] for sl in g.slabs]
        assert redo[0] >= 2 and len(set(redo)) == 1, redo
        assert np.array_equal(g.field("vel"), s.read_field(L.FIELD_VELOCITY))
        assert np.array_equal(g.field("dye"), s.read_field(L.FIELD_DENSITY))
        assert np.array_equal(g.field("p"), s.read_field(L.FIELD_PRESSURE))
    finally:
        g.close()
        s.close()


@pytest.mark.parametrize("mode", ["async-overlap", "blocking-plain"])
def test_projection_as_two_bands_inside_a_slab(mode):
    """Two slabs of 3072 x 2048: large enough that each slab runs its fused projection as two row bands out of step
    (replayed segments only), pressure of the lower band in the band's private plane until read.  Against fs_step on the
    whole 3072 x 4096 grid, N = 40, 7 frames: velocity, dye and pressure bit-identical."""
    from fluidsolver_amd import FluidSolver, lib as L
    from fluidsolver_amd.native_slabs import LocalGroup
    from fluidsolver_amd.script import splat
    W, H, params = 3072, 4096, (1.5e-4, 0.0, 0.5, 10.0, 0.002, 40)
    asynchronous, overlap = MODES[mode]
    g = LocalGroup(W, H, params, 2, device=0, asynchronous=asynchronous, overlap=overlap)
    s = FluidSolver(W, H)
    (s.viscosity, s.diffusion, s.deltaTime, s.vorticityStrength, s.fadeSpeed, s.solverIterations) = params
    s.displayMode = None
    try:
        for f in range(12):
            for ptr in range(2):
                pos, force, col = splat(W, H, 5 * f + 1, ptr, 2)
                g.add_force(pos, force, col)
                s.applyForce(pos, force, col)
            g.update()
            s.update()
            if f == 9:   # a read in mid-run brings the lower band's rows home; the next frame must not mind
                assert np.array_equal(g.field("p"), s.read_field(L.FIELD_PRESSURE))
        g.sync()
        s.sync()
        # (six recordings: velocity roles of period 3 -- two buffers + the spare -- times dye / pressure / CFL slot of period 2)
        gs = [sl.graph_stats for sl in g.slabs]
        # (and the read in mid-run, which runs the owed check early, shifts the phase of the roles: a few more)
        assert all(x["replays"] >= 2 and x["captures"] <= 12 for x in gs), gs
        for name, fid in (("vel", L.FIELD_VELOCITY), ("dye", L.FIELD_DENSITY), ("p", L.FIELD_PRESSURE)):
            a = g.field(name)
            assert np.array_equal(a, s.read_field(fid)), name
            assert np.abs(a).max() > 0
    finally:
        g.close()
        s.close()


def test_written_fields_invalidate_ghosts():
    """fs_write_field on a slab marks its ghost rows stale: the next frame must fetch them."""
    from fluidsolver_amd import FluidSolver, lib as L
    from fluidsolver_amd.native_slabs import LocalGroup
    W, H, params, world = 320, 210, (1.5e-4, 0.0, 0.5, 5.0, 0.002, 9), 3
    rng = np.random.default_rng(7)
    vel = rng.uniform(-2, 2, (2, H, W)).astype(F)
    dye = rng.uniform(0, 1, (3, H, W)).astype(F)
    g = LocalGroup(W, H, params, world, device=0, asynchronous=True)
    s = FluidSolver(W, H)
    (s.viscosity, s.diffusion, s.deltaTime, s.vorticityStrength, s.fadeSpeed, s.solverIterations) = params
    s.displayMode = None
    try:
        for sl in g.slabs:
            sl.write_owned("vel", vel[:, sl.r0:sl.r1])
            sl.write_owned("dye", dye[:, sl.r0:sl.r1])
        s.write_field(L.FIELD_VELOCITY, vel)
        s.write_field(L.FIELD_DENSITY, dye)
        for _ in range(2):
            g.update()
            s.update()
        g.sync()
        s.sync()
        assert np.array_equal(g.field("vel"), s.read_field(L.FIELD_VELOCITY))
        assert np.array_equal(g.field("dye"), s.read_field(L.FIELD_DENSITY))
    finally:
        g.close()
        s.close()


def test_halo_beyond_ghost_capacity_is_reported():
    """A flow faster than the ghost capacity allows must fail loudly (FS_ERR_HALO), never silently."""
    from fluidsolver_amd import lib as L
    from fluidsolver_amd.native_slabs import LocalGroup
    W, H, params = 200, 180, (1.5e-4, 0.0, 0.5, 10.0, 0.002, 8)
    g = LocalGroup(W, H, params, 3, device=0, ghost=6)
    try:
        g.add_force((100.0, 90.0), (0.0, 4000.0), (1.0, 1.0, 1.0))
        with pytest.raises(L.FluidSolverError) as e:
            g.update()
        assert e.value.code == -4 and "ghost capacity" in str(e.value)
    finally:
        g.close()


def test_rccl_transport_single_rank():
    """The RCCL transport on the box's one GPU: communicator of world 1, grouped send/recv to self and
    the all-reduce (fs_comm_selftest), then a slab context on it stepping like a plain context."""
    from fluidsolver_amd import FluidSolver, lib as L
    from fluidsolver_amd.native_slabs import RcclSlab
    from fluidsolver_amd.script import splat
    W, H, params = 384, 256, (1.5e-4, 0.0, 0.5, 10.0, 0.002, 12)
    a = RcclSlab(W, H, params, device=0, rank=0, world=1)
    b = FluidSolver(W, H)
    (b.viscosity, b.diffusion, b.deltaTime, b.vorticityStrength, b.fadeSpeed, b.solverIterations) = params
    b.displayMode = None
    try:
        L.check(a.lib.fs_comm_selftest(a.comm, 3 * 4096))
        for f in range(3):
            pos, force, col = splat(W, H, 3 * f + 1)
            a.add_force(pos, force, col)
            b.applyForce(pos, force, col)
            a.update()
            b.update()
        a.sync()
        b.sync()
        assert np.array_equal(a.owned("vel"), b.read_field(L.FIELD_VELOCITY))
        assert np.array_equal(a.owned("dye"), b.read_field(L.FIELD_DENSITY))
    finally:
        a.close()
        b.close()


def _tracers(n, W, H, seed=3):
    rng = np.random.default_rng(seed)
    z = np.zeros(n, F)
    return [rng.uniform(0, W, n).astype(F), rng.uniform(0, H, n).astype(F), z, z.copy(),
            rng.uniform(0.3, 1.0, n).astype(F), rng.uniform(0.1, 1.0, n).astype(F), z.copy()]


def _sorted_columns(a):
    """[7, n] tracer arrays as a canonically ordered multiset (the slab runtime does not keep an order)."""
    a = np.asarray(a)
    return a[:, np.lexsort(a[::-1])]


@pytest.mark.parametrize("asynchronous", [False, True])
@pytest.mark.parametrize("world", [2, 3, 5])
def test_tracers_on_slabs_equal_single_gpu(world, asynchronous):
    """tracers live on the slab that owns the row their lookup reads and migrate when they
    cross.  After several frames of a strong flow the union of the slabs' tracers must equal the single-GPU
    tracer set bit for bit (as a multiset), every tracer must sit on the rank that owns its row, and some must
    have migrated after the initial distribution."""
    from fluidsolver_amd import FluidSolver, ParticleSystem
    from fluidsolver_amd.native_slabs import LocalGroup
    from fluidsolver_amd.script import splat
    W, H, params, n, frames = 448, 420, (1.5e-4, 0.0, 0.5, 10.0, 0.002, 8), 30000, 6
    arrs = _tracers(n, W, H)
    g = LocalGroup(W, H, params, world, device=0, asynchronous=asynchronous)
    s = FluidSolver(W, H, max_particles=n)
    (s.viscosity, s.diffusion, s.deltaTime, s.vorticityStrength, s.fadeSpeed, s.solverIterations) = params
    s.displayMode = None
    ps = ParticleSystem(s)
    ps.fluidForce, ps.linkToViscosity = 2.0, False       # strong coupling so that tracers really travel
    ps.set_arrays(*arrs)
    try:
        for sl in g.slabs:
            sl.particles_reserve(n, fluidForce=2.0, linkToViscosity=False)
            sl.particles_set(*arrs)
        received_after_first = None
        for f in range(frames):
            for ptr in range(2):
                pos, force, col = splat(W, H, 9 * f + 1, ptr, 2)
                g.add_force(pos, force, col)
                s.applyForce(pos, force, col)
            g.update()
            s.update()
            ps.update(s, (W, H))
            if f == 0:
                received_after_first = sum(sl.stats["tracers_received"] for sl in g.slabs)
        g.sync()
        s.sync()
        parts = [sl.particles_get() for sl in g.slabs]
        assert sum(p.shape[1] for p in parts) == n
        got = _sorted_columns(np.concatenate(parts, axis=1))
        exp = _sorted_columns(np.stack(ps.get_arrays()))
        assert np.array_equal(got, exp)
        assert sum(sl.stats["tracers_received"] for sl in g.slabs) > received_after_first, "nothing migrated after frame 0"
        moved = np.abs(exp[0] - np.sort(arrs[0])).max()
        assert moved > 0
    finally:
        g.close()
        s.close()


def test_tracer_capacity_is_checked():
    from fluidsolver_amd import lib as L
    from fluidsolver_amd.native_slabs import LocalGroup
    W, H, params, n = 256, 240, (1.5e-4, 0.0, 0.5, 0.0, 0.002, 4), 4000
    arrs = _tracers(n, W, H)
    arrs[1][:] = 5.0                                      # every tracer belongs to rank 0
    g = LocalGroup(W, H, params, 2, device=0)
    try:
        for sl in g.slabs:
            sl.particles_reserve(n // 2 + 10)
            sl.particles_set(*arrs)
        with pytest.raises(L.FluidSolverError) as e:
            g.update()
        assert e.value.code == -3 and "capacity" in str(e.value)
    finally:
        g.close()


def _live(a):
    a = np.asarray(a)
    return _sorted_columns(a[:, a[4] != 0])


@pytest.mark.parametrize("world", [2, 3])
def test_spawning_on_slabs_equals_single_gpu(world):
    """ParticleSystem.addParticles on slabs: fs_particles_spawn is collective, every rank appends its index block of the
    new tracers and the next frame re-homes them.  With spawns between the frames the live tracers of all slabs together
    must be the single-GPU live set, bit for bit."""
    from fluidsolver_amd import FluidSolver, ParticleSystem
    from fluidsolver_amd.native_slabs import LocalGroup
    from fluidsolver_amd.script import splat
    W, H, params, n, frames = 448, 420, (1.5e-4, 0.0, 0.5, 10.0, 0.002, 8), 6000, 7
    arrs = _tracers(n, W, H)
    arrs[4][::7] = 0.02                                  # some die on the way: the slabs keep their slots, one GPU drops them
    g = LocalGroup(W, H, params, world, device=0, asynchronous=True)
    s = FluidSolver(W, H, max_particles=4 * n)
    (s.viscosity, s.diffusion, s.deltaTime, s.vorticityStrength, s.fadeSpeed, s.solverIterations) = params
    s.displayMode = None
    ps = ParticleSystem(s)
    ps.fluidForce, ps.linkToViscosity, ps.spawnRadius = 2.0, False, 40.0
    ps.set_arrays(*arrs)
    try:
        for sl in g.slabs:
            sl.particles_reserve(4 * n, fluidForce=2.0, linkToViscosity=False)
            sl.particles_set(*arrs)
        spawned = 0
        for f in range(frames):
            pos, force, col = splat(W, H, 9 * f + 1)
            g.add_force(pos, force, col)
            s.applyForce(pos, force, col)
            at, count, seed = (pos[0], H * (0.1 + 0.13 * f)), 101 + 37 * f, 1000 + f   # near and across the slab borders
            for sl in g.slabs:
                sl.particles_spawn(at, count, radius=40.0, seed=seed)
            ps.addParticles(at, count, seed=seed)
            spawned += count
            g.update()
            s.update()
            ps.update(s, (W, H))
        g.sync()
        s.sync()
        parts = [sl.particles_get() for sl in g.slabs]
        exp = _live(np.stack(ps.get_arrays()))
        got = _live(np.concatenate(parts, axis=1))
        assert exp.shape[1] > n - n // 7 and got.shape == exp.shape
        assert np.array_equal(got, exp)
        assert sum(sl.stats["tracers_received"] for sl in g.slabs) > 0
    finally:
        g.close()
        s.close()


def test_spawn_on_a_full_slab():
    """A slab keeps no age order, so a spawn never pushes tracers out there: dead slots are dropped when room runs out, and a
    rank full of live tracers refuses with FS_ERR_NOMEM."""
    from fluidsolver_amd import lib as L
    from fluidsolver_amd.native_slabs import LocalGroup
    W, H, params, n = 256, 240, (1.5e-4, 0.0, 0.5, 0.0, 0.002, 4), 4000
    arrs = _tracers(n, W, H)
    arrs[4][: n // 2] = 0.0                              # the first rank's index block is all dead slots
    g = LocalGroup(W, H, params, 2, device=0)
    try:
        for sl in g.slabs:
            sl.particles_reserve(n // 2 + 100)
            sl.particles_set(*arrs)
        for sl in g.slabs:                               # 100 each: fits behind the slots in use
            sl.particles_spawn((100.0, 100.0), 200, seed=5)
        assert [sl.lib.fs_particles_count(sl.ctx) for sl in g.slabs] == [n // 2 + 100, n // 2 + 100]
        g.slabs[0].particles_spawn((100.0, 100.0), 600, seed=6)      # rank 0: room only once its dead slots are dropped
        assert g.slabs[0].lib.fs_particles_count(g.slabs[0].ctx) == 100 + 300
        with pytest.raises(L.FluidSolverError) as e:     # rank 1 is full of live tracers
            g.slabs[1].particles_spawn((100.0, 100.0), 600, seed=6)
        assert e.value.code == -3 and "capacity" in str(e.value)
    finally:
        g.close()


@pytest.mark.parametrize("seed", range(1, 1 + int(os.environ.get("FS_FUZZ_SEEDS", "6"))))   # (more walks: FS_FUZZ_SEEDS=100)
def test_random_walks_on_slabs_equal_single_gpu(seed):
    """The slab frame is a state machine too (speculated halo depths checked one frame late, a third velocity buffer whose roles
    rotate, replayed segments keyed by roles and row ranges, splats logged for a repair, ghost depths that survive or do not).
    A random walk over what a host does -- splats of changing strength, frames, parameter and option changes on every rank,
    reads in between, resets, spawns -- on 2-4 concurrent slabs against ONE context making the same moves: fields at random
    checkpoints and at the end, and the live tracers as a multiset, bit for bit."""
    from fluidsolver_amd import FluidSolver, ParticleSystem, lib as L
    from fluidsolver_amd.native_slabs import LocalGroup, make_params
    rng = np.random.default_rng(7000 + seed)
    world = int(rng.integers(2, 5))
    W, H = (320, 96 * world + int(rng.integers(0, 7))) if seed % 2 else (448, 80 * world + 5)
    par = [1.5e-4, 0.0, 0.2, 10.0, 0.002, 6]
    tracers = seed % 3 == 0
    cap = 20000
    g = LocalGroup(W, H, tuple(par), world, device=0, asynchronous=bool(seed % 4), overlap=True)
    s = FluidSolver(W, H, max_particles=cap if tracers else 0)
    s.displayMode = None
    ps = ParticleSystem(s) if tracers else None

    def push():
        (s.viscosity, s.diffusion, s.deltaTime, s.vorticityStrength, s.fadeSpeed, s.solverIterations) = par
        p = make_params(tuple(par))
        for sl in g.slabs:
            L.check(sl.lib.fs_set_params(sl.ctx, C.byref(p)), sl.ctx)
            sl.deltaTime, sl.viscosity = par[2], par[0]

    def check(what):
        g.sync()
        s.sync()
        assert np.array_equal(g.field("vel"), s.read_field(L.FIELD_VELOCITY)), (what, "velocity")
        assert np.array_equal(g.field("dye"), s.read_field(L.FIELD_DENSITY)), (what, "dye")
        if tracers:
            got = _live(np.concatenate([sl.particles_get() for sl in g.slabs], axis=1))
            exp = _live(np.stack(ps.get_arrays()))
            assert got.shape == exp.shape and np.array_equal(got, exp), (what, "tracers")

    log, flow_seen = [], False
    try:
        push()
        if tracers:
            ps.fluidForce, ps.linkToViscosity = 2.0, False
            arrs = _tracers(6000, W, H, seed)
            ps.set_arrays(*arrs)
            for sl in g.slabs:
                sl.particles_reserve(cap, fluidForce=2.0, linkToViscosity=False)
                sl.particles_set(*arrs)
        for it in range(50):
            strength = float(rng.choice([10.0, 50.0, 150.0]))      # halo depths that jump: speculation misses
            for _ in range(int(rng.integers(0, 3))):
                pos = (float(rng.uniform(-5, W + 5)), float(rng.uniform(-5, H + 5)))
                force = (float(rng.uniform(-strength, strength)), float(rng.uniform(-strength, strength)))
                col = tuple(float(x) for x in rng.uniform(0, 40, 3))
                g.add_force(pos, force, col)
                s.applyForce(pos, force, col)
            r = rng.random()
            if tracers and r < 0.3:
                at, count, sd = (float(rng.uniform(0, W)), float(rng.uniform(0, H))), int(rng.choice([1, 10, 300])), int(rng.integers(1, 2 ** 31))
                for sl in g.slabs:
                    sl.particles_spawn(at, count, radius=15.0, seed=sd)
                ps.spawnRadius = 15.0
                ps.addParticles(at, count, seed=sd)
                log.append(("spawn", count))
            r = rng.random()
            if r < 0.08:
                par[0] = float(rng.choice([1.5e-4, 3.0e-4, 0.0])); push(); log.append(("viscosity", par[0]))
            elif r < 0.14:
                par[3] = float(rng.choice([0.0, 10.0, 4.0])); push(); log.append(("vorticity", par[3]))
            elif r < 0.18:
                par[5] = int(rng.choice([6, 3, 8])); push(); log.append(("iterations", par[5]))
            elif r < 0.22:
                par[1] = float(rng.choice([0.0, 2.0e-4])); push(); log.append(("diffusion", par[1]))
            elif r < 0.32:
                opt, val = [(L.OPT_USE_GRAPH, int(rng.integers(0, 2))), (L.OPT_SLAB_DEFER, int(rng.integers(0, 2))),
                            (L.OPT_SLAB_OVERLAP, int(rng.integers(0, 2))), (L.OPT_FUSE, int(rng.integers(0, 5))),
                            (L.OPT_SPLIT_ROWS, int(rng.integers(0, 2))), (L.OPT_SEGMENT_GRAPH_AFTER, int(rng.integers(0, 4)))][int(rng.integers(0, 6))]
                for sl in g.slabs:
                    sl.set_option(opt, val)
                log.append(("option", opt, val))
            elif r < 0.35:
                for sl in g.slabs:
                    L.check(sl.lib.fs_reset(sl.ctx), sl.ctx)
                s.reset(); log.append(("reset",))
            elif r < 0.40:
                k = int(rng.integers(0, world))
                g.slabs[k].owned("dye"); log.append(("read", k))     # one rank looks at its rows: its owed halo check runs now
            try:
                g.update()
            except L.FluidSolverError as e:
                raise AssertionError((str(e), log[-15:], float(np.abs(s.read_field(L.FIELD_VELOCITY)[1]).max())))
            s.update()
            if tracers:
                ps.update(s, (W, H))
            log.append(("frame", it))
            if rng.random() < 0.12:
                check(log[-12:])
            # (confinement without viscosity pumps the flow up; before the advection halo outgrows the ghost rows -- a
            # legitimate FS_ERR_HALO -- start over.  Only the single context is looked at: the slabs are not made to wait)
            vmax = float(np.abs(s.read_field(L.FIELD_VELOCITY)[1]).max())
            flow_seen = flow_seen or vmax > 0
            if vmax * par[2] > 0.3 * g.slabs[0].G:
                for sl in g.slabs:
                    L.check(sl.lib.fs_reset(sl.ctx), sl.ctx)
                s.reset(); log.append(("reset (fast flow)",))
        check(log[-12:])
        assert flow_seen
        if os.environ.get("FS_FUZZ_VERBOSE"):
            print("seed", seed, "world", world, "redo", [sl.stats["redo"] for sl in g.slabs], g.slabs[0].graph_stats,
                  "last halo", g.slabs[0].last_halo)
    finally:
        g.close()
        s.close()


def test_join_waits_are_measured():
    """FS_OPT_SLAB_PROFILE: every wait of a slab's kernels for halo rows is timed on the GPU (bench.py reports the
    sums at N > 1, so the next look at the links has data); results are unchanged by it."""
    from fluidsolver_amd.native_slabs import LocalGroup
    from fluidsolver_amd.script import splat
    W, H, params = 512, 768, (1.5e-4, 0.0, 0.5, 10.0, 0.002, 10)
    runs = []
    for profile in (False, True):
        g = LocalGroup(W, H, params, 3, device=0, asynchronous=True, overlap=True)
        try:
            for sl in g.slabs:
                sl.set_profile(profile)
            for f in range(5):
                pos, force, col = splat(W, H, 4 * f + 1)
                g.add_force(pos, force, col)
                g.update()
            g.sync()
            w = [sl.waits() for sl in g.slabs]
            runs.append((g.field("vel"), g.field("dye"), w))
        finally:
            g.close()
    assert np.array_equal(runs[0][0], runs[1][0]) and np.array_equal(runs[0][1], runs[1][1])
    assert all(sum(x["joins"]) == 0 for x in runs[0][2])
    mid = runs[1][2][1]                      # the middle slab has two neighbours
    assert mid["joins"][0] >= 4 and mid["joins"][1] >= 3, mid
    assert all(v >= 0.0 for v in (mid["vel_ms"], mid["dye_ms"], mid["blocking_ms"]))
    assert sum(runs[1][2][1]["joins"]) > 0 and sum(sl["joins"][0] for sl in runs[1][2]) > 0
