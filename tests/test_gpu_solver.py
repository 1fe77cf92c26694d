"""GPU parity of the whole step (fs_create / fs_add_force / fs_step ...) against the CPU oracle and the
committed golden vectors, through the C ABI.  Bit-exact (see test_gpu_kernels.py for why)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
F = np.float32
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))


def make_solver(W, H, params, radius=5.0, graph=True, depth=None, max_particles=0):
    from fluidsolver_amd import FluidSolver, lib as L
    s = FluidSolver(W, H, max_particles=max_particles)
    (s.viscosity, s.diffusion, s.deltaTime, s.vorticityStrength, s.fadeSpeed, s.solverIterations) = params
    s.forceRadius = radius
    s.displayMode = None
    s.set_option(L.OPT_USE_GRAPH, 1 if graph else 0)
    if depth:
        s.set_option(L.OPT_BLOCK_DEPTH, depth)
    return s


def run_gpu_case(c, graph=True, depth=None):
    from fluidsolver_amd import lib as L
    from fluidsolver_amd.script import splat
    s = make_solver(c["W"], c["H"], c["params"], c["radius"], graph, depth)
    out = {}
    for f in range(c["frames"]):
        if "fixed_splat" in c:
            if f == 0:
                pos, force, col = c["fixed_splat"]
                s.applyForce(pos, force, col)
        else:
            pos, force, col = splat(c["W"], c["H"], f)
            s.applyForce(pos, force, col)
        s.update()
        if f + 1 in c["save_at"]:
            out["vel_%d" % (f + 1)] = s.read_field(L.FIELD_VELOCITY)
            out["den_%d" % (f + 1)] = s.read_field(L.FIELD_DENSITY)
            out["prs_%d" % (f + 1)] = s.read_field(L.FIELD_PRESSURE)
    s.sync()
    s.close()
    return out


@pytest.mark.parametrize("graph", [True, False])
def test_golden_vectors(graph):
    from fluidsolver_amd import lib as L
    from make_golden import CASES
    old = (L.load().fs_config_get(L.OPT_BLOCK_DEPTH_PRESSURE), L.load().fs_config_get(L.OPT_BLOCK_DEPTH_DIFFUSE))
    try:
        for depth in (None, 3):
            for name, case in CASES.items():
                gold = np.load(os.path.join(HERE, "golden", name + ".npz"))
                out = run_gpu_case(case, graph=graph, depth=depth)
                for k in gold.files:
                    assert np.array_equal(out[k], gold[k]), (name, k, graph, depth)
    finally:
        L.load().fs_config_set(L.OPT_BLOCK_DEPTH_PRESSURE, old[0]); L.load().fs_config_set(L.OPT_BLOCK_DEPTH_DIFFUSE, old[1])


def test_fused_and_unfused_frames_are_identical():
    from fluidsolver_amd import lib as L
    from make_golden import CASES
    case = CASES["smoke_50x38"]
    gold = np.load(os.path.join(HERE, "golden", "smoke_50x38.npz"))
    from fluidsolver_amd.script import splat
    for fuse in (0, 1):
        s = make_solver(case["W"], case["H"], case["params"], case["radius"])
        s.set_option(L.OPT_FUSE, fuse)
        for f in range(case["frames"]):
            pos, force, col = splat(case["W"], case["H"], f)
            s.applyForce(pos, force, col)
            s.update()
        assert np.array_equal(s.read_field(L.FIELD_DENSITY), gold["den_6"]), fuse
        assert np.array_equal(s.read_field(L.FIELD_VELOCITY), gold["vel_6"]), fuse


@pytest.mark.parametrize("fuse", [1, 2, 3, 4])
def test_every_fusion_option_equals_the_unfused_frame(oracle, fuse):
    """FS_OPT_FUSE 1 / 2 / 3 / 4 (fade into the dye advection; the projection in one loop; both advections in one
    pass, forced at this small size by 4) against the oracle,
    at a size with interior and edge tiles and N spanning several launches."""
    from fluidsolver_amd import lib as L
    from fluidsolver_amd.script import splat
    from oracle_binding import DENSITY, PRESSURE, VELOCITY, Params
    W, H, params = 700, 330, (1.5e-4, 0.0, 0.5, 10.0, 0.002, 23)
    s = make_solver(W, H, params)
    s.set_option(L.OPT_FUSE, fuse)
    o = oracle.solver(W, H)
    p = Params(*params)
    for f in range(4):
        pos, force, col = splat(W, H, 11 * f)
        s.applyForce(pos, force, col)
        o.add_force(pos, force, col, 5.0, p.deltaTime)
        s.update()
        o.update(p)
    assert np.array_equal(s.read_field(L.FIELD_VELOCITY), o.field(VELOCITY))
    assert np.array_equal(s.read_field(L.FIELD_DENSITY), o.field(DENSITY))
    assert np.array_equal(s.read_field(L.FIELD_PRESSURE), o.field(PRESSURE))


def test_out_of_cache_plan_equals_the_unfused_frame_and_the_oracle(oracle_omp):
    """8192^2: a plane (256 MiB) is as big as the Infinity Cache, so fs_step advects velocity and dye in one pass,
    computes curl + confinement in one pass and keeps the projection in separate passes -- the opposite choices of
    the 3072^2 frame.  The fused frame must equal the unfused one and the (OpenMP) oracle bit for bit."""
    from fluidsolver_amd import lib as L
    from fluidsolver_amd.script import splat
    from oracle_binding import DENSITY, VELOCITY, Params
    W = H = 8192
    params = (1.5e-4, 0.0, 0.5, 10.0, 0.002, 12)
    out = []
    for fuse in (0, 1):
        s = make_solver(W, H, params)
        s.set_option(L.OPT_FUSE, fuse)
        for f in range(3):
            for ptr in range(4):
                pos, force, col = splat(W, H, 9 * f + 1, ptr, 4)
                s.applyForce(pos, force, col)
            s.update()
        out.append((s.read_field(L.FIELD_VELOCITY), s.read_field(L.FIELD_DENSITY)))
        s.close()
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    assert np.abs(out[0][0]).max() > 1.0
    o = oracle_omp.solver(W, H)
    p = Params(*params)
    for f in range(3):
        for ptr in range(4):
            pos, force, col = splat(W, H, 9 * f + 1, ptr, 4)
            o.add_force(pos, force, col, 5.0, 0.5)
        o.update(p)
    assert np.array_equal(out[1][0], o.field(VELOCITY)) and np.array_equal(out[1][1], o.field(DENSITY))


@pytest.mark.parametrize("W,H", [(1, 1), (2, 2), (3, 2), (1, 40), (40, 1), (4, 3), (257, 2), (2, 257), (65, 65)])
def test_degenerate_grids_match_oracle(oracle, W, H):
    """Grids with no interior at all (every cell is a boundary cell), one-cell-wide strips, sizes just past a
    tile / pitch edge: whole frames with every stage on still equal the oracle bit for bit (and do not fault)."""
    import ctypes as C
    from fluidsolver_amd import lib as L
    from oracle_binding import DENSITY, PRESSURE, VELOCITY, Params
    params = (1.5e-4, 0.00001, 0.5, 10.0, 0.002, 11)
    lib = L.load()
    ctx = C.c_void_p()               # the C ABI directly: the FluidSolver mirror never goes below 32 x 32 (Solver:99-106)
    L.check(lib.fs_create(W, H, 0, 0, C.byref(ctx)))
    try:
        fp = L.Params(*params[:5], params[5], L.DISPLAY_NONE, 0.06)
        L.check(lib.fs_set_params(ctx, C.byref(fp)), ctx)
        o = oracle.solver(W, H)
        p = Params(*params)
        for f in range(6):           # enough frames to leave the eager start and replay the recorded graph
            pos, force, col = (W * 0.4 + f, H * 0.6), (7.0, -3.0 - f), (10.0, 20.0, 30.0)
            rgba = (C.c_float * 4)(col[0], col[1], col[2], 0.0)
            L.check(lib.fs_add_force(ctx, pos[0], pos[1], force[0], force[1], rgba, 5.0, 0.5), ctx)
            o.add_force(pos, force, col, 5.0, p.deltaTime)
            L.check(lib.fs_step(ctx), ctx)
            o.update(p)
        L.check(lib.fs_sync(ctx), ctx)
        for fid, oid in ((L.FIELD_VELOCITY, VELOCITY), (L.FIELD_DENSITY, DENSITY), (L.FIELD_PRESSURE, PRESSURE)):
            got = np.empty((L.FIELD_PLANES[fid], H, W), F)
            L.check(lib.fs_read_field(ctx, fid, got.ctypes.data_as(C.c_void_p), got.nbytes), ctx)
            assert np.array_equal(got, o.field(oid)), (fid, got, o.field(oid))
    finally:
        lib.fs_destroy(ctx)


def test_parameters_dragged_every_frame(oracle):
    """A host dragging sliders changes fs_params every frame (graphs would be re-recorded each time: frames run
    eagerly until the parameters settle), then leaves them alone (graph replay): every frame equals the oracle."""
    from fluidsolver_amd import lib as L
    from fluidsolver_amd.script import splat
    from oracle_binding import DENSITY, VELOCITY, Params
    W, H = 200, 136
    s = make_solver(W, H, (1.5e-4, 0.0, 0.5, 10.0, 0.002, 12))
    o = oracle.solver(W, H)
    for f in range(12):
        if f < 7:   # the drag: viscosity, vorticity strength and iteration count all move
            s.viscosity = 1.5e-4 * (1 + f)
            s.vorticityStrength = 10.0 + f
            s.solverIterations = 12 + (f % 3)
        p = Params(s.viscosity, 0.0, 0.5, s.vorticityStrength, 0.002, s.solverIterations)
        pos, force, col = splat(W, H, 9 * f)
        s.applyForce(pos, force, col)
        o.add_force(pos, force, col, 5.0, p.deltaTime)
        s.update()
        o.update(p)
        assert np.array_equal(s.read_field(L.FIELD_VELOCITY), o.field(VELOCITY)), f
    assert np.array_equal(s.read_field(L.FIELD_DENSITY), o.field(DENSITY))


def test_zero_state_stays_zero():
    from fluidsolver_amd import lib as L
    s = make_solver(300, 200, (1.5e-4, 1e-4, 0.5, 10.0, 0.002, 12))
    s.update()
    s.update()
    for f in (L.FIELD_VELOCITY, L.FIELD_DENSITY, L.FIELD_PRESSURE, L.FIELD_DIVERGENCE, L.FIELD_VORTICITY):
        assert not s.read_field(f).any()


@pytest.mark.parametrize("W,H,params", [
    (200, 150, (1.5e-4, 0.0, 0.5, 10.0, 0.002, 40)),       # config-2 physics
    (177, 177, (0.00001, 0.0, 0.59, 0.0, 0.005, 15)),      # "Default" preset at its 177 grid
    (128, 96, (0.00005, 0.00001, 0.4, 10.0, 0.008, 15)),   # "Smoke": dye diffusion on
    (80, 80, (0.0, 0.0, 0.5, 0.0, 0.02, 5)),               # "Minimal": viscosity 0 -> no diffuse
    (96, 64, (1.5e-4, 0.0, 0.5, 0.0, 0.002, 0)),           # N = 0: loops vanish, pressure just cleared
])
def test_frames_match_oracle(oracle, W, H, params):
    from fluidsolver_amd import lib as L
    from fluidsolver_amd.script import splat
    from oracle_binding import DENSITY, PRESSURE, VELOCITY, Params
    s = make_solver(W, H, params)
    o = oracle.solver(W, H)
    p = Params(*params)
    for f in range(12):
        pos, force, col = splat(W, H, f * 7)
        force = (force[0] * F(0.2), force[1] * F(0.2))
        s.applyForce(pos, force, col)
        o.add_force(pos, force, col, 5.0, p.deltaTime)
        s.update()
        o.update(p)
        if f in (0, 5, 11):
            assert np.array_equal(s.read_field(L.FIELD_VELOCITY), o.field(VELOCITY)), f
            assert np.array_equal(s.read_field(L.FIELD_DENSITY), o.field(DENSITY)), f
            assert np.array_equal(s.read_field(L.FIELD_PRESSURE), o.field(PRESSURE)), f
    s.sync()


def test_no_drift_over_100_frames(oracle_omp):
    """north_star's "bounded drift over 100 frames": 100 scripted full-strength frames of BASELINE config 1
    (256 x 256, N = 20, defaults of Solver:139-202) plus vorticity confinement.  The stated tolerance is a max
    relative error of 1e-5 per step; the build is held to zero: velocity, dye and pressure are compared with
    == after frames 1, 10 and 100, so the drift is exactly 0."""
    from fluidsolver_amd import lib as L
    from fluidsolver_amd.script import splat
    from oracle_binding import DENSITY, PRESSURE, VELOCITY, Params
    W = H = 256
    params = (1.5e-4, 0.0, 0.5, 10.0, 0.002, 20)
    s = make_solver(W, H, params)
    o = oracle_omp.solver(W, H)
    p = Params(*params)
    for f in range(100):
        pos, force, col = splat(W, H, f)
        s.applyForce(pos, force, col)
        o.add_force(pos, force, col, 5.0, p.deltaTime)
        s.update()
        o.update(p)
        if f in (0, 9, 99):
            for fid, oid in ((L.FIELD_VELOCITY, VELOCITY), (L.FIELD_DENSITY, DENSITY), (L.FIELD_PRESSURE, PRESSURE)):
                got, exp = s.read_field(fid), o.field(oid)
                assert np.array_equal(got, exp), (f, fid, float(np.abs(got - exp).max()))
    assert np.abs(o.field(VELOCITY)).max() > 1.0 and np.isfinite(o.field(DENSITY)).all()
    s.sync()


def test_update_boundary_invariants():
    from fluidsolver_amd import lib as L
    W, H = 96, 72
    s = make_solver(W, H, (1.5e-4, 0.0, 0.5, 10.0, 0.002, 7))
    rng = np.random.default_rng(9)
    vel = rng.uniform(-2, 2, (2, H, W)).astype(F)
    den = rng.uniform(-1, 3, (3, H, W)).astype(F)
    s.write_field(L.FIELD_VELOCITY, vel)
    s.write_field(L.FIELD_DENSITY, den)
    s.update()
    nv, nd = s.read_field(L.FIELD_VELOCITY), s.read_field(L.FIELD_DENSITY)
    k = F(1.0) - F(0.002)
    for sl in [(slice(None), 0), (slice(None), -1), (0, slice(None)), (-1, slice(None))]:
        for c in range(2):
            assert np.array_equal(nv[c][sl], vel[c][sl])
        for c in range(3):
            assert np.array_equal(nd[c][sl], np.maximum(den[c][sl] * k, F(0)))
        for f in (L.FIELD_PRESSURE, L.FIELD_DIVERGENCE, L.FIELD_VORTICITY):
            assert not s.read_field(f)[0][sl].any()


@pytest.mark.parametrize("in_graph", [1, 0])
def test_splats_as_graph_nodes_match_oracle(oracle, in_graph):
    """FS_OPT_SPLAT_IN_GRAPH: while frames are replayed fs_add_force only queues the splat; fs_step makes the queued splats
    the first nodes of the frame's graph and sets their arguments per replay.  A varying number of splats per frame (0 to
    10: more than the 8 a graph takes), some off the grid, some of radius 0, a read and a timer mark between splat and step
    (which must launch what is queued first): the oracle's bits, and the graphs are really replayed."""
    from fluidsolver_amd import lib as L
    from oracle_binding import DENSITY, PRESSURE, VELOCITY, Params
    W, H, params = 520, 300, (1.5e-4, 0.0, 0.5, 10.0, 0.002, 9)
    s = make_solver(W, H, params)
    s.set_option(L.OPT_SPLAT_IN_GRAPH, in_graph)
    o = oracle.solver(W, H)
    p = Params(*params)
    counts = [1, 1, 1, 2, 2, 0, 3, 3, 1, 10, 2, 2, 1, 1, 3, 3, 0, 0, 2, 2, 1, 1]
    try:
        for f, k in enumerate(counts):
            for j in range(k):
                pos = (40.0 + 37.0 * ((f + 3 * j) % 13), 20.0 + 23.0 * ((2 * f + j) % 12))
                if (f + j) % 7 == 3:
                    pos = (-40.0, 5000.0)                       # off the grid: touches nothing
                radius = 0.0 if (f + j) % 11 == 5 else 4.0 + j  # radius 0: dist < radius never holds
                force, col = (20.0 - 3 * j, -7.0 + f), (10.0 * j, 5.0, 30.0 - f)
                s.forceRadius = radius
                s.applyForce(pos, force, col)
                o.add_force(pos, force, col, radius, p.deltaTime)
            if f == 8:     # a read between splat and step
                assert np.array_equal(s.read_field(L.FIELD_VELOCITY), o.field(VELOCITY))
            if f == 12:    # a timer mark between splat and step: the queued splats are launched in front of it
                L.check(L.load().fs_timer_begin(s.ctx), s.ctx)
            s.update()
            o.update(p)
        s.sync()
        assert np.array_equal(s.read_field(L.FIELD_VELOCITY), o.field(VELOCITY))
        assert np.array_equal(s.read_field(L.FIELD_PRESSURE), o.field(PRESSURE))
        assert np.array_equal(s.read_field(L.FIELD_DENSITY), o.field(DENSITY))
        st = (C.c_long * 3)()
        L.check(L.load().fs_graph_stats(s.ctx, st))
        assert st[0] >= 2, list(st)
    finally:
        s.close()


def test_vorticity_plane_is_refused_when_the_frame_did_not_write_it(oracle):
    """On grids at least 512 wide fs_step computes curl + confinement in one pass and never writes the vorticity plane:
    fs_read_field must say so (FS_ERR_STATE) instead of handing out an old plane; FS_OPT_FUSE 0 keeps the plane, and it
    equals the oracle's; fs_get_option reports what the context runs with."""
    from fluidsolver_amd import lib as L
    from oracle_binding import VORTICITY, Params
    W, H, params = 600, 96, (1.5e-4, 0.0, 0.5, 10.0, 0.002, 6)
    s = make_solver(W, H, params)
    o = oracle.solver(W, H)
    try:
        assert s.get_option(L.OPT_FUSE) == 1 and s.get_option(L.OPT_BLOCK_DEPTH_PRESSURE) == L.load().fs_config_get(L.OPT_BLOCK_DEPTH_PRESSURE)
        s.applyForce((300.0, 40.0), (30.0, -12.0), (100.0, 40.0, 5.0))
        o.add_force((300.0, 40.0), (30.0, -12.0), (100.0, 40.0, 5.0), 5.0, 0.5)
        s.update()
        o.update(Params(*params))
        with pytest.raises(L.FluidSolverError) as e:
            s.read_field(L.FIELD_VORTICITY)
        assert e.value.code == -5 and "vorticity" in str(e.value)
        s.set_option(L.OPT_FUSE, 0)
        s.set_option(L.OPT_BLOCK_DEPTH_PRESSURE, 3)
        assert s.get_option(L.OPT_FUSE) == 0 and s.get_option(L.OPT_BLOCK_DEPTH_PRESSURE) == 3
        s.update()
        o.update(Params(*params))
        assert np.array_equal(s.read_field(L.FIELD_VORTICITY), o.field(VORTICITY))
    finally:
        s.close()


def test_stage_calls_equal_update(oracle):
    """fs_stage() per reference kernel == fs_step(): a host can drive the stages itself."""
    from fluidsolver_amd import lib as L
    W, H, N = 120, 90, 9
    params = (1.5e-4, 1e-5, 0.5, 10.0, 0.002, N)
    a, b = make_solver(W, H, params), make_solver(W, H, params)
    for s in (a, b):
        s.applyForce((60.0, 40.0), (30.0, -12.0), (100.0, 40.0, 5.0))
    a.update()
    for st, n in ((L.STAGE_VORTICITY, 0), (L.STAGE_APPLY_VORTICITY, 0), (L.STAGE_DIFFUSE_VELOCITY, N),
                  (L.STAGE_DIVERGENCE, 0), (L.STAGE_CLEAR_PRESSURE, 0), (L.STAGE_PRESSURE, N),
                  (L.STAGE_SUBTRACT_GRADIENT, 0), (L.STAGE_ADVECT_VELOCITY, 0), (L.STAGE_DIFFUSE_DENSITY, N),
                  (L.STAGE_ADVECT_DENSITY, 0), (L.STAGE_FADE, 0)):
        b.stage(st, n)
    for f in (L.FIELD_VELOCITY, L.FIELD_DENSITY, L.FIELD_PRESSURE, L.FIELD_DIVERGENCE, L.FIELD_VORTICITY):
        assert np.array_equal(a.read_field(f), b.read_field(f))


@pytest.mark.parametrize("W,H,N", [(120, 90, 9), (700, 330, 23), (3072, 3072, 40), (3072, 3072, 31)])
def test_stage_project_equals_the_three_stages(W, H, N):
    """FS_STAGE_PROJECT (divergence + clear + N sweeps + subtractGradient as one loop of launches; two row bands at
    3072^2) == FS_STAGE_DIVERGENCE, FS_STAGE_PRESSURE_CLEARED, FS_STAGE_SUBTRACT_GRADIENT one after the other."""
    from fluidsolver_amd import lib as L
    lib = L.load()
    params = (1.5e-4, 0.0, 0.5, 10.0, 0.002, N)
    a, b = make_solver(W, H, params), make_solver(W, H, params)
    vel = np.empty((2, H, W), np.float32)
    lib.fs_harness_fill_uniform(vel.ctypes.data_as(C.c_void_p), vel.size, -1.0, 1.0, 99)
    for rep in range(2):
        for s in (a, b):
            s.write_field(L.FIELD_VELOCITY, vel)
        a.stage(L.STAGE_PROJECT, N)
        for st, n in ((L.STAGE_DIVERGENCE, 0), (L.STAGE_PRESSURE_CLEARED, N), (L.STAGE_SUBTRACT_GRADIENT, 0)):
            b.stage(st, n)
        assert np.array_equal(a.read_field(L.FIELD_VELOCITY), b.read_field(L.FIELD_VELOCITY))
        assert np.array_equal(a.read_field(L.FIELD_PRESSURE), b.read_field(L.FIELD_PRESSURE))
        # the divergence plane of the fused loop is complete on the interior (the loop stores what its later launches read)
        assert np.array_equal(a.read_field(L.FIELD_DIVERGENCE)[:, 1:-1, 1:-1], b.read_field(L.FIELD_DIVERGENCE)[:, 1:-1, 1:-1])
        vel = a.read_field(L.FIELD_VELOCITY) * np.float32(1.5)
    a.close(); b.close()


def test_value_changes_refresh_the_graph_in_place(oracle):
    """A host that moves a slider: viscosity, deltaTime, fade and vorticity strength change every few frames while the
    frame graphs exist.  The plan is unchanged, so the instantiated graphs get new kernel arguments in place
    (hipGraphExecUpdate) instead of being dropped and instantiated again -- and the fields still equal the oracle's."""
    from fluidsolver_amd import lib as L
    from fluidsolver_amd.script import splat
    from oracle_binding import DENSITY, PRESSURE, VELOCITY, Params
    lib = L.load()
    W, H, N = 300, 200, 12
    base = (1.5e-4, 1e-5, 0.5, 10.0, 0.002, N)
    s = make_solver(W, H, base)
    o = oracle.solver(W, H)
    seq = []
    for f in range(24):
        k = f // 3
        p = (1.5e-4 * (1 + 0.1 * k), 1e-5, 0.5 + 0.01 * k, 10.0 + k, 0.002 + 0.0001 * k, N) if f >= 6 else base
        seq.append(p)
        (s.viscosity, s.diffusion, s.deltaTime, s.vorticityStrength, s.fadeSpeed, s.solverIterations) = p
        pos, force, col = splat(W, H, 5 * f)
        s.applyForce(pos, force, col)
        s.update()
        o.add_force(pos, force, col, 5.0, p[2])
        o.update(Params(*p))
    for fg, fo in ((L.FIELD_VELOCITY, VELOCITY), (L.FIELD_DENSITY, DENSITY), (L.FIELD_PRESSURE, PRESSURE)):
        assert np.array_equal(s.read_field(fg), o.field(fo))
    st = (C.c_long * 3)()
    L.check(lib.fs_graph_stats(s.ctx, st), s.ctx)
    assert st[0] >= 1 and st[1] >= 4, list(st)       # values changed 6 times with graphs in place
    assert st[2] <= 8, list(st)                       # at most one instantiation per ping-pong state
    s.close()


def test_render_modes(oracle):
    from fluidsolver_amd import lib as L
    from oracle_binding import full_grid
    W, H = 100, 60
    s = make_solver(W, H, (1.5e-4, 0.0, 0.5, 10.0, 0.002, 6))
    s.applyForce((50.0, 30.0), (30.0, -12.0), (100.0, 40.0, 5.0))
    for mode in ("Color", "Motion", "Speed", "Vorticity", "Particles Only"):
        s.displayMode = mode
        s.update()
        vel, den, vort = s.read_field(L.FIELD_VELOCITY), s.read_field(L.FIELD_DENSITY), s.read_field(L.FIELD_VORTICITY)
        e = np.zeros((4, H, W), F)
        g = full_grid(W, H)
        z = np.zeros((H, W), F)
        if mode == "Color":
            oracle.renderDensity(g, den[0], den[1], den[2], e, s.brightness)
        elif mode == "Motion":
            oracle.renderVelocity(g, vel[0], vel[1], e, 1.0)
        elif mode == "Speed":
            oracle.renderSpeed(g, vel[0], vel[1], e, 25.0)
        elif mode == "Vorticity":
            oracle.renderDensity(g, vort[0], z, z, e, 10.0)
        assert np.array_equal(s.texture, e), mode


def test_host_api_add_force_and_reset():
    from fluidsolver_amd import FluidSolver, lib as L
    s = FluidSolver(128, 128)
    s.displayMode = None
    s.addForce((100.0, 100.0), (400.0, 400.0), time=0.0)       # first touch: zero force, dye only
    assert not s.read_field(L.FIELD_VELOCITY).any() and s.read_field(L.FIELD_DENSITY)[0].max() == F(50.0)
    s.addForce((110.0, 100.0), (400.0, 400.0), time=0.0)       # 10 px -> 3.2 cells -> force 76.8
    # exact: addForce(at:windowSize:) restated from FluidSolver.swift:409-428 (tests/shader_numpy.py) into the oracle's splat
    import shader_numpy as S
    from oracle_binding import DENSITY, VELOCITY, Oracle
    o = Oracle().solver(128, 128)
    prev = None
    for at in ((100.0, 100.0), (110.0, 100.0)):
        pos, force, col = S.add_force_inputs(at, (400.0, 400.0), (128, 128), prev, s.forceMultiplier, s.colorMultiplier, "Rainbow", 0)
        o.add_force(pos, force, col[:3], s.forceRadius, s.deltaTime)
        prev = pos
    assert force[0] == F(76.8) or abs(float(force[0]) - 76.8) < 1e-4
    assert np.array_equal(s.read_field(L.FIELD_VELOCITY), o.field(VELOCITY)) and s.read_field(L.FIELD_VELOCITY)[0].max() > 30
    assert np.array_equal(s.read_field(L.FIELD_DENSITY), o.field(DENSITY))
    s.endInteraction()
    s.applyPreset("Turbulent")
    assert s.solverIterations == 10 and s.vorticityStrength == 25.0
    s.update()
    s.reset()
    assert not s.read_field(L.FIELD_VELOCITY).any() and not s.read_field(L.FIELD_DENSITY).any()
    s.setResolution(64)
    assert s.read_field(L.FIELD_VELOCITY).shape == (2, 64, 64)
    v = s.getVelocityAt((0.5, 0.5))
    assert v.shape == (2,)


def test_particles_through_context(oracle):
    from fluidsolver_amd import ParticleSystem, lib as L
    from oracle_binding import ParticleParams as OPP, VELOCITY, Params, full_grid
    W, H, n = 160, 120, 20000
    params = (1.5e-4, 0.0, 0.5, 10.0, 0.002, 10)
    s = make_solver(W, H, params, max_particles=n)
    ps = ParticleSystem(s)
    lib = L.load()
    import ctypes as C
    def fill(lo, hi, seed):
        a = np.empty(n, F)
        lib.fs_harness_fill_uniform(a.ctypes.data_as(C.c_void_p), n, lo, hi, seed)
        return a
    arrs = [fill(0, W, 1), fill(0, H, 2), np.zeros(n, F), np.zeros(n, F), fill(0.3, 1.0, 3), fill(0.1, 1.0, 4),
            np.zeros(n, F)]
    ps.set_arrays(*arrs)
    exp = [a.copy() for a in arrs]
    o = oracle.solver(W, H)
    p = Params(*params)
    for f in range(5):
        s.applyForce((80.0, 60.0), (200.0, -90.0), (100.0, 0.0, 0.0))
        o.add_force((80.0, 60.0), (200.0, -90.0), (100.0, 0.0, 0.0), 5.0, 0.5)
        s.update()
        o.update(p)
        ps.update(s, (W, H))
        ov = o.field(VELOCITY)
        oracle.particles_update(full_grid(W, H), ov[0], ov[1], OPP(0.5, 0.6, 3.0, 1, 1.5e-4, W, H), exp)
    got = ps.get_arrays()
    for e, g in zip(exp, got):
        assert np.array_equal(e, g)
    assert (got[2] != 0).any()


def test_full_size_3072_frame_matches_oracle(oracle_omp):
    """BASELINE config 2 at full size: one scripted frame at 3072^2, N=40, against the (OpenMP,
    bit-identical) oracle; plus size-independent properties."""
    from fluidsolver_amd import lib as L
    from fluidsolver_amd.script import splat
    from oracle_binding import DENSITY, PRESSURE, VELOCITY, Params
    W = H = 3072
    params = (1.5e-4, 0.0, 0.5, 10.0, 0.002, 40)
    s = make_solver(W, H, params)
    o = oracle_omp.solver(W, H)
    p = Params(*params)
    for f in range(2):
        pos, force, col = splat(W, H, f + 3)
        s.applyForce(pos, force, col)
        o.add_force(pos, force, col, 5.0, 0.5)
        s.update()
        o.update(p)
    for fg, fo in ((L.FIELD_VELOCITY, VELOCITY), (L.FIELD_DENSITY, DENSITY), (L.FIELD_PRESSURE, PRESSURE)):
        assert np.array_equal(s.read_field(fg), o.field(fo))
    s.sync()


@pytest.mark.parametrize("W,H", [(600, 500), (3072, 1024)])
def test_diffuse_planes_as_two_chains_equal_one_chain(oracle_omp, W, H):
    """FS_OPT_SPLIT_PLANES: the velocity planes (1 + 1) and the dye planes (1 + 2) of the diffuse loops as two
    concurrent launch chains, eager frames and graph replays: same bits as one chain and as the oracle."""
    from fluidsolver_amd import lib as L
    from fluidsolver_amd.script import splat
    from oracle_binding import DENSITY, VELOCITY, Params
    params = (1.5e-4, 0.00002, 0.5, 10.0, 0.002, 21)      # odd number of launches per loop, dye diffusion on
    out = []
    for mode in (0, 1):
        s = make_solver(W, H, params)
        s.set_option(L.OPT_SPLIT_PLANES, mode)
        for f in range(7):
            pos, force, col = splat(W, H, 3 * f + 1)
            s.applyForce(pos, force, col)
            s.update()
        out.append((s.read_field(L.FIELD_VELOCITY), s.read_field(L.FIELD_DENSITY)))
        s.close()
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    o = oracle_omp.solver(W, H)
    p = Params(*params)
    for f in range(7):
        pos, force, col = splat(W, H, 3 * f + 1)
        o.add_force(pos, force, col, 5.0, 0.5)
        o.update(p)
    assert np.array_equal(out[1][0], o.field(VELOCITY)) and np.array_equal(out[1][1], o.field(DENSITY))


@pytest.mark.parametrize("N,depth", [(40, 10), (23, 10), (20, 7)])
def test_projection_as_two_row_bands_equals_one_chain(oracle_omp, N, depth):
    """FS_OPT_SPLIT_ROWS: the fused projection of a 3072 x 3072 frame as two row bands on two streams (band B in
    private planes, its pressure rows copied home on read-back): eager frames and graph replays give the bits of the
    one-chain loop and of the oracle, velocity, dye and pressure, for even and odd launch counts."""
    from fluidsolver_amd import lib as L
    from fluidsolver_amd.script import splat
    from oracle_binding import DENSITY, PRESSURE, VELOCITY, Params
    W = H = 3072
    params = (1.5e-4, 0.0, 0.5, 10.0, 0.002, N)
    out = []
    for mode in (0, 1):
        s = make_solver(W, H, params)
        s.set_option(L.OPT_SPLIT_ROWS, mode)
        s.set_option(L.OPT_BLOCK_DEPTH_PRESSURE, depth)
        for f in range(6):
            pos, force, col = splat(W, H, 4 * f + 2)
            s.applyForce(pos, force, col)
            s.update()
            if f == 2:
                mid_p = s.read_field(L.FIELD_PRESSURE)    # a read-back between frames (settles band B's rows)
        out.append((s.read_field(L.FIELD_VELOCITY), s.read_field(L.FIELD_DENSITY), s.read_field(L.FIELD_PRESSURE), mid_p))
        s.close()
    for a, b in zip(out[0], out[1]):
        assert np.array_equal(a, b)
    o = oracle_omp.solver(W, H)
    p = Params(*params)
    for f in range(6):
        pos, force, col = splat(W, H, 4 * f + 2)
        o.add_force(pos, force, col, 5.0, 0.5)
        o.update(p)
    assert np.array_equal(out[1][0], o.field(VELOCITY)) and np.array_equal(out[1][1], o.field(DENSITY))
    assert np.array_equal(out[1][2], o.field(PRESSURE))


def test_two_bands_beside_the_dye_side_chain(oracle_omp):
    """All three streams of a context at once: the velocity planes as two diffuse chains, then the projection as two
    row bands (streams 1 and 2), with the dye diffusion loop running beside them on the third stream -- eager frames
    and replays equal the oracle."""
    from fluidsolver_amd import lib as L
    from fluidsolver_amd.script import splat
    from oracle_binding import DENSITY, PRESSURE, VELOCITY, Params
    W = H = 3072
    params = (1.5e-4, 0.00002, 0.5, 10.0, 0.002, 20)
    s = make_solver(W, H, params)
    o = oracle_omp.solver(W, H)
    p = Params(*params)
    for f in range(6):
        pos, force, col = splat(W, H, 7 * f + 3)
        s.applyForce(pos, force, col)
        s.update()
        o.add_force(pos, force, col, 5.0, 0.5)
        o.update(p)
    for fg, fo in ((L.FIELD_VELOCITY, VELOCITY), (L.FIELD_DENSITY, DENSITY), (L.FIELD_PRESSURE, PRESSURE)):
        assert np.array_equal(s.read_field(fg), o.field(fo))
    st = (C.c_long * 3)()
    L.check(L.load().fs_graph_stats(s.ctx, st), s.ctx)
    assert st[0] >= 1          # the later frames were replays
    s.close()


def test_config2_fifty_frames_match_oracle(oracle_omp):
    """BASELINE config 2 as bench.py runs it (3072^2, N = 40, one scripted splat per frame, hipGraph replay once
    the parameters have settled), 50 frames: velocity, dye and pressure bit-identical to the oracle at the end and
    at two checkpoints on the way (the jets are well developed by then: the advection gathers across rows)."""
    from fluidsolver_amd import lib as L
    from fluidsolver_amd.script import splat
    from oracle_binding import DENSITY, PRESSURE, VELOCITY, Params
    W = H = 3072
    params = (1.5e-4, 0.0, 0.5, 10.0, 0.002, 40)
    s = make_solver(W, H, params)
    o = oracle_omp.solver(W, H)
    p = Params(*params)
    for f in range(50):
        pos, force, col = splat(W, H, f)
        s.applyForce(pos, force, col)
        o.add_force(pos, force, col, 5.0, 0.5)
        s.update()
        o.update(p)
        if f in (9, 29, 49):
            for fg, fo in ((L.FIELD_VELOCITY, VELOCITY), (L.FIELD_DENSITY, DENSITY), (L.FIELD_PRESSURE, PRESSURE)):
                assert np.array_equal(s.read_field(fg), o.field(fo)), (f, fg)
    assert float(np.abs(o.field(VELOCITY)).max()) > 10.0      # a real flow, not a resting fluid
    s.sync()


def test_config4_grid_frames_match_oracle(oracle_omp):
    """The largest configured grid (BASELINE config 4: 16384^2, N = 40, eight scripted pointers) on one GPU, two
    frames: every plane is 1 GiB, nothing fits the Infinity Cache, the frame runs the out-of-cache plan.  Velocity,
    dye and pressure bit-identical to the (OpenMP) oracle."""
    from fluidsolver_amd import lib as L
    from fluidsolver_amd.script import splat
    from oracle_binding import DENSITY, PRESSURE, VELOCITY, Params
    W = H = 16384
    params = (1.5e-4, 0.0, 0.5, 10.0, 0.002, 40)
    s = make_solver(W, H, params)
    o = oracle_omp.solver(W, H)
    p = Params(*params)
    for f in (1, 2):
        for ptr in range(8):
            pos, force, col = splat(W, H, 5 * f, ptr, 8)
            s.applyForce(pos, force, col)
            o.add_force(pos, force, col, 5.0, 0.5)
        s.update()
        o.update(p)
    for fg, fo in ((L.FIELD_VELOCITY, VELOCITY), (L.FIELD_DENSITY, DENSITY), (L.FIELD_PRESSURE, PRESSURE)):
        got = s.read_field(fg)
        assert np.array_equal(got, o.field(fo)), fg
        assert np.abs(got).max() > 0
        del got
    s.close()


def test_config5_step_matches_oracle(oracle_omp):
    """BASELINE config 5's own step on one GPU: 8192^2 with N = 100 (eight launches per pressure loop, thirteen per
    diffuse loop, the out-of-cache plan), two frames of two scripted pointers: velocity, dye and pressure bit-identical
    to the (OpenMP) oracle.  (The slab form of the same step is compared with this context in test_gpu_native_slabs.py.)"""
    from fluidsolver_amd import lib as L
    from fluidsolver_amd.script import splat
    from oracle_binding import DENSITY, PRESSURE, VELOCITY, Params
    W = H = 8192
    params = (1.5e-4, 0.0, 0.5, 10.0, 0.002, 100)
    s = make_solver(W, H, params)
    o = oracle_omp.solver(W, H)
    p = Params(*params)
    for f in (1, 2):
        for ptr in range(2):
            pos, force, col = splat(W, H, 5 * f, ptr, 2)
            s.applyForce(pos, force, col)
            o.add_force(pos, force, col, 5.0, 0.5)
        s.update()
        o.update(p)
    for fg, fo in ((L.FIELD_VELOCITY, VELOCITY), (L.FIELD_DENSITY, DENSITY), (L.FIELD_PRESSURE, PRESSURE)):
        got = s.read_field(fg)
        assert np.array_equal(got, o.field(fo)), fg
        assert np.abs(got).max() > 0
        del got
    s.close()


def test_full_size_properties():
    """Properties that need no oracle, at 3072^2: linearity of the Jacobi solve in div (power-of-two
    scaling is exact), blocked == unblocked sweeps, zero-velocity advect identity."""
    import ctypes as C
    import torch
    from fluidsolver_amd import lib as L
    from gpu_util import grid, ptr, stream, stride, sync
    lib = L.load()
    W = H = 3072
    g = grid(W, H)
    gen = torch.Generator(device="cuda").manual_seed(5)
    div = (torch.rand((1, H, W), generator=gen, device="cuda") * 2 - 1)
    pa, pb = torch.zeros_like(div), torch.zeros_like(div)
    in_b = C.c_int()
    L.check(lib.fs_k_pressureSolve_n(C.byref(g), ptr(div), ptr(pa), ptr(pb), 40, 1, C.byref(in_b), stream()))
    sync()
    ref = (pb if in_b.value else pa).clone()
    # unblocked: 40 launches of one sweep
    old = (lib.fs_config_get(L.OPT_BLOCK_DEPTH_PRESSURE), lib.fs_config_get(L.OPT_BLOCK_DEPTH_DIFFUSE))
    lib.fs_config_set(L.OPT_BLOCK_DEPTH, 1)
    qa, qb = torch.zeros_like(div), torch.zeros_like(div)
    L.check(lib.fs_k_pressureSolve_n(C.byref(g), ptr(div), ptr(qa), ptr(qb), 40, 1, C.byref(in_b), stream()))
    lib.fs_config_set(L.OPT_BLOCK_DEPTH_PRESSURE, old[0]); lib.fs_config_set(L.OPT_BLOCK_DEPTH_DIFFUSE, old[1])
    sync()
    assert torch.equal(ref, qb if in_b.value else qa)
    # linearity: solve(4*div) == 4*solve(div) exactly
    div4 = div * 4
    L.check(lib.fs_k_pressureSolve_n(C.byref(g), ptr(div4), ptr(qa), ptr(qb), 40, 1, C.byref(in_b), stream()))
    sync()
    assert torch.equal(ref * 4, qb if in_b.value else qa)
    assert float(ref.abs().max()) > 0 and not ref[0, 0].any() and not ref[0, :, 0].any()
    # advect with zero velocity is the identity
    vel = torch.zeros((2, H, W), device="cuda")
    out = torch.empty((1, H, W), device="cuda")
    L.check(lib.fs_k_advect(C.byref(g), ptr(vel, 0), ptr(vel, 1), ptr(div), ptr(out), 1, stride(div), 0.5, None,
                            stream()))
    sync()
    assert torch.equal(out, div)


@pytest.mark.parametrize("W,H,params,frames", [
    (300, 200, (1.5e-4, 0.0, 0.5, 10.0, 0.002, 20), 10),
    (261, 130, (1.5e-4, 1e-5, 0.5, 10.0, 0.002, 7), 6),     # dye diffusion on, odd loop count, width past a tile edge
])
def test_frames_match_second_restatement(W, H, params, frames):
    """fs_add_force / fs_step against the numpy restatement of the shaders (tests/shader_numpy.py), not the C oracle:
    the GPU path agrees bit for bit with BOTH independently written CPU readings of the reference."""
    import shader_numpy as S
    from fluidsolver_amd import lib as L
    from fluidsolver_amd.script import splat
    s = make_solver(W, H, params)
    r = S.Solver(W, H)
    for f in range(frames):
        pos, force, col = splat(W, H, 7 * f)
        s.applyForce(pos, force, col)
        r.add_force(pos, force, col[:3], 5.0, params[2])
        s.update()
        r.update(*params)
    assert np.array_equal(s.read_field(L.FIELD_VELOCITY), np.stack(r.vel))
    assert np.array_equal(s.read_field(L.FIELD_DENSITY), np.stack(r.dens))
    assert np.array_equal(s.read_field(L.FIELD_PRESSURE)[0], r.p)
    assert float(np.abs(np.stack(r.vel)).max()) > 0 and float(np.stack(r.dens).max()) > 0
    s.close()


@pytest.mark.gpu
def test_spawn_on_device_matches_oracle(oracle):
    # fs_particles_spawn (ParticleSystem.addParticles, ParticleSystem.swift:186-231) through the context: interleaved with
    # tracer updates that kill some tracers, up to and beyond the cap, against the oracle's sequential restatement
    import ctypes as C
    from fluidsolver_amd import FluidSolver, ParticleSystem, lib as L
    from oracle_binding import ParticleParams as OPP, full_grid, VELOCITY
    lib = L.load()
    W, H, cap = 160, 120, 5000
    s = FluidSolver(W, H, max_particles=cap)
    ps = ParticleSystem(s)
    ps.momentum = 0.9
    rng = np.random.default_rng(5)
    vel = rng.uniform(-60, 60, (2, H, W)).astype(np.float32)      # fast flow: tracers leave the window and die (:282-288)
    s.write_field(L.FIELD_VELOCITY, vel)
    exp = [np.zeros(0, np.float32) for _ in range(7)]
    try:
        for step, (x, y, count, seed) in enumerate([(80.0, 60.0, 700, 11), (10.0, 5.0, 1500, 12), (150.0, 110.0, 2500, 13),
                                                    (80.0, 60.0, 0, 14), (5.0, 115.0, 1200, 15), (80.0, 60.0, 6000, 16),
                                                    (20.0, 20.0, 1, 17)]):
            L.check(lib.fs_particles_spawn(s.ctx, x, y, count, 15.0, seed), s.ctx)
            exp = oracle.particles_spawn(cap, exp, x, y, count, 15.0, seed)
            assert ps.particleCount == len(exp[0])
            got = ps.get_arrays()
            assert all(np.array_equal(a, b) for a, b in zip(got, exp)), step
            for _ in range(4):
                ps.update(s, (W, H))
                if len(exp[0]):
                    oracle.particles_update(full_grid(W, H), vel[0], vel[1], OPP(0.9, 0.6, 3.0, 1, s.viscosity, W, H), exp)
            got = ps.get_arrays()
            assert all(np.array_equal(a, b) for a, b in zip(got, exp)), step
        assert (exp[4] == 0).any() and (exp[4] != 0).any()          # the run did see dead slots and live tracers
        # a large population: more blocks than one scan segment holds
        big = FluidSolver(W, H, max_particles=300000)
        pb = ParticleSystem(big)
        e2 = [np.zeros(0, np.float32) for _ in range(7)]
        for x, count, seed in ((40.0, 200000, 21), (90.0, 150000, 22)):
            L.check(lib.fs_particles_spawn(big.ctx, x, 60.0, count, 15.0, seed), big.ctx)
            e2 = oracle.particles_spawn(300000, e2, x, 60.0, count, 15.0, seed)
            assert all(np.array_equal(a, b) for a, b in zip(pb.get_arrays(), e2))
        big.close()
    finally:
        s.close()


@pytest.mark.gpu
def test_spawning_every_frame_keeps_the_frame_graphs(oracle):
    # the interactive pattern of ContentView.swift:221-226: ten tracers per pointer event, every frame, tracer update inside the
    # frame's graph.  No call looks at the tracers in between, so no spawn waits for the device and -- once recorded -- the
    # graphs are replayed (no further instantiation); tracers die on the way (fast flow), and the final arrays, by slot,
    # equal the oracle's spawn / update sequence
    from fluidsolver_amd import FluidSolver, ParticleSystem, lib as L
    from oracle_binding import ParticleParams as OPP, VELOCITY, Params, full_grid
    lib = L.load()
    W, H, cap, params = 160, 120, 40000, (1.5e-4, 0.0, 0.5, 0.0, 0.002, 4)
    s = FluidSolver(W, H, max_particles=cap)
    (s.viscosity, s.diffusion, s.deltaTime, s.vorticityStrength, s.fadeSpeed, s.solverIterations) = params
    s.displayMode = None
    s.set_option(L.OPT_TRACER_SORT, 5)
    ps = ParticleSystem(s)
    ps.momentum = 0.9
    ps.stepWithSolver((W, H))
    o = oracle.solver(W, H)
    p = Params(*params)
    exp = [np.zeros(0, np.float32) for _ in range(7)]
    st = (C.c_long * 3)()
    try:
        inst = []
        for f in range(60):
            pos, force, col = (30.0 + 2 * f, 60.0), (900.0, 300.0 - 10 * f), (10.0, 20.0, 30.0)
            s.applyForce(pos, force, col)
            o.add_force(pos, force, col, 5.0, p.deltaTime)
            count = 2000 if f == 0 else 10          # (a first burst so that the arrays are large enough to be kept sorted)
            L.check(lib.fs_particles_spawn(s.ctx, pos[0], pos[1], count, 15.0, 100 + f), s.ctx)
            exp = oracle.particles_spawn(cap, exp, pos[0], pos[1], count, 15.0, 100 + f)
            s.update()
            o.update(p)
            vel = o.field(VELOCITY)
            oracle.particles_update(full_grid(W, H), vel[0], vel[1], OPP(0.9, 0.6, 3.0, 1, s.viscosity, W, H), exp)
            L.check(lib.fs_graph_stats(s.ctx, st))
            inst.append(st[2])
        assert inst[-1] == inst[20], inst            # nothing instantiated over the last 40 frames
        assert ps.particleCount == len(exp[0])
        got = ps.get_arrays()
        for name, a, b in zip("px py vx vy life mass age".split(), got, exp):
            assert np.array_equal(a, b), name
        assert (exp[4] == 0).any() and (exp[4] != 0).any() and len(exp[0]) < 2000 + 59 * 10   # tracers died and were dropped
    finally:
        s.close()


@pytest.mark.gpu
def test_config3_full_size_matches_oracle(oracle_omp):
    # BASELINE config 3 as bench.py runs it: 3072^2, N = 40, 1,048,576 tracers seeded as SURVEY 8(d) (xorshift32
    # 0x9E3779B9), the tracer update as the last node of the frame (FS_OPT_WITH_PARTICLES; frame 3 is a graph replay):
    # fields and all seven tracer arrays against the oracle
    import ctypes as C
    import bench
    from fluidsolver_amd import lib as L
    from fluidsolver_amd.script import splat
    from oracle_binding import DENSITY, VELOCITY, ParticleParams as OPP, Params, full_grid
    W = H = 3072
    wl = bench.WORKLOADS["config3"]
    s, ps = bench.make_single(W, H, wl["params"], wl["particles"])
    try:
        exp = ps.get_arrays()
        assert len(exp[0]) == 1 << 20 and (exp[4] != 0).all()
        o = oracle_omp.solver(W, H)
        p = Params(*wl["params"])
        for f in range(4):
            bench.frame_single(s, ps, f, W, H)
            pos, force, col = splat(W, H, f)
            o.add_force(pos, force, col, 5.0, p.deltaTime)
            o.update(p)
            vel = o.field(VELOCITY)
            oracle_omp.particles_update(full_grid(W, H), vel[0], vel[1], OPP(0.5, 0.6, 3.0, 1, wl["params"][0], W, H), exp)
        s.sync()
        assert np.array_equal(s.read_field(L.FIELD_VELOCITY), o.field(VELOCITY))
        assert np.array_equal(s.read_field(L.FIELD_DENSITY), o.field(DENSITY))
        got = ps.get_arrays()
        for name, a, b in zip("px py vx vy life mass age".split(), got, exp):
            assert np.array_equal(a, b), name
        assert (got[2] != 0).any()   # some tracers did pick up fluid velocity
    finally:
        s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("in_graph", [False, True])
def test_tracer_sort_is_invisible(in_graph):
    # FS_OPT_TRACER_SORT: the tracer arrays are re-ordered by the cache line their lookup reads every few frames (here
    # every 3rd), and every call that exposes slot order restores it.  Against the same run with the sort off: all seven
    # arrays after every kind of access in mid-run -- get, line vertices, spawning at the cap (age order decides who is
    # dropped), a partial fs_particles_set -- and at the end, by slot, bit for bit.
    from fluidsolver_amd import FluidSolver, ParticleSystem, lib as L
    from fluidsolver_amd.script import splat
    W, H, n, cap = 768, 640, 60000, 64000
    rng = np.random.default_rng(11)
    z = np.zeros(n, F)
    arrs = [rng.uniform(0, W, n).astype(F), rng.uniform(0, H, n).astype(F), z, z.copy(),
            rng.uniform(0.3, 1.0, n).astype(F), rng.uniform(0.1, 1.0, n).astype(F), z.copy()]
    arrs[4][::7] = 0.0   # dead slots stay where they are
    runs = []
    for every in (3, 0):
        s = FluidSolver(W, H, max_particles=cap)
        (s.viscosity, s.diffusion, s.deltaTime, s.vorticityStrength, s.fadeSpeed, s.solverIterations) = (1.5e-4, 0.0, 0.5, 10.0, 0.002, 12)
        s.displayMode = None
        s.set_option(L.OPT_TRACER_SORT, every)
        ps = ParticleSystem(s)
        ps.fluidForce, ps.linkToViscosity = 2.0, False
        ps.set_arrays(*arrs)
        if in_graph:
            ps.stepWithSolver((W, H))
        seen = []
        try:
            for f in range(24):
                pos, force, col = splat(W, H, 6 * f + 1)
                s.applyForce(pos, force, col)
                s.update()
                if not in_graph:
                    ps.update(s, (W, H))
                if f == 5:
                    seen.append(ps.get_arrays())
                if f == 8:
                    seen.append(list(ps.lineVertices((W, H), 1.5)))
                if f in (10, 11, 17):
                    ps.addParticles((300.0 + f, 200.0), count=3000, seed=77 + f)     # 60000 + 9000 > cap: the oldest go
                if f == 14:
                    k = 30000
                    L.check(s._lib.fs_particles_set(s.ctx, k, *[a[:k].ctypes.data_as(C.c_void_p) for a in arrs]), s.ctx)
                    seen.append([np.array([ps.particleCount], F)])
            s.sync()
            seen.append(ps.get_arrays())
            runs.append(seen)
        finally:
            s.close()
    for a, b in zip(*runs):
        assert len(a) == len(b)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
    assert (runs[0][-1][2] != 0).any() and len(runs[0][-1][0]) > 16384   # (still above the size below which nothing is sorted)


@pytest.mark.gpu
def test_particles_in_the_reference_layout():
    # fs_particles_set_aos / _get_aos: [Particle] as Swift and MSL lay it out (48-byte stride, ParticleSystem.swift:18-31)
    # against the seven planar arrays: same tracers after an update; `color` bytes of the caller's buffer are left alone
    from fluidsolver_amd import FluidSolver, ParticleSystem, lib as L
    W, H, n = 256, 200, 5000
    rng = np.random.default_rng(5)
    arrs = [rng.uniform(0, W, n).astype(F), rng.uniform(0, H, n).astype(F), rng.uniform(-3, 3, n).astype(F), rng.uniform(-3, 3, n).astype(F),
            rng.uniform(0.3, 1.0, n).astype(F), rng.uniform(0.1, 1.0, n).astype(F), rng.uniform(0, 0.5, n).astype(F)]
    aos = np.zeros((n, 12), F)
    for col, a in zip((0, 1, 2, 3, 8, 9, 10), arrs):
        aos[:, col] = a
    aos[:, 4:8] = (1.0, 1.0, 1.0, 0.75)   # color
    aos[:, 11] = 123.0                    # padding
    out = []
    for layout in ("soa", "aos"):
        s = FluidSolver(W, H, max_particles=n)
        (s.viscosity, s.diffusion, s.deltaTime, s.vorticityStrength, s.fadeSpeed, s.solverIterations) = (1.5e-4, 0.0, 0.5, 10.0, 0.002, 6)
        s.displayMode = None
        ps = ParticleSystem(s)
        try:
            if layout == "soa":
                ps.set_arrays(*arrs)
            else:
                L.check(s._lib.fs_particles_set_aos(s.ctx, n, aos.ctypes.data_as(C.c_void_p), 48), s.ctx)
            s.applyForce((128.0, 100.0), (40.0, 25.0), (10.0, 20.0, 30.0))
            s.update()
            ps.update(s, (W, H))
            if layout == "soa":
                out.append(ps.get_arrays())
            else:
                back = np.full((n, 12), 7.0, F)
                L.check(s._lib.fs_particles_get_aos(s.ctx, n, back.ctypes.data_as(C.c_void_p), 48), s.ctx)
                assert (back[:, 4:8] == 7.0).all() and (back[:, 11] == 7.0).all()
                out.append([back[:, col].copy() for col in (0, 1, 2, 3, 8, 9, 10)])
                assert s._lib.fs_particles_set_aos(s.ctx, n, aos.ctypes.data_as(C.c_void_p), 40) == -1
        finally:
            s.close()
    for a, b in zip(*out):
        assert np.array_equal(a, b)
    assert not np.array_equal(out[0][0], arrs[0])


@pytest.mark.gpu
def test_jacobi_tuning_is_per_context():
    # two contexts in one process: fs_set_option on one (block depth, tile shape) leaves the other's launches alone
    # (they used to share process-wide statics), and both still produce the same bits
    from fluidsolver_amd import FluidSolver, lib as L
    from fluidsolver_amd.script import splat
    lib = L.load()
    W, H = 700, 330
    params = (1.5e-4, 0.0, 0.5, 10.0, 0.002, 29)
    a, b = FluidSolver(W, H), FluidSolver(W, H)
    try:
        for s in (a, b):
            (s.viscosity, s.diffusion, s.deltaTime, s.vorticityStrength, s.fadeSpeed, s.solverIterations) = params
            s.displayMode = None
        a.set_option(L.OPT_BLOCK_DEPTH, 3)
        a.set_option(L.OPT_TILE_VARIANT, 0)
        assert lib.fs_config_get(L.OPT_BLOCK_DEPTH_PRESSURE) == 14 and lib.fs_config_get(L.OPT_TILE_VARIANT) == 8
        import ctypes as C
        from oracle_binding import full_grid   # noqa: F401  (plan dump needs no GPU; the launch counts show the depth in use)
        for f in range(3):
            for s in (a, b):
                pos, force, col = splat(W, H, f)
                s.applyForce(pos, force, col)
                s.update()
        a.sync(); b.sync()
        for fld in (L.FIELD_VELOCITY, L.FIELD_DENSITY, L.FIELD_PRESSURE):
            assert np.array_equal(a.read_field(fld), b.read_field(fld))
        # 29 sweeps: depth 3 -> 10 launches (even: result in the first buffer), depth 14 -> 3 launches (odd): the two contexts'
        # pressure ping-pong sides differ after a frame, which only happens if each used its own depth
        pa, pb = C.c_void_p(), C.c_void_p()
        L.check(lib.fs_field_device_ptr(a.ctx, L.FIELD_PRESSURE, 0, C.byref(pa), None), a.ctx)
        a.update(); a.sync()
        L.check(lib.fs_field_device_ptr(a.ctx, L.FIELD_PRESSURE, 0, C.byref(pb), None), a.ctx)
        assert pa.value == pb.value        # even number of launches: same side every frame
        L.check(lib.fs_field_device_ptr(b.ctx, L.FIELD_PRESSURE, 0, C.byref(pa), None), b.ctx)
        b.update(); b.sync()
        L.check(lib.fs_field_device_ptr(b.ctx, L.FIELD_PRESSURE, 0, C.byref(pb), None), b.ctx)
        assert pa.value != pb.value        # odd number of launches: the sides alternate
    finally:
        a.close(); b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(1, 1 + int(os.environ.get("FS_FUZZ_SEEDS", "6"))))   # (more walks: FS_FUZZ_SEEDS=100)
def test_random_call_sequences_match_oracle(oracle, seed):
    # The host side of a context is a state machine (graphs keyed by buffer roles / tracer bound / pending splats, parameter
    # epochs, splats put off, tracers kept sorted, spawns that put their compaction off).  A random walk over the calls a host
    # makes -- splats, spawns, frames, parameter and option changes, reads in between, resets -- against the oracle making the same
    # moves: fields and all seven tracer arrays (by slot) at random checkpoints and at the end, bit for bit.
    from fluidsolver_amd import FluidSolver, ParticleSystem, lib as L
    from oracle_binding import DENSITY, PRESSURE, VELOCITY, ParticleParams as OPP, Params, full_grid
    lib = L.load()
    rng = np.random.default_rng(1000 + seed)
    W, H, cap = (192, 160, 30000) if seed % 2 else (320, 96, 24000)
    par = dict(viscosity=1.5e-4, diffusion=0.0, deltaTime=0.5, vorticityStrength=10.0, fadeSpeed=0.002, solverIterations=6)
    s = FluidSolver(W, H, max_particles=cap)
    s.displayMode = None
    ps = ParticleSystem(s)
    ps.momentum = 0.8
    in_graph = bool(seed % 3)
    o = oracle.solver(W, H)
    exp = [np.zeros(0, F) for _ in range(7)]

    def push():
        for k, v in par.items():
            setattr(s, k, v)
        if in_graph:
            ps.stepWithSolver((W, H))

    def check(what):
        s.sync()
        assert np.array_equal(s.read_field(L.FIELD_VELOCITY), o.field(VELOCITY)), (what, "velocity")
        assert np.array_equal(s.read_field(L.FIELD_DENSITY), o.field(DENSITY)), (what, "dye")
        assert np.array_equal(s.read_field(L.FIELD_PRESSURE), o.field(PRESSURE)), (what, "pressure")
        assert ps.particleCount == len(exp[0]), what
        for name, a, b in zip("px py vx vy life mass age".split(), ps.get_arrays(), exp):
            assert np.array_equal(a, b), (what, name)

    push()
    s.set_option(L.OPT_TRACER_SORT, int(rng.integers(2, 6)))
    log = []
    try:
        for it in range(70):
            for _ in range(int(rng.integers(0, 3))):
                pos = (float(rng.uniform(-5, W + 5)), float(rng.uniform(-5, H + 5)))
                force = (float(rng.uniform(-900, 900)), float(rng.uniform(-900, 900)))
                col = tuple(float(x) for x in rng.uniform(0, 40, 3))
                s.applyForce(pos, force, col)
                o.add_force(pos, force, col, s.forceRadius, par["deltaTime"])
            r = rng.random()
            if it == 0 or r < 0.45:
                count = 18000 if it == 0 else int(rng.choice([0, 1, 10, 10, 300, 9000]))
                x, y, sd = float(rng.uniform(0, W)), float(rng.uniform(0, H)), int(rng.integers(1, 2 ** 31))
                L.check(lib.fs_particles_spawn(s.ctx, x, y, count, 15.0, sd), s.ctx)
                exp = oracle.particles_spawn(cap, exp, x, y, count, 15.0, sd)
                log.append(("spawn", count))
            r = rng.random()
            if r < 0.08:
                par["viscosity"] = float(rng.choice([1.5e-4, 3.0e-4, 0.0]))
                push(); log.append(("viscosity", par["viscosity"]))
            elif r < 0.14:
                par["vorticityStrength"] = float(rng.choice([0.0, 10.0, 4.0]))
                push(); log.append(("vorticity", par["vorticityStrength"]))
            elif r < 0.18:
                par["solverIterations"] = int(rng.choice([6, 3, 9]))
                push(); log.append(("iterations", par["solverIterations"]))
            elif r < 0.22:
                par["diffusion"] = float(rng.choice([0.0, 2.0e-4]))
                push(); log.append(("diffusion", par["diffusion"]))
            elif r < 0.27:
                opt, val = [(L.OPT_USE_GRAPH, int(rng.integers(0, 2))), (L.OPT_SPLAT_IN_GRAPH, int(rng.integers(0, 2))),
                            (L.OPT_FUSE, int(rng.integers(0, 5))), (L.OPT_SPLIT_ROWS, int(rng.integers(0, 2))),
                            (L.OPT_SPLIT_PLANES, int(rng.integers(0, 2))), (L.OPT_TRACER_SORT, int(rng.integers(0, 6)))][int(rng.integers(0, 6))]
                s.set_option(opt, val); log.append(("option", opt, val))
            elif r < 0.30:
                s.reset(); o.reset(); log.append(("reset",))
            elif r < 0.34:
                list(ps.lineVertices((W, H), 1.5)); log.append(("lines",))
            elif r < 0.38:
                assert ps.particleCount == len(exp[0]), log
            s.update()
            if not in_graph:
                ps.update(s, (W, H))
            o.update(Params(*[par[k] for k in ("viscosity", "diffusion", "deltaTime", "vorticityStrength", "fadeSpeed", "solverIterations")]))
            if len(exp[0]):
                vel = o.field(VELOCITY)
                oracle.particles_update(full_grid(W, H), vel[0], vel[1], OPP(0.8, 0.6, 3.0, 1, par["viscosity"], W, H), exp)
            log.append(("frame", it))
            if rng.random() < 0.12:
                check(log[-12:])
        check(log[-12:])
        assert (exp[4] != 0).any()
    finally:
        s.close()
