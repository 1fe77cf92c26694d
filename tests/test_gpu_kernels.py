"""GPU parity: every fs_k_* entry point against the CPU oracle, bit for bit, through the C ABI.

Tolerance: none.  The HIP library is built with -ffp-contract=off, divide/sqrt are correctly
rounded on both sides and the splat's exp is evaluated in fp64 and rounded once on both sides, so
results must be identical (compared with ==, i.e. -0.0 == +0.0).
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

F = np.float32
SIZES = [(256, 256), (177, 131), (64, 3), (33, 70), (520, 37), (5, 5), (3, 200)]


@pytest.fixture(scope="module")
def lib():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from fluidsolver_amd import lib as L
    return L.load()


def rnd(shape, seed, lo=-2.0, hi=2.0):
    return np.random.default_rng(seed).uniform(lo, hi, shape).astype(F)


def ok(rc):
    from fluidsolver_amd import lib as L
    L.check(rc)


@pytest.mark.parametrize("W,H", SIZES)
def test_divergence_vorticity(lib, oracle, W, H):
    from gpu_util import dev_planes, grid, host_planes, ptr, stream, sync
    from oracle_binding import full_grid
    vel = rnd((2, H, W), 1)
    exp_d, exp_w = np.zeros((H, W), F), np.zeros((H, W), F)
    oracle.divergence(full_grid(W, H), vel[0], vel[1], exp_d)
    oracle.calculateVorticity(full_grid(W, H), vel[0], vel[1], exp_w)
    dv = dev_planes(vel)
    out = dev_planes(np.full((2, H, W), 9.0, F))
    g = grid(W, H)
    ok(lib.fs_k_divergence(C.byref(g), ptr(dv, 0), ptr(dv, 1), ptr(out, 0), stream()))
    ok(lib.fs_k_calculateVorticity(C.byref(g), ptr(dv, 0), ptr(dv, 1), ptr(out, 1), stream()))
    sync()
    got = host_planes(out, W)
    assert np.array_equal(got[0], exp_d)
    assert np.array_equal(got[1], exp_w)


@pytest.mark.parametrize("W,H", SIZES)
def test_subtract_gradient_and_apply_vorticity(lib, oracle, W, H):
    from gpu_util import dev_planes, grid, host_planes, ptr, stream, sync
    from oracle_binding import full_grid
    vel, p, w = rnd((2, H, W), 2), rnd((H, W), 3), rnd((H, W), 4, -5, 5)
    w[H // 2, : W // 2] = 0.0   # zero-gradient cells exercise the +1e-5 guard
    g = grid(W, H)
    e = vel.copy()
    oracle.subtractGradient(full_grid(W, H), p, e[0], e[1])
    dv, dp = dev_planes(vel), dev_planes(p)
    ok(lib.fs_k_subtractGradient(C.byref(g), ptr(dp), ptr(dv, 0), ptr(dv, 1), stream()))
    sync()
    assert np.array_equal(host_planes(dv, W), e)
    # the same pass with the CFL reduction folded in, on a row range (the slab runtime's call): same
    # velocity bits on the computed rows, untouched elsewhere, max |v| of exactly those rows
    import torch
    yb, ye = (1, H - 1) if H > 4 else (0, H)
    e2 = vel.copy()
    from oracle_binding import Grid as OGrid
    oracle.subtractGradient(OGrid(W, H, 0, H, yb, ye), p, e2[0], e2[1])
    dv2 = dev_planes(vel)
    cfl = torch.zeros(1, dtype=torch.float32, device="cuda")
    g2 = grid(W, H, yb=yb, ye=ye)
    ok(lib.fs_k_subtractGradient_cfl(C.byref(g2), ptr(dp), ptr(dv2, 0), ptr(dv2, 1), C.c_void_p(cfl.data_ptr()), stream()))
    sync()
    assert np.array_equal(host_planes(dv2, W), e2)
    assert float(cfl.item()) == float(np.abs(e2[1][yb:ye]).max())
    ref = torch.zeros(1, dtype=torch.float32, device="cuda")
    ok(lib.fs_k_max_abs(C.byref(g2), ptr(dv2, 1), C.c_void_p(ref.data_ptr()), stream()))
    sync()
    assert float(ref.item()) == float(cfl.item())
    e = vel.copy()
    oracle.applyVorticity(full_grid(W, H), w, e[0], e[1], 10.0, 0.5)
    dv, dw = dev_planes(vel), dev_planes(w)
    ok(lib.fs_k_applyVorticity(C.byref(g), ptr(dw), ptr(dv, 0), ptr(dv, 1), 10.0, 0.5, stream()))
    sync()
    assert np.array_equal(host_planes(dv, W), e)


@pytest.mark.parametrize("W,H", SIZES)
def test_fade_clear_render(lib, oracle, W, H):
    from gpu_util import dev_planes, grid, host_planes, ptr, stream, stride, sync
    from oracle_binding import full_grid
    den, vel = rnd((3, H, W), 5, -1, 4), rnd((2, H, W), 6)
    g, og = grid(W, H), full_grid(W, H)
    e = den.copy()
    for c in range(3):
        oracle.fadeDensity(og, e[c], 0.002)
    dd = dev_planes(den)
    ok(lib.fs_k_fadeDensity(C.byref(g), ptr(dd), 3, stride(dd), 0.002, stream()))
    sync()
    assert np.array_equal(host_planes(dd, W), e)
    # render kernels
    dd, dv = dev_planes(den), dev_planes(vel)
    out = dev_planes(np.zeros((4, H, W), F))
    e4 = np.zeros((4, H, W), F)
    oracle.renderDensity(og, den[0], den[1], den[2], e4, 0.06)
    ok(lib.fs_k_renderDensity(C.byref(g), ptr(dd), stride(dd), 3, ptr(out), stride(out), 0.06, stream()))
    sync()
    assert np.array_equal(host_planes(out, W), e4)
    oracle.renderVelocity(og, vel[0], vel[1], e4, 1.0)
    ok(lib.fs_k_renderVelocity(C.byref(g), ptr(dv, 0), ptr(dv, 1), ptr(out), stride(out), 1.0, stream()))
    sync()
    assert np.array_equal(host_planes(out, W), e4)
    oracle.renderSpeed(og, vel[0], vel[1], e4, 25.0)
    ok(lib.fs_k_renderSpeed(C.byref(g), ptr(dv, 0), ptr(dv, 1), ptr(out), stride(out), 25.0, stream()))
    sync()
    assert np.array_equal(host_planes(out, W), e4)
    ok(lib.fs_k_clear(C.byref(g), ptr(out), 4, stride(out), stream()))
    sync()
    assert not host_planes(out, W).any()


@pytest.mark.parametrize("W,H", SIZES)
@pytest.mark.parametrize("speed", [0.0, 3.0, 400.0])
def test_advect(lib, oracle, W, H, speed):
    import torch
    from gpu_util import dev_planes, grid, host_planes, ptr, stream, stride, sync
    from oracle_binding import full_grid
    vel = rnd((2, H, W), 7, -speed, speed) if speed else np.zeros((2, H, W), F)
    den = rnd((3, H, W), 8)
    g, og = grid(W, H), full_grid(W, H)
    err = torch.zeros(1, dtype=torch.int32, device="cuda")
    # dye by velocity (3 planes)
    e = np.zeros((3, H, W), F)
    for c in range(3):
        assert oracle.advect(og, vel[0], vel[1], den[c], e[c], 0.5) == 0
    dv, dd, do = dev_planes(vel), dev_planes(den), dev_planes(np.zeros((3, H, W), F))
    ok(lib.fs_k_advect(C.byref(g), ptr(dv, 0), ptr(dv, 1), ptr(dd), ptr(do), 3, stride(dd), 0.5,
                       C.c_void_p(err.data_ptr()), stream()))
    sync()
    assert np.array_equal(host_planes(do, W), e)
    # velocity by itself (2 planes, in aliases u/v)
    e = np.zeros((2, H, W), F)
    for c in range(2):
        oracle.advect(og, vel[0], vel[1], vel[c], e[c], 0.5)
    do = dev_planes(np.zeros((2, H, W), F))
    ok(lib.fs_k_advect(C.byref(g), ptr(dv, 0), ptr(dv, 1), ptr(dv), ptr(do), 2, stride(dv), 0.5,
                       C.c_void_p(err.data_ptr()), stream()))
    sync()
    assert np.array_equal(host_planes(do, W), e)
    assert int(err.item()) == 0


@pytest.mark.parametrize("W,H,pos,radius", [(64, 64, (32.0, 30.0), 5.0), (177, 131, (3.3, 128.9), 7.5),
                                            (256, 256, (128.0, 128.0), 50.0), (40, 40, (-3.0, 20.0), 6.0),
                                            (40, 40, (200.0, 20.0), 6.0), (33, 70, (16.5, 35.25), 0.5)])
def test_add_forces(lib, oracle, W, H, pos, radius):
    from gpu_util import dev_planes, grid, host_planes, ptr, stream, sync
    from oracle_binding import full_grid
    st = rnd((5, H, W), 9)
    e = st.copy()
    oracle.addForces(full_grid(W, H), e[0], e[1], e[2], e[3], e[4], pos, (24.0, -7.5), (100.0, 50.0, 12.5),
                     radius, 0.5)
    d = dev_planes(st)
    g = grid(W, H)
    ok(lib.fs_k_addForces(C.byref(g), ptr(d, 0), ptr(d, 1), ptr(d, 2), ptr(d, 3), ptr(d, 4), pos[0], pos[1],
                          24.0, -7.5, 100.0, 50.0, 12.5, radius, 0.5, stream()))
    sync()
    assert np.array_equal(host_planes(d, W), e)


def test_splat_kat_on_gpu(lib):
    # radius 5 at an integer position touches exactly 69 cells; centre gets force*dt
    from gpu_util import dev_planes, grid, host_planes, ptr, stream, sync
    W = H = 64
    d = dev_planes(np.zeros((5, H, W), F))
    g = grid(W, H)
    ok(lib.fs_k_addForces(C.byref(g), ptr(d, 0), ptr(d, 1), ptr(d, 2), ptr(d, 3), ptr(d, 4), 32.0, 30.0, 24.0,
                          -7.0, 100.0, 50.0, 25.0, 5.0, 0.5, stream()))
    sync()
    h = host_planes(d, W)
    assert all(np.count_nonzero(h[c]) == 69 for c in range(5))
    assert h[0, 30, 32] == F(12.0) and h[1, 30, 32] == F(-3.5) and h[2, 30, 32] == F(50.0)
    assert h[0, 33, 36] == 0.0


@pytest.mark.parametrize("W,H", SIZES + [(600, 150)])
@pytest.mark.parametrize("n", [1, 2, 7, 8, 9, 20])
def test_pressure_sweeps(lib, oracle, W, H, n):
    from gpu_util import dev_planes, grid, host_planes, ptr, stream, sync
    from oracle_binding import full_grid
    div, p0 = rnd((H, W), 10, -1, 1), rnd((H, W), 11)
    og = full_grid(W, H)
    # oracle: n single sweeps from p0, and from zero
    def oracle_sweeps(p):
        a, b = p.copy(), np.zeros((H, W), F)
        for _ in range(n):
            oracle.pressureSolve(og, div, a, b)
            a, b = b, a
        return a
    e_from_p0, e_from_zero = oracle_sweeps(p0), oracle_sweeps(np.zeros((H, W), F))
    g = grid(W, H)
    dd = dev_planes(div)
    for p_is_zero, exp in ((0, e_from_p0), (1, e_from_zero)):
        pa, pb = dev_planes(p0), dev_planes(np.full((H, W), 7.0, F))
        in_b = C.c_int(-1)
        ok(lib.fs_k_pressureSolve_n(C.byref(g), ptr(dd), ptr(pa), ptr(pb), n, p_is_zero, C.byref(in_b), stream()))
        sync()
        got = host_planes(pb if in_b.value else pa, W)[0]
        assert np.array_equal(got, exp)
    if n == 1:
        pa, pb = dev_planes(p0), dev_planes(np.full((H, W), 7.0, F))
        ok(lib.fs_k_pressureSolve(C.byref(g), ptr(dd), ptr(pa), ptr(pb), stream()))
        sync()
        assert np.array_equal(host_planes(pb, W)[0], e_from_p0)
        assert np.array_equal(host_planes(pa, W)[0], p0)      # input untouched


@pytest.mark.parametrize("depth", [1, 3, 4, 8, 12, 16])
def test_block_depth_does_not_change_bits(lib, oracle, depth):
    # temporal-blocked T-sweep kernel == T single sweeps, bit for bit, for every block depth
    from fluidsolver_amd import lib as L
    from gpu_util import dev_planes, grid, host_planes, ptr, stream, sync
    from oracle_binding import full_grid
    W, H, n = 300, 260, 19
    div = rnd((H, W), 12, -1, 1)
    og = full_grid(W, H)
    a, b = np.zeros((H, W), F), np.zeros((H, W), F)
    for _ in range(n):
        oracle.pressureSolve(og, div, a, b)
        a, b = b, a
    old = (lib.fs_config_get(L.OPT_BLOCK_DEPTH_PRESSURE), lib.fs_config_get(L.OPT_BLOCK_DEPTH_DIFFUSE))
    try:
        ok(lib.fs_config_set(L.OPT_BLOCK_DEPTH, depth))
        g = grid(W, H)
        dd, pa, pb = dev_planes(div), dev_planes(np.zeros((H, W), F)), dev_planes(np.zeros((H, W), F))
        in_b = C.c_int(-1)
        ok(lib.fs_k_pressureSolve_n(C.byref(g), ptr(dd), ptr(pa), ptr(pb), n, 1, C.byref(in_b), stream()))
        sync()
        assert np.array_equal(host_planes(pb if in_b.value else pa, W)[0], a)
    finally:
        lib.fs_config_set(L.OPT_BLOCK_DEPTH_PRESSURE, old[0]); lib.fs_config_set(L.OPT_BLOCK_DEPTH_DIFFUSE, old[1])


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 5, 6, 7])
@pytest.mark.parametrize("depth", [1, 5, 10])
def test_every_tile_variant_matches_the_oracle(lib, oracle, variant, depth):
    # every register-tile shape (rows per lane x waves) of the blocked kernel, both loops, edge and interior tiles
    from fluidsolver_amd import lib as L
    from gpu_util import dev_planes, grid, host_planes, ptr, stream, sync
    from oracle_binding import full_grid
    W, H, n = 700, 330, 13
    div = rnd((H, W), 21, -1, 1)
    vel = rnd((2, H, W), 22, -2, 2)
    og = full_grid(W, H)
    a, b = np.zeros((H, W), F), np.zeros((H, W), F)
    va, vb = vel.copy(), np.zeros_like(vel)
    for _ in range(n):
        oracle.pressureSolve(og, div, a, b)
        a, b = b, a
        for pl in range(2):
            oracle.diffuse(og, va[pl], vb[pl], 1.5e-4, 0.5)
        va, vb = vb, va
    old = (lib.fs_config_get(L.OPT_BLOCK_DEPTH_PRESSURE), lib.fs_config_get(L.OPT_BLOCK_DEPTH_DIFFUSE),
           lib.fs_config_get(L.OPT_TILE_VARIANT))
    try:
        ok(lib.fs_config_set(L.OPT_BLOCK_DEPTH, depth))
        ok(lib.fs_config_set(L.OPT_TILE_VARIANT, variant))
        g = grid(W, H)
        dd, pa, pb = dev_planes(div), dev_planes(np.zeros((H, W), F)), dev_planes(np.zeros((H, W), F))
        in_b = C.c_int(-1)
        ok(lib.fs_k_pressureSolve_n(C.byref(g), ptr(dd), ptr(pa), ptr(pb), n, 1, C.byref(in_b), stream()))
        sync()
        assert np.array_equal(host_planes(pb if in_b.value else pa, W)[0], a)
        da, db = dev_planes(vel), dev_planes(np.zeros_like(vel))
        ok(lib.fs_k_diffuse_n(C.byref(g), ptr(da), ptr(db), 2, da.stride(0), 1.5e-4, 0.5, n, C.byref(in_b), stream()))
        sync()
        assert np.array_equal(host_planes(db if in_b.value else da, W), va)
    finally:
        lib.fs_config_set(L.OPT_BLOCK_DEPTH_PRESSURE, old[0]); lib.fs_config_set(L.OPT_BLOCK_DEPTH_DIFFUSE, old[1])
        lib.fs_config_set(L.OPT_TILE_VARIANT, old[2])


@pytest.mark.parametrize("W,H", SIZES + [(600, 150)])
@pytest.mark.parametrize("n", [1, 5, 8, 17])
def test_diffuse_sweeps(lib, oracle, W, H, n):
    from gpu_util import dev_planes, grid, host_planes, ptr, stream, stride, sync
    from oracle_binding import full_grid
    src = rnd((3, H, W), 13)
    og = full_grid(W, H)
    visc, dt = 1.5e-4, 0.5
    e = np.zeros((3, H, W), F)
    for c in range(3):
        a, b = src[c].copy(), np.zeros((H, W), F)
        for _ in range(n):
            oracle.diffuse(og, a, b, visc, dt)
            a, b = b, a
        e[c] = a
    g = grid(W, H)
    da, db = dev_planes(src), dev_planes(np.full((3, H, W), 7.0, F))
    in_b = C.c_int(-1)
    ok(lib.fs_k_diffuse_n(C.byref(g), ptr(da), ptr(db), 3, stride(da), visc, dt, n, C.byref(in_b), stream()))
    sync()
    assert np.array_equal(host_planes(db if in_b.value else da, W), e)
    if n == 1:
        da, db = dev_planes(src), dev_planes(np.full((3, H, W), 7.0, F))
        ok(lib.fs_k_diffuse(C.byref(g), ptr(da), ptr(db), 3, stride(da), visc, dt, stream()))
        sync()
        assert np.array_equal(host_planes(db, W), e)


def test_slab_kernels_match_full_grid(lib, oracle):
    """Row-slab form (y0/rows/yb/ye) of every stencil kernel equals the oracle's slab form and the
    full-grid result on the computed rows."""
    from gpu_util import dev_planes, grid, host_planes, ptr, stream, stride, sync
    from oracle_binding import Grid as OGrid
    W, H = 200, 180
    vel, p, div, den = rnd((2, H, W), 20), rnd((H, W), 21), rnd((H, W), 22, -1, 1), rnd((3, H, W), 23)
    n = 6
    for (y0, y1, yb, ye) in [(0, 80, 0, 60), (50, 130, 60, 120), (110, 180, 120, 180)]:
        rows = y1 - y0
        og = OGrid(W, H, y0, rows, yb, ye)
        g = grid(W, H, None, y0, rows, yb, ye)
        sl = slice(y0, y1)
        # pressure n sweeps: needs n halo rows (present: >= 10 here)
        a, b = p[sl].copy(), np.zeros((rows, W), F)
        for k in range(n):
            rem = n - 1 - k
            og_k = OGrid(W, H, y0, rows, max(0, yb - rem), min(H, ye + rem))
            oracle.pressureSolve(og_k, div[sl].copy(), a, b)
            a, b = b, a
        dd, pa, pb = dev_planes(div[sl]), dev_planes(p[sl]), dev_planes(np.zeros((rows, W), F))
        in_b = C.c_int(-1)
        ok(lib.fs_k_pressureSolve_n(C.byref(g), ptr(dd), ptr(pa), ptr(pb), n, 0, C.byref(in_b), stream()))
        sync()
        got = host_planes(pb if in_b.value else pa, W)[0]
        assert np.array_equal(got[yb - y0:ye - y0], a[yb - y0:ye - y0])
        # divergence / vorticity / subtract gradient on the slab
        e = np.zeros((rows, W), F)
        oracle.divergence(og, vel[0][sl].copy(), vel[1][sl].copy(), e)
        dv, do = dev_planes(vel[:, sl]), dev_planes(np.zeros((rows, W), F))
        ok(lib.fs_k_divergence(C.byref(g), ptr(dv, 0), ptr(dv, 1), ptr(do), stream()))
        sync()
        assert np.array_equal(host_planes(do, W)[0][yb - y0:ye - y0], e[yb - y0:ye - y0])
        # advect dye with small velocities (halo 10 rows is enough)
        small = (vel * F(2.0)).astype(F)
        e3 = np.zeros((3, rows, W), F)
        for c in range(3):
            assert oracle.advect(og, small[0][sl].copy(), small[1][sl].copy(), den[c][sl].copy(), e3[c], 0.5) == 0
        dv, dd3, do3 = dev_planes(small[:, sl]), dev_planes(den[:, sl]), dev_planes(np.zeros((3, rows, W), F))
        ok(lib.fs_k_advect(C.byref(g), ptr(dv, 0), ptr(dv, 1), ptr(dd3), ptr(do3), 3, stride(dd3), 0.5, None,
                           stream()))
        sync()
        assert np.array_equal(host_planes(do3, W)[:, yb - y0:ye - y0], e3[:, yb - y0:ye - y0])


def test_advect_flags_insufficient_halo(lib):
    import torch
    from gpu_util import dev_planes, grid, ptr, stream, stride, sync
    W, H = 64, 40
    u, v = np.zeros((12, W), F), np.full((12, W), 8.0, F)
    dv, ds, do = dev_planes(np.stack([u, v])), dev_planes(np.ones((1, 12, W), F)), dev_planes(np.zeros((1, 12, W), F))
    err = torch.zeros(1, dtype=torch.int32, device="cuda")
    g = grid(W, H, None, 10, 12, 12, 20)
    ok(lib.fs_k_advect(C.byref(g), ptr(dv, 0), ptr(dv, 1), ptr(ds), ptr(do), 1, stride(ds), 0.5,
                       C.c_void_p(err.data_ptr()), stream()))
    sync()
    assert int(err.item()) != 0


def test_jacobi_rejects_missing_halo(lib):
    from fluidsolver_amd import lib as L
    from gpu_util import dev_planes, grid, ptr, stream
    W, H = 64, 100
    z = dev_planes(np.zeros((3, 20, W), F))
    g = grid(W, H, None, 40, 20, 42, 58)   # 2 halo rows, 5 sweeps asked
    rc = lib.fs_k_pressureSolve_n(C.byref(g), ptr(z, 0), ptr(z, 1), ptr(z, 2), 5, 0, None, stream())
    assert rc == -1 and "need rows" in L.last_error()


def test_max_abs(lib):
    import torch
    from gpu_util import dev_planes, grid, ptr, stream, sync
    W, H = 177, 90
    a = rnd((H, W), 30, -3, 3)
    a[40, 100] = -77.5
    d = dev_planes(a)
    out = torch.zeros(1, dtype=torch.float32, device="cuda")
    g = grid(W, H, None, 0, H, 10, 80)
    ok(lib.fs_k_max_abs(C.byref(g), ptr(d), C.c_void_p(out.data_ptr()), stream()))
    sync()
    assert float(out.item()) == 77.5
    out.zero_()
    g = grid(W, H, None, 0, H, 41, 80)
    ok(lib.fs_k_max_abs(C.byref(g), ptr(d), C.c_void_p(out.data_ptr()), stream()))
    sync()
    assert float(out.item()) == float(np.abs(a[41:80]).max())


def test_particles(lib, oracle):
    from fluidsolver_amd import lib as L
    from gpu_util import dev_planes, grid, ptr, stream, sync
    from oracle_binding import ParticleParams as OPP, full_grid
    import torch
    W, H, n = 300, 200, 100003
    vel = rnd((2, H, W), 31, -40, 40)
    rng = np.random.default_rng(32)
    px = rng.uniform(-20, W + 20, n).astype(F)
    py = rng.uniform(-20, H + 20, n).astype(F)
    vx, vy = rng.uniform(-3, 3, n).astype(F), rng.uniform(-3, 3, n).astype(F)
    life = rng.uniform(0.0, 1.0, n).astype(F)
    life[::7] = 0.0
    life[1::11] = 0.0101
    mass = rng.uniform(0.1, 1.0, n).astype(F)
    age = rng.uniform(0.0, 1.0, n).astype(F)
    host = [px, py, vx, vy, life, mass, age]
    for link in (1, 0):
        exp = [a.copy() for a in host]
        dev = [torch.from_numpy(a.copy()).cuda() for a in host]
        dv = dev_planes(vel)
        g = grid(W, H)
        for _ in range(3):
            oracle.particles_update(full_grid(W, H), vel[0], vel[1], OPP(0.5, 0.6, 3.0, link, 1.5e-4, W, H), exp)
            pp = L.ParticleParams(0.5, 0.6, 3.0, link, 1.5e-4, W, H)
            ok(lib.fs_k_particles_update(C.byref(g), ptr(dv, 0), ptr(dv, 1), C.byref(pp), n,
                                         *[C.c_void_p(t.data_ptr()) for t in dev], stream()))
        sync()
        for e, t in zip(exp, dev):
            assert np.array_equal(e, t.cpu().numpy())


@pytest.mark.parametrize("den", [1.0 + 4.0 * 707.7888, 1.0000001, 3.0, 2831.1553, 1.9999999, 5.0e7, 1.0 + 4.0 * 0.0377])
def test_constant_division_is_ieee_exact_for_all_floats(lib, den):
    """The diffuse kernel divides by the launch-constant (1 + 4a) through an fp64 reciprocal; this checks
    all 2^32 fp32 bit patterns against the IEEE division on the device (0 mismatches, bits compared)."""
    bad = C.c_ulonglong(123)
    ok(lib.fs_selftest_divide(F(den), C.byref(bad)))
    assert bad.value == 0


@pytest.mark.parametrize("W,H", [(177, 131), (256, 256), (33, 70)])
def test_advect_fade_equals_advect_then_fade(lib, oracle, W, H):
    from gpu_util import dev_planes, grid, host_planes, ptr, stream, stride, sync
    from oracle_binding import full_grid
    vel, den = rnd((2, H, W), 40, -6, 6), rnd((3, H, W), 41, -1, 3)
    g, og = grid(W, H), full_grid(W, H)
    e = np.zeros((3, H, W), F)
    for c in range(3):
        oracle.advect(og, vel[0], vel[1], den[c], e[c], 0.5)
        oracle.fadeDensity(og, e[c], 0.002)
    dv, dd, do = dev_planes(vel), dev_planes(den), dev_planes(np.zeros((3, H, W), F))
    ok(lib.fs_k_advect_fade(C.byref(g), ptr(dv, 0), ptr(dv, 1), ptr(dd), ptr(do), 3, stride(dd), 0.5, 0.002, None,
                            stream()))
    sync()
    assert np.array_equal(host_planes(do, W), e)


def test_particle_line_vertices(lib, oracle):
    """particleVertex (Shaders:446-476) as a compute pass: (pos - vel * scale, pos) in NDC, alpha = life."""
    import torch
    n = 10007
    rng = np.random.default_rng(12)
    px, py = rng.uniform(-60, 860, n).astype(F), rng.uniform(-60, 660, n).astype(F)
    vx, vy = rng.uniform(-30, 30, n).astype(F), rng.uniform(-30, 30, n).astype(F)
    life = rng.uniform(0, 1, n).astype(F)
    life[::7] = 0.0
    exy, ealpha = oracle.particle_lines(px, py, vx, vy, life, (800.0, 600.0), 1.5)
    dev = [torch.from_numpy(a).cuda() for a in (px, py, vx, vy, life)]
    xy = torch.full((n, 4), float("nan"), dtype=torch.float32, device="cuda")
    alpha = torch.full((n,), float("nan"), dtype=torch.float32, device="cuda")
    from gpu_util import stream, sync
    ok(lib.fs_k_particle_lines(n, *[C.c_void_p(t.data_ptr()) for t in dev], 800.0, 600.0, 1.5,
                               C.c_void_p(xy.data_ptr()), C.c_void_p(alpha.data_ptr()), stream()))
    sync()
    assert np.array_equal(xy.cpu().numpy(), exy) and np.array_equal(alpha.cpu().numpy(), ealpha)
    # a tracer at rest in the window centre is the degenerate line (0, -0) -> (0, -0)
    one = oracle.particle_lines(np.array([400.0], F), np.array([300.0], F), np.zeros(1, F), np.zeros(1, F), np.ones(1, F),
                                (800.0, 600.0), 1.0)
    assert np.all(one[0] == 0) and one[1][0] == 1.0


@pytest.mark.parametrize("W,H", SIZES + [(600, 150), (1000, 333)])
@pytest.mark.parametrize("n", [1, 4, 10, 13, 27])
def test_pressure_loop_with_gradient_epilogue(lib, oracle, W, H, n):
    """fs_k_pressureSolve_n_gradient == n pressureSolve sweeps + subtractGradient (Solver:533-547), bit for bit in
    p, u and v: the gradient is taken from the last launch's registers, one ring deeper in the tile."""
    from gpu_util import dev_planes, grid, host_planes, ptr, stream, sync
    from oracle_binding import full_grid
    div, vel = rnd((H, W), 31, -1, 1), rnd((2, H, W), 32, -2, 2)
    og = full_grid(W, H)
    a, b = np.zeros((H, W), F), np.zeros((H, W), F)
    for _ in range(n):
        oracle.pressureSolve(og, div, a, b)
        a, b = b, a
    e = vel.copy()
    oracle.subtractGradient(og, a, e[0], e[1])
    g = grid(W, H)
    dd, dv = dev_planes(div), dev_planes(vel)
    pa, pb = dev_planes(rnd((H, W), 33)), dev_planes(np.zeros((H, W), F))   # pa holds garbage: p_is_zero ignores it
    in_b = C.c_int(-1)
    ok(lib.fs_k_pressureSolve_n_gradient(C.byref(g), ptr(dd), ptr(pa), ptr(pb), n, 1, ptr(dv, 0), ptr(dv, 1),
                                         C.byref(in_b), stream()))
    sync()
    assert np.array_equal(host_planes(pb if in_b.value else pa, W)[0], a)
    assert np.array_equal(host_planes(dv, W), e)


def test_pressure_gradient_epilogue_on_a_row_range(lib, oracle):
    """The same on a slab-like buffer: rows [40, 200) of a 260-row domain held with 45 ghost rows, computing rows
    [60, 180): needs n + 1 rows of div around them, touches no velocity row outside [60, 180)."""
    from gpu_util import dev_planes, grid, host_planes, ptr, stream, sync
    from oracle_binding import full_grid
    W, H, n = 333, 260, 12
    div, vel = rnd((H, W), 41, -1, 1), rnd((2, H, W), 42, -2, 2)
    og = full_grid(W, H)
    a, b = np.zeros((H, W), F), np.zeros((H, W), F)
    for _ in range(n):
        oracle.pressureSolve(og, div, a, b)
        a, b = b, a
    e = vel.copy()
    oracle.subtractGradient(og, a, e[0], e[1])
    y0, rows, yb, ye = 40, 160, 60, 180
    g = grid(W, H, y0=y0, rows=rows, yb=yb, ye=ye)
    dd, dv = dev_planes(div[y0:y0 + rows]), dev_planes(vel[:, y0:y0 + rows])
    pa, pb = dev_planes(np.zeros((rows, W), F)), dev_planes(np.zeros((rows, W), F))
    in_b = C.c_int(-1)
    ok(lib.fs_k_pressureSolve_n_gradient(C.byref(g), ptr(dd), ptr(pa), ptr(pb), n, 1, ptr(dv, 0), ptr(dv, 1),
                                         C.byref(in_b), stream()))
    sync()
    got = host_planes(dv, W)
    assert np.array_equal(got[:, yb - y0:ye - y0], e[:, yb:ye])
    assert np.array_equal(got[:, :yb - y0], vel[:, y0:yb]) and np.array_equal(got[:, ye - y0:], vel[:, ye:y0 + rows])
    assert np.array_equal(host_planes(pb if in_b.value else pa, W)[0][yb - y0:ye - y0], a[yb:ye])
    # one ghost row too few is refused, not computed wrongly
    g2 = grid(W, H, y0=y0, rows=rows, yb=y0 + n, ye=ye)
    assert lib.fs_k_pressureSolve_n_gradient(C.byref(g2), ptr(dd), ptr(pa), ptr(pb), n, 1, ptr(dv, 0), ptr(dv, 1),
                                             C.byref(in_b), stream()) == -1


@pytest.mark.parametrize("W,H", SIZES + [(600, 150), (1000, 333)])
@pytest.mark.parametrize("n", [1, 10, 11, 27, 40])
def test_projection_in_one_loop(lib, oracle, W, H, n):
    """fs_k_project_n == divergence + clear + n pressureSolve sweeps + subtractGradient (Solver:522-547), bit for bit
    in div, p, u and v: divergence from the first launch's tile, gradient from the last launch's registers."""
    from gpu_util import dev_planes, grid, host_planes, ptr, stream, sync
    from oracle_binding import full_grid
    vel = rnd((2, H, W), 52, -2, 2)
    og = full_grid(W, H)
    div = np.zeros((H, W), F)
    oracle.divergence(og, vel[0], vel[1], div)
    a, b = np.zeros((H, W), F), np.zeros((H, W), F)
    for _ in range(n):
        oracle.pressureSolve(og, div, a, b)
        a, b = b, a
    e = vel.copy()
    oracle.subtractGradient(og, a, e[0], e[1])
    g = grid(W, H)
    dv, dd = dev_planes(vel), dev_planes(rnd((H, W), 53))                   # div holds garbage beforehand
    pa, pb = dev_planes(rnd((H, W), 54)), dev_planes(rnd((H, W), 55))       # and so do both pressure planes
    in_b = C.c_int(-1)
    ok(lib.fs_k_project_n(C.byref(g), ptr(dv, 0), ptr(dv, 1), ptr(dd), ptr(pa), ptr(pb), n, C.byref(in_b), stream()))
    sync()
    assert np.array_equal(host_planes(dd, W)[0], div)
    assert np.array_equal(host_planes(pb if in_b.value else pa, W)[0], a)
    assert np.array_equal(host_planes(dv, W), e)


@pytest.mark.parametrize("n", [3, 12, 25])
def test_projection_with_cfl_on_a_row_range(lib, oracle, n):
    """fs_k_project_n_cfl as a slab calls it: rows [60, 180) of a 260-row domain from a buffer holding rows
    [30, 215) -- p, u, v of those rows equal the unsplit oracle's, no other velocity row is touched, and the CFL
    value is max |v| of exactly those rows after the update."""
    import torch
    from gpu_util import dev_planes, grid, host_planes, ptr, stream, sync
    from oracle_binding import full_grid
    W, H = 333, 260
    vel = rnd((2, H, W), 62, -2, 2)
    og = full_grid(W, H)
    div = np.zeros((H, W), F)
    oracle.divergence(og, vel[0], vel[1], div)
    a, b = np.zeros((H, W), F), np.zeros((H, W), F)
    for _ in range(n):
        oracle.pressureSolve(og, div, a, b)
        a, b = b, a
    e = vel.copy()
    oracle.subtractGradient(og, a, e[0], e[1])
    y0, rows, yb, ye = 30, 185, 60, 180
    g = grid(W, H, y0=y0, rows=rows, yb=yb, ye=ye)
    dv, dd = dev_planes(vel[:, y0:y0 + rows]), dev_planes(rnd((rows, W), 63))
    pa, pb = dev_planes(rnd((rows, W), 64)), dev_planes(rnd((rows, W), 65))
    cfl = torch.zeros(1, dtype=torch.float32, device="cuda")
    in_b = C.c_int(-1)
    ok(lib.fs_k_project_n_cfl(C.byref(g), ptr(dv, 0), ptr(dv, 1), ptr(dd), ptr(pa), ptr(pb), n, C.c_void_p(cfl.data_ptr()),
                              C.byref(in_b), stream()))
    sync()
    got = host_planes(dv, W)
    assert np.array_equal(got[:, yb - y0:ye - y0], e[:, yb:ye])
    assert np.array_equal(got[:, :yb - y0], vel[:, y0:yb]) and np.array_equal(got[:, ye - y0:], vel[:, ye:y0 + rows])
    assert np.array_equal(host_planes(pb if in_b.value else pa, W)[0][yb - y0:ye - y0], a[yb:ye])
    assert float(cfl.item()) == float(np.abs(e[1][yb:ye]).max())


@pytest.mark.parametrize("W,H", SIZES + [(600, 150)])
@pytest.mark.parametrize("speed", [0.3, 6.0, 40.0])
@pytest.mark.parametrize("fade", [0, 1])
def test_both_advections_in_one_pass(lib, oracle, W, H, speed, fade):
    """fs_k_advect_both == advect(velocity by itself) then advect(dye by the NEW velocity) [+ fadeDensity]
    (Solver:550-591), bit for bit, for slow, fast and domain-crossing flows."""
    from gpu_util import dev_planes, grid, host_planes, ptr, stream, stride, sync
    from oracle_binding import full_grid
    vel, den = rnd((2, H, W), 70, -speed, speed), rnd((3, H, W), 71, -1, 3)
    g, og = grid(W, H), full_grid(W, H)
    ev = np.zeros((2, H, W), F)
    for c in range(2):
        oracle.advect(og, vel[0], vel[1], vel[c], ev[c], 0.5)
    ed = np.zeros((3, H, W), F)
    for c in range(3):
        oracle.advect(og, ev[0], ev[1], den[c], ed[c], 0.5)
        if fade:
            oracle.fadeDensity(og, ed[c], 0.002)
    dv, dvo = dev_planes(vel), dev_planes(np.zeros((2, H, W), F))
    dd, ddo = dev_planes(den), dev_planes(np.zeros((3, H, W), F))
    ok(lib.fs_k_advect_both(C.byref(g), ptr(dv, 0), ptr(dv, 1), ptr(dvo, 0), ptr(dvo, 1), ptr(dd), ptr(ddo), stride(dd), 0.5,
                            fade, 0.002, None, stream()))
    sync()
    assert np.array_equal(host_planes(dvo, W), ev)
    assert np.array_equal(host_planes(ddo, W), ed)


@pytest.mark.parametrize("W,H", SIZES + [(600, 150), (1, 9), (2, 2)])
def test_vorticity_and_confinement_in_one_pass(lib, oracle, W, H):
    """fs_k_vorticity_apply == calculateVorticity + applyVorticity (Solver:486-502), out of place, bit for bit --
    on the whole grid and on a row range of a slab-like buffer."""
    from gpu_util import dev_planes, grid, host_planes, ptr, stream, sync
    from oracle_binding import full_grid
    vel = rnd((2, H, W), 81, -3, 3)
    vel[:, H // 2, : W // 2] = 0.0        # zero-gradient cells exercise the +1e-5 guard
    og = full_grid(W, H)
    w = np.zeros((H, W), F)
    oracle.calculateVorticity(og, vel[0], vel[1], w)
    e = vel.copy()
    oracle.applyVorticity(og, w, e[0], e[1], 10.0, 0.5)
    g = grid(W, H)
    dv, do = dev_planes(vel), dev_planes(rnd((2, H, W), 82))
    ok(lib.fs_k_vorticity_apply(C.byref(g), ptr(dv, 0), ptr(dv, 1), ptr(do, 0), ptr(do, 1), 10.0, 0.5, stream()))
    sync()
    assert np.array_equal(host_planes(do, W), e)
    assert np.array_equal(host_planes(dv, W), vel)                 # the input is left alone
    assert lib.fs_k_vorticity_apply(C.byref(g), ptr(dv, 0), ptr(dv, 1), ptr(dv, 0), ptr(dv, 1), 10.0, 0.5, stream()) == -1
    if H >= 12:
        y0, rows = 2, H - 4
        yb, ye = y0 + 2, y0 + rows - 2
        g2 = grid(W, H, y0=y0, rows=rows, yb=yb, ye=ye)
        dv2, do2 = dev_planes(vel[:, y0:y0 + rows]), dev_planes(np.zeros((2, rows, W), F))
        ok(lib.fs_k_vorticity_apply(C.byref(g2), ptr(dv2, 0), ptr(dv2, 1), ptr(do2, 0), ptr(do2, 1), 10.0, 0.5, stream()))
        sync()
        assert np.array_equal(host_planes(do2, W)[:, yb - y0:ye - y0], e[:, yb:ye])
