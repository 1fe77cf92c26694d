"""The scalar C oracle against a second, independently written restatement of the same shaders (tests/shader_numpy.py:
whole-array numpy): every kernel on random fields and whole scripted runs must agree BIT FOR BIT.  The reference has
no golden vectors (parity unpinned); this is the N-version check that stands in for them."""
import numpy as np
import pytest

import shader_numpy as S
from oracle_binding import DENSITY, DIVERGENCE, PRESSURE, VELOCITY, VORTICITY, Oracle, Params, full_grid

SIZES = [(37, 29), (64, 64), (130, 67), (3, 3), (5, 3), (2, 7), (1, 1)]


@pytest.fixture(scope="module")
def oracle():
    return Oracle()


def rnd(rng, H, W, scale=1.0):
    return (rng.standard_normal((H, W)) * scale).astype(np.float32)


def same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("W,H", SIZES)
def test_kernels_agree_bit_for_bit(oracle, W, H):
    rng = np.random.default_rng(W * 1000 + H)
    g = full_grid(W, H)
    u, v, w, p, d = (rnd(rng, H, W, s) for s in (3.0, 3.0, 2.0, 1.0, 1.0))
    dye = [np.abs(rnd(rng, H, W, 50.0)) - 5.0 for _ in range(3)]   # some negatives: fade clamps them

    out = np.full((H, W), np.nan, np.float32)
    oracle.divergence(g, u, v, out)
    assert same(out, S.divergence(u, v))
    oracle.calculateVorticity(g, u, v, out)
    assert same(out, S.calculateVorticity(u, v))
    oracle.pressureSolve(g, d, p, out)
    assert same(out, S.pressureSolve(d, p))
    for visc in (1.5e-4, 1e-5, 0.01):
        oracle.diffuse(g, u, out, visc, 0.5)
        assert same(out, S.diffuse(u, visc, 0.5))

    uu, vv = u.copy(), v.copy()
    oracle.subtractGradient(g, p, uu, vv)
    eu, ev = S.subtractGradient(p, u, v)
    assert same(uu, eu) and same(vv, ev)

    uu, vv = u.copy(), v.copy()
    oracle.applyVorticity(g, w, uu, vv, 10.0, 0.5)
    eu, ev = S.applyVorticity(w, u, v, 10.0, 0.5)
    assert same(uu, eu) and same(vv, ev)

    for dt, s in ((0.5, 1.0), (0.5, 40.0), (1.0, 1000.0)):          # sub-cell, many cells, clamped at the walls
        us, vs = (u * s).astype(np.float32), (v * s).astype(np.float32)
        for src in (us, vs, dye[0]):
            assert oracle.advect(g, us, vs, src, out, dt) == 0
            assert same(out, S.advect(us, vs, [src], dt)[0])

    dd = dye[1].copy()
    oracle.fadeDensity(g, dd, 0.002)
    assert same(dd, S.fadeDensity(dye[1], 0.002))

    out4 = np.full((4, H, W), np.nan, np.float32)
    oracle.renderDensity(g, dye[0], dye[1], dye[2], out4, 1.3)
    assert same(out4, np.stack(S.renderDensity(dye, 1.3)))
    oracle.renderVelocity(g, u, v, out4, 0.05)
    assert same(out4, np.stack(S.renderVelocity(u, v, 0.05)))
    oracle.renderSpeed(g, u, v, out4, 0.05)
    assert same(out4, np.stack(S.renderSpeed(u, v, 0.05)))


@pytest.mark.parametrize("pos,radius", [((20.3, 11.7), 5.0), ((0.0, 0.0), 9.5), ((36.0, 28.0), 50.0), ((18.0, 14.0), 0.4)])
def test_splat_agrees_bit_for_bit(oracle, pos, radius):
    W, H = 37, 29
    rng = np.random.default_rng(7)
    g = full_grid(W, H)
    vel = [rnd(rng, H, W) for _ in range(2)]
    dens = [rnd(rng, H, W) for _ in range(3)]
    a = [x.copy() for x in vel + dens]
    oracle.addForces(g, *a, pos, (31.0, -17.5), (100.0, 40.0, 3.0), radius, 0.5)
    S.addForces(vel, dens, pos, (31.0, -17.5), (100.0, 40.0, 3.0), radius, 0.5)
    for x, y in zip(a, vel + dens):
        assert same(x, y)


@pytest.mark.parametrize("W,H,params,frames", [
    (96, 80, (1.5e-4, 0.0, 0.5, 10.0, 0.002, 20), 12),      # the defaults of config 1 + vorticity
    (80, 96, (1.5e-4, 1e-5, 0.5, 10.0, 0.002, 7), 8),       # dye diffusion on, odd iteration count (swap parity)
    (64, 64, (0.0, 0.0, 0.5, 0.0, 0.01, 5), 6),             # viscosity and vorticity off: both branches skipped
])
def test_scripted_frames_agree_bit_for_bit(oracle, W, H, params, frames):
    """splat + update() for several frames: step order, loop counts, swaps and conditions (Solver:477-591)"""
    import math
    o = oracle.solver(W, H)
    s = S.Solver(W, H)
    p = Params(*params)
    prev = None
    for f in range(frames):
        pos = (W * (0.5 + 0.35 * math.sin(2 * math.pi * 3 * f / 60.0)),
               H * (0.5 + 0.35 * math.sin(2 * math.pi * 2 * f / 60.0 + math.pi / 2)))
        force = (0.0, 0.0) if prev is None else ((pos[0] - prev[0]) * 24.0, (pos[1] - prev[1]) * 24.0)
        prev = pos
        col = (100.0 * (f % 3 == 0), 100.0 * (f % 3 == 1), 100.0 * (f % 3 == 2))
        o.add_force(pos, force, col, 5.0, params[2])
        s.add_force(pos, force, col, 5.0, params[2])
        o.update(p)
        s.update(*params)
        assert same(o.field(VELOCITY), np.stack(s.vel)), f
        assert same(o.field(DENSITY), np.stack(s.dens)), f
        assert same(o.field(PRESSURE)[0], s.p), f
        assert same(o.field(DIVERGENCE)[0], s.div), f
        if params[3] > 0:
            assert same(o.field(VORTICITY)[0], s.w), f
    assert float(np.abs(np.stack(s.vel)).max()) > 0 and float(np.stack(s.dens).max()) > 0


@pytest.mark.parametrize("link", [0, 1])
def test_tracer_update_agrees_bit_for_bit(oracle, link):
    from oracle_binding import ParticleParams
    W, H, n = 61, 47, 5000
    rng = np.random.default_rng(11 + link)
    g = full_grid(W, H)
    u, v = rnd(rng, H, W, 30.0), rnd(rng, H, W, 30.0)
    window = (800.0, 600.0)
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)   # noqa: E731
    arrs = [f32(rng.uniform(-60, 860, n)), f32(rng.uniform(-60, 660, n)),        # some start off screen
            f32(rng.standard_normal(n) * 3), f32(rng.standard_normal(n) * 3),
            f32(rng.uniform(0.0, 1.0, n)), f32(rng.uniform(0.1, 1.0, n)), f32(rng.uniform(0.0, 3.0, n))]
    arrs[4][::7] = 0.0            # dead tracers are skipped
    arrs[4][1::7] = 0.01005       # dies by the fade of this update
    got = [a.copy() for a in arrs]
    exp = arrs
    pp = ParticleParams(0.9, 0.6, 0.5, link, 0.004, *window)
    for _ in range(5):
        oracle.particles_update(g, u, v, pp, got)
        exp = S.particles_update(u, v, exp, 0.9, 0.6, 0.5, link, 0.004, window)
        for a, b in zip(got, exp):
            assert same(a, b)
    assert (got[4] == 0).sum() > (arrs[4] == 0).sum()


@pytest.mark.parametrize("name", ["config1_256", "script_64", "smoke_50x38"])
def test_golden_vectors_equal_the_second_restatement(name):
    """The committed golden vectors (tests/golden/*.npz, generated from the C oracle) reproduced by the numpy
    restatement alone: every fixture is the bit-exact output of BOTH restatements."""
    import os
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    from make_golden import CASES
    from fluidsolver_amd.script import splat
    c = CASES[name]
    gold = np.load(os.path.join(here, "golden", name + ".npz"))
    s = S.Solver(c["W"], c["H"])
    seen = 0
    for f in range(c["frames"]):
        if "fixed_splat" in c:
            if f == 0:
                pos, force, col = c["fixed_splat"]
                s.add_force(pos, force, col, c["radius"], c["params"][2])
        else:
            pos, force, col = splat(c["W"], c["H"], f)
            s.add_force(pos, force, col, c["radius"], c["params"][2])
        s.update(*c["params"])
        if f + 1 in c["save_at"]:
            assert same(gold["vel_%d" % (f + 1)], np.stack(s.vel)), (name, f + 1)
            assert same(gold["den_%d" % (f + 1)], np.stack(s.dens)), (name, f + 1)
            assert same(gold["prs_%d" % (f + 1)][0], s.p), (name, f + 1)
            seen += 1
    assert seen == len(c["save_at"])


@pytest.mark.parametrize("cap,n,count", [(100, 60, 30), (100, 90, 30), (50, 50, 10), (40, 30, 100), (10, 0, 3), (64, 64, 0),
                                         (1000, 999, 2), (7, 7, 7)])
def test_spawn_agrees_bit_for_bit(oracle, cap, n, count):
    F = np.float32
    # ParticleSystem.addParticles (ParticleSystem.swift:186-231): the sequential C restatement (remove dead, then per new
    # tracer removeFirst() at the cap + append) against the whole-array numpy one (last `cap` of live + new)
    rng = np.random.default_rng(cap * 1000 + n)
    arrs = [rng.uniform(0, 100, n).astype(F) for _ in range(7)]
    arrs[4][::3] = 0
    for seed in (1234, 0, 0x9E3779B9, 0xFFFFFFFF):
        a = oracle.particles_spawn(cap, arrs, 50.0, 60.0, count, 15.0, seed)
        b = S.particles_spawn(cap, arrs, 50.0, 60.0, count, 15.0, seed)
        assert len(a[0]) == min(cap, int((arrs[4] != 0).sum()) + count)
        assert all(np.array_equal(x, y) for x, y in zip(a, b))


def test_all_palettes_all_phases_agree():
    # ColorPalette.getColor(for:) (ColorPalette.swift:40-128) at the 360 phases addForce can produce (FluidSolver.swift:
    # 422-423): the host mirror's scalar restatement against the whole-array one of this directory, 8 x 360 colours
    from fluidsolver_amd.presets import PALETTES, palette_color
    frames = np.arange(360)
    assert len(PALETTES) == 8
    for name in PALETTES:
        whole = S.palette_colors(name, frames)
        each = np.stack([palette_color(name, np.float32(k) / np.float32(360.0)) for k in frames])
        assert whole.shape == (360, 4) and np.array_equal(whole, each), name
        assert (whole[:, 3] == 1).all() and whole[:, :3].max() <= 1.0 + 1e-6   # (Ocean's red dips to -0.1: 0.2 + 0.3 sin, :49)
    # known answers: rainbow at phase 0 is pure red, at 120/360 pure green; fire starts red and ends yellow-white
    assert S.palette_colors("Rainbow", [0])[0].tolist() == [1, 0, 0, 1] and S.palette_colors("Rainbow", [120])[0].tolist() == [0, 1, 0, 1]
    assert S.palette_colors("Fire", [0])[0].tolist() == [1, 0, 0, 1]
