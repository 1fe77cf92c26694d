"""fs_particle_lines_static (the entry point that allocates nothing) against fs_particle_lines, through the C ABI.

The static variant stages the vertices in a file-scope device array of the library instead of a per-call
hipMalloc.  Bit-exact equality with the allocating entry point (and with the oracle) on two calls in a row, the
second with a smaller input (the array keeps the first call's contents), plus the size limit."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
F = np.float32


def _tracers(n, seed, W, H):
    rng = np.random.default_rng(seed)
    px = rng.uniform(0, W, n).astype(F)
    py = rng.uniform(0, H, n).astype(F)
    vx = rng.uniform(-30, 30, n).astype(F)
    vy = rng.uniform(-30, 30, n).astype(F)
    life = rng.uniform(0.3, 1.0, n).astype(F)
    life[::7] = 0.0   # dead slots are drawn with alpha 0, like any other
    mass = rng.uniform(0.1, 1.0, n).astype(F)
    return [px, py, vx, vy, life, mass, np.zeros(n, F)]


def _lines(lib, L, s, fn, n, view, scale):
    xy, alpha = np.full((n, 4), np.nan, F), np.full(n, np.nan, F)
    L.check(fn(s.ctx, view[0], view[1], scale, xy.ctypes.data_as(C.c_void_p), alpha.ctypes.data_as(C.c_void_p)), s.ctx)
    return xy, alpha


def test_static_lines_equal_allocating_entry_point(oracle):
    from fluidsolver_amd import FluidSolver, ParticleSystem, lib as L
    lib = L.load()
    W, H = 320, 200
    s = FluidSolver(W, H, max_particles=L.LINES_STATIC_MAX + 1000)
    ps = ParticleSystem(s)
    view, scale = (800.0, 600.0), 1.5
    try:
        for n, seed in ((100003, 1), (4099, 2)):   # second call smaller: the array still holds the first call's vertices
            arrs = _tracers(n, seed, W, H)
            ps.set_arrays(*arrs)
            exy, ealpha = _lines(lib, L, s, lib.fs_particle_lines, n, view, scale)
            import torch
            free_before = torch.cuda.mem_get_info()[0]
            gxy, galpha = _lines(lib, L, s, lib.fs_particle_lines_static, n, view, scale)
            # "allocates nothing": the device's free memory is the same after the call as before it
            assert torch.cuda.mem_get_info()[0] == free_before
            assert np.array_equal(gxy, exy) and np.array_equal(galpha, ealpha)
            oxy, oalpha = oracle.particle_lines(*arrs[:5], view, scale)
            assert np.array_equal(gxy, oxy.reshape(n, 4)) and np.array_equal(galpha, oalpha)
            assert not np.isnan(gxy).any() and np.array_equal(galpha, arrs[4])
        # at the limit it still works; one more tracer is refused with a clear error and touches nothing
        n = L.LINES_STATIC_MAX
        ps.set_arrays(*_tracers(n, 3, W, H))
        exy, ealpha = _lines(lib, L, s, lib.fs_particle_lines, n, view, scale)
        gxy, galpha = _lines(lib, L, s, lib.fs_particle_lines_static, n, view, scale)
        assert np.array_equal(gxy, exy) and np.array_equal(galpha, ealpha)
        ps.set_arrays(*_tracers(n + 1, 4, W, H))
        xy, alpha = np.full((n + 1, 4), 7.0, F), np.full(n + 1, 7.0, F)
        rc = lib.fs_particle_lines_static(s.ctx, view[0], view[1], scale, xy.ctypes.data_as(C.c_void_p),
                                          alpha.ctypes.data_as(C.c_void_p))
        assert rc == -1 and b"FS_LINES_STATIC_MAX" in lib.fs_last_error(s.ctx)
        assert (xy == 7.0).all() and (alpha == 7.0).all()
        # a second context on the same GPU shares the array and gets the same pointer's worth of results
        s2 = FluidSolver(64, 64, max_particles=5000)
        ps2 = ParticleSystem(s2)
        arrs = _tracers(5000, 5, 64, 64)
        ps2.set_arrays(*arrs)
        gxy, galpha = _lines(lib, L, s2, lib.fs_particle_lines_static, 5000, view, scale)
        oxy, oalpha = oracle.particle_lines(*arrs[:5], view, scale)
        assert np.array_equal(gxy, oxy.reshape(5000, 4)) and np.array_equal(galpha, oalpha)
        s2.close()
    finally:
        s.close()
