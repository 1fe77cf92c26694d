"""C-ABI checks that need no GPU: the library loads, exports every symbol include/fluidsolver.h
declares, the ctypes table matches the header, and argument validation rejects bad geometry before
any device work."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "fluidsolver.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fs_[a-zA-Z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from fluidsolver_amd import lib as L
    lib = L.load()
    names = header_functions()
    assert len(names) >= 40
    for n in names:
        assert hasattr(lib, n), "missing export " + n
    assert sorted(L.SIGNATURES) == names


def test_version_and_pitch():
    from fluidsolver_amd import lib as L
    lib = L.load()
    assert b"gfx950" in lib.fs_version()
    assert lib.fs_pitch_for_width(1) == 64 and lib.fs_pitch_for_width(64) == 64
    assert lib.fs_pitch_for_width(65) == 128 and lib.fs_pitch_for_width(3072) == 3072
    assert lib.fs_pitch_for_width(0) == 0


def test_bad_geometry_is_rejected_without_a_gpu():
    from fluidsolver_amd import lib as L
    lib = L.load()
    bad = [L.Grid(0, 10, 64, 0, 10, 0, 10),      # empty
           L.Grid(100, 10, 100, 0, 10, 0, 10),   # pitch not a multiple of 64
           L.Grid(100, 10, 64, 0, 10, 0, 10),    # pitch < width
           L.Grid(64, 10, 64, 5, 10, 5, 15),     # rows leave the domain
           L.Grid(64, 10, 64, 0, 10, 0, 11),     # compute rows outside buffer
           L.Grid(64, 10, 64, 2, 6, 1, 5)]       # yb < y0
    for g in bad:
        assert lib.fs_k_divergence(C.byref(g), None, None, None, None) == -1
        assert L.last_error()
    good = L.Grid(64, 10, 64, 0, 10, 0, 10)
    assert lib.fs_k_divergence(C.byref(good), None, None, None, None) == -1   # null pointers
    assert "null" in L.last_error()
    with pytest.raises(L.FluidSolverError):
        L.check(-1)


def test_default_params_match_reference():
    from fluidsolver_amd import lib as L
    lib = L.load()
    p = L.Params()
    assert lib.fs_default_params(C.byref(p)) == 0
    # FluidSolver.swift:139-202
    assert abs(p.viscosity - 0.00015) < 1e-9 and p.diffusion == 0.0 and p.deltaTime == 0.5
    assert p.vorticityStrength == 0.0 and abs(p.fadeSpeed - 0.002) < 1e-9 and p.solverIterations == 10
    pp = L.ParticleParams()
    assert lib.fs_default_particle_params(C.byref(pp), 640.0, 480.0) == 0
    # ParticleSystem.swift:63-104
    assert pp.momentum == 0.5 and abs(pp.fluidForce - 0.6) < 1e-7 and pp.fluidDecayRate == 3.0
    assert pp.linkToViscosity == 1 and pp.windowW == 640.0 and pp.windowH == 480.0


def test_harness_rng_matches_oracle(oracle):
    import numpy as np
    from fluidsolver_amd import lib as L
    lib = L.load()
    a = np.empty(1000, np.float32)
    assert lib.fs_harness_fill_uniform(a.ctypes.data_as(C.c_void_p), 1000, -1.0, 1.0, 1234) == 0
    b = oracle.fill_uniform(1000, -1.0, 1.0, 1234)
    assert np.array_equal(a, b) and a.min() >= -1.0 and a.max() < 1.0 and a.std() > 0.5


def test_presets_and_palette():
    import numpy as np
    from fluidsolver_amd import PRESETS, PRESETS_BY_NAME, palette_color, PALETTES
    assert len(PRESETS) == 10 and PRESETS_BY_NAME["Default"].solverIterations == 15   # FluidPresets.swift:28-41
    assert PRESETS_BY_NAME["Smoke"].diffusion == 0.00001 and PRESETS_BY_NAME["Minimal"].viscosity == 0.0
    assert np.allclose(palette_color("Rainbow", 0.0), [1, 0, 0, 1])                  # ColorPalette.swift:44
    assert np.allclose(palette_color("Rainbow", 1.0 / 3.0), [0, 1, 0, 1], atol=1e-6)
    assert np.allclose(palette_color("Fire", 0.75), [1, 1, 0.5, 1])
    for p in PALETTES:
        c = palette_color(p, 0.37)
        assert c.shape == (4,) and c[3] == 1.0 and np.isfinite(c).all() and (c[:3] <= 1.0 + 1e-6).all()
    assert palette_color("Plasma", 0.37)[1] < 0   # 0.3 + 0.5*sin(.) goes negative in the reference too (:65)


def test_script_is_deterministic_fp32():
    import numpy as np
    from fluidsolver_amd.script import splat
    pos, force, col = splat(3072, 3072, 0)
    assert pos == (np.float32(1536.0), np.float32(3072 * 0.85)) and force == (0, 0) and col[0] == 100.0
    p1, f1, _ = splat(3072, 3072, 7)
    p0, _, _ = splat(3072, 3072, 6)
    assert f1[0] == (p1[0] - p0[0]) * np.float32(24.0)
    assert all(isinstance(x, np.float32) for x in (*p1, *f1))
