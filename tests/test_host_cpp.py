"""The C++ host mirror (host/FluidSolver.hpp): compiles and links against the C ABI (no GPU), and on
the GPU box produces the same fields as the Python host driving the same calls."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_host_builds():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    ge.build_host_demo()
    assert os.path.exists(os.path.join(ROOT, "host", "demo"))
    src = open(os.path.join(ROOT, "host", "FluidSolver.hpp")).read()
    for name in ("setResolution", "reset", "applyPreset", "addForce", "endInteraction", "update", "texture",
                 "getVelocityAt", "class ParticleSystem", "lineVertices"):
        assert name in src


@pytest.mark.gpu
def test_cpp_host_matches_python_host():
    from fluidsolver_amd import FluidSolver, lib as L
    n, frames = 128, 5
    out = subprocess.run([os.path.join(ROOT, "host", "demo"), str(n), str(frames)], capture_output=True, text=True,
                         timeout=120)
    assert out.returncode == 0, out.stderr
    _, _, sv, st = out.stdout.split()
    s = FluidSolver(n, n)
    s.applyPreset("Turbulent")
    for f in range(frames):
        s.addForce((200.0 + 20.0 * f, 300.0 + 7.0 * f), (640.0, 640.0), time=np.float32(f % 360) / np.float32(360.0))
        s.update()
    s.sync()
    vel = s.read_field(L.FIELD_VELOCITY).astype(np.float64)
    tex = s.texture.astype(np.float64)
    assert float(sv) == pytest.approx(float((vel * vel).sum()), rel=1e-12)
    assert float(st) == pytest.approx(float(tex.sum()), rel=1e-12)
    assert float(sv) > 0
