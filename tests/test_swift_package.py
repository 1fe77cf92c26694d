"""The Swift host package (host/swift/) cannot be compiled in this image (no Swift toolchain).  What CAN be
checked without one: every C symbol it calls is declared by include/fluidsolver.h, and the preset table it carries
equals the tested Python table."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "host", "swift", "Sources", "FluidSolverCore")


def _swift():
    return {n: open(os.path.join(SRC, n)).read() for n in sorted(os.listdir(SRC)) if n.endswith(".swift")}


def test_swift_package_calls_only_declared_symbols():
    from test_abi import header_functions
    declared = set(header_functions())
    called = set()
    for text in _swift().values():
        called |= set(re.findall(r"\b(fs_[a-z_A-Z0-9]+)\s*\(", text))
    called -= {"fs_params", "fs_particle_params"}          # struct initialisers, not functions
    assert called and called <= declared, sorted(called - declared)
    header = open(os.path.join(ROOT, "include", "fluidsolver.h")).read()
    for const in ("FS_OK", "FS_FIELD_DISPLAY", "FS_FIELD_VELOCITY"):
        assert const in header
    for cls in ("final class FluidSolver", "final class ParticleSystem", "struct FluidPreset"):
        assert any(cls in t for t in _swift().values())
    assert os.path.exists(os.path.join(ROOT, "host", "swift", "Package.swift"))
    assert "fluidsolver_hip" in open(os.path.join(ROOT, "host", "swift", "Sources", "CFluidSolver", "module.modulemap")).read()


def test_swift_presets_equal_the_python_table():
    from fluidsolver_amd import PRESETS
    text = _swift()["Presets.swift"]
    rows = re.findall(r'FluidPreset\(name: "([^"]+)", viscosity: ([\d.]+), diffusion: ([\d.]+), deltaTime: ([\d.]+), '
                      r'vorticityStrength: ([\d.]+), fadeSpeed: ([\d.]+), solverIterations: (\d+), forceMultiplier: ([\d.]+), '
                      r'colorMultiplier: ([\d.]+), brightness: ([\d.]+)\)', text)
    assert len(rows) == len(PRESETS) == 10
    for r, p in zip(rows, PRESETS):
        assert r[0] == p.name
        got = [float(x) for x in r[1:6]] + [int(r[6])] + [float(x) for x in r[7:]]
        exp = [p.viscosity, p.diffusion, p.deltaTime, p.vorticityStrength, p.fadeSpeed, p.solverIterations,
               p.forceMultiplier, p.colorMultiplier, p.brightness]
        assert got == exp, (p.name, got, exp)
