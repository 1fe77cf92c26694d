"""ctypes binding of the CPU oracle (oracle/liboracle*.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package (fluidsolver_amd) never
imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

FP = C.POINTER(C.c_float)

VELOCITY, DENSITY, PRESSURE, DIVERGENCE, VORTICITY = range(5)
NPLANES = {VELOCITY: 2, DENSITY: 3, PRESSURE: 1, DIVERGENCE: 1, VORTICITY: 1}


class Grid(C.Structure):
    _fields_ = [("W", C.c_int), ("H", C.c_int), ("y0", C.c_int), ("rows", C.c_int),
                ("yb", C.c_int), ("ye", C.c_int)]


class Params(C.Structure):
    _fields_ = [("viscosity", C.c_float), ("diffusion", C.c_float), ("deltaTime", C.c_float),
                ("vorticityStrength", C.c_float), ("fadeSpeed", C.c_float),
                ("solverIterations", C.c_int)]


class ParticleParams(C.Structure):
    _fields_ = [("momentum", C.c_float), ("fluidForce", C.c_float), ("fluidDecayRate", C.c_float),
                ("linkToViscosity", C.c_int), ("viscosity", C.c_float),
                ("windowW", C.c_float), ("windowH", C.c_float)]


def build_oracle():
    """Compile the oracle if the .so files are missing or stale (gcc, seconds)."""
    src = os.path.join(ORACLE_DIR, "fluid_oracle.c")
    for name in ("liboracle.so", "liboracle_omp.so"):
        so = os.path.join(ORACLE_DIR, name)
        if (not os.path.exists(so)) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", ORACLE_DIR], stdout=subprocess.DEVNULL)
            break


def _p(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(FP)


def full_grid(W, H):
    return Grid(W, H, 0, H, 0, H)


class Oracle:
    def __init__(self, omp=False):
        build_oracle()
        self.lib = C.CDLL(os.path.join(ORACLE_DIR, "liboracle_omp.so" if omp else "liboracle.so"))
        L = self.lib
        G = C.POINTER(Grid)
        f = C.c_float
        L.fo_addForces.argtypes = [G, FP, FP, FP, FP, FP] + [f] * 9
        L.fo_diffuse.argtypes = [G, FP, FP, f, f]
        L.fo_advect.argtypes = [G, FP, FP, FP, FP, f]
        L.fo_advect.restype = C.c_long
        L.fo_divergence.argtypes = [G, FP, FP, FP]
        L.fo_pressureSolve.argtypes = [G, FP, FP, FP]
        L.fo_subtractGradient.argtypes = [G, FP, FP, FP]
        L.fo_calculateVorticity.argtypes = [G, FP, FP, FP]
        L.fo_applyVorticity.argtypes = [G, FP, FP, FP, f, f]
        L.fo_fadeDensity.argtypes = [G, FP, f]
        L.fo_clear.argtypes = [G, FP]
        L.fo_renderDensity.argtypes = [G, FP, FP, FP, FP, f]
        L.fo_renderVelocity.argtypes = [G, FP, FP, FP, f]
        L.fo_renderSpeed.argtypes = [G, FP, FP, FP, f]
        L.fo_create.argtypes = [C.c_int, C.c_int]
        L.fo_create.restype = C.c_void_p
        L.fo_destroy.argtypes = [C.c_void_p]
        L.fo_reset.argtypes = [C.c_void_p]
        L.fo_add_force.argtypes = [C.c_void_p] + [f] * 9
        L.fo_update.argtypes = [C.c_void_p, C.POINTER(Params)]
        L.fo_field.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.fo_field.restype = FP
        L.fo_particles_update.argtypes = [G, FP, FP, C.POINTER(ParticleParams), C.c_long] + [FP] * 7
        L.fo_particles_spawn.argtypes = [C.c_long, C.POINTER(C.c_long)] + [FP] * 7 + [f, f, C.c_long, f, C.c_uint]
        L.fo_particles_spawn.restype = None
        L.fo_particle_lines.argtypes = [C.c_long] + [FP] * 5 + [f] * 3 + [FP] * 2
        L.fo_fill_uniform.argtypes = [FP, C.c_long, f, f, C.c_uint]
        L.fo_omp_threads.restype = C.c_int

    # ---- per-kernel wrappers on numpy planes (shape [rows, W], float32) ----
    def threads(self):
        return self.lib.fo_omp_threads()

    def addForces(self, g, u, v, r, gch, b, pos, force, color, radius, dt):
        self.lib.fo_addForces(C.byref(g), _p(u), _p(v), _p(r), _p(gch), _p(b),
                              pos[0], pos[1], force[0], force[1],
                              color[0], color[1], color[2], radius, dt)

    def diffuse(self, g, src, dst, visc, dt):
        self.lib.fo_diffuse(C.byref(g), _p(src), _p(dst), visc, dt)

    def advect(self, g, u, v, src, dst, dt):
        return self.lib.fo_advect(C.byref(g), _p(u), _p(v), _p(src), _p(dst), dt)

    def divergence(self, g, u, v, div):
        self.lib.fo_divergence(C.byref(g), _p(u), _p(v), _p(div))

    def pressureSolve(self, g, div, p, pout):
        self.lib.fo_pressureSolve(C.byref(g), _p(div), _p(p), _p(pout))

    def subtractGradient(self, g, p, u, v):
        self.lib.fo_subtractGradient(C.byref(g), _p(p), _p(u), _p(v))

    def calculateVorticity(self, g, u, v, w):
        self.lib.fo_calculateVorticity(C.byref(g), _p(u), _p(v), _p(w))

    def applyVorticity(self, g, w, u, v, strength, dt):
        self.lib.fo_applyVorticity(C.byref(g), _p(w), _p(u), _p(v), strength, dt)

    def fadeDensity(self, g, d, fade):
        self.lib.fo_fadeDensity(C.byref(g), _p(d), fade)

    def clear(self, g, t):
        self.lib.fo_clear(C.byref(g), _p(t))

    def renderDensity(self, g, r, gch, b, out4, brightness):
        self.lib.fo_renderDensity(C.byref(g), _p(r), _p(gch), _p(b), _p(out4), brightness)

    def renderVelocity(self, g, u, v, out4, scale):
        self.lib.fo_renderVelocity(C.byref(g), _p(u), _p(v), _p(out4), scale)

    def renderSpeed(self, g, u, v, out4, scale):
        self.lib.fo_renderSpeed(C.byref(g), _p(u), _p(v), _p(out4), scale)

    def particle_lines(self, px, py, vx, vy, life, view, scale):
        n = len(px)
        xy, alpha = np.empty((n, 4), np.float32), np.empty(n, np.float32)
        self.lib.fo_particle_lines(n, _p(px), _p(py), _p(vx), _p(vy), _p(life), view[0], view[1], scale, _p(xy), _p(alpha))
        return xy, alpha

    def particles_spawn(self, cap, arrs, x, y, count, radius, seed):
        """arrs: seven arrays of the slots in use; returns the seven arrays after the spawn (dead slots dropped)."""
        n = C.c_long(len(arrs[0]))
        buf = [np.concatenate([np.asarray(a, np.float32), np.zeros(max(0, cap - len(a)), np.float32)]) for a in arrs]
        self.lib.fo_particles_spawn(cap, C.byref(n), *[_p(b) for b in buf], x, y, count, radius, seed)
        return [b[:n.value].copy() for b in buf]

    def particles_update(self, g, u, v, pp, arrs):
        n = arrs[0].shape[0]
        self.lib.fo_particles_update(C.byref(g), _p(u), _p(v), C.byref(pp), n, *[_p(a) for a in arrs])

    def fill_uniform(self, n, lo, hi, seed):
        out = np.empty(n, dtype=np.float32)
        self.lib.fo_fill_uniform(_p(out), n, lo, hi, seed)
        return out

    def solver(self, W, H):
        return OracleSolver(self, W, H)


class OracleSolver:
    """Whole-solver oracle object (FluidSolver restatement, full grid)."""

    def __init__(self, oracle, W, H):
        self.o = oracle
        self.W, self.H = W, H
        self.h = oracle.lib.fo_create(W, H)
        assert self.h

    def __del__(self):
        try:
            self.o.lib.fo_destroy(self.h)
        except Exception:
            pass

    def reset(self):
        self.o.lib.fo_reset(self.h)

    def add_force(self, pos, force, color, radius, dt):
        self.o.lib.fo_add_force(self.h, pos[0], pos[1], force[0], force[1],
                                color[0], color[1], color[2], radius, dt)

    def update(self, params):
        self.o.lib.fo_update(self.h, C.byref(params))

    def plane(self, field, c):
        """numpy VIEW of plane c of a field (current ping-pong side)."""
        ptr = self.o.lib.fo_field(self.h, field, c)
        return np.ctypeslib.as_array(ptr, shape=(self.H, self.W))

    def field(self, field):
        """copy [planes, H, W]"""
        return np.stack([self.plane(field, c).copy() for c in range(NPLANES[field])])

    def set_field(self, field, arr):
        for c in range(NPLANES[field]):
            self.plane(field, c)[...] = arr[c]
