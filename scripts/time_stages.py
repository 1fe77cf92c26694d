"""HIP-event time of each stage of update() on the config-2 state after some scripted frames.
usage: python scripts/time_stages.py [W] [H] [frames]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fluidsolver_amd import FluidSolver, lib as L
from fluidsolver_amd.script import splat
W = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
H = int(sys.argv[2]) if len(sys.argv) > 2 else W
FR = int(sys.argv[3]) if len(sys.argv) > 3 else 30
lib = L.load()
s = FluidSolver(W, H); s.displayMode = None
(s.viscosity, s.diffusion, s.deltaTime, s.vorticityStrength, s.fadeSpeed, s.solverIterations) = (1.5e-4, 0.0, 0.5, 10.0, 0.002, 40)
for f in range(FR):
    pos, force, col = splat(W, H, f); s.applyForce(pos, force, col); s.update()
s.sync()
ms = C.c_float()
def t(stage, n=0, reps=20):
    s.stage(stage, n); s.sync()
    L.check(lib.fs_timer_begin(s.ctx), s.ctx)
    for _ in range(reps):
        L.check(lib.fs_stage(s.ctx, stage, n), s.ctx)
    L.check(lib.fs_timer_end(s.ctx, C.byref(ms)), s.ctx)
    return ms.value * 1e3 / reps
names = [("vorticity", L.STAGE_VORTICITY, 0), ("applyVorticity", L.STAGE_APPLY_VORTICITY, 0), ("diffuse x40", L.STAGE_DIFFUSE_VELOCITY, 40),
         ("divergence", L.STAGE_DIVERGENCE, 0), ("pressure x40", L.STAGE_PRESSURE, 40), ("subtractGradient", L.STAGE_SUBTRACT_GRADIENT, 0),
         ("advect velocity", L.STAGE_ADVECT_VELOCITY, 0), ("advect dye", L.STAGE_ADVECT_DENSITY, 0), ("fade", L.STAGE_FADE, 0)]
tot = 0.0
for rep in range(2):
    for name, st, n in names:
        us = t(st, n)
        if rep: print("%-18s %8.1f us" % (name, us), flush=True)
s.set_option(L.OPT_USE_GRAPH, 1)
L.check(lib.fs_timer_begin(s.ctx), s.ctx)
for f in range(50):
    s.update()
L.check(lib.fs_timer_end(s.ctx, C.byref(ms)), s.ctx)
print("update() graph     %8.1f us" % (ms.value * 1e3 / 50))
s.set_option(L.OPT_USE_GRAPH, 0)
s.update(); s.sync()
import time
t0 = time.perf_counter()
L.check(lib.fs_timer_begin(s.ctx), s.ctx)
for f in range(50):
    s.update()
t1 = time.perf_counter()
L.check(lib.fs_timer_end(s.ctx, C.byref(ms)), s.ctx)
print("update() eager     %8.1f us  (host enqueue %.1f us per frame)" % (ms.value * 1e3 / 50, (t1 - t0) * 1e6 / 50))
