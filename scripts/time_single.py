"""Single context, wall time per frame.  usage: time_single.py [frames] [W] [H]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fluidsolver_amd import FluidSolver, lib as L
from fluidsolver_amd.script import splat
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
W = int(sys.argv[2]) if len(sys.argv) > 2 else 3072
H = int(sys.argv[3]) if len(sys.argv) > 3 else 3072
s = FluidSolver(W, H); s.displayMode = None
(s.viscosity, s.diffusion, s.deltaTime, s.vorticityStrength, s.fadeSpeed, s.solverIterations) = (1.5e-4, 0.0, 0.5, 10.0, 0.002, int(os.environ.get("FS_SWEEPS", "40")))
def step(f):
    pos, force, col = splat(W, H, f); s.applyForce(pos, force, col); s.update()
for f in range(20):
    step(f)
s.sync()
import ctypes as C
ms = C.c_float()
lib = L.load(); L.check(lib.fs_timer_begin(s.ctx), s.ctx)
t0 = time.perf_counter()
for f in range(20, 20 + n):
    step(f)
te = time.perf_counter() - t0
L.check(lib.fs_timer_end(s.ctx, C.byref(ms)), s.ctx)
t = ms.value * 1e-3
s.sync()
tw = time.perf_counter() - t0
print("single %dx%d: %.1f us/frame (host loop %.1f; wall to the end of a sync %.1f)" % (W, H, t / n * 1e6, te / n * 1e6, tw / n * 1e6), flush=True)
s.close()
