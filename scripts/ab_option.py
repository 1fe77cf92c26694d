"""update() at config 2 (FS_W / FS_H: another grid, FS_FRAMES: frames per timing) with one context option off / on, interleaved A/B in one process:
ab_option.py OPT_NAME [off on]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fluidsolver_amd import FluidSolver, lib as L
from fluidsolver_amd.script import splat
opt = getattr(L, sys.argv[1])
vals = [int(v) for v in sys.argv[2:4]] if len(sys.argv) > 3 else [0, 1]
W = int(os.environ.get("FS_W", "3072")); H = int(os.environ.get("FS_H", str(W)))
n_timed = int(os.environ.get("FS_FRAMES", "40"))
lib = L.load()
s = FluidSolver(W, H); s.displayMode = None
(s.viscosity, s.diffusion, s.deltaTime, s.vorticityStrength, s.fadeSpeed, s.solverIterations) = (1.5e-4, 0.0, 0.5, 10.0, 0.002, 40)
f = 0
for _ in range(30):
    pos, force, col = splat(W, H, f); s.applyForce(pos, force, col); s.update(); f += 1
ms = C.c_float()
for rep in range(4):
    for v in vals:
        s.set_option(opt, v)
        for _ in range(4):
            pos, force, col = splat(W, H, f); s.applyForce(pos, force, col); s.update(); f += 1
        s.sync()
        L.check(lib.fs_timer_begin(s.ctx), s.ctx)
        for _ in range(n_timed):
            pos, force, col = splat(W, H, f); s.applyForce(pos, force, col); s.update(); f += 1
        L.check(lib.fs_timer_end(s.ctx, C.byref(ms)), s.ctx)
        if rep: print("%s = %d   %7.1f us/frame" % (sys.argv[1], v, ms.value * 1e3 / n_timed), flush=True)
