"""update() at config 2 with the diffuse planes as the projection as one chain / two row bands, A/B in one process."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from fluidsolver_amd import FluidSolver, lib as L
from fluidsolver_amd.script import splat
W = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
H = int(sys.argv[2]) if len(sys.argv) > 2 else W
lib = L.load()
s = FluidSolver(W, H); s.displayMode = None
(s.viscosity, s.diffusion, s.deltaTime, s.vorticityStrength, s.fadeSpeed, s.solverIterations) = (1.5e-4, float(os.environ.get("DIFFUSION", "0")), 0.5, 10.0, 0.002, 40)
f = 0
for _ in range(30):
    pos, force, col = splat(W, H, f); s.applyForce(pos, force, col); s.update(); f += 1
ms = C.c_float()
fields = {}
for d in [int(x) for x in os.environ.get("DELAYS", "").split()]:
    s.set_option(L.OPT_SPLIT_ROWS, 1); s.set_option(L.OPT_BAND_DELAY, d)
    for _ in range(4): s.update()
    s.sync()
    L.check(lib.fs_timer_begin(s.ctx), s.ctx)
    for _ in range(50): s.update()
    L.check(lib.fs_timer_end(s.ctx, C.byref(ms)), s.ctx)
    print("two bands, delay %5d ticks %7.1f us/frame" % (d, ms.value * 1e3 / 50), flush=True)
s.set_option(L.OPT_BAND_DELAY, 100)
for rep in range(3):
    for mode, name in ((0, "one chain"), (1, "two bands")):
        s.set_option(L.OPT_SPLIT_ROWS, mode)
        for _ in range(4): s.update()
        s.sync()
        L.check(lib.fs_timer_begin(s.ctx), s.ctx)
        for _ in range(50): s.update()
        L.check(lib.fs_timer_end(s.ctx, C.byref(ms)), s.ctx)
        if rep: print("%-12s %7.1f us/frame" % (name, ms.value * 1e3 / 50), flush=True)
# same bits: two solvers from the same start
out = []
for mode in (0, 1):
    t = FluidSolver(W, H); t.displayMode = None
    (t.viscosity, t.diffusion, t.deltaTime, t.vorticityStrength, t.fadeSpeed, t.solverIterations) = (1.5e-4, float(os.environ.get("DIFFUSION", "0")), 0.5, 10.0, 0.002, 40)
    t.set_option(L.OPT_SPLIT_ROWS, mode)
    for f in range(8):
        pos, force, col = splat(W, H, f); t.applyForce(pos, force, col); t.update()
    out.append((t.read_field(L.FIELD_VELOCITY), t.read_field(L.FIELD_PRESSURE))); t.close()
print("same bits:", np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1]), float(np.abs(out[0][0]).max()))
