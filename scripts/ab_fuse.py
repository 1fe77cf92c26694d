"""update() as a hipGraph at config 2 with the fusion options A/B-ed in one process (interleaved)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fluidsolver_amd import FluidSolver, lib as L
from fluidsolver_amd.script import splat
W = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
H = int(sys.argv[2]) if len(sys.argv) > 2 else W
lib = L.load()
s = FluidSolver(W, H); s.displayMode = None
(s.viscosity, s.diffusion, s.deltaTime, s.vorticityStrength, s.fadeSpeed, s.solverIterations) = (1.5e-4, 0.0, 0.5, 10.0, 0.002, 40)
for f in range(30):
    pos, force, col = splat(W, H, f); s.applyForce(pos, force, col); s.update()
ms = C.c_float()
for rep in range(3):
    for mode, name in ((1, "default size rules"), (2, "fade fused only"), (0, "nothing fused")):
        s.set_option(L.OPT_FUSE, mode)
        for _ in range(4): s.update()
        s.sync()
        L.check(lib.fs_timer_begin(s.ctx), s.ctx)
        for _ in range(50): s.update()
        L.check(lib.fs_timer_end(s.ctx, C.byref(ms)), s.ctx)
        if rep: print("%-24s %7.1f us/frame" % (name, ms.value * 1e3 / 50), flush=True)
