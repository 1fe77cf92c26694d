"""update() of the config-2 step with several SETS of context options, interleaved in one process:
    ab_sets.py W H "OPT_A=1,OPT_B=7" "OPT_A=0" ...   (FS_SWEEPS, FS_POINTERS from the environment)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fluidsolver_amd import FluidSolver, lib as L
from fluidsolver_amd.script import splat
W, H = int(sys.argv[1]), int(sys.argv[2])
sets = [[(getattr(L, kv.split("=")[0]), int(kv.split("=")[1])) for kv in a.split(",") if kv] for a in sys.argv[3:]]
N, P = int(os.environ.get("FS_SWEEPS", "40")), int(os.environ.get("FS_POINTERS", "1"))
frames = int(os.environ.get("FS_FRAMES", "40"))
lib = L.load()
s = FluidSolver(W, H); s.displayMode = None
(s.viscosity, s.diffusion, s.deltaTime, s.vorticityStrength, s.fadeSpeed, s.solverIterations) = (1.5e-4, 0.0, 0.5, 10.0, 0.002, N)
f = [0]
def step():
    for ptr in range(P):
        pos, force, col = splat(W, H, f[0], ptr, P); s.applyForce(pos, force, col)
    s.update(); f[0] += 1
for _ in range(20):
    step()
ms = C.c_float()
for rep in range(4):
    for name, opts in zip(sys.argv[3:], sets):
        for k, v in opts:
            s.set_option(k, v)
        for _ in range(4):
            step()
        s.sync()
        L.check(lib.fs_timer_begin(s.ctx), s.ctx)
        for _ in range(frames):
            step()
        L.check(lib.fs_timer_end(s.ctx, C.byref(ms)), s.ctx)
        if rep: print("%dx%d  %-60s %9.1f us/frame" % (W, H, name, ms.value * 1e3 / frames), flush=True)
