"""update() at config 2 under several option sets, interleaved in one process.
usage: ab_configs.py "NAME=V,NAME=V" "NAME=V" ...   (names are fluidsolver_amd.lib OPT_* constants; "" = defaults)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fluidsolver_amd import FluidSolver, lib as L
from fluidsolver_amd.script import splat
DEFAULTS = {"OPT_TILE_VARIANT_DIFFUSE": 4, "OPT_BLOCK_DEPTH_DIFFUSE": 8, "OPT_BLOCK_DEPTH_PRESSURE": 10}
cfgs = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in a.split(",") if kv) for a in sys.argv[1:]]
W = H = 3072
lib = L.load()
base = {k: lib.fs_config_get(getattr(L, k)) for k in DEFAULTS}
s = FluidSolver(W, H); s.displayMode = None
(s.viscosity, s.diffusion, s.deltaTime, s.vorticityStrength, s.fadeSpeed, s.solverIterations) = (1.5e-4, 0.0, 0.5, 10.0, 0.002, 40)
f = 0
def frames(n):
    global f
    for _ in range(n):
        pos, force, col = splat(W, H, f); s.applyForce(pos, force, col); s.update(); f += 1
frames(30)
ms = C.c_float()
tot = [0.0] * len(cfgs)
R = 3
for rep in range(R + 1):
    for i, cfg in enumerate(cfgs):
        for k, v in {**base, **cfg}.items():
            s.set_option(getattr(L, k), v)
        frames(4); s.sync()
        L.check(lib.fs_timer_begin(s.ctx), s.ctx)
        frames(40)
        L.check(lib.fs_timer_end(s.ctx, C.byref(ms)), s.ctx)
        if rep: tot[i] += ms.value * 1e3 / 40
for i, cfg in enumerate(cfgs):
    print("%-70s %7.1f us/frame" % (sys.argv[1 + i] or "(defaults)", tot[i] / R), flush=True)
