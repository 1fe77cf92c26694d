"""Whole config-2 frames on a grid beyond the caches for a range of Jacobi depths and tile shapes, one knob at a time around the
defaults.  usage: tune_big.py W H [frames]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fluidsolver_amd import FluidSolver, lib as L
from fluidsolver_amd.script import splat
W = int(sys.argv[1]); H = int(sys.argv[2]); n = int(sys.argv[3]) if len(sys.argv) > 3 else 20
lib = L.load()
s = FluidSolver(W, H); s.displayMode = None
(s.viscosity, s.diffusion, s.deltaTime, s.vorticityStrength, s.fadeSpeed, s.solverIterations) = (1.5e-4, 0.0, 0.5, 10.0, 0.002, 40)
f = 0
def frames(k):
    global f
    for _ in range(k):
        pos, force, col = splat(W, H, f); s.applyForce(pos, force, col); s.update(); f += 1
frames(12)
ms = C.c_float()
def frame_us():
    frames(4); s.sync()
    L.check(lib.fs_timer_begin(s.ctx), s.ctx)
    frames(n)
    L.check(lib.fs_timer_end(s.ctx, C.byref(ms)), s.ctx)
    return ms.value * 1e3 / n
TILES = {4: "4x16", 7: "5x16", 6: "6x16", 2: "8x16"}
base = {L.OPT_BLOCK_DEPTH_PRESSURE: 14, L.OPT_BLOCK_DEPTH_DIFFUSE: 8, L.OPT_TILE_VARIANT: 8, L.OPT_TILE_VARIANT_DIFFUSE: 8}
def run(label, **kw):
    for k, v in base.items():
        s.set_option(k, v)
    for k, v in kw.items():
        s.set_option(getattr(L, k), v)
    print("%-34s %9.1f us/frame" % (label, frame_us()), flush=True)
run("defaults")
for d in (10, 12, 14, 16, 20):
    run("pressure depth %d" % d, OPT_BLOCK_DEPTH_PRESSURE=d)
for tv in (4, 7, 6):
    run("pressure tile %s" % TILES[tv], OPT_TILE_VARIANT=tv)
for tv, d in ((6, 16), (6, 20), (7, 20)):
    run("pressure tile %s depth %d" % (TILES[tv], d), OPT_TILE_VARIANT=tv, OPT_BLOCK_DEPTH_PRESSURE=d)
for d in (6, 8, 10, 14):
    run("diffuse depth %d" % d, OPT_BLOCK_DEPTH_DIFFUSE=d)
for tv in (4, 7, 6, 2):
    run("diffuse tile %s" % TILES[tv], OPT_TILE_VARIANT_DIFFUSE=tv)
for tv, d in ((6, 10), (6, 14), (2, 14), (2, 10)):
    run("diffuse tile %s depth %d" % (TILES[tv], d), OPT_TILE_VARIANT_DIFFUSE=tv, OPT_BLOCK_DEPTH_DIFFUSE=d)
run("defaults again")
s.close()
