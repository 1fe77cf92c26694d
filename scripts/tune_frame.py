"""Whole config-2 frames (hipGraph replay) for a range of Jacobi depths / tile variants: tune_frame.py [W H]"""
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
f = 0
for _ in range(30):
    pos, force, col = splat(W, H, f); s.applyForce(pos, force, col); s.update(); f += 1
ms = C.c_float()
def frame_us(n=50):
    s.set_option(L.OPT_USE_GRAPH, 1)      # drops the recorded graphs: the knobs below are baked into them
    for _ in range(5): s.update()
    s.sync()
    L.check(lib.fs_timer_begin(s.ctx), s.ctx)
    for _ in range(n): s.update()
    L.check(lib.fs_timer_end(s.ctx, C.byref(ms)), s.ctx)
    return ms.value * 1e3 / n
print("default                      %7.1f us/frame" % frame_us(), flush=True)
for dd in (5, 6, 7, 8, 10, 14, 20):
    (lib.fs_config_set(L.OPT_BLOCK_DEPTH_DIFFUSE, dd), lib.fs_set_option(s.ctx, L.OPT_BLOCK_DEPTH_DIFFUSE, dd))[1]
    print("diffuse depth %2d             %7.1f us/frame" % (dd, frame_us()), flush=True)
(lib.fs_config_set(L.OPT_BLOCK_DEPTH_DIFFUSE, 8), lib.fs_set_option(s.ctx, L.OPT_BLOCK_DEPTH_DIFFUSE, 8))[1]
for dp in (5, 7, 8, 10, 14, 20):
    (lib.fs_config_set(L.OPT_BLOCK_DEPTH_PRESSURE, dp), lib.fs_set_option(s.ctx, L.OPT_BLOCK_DEPTH_PRESSURE, dp))[1]
    print("pressure depth %2d            %7.1f us/frame" % (dp, frame_us()), flush=True)
(lib.fs_config_set(L.OPT_BLOCK_DEPTH_PRESSURE, 10), lib.fs_set_option(s.ctx, L.OPT_BLOCK_DEPTH_PRESSURE, 10))[1]
for tv in (0, 2, 4, 6, 7):
    (lib.fs_config_set(L.OPT_TILE_VARIANT_DIFFUSE, tv), lib.fs_set_option(s.ctx, L.OPT_TILE_VARIANT_DIFFUSE, tv))[1]
    print("diffuse tile variant %d       %7.1f us/frame" % (tv, frame_us()), flush=True)
(lib.fs_config_set(L.OPT_TILE_VARIANT_DIFFUSE, 4), lib.fs_set_option(s.ctx, L.OPT_TILE_VARIANT_DIFFUSE, 4))[1]
for sp in (0, 1):
    s.set_option(L.OPT_SPLIT_PLANES, sp)
    print("split planes %d               %7.1f us/frame" % (sp, frame_us()), flush=True)
s.set_option(L.OPT_SPLIT_PLANES, 1)
for tv in (0, 4, 6, 7):
    s.set_option(L.OPT_TILE_VARIANT, tv); s.set_option(L.OPT_TILE_VARIANT_DIFFUSE, 4)
    print("pressure tile variant %d      %7.1f us/frame" % (tv, frame_us()), flush=True)
s.set_option(L.OPT_TILE_VARIANT, 4)
print("default again                %7.1f us/frame" % frame_us(), flush=True)
