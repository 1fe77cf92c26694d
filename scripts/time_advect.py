"""Dye advection + fade (k_advect<3, true>) at 3072^2 on zero / slow / fast velocity against a pure streaming pass of
comparable traffic (fade of 3 planes in place + the curl pass): is the gather what costs, or the bytes?"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fluidsolver_amd import FluidSolver, lib as L
W = H = 3072
lib = L.load()
s = FluidSolver(W, H); s.displayMode = None
g = L.full_grid(W, H)
def ptr(field, plane):
    p = C.c_void_p(); L.check(lib.fs_field_device_ptr(s.ctx, field, plane, C.byref(p), None)); return p
dye = np.empty((3, H, W), np.float32); lib.fs_harness_fill_uniform(dye.ctypes.data_as(C.c_void_p), dye.size, 0.0, 1.0, 77)
s.write_field(L.FIELD_DENSITY, dye)
st = lib.fs_stream(s.ctx)
ms = C.c_float()
def timed(fn, n=40):
    for _ in range(5): fn()
    s.sync()
    L.check(lib.fs_timer_begin(s.ctx), s.ctx)
    for _ in range(n): fn()
    L.check(lib.fs_timer_end(s.ctx, C.byref(ms)), s.ctx)
    return ms.value * 1e3 / n
pu, pv, pd = ptr(L.FIELD_VELOCITY, 0), ptr(L.FIELD_VELOCITY, 1), ptr(L.FIELD_DENSITY, 0)
# out-of-place target: pressure / divergence / vorticity planes are contiguous? use the display planes (4 planes)
po = ptr(L.FIELD_DISPLAY, 0)
plane = C.c_size_t(L.pitch_for_width(W) * H)
for name, amp in (("zero velocity", 0.0), ("|v| dt < 0.5 cell", 1.0), ("|v| dt < 8 cells", 16.0), ("|v| dt < 64 cells", 128.0)):
    vel = np.empty((2, H, W), np.float32)
    lib.fs_harness_fill_uniform(vel.ctypes.data_as(C.c_void_p), vel.size, -amp, amp, 4321)
    if amp == 0.0: vel[:] = 0
    s.write_field(L.FIELD_VELOCITY, vel)
    t = timed(lambda: L.check(lib.fs_k_advect_fade(C.byref(g), pu, pv, pd, po, 3, plane, C.c_float(0.5), C.c_float(0.002), None, st)))
    print("dye advect + fade, %-20s %6.1f us" % (name, t), flush=True)
t = timed(lambda: L.check(lib.fs_k_fadeDensity(C.byref(g), pd, 3, plane, C.c_float(0.0), st)))
print("fade of 3 planes in place (226 MB)      %6.1f us" % t)
t = timed(lambda: L.check(lib.fs_k_advect(C.byref(g), pu, pv, pu, po, 2, plane, C.c_float(0.5), None, st)))
print("velocity advect (|v| dt < 64)           %6.1f us" % t)
