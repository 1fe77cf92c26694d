"""curl + confinement at 3072^2: the two streaming kernels against the fused pass (FS_VORT_MARCH=0: its re-reading form)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fluidsolver_amd import FluidSolver, lib as L
W = H = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
lib = L.load()
s = FluidSolver(W, H); s.displayMode = None
vel = np.empty((2, H, W), np.float32)
lib.fs_harness_fill_uniform(vel.ctypes.data_as(C.c_void_p), vel.size, -1.0, 1.0, 4321)
s.write_field(L.FIELD_VELOCITY, vel)
g = L.full_grid(W, H)
pu, pv, pw, pitch = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_int()
L.check(lib.fs_field_device_ptr(s.ctx, L.FIELD_VELOCITY, 0, C.byref(pu), C.byref(pitch)))
L.check(lib.fs_field_device_ptr(s.ctx, L.FIELD_VELOCITY, 1, C.byref(pv), None))
L.check(lib.fs_field_device_ptr(s.ctx, L.FIELD_VORTICITY, 0, C.byref(pw), None))
po0, po1 = C.c_void_p(), C.c_void_p()
L.check(lib.fs_field_device_ptr(s.ctx, L.FIELD_DENSITY, 0, C.byref(po0), None))   # two dye planes as the out-of-place target
L.check(lib.fs_field_device_ptr(s.ctx, L.FIELD_DENSITY, 1, C.byref(po1), None))
st = lib.fs_stream(s.ctx)
ms = C.c_float()
def timed(fn, n=50):
    for _ in range(5): fn()
    s.sync()
    L.check(lib.fs_timer_begin(s.ctx), s.ctx)
    for _ in range(n): fn()
    L.check(lib.fs_timer_end(s.ctx, C.byref(ms)), s.ctx)
    return ms.value * 1e3 / n
def two():
    L.check(lib.fs_k_calculateVorticity(C.byref(g), pu, pv, pw, st))
    L.check(lib.fs_k_applyVorticity(C.byref(g), pw, pu, pv, C.c_float(1e-6), C.c_float(0.5), st))
def fused():
    L.check(lib.fs_k_vorticity_apply(C.byref(g), pu, pv, po0, po1, C.c_float(1e-6), C.c_float(0.5), st))
print("%dx%d two kernels %.1f us   fused (FS_VORT_MARCH=%s) %.1f us" % (W, H, timed(two), os.environ.get("FS_VORT_MARCH", "1"), timed(fused)))
