"""Pressure loop (40 sweeps, cleared) + subtractGradient: two calls against the fused epilogue, HIP-event timed."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fluidsolver_amd import FluidSolver, lib as L
W = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
H = int(sys.argv[2]) if len(sys.argv) > 2 else W
lib = L.load()
s = FluidSolver(W, H); s.displayMode = None
rng = np.random.default_rng(0)
s.write_field(L.FIELD_DIVERGENCE, rng.uniform(-1, 1, (1, H, W)).astype(np.float32))
s.write_field(L.FIELD_VELOCITY, rng.uniform(-1, 1, (2, H, W)).astype(np.float32))
def dptr(field, plane=0):
    p, pitch = C.c_void_p(), C.c_int()
    L.check(lib.fs_field_device_ptr(s.ctx, field, plane, C.byref(p), C.byref(pitch)))
    return p
g = L.full_grid(W, H)
div, u, v, pa = dptr(L.FIELD_DIVERGENCE), dptr(L.FIELD_VELOCITY, 0), dptr(L.FIELD_VELOCITY, 1), dptr(L.FIELD_PRESSURE)
s2 = FluidSolver(W, H)   # only lends its pressure plane as the ping-pong partner
pbp, _pitch = C.c_void_p(), C.c_int()
L.check(lib.fs_field_device_ptr(s2.ctx, L.FIELD_PRESSURE, 0, C.byref(pbp), C.byref(_pitch)))
st = C.c_void_p(lib.fs_stream(s.ctx))
ib, ms = C.c_int(), C.c_float()
def two():
    L.check(lib.fs_k_pressureSolve_n(C.byref(g), div, pa, pbp, 40, 1, C.byref(ib), st))
    L.check(lib.fs_k_subtractGradient(C.byref(g), pbp if ib.value else pa, u, v, st))
def fused():
    L.check(lib.fs_k_pressureSolve_n_gradient(C.byref(g), div, pa, pbp, 40, 1, u, v, C.byref(ib), st))
def only():
    L.check(lib.fs_k_pressureSolve_n(C.byref(g), div, pa, pbp, 40, 1, C.byref(ib), st))
for rep in range(3):
    for name, fn in (("pressure x40 alone", only), ("pressure x40 + subtractGradient", two), ("pressure x40 with gradient epilogue", fused)):
        fn(); s.sync()
        L.check(lib.fs_timer_begin(s.ctx), s.ctx)
        for _ in range(20):
            fn()
        L.check(lib.fs_timer_end(s.ctx, C.byref(ms)), s.ctx)
        if rep: print("%-40s %7.1f us" % (name, ms.value * 1e3 / 20), flush=True)
