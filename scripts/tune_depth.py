"""Temporal-blocking depth of the two Jacobi loops at a given grid size: 40 sweeps of each loop as update()
runs them, HIP-event timed, for a range of depths.  usage: tune_depth.py W H"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from fluidsolver_amd import FluidSolver, lib as L
W = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
H = int(sys.argv[2]) if len(sys.argv) > 2 else W
lib = L.load()
s = FluidSolver(W, H); s.displayMode = None
(s.viscosity, s.diffusion, s.deltaTime, s.vorticityStrength, s.fadeSpeed, s.solverIterations) = (1.5e-4, 0.0, 0.5, 10.0, 0.002, 40)
s.update()
buf = np.empty((2, H, W), np.float32)
lib.fs_harness_fill_uniform(buf.ctypes.data_as(C.c_void_p), buf.size, -1.0, 1.0, 1234)
s.write_field(L.FIELD_VELOCITY, buf)
s.write_field(L.FIELD_DIVERGENCE, buf[:1])
del buf
ms = C.c_float()
reps = 3 if W * H > (1 << 26) else 10
for opt, stage, name in ((L.OPT_BLOCK_DEPTH_PRESSURE, L.STAGE_PRESSURE_CLEARED, "pressure"),
                         (L.OPT_BLOCK_DEPTH_DIFFUSE, L.STAGE_DIFFUSE_VELOCITY, "diffuse(velocity)")):
    keep = lib.fs_config_get(opt)
    for depth in (5, 6, 7, 8, 10, 12, 14, 16, 20):
        (lib.fs_config_set(opt, depth), lib.fs_set_option(s.ctx, opt, depth))[1]
        for rep in range(2):
            L.check(lib.fs_stage(s.ctx, stage, 40), s.ctx); s.sync()
            L.check(lib.fs_timer_begin(s.ctx), s.ctx)
            for _ in range(reps):
                L.check(lib.fs_stage(s.ctx, stage, 40), s.ctx)
            L.check(lib.fs_timer_end(s.ctx, C.byref(ms)), s.ctx)
        print("%-18s depth %2d  %9.1f us per 40 sweeps" % (name, depth, ms.value * 1e3 / reps), flush=True)
    (lib.fs_config_set(opt, keep), lib.fs_set_option(s.ctx, opt, keep))[1]
