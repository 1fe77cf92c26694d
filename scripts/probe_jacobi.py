"""Timing probes of the Jacobi block kernel: memory-only (no sweeps) and compute-only (no loads) variants."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fluidsolver_amd import FluidSolver, lib as L
W = H = 3072; N = 40
lib = L.load()
s = FluidSolver(W, H); s.displayMode = None; s.solverIterations = N
rng = np.random.default_rng(0)
s.write_field(L.FIELD_DIVERGENCE, rng.uniform(-1, 1, (1, H, W)).astype(np.float32))
s.write_field(L.FIELD_VELOCITY, rng.uniform(-1, 1, (2, H, W)).astype(np.float32))
ms = C.c_float()
def t(stage):
    s.stage(stage, N); s.sync()
    reps = 10
    L.check(lib.fs_timer_begin(s.ctx), s.ctx)
    for _ in range(reps):
        L.check(lib.fs_stage(s.ctx, stage, N), s.ctx)
    L.check(lib.fs_timer_end(s.ctx, C.byref(ms)), s.ctx)
    return ms.value * 1e3 / reps
for rep in range(2):
  for depth in (10,):
    L.check((lib.fs_config_set(L.OPT_BLOCK_DEPTH, depth), lib.fs_set_option(s.ctx, L.OPT_BLOCK_DEPTH, depth))[1])
    for name, mode in (("full", 0), ("no sweeps (memory only)", 16), ("no loads (compute+store)", 32), ("neither", 48)):
        L.check((lib.fs_config_set(L.OPT_STAGGER_MODE, mode), lib.fs_set_option(s.ctx, L.OPT_STAGGER_MODE, mode))[1]); L.check((lib.fs_config_set(L.OPT_STAGGER, 0), lib.fs_set_option(s.ctx, L.OPT_STAGGER, 0))[1])
        print("depth %2d %-28s pressure x%d: %7.1f us  diffuse x%d: %7.1f us" % (depth, name, N, t(L.STAGE_PRESSURE), N, t(L.STAGE_DIFFUSE_VELOCITY)), flush=True)
