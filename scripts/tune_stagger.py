"""Sweep the Jacobi stagger delay/mode at a given size; prints us per N-sweep loop (HIP events)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fluidsolver_amd import FluidSolver, lib as L

W = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
H = int(sys.argv[2]) if len(sys.argv) > 2 else W
N = int(sys.argv[3]) if len(sys.argv) > 3 else 40
lib = L.load()
s = FluidSolver(W, H)
s.displayMode = None
s.solverIterations = N
rng = np.random.default_rng(0)
s.write_field(L.FIELD_DIVERGENCE, rng.uniform(-1, 1, (1, H, W)).astype(np.float32))
s.write_field(L.FIELD_VELOCITY, rng.uniform(-1, 1, (2, H, W)).astype(np.float32))
ms = C.c_float()
VAR = int(sys.argv[4]) if len(sys.argv) > 4 else 4
DP = int(sys.argv[5]) if len(sys.argv) > 5 else 10
DD = int(sys.argv[6]) if len(sys.argv) > 6 else 8
L.check((lib.fs_config_set(L.OPT_TILE_VARIANT, VAR), lib.fs_set_option(s.ctx, L.OPT_TILE_VARIANT, VAR))[1])
L.check((lib.fs_config_set(L.OPT_BLOCK_DEPTH_PRESSURE, DP), lib.fs_set_option(s.ctx, L.OPT_BLOCK_DEPTH_PRESSURE, DP))[1]); L.check((lib.fs_config_set(L.OPT_BLOCK_DEPTH_DIFFUSE, DD), lib.fs_set_option(s.ctx, L.OPT_BLOCK_DEPTH_DIFFUSE, DD))[1])
print("variant", VAR, "depth pressure", DP, "diffuse", DD)
for mode in (0, 3):
    for ticks in (0, 200, 400, 600, 800, 1000):
        L.check((lib.fs_config_set(L.OPT_STAGGER_MODE, mode), lib.fs_set_option(s.ctx, L.OPT_STAGGER_MODE, mode))[1]); L.check((lib.fs_config_set(L.OPT_STAGGER, ticks), lib.fs_set_option(s.ctx, L.OPT_STAGGER, ticks))[1])
        row = []
        for stage in (L.STAGE_PRESSURE, L.STAGE_DIFFUSE_VELOCITY):
            s.stage(stage, N); s.sync()
            reps = 10
            L.check(lib.fs_timer_begin(s.ctx), s.ctx)
            for _ in range(reps):
                L.check(lib.fs_stage(s.ctx, stage, N), s.ctx)
            L.check(lib.fs_timer_end(s.ctx, C.byref(ms)), s.ctx)
            row.append(ms.value * 1e3 / reps)
        print("mode %d stagger %5d ticks  pressure x%d: %8.1f us   diffuse x%d: %8.1f us" % (mode, ticks, N, row[0], N, row[1]), flush=True)
