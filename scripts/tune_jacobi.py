"""Sweep Jacobi tile variants x block depths at a given size; prints us per 40-sweep loop (HIP events).
usage: python scripts/tune_jacobi.py [W] [H] [sweeps]"""
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
names = {0: "8x8", 1: "16x8", 2: "8x16", 3: "16x4", 4: "4x16", 5: "12x8", 6: "6x16", 7: "5x16"}
th = {0: 64, 1: 128, 2: 128, 3: 64, 4: 64, 5: 96, 6: 96, 7: 80}
ms = C.c_float()
res = []
VARS = [int(v) for v in sys.argv[4].split(',')] if len(sys.argv) > 4 else range(8)
for var in VARS:
    for depth in ([int(d) for d in os.environ["FS_DEPTHS"].split(',')] if os.environ.get("FS_DEPTHS") else (5, 7, 8, 10, 12, 14, 16, 20)):
        if th[var] - 2 * depth < 8:
            continue
        L.check((lib.fs_config_set(L.OPT_TILE_VARIANT, var), lib.fs_set_option(s.ctx, L.OPT_TILE_VARIANT, var))[1]); L.check((lib.fs_config_set(L.OPT_BLOCK_DEPTH, depth), lib.fs_set_option(s.ctx, L.OPT_BLOCK_DEPTH, depth))[1])
        row = [names[var], depth]
        for stage in (L.STAGE_PRESSURE, L.STAGE_DIFFUSE_VELOCITY):
            s.stage(stage, N); s.sync()
            reps = 5
            L.check(lib.fs_timer_begin(s.ctx), s.ctx)
            for _ in range(reps):
                L.check(lib.fs_stage(s.ctx, stage, N), s.ctx)
            L.check(lib.fs_timer_end(s.ctx, C.byref(ms)), s.ctx)
            row.append(ms.value * 1e3 / reps)
        res.append(row)
        print("tile %-5s depth %2d  pressure x%d: %8.1f us   diffuse(2 planes) x%d: %8.1f us" % (row[0], row[1], N, row[2], N, row[3]), flush=True)
best_p = min(res, key=lambda r: r[2]); best_d = min(res, key=lambda r: r[3])
print("best pressure", best_p, "best diffuse", best_d)
