"""Diffuse loop (2 velocity planes x 40 sweeps) per tile variant and depth, warmed up, 20 reps each,
interleaved twice so clock ramp does not favour a variant.  usage: python scripts/tune_diffuse_tile.py [W] [H]"""
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
s.write_field(L.FIELD_VELOCITY, rng.uniform(-1, 1, (2, H, W)).astype(np.float32))
ms = C.c_float()
names = {4: "4x16", 7: "5x16", 6: "6x16"}
for rep in range(3):
    for var in (4, 7):
        for depth in (6, 7, 8, 9, 10, 12):
            L.check((lib.fs_config_set(L.OPT_TILE_VARIANT_DIFFUSE, var), lib.fs_set_option(s.ctx, L.OPT_TILE_VARIANT_DIFFUSE, var))[1]); L.check((lib.fs_config_set(L.OPT_BLOCK_DEPTH_DIFFUSE, depth), lib.fs_set_option(s.ctx, L.OPT_BLOCK_DEPTH_DIFFUSE, depth))[1])
            s.stage(L.STAGE_DIFFUSE_VELOCITY, 40); s.sync()
            L.check(lib.fs_timer_begin(s.ctx), s.ctx)
            for _ in range(20):
                L.check(lib.fs_stage(s.ctx, L.STAGE_DIFFUSE_VELOCITY, 40), s.ctx)
            L.check(lib.fs_timer_end(s.ctx, C.byref(ms)), s.ctx)
            if rep:
                print("rep %d tile %-5s depth %2d diffuse x40: %7.1f us" % (rep, names[var], depth, ms.value * 1e3 / 20), flush=True)
