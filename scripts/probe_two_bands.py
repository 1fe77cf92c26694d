"""Upper bound of what row-band chains can gain: the pressure / diffuse loop of ONE 3072x3072 context against the same
loops of TWO (or four) independent contexts of 3072x1536 (3072x768) enqueued side by side on their own streams.
No halo rows between the bands here, so this is the ceiling, not the figure."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from fluidsolver_amd import lib as L
lib = L.load()
W = 3072
PAR = (1.5e-4, 0.0, 0.5, 10.0, 0.002, 40)

def run(K, stage, reps=30):
    H = 3072 // K
    ss = [bench.make_single(W, H, PAR, 0)[0] for _ in range(K)]
    for s in ss:
        d = np.empty((1, H, W), np.float32); lib.fs_harness_fill_uniform(d.ctypes.data_as(C.c_void_p), d.size, -1.0, 1.0, 1234)
        s.write_field(L.FIELD_DIVERGENCE, d)
        v = np.empty((2, H, W), np.float32); lib.fs_harness_fill_uniform(v.ctypes.data_as(C.c_void_p), v.size, -1.0, 1.0, 4321)
        s.write_field(L.FIELD_VELOCITY, v)
    for s in ss: s.stage(stage, 40)
    for s in ss: s.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        for s in ss: L.check(lib.fs_stage(s.ctx, stage, 40), s.ctx)
    for s in ss: s.sync()
    dt = (time.perf_counter() - t0) / reps * 1e6
    for s in ss: s.close()
    return dt

for name, stage in (("pressure", L.STAGE_PRESSURE_CLEARED), ("diffuse velocity", L.STAGE_DIFFUSE_VELOCITY)):
    for K in (1, 2, 4):
        print("%-17s %d band(s): %7.1f us per 40 sweeps" % (name, K, run(K, stage)), flush=True)
