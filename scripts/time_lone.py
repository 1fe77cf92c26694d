"""One slab alone on the GPU (loopback communicator), wall time per frame without a profiler.
usage: time_lone.py [frames] [W] [rows_per_slab] [world] [rank] [KEY=VALUE options ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fluidsolver_amd import lib as L
from fluidsolver_amd.native_slabs import LoopbackSlab
from fluidsolver_amd.script import splat
a = [x for x in sys.argv[1:] if "=" not in x]
n = int(a[0]) if len(a) > 0 else 100
W = int(a[1]) if len(a) > 1 else 3072
rows = int(a[2]) if len(a) > 2 else 3072
K = int(a[3]) if len(a) > 3 else 2
rank = int(a[4]) if len(a) > 4 else 0
params = (1.5e-4, 0.0, 0.5, 10.0, 0.002, int(os.environ.get("FS_SWEEPS", "40")))
sl = LoopbackSlab(W, rows * K, params, device=0, rank=rank, world=K)
for kv in sys.argv[1:]:
    if "=" in kv:
        k, v = kv.split("=")
        sl.set_option(getattr(L, "OPT_" + k), int(v))
def step(f):
    pos, force, col = splat(W, rows * K, f, rank, K, 1.0)
    sl.add_force(pos, force, col)
    sl.update()
for f in range(20):
    step(f)
sl.sync()
for f in range(20, 32):
    step(f)
sl.host_times(); sl.host_breakdown()
# GPU time between two events on the slab's stream: the host may run a frame ahead of the GPU (no sync here: it would run
# the owed halo check early and shift the phase of the buffer roles, so that the first frames after it record new segments)
import ctypes as C
ms = C.c_float()
L.check(sl.lib.fs_timer_begin(sl.ctx), sl.ctx)
t0 = time.perf_counter()
for f in range(32, 32 + n):
    step(f)
te = time.perf_counter() - t0
L.check(sl.lib.fs_timer_end(sl.ctx, C.byref(ms)), sl.ctx)
t = ms.value * 1e-3
sl.sync()
ht, hb = sl.host_times(), sl.host_breakdown()
print("%s: %.1f us/frame (host loop %.1f; inside fs_step %.1f, waiting %.1f; %s) %s" % (
    " ".join(sys.argv[1:]), t / n * 1e6, te / n * 1e6, ht["step_us"], ht["wait_cfl_us"],
    ", ".join("%s %.0f" % (k, v / n) for k, v in hb.items() if v), sl.graph_stats), flush=True)
sl.close()
