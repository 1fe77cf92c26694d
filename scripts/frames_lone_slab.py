"""N frames of ONE slab alone on the GPU (loopback communicator, see fs_comm_create_loopback): the program for kernel traces
of a rank's frame.  usage: frames_lone_slab.py [frames] [W] [rows_per_slab] [world] [rank] [KEY=VALUE options ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fluidsolver_amd import lib as L
from fluidsolver_amd.native_slabs import LoopbackSlab
from fluidsolver_amd.script import splat
a = [x for x in sys.argv[1:] if "=" not in x]
n = int(a[0]) if len(a) > 0 else 30
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
for f in range(n):
    pos, force, col = splat(W, rows * K, f, rank, K, 1.0)
    sl.add_force(pos, force, col)
    sl.update()
sl.sync()
print(sl.graph_stats, sl.stats)
sl.close()
