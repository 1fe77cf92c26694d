"""N frames of K row slabs of 3072 rows each in one process on one GPU (async transport, overlapped plan): the program for
kernel traces of the slab path.  usage: frames_slabs.py [frames] [K]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fluidsolver_amd.native_slabs import LocalGroup
from fluidsolver_amd.script import splat
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
K = int(sys.argv[2]) if len(sys.argv) > 2 else 2
W, H = 3072, 3072 * K
g = LocalGroup(W, H, (1.5e-4, 0.0, 0.5, 10.0, 0.002, 40), K, device=0, asynchronous=True, overlap=True)
for f in range(n):
    for ptr in range(K):
        pos, force, col = splat(W, H, f, ptr, K)
        g.add_force(pos, force, col)
    g.update()
g.sync(); g.close()
