#!/usr/bin/env python3
"""Advection halo h = ceil(max|v_y| dt) + 2 per frame of the scripted workloads (2 slabs, in-process):
the series the halo speculation has to predict."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fluidsolver_amd.native_slabs import LocalGroup
from fluidsolver_amd.script import splat
PARAMS = (1.5e-4, 0.0, 0.5, 10.0, 0.002, 40)
for W, H, K, frames in ((3072, 6144, 2, 240), (16384, 4096, 2, 80), (16384, 16384, 8, 60)):
    g = LocalGroup(W, H, PARAMS, 2, device=0, asynchronous=True, overlap=False)
    hs = []
    for f in range(frames):
        for ptr in range(K):
            pos, force, col = splat(W, H, f, ptr, K)
            g.add_force(pos, force, col)
        g.update()
        hs.append(g.slabs[0].last_halo)
    g.close()
    print("%dx%d, %d pointers:" % (W, H, K), " ".join(map(str, hs)), flush=True)
