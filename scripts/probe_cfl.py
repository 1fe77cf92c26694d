#!/usr/bin/env python3
"""max |v| dt of the scripted workloads over the first frames: sizes the advection halo
h = ceil(max|v_y| dt) + 2 that a row-slab run has to exchange per frame."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fluidsolver_amd import FluidSolver, lib as L
from fluidsolver_amd.script import splat

def run(W, H, n_ptr, frames, every):
    s = FluidSolver(W, H)
    (s.viscosity, s.diffusion, s.deltaTime, s.vorticityStrength, s.fadeSpeed, s.solverIterations) = (1.5e-4, 0.0, 0.5, 10.0, 0.002, 40)
    s.displayMode = None
    for f in range(frames):
        for p in range(n_ptr):
            pos, force, col = splat(W, H, f, p, n_ptr)
            s.applyForce(pos, force, col)
        s.update()
        if (f + 1) % every == 0:
            v = s.read_field(L.FIELD_VELOCITY)
            print("%dx%d ptr=%d frame %3d  max|u|dt=%.1f  max|v|dt=%.1f  force=%s" % (W, H, n_ptr, f + 1, np.abs(v[0]).max() * 0.5, np.abs(v[1]).max() * 0.5, np.round(force, 1)), flush=True)
    s.close()

run(3072, 3072, 1, 120, 20)
run(3072, 3072 * 2, 2, 60, 20)
run(16384, 16384, 8, 40, 10)
