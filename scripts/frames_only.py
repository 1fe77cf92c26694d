"""N frames of config 2 (scripted splats), nothing else: the program for kernel traces.  usage: frames_only.py [frames] [split_rows] [graph]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fluidsolver_amd import FluidSolver, lib as L
from fluidsolver_amd.script import splat
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
W = int(os.environ.get("FS_W", "3072")); H = int(os.environ.get("FS_H", str(W)))
s = FluidSolver(W, H); s.displayMode = None
(s.viscosity, s.diffusion, s.deltaTime, s.vorticityStrength, s.fadeSpeed, s.solverIterations) = (1.5e-4, 0.0, 0.5, 10.0, 0.002, 40)
if len(sys.argv) > 2: s.set_option(L.OPT_SPLIT_ROWS, int(sys.argv[2]))
if len(sys.argv) > 3: s.set_option(L.OPT_USE_GRAPH, int(sys.argv[3]))
for f in range(n):
    pos, force, col = splat(W, H, f); s.applyForce(pos, force, col); s.update()
s.sync(); s.close()
