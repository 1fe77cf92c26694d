"""Config-2 frames (splat + update(), as bench.py runs them) replayed from the frame graph against the same launches
enqueued eagerly, A/B in one process, wall-clock timed like bench.py.  usage: ab_graph.py [W] [H]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fluidsolver_amd import FluidSolver, lib as L
from fluidsolver_amd.script import splat
W = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
H = int(sys.argv[2]) if len(sys.argv) > 2 else W
s = FluidSolver(W, H); s.displayMode = None
(s.viscosity, s.diffusion, s.deltaTime, s.vorticityStrength, s.fadeSpeed, s.solverIterations) = (1.5e-4, 0.0, 0.5, 10.0, 0.002, 40)
f = 0
def run(n):
    global f
    for _ in range(n):
        pos, force, col = splat(W, H, f); s.applyForce(pos, force, col); s.update(); f += 1
run(30)
for rep in range(4):
    for mode, name in ((1, "graph"), (0, "eager")):
        s.set_option(L.OPT_USE_GRAPH, mode)
        run(10); s.sync()
        t0 = time.perf_counter(); run(100); s.sync(); el = time.perf_counter() - t0
        if rep: print("%-6s %7.1f us/frame" % (name, el * 1e4), flush=True)
