"""Frames whose parameter VALUES change every frame (a slider being dragged): the instantiated frame graphs get new
kernel arguments in place (hipGraphExecUpdate) -- against eager launches and against steady replays.
usage: time_param_change.py [W H]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fluidsolver_amd import FluidSolver, lib as L
from fluidsolver_amd.script import splat
W = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
H = int(sys.argv[2]) if len(sys.argv) > 2 else W
lib = L.load()
s = FluidSolver(W, H); s.displayMode = None
(s.viscosity, s.diffusion, s.deltaTime, s.vorticityStrength, s.fadeSpeed, s.solverIterations) = (1.5e-4, 0.0, 0.5, 10.0, 0.002, 40)
f = 0
def frames(n, drag):
    global f
    t_host = 0.0
    s.sync(); t0 = time.perf_counter()
    for _ in range(n):
        if drag: s.vorticityStrength = 10.0 + 0.01 * (f % 50)
        pos, force, col = splat(W, H, f)
        h0 = time.perf_counter(); s.applyForce(pos, force, col); s.update(); t_host += time.perf_counter() - h0
        f += 1
    s.sync()
    return (time.perf_counter() - t0) / n * 1e6, t_host / n * 1e6
frames(20, False)
st = (C.c_long * 3)()
for name, graph, drag in (("graph replays, steady values", 1, False), ("graph, values change every frame", 1, True), ("eager, values change every frame", 0, True), ("eager, steady values", 0, False)):
    s.set_option(L.OPT_USE_GRAPH, graph)
    frames(12, False)     # settle: graphs recorded for every ping-pong state
    wall, host = frames(60, drag)
    L.check(lib.fs_graph_stats(s.ctx, st), s.ctx)
    print("%dx%d %-36s %7.1f us/frame  (host %6.1f us/frame)   graphs %d, refreshed in place %d, instantiated %d" % (W, H, name, wall, host, st[0], st[1], st[2]), flush=True)
