#!/usr/bin/env python3
"""K row slabs of one grid on ONE GPU (in-process groups) against the same grid as a single context:
the extra cost of the slab plan (ghost rows, rims, event waits, host enqueue) per frame, for the
blocking / asynchronous transports and with the halo traffic exposed / hidden.  The K slabs share the
GPU, so this is not a scaling measurement: (t_group - t_single) / t_single is the overhead the plan
adds when the wire is free."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fluidsolver_amd import FluidSolver
from fluidsolver_amd.native_slabs import LocalGroup
from fluidsolver_amd.script import splat

PARAMS = (1.5e-4, 0.0, 0.5, 10.0, 0.002, 40)


def timed(step, sync, warm, frames):
    for f in range(warm):
        step(f)
    sync()
    t0 = time.perf_counter()
    for f in range(warm, warm + frames):
        step(f)
    enq = (time.perf_counter() - t0) * 1e3 / frames     # host side only: the calls have returned
    sync()
    return (time.perf_counter() - t0) * 1e3 / frames, enq


def main():
    W, rows_per_slab = int(sys.argv[1]), int(sys.argv[2])
    frames = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    for K in ([int(k) for k in sys.argv[4].split(',')] if len(sys.argv) > 4 else (2, 4)):
        H = rows_per_slab * K
        s = FluidSolver(W, H)
        (s.viscosity, s.diffusion, s.deltaTime, s.vorticityStrength, s.fadeSpeed, s.solverIterations) = PARAMS
        s.displayMode = None

        def step1(f):
            for ptr in range(K):
                pos, force, col = splat(W, H, f, ptr, K)
                s.applyForce(pos, force, col)
            s.update()
        t1, e1 = timed(step1, s.sync, 10, frames)
        s.close()
        print("%dx%d single context            %8.3f ms/frame  host enqueue %6.1f us/frame" % (W, H, t1, e1 * 1e3), flush=True)
        for name, (asy, ovl) in (("blocking plain", (False, False)), ("async plain", (True, False)),
                                 ("blocking overlap", (False, True)), ("async overlap", (True, True))):
            g = LocalGroup(W, H, PARAMS, K, device=0, asynchronous=asy, overlap=ovl)
            if os.environ.get("FS_SPLIT") is not None:
                from fluidsolver_amd import lib as L
                for sl in g.slabs:
                    L.check(sl.lib.fs_set_option(sl.ctx, L.OPT_SPLIT_PLANES, int(os.environ["FS_SPLIT"])), sl.ctx)

            def stepk(f):
                for ptr in range(K):
                    pos, force, col = splat(W, H, f, ptr, K)
                    g.add_force(pos, force, col)
                g.update()
            g.slabs[0].host_times()
            tk, ek = timed(stepk, g.sync, 10, frames)
            ht = g.slabs[0].host_times()
            ek = (ht["step_us"] - ht["wait_cfl_us"]) / 1e3
            st = g.slabs[0].stats
            print("%dx%d %d slabs %-17s %8.3f ms/frame  (+%5.1f%%)  host enqueue %6.1f us/frame inside fs_group_step without the CFL waits (all %d slabs from one thread), halo %d rows, %.1f rounds/frame, %d re-plans"
                  % (W, H, K, name, tk, 100 * (tk - t1) / t1, ek * 1e3, K, g.slabs[0].last_halo, st["exchanges"] / st["frames"], st["redo"]),
                  flush=True)
            g.close()


main()
