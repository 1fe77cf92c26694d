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

PARAMS = (1.5e-4, 0.0, 0.5, 10.0, 0.002, int(os.environ.get("FS_SWEEPS", "40")))
MODES = (("blocking plain", (False, False)), ("async plain", (True, False)), ("blocking overlap", (False, True)),
         ("async overlap", (True, True)))
if os.environ.get("FS_MODES"):   # e.g. FS_MODES="async plain,async overlap"
    MODES = tuple(m for m in MODES if m[0] in os.environ["FS_MODES"].split(","))


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
        for name, (asy, ovl) in MODES:
            g = LocalGroup(W, H, PARAMS, K, device=0, asynchronous=asy, overlap=ovl)
            for key in ("FS_USE_GRAPH", "FS_SLAB_DEFER", "FS_SPLIT_ROWS", "FS_SLAB_SERIALIZE"):   # A/B hooks: FS_USE_GRAPH=0 etc.
                if os.environ.get(key) is not None:
                    from fluidsolver_amd import lib as L
                    for sl in g.slabs:
                        sl.set_option(getattr(L, "OPT_" + key[3:]), int(os.environ[key]))
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
            for f in range(10):
                stepk(f - 10)
            g.sync()
            for sl in g.slabs:
                sl.host_breakdown()
            tk, ek = timed(stepk, g.sync, 10, frames)
            hb = [sl.host_breakdown() for sl in g.slabs]
            ht = g.slabs[0].host_times()
            ek = (ht["step_us"] - ht["wait_cfl_us"]) / 1e3
            st = g.slabs[0].stats
            gs = g.slabs[0].graph_stats
            print("%dx%d %d slabs %-17s %8.3f ms/frame  (+%5.1f%%)  host: %6.1f us/frame inside fs_group_step for all %d slabs = %5.1f us per slab and frame, + %5.1f us waiting for the CFL value; halo %d rows, %.1f rounds/frame, %d re-plans; segments %d replayed / %d recorded / %d eager"
                  % (W, H, K, name, tk, 100 * (tk - t1) / t1, ek * 1e3, K, ek * 1e3 / K, ht["wait_cfl_us"], g.slabs[0].last_halo,
                     st["exchanges"] / st["frames"], st["redo"], gs["replays"], gs["captures"], gs["eager"]),
                  flush=True)
            print("      host us per frame, slab 0 (incl. its 10 warm-up frames): " + ", ".join("%s %.1f" % (k, v / (frames + 10)) for k, v in hb[0].items()), flush=True)
            g.close()
        lone(W, H, K, rows_per_slab, frames, t1)


def lone(W, H, K, rows_per_slab, frames, t1):
    """ONE slab of the K, alone on the GPU, stepped like a rank of an RCCL run (loopback communicator: transfers of the
    real sizes out of its own rows, so the fields mean nothing and only the timing counts) against the same number of
    rows as one single-GPU context: what a rank's frame costs on a GPU of its own when the wire is free."""
    from fluidsolver_amd import lib as L
    from fluidsolver_amd.native_slabs import LoopbackSlab
    s = FluidSolver(W, rows_per_slab)
    (s.viscosity, s.diffusion, s.deltaTime, s.vorticityStrength, s.fadeSpeed, s.solverIterations) = PARAMS
    s.displayMode = None

    def step1(f):
        pos, force, col = splat(W, rows_per_slab, f)
        s.applyForce(pos, force, col)
        s.update()
    ts, es = timed(step1, s.sync, 10, frames)
    s.close()
    print("%dx%d single context            %8.3f ms/frame  host enqueue %6.1f us/frame   (the rows of ONE slab as a grid of their own)"
          % (W, rows_per_slab, ts, es * 1e3), flush=True)
    rank = K // 2 if K > 2 else 0     # an inner slab (two neighbours) when there is one
    for name, opts in (("defaults", {}), ("one chain", {"SPLIT_ROWS": 0}), ("eager", {"USE_GRAPH": 0}),
                       ("round 2: eager, one chain, wait in fs_step", {"USE_GRAPH": 0, "SPLIT_ROWS": 0, "SLAB_DEFER": 0})):
        for ovl in (True, False):
            sl = LoopbackSlab(W, H, PARAMS, device=0, rank=rank, world=K, overlap=ovl)
            for k, v in opts.items():
                sl.set_option(getattr(L, "OPT_" + k), v)

            def stepk(f):
                # the splats of every pointer, like a rank of a real run; a weak one: the fields are not physical here
                for ptr in range(K):
                    pos, force, col = splat(W, H, f, ptr, K, 1.0)
                    sl.add_force(pos, force, col)
                sl.update()
            for f in range(10):
                stepk(f)
            sl.sync()
            # (a sync runs the owed halo check early, which shifts the phase of the buffer roles: the segments of the new
            # phase are recorded in these frames; the host is never more than a frame ahead of the GPU, so starting the
            # clock without another sync costs at most one frame in `frames`)
            for f in range(10, 20):
                stepk(f)
            sl.host_times(); sl.host_breakdown()
            t0 = time.perf_counter()
            for f in range(20, 20 + frames):
                stepk(f)
            ek = (time.perf_counter() - t0) * 1e3 / frames
            sl.sync()
            tk = (time.perf_counter() - t0) * 1e3 / frames
            ht, hb, gs = sl.host_times(), sl.host_breakdown(), sl.graph_stats
            print("%dx%d rank %d of %d ALONE (%s, %s)  %8.3f ms/frame  (+%5.1f%% over its rows as one context; x %d = +%5.1f%% over the whole grid)  halo %d rows, %d re-plans"
                  % (W, sl.r1 - sl.r0, rank, K, name, "overlap" if ovl else "plain", tk, 100 * (tk - ts) / ts, K, 100 * (K * tk - t1) / t1,
                     sl.last_halo, sl.stats["redo"]), flush=True)
            print("      host us per frame: inside fs_step %.1f (of it waiting for the CFL value %.1f): " % (ht["step_us"], ht["wait_cfl_us"])
                  + ", ".join("%s %.1f" % (k, v / frames) for k, v in hb.items()) + "; python loop incl. %d splats %.1f" % (K, ek * 1e3), flush=True)
            sl.close()


main()
