#!/usr/bin/env python3
"""bench.py -- throughput of the stable-fluids step on MI355X (driver contract: see README/DESIGN.md).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload config2|config3|config4|config5]

A "step" is one frame: one scripted splat + FluidSolver.update() (+ the tracer
update for config3).  N = 1 runs the workload on one GPU through the solver context (hipGraph);
N > 1 (launched by torch.distributed.run, one rank per GPU) runs row slabs through the library's
native slab runtime: halo rows by ncclSend/ncclRecv over xGMI.  Rank 0 prints ONE JSON line.

value = cells x frames/s / 1e9 (Gcell-updates/s of whole frames), inputs resident in HBM.
roofline = the pressure Jacobi kernel (k_jacobi_block<pressure>), effective bandwidth =
ALGORITHMIC bytes (12 B/cell/sweep) / launch time measured with HIP events.
cpu_baseline = the scalar CPU oracle (oracle/) timed on this host on a bounded sample.
"""
import argparse
import ctypes as C
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X nominal HBM3E peak (MI355X_MICROARCH.md: 8.0 TB/s spec)
ALG_BYTES_PRESSURE_SWEEP = 12   # read div 4 + read p 4 + write p 4
# full frame, config 2: 12 + 20 + 40*16 + 12 + 4 + 40*12 + 20 + 16 + 32 + 24
ALG_BYTES_FRAME_CONFIG2 = 1260

WORKLOADS = {
    # name: (W, H per GPU or total, params(visc, diff, dt, vort, fade, N), particles, scaling)
    "config2": dict(W=3072, H=3072, params=(1.5e-4, 0.0, 0.5, 10.0, 0.002, 40), particles=0, scaling="weak",
                    desc="3072x3072 full step: vorticity + diffuse + 40 pressure iters + RGB dye advect"),
    "config3": dict(W=3072, H=3072, params=(1.5e-4, 0.0, 0.5, 10.0, 0.002, 40), particles=1 << 20, scaling="weak",
                    desc="config2 + 1,048,576 tracer particles"),
    "config4": dict(W=16384, H=16384, params=(1.5e-4, 0.0, 0.5, 10.0, 0.002, 40), particles=0, scaling="strong", pointers=8,
                    desc="16384x16384 full step, row slabs, strong scaling"),
    "config5": dict(W=8192, H=8192, params=(1.5e-4, 0.0, 0.5, 10.0, 0.002, 100), particles=0, scaling="weak",
                    desc="8192x8192 per GPU, 100 pressure iters, weak scaling"),
}


# Watchdog: a multi-rank run that stops making progress (a halo exchange that never completes, a rank that died)
# must still end with ONE JSON line that says where it stopped, not with a silent hang.  The main thread names the
# phase it is in; if the whole run outlives FS_BENCH_WATCHDOG_S seconds (default 900; 0 = off) every rank leaves
# through os._exit and rank 0 prints what it has: the finished main line if only the untimed extra hung, else an
# error line (no number is made up).
STATE = {"phase": "start", "out": None}


def phase(name):
    STATE["phase"] = name


def start_watchdog():
    limit = float(os.environ.get("FS_BENCH_WATCHDOG_S", "900"))
    if limit <= 0:
        return

    def bark():
        time.sleep(limit)
        rank0 = int(os.environ.get("RANK", "0")) == 0
        msg = "watchdog: no result after %.0f s, stopped in phase '%s'" % (limit, STATE["phase"])
        sys.stderr.write("[bench rank %s] %s\n" % (os.environ.get("RANK", "0"), msg))
        sys.stderr.flush()
        if rank0:
            out = STATE["out"]
            if out is not None:
                out = dict(out, config4_strong={"error": msg})
            else:
                out = {"metric": "Gcell-updates/s, full stable-fluids step", "value": None, "unit": "Gcell-updates/s",
                       "n_gpus": int(os.environ.get("WORLD_SIZE", "1")), "error": msg}
            print(json.dumps(out), flush=True)
        else:
            time.sleep(5.0)   # the launcher ends every rank as soon as one has left: rank 0 prints first
        os._exit(3)
    threading.Thread(target=bark, daemon=True).start()


def ctx_option(s, key):
    """An option as the context of `s` (FluidSolver or slab) runs it -- not the process default."""
    from fluidsolver_amd import lib as L
    v = C.c_int()
    L.check(L.load().fs_get_option(s.ctx, key, C.byref(v)), s.ctx)
    return v.value


def make_single(W, H, params, particles):
    from fluidsolver_amd import FluidSolver, ParticleSystem, lib as L
    s = FluidSolver(W, H, max_particles=particles)
    (s.viscosity, s.diffusion, s.deltaTime, s.vorticityStrength, s.fadeSpeed, s.solverIterations) = params
    s.displayMode = None   # headless: the render pass is a "next" row, not part of the sim
    ps = None
    if particles:
        lib = L.load()
        ps = ParticleSystem(s)

        def fill(lo, hi, seed):
            a = np.empty(particles, np.float32)
            lib.fs_harness_fill_uniform(a.ctypes.data_as(C.c_void_p), particles, lo, hi, seed)
            return a
        z = np.zeros(particles, np.float32)
        # xorshift32 seed 0x9E3779B9, positions uniform over the grid,
        # mass ~ U(0.1,1), life ~ U(0.3,1), age = 0
        ps.set_arrays(fill(0, W, 0x9E3779B9), fill(0, H, 0x9E3779B9 ^ 1), z, z, fill(0.3, 1.0, 0x9E3779B9 ^ 2),
                      fill(0.1, 1.0, 0x9E3779B9 ^ 3), z)
        ps.stepWithSolver((W, H))   # the tracer update is the last node of the frame's graph
    return s, ps


def frame_single(s, ps, f, W, H, n_pointers=1):
    from fluidsolver_amd.script import splat
    for ptr in range(n_pointers):
        pos, force, col = splat(W, H, f, ptr, n_pointers)
        s.applyForce(pos, force, col)
    s.update()   # with tracers: FluidSolver.update() + ParticleSystem.update() (MetalView.swift:152-155) in one graph


def jacobi_microbench(s, W, H, sweeps=40, reps=25):
    """SURVEY 8(d) Jacobi micro-benchmark: the pressure loop alone, exactly as update() runs it (p = 0 folded
    into the first launch, Solver:528-541), on div ~ U(-1,1) (xorshift32 seed 1234), HIP-event timed on the
    context's stream.  Returns (ms per launch, sweeps per launch, ms per sweep)."""
    from fluidsolver_amd import lib as L
    lib = L.load()
    div = np.empty((1, H, W), np.float32)
    lib.fs_harness_fill_uniform(div.ctypes.data_as(C.c_void_p), div.size, -1.0, 1.0, 1234)
    s.write_field(L.FIELD_DIVERGENCE, div)
    s.write_field(L.FIELD_PRESSURE, np.zeros((1, H, W), np.float32))
    depth = ctx_option(s, L.OPT_BLOCK_DEPTH_PRESSURE)   # (fs_stage runs the context's tuning)
    s.stage(L.STAGE_PRESSURE_CLEARED, sweeps)   # warm-up
    s.sync()
    ms = C.c_float()
    L.check(lib.fs_timer_begin(s.ctx), s.ctx)
    for _ in range(reps):
        L.check(lib.fs_stage(s.ctx, L.STAGE_PRESSURE_CLEARED, sweeps), s.ctx)
    L.check(lib.fs_timer_end(s.ctx, C.byref(ms)), s.ctx)
    launches = reps * ((sweeps + depth - 1) // depth)
    return ms.value / launches, sweeps * reps / float(launches), ms.value / (reps * sweeps)


def diffuse_microbench(s, W, H, sweeps=40, reps=10):
    """The same for `diffuse` on the velocity: 40 sweeps of both planes as update() runs them
    (Solver:506-517), velocity ~ U(-1,1).  Returns (ms per launch, sweeps per launch, ms per sweep)."""
    from fluidsolver_amd import lib as L
    lib = L.load()
    vel = np.empty((2, H, W), np.float32)
    lib.fs_harness_fill_uniform(vel.ctypes.data_as(C.c_void_p), vel.size, -1.0, 1.0, 4321)
    s.write_field(L.FIELD_VELOCITY, vel)
    depth = ctx_option(s, L.OPT_BLOCK_DEPTH_DIFFUSE)
    s.stage(L.STAGE_DIFFUSE_VELOCITY, sweeps)   # warm-up
    s.sync()
    ms = C.c_float()
    L.check(lib.fs_timer_begin(s.ctx), s.ctx)
    for _ in range(reps):
        L.check(lib.fs_stage(s.ctx, L.STAGE_DIFFUSE_VELOCITY, sweeps), s.ctx)
    L.check(lib.fs_timer_end(s.ctx, C.byref(ms)), s.ctx)
    launches = reps * ((sweeps + depth - 1) // depth)
    return ms.value / launches, sweeps * reps / float(launches), ms.value / (reps * sweeps)


def project_microbench(s, W, H, sweeps=40, reps=20):
    """The projection of update() alone (Solver:522-547: divergence, clear, `sweeps` pressure sweeps, subtractGradient) the
    way a frame runs it -- one loop of launches, as two row bands when FS_OPT_SPLIT_ROWS applies -- and as one chain,
    on velocity ~ U(-1,1), HIP-event timed.  Returns {"us", "us_one_chain"} per projection."""
    from fluidsolver_amd import lib as L
    lib = L.load()
    vel = np.empty((2, H, W), np.float32)
    lib.fs_harness_fill_uniform(vel.ctypes.data_as(C.c_void_p), vel.size, -1.0, 1.0, 4321)
    out = {}
    ms = C.c_float()
    for key, mode in (("us_one_chain", 0), ("us", 1)):
        s.set_option(L.OPT_SPLIT_ROWS, mode)
        s.write_field(L.FIELD_VELOCITY, vel)
        s.stage(L.STAGE_PROJECT, sweeps)
        s.sync()
        L.check(lib.fs_timer_begin(s.ctx), s.ctx)
        for _ in range(reps):
            L.check(lib.fs_stage(s.ctx, L.STAGE_PROJECT, sweeps), s.ctx)
        L.check(lib.fs_timer_end(s.ctx, C.byref(ms)), s.ctx)
        out[key] = ms.value * 1e3 / reps
    return out


def cpu_baseline(W, H, params, budget_s=20.0, omp=False):
    """The scalar oracle ("port", 1 thread; omp=True: the same code with OpenMP over rows on the box's CPU share,
    bit-identical) on a bounded sample of the same workload: whole frames of the same scripted input, as many as
    fit the budget (at least 1)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_binding import Oracle, Params
    from fluidsolver_amd.script import splat
    o = Oracle(omp=omp)
    s = o.solver(W, H)
    p = Params(*params)
    frames, t0 = 0, time.perf_counter()
    while True:
        pos, force, col = splat(W, H, frames)
        s.add_force(pos, force, col, 5.0, p.deltaTime)
        s.update(p)
        frames += 1
        el = time.perf_counter() - t0
        if el > budget_s or el * (frames + 1) / frames > budget_s * 1.5:
            break
    fps = frames / el
    threads = o.threads() if omp else 1
    return dict(value=W * H * fps / 1e9, unit="Gcell-updates/s", cores=threads, kind="port",
                sample="%d whole frame(s) of the same %dx%d scripted workload, scalar C oracle (gcc -O2 "
                       "-ffp-contract=off%s), %.1f s" % (frames, W, H, ", OpenMP over rows" if omp else "", el),
                frames_per_s=fps, host_cpus=os.cpu_count())


def slab_selfcheck(dist, rank, world, local_rank):
    """N > 1 only: a small grid stepped as `world` row slabs over RCCL must equal the same grid stepped on
    one GPU, bit for bit (velocity, dye, pressure).  Untimed; the verdict goes into the JSON line."""
    import torch
    from fluidsolver_amd import FluidSolver, lib as L
    from fluidsolver_amd.native_slabs import RcclSlab
    from fluidsolver_amd.script import splat
    W, H, frames = 640, 96 * world, 4
    params = (1.5e-4, 0.00001, 0.5, 10.0, 0.002, 12)
    s = RcclSlab(W, H, params, device=local_rank, rank=rank, world=world, dist=dist)
    for f in range(frames):
        s.frame(5 * f + 1, n_pointers=world, forceMultiplier=2.0)
    s.sync()
    mine = np.concatenate([s.owned("vel"), s.owned("dye"), s.owned("p")], axis=0)   # [6, H/world, W]
    halo = s.last_halo
    s.close()
    t = torch.from_numpy(np.ascontiguousarray(mine))
    parts = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
    dist.gather(t, parts, dst=0)
    verdict = [""]
    if rank == 0:
        got = np.concatenate([p.numpy() for p in parts], axis=1)
        ref = FluidSolver(W, H, device=local_rank)
        (ref.viscosity, ref.diffusion, ref.deltaTime, ref.vorticityStrength, ref.fadeSpeed, ref.solverIterations) = params
        ref.displayMode = None
        for f in range(frames):
            for ptr in range(world):
                pos, force, col = splat(W, H, 5 * f + 1, ptr, world, 2.0)
                ref.applyForce(pos, force, col)
            ref.update()
        ref.sync()
        exp = np.concatenate([ref.read_field(L.FIELD_VELOCITY), ref.read_field(L.FIELD_DENSITY),
                              ref.read_field(L.FIELD_PRESSURE)], axis=0)
        ref.close()
        same = np.array_equal(got, exp) and bool(np.abs(exp[:2]).max() > 0)
        verdict[0] = ("bit-identical to 1 GPU" if same else "MISMATCH against 1 GPU (%d cells differ)"
                      % int((got != exp).sum())) + " on %dx%d, %d frames, advection halo %d rows" % (W, H, frames, halo)
    dist.broadcast_object_list(verdict, src=0)
    return verdict[0]


def config4_extra(world, rank, local_rank, dist, frames=10, warm=4):
    """Untimed-by-the-driver extra, reported beside the main line at every N: the 16384^2 grid of config 4
    (strong scaling: the SAME grid on 1 GPU or split into N row slabs, always eight pointers), a few frames.
    t_1 / (N t_N) of these figures across the driver's N = 1, 2, 4, 8 runs is the strong-scaling efficiency
    north_star asks for."""
    import torch
    wl = WORKLOADS["config4"]
    W, H, params = wl["W"], wl["H"], wl["params"]
    if world == 1:
        s, _ = make_single(W, H, params, 0)
        step = lambda f: frame_single(s, None, f, W, H, 8)   # noqa: E731
        sync = lambda: (s.sync(), torch.cuda.synchronize())   # noqa: E731
    else:
        from fluidsolver_amd.native_slabs import RcclSlab
        s = RcclSlab(W, H, params, device=local_rank, rank=rank, world=world, dist=dist)
        step = lambda f: s.frame(f, n_pointers=8)   # noqa: E731
        sync = lambda: (s.sync(), torch.cuda.synchronize(), dist.barrier())   # noqa: E731
    def timed(f0):
        for f in range(f0, f0 + warm):
            step(f)
        sync()
        t0 = time.perf_counter()
        for f in range(f0 + warm, f0 + warm + frames):
            step(f)
        sync()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    out = {"grid": [W, H], "frames": frames, "pointers": 8}
    if world == 1:
        el = timed(0)
        # the two Jacobi loops alone at this size (what roofline.kernels.*_16384 is computed from)
        pl, pspl, psw = jacobi_microbench(s, W, H, reps=3)
        dl, dspl, dsw = diffuse_microbench(s, W, H, reps=3)
        out["jacobi"] = {"pressure_us_per_launch": pl * 1e3, "pressure_sweeps_per_launch": pspl,
                         "diffuse_us_per_launch": dl * 1e3, "diffuse_sweeps_per_launch": dspl,
                         "pressure_gcell_updates_per_s_per_sweep": W * H / (psw * 1e-3) / 1e9,
                         "diffuse_gcell_updates_per_s_per_sweep": W * H / (dsw * 1e-3) / 1e9}
    else:
        # both slab plans, with every wait for halo rows timed (the event records cost ~1 us per join)
        s.set_profile(True)
        per_plan = {}
        for k, mode in enumerate((0, 1)):
            s.set_overlap(mode)
            s.waits()
            e = timed(k * (warm + frames))
            w = s.waits()
            wt = torch.tensor([w["vel_ms"], w["dye_ms"], w["blocking_ms"]], dtype=torch.float64)
            dist.all_reduce(wt, op=dist.ReduceOp.MAX)
            n = float(warm + frames)
            per_plan["overlap" if mode else "plain"] = {
                "ms_per_frame": e * 1e3 / frames,
                "exposed_wait_us_per_frame": {"velocity_halo": float(wt[0]) * 1e3 / n, "dye_halo": float(wt[1]) * 1e3 / n,
                                              "blocking_exchanges": float(wt[2]) * 1e3 / n}}
        el = min(v["ms_per_frame"] for v in per_plan.values()) * frames / 1e3
        st = s.stats
        out.update(plans=per_plan, last_advect_halo=s.last_halo, advection_replans=st["redo"],
                   halo_rows_sent_per_frame=st["rows_sent"] / float(max(1, st["frames"])))
    s.close()
    out.update(ms_per_frame=el * 1e3 / frames, frames_per_s=frames / el,
               gcell_updates_per_s=W * H * frames / el / 1e9)
    return out


ROOFLINE_INPUTS = os.path.join("profiles", "roofline_inputs.json")
VALU_PEAK_LANE_OPS = 256 * 4 * 32 * 2.4e9   # 256 CUs x 4 SIMD-32 x 2.4 GHz (MI355X_MICROARCH.md); measured 67-73 T at the
                                            # 2.31 GHz the chip holds under load: profiles/r02_valu_issue.log


def load_roofline_inputs():
    """PMC traffic + SQ counters of the two Jacobi kernels (scripts/profile_jacobi.sh + scripts/roofline_inputs.py).
    They describe the kernels of ONE source tree: used only if its sha equals this tree's, else (None, reason)."""
    from fluidsolver_amd.build import source_sha
    try:
        with open(os.path.join(ROOT, ROOFLINE_INPUTS)) as f:
            ri = json.load(f)
    except Exception as e:
        return None, "%s unreadable: %s" % (ROOFLINE_INPUTS, e)
    sha = source_sha()
    if ri.get("source_sha") != sha:
        return None, "%s was measured on kernel sources %s, this tree is %s: PMC-derived figures withheld" % (
            ROOFLINE_INPUTS, ri.get("source_sha"), sha)
    return ri, None


def physical_roofline(ri, key, us_per_launch):
    """What the launch really does to the memory system and the VALUs: PMC bytes / live launch time / 8 TB/s, and the
    VALU pipe occupancy (busy quad-cycles x 4 / SIMD-cycles of the launch, 2.4 GHz nominal)."""
    e = ri["kernels"].get(key)
    if not e or not us_per_launch:
        return None
    t = us_per_launch * 1e-6
    in_cache = key.endswith("_3072")
    return {"kernel": e["kernel"], "us_per_launch_live": us_per_launch, "us_per_launch_traced": e["traced_ns"] / 1e3,
            "bytes_are": ("L2 <-> fabric bytes: at this size the loop's planes live in the 256 MB Infinity Cache, so hbm_frac here is a "
                          "FABRIC fraction of the 8 TB/s figure, not DRAM traffic" if in_cache else
                          "L2 <-> fabric bytes = DRAM bytes at this size (the planes are 1 GiB each)"),
            "hbm_bytes_per_launch_pmc": e["hbm_bytes"], "hbm_GBps": e["hbm_bytes"] / t / 1e9,
            "hbm_frac": e["hbm_bytes"] / t / (HBM_PEAK_GBS * 1e9),
            "valu_busy_frac": e["sq"]["SQ_ACTIVE_INST_VALU"] * 4.0 / (256 * 4 * t * 2.4e9),
            "valu_insts_per_launch": e["sq"]["SQ_INSTS_VALU"], "valu_cycles_per_inst": e.get("valu_cycles_per_inst")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="config2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--depth", type=int, default=0, help="Jacobi temporal-blocking depth (0 = library default)")
    args = ap.parse_args()

    start_watchdog()
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("FS_BENCH_DEVICE"):   # test hook: rehearse N ranks on one GPU (with tests/mock_rccl)
        local_rank = int(os.environ["FS_BENCH_DEVICE"])
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs one rank per GPU: launch with python -m torch.distributed.run "
                             "--nproc-per-node %d bench.py --gpus %d ..." % (args.gpus, args.gpus, args.gpus))
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback)"
    torch.cuda.set_device(local_rank)

    from fluidsolver_amd import lib as L
    lib = L.load()
    if args.depth:
        L.check(lib.fs_config_set(L.OPT_BLOCK_DEPTH, args.depth))
    wl = WORKLOADS[args.workload]
    W, Hg = wl["W"], wl["H"]
    params = wl["params"]

    # FS_BENCH_FORCE_SLABS=1 (test hook): take the multi-rank code path with a single rank, so that the
    # control plane, the slab self-check and the JSON of an N > 1 run can be rehearsed on a 1-GPU box.
    slabs = world > 1 or os.environ.get("FS_BENCH_FORCE_SLABS") == "1"
    if not slabs:
        H = Hg
        s, ps = make_single(W, H, params, wl["particles"])
        n_ptr = wl.get("pointers", 1)
        step = lambda f: frame_single(s, ps, f, W, H, n_ptr)   # noqa: E731
        barrier = lambda: None                          # noqa: E731
        sync_all = lambda: (s.sync(), torch.cuda.synchronize())   # noqa: E731
        total_cells = W * H
        parallelism = "1 GPU"
    else:
        import torch.distributed as dist
        from fluidsolver_amd.native_slabs import RcclSlab
        # Control plane (RCCL unique id, barriers, max of the timings): torch.distributed over gloo.
        # Data plane (halo rows, CFL all-reduce): ncclSend/ncclRecv/ncclAllReduce inside the library.
        # (gloo announces its connections on fd 1; stdout is reserved for the one JSON line, so it goes to stderr)
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        phase("gloo rendezvous")
        try:
            dist.init_process_group("gloo")
        finally:
            os.dup2(saved_fd, 1)
            os.close(saved_fd)
        H = Hg * world if wl["scaling"] == "weak" else Hg
        phase("RCCL communicator + slab context")
        s = RcclSlab(W, H, params, device=local_rank, rank=rank, world=world, dist=dist)
        phase("RCCL self-test (grouped send/recv + all-reduce)")
        L.check(lib.fs_comm_selftest(s.comm, 1 << 16))
        phase("slab self-check (small grid as N slabs against 1 GPU)")
        slab_check = slab_selfcheck(dist, rank, world, local_rank)
        n_ptr = wl.get("pointers", world)   # weak scaling: one pointer per slab; config 4: always eight
        step = lambda f: s.frame(f, n_pointers=n_ptr)   # noqa: E731
        barrier = dist.barrier
        sync_all = lambda: (s.sync(), torch.cuda.synchronize())   # noqa: E731
        total_cells = W * H
        parallelism = "%d row slabs (RCCL send/recv halo exchange over xGMI)" % world
        # Untimed A/B of the two frame plans on this machine's links (FS_OPT_SLAB_OVERLAP: halo traffic
        # hidden behind the kernels / exchange-wait-compute); the timed run uses the faster one.  Same
        # results either way (bit-identical fields), every rank takes the same decision (max over ranks).
        forced = os.environ.get("FS_BENCH_SLAB_OVERLAP")
        ab, f0 = {}, 0
        for mode in ((0, 1) if forced is None else (int(forced),)):   # plain first: the halo history is warm for the guess
            phase("untimed A/B of the slab plans, %s" % ("overlapped" if mode else "plain"))
            s.set_overlap(mode)
            for f in range(f0, f0 + 4):
                step(f)
            sync_all(); barrier()
            t0 = time.perf_counter()
            for f in range(f0 + 4, f0 + 12):
                step(f)
            sync_all(); barrier()
            t = torch.tensor([(time.perf_counter() - t0) / 8.0], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ab[mode] = float(t.item()) * 1e3
            f0 += 12
        overlap = min(ab, key=ab.get)
        s.set_overlap(overlap)
        # untimed settle-in: the exchange-free runs of launches of a slab's frame are recorded as hipGraphs once the options have
        # stayed put for two frames, one recording per combination of buffer roles (up to twelve): all done before the clock starts
        phase("settle-in frames (segment graphs are recorded)")
        for f in range(f0, f0 + 16):
            step(f)
        f0 += 16

    f0 = f0 if slabs else 0
    phase("warm-up + timed frames")
    for f in range(f0, f0 + args.warmup):
        step(f)
    sync_all()
    barrier()
    sync_all()
    if slabs:
        s.host_times()     # reset the library's host-time accumulators
        s.host_breakdown()
    t0 = time.perf_counter()
    for f in range(f0 + args.warmup, f0 + args.warmup + args.steps):
        step(f)
    t_enq = time.perf_counter() - t0     # host side only: the calls have returned, the GPU may still be running
    sync_all()
    barrier()
    sync_all()
    elapsed = time.perf_counter() - t0
    host_t = s.host_times() if slabs else None
    if slabs:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ms_per_step = elapsed * 1e3 / args.steps
    fps = args.steps / elapsed
    out = {
        "metric": "Gcell-updates/s, full stable-fluids step (cells x frames/s); frames/s and Jacobi "
                  "Gcell-updates/s per sweep alongside",
        "value": total_cells * fps / 1e9,
        "unit": "Gcell-updates/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": wl["scaling"],
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic (scripted Lissajous splats, SURVEY 8(d); zero initial state)",
        "config": {"workload": "%s: %s" % (args.workload, wl["desc"]), "grid": [W, H],
                   "solverIterations": params[5], "particles": wl["particles"], "parallelism": parallelism,
                   "jacobi_block_depth": {"pressure": ctx_option(s, L.OPT_BLOCK_DEPTH_PRESSURE),
                                          "diffuse": ctx_option(s, L.OPT_BLOCK_DEPTH_DIFFUSE)}},
        "frames_per_s": fps,
        "host_enqueue_us_per_frame": t_enq * 1e6 / args.steps,
    }

    phase("micro-benchmarks / CPU baseline / exposed waits")
    if not slabs:
        alg = ALG_BYTES_FRAME_CONFIG2 if params[5] == 40 else None
        if alg:
            # algorithmic bytes per frame (per-sweep bytes x sweeps: the temporally blocked loops move them once per launch,
            # so this is an EFFECTIVE rate and exceeds the peak) ...
            out["frame_effective_GBps"] = alg * total_cells * fps / 1e9
            out["frame_effective_over_hbm_peak"] = out["frame_effective_GBps"] / HBM_PEAK_GBS
            # ... and the physical one: PMC bytes of one frame (profiles/frame_traffic.json, scripts/frame_traffic.py) over the
            # frame time measured here, withheld when any kernel source has changed since the PMC passes
            try:
                from fluidsolver_amd.build import source_sha_all
                ft = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "frame_traffic.json")))
                if ft.get("source_sha_all") == source_sha_all() and (W, H) == (3072, 3072) and not wl["particles"]:
                    out["frame_traffic_bytes_pmc"] = ft["bytes_per_frame"]
                    out["frame_hbm_frac"] = ft["bytes_per_frame"] / (ms_per_step * 1e-3) / (HBM_PEAK_GBS * 1e9)
                else:
                    out["frame_traffic_withheld"] = "profiles/frame_traffic.json was measured on other sources or another workload"
            except (OSError, ValueError, KeyError) as e:
                out["frame_traffic_withheld"] = "%s: %s" % (type(e).__name__, e)
        if params[5] >= 1:
            # untimed A/B of the two-band projection where it matters, in the frame's graph (interleaved, 2 x 30 frames each)
            ab, fa = {0: [], 1: []}, args.warmup + args.steps
            for rep in range(2):
                for mode in (0, 1):
                    s.set_option(L.OPT_SPLIT_ROWS, mode)
                    for f in range(fa, fa + 4):
                        step(f)
                    s.sync()
                    t0 = time.perf_counter()
                    for f in range(fa + 4, fa + 34):
                        step(f)
                    s.sync()
                    ab[mode].append((time.perf_counter() - t0) / 30.0 * 1e6)
                    fa += 34
            s.set_option(L.OPT_SPLIT_ROWS, 1)
        ms_launch, sweeps_per_launch, ms_sweep = jacobi_microbench(s, W, H)
        achieved = ALG_BYTES_PRESSURE_SWEEP * total_cells * sweeps_per_launch / (ms_launch * 1e-3) / 1e9
        out["jacobi"] = {"gcell_updates_per_s_per_sweep": total_cells / (ms_sweep * 1e-3) / 1e9,
                         "us_per_sweep": ms_sweep * 1e3, "us_per_launch": ms_launch * 1e3,
                         "sweeps_per_launch": sweeps_per_launch}
        dl, dspl, dsw = diffuse_microbench(s, W, H)
        out["jacobi_diffuse_velocity"] = {"gcell_updates_per_s_per_sweep": total_cells / (dsw * 1e-3) / 1e9,
                                          "us_per_sweep": dsw * 1e3, "us_per_launch": dl * 1e3, "sweeps_per_launch": dspl,
                                          "effective_GBps": 16.0 * total_cells / (dsw * 1e-3) / 1e9,
                                          "note": "both velocity planes per sweep, 16 algorithmic B/cell/sweep"}
        if params[5] >= 1:
            pj = project_microbench(s, W, H, sweeps=params[5])
            out["projection"] = {"sweeps": params[5], "frame_us_two_bands": sum(ab[1]) / 2, "frame_us_one_chain": sum(ab[0]) / 2,
                                 "eager_stage_us_two_bands": pj["us"], "eager_stage_us_one_chain": pj["us_one_chain"],
                                 "note": "Solver:522-547 (divergence, clear, sweeps, subtractGradient) as one loop of launches.  frame_us_*: whole "
                                         "frames (graph replays) with the loop as two row bands out of step (FS_OPT_SPLIT_ROWS, default) / as one "
                                         "chain, interleaved after the timed region.  eager_stage_us_*: the loop alone through FS_STAGE_PROJECT, "
                                         "eager launches back to back -- there the fork/join events cost more than the bands gain"}
        # Contract fields: achieved = ALGORITHMIC bytes per launch / live launch time (the EFFECTIVE bandwidth: T
        # sweeps share one pass over memory, so it may exceed the HBM peak and is NOT a fraction of any hardware
        # limit).  The physical figures follow: hbm_frac (PMC bytes / live time / 8 TB/s) and valu_busy_frac.
        ri, why = load_roofline_inputs()
        roof = {"kernel": "k_jacobi_block<pressure> (%g sweeps/launch)" % sweeps_per_launch,
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "effective_GBps": achieved, "traffic": None,
                "note": "achieved/frac/effective_GBps = algorithmic bytes (12 B/cell/sweep x cells x sweeps per launch) / "
                        "HIP-event launch time: T sweeps share one pass over memory, so this is an effective figure "
                        "and may exceed 1; hbm_frac = PMC bytes / launch time / 8 TB/s is the physical one (at 3072^2 a fabric "
                        "fraction: the planes sit in the Infinity Cache) and valu_issue.frac the share of the VALU pipes; the "
                        "loop is bound by neither (DESIGN.md section 3)"}
        if ri is None:
            roof["traffic_withheld"] = why
        elif (W, H) == (3072, 3072):
            kern = {"pressure_3072": physical_roofline(ri, "pressure_3072", ms_launch * 1e3),
                    "diffuse_3072": physical_roofline(ri, "diffuse_3072", dl * 1e3)}
            roof["traffic"] = kern["pressure_3072"]["hbm_bytes_per_launch_pmc"] if kern["pressure_3072"] else None
            roof["hbm_frac"] = kern["pressure_3072"]["hbm_frac"] if kern["pressure_3072"] else None
            roof["kernels"] = kern
            roof["traffic_source"] = "%s (kernel sources %s)" % (ROOFLINE_INPUTS, ri["source_sha"])
            roof["valu_issue"] = {"peak_lane_ops_per_s": VALU_PEAK_LANE_OPS,
                                  "frac": kern["pressure_3072"]["valu_busy_frac"] if kern["pressure_3072"] else None,
                                  "note": "SQ_ACTIVE_INST_VALU x 4 cycles / (1024 SIMDs x launch time x 2.4 GHz): the share "
                                          "of the launch in which the VALU pipes are occupied; instruction costs in "
                                          "profiles/r02_valu_issue.log (SIMD-32: add/mul/fma 2 cycles, packed / DPP / fp64 4)"}
        out["roofline"] = roof
        if not args.no_cpu_baseline and rank == 0:
            out["cpu_baseline"] = cpu_baseline(W, H, params)
            out["cpu_baseline_all_cores"] = cpu_baseline(W, H, params, budget_s=8.0, omp=True)
    if slabs:
        # untimed: a few more frames with every wait for halo rows timed on the GPU (what was NOT hidden)
        s.set_profile(True)
        s.waits()
        pf = 10
        for f in range(f0 + args.warmup + args.steps, f0 + args.warmup + args.steps + pf):
            step(f)
        w = s.waits()
        s.set_profile(False)
        wt = torch.tensor([w["vel_ms"], w["dye_ms"], w["blocking_ms"]], dtype=torch.float64)
        dist.all_reduce(wt, op=dist.ReduceOp.MAX)
        st = s.stats
        exposed = {"velocity_halo": float(wt[0]) * 1e3 / pf, "dye_halo": float(wt[1]) * 1e3 / pf,
                   "blocking_exchanges": float(wt[2]) * 1e3 / pf,
                   "note": "us per frame the kernels waited at joins (max over ranks, HIP events, 10 extra frames)"}
        te = torch.tensor([t_enq, host_t["step_us"] - host_t["wait_cfl_us"], host_t["wait_cfl_us"]], dtype=torch.float64)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        out["slab"] = {"exposed_wait_us_per_frame": exposed, "check": slab_check, "segment_graphs": s.graph_stats,
                       "host_breakdown_us_per_frame": {k: v / float(args.steps + pf) for k, v in s.host_breakdown().items()},
                       "host_calls_us_per_frame": float(te[0]) * 1e6 / args.steps,
                       "host_enqueue_us_per_frame": float(te[1]), "host_wait_cfl_us_per_frame": float(te[2]),
                       "host_enqueue_note": "max over ranks, timed frames.  host_calls = wall time of the fs_add_force + fs_step calls before any "
                                            "synchronisation of the caller; inside fs_step (fs_slab_host_times) host_wait_cfl = blocked on the CFL "
                                            "value of the PREVIOUS frame, looked at only after this frame's stencil chain has been enqueued (the "
                                            "host runs one frame ahead of the GPU, so this wait is where it idles, not where the GPU does), "
                                            "host_enqueue = the rest: planning, one segment replay, ~8 eager launches and the exchanges.  The frame "
                                            "is host-bound when host_enqueue approaches ms_per_step", "rows_per_gpu": s.r1 - s.r0, "ghost_rows": s.G, "last_advect_halo": s.last_halo,
                       "exchange_rounds_per_frame": st["exchanges"] / float(max(1, st["frames"])),
                       "halo_rows_sent_per_frame": st["rows_sent"] / float(max(1, st["frames"])),
                       "overlap": bool(overlap), "advection_replans": st["redo"],
                       "untimed_ab_ms_per_frame": {"overlap": ab.get(1), "plain": ab.get(0)}}
    if args.workload == "config2" and os.environ.get("FS_BENCH_NO_CONFIG4") != "1":
        STATE["out"] = out
        phase("config4_strong extra (16384^2)")
        try:
            s.close()
            out["config4_strong"] = config4_extra(world if slabs else 1, rank, local_rank, dist if slabs else None)
            j16 = out["config4_strong"].get("jacobi")
            if j16 and not slabs and out.get("roofline", {}).get("kernels") is not None:
                ri, _ = load_roofline_inputs()
                out["roofline"]["kernels"]["pressure_16384"] = physical_roofline(ri, "pressure_16384", j16["pressure_us_per_launch"])
                out["roofline"]["kernels"]["diffuse_16384"] = physical_roofline(ri, "diffuse_16384", j16["diffuse_us_per_launch"])
        except Exception as e:   # the extra must never cost the main line
            out["config4_strong"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if slabs:
        dist.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    except Exception as e:   # a failed run still says so on the one JSON line (no number is made up)
        import traceback
        traceback.print_exc()
        if int(os.environ.get("RANK", "0")) == 0:
            print(json.dumps({"metric": "Gcell-updates/s, full stable-fluids step", "value": None,
                              "unit": "Gcell-updates/s", "n_gpus": int(os.environ.get("WORLD_SIZE", "1")),
                              "error": "%s: %s" % (type(e).__name__, e)}), flush=True)
        sys.exit(1)
