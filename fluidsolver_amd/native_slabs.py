"""Host side of the native row-slab path (layer 3 of include/fluidsolver.h): the frame planner, the
kernels and the RCCL halo exchange all live in libfluidsolver_hip.so; this file only creates the
communicator and forwards calls.

    RcclSlab   one rank of a multi-process run (one process per GPU; launched by torch.distributed.run).
               The 128-byte RCCL unique id travels from rank 0 over an already initialised
               torch.distributed group (any backend) -- the control plane; halo rows never touch torch.
    LocalGroup all K slabs in this process (tests on a 1-GPU box), stepped in lock step.

The reference has one GPU and no counterpart (FluidSolver.swift:269).  slabs.py holds the same
planner in Python as an executable specification; tests/test_slab_plan.py compares the two.
"""
import ctypes as C

import numpy as np

from . import lib as L
from .script import splat

F = np.float32
FIELDS = {"vel": L.FIELD_VELOCITY, "dye": L.FIELD_DENSITY, "p": L.FIELD_PRESSURE, "div": L.FIELD_DIVERGENCE,
          "vort": L.FIELD_VORTICITY}


def make_params(params, display=L.DISPLAY_NONE, brightness=0.06):
    """params = (viscosity, diffusion, deltaTime, vorticityStrength, fadeSpeed, solverIterations)."""
    v, d, dt, vs, fade, n = params
    return L.Params(v, d, dt, vs, fade, int(n), display, brightness)


def plan_dump(W, H, world, rank, ghost, params, halos, overlap=True):
    """Text of the frame plans of len(halos) consecutive update() calls (host only, no GPU)."""
    lib = L.load()
    p = make_params(params)
    hs = (C.c_int * max(1, len(halos)))(*halos)
    buf = C.create_string_buffer(1 << 20)
    L.check(lib.fs_slab_plan_dump(W, H, world, rank, -1 if ghost is None else int(ghost), 1 if overlap else 0,
                                  C.byref(p), len(halos), hs, buf, len(buf)))
    return buf.value.decode()


class _Slab:
    """One fs_ctx created by fs_create_slab."""

    def __init__(self, W, H, params, device, comm, rank, ghost, radius=5.0, overlap=True):
        self.lib = L.load()
        self.W, self.H = int(W), int(H)
        self.radius = float(radius)
        self.deltaTime = float(params[2])
        self.viscosity = float(params[0])
        ctx = C.c_void_p()
        L.check(self.lib.fs_create_slab(self.W, self.H, int(device), comm, int(rank), -1 if ghost is None else int(ghost),
                                        C.byref(ctx)))
        self.ctx = ctx
        p = make_params(params)
        L.check(self.lib.fs_set_params(self.ctx, C.byref(p)), self.ctx)
        self.set_overlap(overlap)
        info = (C.c_int * 6)()
        L.check(self.lib.fs_slab_info(self.ctx, info))
        self.r0, self.r1, self.b0, self.b1, self.G = info[0], info[1], info[2], info[3], info[4]

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.fs_destroy(self.ctx)
            self.ctx = None

    def set_overlap(self, on):
        """FS_OPT_SLAB_OVERLAP: hide the halo traffic behind the kernels (every rank must agree)."""
        L.check(self.lib.fs_set_option(self.ctx, L.OPT_SLAB_OVERLAP, 1 if on else 0), self.ctx)

    def add_force(self, pos, force, color, dt=None):
        col = (C.c_float * 4)(float(color[0]), float(color[1]), float(color[2]), 0.0)
        L.check(self.lib.fs_add_force(self.ctx, float(pos[0]), float(pos[1]), float(force[0]), float(force[1]), col,
                                      self.radius, float(self.deltaTime if dt is None else dt)), self.ctx)

    def sync(self):
        L.check(self.lib.fs_sync(self.ctx), self.ctx)

    def owned(self, name):
        """The rows this slab owns of a field, [planes, r1-r0, W] on the host."""
        fid = FIELDS[name]
        out = np.empty((L.FIELD_PLANES[fid], self.r1 - self.r0, self.W), dtype=F)
        L.check(self.lib.fs_read_field(self.ctx, fid, out.ctypes.data_as(C.c_void_p), out.nbytes), self.ctx)
        return out

    def write_owned(self, name, arr):
        fid = FIELDS[name]
        a = np.ascontiguousarray(arr, dtype=F)
        assert a.shape == (L.FIELD_PLANES[fid], self.r1 - self.r0, self.W), a.shape
        L.check(self.lib.fs_write_field(self.ctx, fid, a.ctypes.data_as(C.c_void_p), a.nbytes), self.ctx)

    @property
    def last_halo(self):
        info = (C.c_int * 6)()
        L.check(self.lib.fs_slab_info(self.ctx, info))
        return info[5]

    @property
    def stats(self):
        st = (C.c_long * 5)()
        L.check(self.lib.fs_slab_stats(self.ctx, st))
        return {"exchanges": st[0], "rows_sent": st[1], "frames": st[2], "redo": st[3], "tracers_received": st[4]}

    def set_fuse(self, mode):
        """FS_OPT_FUSE (every rank must agree): 1 = default size rules, 4 = every fusion at any size, 0 = none."""
        L.check(self.lib.fs_set_option(self.ctx, L.OPT_FUSE, int(mode)), self.ctx)

    def set_profile(self, on):
        """FS_OPT_SLAB_PROFILE: time every wait of the kernels for halo rows (read with waits())."""
        L.check(self.lib.fs_set_option(self.ctx, L.OPT_SLAB_PROFILE, 1 if on else 0), self.ctx)

    def waits(self):
        """Since the last call: ms the kernels waited for the velocity halo, the dye halo and blocking exchanges,
        and the number of waits timed.  Blocks."""
        ms, n = (C.c_double * 3)(), (C.c_long * 3)()
        L.check(self.lib.fs_slab_waits(self.ctx, ms, n), self.ctx)
        return {"vel_ms": ms[0], "dye_ms": ms[1], "blocking_ms": ms[2], "joins": [n[0], n[1], n[2]]}

    def set_option(self, key, value):
        L.check(self.lib.fs_set_option(self.ctx, int(key), int(value)), self.ctx)

    @property
    def graph_stats(self):
        """Segments (exchange-free runs of launches) of this slab's frames: graphs held, replays, recordings, eager runs."""
        st = (C.c_long * 4)()
        L.check(self.lib.fs_slab_graph_stats(self.ctx, st))
        return {"graphs": st[0], "replays": st[1], "captures": st[2], "eager": st[3]}

    def host_breakdown(self):
        """Since the last call: host us spent planning / replaying / recording segments / launching eagerly / enqueueing
        exchanges / the CFL reduction / joins / tracer steps."""
        us = (C.c_double * 8)()
        L.check(self.lib.fs_slab_host_breakdown(self.ctx, us))
        return dict(zip(("plan", "replay", "record", "eager", "exchange", "cfl", "join", "tracers"), [float(x) for x in us]))

    def host_times(self):
        """Since the last call: us per frame the host spent inside fs_step, and of that blocked on the CFL value."""
        us, n = (C.c_double * 2)(), C.c_long()
        L.check(self.lib.fs_slab_host_times(self.ctx, us, C.byref(n)), self.ctx)
        k = max(1, n.value)
        return {"step_us": us[0] / k, "wait_cfl_us": us[1] / k, "frames": n.value}

    # ---- tracers: each lives on the rank that owns the row its lookup reads ----
    def particles_reserve(self, capacity, window=None, **params):
        """Room for `capacity` tracers on THIS rank; turns the tracer step of update() on.  params: fields of
        fs_particle_params (momentum, fluidForce, fluidDecayRate, linkToViscosity) that differ from the defaults."""
        L.check(self.lib.fs_particles_reserve(self.ctx, int(capacity)), self.ctx)
        pp = L.ParticleParams()
        w = window or (float(self.W), float(self.H))
        L.check(self.lib.fs_default_particle_params(C.byref(pp), float(w[0]), float(w[1])))
        pp.viscosity = self.viscosity
        for k, v in params.items():
            setattr(pp, k, type(getattr(pp, k))(v))
        L.check(self.lib.fs_set_particle_params(self.ctx, C.byref(pp)), self.ctx)
        L.check(self.lib.fs_set_option(self.ctx, L.OPT_WITH_PARTICLES, 1), self.ctx)

    def particles_set(self, px, py, vx, vy, life, mass, age):
        """The SAME global arrays on every rank; this rank keeps its index block until the next update()."""
        arrs = [np.ascontiguousarray(a, dtype=F) for a in (px, py, vx, vy, life, mass, age)]
        L.check(self.lib.fs_particles_set(self.ctx, len(arrs[0]), *[a.ctypes.data_as(C.c_void_p) for a in arrs]), self.ctx)

    def particles_spawn(self, at, count, radius=15.0, seed=0):
        """ParticleSystem.addParticles on slabs (fs_particles_spawn): the same call on every rank; this rank appends its
        index block of the new tracers, the next update() re-homes them."""
        L.check(self.lib.fs_particles_spawn(self.ctx, float(at[0]), float(at[1]), int(count), float(radius), int(seed) & 0xFFFFFFFF),
                self.ctx)

    def particles_get(self):
        """This rank's tracers, [7, n] (px, py, vx, vy, life, mass, age), in no particular order."""
        n = self.lib.fs_particles_count(self.ctx)
        out = np.empty((7, n), dtype=F)
        L.check(self.lib.fs_particles_get(self.ctx, n, *[out[i].ctypes.data_as(C.c_void_p) for i in range(7)]), self.ctx)
        return out


class RcclSlab(_Slab):
    """One rank's slab of a multi-process run: halos over RCCL send/recv inside the library."""

    def __init__(self, W, H, params, device, rank, world, dist=None, ghost=None, overlap=True):
        lib = L.load()
        ident = np.zeros(L.COMM_ID_BYTES, np.uint8)
        if rank == 0:
            L.check(lib.fs_comm_unique_id(ident.ctypes.data_as(C.c_void_p)))
        if world > 1 and hasattr(dist, "broadcast_id"):
            ident = dist.broadcast_id(ident, rank)      # any other control plane (a file, MPI, ...)
        elif world > 1:
            if dist is None:
                raise ValueError("world > 1 needs an initialised torch.distributed group to hand out the RCCL id")
            import torch
            t = torch.from_numpy(ident)
            if dist.get_backend() == "nccl":
                t = t.cuda(device)
            dist.broadcast(t, 0)
            ident = t.cpu().numpy().copy()
        comm = C.c_void_p()
        L.check(lib.fs_comm_create_rccl(int(world), int(rank), int(device), ident.ctypes.data_as(C.c_void_p),
                                        C.byref(comm)))
        self.comm = comm
        self.rank, self.world = rank, world
        super().__init__(W, H, params, device, comm, rank, ghost, overlap=overlap)

    def update(self):
        L.check(self.lib.fs_step(self.ctx), self.ctx)

    def frame(self, f, n_pointers=1, forceMultiplier=24.0, colorMultiplier=100.0):
        """One scripted frame of SURVEY.md 8(d): one splat per virtual pointer (every rank applies every
        splat to the rows it holds), then update()."""
        for ptr in range(n_pointers):
            pos, force, col = splat(self.W, self.H, f, ptr, n_pointers, forceMultiplier, colorMultiplier)
            self.add_force(pos, force, col)
        self.update()

    def close(self):
        super().close()
        if getattr(self, "comm", None):
            self.lib.fs_comm_destroy(self.comm)
            self.comm = None


class LoopbackSlab(_Slab):
    """Measurement aid: rank `rank` of a `world`-rank run whose peers do not exist (fs_comm_create_loopback).  Stepped like
    an RcclSlab; does everything that rank would do, with device copies of its own rows in place of the halo rows it would
    receive.  The fields are meaningless; the TIMING is one rank alone on its GPU with a free wire."""

    def __init__(self, W, H, params, device, rank, world, ghost=None, overlap=True):
        lib = L.load()
        comm = C.c_void_p()
        L.check(lib.fs_comm_create_loopback(int(world), int(rank), C.byref(comm)))
        self.comm = comm
        self.rank, self.world = rank, world
        super().__init__(W, H, params, device, comm, rank, ghost, overlap=overlap)

    update = RcclSlab.update
    frame = RcclSlab.frame

    def close(self):
        _Slab.close(self)
        if getattr(self, "comm", None):
            self.lib.fs_comm_destroy(self.comm)
            self.comm = None


class LocalGroup:
    """K slabs of one W x H solver in this process, exchanging by device memcpy, stepped in lock step
    (fs_group_step).  Exists to test the slab planner and kernels on a 1-GPU box.  asynchronous=True:
    the copies run on the slabs' exchange streams, ordered only by the events the RCCL path uses, so
    the K slabs execute concurrently like K ranks."""

    def __init__(self, W, H, params, world, device=0, ghost=None, asynchronous=False, overlap=True):
        self.lib = L.load()
        comm = C.c_void_p()
        make = self.lib.fs_comm_create_local_async if asynchronous else self.lib.fs_comm_create_local
        L.check(make(int(world), C.byref(comm)))
        self.comm = comm
        self.world = world
        self.W, self.H = int(W), int(H)
        self.slabs = [_Slab(W, H, params, device, comm, r, ghost, overlap=overlap) for r in range(world)]
        self._arr = (C.c_void_p * world)(*[s.ctx for s in self.slabs])

    def add_force(self, pos, force, color, dt=None):
        for s in self.slabs:
            s.add_force(pos, force, color, dt)

    def update(self):
        rc = self.lib.fs_group_step(self._arr, self.world)
        if rc:
            raise L.FluidSolverError(rc, L.last_error(None) or L.last_error(self.slabs[0].ctx))

    def sync(self):
        for s in self.slabs:
            s.sync()

    def field(self, name):
        """The whole field, stitched from the owned rows of every slab."""
        return np.concatenate([s.owned(name) for s in self.slabs], axis=1)

    def close(self):
        for s in self.slabs:
            s.close()
        if self.comm:
            self.lib.fs_comm_destroy(self.comm)
            self.comm = None
