"""Row-slab multi-GPU driver of the stable-fluids step: one process per GPU, halo exchange over
torch.distributed (backend "nccl" = RCCL over xGMI on MI355X; "gloo" in the CPU tests).

The reference has one GPU and no communication (FluidSolver.swift:269); this layer is new work whose
contract is SURVEY.md 8(e):

  * GPU g owns the contiguous rows [r0, r1) of the W x H grid and all columns; its buffers also hold
    up to G ghost rows on each interior side, so halo rows are contiguous in memory.
  * Every stage is the SAME fs_k_* kernel the single-GPU path runs, launched on a row range, so a
    k-GPU run equals the 1-GPU run bit for bit.
  * Ghost rows are not refreshed after every stencil.  Each field carries a "valid depth" (how many
    ghost rows still equal the neighbour's owned rows); a stencil of radius 1 computed on the owned
    rows +- d consumes one row of depth.  Exchanges happen lazily, only when the depth runs out, and
    fetch exactly the depth the rest of the frame needs (2N+4 rows of velocity for vorticity ->
    N diffuse sweeps -> divergence -> N pressure sweeps -> gradient), so a frame costs two exchange
    rounds (velocity at frame start; velocity + dye before advection) instead of one per sweep:
    the exchange is latency-bound, and redundant ghost-row compute is cheap.
  * overlap=True (default, the native runtime's FS_OPT_SLAB_OVERLAP): the velocity advection also
    computes the ghost rows the next frame's chain needs (so the frame-start round merges into the
    pre-advection one), advection is split into interior rows and two rims, and the dye halo is
    fetched at frame start from a guessed depth.  Here everything is synchronous -- this file is the
    executable specification of the PLAN (which rows, which depths, in which order); the native
    runtime runs the same plan with the transfers on a second stream.
  * The semi-Lagrangian backtrace is unbounded (Shaders:133-137), so the advection halo is sized per
    frame from the CFL bound: h = ceil(max|v_y| * dt) + 2 rows, max over all ranks (one 4-byte
    all-reduce).  If h exceeds the ghost capacity the source field is gathered whole (slow path).

Compute goes through a backend object; the product backend is HipBackend (ctypes -> libfluidsolver_hip.so)
and is the only one this package constructs.  Tests inject a CPU backend to exercise the exchange
logic under gloo.
"""
import ctypes as C
import math

import torch

from . import lib as L
from .script import splat

INF = 1 << 30


class HipBackend:
    """fs_k_* entry points on torch CUDA tensors shaped [planes, rows, pitch]."""

    def __init__(self):
        self.lib = L.load()
        if not torch.cuda.is_available():
            raise RuntimeError("HipBackend needs a GPU (there is no CPU fallback)")

    @staticmethod
    def _s():
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    @staticmethod
    def _p(t, plane=0):
        return C.c_void_p(t[plane].data_ptr())

    def alloc(self, planes, rows, pitch, device):
        return torch.zeros((planes, rows, pitch), dtype=torch.float32, device=device)

    def alloc_int(self, device):
        return torch.zeros(1, dtype=torch.int32, device=device)

    def addForces(self, g, vel, dye, pos, force, color, radius, dt):
        L.check(self.lib.fs_k_addForces(C.byref(g), self._p(vel, 0), self._p(vel, 1), self._p(dye, 0), self._p(dye, 1),
                                        self._p(dye, 2), pos[0], pos[1], force[0], force[1], color[0], color[1],
                                        color[2], radius, dt, self._s()))

    def vorticity(self, g, vel, w):
        L.check(self.lib.fs_k_calculateVorticity(C.byref(g), self._p(vel, 0), self._p(vel, 1), self._p(w), self._s()))

    def vorticity_apply(self, g, vel, vel_out, strength, dt):
        L.check(self.lib.fs_k_vorticity_apply(C.byref(g), self._p(vel, 0), self._p(vel, 1), self._p(vel_out, 0),
                                              self._p(vel_out, 1), strength, dt, self._s()))

    def applyVorticity(self, g, w, vel, strength, dt):
        L.check(self.lib.fs_k_applyVorticity(C.byref(g), self._p(w), self._p(vel, 0), self._p(vel, 1), strength, dt,
                                             self._s()))

    def diffuse_n(self, g, a, b, n, visc, dt):
        in_b = C.c_int(0)
        L.check(self.lib.fs_k_diffuse_n(C.byref(g), self._p(a), self._p(b), a.shape[0], a.stride(0), visc, dt, n,
                                        C.byref(in_b), self._s()))
        return bool(in_b.value)

    def divergence(self, g, vel, div):
        L.check(self.lib.fs_k_divergence(C.byref(g), self._p(vel, 0), self._p(vel, 1), self._p(div), self._s()))

    def pressure_n(self, g, div, pa, pb, n, p_is_zero):
        in_b = C.c_int(0)
        L.check(self.lib.fs_k_pressureSolve_n(C.byref(g), self._p(div), self._p(pa), self._p(pb), n,
                                              1 if p_is_zero else 0, C.byref(in_b), self._s()))
        return bool(in_b.value)

    def project_n(self, g, vel, div, pa, pb, n):
        in_b = C.c_int(0)
        L.check(self.lib.fs_k_project_n(C.byref(g), self._p(vel, 0), self._p(vel, 1), self._p(div), self._p(pa),
                                        self._p(pb), n, C.byref(in_b), self._s()))
        return bool(in_b.value)

    def project_n_cfl(self, g, vel, div, pa, pb, n, out):
        out.zero_()
        in_b = C.c_int(0)
        L.check(self.lib.fs_k_project_n_cfl(C.byref(g), self._p(vel, 0), self._p(vel, 1), self._p(div), self._p(pa),
                                            self._p(pb), n, C.c_void_p(out.data_ptr()), C.byref(in_b), self._s()))
        return bool(in_b.value)

    def clear(self, g, t):
        L.check(self.lib.fs_k_clear(C.byref(g), self._p(t), t.shape[0], t.stride(0), self._s()))

    def subtractGradient(self, g, p, vel):
        L.check(self.lib.fs_k_subtractGradient(C.byref(g), self._p(p), self._p(vel, 0), self._p(vel, 1), self._s()))

    def advect(self, g, vel, src, dst, dt, err):
        L.check(self.lib.fs_k_advect(C.byref(g), self._p(vel, 0), self._p(vel, 1), self._p(src), self._p(dst),
                                     src.shape[0], src.stride(0), dt, C.c_void_p(err.data_ptr()), self._s()))

    def advect_fade(self, g, vel, src, dst, dt, rate, err):
        L.check(self.lib.fs_k_advect_fade(C.byref(g), self._p(vel, 0), self._p(vel, 1), self._p(src), self._p(dst),
                                          src.shape[0], src.stride(0), dt, rate, C.c_void_p(err.data_ptr()), self._s()))

    def advect_both(self, g, vel, vel_out, dye, dye_out, dt, rate, err):
        L.check(self.lib.fs_k_advect_both(C.byref(g), self._p(vel, 0), self._p(vel, 1), self._p(vel_out, 0),
                                          self._p(vel_out, 1), self._p(dye), self._p(dye_out), dye.stride(0), dt, 1, rate,
                                          C.c_void_p(err.data_ptr()), self._s()))

    def fade(self, g, dye, rate):
        L.check(self.lib.fs_k_fadeDensity(C.byref(g), self._p(dye), dye.shape[0], dye.stride(0), rate, self._s()))

    def max_abs(self, g, plane, out):
        out.zero_()
        L.check(self.lib.fs_k_max_abs(C.byref(g), C.c_void_p(plane.data_ptr()), C.c_void_p(out.data_ptr()), self._s()))

    def subtractGradient_cfl(self, g, p, vel, out):
        out.zero_()
        L.check(self.lib.fs_k_subtractGradient_cfl(C.byref(g), self._p(p), self._p(vel, 0), self._p(vel, 1),
                                                   C.c_void_p(out.data_ptr()), self._s()))


class Comm:
    """Thin wrapper of torch.distributed point-to-point + the 4-byte max all-reduce.
    stage_through_host=True copies CUDA halos through pinned host memory (gloo cannot move device
    memory); it exists for 2-rank tests on a 1-GPU box and is never used with RCCL."""

    def __init__(self, group=None, stage_through_host=False):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.stage = stage_through_host

    def exchange(self, sends, recvs):
        """sends / recvs: lists of (peer_rank, contiguous tensor view)."""
        dist = self.dist
        if self.stage:
            s_host = [(p, t.cpu()) for p, t in sends]
            r_host = [(p, torch.empty(t.shape, dtype=t.dtype)) for p, t in recvs]
            ops = [dist.P2POp(dist.isend, t, p, self.group) for p, t in s_host]
            ops += [dist.P2POp(dist.irecv, t, p, self.group) for p, t in r_host]
            for r in dist.batch_isend_irecv(ops):
                r.wait()
            for (_, dst), (_, src) in zip(recvs, r_host):
                dst.copy_(src)
            return
        ops = [dist.P2POp(dist.isend, t, p, self.group) for p, t in sends]
        ops += [dist.P2POp(dist.irecv, t, p, self.group) for p, t in recvs]
        if ops:
            for r in dist.batch_isend_irecv(ops):
                r.wait()

    def allreduce_max(self, t):
        if self.stage and t.is_cuda:
            h = t.cpu()
            self.dist.all_reduce(h, op=self.dist.ReduceOp.MAX, group=self.group)
            t.copy_(h)
        else:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)

    def broadcast(self, t, src):
        if self.stage and t.is_cuda:
            h = t.cpu()
            self.dist.broadcast(h, src, group=self.group)
            t.copy_(h)
        else:
            self.dist.broadcast(t, src, group=self.group)


def slab_rows(H, world, rank):
    """Contiguous split of H rows: the first H % world ranks get one extra row."""
    base, rem = divmod(H, world)
    r0 = rank * base + min(rank, rem)
    return r0, r0 + base + (1 if rank < rem else 0)


class SlabSolver:
    """One rank's slab of a W x H solver.  params = (viscosity, diffusion, deltaTime, vorticityStrength,
    fadeSpeed, solverIterations), the fields of fs_params that update() reads."""

    def __init__(self, W, H, params, device=None, comm=None, backend=None, ghost=None, radius=5.0, overlap=True):
        self.W, self.H = int(W), int(H)
        self.overlap = bool(overlap)
        (self.viscosity, self.diffusion, self.deltaTime, self.vorticityStrength, self.fadeSpeed,
         self.solverIterations) = params
        self.radius = float(radius)
        self.comm = comm
        if comm is None:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                self.comm = Comm()
        self.rank = self.comm.rank if self.comm else 0
        self.world = self.comm.world if self.comm else 1
        self.be = backend if backend is not None else HipBackend()
        if device is None:
            device = "cuda"
        elif isinstance(device, int):
            device = "cuda:%d" % device
        self.device = torch.device(device)
        self.pitch = L.pitch_for_width(self.W)
        self.r0, self.r1 = slab_rows(self.H, self.world, self.rank)
        own_min = self.H // self.world
        if self.world > 1 and own_min < 4:
            raise ValueError("slabs of %d rows are too thin for %d ranks" % (own_min, self.world))
        self.G = 0 if self.world == 1 else min(own_min, int(ghost) if ghost is not None else own_min)
        if self.world > 1 and self.G < 2:
            raise ValueError("ghost capacity must be >= 2 rows")
        self.b0 = max(0, self.r0 - self.G)
        self.b1 = min(self.H, self.r1 + self.G)
        rows = self.b1 - self.b0
        al = lambda p: self.be.alloc(p, rows, self.pitch, self.device)   # noqa: E731
        self.f = {"vel": al(2), "vel_t": al(2), "dye": al(3), "dye_t": al(3), "p": al(1), "p_t": al(1),
                  "div": al(1), "vort": al(1)}
        self.err = self.be.alloc_int(self.device)
        self.cfl = torch.zeros(1, dtype=torch.float32, device=self.device)
        # valid ghost depth per field; zero-initialised fields are valid everywhere
        self.valid = {k: INF for k in ("vel", "dye", "p", "div", "vort")}
        self.stats = {"exchanges": 0, "rows_sent": 0, "gathers": 0}
        self._full = {}

    # ---- geometry ------------------------------------------------------------------------------
    def grid(self, depth):
        """fs_grid computing the owned rows +- depth (clipped to the buffer / domain)."""
        yb = max(self.b0, self.r0 - depth)
        ye = min(self.b1, self.r1 + depth)
        return L.Grid(self.W, self.H, self.pitch, self.b0, self.b1 - self.b0, yb, ye)

    def _rows(self, t, y0, y1):
        return t[:, y0 - self.b0:y1 - self.b0, :]

    # ---- halo exchange -------------------------------------------------------------------------
    def _join(self, name):
        """The native runtime waits here for the asynchronous exchange of `name`; nothing to do here."""

    def _ensure(self, wants, tag="exchange"):
        """wants: list of (field, depth).  Refresh the ghost rows of every field whose valid depth is
        short, in ONE batched send/recv round.  Depths are capped at the ghost capacity."""
        if self.world == 1:
            return
        todo = [(n, min(d, self.G)) for n, d in wants if self.valid[n] < min(d, self.G)]
        if not todo:
            return
        sends, recvs = [], []
        up, dn = self.rank - 1, self.rank + 1
        for name, d in todo:
            t = self.f[name]
            for pl in range(t.shape[0]):
                if up >= 0:
                    sends.append((up, t[pl, self.r0 - self.b0:self.r0 - self.b0 + d, :]))
                    recvs.append((up, t[pl, self.r0 - d - self.b0:self.r0 - self.b0, :]))
                if dn < self.world:
                    sends.append((dn, t[pl, self.r1 - d - self.b0:self.r1 - self.b0, :]))
                    recvs.append((dn, t[pl, self.r1 - self.b0:self.r1 - self.b0 + d, :]))
                self.stats["rows_sent"] += d * ((up >= 0) + (dn < self.world))
        self.comm.exchange(sends, recvs)
        self.stats["exchanges"] += 1
        for name, d in todo:
            self.valid[name] = d

    # ---- Jacobi loops with lazy re-exchange ----------------------------------------------------
    def _sweeps(self, kind, name, N, need_out, with_gradient=False):
        """N sweeps on field `name` (ping-pong with name_t).  One sweep computed on the owned rows +- d
        reads the previous iterate on +- (d+1), i.e. consumes one row of ghost depth.  The sweeps run in
        as few temporally blocked chunks as the available depth allows; ghosts are re-exchanged only
        when the depth is used up.  The caller checks the final depth against need_out."""
        multi = self.world > 1
        remaining = N
        while remaining > 0:
            if not multi:
                n, out_depth = remaining, 0
            else:
                want = need_out + remaining          # depth that finishes the loop with no further exchange
                if self.valid[name] < 1:
                    self._ensure([(name, want)])
                a = min(self.valid[name], self.G)
                if a >= want:
                    n, out_depth = remaining, need_out
                else:
                    n = min(remaining, a)
                    out_depth = a - n
                if kind == "pressure":
                    # the sweep at depth d reads div at depth d; the deepest sweep of the chunk is out_depth + n - 1
                    if self.valid["div"] < out_depth + n - 1:
                        self._ensure([("div", need_out + remaining - 1)])
                    vd = min(self.valid["div"], self.G)
                    if vd < out_depth + n - 1:
                        n = min(n, vd + 1)
                        out_depth = min(out_depth, vd - (n - 1))
                assert n >= 1 and out_depth >= 0 and out_depth + n <= self.G, (n, out_depth, self.G)
            g = self.grid(out_depth)
            src, dst = self.f[name], self.f[name + "_t"]
            if kind == "pressure" and with_gradient:
                in_b = self.be.project_n(g, self.f["vel"], self.f["div"], src, dst, n)     # :522-547 in one loop
                self._p_zero = False
            elif kind == "pressure":
                in_b = self.be.pressure_n(g, self.f["div"], src, dst, n, self._p_zero)
                self._p_zero = False
            else:
                coeff = self.viscosity if name == "vel" else self.diffusion
                in_b = self.be.diffuse_n(g, src, dst, n, float(coeff), float(self.deltaTime))
            if in_b:
                self.f[name], self.f[name + "_t"] = dst, src
            if multi:
                self.valid[name] = out_depth
            remaining -= n

    def _need(self, name, depth):
        """Refresh `name` if its valid ghost depth is short of `depth` (capped at the capacity)."""
        if self.world > 1 and self.valid[name] < min(depth, self.G):
            self._ensure([(name, depth)])

    # ---- the frame -----------------------------------------------------------------------------
    def add_force(self, pos, force, color, dt=None):
        """applyForce (FluidSolver.swift:444-470): every slab applies the splat to all rows it holds
        (owned + ghost), so ghost depths are unchanged."""
        g = self.grid(self.G)
        self.be.addForces(g, self.f["vel"], self.f["dye"], [float(x) for x in pos], [float(x) for x in force],
                          [float(x) for x in color], self.radius, float(self.deltaTime if dt is None else dt))

    def _chain_needs(self):
        """Ghost depths the velocity chain needs, backwards from subtractGradient on the owned rows:
        p(1) <- N sweeps <- div(N) <- vel(N+1) <- N diffuse sweeps <- vel(2N+1) <- applyVorticity
        <- vort(2N+2) <- vel(2N+3)"""
        N = int(self.solverIterations)
        need_p = 1
        need_div = need_p + max(N - 1, 0)
        need_v2 = need_div + 1
        need_v1 = need_v2 + (N if self.viscosity > 0 else 0)
        need_v0 = need_v1 + (2 if self.vorticityStrength > 0 else 0)
        return need_p, need_div, need_v2, need_v1, need_v0

    def update(self, h_spec=None):
        """FluidSolver.update(), FluidSolver.swift:477-591, on this slab.  h_spec: the advection halo the
        frame is expected to need (the native runtime guesses it from the previous frame)."""
        N = int(self.solverIterations)
        dt = float(self.deltaTime)
        vort_on, visc_on, ddiff_on = self.vorticityStrength > 0, self.viscosity > 0, self.diffusion > 0
        multi = self.world > 1
        V = self.valid
        eff = lambda name: min(V[name], self.G)   # noqa: E731  usable depth: valid AND present in the buffer
        need_p, need_div, need_v2, need_v1, need_v0 = self._chain_needs()
        ddn = N if (ddiff_on and N > 0) else 0
        if not (multi and self.overlap and ddn == 0):
            h_spec = None
        if h_spec is not None:
            self._ensure([("dye", h_spec)], "exchange_async")                  # under the whole stencil chain

        hoisted = ddn and not multi
        if hoisted:
            # one GPU: the dye diffusion touches nothing but the dye, the native runtime runs it beside the whole
            # velocity chain (a third stream); here it simply comes first                     :562-574
            self._sweeps("diffuse", "dye", N, 0)
        if vort_on:                                                            # :486-502
            self._need("vel", need_v0)
            dw = min(need_v1 + 1, eff("vel") - 1) if multi else 0
            dv = min(need_v1, dw - 1, eff("vel")) if multi else 0
            # out-of-cache planes: both passes in one, out of place, the vorticity plane never touches memory
            big = self.pitch * (self.H // self.world) * 4.0 >= 256.0 * 1024 * 1024
            if (big or self.W >= 512) and hasattr(self.be, "vorticity_apply"):   # (the marching form wins wherever it applies)
                self.be.vorticity_apply(self.grid(dv), self.f["vel"], self.f["vel_t"],
                                        float(self.vorticityStrength), dt)     # :486-502
                self.f["vel"], self.f["vel_t"] = self.f["vel_t"], self.f["vel"]
            else:
                self.be.vorticity(self.grid(dw), self.f["vel"], self.f["vort"])    # :488-491
                self.be.applyVorticity(self.grid(dv), self.f["vort"], self.f["vel"],
                                       float(self.vorticityStrength), dt)          # :494-501
            if multi:
                V["vort"], V["vel"] = dw, dv
        if visc_on and N > 0:                                                  # :505-518
            self._need("vel", need_v1)
            self._sweeps("diffuse", "vel", N, need_v2)
        self._need("vel", need_v2)
        # The projection in one loop (fs_k_project_n): the divergence is computed by the first pressure launch and the
        # gradient subtracted by the last, from registers -- at every size since round 3 (slab.hip plan_a).  A slab needs
        # N + 1 ghost rows of velocity for it.
        grad_fused = (N > 0 and hasattr(self.be, "project_n")
                      and (not multi or (self.G >= N + 1 and hasattr(self.be, "project_n_cfl"))))
        dd = min(need_div, eff("vel") - 1) if multi else 0
        if not grad_fused:
            self.be.divergence(self.grid(dd), self.f["vel"], self.f["div"])    # :522-525
        if multi:
            V["div"] = dd
        self._p_zero = True                                                    # clear folded in, :528-530
        V["p"] = INF
        if grad_fused and multi:
            in_b = self.be.project_n_cfl(self.grid(0), self.f["vel"], self.f["div"], self.f["p"], self.f["p_t"], N,
                                         self.cfl)                             # :522-547 + the CFL bound
            if in_b:
                self.f["p"], self.f["p_t"] = self.f["p_t"], self.f["p"]
            V["p"], V["div"] = 0, 0
            self._p_zero = False
        elif N > 0:
            self._sweeps("pressure", "p", N, need_p, grad_fused)               # :533-541 (+ :544-547)
        else:
            self.be.clear(self.grid(self.G), self.f["p"])
        if not grad_fused:
            self._need("p", 1)
        # ---- advection: halo from the CFL bound of the backtrace ----
        h = 0
        if grad_fused and not multi:
            pass
        elif not multi:
            self.be.subtractGradient(self.grid(0), self.f["p"], self.f["vel"])     # :544-547
        else:
            # ... with max |v_y| of the owned rows in the same pass, then max over ranks
            if not grad_fused:
                self.be.subtractGradient_cfl(self.grid(0), self.f["p"], self.f["vel"], self.cfl)
            V["vel"] = 0
            self.comm.allreduce_max(self.cfl)
            # bilinear taps lie within ceil(|v_y| dt) + 1 rows; +1 more because the dye is traced through
            # the ADVECTED velocity, whose fp32 mix() can exceed the old maximum by rounding only
            h = int(math.ceil(float(self.cfl.item()) * abs(dt))) + 2
            self.last_halo = h
            if h_spec is not None and h_spec >= h:
                h = h_spec                                                     # the guess holds: plan with it
        if multi and h > self.G:                                               # slow path: gather whole fields
            self._advect_gathered("vel")
            if ddn:
                self._need("dye", ddn)
                self._sweeps("diffuse", "dye", N, 0)
            self._advect_gathered("dye")
            self.be.fade(self.grid(0), self.f["dye"], float(self.fadeSpeed))   # :587-591
            V["dye"] = 0
            return
        fused = hasattr(self.be, "advect_fade")
        # both advections of the same rows in one pass: pays only when a plane of the slab outgrows the Infinity Cache
        both = (fused and hasattr(self.be, "advect_both")
                and self.pitch * (self.H // self.world) * 4.0 >= 256.0 * 1024 * 1024)
        # overlap: the advected velocity is also computed on the ghost rows the next frame's chain
        # consumes (as far as the ghost capacity allows), from a velocity halo of h + adv rows
        adv = max(0, min(need_v0, self.G - h)) if (multi and self.overlap) else 0
        split = multi and self.overlap and ddn == 0 and self.H // self.world > 2 * h + 8
        if split:
            self._ensure([("vel", h + adv), ("dye", h)], "exchange_async")
            i0 = self.r0 + (h if self.r0 > 0 else 0)
            i1 = self.r1 - (h if self.r1 < self.H else 0)
            old, new = self.f["vel"], self.f["vel_t"]
            # interior rows: no tap of their backtrace leaves the owned rows    :550-558, :577-591
            if both:
                self._advect_both_rows(i0, i1)
            else:
                self._advect_rows("vel", old, i0, i1, False)
                self._advect_rows("dye", new, i0, i1, fused)
            self._join("vel")
            self._advect_rows("vel", old, self.r0 - adv, i0, False)
            self._advect_rows("vel", old, i1, self.r1 + adv, False)
            self.f["vel"], self.f["vel_t"] = new, old                          # :558
            V["vel"] = adv
            self._join("dye")
            self._advect_rows("dye", new, self.r0, i0, fused)
            self._advect_rows("dye", new, i1, self.r1, fused)
            self.f["dye"], self.f["dye_t"] = self.f["dye_t"], self.f["dye"]    # :584
        else:
            if multi:
                self._ensure([("vel", h + adv), ("dye", h + ddn)])             # one batched round
                self._join("dye")
            if both and not multi and not ddn:
                self._advect_both_rows(self.r0, self.r1)                       # :550-558 + :577-591 in one pass
                self.f["vel"], self.f["vel_t"] = self.f["vel_t"], self.f["vel"]
            else:
                self._advect_rows("vel", self.f["vel"], self.r0 - adv, self.r1 + adv, False)   # :550-558
                self.f["vel"], self.f["vel_t"] = self.f["vel_t"], self.f["vel"]
                if multi:
                    V["vel"] = adv
                if ddn and not hoisted:
                    self._sweeps("diffuse", "dye", N, h)                       # :562-574
                    self._need("dye", h)
                self._advect_rows("dye", self.f["vel"], self.r0, self.r1, fused)   # :577-584 (+ :587-591 when fused)
            self.f["dye"], self.f["dye_t"] = self.f["dye_t"], self.f["dye"]
        if not fused:
            self.be.fade(self.grid(0), self.f["dye"], float(self.fadeSpeed))   # :587-591
        if multi:
            V["dye"] = 0

    def _advect_both_rows(self, yb, ye):
        yb = max(yb, self.b0)
        ye = max(yb, min(ye, self.b1))
        g = L.Grid(self.W, self.H, self.pitch, self.b0, self.b1 - self.b0, yb, ye)
        self.be.advect_both(g, self.f["vel"], self.f["vel_t"], self.f["dye"], self.f["dye_t"], float(self.deltaTime),
                            float(self.fadeSpeed), self.err)

    def _advect_rows(self, name, vel, yb, ye, fade):
        """advect field `name` through `vel` on rows [yb, ye) (clipped to the buffer), into its partner."""
        yb = max(yb, self.b0)
        ye = max(yb, min(ye, self.b1))
        g = L.Grid(self.W, self.H, self.pitch, self.b0, self.b1 - self.b0, yb, ye)
        src, dst = self.f[name], self.f[name + "_t"]
        if fade:
            self.be.advect_fade(g, vel, src, dst, float(self.deltaTime), float(self.fadeSpeed), self.err)
        else:
            self.be.advect(g, vel, src, dst, float(self.deltaTime), self.err)

    def _gather_full(self, name):
        """Slow path: every rank receives the whole field (row blocks broadcast by their owners)."""
        t = self.f[name]
        key = (name, t.shape[0])
        if key not in self._full:
            self._full[key] = self.be.alloc(t.shape[0], self.H, self.pitch, self.device)
        full = self._full[key]
        for r in range(self.world):
            a, b = slab_rows(self.H, self.world, r)
            if r == self.rank:
                full[:, a:b, :].copy_(self._rows(t, a, b))
            for pl in range(t.shape[0]):
                self.comm.broadcast(full[pl, a:b, :], r)
        self.stats["gathers"] += 1
        return full

    def _advect_gathered(self, name):
        vel_full = self._gather_full("vel")
        src_full = vel_full if name == "vel" else self._gather_full(name)
        key = (name + "_out", src_full.shape[0])
        if key not in self._full:
            self._full[key] = self.be.alloc(src_full.shape[0], self.H, self.pitch, self.device)
        out_full = self._full[key]
        g = L.Grid(self.W, self.H, self.pitch, 0, self.H, self.r0, self.r1)
        self.be.advect(g, vel_full, src_full, out_full, float(self.deltaTime), self.err)
        src, dst = self.f[name], self.f[name + "_t"]
        self._rows(dst, self.r0, self.r1).copy_(out_full[:, self.r0:self.r1, :])
        self.f[name], self.f[name + "_t"] = dst, src
        self.valid[name] = 0

    def frame(self, f, n_pointers=1, forceMultiplier=24.0, colorMultiplier=100.0):
        """One scripted frame of SURVEY.md 8(d): one splat per virtual pointer, then update()."""
        for ptr in range(n_pointers):
            pos, force, col = splat(self.W, self.H, f, ptr, n_pointers, forceMultiplier, colorMultiplier)
            self.add_force(pos, force, col)
        self.update()

    def sync(self):
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        if int(self.err.item()) != 0:
            self.err.zero_()
            raise L.FluidSolverError(-4, "advect backtrace left the buffered rows (halo too small)")

    def owned(self, name):
        """The rows this rank owns of a field, [planes, r1-r0, W] on the host."""
        return self._rows(self.f[name], self.r0, self.r1)[:, :, :self.W].cpu().numpy()
