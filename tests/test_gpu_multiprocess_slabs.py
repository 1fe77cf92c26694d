"""One PROCESS per rank through fs_comm_create_rccl / fs_step -- the code path of a real multi-GPU run --
on the 1-GPU test box.  RCCL refuses several ranks on one device, so FS_RCCL_LIBRARY points the library at
tests/mock_rccl/libmock_rccl.so, a test double of the nine RCCL calls the runtime uses (messages over
POSIX shared memory; it checks that every send meets a receive of the same element count, in FIFO order
per pair of ranks, and turns a collective issued out of order into a timeout).  What this covers that
the in-process groups cannot: the per-rank segment lists (who sends which rows to whom, in which order),
the id hand-out, and that no rank ever waits for a message nobody sends.  The k-process result must equal
the unsplit CPU oracle bit for bit."""
import multiprocessing as mp
import os
import sys
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
MOCK = os.path.join(HERE, "mock_rccl", "libmock_rccl.so")
F = np.float32


class FileRendezvous:
    """Control plane of the test: rank 0 publishes the 128-byte communicator id in a file."""

    def __init__(self, path):
        self.path = path

    def broadcast_id(self, ident, rank):
        if rank == 0:
            with open(self.path + ".tmp", "wb") as f:
                f.write(ident.tobytes())
            os.rename(self.path + ".tmp", self.path)
            return ident
        t0 = time.time()
        while not os.path.exists(self.path):
            if time.time() - t0 > 60:
                raise TimeoutError("rank 0 never published the communicator id")
            time.sleep(0.01)
        with open(self.path, "rb") as f:
            return np.frombuffer(f.read(), np.uint8).copy()


def _scale(fscale, f):
    """force scale of frame f: a scalar, or one value per frame (a jump makes the halo guess fail)"""
    return fscale[f] if isinstance(fscale, tuple) else fscale


def _reference(case):
    from oracle_binding import DENSITY, PRESSURE, VELOCITY, Oracle, Params
    from fluidsolver_amd.script import splat
    W, H, params, ghost, frames, fscale = case
    o = Oracle(omp=True).solver(W, H)
    p = Params(*params)
    for f in range(frames):
        for ptr in range(2):
            pos, force, col = splat(W, H, 5 * f + 1, ptr, 2)
            sc = F(_scale(fscale, f))
            o.add_force(pos, (force[0] * sc, force[1] * sc), col, 5.0, p.deltaTime)
        o.update(p)
    return o.field(VELOCITY), o.field(DENSITY), o.field(PRESSURE)


def _worker(rank, world, case, outdir, overlap):
    os.environ["FS_RCCL_LIBRARY"] = MOCK
    sys.path.insert(0, ROOT)
    from fluidsolver_amd import lib as L
    from fluidsolver_amd.native_slabs import RcclSlab
    from fluidsolver_amd.script import splat
    assert "torch" not in sys.modules      # torch would bring the real librccl into the process
    W, H, params, ghost, frames, fscale = case
    s = RcclSlab(W, H, params, device=0, rank=rank, world=world, dist=FileRendezvous(os.path.join(outdir, "id")),
                 ghost=ghost, overlap=overlap)
    try:
        L.check(s.lib.fs_comm_selftest(s.comm, 4096))
        halos = []
        for f in range(frames):
            for ptr in range(2):
                pos, force, col = splat(W, H, 5 * f + 1, ptr, 2)
                sc = F(_scale(fscale, f))
                s.add_force(pos, (force[0] * sc, force[1] * sc), col)
            s.update()
            halos.append(s.last_halo)
        s.sync()
        st = s.stats
        np.savez(os.path.join(outdir, "rank%d.npz" % rank), vel=s.owned("vel"), dye=s.owned("dye"), p=s.owned("p"),
                 r0=s.r0, r1=s.r1, halos=np.array(halos), redo=st["redo"], exchanges=st["exchanges"])
    finally:
        s.close()


_REF = {}
CASES = {
    # name: (W, H, params, ghost capacity, frames, force scale)
    "config2_physics": (600, 500, (1.5e-4, 0.0, 0.5, 10.0, 0.002, 40), None, 4, 0.02),
    "fast_flow_replans": (320, 720, (1.5e-4, 0.0, 0.5, 10.0, 0.002, 10), None, 6, (0.2, 0.2, 3.0, 1.0, 0.2, 4.0)),
    "thin_ghost_rechunks": (300, 260, (1.5e-4, 0.00001, 0.5, 10.0, 0.002, 20), 12, 3, 0.02),
    "multi_hop_ghost": (200, 180, (1.5e-4, 0.0, 0.5, 10.0, 0.002, 8), 180, 3, 6.0),
}


@pytest.mark.skipif(not os.path.exists(MOCK), reason="tests/mock_rccl/libmock_rccl.so not built (__graft_entry__.build())")
@pytest.mark.parametrize("overlap", [True, False])
@pytest.mark.parametrize("world", [2, 3, 4])
@pytest.mark.parametrize("name", sorted(CASES))
def test_rank_per_process_equals_oracle(tmp_path, oracle_omp, world, name, overlap):
    case = CASES[name]
    if name == "multi_hop_ghost" and world < 4:
        pytest.skip("needs slabs thinner than the halo")
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, world, case, str(tmp_path), overlap)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(150)
    hung = [p for p in procs if p.is_alive()]
    for p in hung:
        p.kill()
    assert not hung, "ranks hung"
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    if name not in _REF:
        _REF[name] = _reference(case)
    vel, dye, prs = _REF[name]
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        r0, r1 = int(d["r0"]), int(d["r1"])
        assert np.array_equal(d["vel"], vel[:, r0:r1]), (name, "vel", r)
        assert np.array_equal(d["dye"], dye[:, r0:r1]), (name, "dye", r)
        assert np.array_equal(d["p"], prs[:, r0:r1]), (name, "p", r)
        if name == "fast_flow_replans" and overlap:
            assert int(d["redo"]) >= 1, d["halos"]
    assert np.abs(vel).max() > 0


def _tracer_worker(rank, world, W, H, params, frames, n, outdir):
    os.environ["FS_RCCL_LIBRARY"] = MOCK
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    from fluidsolver_amd.native_slabs import RcclSlab
    from fluidsolver_amd.script import splat
    from test_gpu_native_slabs import _tracers
    s = RcclSlab(W, H, params, device=0, rank=rank, world=world, dist=FileRendezvous(os.path.join(outdir, "id")))
    try:
        s.particles_reserve(n, fluidForce=2.0, linkToViscosity=False)
        s.particles_set(*_tracers(n, W, H))
        for f in range(frames):
            for ptr in range(2):
                pos, force, col = splat(W, H, 9 * f + 1, ptr, 2)
                s.add_force(pos, force, col)
            s.update()
        s.sync()
        np.savez(os.path.join(outdir, "tracers%d.npz" % rank), t=s.particles_get(), received=s.stats["tracers_received"])
    finally:
        s.close()


@pytest.mark.skipif(not os.path.exists(MOCK), reason="tests/mock_rccl/libmock_rccl.so not built (__graft_entry__.build())")
@pytest.mark.parametrize("world", [2, 4])
def test_tracers_rank_per_process_equal_single_gpu(tmp_path, world):
    """Tracer migration through the RCCL code path (count matrix by all-reduce, payload by send/recv), one
    process per rank: the union of the ranks' tracers equals the single-GPU tracer set bit for bit."""
    from fluidsolver_amd import FluidSolver, ParticleSystem
    from fluidsolver_amd.script import splat
    from test_gpu_native_slabs import _sorted_columns, _tracers
    W, H, params, n, frames = 448, 420, (1.5e-4, 0.0, 0.5, 10.0, 0.002, 8), 20000, 5
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_tracer_worker, args=(r, world, W, H, params, frames, n, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    s = FluidSolver(W, H, max_particles=n)
    (s.viscosity, s.diffusion, s.deltaTime, s.vorticityStrength, s.fadeSpeed, s.solverIterations) = params
    s.displayMode = None
    ps = ParticleSystem(s)
    ps.fluidForce, ps.linkToViscosity = 2.0, False
    ps.set_arrays(*_tracers(n, W, H))
    for f in range(frames):
        for ptr in range(2):
            pos, force, col = splat(W, H, 9 * f + 1, ptr, 2)
            s.applyForce(pos, force, col)
        s.update()
        ps.update(s, (W, H))
    s.sync()
    exp = _sorted_columns(np.stack(ps.get_arrays()))
    s.close()
    for p in procs:
        p.join(150)
    hung = [p for p in procs if p.is_alive()]
    for p in hung:
        p.kill()
    assert not hung, "ranks hung"
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    parts = [np.load(os.path.join(str(tmp_path), "tracers%d.npz" % r)) for r in range(world)]
    got = _sorted_columns(np.concatenate([d["t"] for d in parts], axis=1))
    assert got.shape == exp.shape and np.array_equal(got, exp)
    assert sum(int(d["received"]) for d in parts) > 0
