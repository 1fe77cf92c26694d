"""Multi-rank row-slab path (fluidsolver_amd/slabs.py) under gloo on the CPU, world_size 2 and 3.

The slab driver, halo planner and exchange are the product code; only the per-kernel compute is
swapped for the oracle backend (tests/oracle_backend.py) because this container has no GPU.  The
k-slab result must equal the unsplit oracle bit for bit (SURVEY.md 8(e) "Exactness").
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
F = np.float32


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _spawn(worker, world, case, outdir):
    """mp.spawn on a free port; the port is free when it is picked, not necessarily when rank 0 binds it (shared test
    boxes): a rendezvous that finds it taken is retried on another one."""
    for attempt in range(4):
        try:
            mp.spawn(worker, args=(world, _free_port(), case, outdir), nprocs=world, join=True)
            return
        except Exception as e:   # noqa: BLE001
            if "EADDRINUSE" not in str(e) or attempt == 3:
                raise


def _worker(rank, world, port, case, outdir):
    sys.path.insert(0, HERE)
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle_backend import OracleBackend
        from fluidsolver_amd.slabs import Comm, SlabSolver
        W, H, params, ghost, frames, fscale = case
        s = SlabSolver(W, H, params, device="cpu", comm=Comm(), backend=OracleBackend(), ghost=ghost)
        halos = []
        for f in range(frames):
            from fluidsolver_amd.script import splat
            for ptr in range(2):
                pos, force, col = splat(W, H, 5 * f + 1, ptr, 2)
                s.add_force(pos, (force[0] * F(fscale), force[1] * F(fscale)), col)
            s.update()
            halos.append(getattr(s, "last_halo", 0))
        s.sync()
        np.savez(os.path.join(outdir, "rank%d.npz" % rank), vel=s.owned("vel"), dye=s.owned("dye"), p=s.owned("p"),
                 r0=s.r0, r1=s.r1, exchanges=s.stats["exchanges"], gathers=s.stats["gathers"], halos=np.array(halos))
    finally:
        dist.destroy_process_group()


def _reference(case):
    from oracle_binding import DENSITY, PRESSURE, VELOCITY, Oracle, Params
    from fluidsolver_amd.script import splat
    W, H, params, ghost, frames, fscale = case
    o = Oracle().solver(W, H)
    p = Params(*params)
    for f in range(frames):
        for ptr in range(2):
            pos, force, col = splat(W, H, 5 * f + 1, ptr, 2)
            o.add_force(pos, (force[0] * F(fscale), force[1] * F(fscale)), col, 5.0, p.deltaTime)
        o.update(p)
    return o.field(VELOCITY), o.field(DENSITY), o.field(PRESSURE)


CASES = {
    # name: (W, H, params, ghost capacity, frames, force scale)
    "deep_halo_one_exchange": (96, 120, (1.5e-4, 0.0, 0.5, 10.0, 0.002, 6), None, 4, 0.05),
    "thin_ghost_rechunks": (80, 90, (1.5e-4, 0.0, 0.5, 10.0, 0.002, 9), 5, 3, 0.05),
    "dye_diffusion": (70, 84, (0.00005, 0.00001, 0.4, 10.0, 0.008, 5), None, 3, 0.05),
    "no_viscosity_no_vorticity": (64, 61, (0.0, 0.0, 0.5, 0.0, 0.02, 4), None, 3, 0.05),
    "fast_flow_gather_fallback": (64, 60, (1.5e-4, 0.00001, 0.5, 10.0, 0.002, 4), 6, 3, 3.0),
}


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("name", sorted(CASES))
def test_slabs_equal_unsplit_oracle(tmp_path, world, name):
    case = CASES[name]
    _spawn(_worker, world, case, str(tmp_path))
    vel, dye, prs = _reference(case)
    rows = 0
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        r0, r1 = int(d["r0"]), int(d["r1"])
        assert r0 == rows
        rows = r1
        assert np.array_equal(d["vel"], vel[:, r0:r1]), (name, "vel", r)
        assert np.array_equal(d["dye"], dye[:, r0:r1]), (name, "dye", r)
        assert np.array_equal(d["p"], prs[:, r0:r1]), (name, "p", r)
        frames = case[4]
        if name == "deep_halo_one_exchange":
            # two exchange rounds per frame: velocity at frame start, velocity+dye before advection
            # (frame 0 starts from all-zero fields whose ghosts are valid, so it needs only one)
            assert int(d["exchanges"]) <= 2 * frames and int(d["gathers"]) == 0
        if name == "fast_flow_gather_fallback":
            assert int(d["gathers"]) > 0 and d["halos"].max() > 6
    assert rows == case[1]
    assert np.abs(vel).max() > 0


def test_slab_rows_partition():
    from fluidsolver_amd.slabs import slab_rows
    for H, world in ((16384, 8), (3072, 8), (100, 3), (61, 4)):
        edges = [slab_rows(H, world, r) for r in range(world)]
        assert edges[0][0] == 0 and edges[-1][1] == H
        assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
        sizes = [b - a for a, b in edges]
        assert max(sizes) - min(sizes) <= 1
    assert slab_rows(16384, 8, 3) == (6144, 8192)


def test_single_rank_slab_equals_oracle():
    """world = 1: no exchange, the slab driver is just another host of the same kernels."""
    sys.path.insert(0, HERE)
    from oracle_backend import OracleBackend
    from fluidsolver_amd.slabs import SlabSolver
    case = CASES["dye_diffusion"]
    W, H, params, ghost, frames, fscale = case
    s = SlabSolver(W, H, params, device="cpu", backend=OracleBackend())
    from fluidsolver_amd.script import splat
    for f in range(frames):
        for ptr in range(2):
            pos, force, col = splat(W, H, 5 * f + 1, ptr, 2)
            s.add_force(pos, (force[0] * F(fscale), force[1] * F(fscale)), col)
        s.update()
    vel, dye, prs = _reference(case)
    assert np.array_equal(s.owned("vel"), vel) and np.array_equal(s.owned("dye"), dye)
    assert np.array_equal(s.owned("p"), prs)
