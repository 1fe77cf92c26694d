"""Row slabs with the HIP kernels: 2 and 3 ranks share the one GPU of the test box, halos staged
through the host over gloo (RCCL needs one GPU per rank; the exchange bytes are the same).  The
k-slab HIP result must equal the unsplit CPU oracle bit for bit."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
F = np.float32


def _worker(rank, world, port, case, outdir):
    sys.path.insert(0, HERE)
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fluidsolver_amd.script import splat
        from fluidsolver_amd.slabs import Comm, SlabSolver
        W, H, params, ghost, frames, fscale = case
        torch.cuda.set_device(0)
        s = SlabSolver(W, H, params, device=0, comm=Comm(stage_through_host=True), ghost=ghost)
        for f in range(frames):
            for ptr in range(2):
                pos, force, col = splat(W, H, 5 * f + 1, ptr, 2)
                s.add_force(pos, (force[0] * F(fscale), force[1] * F(fscale)), col)
            s.update()
        s.sync()
        np.savez(os.path.join(outdir, "rank%d.npz" % rank), vel=s.owned("vel"), dye=s.owned("dye"), p=s.owned("p"),
                 r0=s.r0, r1=s.r1, exchanges=s.stats["exchanges"], gathers=s.stats["gathers"])
    finally:
        dist.destroy_process_group()


GPU_CASES = {
    "config2_physics": (600, 500, (1.5e-4, 0.0, 0.5, 10.0, 0.002, 40), None, 3, 0.02),
    "thin_ghost_rechunks": (300, 260, (1.5e-4, 0.00001, 0.5, 10.0, 0.002, 20), 12, 3, 0.02),
    "gather_fallback": (200, 180, (1.5e-4, 0.0, 0.5, 10.0, 0.002, 8), 6, 3, 8.0),
}


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("name", sorted(GPU_CASES))
def test_hip_slabs_equal_oracle(tmp_path, oracle_omp, world, name):
    import test_slabs as T
    case = GPU_CASES[name]
    T._spawn(_worker, world, case, str(tmp_path))
    vel, dye, prs = T._reference(case)
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        r0, r1 = int(d["r0"]), int(d["r1"])
        assert np.array_equal(d["vel"], vel[:, r0:r1]), (name, "vel", r)
        assert np.array_equal(d["dye"], dye[:, r0:r1]), (name, "dye", r)
        assert np.array_equal(d["p"], prs[:, r0:r1]), (name, "p", r)
        if name == "gather_fallback":
            assert int(d["gathers"]) > 0
    assert np.abs(vel).max() > 0


def test_single_rank_slab_on_gpu_equals_context():
    """world = 1 SlabSolver (torch-owned buffers + fs_k_* calls) == FluidSolver context (fs_step)."""
    import test_slabs as T
    from fluidsolver_amd import FluidSolver, lib as L
    from fluidsolver_amd.script import splat
    from fluidsolver_amd.slabs import SlabSolver
    W, H, params = 500, 300, (1.5e-4, 0.00001, 0.5, 10.0, 0.002, 17)
    a = SlabSolver(W, H, params, device=0)
    b = FluidSolver(W, H)
    (b.viscosity, b.diffusion, b.deltaTime, b.vorticityStrength, b.fadeSpeed, b.solverIterations) = params
    b.displayMode = None
    for f in range(4):
        pos, force, col = splat(W, H, 3 * f + 1)
        force = (force[0] * F(0.05), force[1] * F(0.05))
        a.add_force(pos, force, col)
        b.applyForce(pos, force, col)
        a.update()
        b.update()
    a.sync()
    b.sync()
    assert np.array_equal(a.owned("vel"), b.read_field(L.FIELD_VELOCITY))
    assert np.array_equal(a.owned("dye"), b.read_field(L.FIELD_DENSITY))
    assert np.array_equal(a.owned("p"), b.read_field(L.FIELD_PRESSURE))
