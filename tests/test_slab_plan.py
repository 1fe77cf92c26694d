"""The native frame planner (fluidsolver_amd/csrc/slab.hip, dumped by fs_slab_plan_dump -- host code,
no GPU) against the Python planner (fluidsolver_amd/slabs.py), which the gloo tests of test_slabs.py
prove bit-exact against the unsplit oracle.  Same stages, same row ranges, same exchange depths, in
the same order, for every rank and over several frames (ghost depths carry over between frames)."""
import os
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from fluidsolver_amd import native_slabs as NS  # noqa: E402
from fluidsolver_amd.slabs import SlabSolver  # noqa: E402


class _Recorder:
    """Backend that records what the Python planner launches instead of computing it."""

    def __init__(self, log):
        self.log = log
        self.next_cfl = 0.0

    def alloc(self, planes, rows, pitch, device):
        return torch.zeros((planes, 1, 1))     # the planner never looks inside

    def alloc_int(self, device):
        return torch.zeros(1, dtype=torch.int32)

    def _rec(self, name, g, n=0):
        self.log.append("%s %d %d %d" % (name, g.yb, g.ye, n))

    def vorticity(self, g, vel, w): self._rec("vorticity", g)
    def applyVorticity(self, g, w, vel, s, dt): self._rec("applyVorticity", g)
    def vorticity_apply(self, g, vel, vel_out, s, dt): self._rec("vorticity_apply", g)
    def divergence(self, g, vel, div): self._rec("divergence", g)
    def clear(self, g, t): self._rec("clear", g)
    def subtractGradient(self, g, p, vel): self._rec("subtractGradient", g)
    def fade(self, g, dye, rate): self._rec("fade", g)
    def advect(self, g, vel, src, dst, dt, err): self._rec("advect", g, src.shape[0])
    def advect_fade(self, g, vel, src, dst, dt, rate, err): self._rec("advect_fade", g, src.shape[0])
    def advect_both(self, g, vel, vel_out, dye, dye_out, dt, rate, err): self._rec("advect_both", g, 5)

    def diffuse_n(self, g, a, b, n, visc, dt):
        self._rec("diffuse_n", g, n)
        return bool(self._launches(1, n) & 1)

    def pressure_n(self, g, div, pa, pb, n, p_is_zero):
        self._rec("pressure_n", g, n)
        return bool(self._launches(0, n) & 1)

    def project_n(self, g, vel, div, pa, pb, n):
        self._rec("project_n", g, n)
        return bool(self._launches(0, n) & 1)

    def project_n_cfl(self, g, vel, div, pa, pb, n, out):
        self._rec("project_n_cfl", g, n)
        out.fill_(self.next_cfl)
        return bool(self._launches(0, n) & 1)

    @staticmethod
    def _launches(op, n):
        from fluidsolver_amd import lib as L
        depth = L.load().fs_config_get(L.OPT_BLOCK_DEPTH_DIFFUSE if op else L.OPT_BLOCK_DEPTH_PRESSURE)
        return (n + depth - 1) // depth

    def subtractGradient_cfl(self, g, p, vel, out):
        self._rec("subtractGradient_cfl", g)
        out.fill_(self.next_cfl)


class _Comm:
    def __init__(self, rank, world, log):
        self.rank, self.world, self.log = rank, world, log

    def allreduce_max(self, t):
        self.log.append("cfl_allreduce")


class _PlanOnly(SlabSolver):
    def _ensure(self, wants, tag="exchange"):
        if self.world == 1:
            return
        todo = [(n, min(d, self.G)) for n, d in wants if self.valid[n] < min(d, self.G)]
        if todo:
            self.comm.log.append(tag + " " + " ".join("%s:%d" % t for t in todo))
            for n, d in todo:
                self.valid[n] = d

    def _join(self, name):
        self.comm.log.append("join " + name)


def python_plan(W, H, world, rank, ghost, params, halos, overlap=True):
    log = []
    be = _Recorder(log)
    s = _PlanOnly(W, H, params, device="cpu", comm=_Comm(rank, world, log) if world > 1 else None, backend=be,
                  ghost=ghost, overlap=overlap)
    s.comm = _Comm(rank, world, log) if world > 1 else None
    dt = abs(params[2])
    for f, h in enumerate(halos):
        be.next_cfl = (h - 2.5) / dt if world > 1 else 0.0   # ceil(cfl * dt) + 2 == h
        s.update(h_spec=h if f > 0 else None)                # frames after the first guess their halo (here: right)
        if world > 1:
            assert s.last_halo == h
    return "\n".join(log) + "\n"


CASES = {
    # name: (W, H, params, ghost, halos per frame)
    "config2_8gpu": (16384, 16384, (1.5e-4, 0.0, 0.5, 10.0, 0.002, 40), None, [2, 9, 30]),
    "config5_weak": (8192, 65536, (1.5e-4, 0.0, 0.5, 10.0, 0.002, 100), None, [3, 3]),
    "thin_ghost_rechunks": (300, 260, (1.5e-4, 0.00001, 0.5, 10.0, 0.002, 20), 12, [3, 5, 12]),
    "tiny_ghost": (128, 96, (1.5e-4, 0.0, 0.25, 10.0, 0.002, 7), 2, [2, 2]),
    "dye_diffusion": (70, 84, (0.00005, 0.00001, 0.5, 10.0, 0.008, 5), None, [4, 6, 2]),
    "no_viscosity_no_vorticity": (64, 61, (0.0, 0.0, 0.5, 0.0, 0.02, 4), None, [3, 3]),
    "zero_iterations": (64, 64, (1.5e-4, 0.0, 0.5, 10.0, 0.002, 0), None, [2, 2]),
    "uneven_rows": (100, 103, (1.5e-4, 0.0, 0.5, 0.0, 0.002, 11), 9, [4, 9]),
}


@pytest.mark.parametrize("overlap", [True, False])
@pytest.mark.parametrize("world", [1, 2, 3, 8])
@pytest.mark.parametrize("name", sorted(CASES))
def test_native_plan_equals_python_plan(name, world, overlap):
    W, H, params, ghost, halos = CASES[name]
    if H // world < 4:
        pytest.skip("slabs too thin")
    for rank in sorted({0, world // 2, world - 1}):
        native = NS.plan_dump(W, H, world, rank, ghost, params, halos, overlap)
        python = python_plan(W, H, world, rank, ghost, params, halos, overlap)
        assert native == python, "rank %d of %d\n--- native ---\n%s--- python ---\n%s" % (rank, world, native, python)
        assert "subtractGradient" in native or "project_n" in native
        if world > 1:
            assert "cfl_allreduce" in native


def test_deep_halo_plan_has_two_exchange_rounds_per_frame():
    """Config 4 on 8 GPUs without overlap: after the first frame (all-zero fields need no exchange) every
    frame costs exactly two exchange rounds -- velocity at frame start, velocity + dye before advection."""
    W, H, params, ghost, _ = CASES["config2_8gpu"]
    text = NS.plan_dump(W, H, 8, 3, None, params, [5, 5, 5], overlap=False)
    frames = text.split("cfl_allreduce\n")
    assert len(frames) == 4
    assert text.count("exchange") == 1 + 2 + 2      # frame 0: only the pre-advection round
    assert "exchange vel:83\n" in text              # 2N + 3 rows of velocity in one round
    assert "exchange vel:5 dye:5\n" in text
    assert "async" not in text and "join" not in text.replace("join dye\n", "")


def test_overlap_plan_hides_both_rounds():
    """Config 4 on 8 GPUs with overlap: two asynchronous rounds per frame and no blocking one -- the dye
    halo at frame start (joined before the dye rims), velocity h + 2N+3 rows after the projection (joined
    before the velocity rims); the interior rows are advected in between; the next frame starts without
    an exchange because the advected velocity was also computed on 2N+3 ghost rows."""
    W, H, params, ghost, _ = CASES["config2_8gpu"]
    text = NS.plan_dump(W, H, 8, 3, None, params, [5, 7, 7])
    lines = text.split("\n")
    assert "exchange vel:83" not in lines
    frame2 = text.split("cfl_allreduce\n")[2]                  # advection of frame 1 + chain of frame 2
    f = frame2.split("\n")
    r0, r1 = 3 * 2048, 4 * 2048
    assert f[0] == "exchange_async vel:90"                      # 7 + 83
    assert f[1] == "advect %d %d 2" % (r0 + 7, r1 - 7)
    assert f[2] == "advect_fade %d %d 3" % (r0 + 7, r1 - 7)
    assert f[3] == "join vel"
    assert f[4] == "advect %d %d 2" % (r0 - 83, r0 + 7) and f[5] == "advect %d %d 2" % (r1 - 7, r1 + 83)
    assert f[6] == "join dye"
    assert f[7] == "advect_fade %d %d 3" % (r0, r0 + 7) and f[8] == "advect_fade %d %d 3" % (r1 - 7, r1)
    assert f[9] == "exchange_async dye:7"                       # frame 2 begins: dye halo under the chain
    assert f[10].startswith("vorticity_apply %d %d" % (r0 - 81, r1 + 81))   # curl + confinement in one marching pass (W >= 512)
    # one GPU, 16384^2: a plane (1 GiB) outgrows the Infinity Cache -> both advections in one pass
    one = NS.plan_dump(W, H, 1, 0, None, params, [0])
    assert "advect_both 0 16384 5\n" in one and "advect_fade" not in one
    assert "vorticity_apply 0 16384 0\n" in one and "applyVorticity" not in one
    assert not any(x.startswith("exchange ") for x in f)


def test_planner_host_code_is_clean_under_asan_and_ubsan(tmp_path):
    """The library's host code that plans frames, compiled with hipcc and ASan + UBSan on the HOST side only
    (the device code is untouched; no GPU is needed: the plan dump runs on plan-only contexts)."""
    import subprocess
    root = os.path.dirname(HERE)
    csrc = os.path.join(root, "fluidsolver_amd", "csrc")
    exe = str(tmp_path / "plan_check")
    cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-O1", "-g", "-std=c++17", "-ffp-contract=off",
           "--offload-arch=gfx950", "-Xarch_host", "-fsanitize=address", "-Xarch_host", "-fsanitize=undefined",
           "-Xarch_host", "-fno-sanitize-recover=undefined",
           "-I", os.path.join(root, "include"), "-I", csrc, "-o", exe, os.path.join(HERE, "host_sanitize", "plan_check.cpp")]
    cmd += [os.path.join(csrc, n) for n in ("passes.hip", "jacobi.hip", "context.hip", "slab.hip")] + ["-ldl"]
    subprocess.run(cmd, check=True, capture_output=True, timeout=600)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert out.returncode == 0 and "plan_check ok" in out.stdout, out.stdout[-1500:] + out.stderr[-3000:]
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr, out.stderr[-3000:]
