"""Generates the golden vectors in this directory FROM THE CPU ORACLE of this repo.

The reference ships no fixtures and cannot run here (Metal/Swift only), so these vectors are
"parity unpinned" with respect to on-device Metal output: they pin the oracle against regressions
and give the GPU box a committed expected output.  Run from the repo root:

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle_binding import DENSITY, PRESSURE, VELOCITY, Oracle, Params  # noqa: E402

from fluidsolver_amd.script import splat  # noqa: E402

CASES = {
    # BASELINE.json config 1 / SURVEY 8(d): 256^2, one splat at (128,128), force (24,0), colour (100,0,0),
    # radius 5, dt 0.5, one update() with N=20, nu=1.5e-4, no vorticity, fade 0.002
    "config1_256": dict(W=256, H=256, frames=1, params=(1.5e-4, 0.0, 0.5, 0.0, 0.002, 20), radius=5.0,
                        fixed_splat=((128.0, 128.0), (24.0, 0.0), (100.0, 0.0, 0.0)), save_at=(1,)),
    # config-2 physics (N=40, vorticity 10) on a small grid, scripted Lissajous splats
    "script_64": dict(W=64, H=64, frames=100, params=(1.5e-4, 0.0, 0.5, 10.0, 0.002, 40), radius=5.0,
                      save_at=(1, 10, 100)),
    # "Smoke" preset (dye diffusion on), non-square grid whose width is not a multiple of 4
    "smoke_50x38": dict(W=50, H=38, frames=6, params=(0.00005, 0.00001, 0.4, 10.0, 0.008, 15), radius=4.0,
                        save_at=(6,)),
}


def run_case(o, c):
    s = o.solver(c["W"], c["H"])
    p = Params(*c["params"])
    out = {}
    for f in range(c["frames"]):
        if "fixed_splat" in c:
            if f == 0:
                pos, force, col = c["fixed_splat"]
                s.add_force(pos, force, col, c["radius"], p.deltaTime)
        else:
            pos, force, col = splat(c["W"], c["H"], f)
            s.add_force(pos, force, col, c["radius"], p.deltaTime)
        s.update(p)
        if f + 1 in c["save_at"]:
            out["vel_%d" % (f + 1)] = s.field(VELOCITY)
            out["den_%d" % (f + 1)] = s.field(DENSITY)
            out["prs_%d" % (f + 1)] = s.field(PRESSURE)
    return out


if __name__ == "__main__":
    o = Oracle()
    for name, c in CASES.items():
        out = run_case(o, c)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, {k: (v.shape, float(np.abs(v).max())) for k, v in out.items()})
