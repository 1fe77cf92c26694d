"""The oracle reproduces the committed golden vectors (tests/golden/*.npz, made by make_golden.py)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))


def test_oracle_reproduces_golden(oracle):
    from make_golden import CASES, run_case
    for name, case in CASES.items():
        gold = np.load(os.path.join(HERE, "golden", name + ".npz"))
        out = run_case(oracle, case)
        assert sorted(out) == sorted(gold.files)
        for k in out:
            assert np.array_equal(out[k], gold[k]), (name, k)


def test_golden_config1_sanity():
    g = np.load(os.path.join(HERE, "golden", "config1_256.npz"))
    vel, den = g["vel_1"], g["den_1"]
    assert vel.shape == (2, 256, 256) and den.shape == (3, 256, 256)
    # boundary velocity untouched (zero), only the red channel carries dye
    assert not vel[:, 0].any() and not vel[:, :, 0].any()
    assert den[0].max() > 10 and not den[1].any() and not den[2].any()
