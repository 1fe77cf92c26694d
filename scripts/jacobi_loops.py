"""The two Jacobi loops alone, as update() runs them (pressure: clear folded in; diffuse: both velocity planes), on
U(-1,1) data -- the program the PMC / kernel-trace passes of scripts/profile_jacobi.sh are collected on.
usage: python3 scripts/jacobi_loops.py [--size 3072] [--reps 10] [--sweeps 40]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=3072)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--sweeps", type=int, default=40)
    a = ap.parse_args()
    import bench
    s, _ = bench.make_single(a.size, a.size, (1.5e-4, 0.0, 0.5, 10.0, 0.002, a.sweeps), 0)
    pl, pspl, psw = bench.jacobi_microbench(s, a.size, a.size, sweeps=a.sweeps, reps=a.reps)
    dl, dspl, dsw = bench.diffuse_microbench(s, a.size, a.size, sweeps=a.sweeps, reps=a.reps)
    s.close()
    print(json.dumps({"size": a.size, "pressure": {"us_per_launch": pl * 1e3, "sweeps_per_launch": pspl, "us_per_sweep": psw * 1e3},
                      "diffuse_velocity": {"us_per_launch": dl * 1e3, "sweeps_per_launch": dspl, "us_per_sweep": dsw * 1e3}}))
