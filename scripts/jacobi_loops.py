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
    ap.add_argument("--variant", type=int, default=-1, help="tile variant of the pressure loop (-1: library default)")
    ap.add_argument("--variant-diffuse", type=int, default=-1)
    ap.add_argument("--depth", type=int, default=0, help="block depth of the pressure loop (0: library default)")
    ap.add_argument("--depth-diffuse", type=int, default=0)
    ap.add_argument("--probe", type=int, default=0, help="FS_OPT_STAGGER_MODE bits of the FS_PROBE build (FS_PROBE_LIB=1; 16: no sweeps, 32: no loads, 64: 3x the sweeps, 128: edge path everywhere): timing only, wrong fields")
    a = ap.parse_args()
    import bench
    from fluidsolver_amd import lib as L
    lib = L.load()
    if a.variant >= 0:
        L.check(lib.fs_config_set(L.OPT_TILE_VARIANT, a.variant))
    if a.variant_diffuse >= 0:
        L.check(lib.fs_config_set(L.OPT_TILE_VARIANT_DIFFUSE, a.variant_diffuse))
    if a.probe:
        L.check(lib.fs_config_set(L.OPT_STAGGER_MODE, a.probe))
    if a.depth:
        L.check(lib.fs_config_set(L.OPT_BLOCK_DEPTH_PRESSURE, a.depth))
    if a.depth_diffuse:
        L.check(lib.fs_config_set(L.OPT_BLOCK_DEPTH_DIFFUSE, a.depth_diffuse))
    s, _ = bench.make_single(a.size, a.size, (1.5e-4, 0.0, 0.5, 10.0, 0.002, a.sweeps), 0)
    pl, pspl, psw = bench.jacobi_microbench(s, a.size, a.size, sweeps=a.sweeps, reps=a.reps)
    dl, dspl, dsw = bench.diffuse_microbench(s, a.size, a.size, sweeps=a.sweeps, reps=a.reps)
    s.close()
    print(json.dumps({"size": a.size, "variant": [lib.fs_config_get(L.OPT_TILE_VARIANT), lib.fs_config_get(L.OPT_TILE_VARIANT_DIFFUSE)],
                      "depth": [lib.fs_config_get(L.OPT_BLOCK_DEPTH_PRESSURE), lib.fs_config_get(L.OPT_BLOCK_DEPTH_DIFFUSE)], "pressure": {"us_per_launch": pl * 1e3, "sweeps_per_launch": pspl, "us_per_sweep": psw * 1e3},
                      "diffuse_velocity": {"us_per_launch": dl * 1e3, "sweeps_per_launch": dspl, "us_per_sweep": dsw * 1e3}}))
