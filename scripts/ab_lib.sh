#!/bin/bash
# A/B of two builds of the library on ONE box: scripts/ab_lib.sh <libA> <libB> [rounds]; each round runs the
# config-2 bench (no CPU baseline, no 16384^2 extra) with A, then with B, and prints frame and Jacobi times.
set -e -o pipefail
A=$1; B=$2; N=${3:-2}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export FS_BENCH_NO_CONFIG4=1
for i in $(seq 1 $N); do
  for L in $A $B; do
    cp $R/$L $R/fluidsolver_amd/libfluidsolver_hip.so
    timeout -k 10 200 python3 $R/bench.py --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-22s frame %.1f us  pressure %.2f us/launch  diffuse %.2f us/launch' % ('$L', d['ms_per_step']*1e3, d['jacobi']['us_per_launch'], d['jacobi_diffuse_velocity']['us_per_launch']))"
  done
done
