#!/bin/bash
# A/B of several builds of the library on ONE box: scripts/ab_libs.sh <rounds> <lib> [<lib> ...]; each round runs the
# config-2 bench (no CPU baseline, no 16384^2 extra) with each build in turn and prints frame and Jacobi times.
set -e -o pipefail
N=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export FS_BENCH_NO_CONFIG4=1
for i in $(seq 1 $N); do
  for L in "$@"; do
    cp $R/$L $R/fluidsolver_amd/libfluidsolver_hip.so
    timeout -k 10 200 python3 $R/bench.py --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-28s frame %.1f us  pressure %.2f us/launch  diffuse %.2f us/launch' % ('$L', d['ms_per_step']*1e3, d['jacobi']['us_per_launch'], d['jacobi_diffuse_velocity']['us_per_launch']))"
  done
done
