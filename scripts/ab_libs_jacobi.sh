#!/bin/bash
# A/B of several builds of the library on ONE box, the two Jacobi loops alone: scripts/ab_libs_jacobi.sh <size> <rounds> <lib> [<lib> ...]
set -e -o pipefail
S=$1; N=$2; shift; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for i in $(seq 1 $N); do
  for L in "$@"; do
    cp $R/$L $R/fluidsolver_amd/libfluidsolver_hip.so
    timeout -k 10 200 python3 $R/scripts/jacobi_loops.py --size $S --reps 10 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-28s %d^2  pressure %.1f us/launch  diffuse %.1f us/launch' % ('$L', d['size'], d['pressure']['us_per_launch'], d['diffuse_velocity']['us_per_launch']))"
  done
done
