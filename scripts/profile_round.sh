#!/bin/bash
# Per-round profile of the default bench workload on the GPU box (run through gpurun):
#   1. rocprofv3 --kernel-trace --stats   -> per-kernel time
#   2. two --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, no tracing) -> HBM bytes per kernel launch
# usage: scripts/profile_round.sh <tag>      outputs under gpurun_out/prof_<tag>*/
set -e -o pipefail
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
export FS_BENCH_NO_CONFIG4=1   # profile the config-2 frame only (the 16384^2 extra would mix into the per-kernel averages)
timeout -s KILL 240 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_$TAG -o bench -- python3 $R/bench.py --steps 25 --warmup 5 --no-cpu-baseline > $OUT/prof_$TAG.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE -f csv -d $OUT/pmc_fetch_$TAG -o bench -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/pmc_fetch_$TAG.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE -f csv -d $OUT/pmc_write_$TAG -o bench -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/pmc_write_$TAG.log 2>&1
find $OUT/prof_$TAG $OUT/pmc_fetch_$TAG $OUT/pmc_write_$TAG -name "*.csv" | head -20
