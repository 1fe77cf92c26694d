#!/bin/bash
# SQ + TCC counter passes and a kernel trace of the two Jacobi loops alone (scripts/jacobi_loops.py), one run per pass.
# usage: [JL_ARGS="--variant 8 ..."] scripts/profile_jacobi.sh <tag> <size>     outputs under gpurun_out/jac_<tag>_<size>_*/
set -e -o pipefail
TAG=${1:-x}
SIZE=${2:-3072}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
P=jac_${TAG}_${SIZE}
cd /tmp && export TMPDIR=/tmp
run() {   # name, rocprofv3 options...
    local name=$1; shift
    timeout -s KILL 150 rocprofv3 "$@" -f csv -d $OUT/${P}_$name -o j -- python3 $R/scripts/jacobi_loops.py --size $SIZE --reps 5 $JL_ARGS > $OUT/${P}_$name.log 2>&1
}
run trace --kernel-trace --stats
run sq1 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY
run sq2 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM
run fetch --pmc FETCH_SIZE
run write --pmc WRITE_SIZE
run grbm --pmc GRBM_GUI_ACTIVE
python3 $R/scripts/pmc_summary.py $OUT/${P}_pmc.json $(find $OUT/${P}_sq1 $OUT/${P}_sq2 $OUT/${P}_fetch $OUT/${P}_write $OUT/${P}_grbm -name "*counter_collection.csv") > $OUT/${P}_pmc.txt
find $OUT/${P}_trace -name "*kernel_stats.csv" -exec cp {} $OUT/${P}_kernel_stats.csv \;
