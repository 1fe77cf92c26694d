#!/bin/bash
# PMC bytes of one config-2 frame on the current tree -> profiles/frame_traffic.json (the part of final_round_measurements.sh that
# bench.py's frame_traffic_* fields depend on; run through gpurun after any change to the kernel sources).  usage: frame_traffic_only.sh <tag>
set -o pipefail
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE -f csv -d $O/ft_f_$TAG -o f -- python3 $R/scripts/frames_only.py 30 > $O/${TAG}_ft_f.log 2>&1 || exit 1
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE -f csv -d $O/ft_w_$TAG -o f -- python3 $R/scripts/frames_only.py 30 > $O/${TAG}_ft_w.log 2>&1 || exit 1
cd $R
python3 scripts/frame_traffic.py $(find $O/ft_f_$TAG -name "*counter_collection.csv" | head -1) $(find $O/ft_w_$TAG -name "*counter_collection.csv" | head -1) 30 $O/${TAG}_frame_traffic.json > $O/${TAG}_frame_traffic.txt 2>&1
rm -rf $O/ft_f_$TAG $O/ft_w_$TAG
cat $O/${TAG}_frame_traffic.txt | tail -12
