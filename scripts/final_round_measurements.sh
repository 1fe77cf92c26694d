#!/bin/bash
# The measurements of a round's final tree, one after the other on one box (run through gpurun):
#   kernel stats + FETCH/WRITE_SIZE of the default bench (profile_round.sh), PMC bytes of one frame (frame_traffic.py),
#   the bench lines of configs 2 and 3, one rank ALONE against its rows as one context (time_lone.py) for the three slab shapes.
# usage: scripts/final_round_measurements.sh <tag>
set -o pipefail
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
cd $R
bash scripts/profile_round.sh $TAG > $O/${TAG}_profile_round.log 2>&1 || exit 1
cp $(find $O/prof_$TAG -name "*kernel_stats.csv" | head -1) $O/${TAG}_kernel_stats_bench.csv
python3 scripts/pmc_traffic.py $(find $O/pmc_fetch_$TAG -name "*counter_collection.csv" | head -1) $(find $O/pmc_write_$TAG -name "*counter_collection.csv" | head -1) > $O/${TAG}_traffic.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE -f csv -d $O/ft_f_$TAG -o f -- python3 $R/scripts/frames_only.py 30 > $O/${TAG}_ft_f.log 2>&1 || exit 1
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE -f csv -d $O/ft_w_$TAG -o f -- python3 $R/scripts/frames_only.py 30 > $O/${TAG}_ft_w.log 2>&1 || exit 1
cd $R
python3 scripts/frame_traffic.py $(find $O/ft_f_$TAG -name "*counter_collection.csv" | head -1) $(find $O/ft_w_$TAG -name "*counter_collection.csv" | head -1) 30 $O/${TAG}_frame_traffic.json > $O/${TAG}_frame_traffic.txt 2>&1
rm -rf $O/prof_$TAG $O/pmc_fetch_$TAG $O/pmc_write_$TAG $O/ft_f_$TAG $O/ft_w_$TAG
cp $O/${TAG}_frame_traffic.json $R/profiles/frame_traffic.json   # so that the bench line below carries it
timeout -k 10 300 python3 bench.py --steps 200 --warmup 20 > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err || exit 1
FS_BENCH_NO_CONFIG4=1 timeout -k 10 200 python3 bench.py --workload config3 --steps 200 --warmup 20 --no-cpu-baseline > $O/${TAG}_bench_config3.json 2> $O/${TAG}_bench_config3.err || exit 1
(
echo "scripts/time_single.py / scripts/time_lone.py: ONE rank of a slab run alone on the GPU (loopback communicator: the transfers of a real"
echo "rank out of its own rows, a free wire) against the same rows as one single-GPU context; GPU time between two events."
timeout -k 10 100 python3 scripts/time_single.py 200 3072 3072 && timeout -k 10 100 python3 scripts/time_lone.py 200 3072 3072 2 0 && timeout -k 10 100 python3 scripts/time_lone.py 200 3072 3072 8 3 &&
timeout -k 10 100 python3 scripts/time_lone.py 200 3072 3072 8 3 USE_GRAPH=0 SLAB_DEFER=0 SPLIT_ROWS=0 &&
timeout -k 10 100 python3 scripts/time_single.py 60 16384 2048 && timeout -k 10 100 python3 scripts/time_lone.py 60 16384 2048 8 3 && timeout -k 10 100 python3 scripts/time_lone.py 60 16384 2048 8 0 &&
timeout -k 10 200 python3 scripts/time_single.py 30 16384 16384 &&
FS_SWEEPS=100 timeout -k 10 100 python3 scripts/time_single.py 30 8192 8192 && FS_SWEEPS=100 timeout -k 10 100 python3 scripts/time_lone.py 30 8192 8192 8 3 && FS_SWEEPS=100 timeout -k 10 100 python3 scripts/time_lone.py 30 8192 8192 2 0
) > $O/${TAG}_time_lone.log 2>&1
cat $O/${TAG}_time_lone.log
