# one rank ALONE (loopback) against its rows as one context, the three slab shapes (the block at the end of final_round_measurements.sh)
cd ${GRAFT_REPO_ROOT:-.}
timeout -k 10 100 python3 scripts/time_single.py 200 3072 3072 && timeout -k 10 100 python3 scripts/time_lone.py 200 3072 3072 2 0 && timeout -k 10 100 python3 scripts/time_lone.py 200 3072 3072 8 3 &&
timeout -k 10 100 python3 scripts/time_single.py 200 3072 1536 && timeout -k 10 100 python3 scripts/time_lone.py 200 3072 1536 4 0 && timeout -k 10 100 python3 scripts/time_lone.py 200 3072 1536 4 1 &&
timeout -k 10 100 python3 scripts/time_single.py 60 16384 2048 && timeout -k 10 100 python3 scripts/time_lone.py 60 16384 2048 8 3 && timeout -k 10 100 python3 scripts/time_lone.py 60 16384 2048 8 0 &&
FS_SWEEPS=100 timeout -k 10 100 python3 scripts/time_single.py 30 8192 8192 && FS_SWEEPS=100 timeout -k 10 100 python3 scripts/time_lone.py 30 8192 8192 8 3 && FS_SWEEPS=100 timeout -k 10 100 python3 scripts/time_lone.py 30 8192 8192 2 0
