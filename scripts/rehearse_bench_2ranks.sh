#!/bin/bash
# bench.py --gpus 2 end to end on a ONE-GPU box: one process per rank under torch.distributed.run, both ranks on GPU 0
# (FS_BENCH_DEVICE), halo rows and the CFL all-reduce through tests/mock_rccl (a test double of RCCL over shared memory).
# Not a measurement -- the ranks share the GPU --: it shows that the multi-rank code path of the final tree runs and that
# its slab self-check is bit-identical.  usage: scripts/rehearse_bench_2ranks.sh [ranks] [steps] > out.json
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
N=${1:-2}
STEPS=${2:-20}
cd $R
export FS_RCCL_LIBRARY=$R/tests/mock_rccl/libmock_rccl.so FS_BENCH_DEVICE=0 FS_BENCH_WATCHDOG_S=400 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 500 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus $N --steps $STEPS --warmup 5
