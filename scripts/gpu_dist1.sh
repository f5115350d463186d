#!/bin/bash
# exercise the N>1 code path of bench.py on ONE GPU: torchrun with a single rank, both collectives
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
export SA_BENCH_FORCE_DIST=1
( time timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --warmup 5 --docs ${DOCS:-1250000} --steps 50 --no-cpu-baseline ) > $O/dist1_rccl.log 2>&1
( time timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --warmup 5 --docs ${DOCS:-1250000} --steps 50 --no-cpu-baseline --collective torch ) > $O/dist1_torch.log 2>&1
( SA_BENCH_FORCE_DIST=0 timeout 900 python bench.py --gpus 1 --warmup 5 --docs ${DOCS:-1250000} --steps 50 --no-cpu-baseline ) > $O/dist1_single.log 2>&1
tail -n 3 $O/dist1_rccl.log $O/dist1_torch.log
exit 0
