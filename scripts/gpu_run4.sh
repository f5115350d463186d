#!/bin/bash
# GPU tests + dist path on one rank + large-k sweep
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1
tail -n 3 $O/pytest_gpu.log
bash scripts/gpu_dist1.sh
( timeout 900 python scripts/sweep.py --tiles 8192 --ks 10,100,1000 --steps 10 ) > $O/sweep_k.log 2>&1
cat $O/sweep_k.log
exit 0
