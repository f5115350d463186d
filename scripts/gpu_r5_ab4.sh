#!/bin/bash
# round 5: GPU tests of the grouped path + A/B previous build vs this one
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd $R
( timeout 600 python -m pytest tests/test_group.py tests/test_bm25.py tests/test_config_10m.py -m gpu -q -x ) > $O/pytest_group_gpu.log 2>&1
tail -3 $O/pytest_group_gpu.log
bash scripts/gpu_r5_ab3.sh "SA_SPARSE=0" 10,100,1000 baseline,distinct
exit 0
