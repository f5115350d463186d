#!/bin/bash
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd $R
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1
( time timeout 900 python scripts/phrase_bench.py ) > $O/phrase_bench.log 2>&1
tail -n 5 $O/pytest_gpu.log
exit 0
