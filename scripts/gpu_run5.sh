#!/bin/bash
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1
tail -n 5 $O/pytest_gpu.log
( timeout 900 python scripts/phrase_bench.py ) > $O/phrase_bench.log 2>&1
tail -n 2 $O/phrase_bench.log
( timeout 900 python bench.py --steps 20 --warmup 3 --no-cpu-baseline ) > $O/bench_quick.log 2>&1
tail -n 1 $O/bench_quick.log
exit 0
