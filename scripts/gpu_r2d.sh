#!/bin/bash
# full GPU pass: all gpu tests, smoke, bench (with its own PMC children and the reference CPU baseline), comm path
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd $R
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1
( time timeout 1200 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1
( time timeout 600 python bench.py --corpus-cache /tmp/corpus ) > $O/bench.log 2>&1
( time SA_BENCH_FORCE_COMM=1 timeout 300 python bench.py --corpus-cache /tmp/corpus --no-cpu-baseline --no-pmc ) > $O/bench_comm1.log 2>&1
tail -5 $O/pytest_gpu.log
exit 0
