#!/bin/bash
# impact-stream pass: bench (default route), same-box A/B of the routes, gpu tests under the default and route 1
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
( time timeout 600 python bench.py --corpus-cache /tmp/corpus ) > $O/bench.log 2>&1
tail -n 1 $O/bench.log | cut -c1-600
( time timeout 400 python scripts/imp_ab.py --corpus-cache /tmp/corpus ) > $O/imp_ab.log 2>&1
cat $O/imp_ab.log | cut -c1-330
( time timeout 900 python -m pytest tests -m gpu -q -x ) > $O/pytest_gpu.log 2>&1
tail -n 3 $O/pytest_gpu.log
( time SA_IMPACT=1 timeout 300 python -m pytest tests/test_bm25.py tests/test_fuzz.py -m gpu -q -x ) > $O/pytest_gpu_route1.log 2>&1
tail -n 3 $O/pytest_gpu_route1.log
exit 0
