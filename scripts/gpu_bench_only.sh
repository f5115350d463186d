#!/bin/bash
# bench legs only (profiles/pmc_traffic.json already collected): N=1 k=10/100/1000 + one-rank distributed path
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
( time timeout 900 python bench.py --corpus-cache /tmp/corpus ) > $O/bench.log 2>&1
( time timeout 900 python bench.py --corpus-cache /tmp/corpus --k 100 --no-cpu-baseline --steps 10 ) > $O/bench_k100.log 2>&1
( time timeout 900 python bench.py --corpus-cache /tmp/corpus --k 1000 --no-cpu-baseline --steps 5 ) > $O/bench_k1000.log 2>&1
DOCS=1250000 bash scripts/gpu_dist1.sh
tail -n 1 $O/bench.log | cut -c1-400
exit 0
