#!/bin/bash
# quick GPU check: bench without the CPU leg, BM25 parity tests, A/B of the posting routes
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
( time timeout 300 python bench.py --corpus-cache /tmp/corpus --no-cpu-baseline --steps 10 ) > $O/bench_quick.log 2>&1
grep '^{"metric' $O/bench_quick.log | cut -c1-140
( timeout 120 python -m pytest tests/test_bm25.py tests/test_fuzz.py -m gpu -q -x ) 2>&1 | tail -n 7 | head -n 3
( timeout 300 python scripts/imp_ab.py --corpus-cache /tmp/corpus --docs 10000000,1250000 --ks 10,1000 --routes 0,1,0,1 ${IMP_AB_ARGS} ) 2>&1 | grep '^{' | grep '"sparse": "0"' | cut -c1-200
exit 0
