#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
export TMPDIR=/tmp
cd $R
( timeout 900 python -m pytest tests/test_phrase.py -m gpu -x -q -k "slop or span" ) > $O/phrase_tests.log 2>&1
tail -2 $O/phrase_tests.log
bash scripts/gpu_slop_pmc.sh
python scripts/slop_pmc.py $O
( timeout 300 python scripts/slop_bench.py ) > $O/slop_bench.log 2>&1
grep "^{" $O/slop_bench.log | tail -1 | cut -c1-1500
