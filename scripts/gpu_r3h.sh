#!/bin/bash
# round 3: slop phrases of a batch in shared launches -- parity tests and the phrase / slop legs of the bench
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd $R
( time timeout 900 python -m pytest tests/test_phrase.py tests/test_reset.py tests/test_setops.py tests/test_sharded.py -m gpu -q -x ) > $O/pytest_r3h.log 2>&1
tail -3 $O/pytest_r3h.log
( timeout 600 python scripts/slop_bench.py ) > $O/slop_bench_r3h.log 2>&1
( SA_SPAN_MULTI=0 timeout 600 python scripts/slop_bench.py ) > $O/slop_bench_r3h_single.log 2>&1
grep "^{" $O/slop_bench_r3h.log $O/slop_bench_r3h_single.log
exit 0
