#!/bin/bash
# round 3, first GPU pass: fresh-batch path (sa_batch_reset) on hardware -- parity tests, the bench line with its
# fresh / replay legs and the new phrase / slop legs
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd $R
( time timeout 600 python -m pytest tests/test_reset.py tests/test_group.py tests/test_sharded.py -m gpu -q -x ) > $O/pytest_r3a.log 2>&1
( time timeout 900 python bench.py --corpus-cache /tmp/corpus ) > $O/bench_r3a.log 2>&1
tail -3 $O/pytest_r3a.log
tail -c 1500 $O/bench_r3a.log
exit 0
