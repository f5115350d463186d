#!/bin/bash
# round 2, second GPU pass: grouped kernel parity + A/B, PMC child diagnosis
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd $R
( time timeout 900 python -m pytest tests/test_group.py tests/test_bm25.py tests/test_sharded.py -m gpu -q -x ) > $O/pytest_group.log 2>&1
( time timeout 900 python scripts/group_ab.py --corpus-cache /tmp/corpus ) > $O/group_ab.log 2>&1
# PMC diagnosis: a stand-alone child run exactly as bench.py would start it
cd /tmp
( time timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_try -- python $R/bench.py --pmc-child fetch --docs 10000000 --no-cpu-baseline --no-pmc --corpus-cache /tmp/corpus ) > $O/pmc_try.log 2>&1
ls -R $O/pmc_try | head -20 >> $O/pmc_try.log
cd $R
( time timeout 900 python bench.py --corpus-cache /tmp/corpus --cpu-seconds 5 ) > $O/bench2.log 2>&1
tail -3 $O/pytest_group.log
exit 0
