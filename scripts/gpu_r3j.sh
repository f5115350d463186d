#!/bin/bash
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd $R
( timeout 600 python -m pytest tests/test_group.py tests/test_reset.py tests/test_config_scale.py -m gpu -q -x ) > $O/pytest_r3j.log 2>&1
tail -2 $O/pytest_r3j.log
( timeout 600 python scripts/ab.py --corpus-cache /tmp/corpus --ks 10,100,1000 --qsets baseline --envs "SA_GROUP_DENSE=1;SA_GROUP_DENSE=0" ) > $O/ab_r3j.log 2>&1
( timeout 600 python scripts/ab.py --docs 1250000 --steps 50 --ks 10 --qsets baseline --envs "SA_GROUP_DENSE=1;SA_GROUP_DENSE=0" ) >> $O/ab_r3j.log 2>&1
grep "^{" $O/ab_r3j.log
exit 0
