#!/bin/bash
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd $R
( time timeout 600 python scripts/ab.py --corpus-cache /tmp/corpus --ks 10,1000 --qsets baseline --envs "SA_GROUP_WARM=16;SA_GROUP_WARM=8;SA_GROUP_WARM=4;SA_GROUP_WARM=32;SA_GROUP_WARM=64;SA_GROUP_WARM=128" ) > $O/ab_r3j.log 2>&1
( time timeout 600 python scripts/ab.py --docs 1250000 --steps 50 --ks 10 --qsets baseline --envs "SA_GROUP_WARM=16;SA_GROUP_WARM=8;SA_GROUP_WARM=4;SA_GROUP_WARM=2" ) >> $O/ab_r3j.log 2>&1
grep "^{" $O/ab_r3j.log
exit 0
