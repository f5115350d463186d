#!/bin/bash
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd $R
( time timeout 600 python scripts/ab.py --corpus-cache /tmp/corpus --ks 10,100 --qsets distinct,baseline --envs "SA_GROUP_MIN=2;SA_GROUP_MIN=1;SA_GROUP_MIN=1,SA_GROUP_LOOSE=0;SA_SPARSE=1" ) > $O/ab_r3j.log 2>&1
grep "^{" $O/ab_r3j.log
exit 0
