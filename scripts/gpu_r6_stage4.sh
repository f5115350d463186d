#!/bin/bash
# round 6: the staged-tile route by k and query set (route rule), with the plan's candidate estimate traced
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd $R
C=/tmp/corpus
( time timeout 900 python scripts/ab.py --corpus-cache $C --ks 1,10,32,100,1000 --qsets baseline,distinct,hot --envs "SA_SPARSE=0;stage=1,trace=1" ) > $O/ab_stage4.log 2>&1
grep -v "^+" $O/ab_stage4.log | grep -E "sa_stage_plan|lib" | uniq | tail -60
exit 0
