#!/bin/bash
# round 6: first measurement of the staged-tile route (sa_stage.hip): GPU parity tests of the route, then a same-box A/B against the
# grouped overlay kernel on the bench workload
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd $R
C=/tmp/corpus
( time timeout 900 python -m pytest tests/test_stage.py -m gpu -q -x ) > $O/pytest_stage.log 2>&1
tail -5 $O/pytest_stage.log
( time timeout 900 python scripts/ab.py --corpus-cache $C --ks 10 --qsets baseline,distinct --envs "SA_SPARSE=0;stage=1;stage=1,stage_docs=256;stage=1,stage_docs=384;stage=1,stage_wgs=1" ) > $O/ab_stage1.log 2>&1
cat $O/ab_stage1.log | grep -v "^+" | tail -20
( time timeout 600 python scripts/ab.py --corpus-cache $C --ks 100,1000 --qsets baseline --envs "SA_SPARSE=0;stage=1" ) > $O/ab_stage1_k.log 2>&1
cat $O/ab_stage1_k.log | grep -v "^+" | tail -20
exit 0
