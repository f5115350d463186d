#!/bin/bash
# round 6: the staged-tile kernel, product build vs the grouped overlay + the probe build's cycles per phase
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd $R
C=/tmp/corpus
( time timeout 600 python -m pytest tests/test_stage.py -m gpu -q -x ) > $O/pytest_stage.log 2>&1
tail -3 $O/pytest_stage.log
( time timeout 900 python scripts/ab.py --corpus-cache $C --ks 10 --qsets baseline,distinct --envs "SA_SPARSE=0;stage=1;stage=1,stage_docs=256;stage=1,stage_docs=384" ) > $O/ab_stage2.log 2>&1
grep -v "^+" $O/ab_stage2.log | tail -12
( time timeout 900 python scripts/ab.py --corpus-cache $C --ks 10 --qsets baseline,distinct --libs build/libsearcharray_hip_probe.so --envs "stage=1;stage=1,stage_docs=256" ) > $O/ab_stage2_probe.log 2>&1
grep -v "^+" $O/ab_stage2_probe.log | tail -12
exit 0
