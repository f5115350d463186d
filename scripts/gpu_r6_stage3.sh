#!/bin/bash
# round 6: the staged-tile kernel with probe rows: product build vs the grouped overlay, tile sizes, and the probe build's cycles per phase
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd $R
C=/tmp/corpus
( time timeout 600 python -m pytest tests/test_stage.py -m gpu -q -x ) > $O/pytest_stage.log 2>&1
tail -3 $O/pytest_stage.log
( time timeout 900 python scripts/ab.py --corpus-cache $C --ks 10 --qsets baseline --envs "SA_SPARSE=0;stage=1;stage=1,stage_docs=512;stage=1,stage_docs=1024;stage=1,stage_docs=2048;stage=1,stage_docs=4096;stage=1,stage_probe=0" ) > $O/ab_stage3.log 2>&1
grep -v "^+" $O/ab_stage3.log | tail -12
( time timeout 900 python scripts/ab.py --corpus-cache $C --ks 10 --qsets baseline --libs build/libsearcharray_hip_probe.so --envs "stage=1;stage=1,stage_docs=2048" ) > $O/ab_stage3_probe.log 2>&1
grep -v "^+" $O/ab_stage3_probe.log | tail -12
( time timeout 900 python scripts/ab.py --corpus-cache $C --ks 100,1000 --qsets baseline --envs "SA_SPARSE=0;stage=1" ) > $O/ab_stage3_k.log 2>&1
grep -v "^+" $O/ab_stage3_k.log | tail -12
exit 0
