#!/bin/bash
# round 5: GPU tests + smoke + a short A/B of this library against round 4's on the same box
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd $R
C=/tmp/corpus
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1
( time timeout 1800 python -m pytest tests -m gpu -q -x ) > $O/pytest_gpu.log 2>&1
( time timeout 600 python scripts/ab.py --corpus-cache $C --ks 10,100,1000 --qsets baseline,distinct --libs build/libsearcharray_hip_r04.so,searcharray_amd/libsearcharray_hip.so --envs "SA_SPARSE=0" ) > $O/ab_r5.log 2>&1
exit 0
