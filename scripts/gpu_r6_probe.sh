#!/bin/bash
# round 6: the staged-tile kernel's cycles per phase (probe build)
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd $R
C=/tmp/corpus
( time timeout 900 python scripts/ab.py --corpus-cache $C --ks 10 --qsets baseline --libs build/libsearcharray_hip_probe.so --envs "${1:-stage=1}" ) > $O/ab_probe.log 2>&1
grep -v "^+" $O/ab_probe.log | tail -12
exit 0
