#!/bin/bash
# round 5: A/B of option sets on the current library vs the previous build
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd $R
ENVS=${1:-"SA_SPARSE=0"}
( timeout 400 python scripts/ab.py --ks ${2:-10} --qsets ${3:-baseline} --libs build/libsearcharray_hip_r5a.so --envs "SA_SPARSE=0" ) > $O/ab3.log 2>&1
( timeout 400 python scripts/ab.py --ks ${2:-10} --qsets ${3:-baseline} --libs searcharray_amd/libsearcharray_hip.so --envs "$ENVS" ) >> $O/ab3.log 2>&1
( timeout 300 python scripts/ab.py --ks 10 --docs 1250000 --qsets baseline --libs build/libsearcharray_hip_r5a.so --envs "SA_SPARSE=0" --steps 50 ) >> $O/ab3.log 2>&1
( timeout 300 python scripts/ab.py --ks 10 --docs 1250000 --qsets baseline --libs searcharray_amd/libsearcharray_hip.so --envs "$ENVS" --steps 50 ) >> $O/ab3.log 2>&1
grep "^{" $O/ab3.log | cut -c1-400
exit 0
