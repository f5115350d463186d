#!/bin/bash
# round 3: A/B of the grouped kernel's occupancy-4 layout against the previous build, 10 M docs and a rank-sized shard
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd $R
L=searcharray_amd/libsearcharray_hip_occ4.so,searcharray_amd/libsearcharray_hip_v3.so,searcharray_amd/libsearcharray_hip.so
( time timeout 600 python -m pytest tests/test_group.py tests/test_reset.py -m gpu -q -x ) > $O/pytest_r3b.log 2>&1
( time timeout 600 python scripts/ab.py --corpus-cache /tmp/corpus --libs $L --ks 10,100,1000 --qsets baseline,distinct ) > $O/ab_r3b.log 2>&1
( time timeout 300 python scripts/ab.py --docs 1250000 --libs $L --ks 10 --qsets baseline --steps 50 ) > $O/ab_r3b_rank.log 2>&1
tail -3 $O/pytest_r3b.log
grep "^{" $O/ab_r3b.log $O/ab_r3b_rank.log
exit 0
