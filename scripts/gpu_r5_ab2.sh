#!/bin/bash
# round 5: same-box A/B of the grouped kernel, previous build vs this one (10 M docs, rank-sized shard)
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd $R
L=${1:-build/libsearcharray_hip_r5a.so,searcharray_amd/libsearcharray_hip.so}
( timeout 400 python scripts/ab.py --ks ${2:-10,100,1000} --qsets baseline,distinct --libs $L --envs "SA_SPARSE=0" ) > $O/ab2.log 2>&1
( timeout 300 python scripts/ab.py --ks 10 --docs 1250000 --qsets baseline --libs $L --envs "SA_SPARSE=0" --steps 50 ) >> $O/ab2.log 2>&1
grep "^{" $O/ab2.log | cut -c1-400
exit 0
