#!/bin/bash
# round 5: the grouped kernel at 4 / 3 / 2 / 1 resident waves per SIMD (unused dynamic LDS in the -DSA_PROBE build)
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd $R
rm -f $O/occupancy.log
for pad in 0 3200 10240 30720; do
  ( SA_PROBE_LDS_PAD=$pad timeout 300 python scripts/ab.py --ks 10 --qsets baseline --libs build/libsearcharray_hip_probe.so --envs "SA_SPARSE=0" ) 2>&1 | grep "^{" | sed "s/^{/{\"lds_pad\": $pad, /" >> $O/occupancy.log
done
cat $O/occupancy.log
exit 0
