#!/bin/bash
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
rm -rf $O/prof_k1000
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
( timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_k1000 -- python $R/scripts/ab.py --corpus-cache /tmp/corpus --ks 1000 --qsets baseline --steps 12 ) > $O/prof_k1000.log 2>&1
grep "^{" $O/prof_k1000.log
f=$(find $O/prof_k1000 -name "*kernel_stats.csv" | head -1); grep -E "bm25|merge|Name" $f | cut -c1-150
find $O -name "*.db" -delete 2>/dev/null
exit 0
