#!/bin/bash
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
rm -rf $O/prof_slopb
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
( timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_slopb -- python $R/scripts/slop_batch_prof.py slop ) > $O/prof_slopb.log 2>&1
grep "^{" $O/prof_slopb.log
f=$(find $O/prof_slopb -name "*kernel_stats.csv" | head -1); head -20 $f | cut -c1-160
find $O -name "*.db" -delete 2>/dev/null
find $O -type f -size +8M -delete 2>/dev/null
exit 0
