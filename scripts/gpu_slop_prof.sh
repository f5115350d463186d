#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O; rm -rf $O/prof_slop2 $O/prof_slop3
export TMPDIR=/tmp
cd /tmp
( timeout 200 python $R/scripts/slop_heavy.py --terms 2,3 --reps 3 ) > $O/slop_ce.log 2>&1
grep '^{' $O/slop_ce.log
for t in 2 3; do
( timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_slop$t -- python $R/scripts/slop_heavy.py --terms $t --reps 5 ) > $O/prof_slop$t.log 2>&1
grep "^{" $O/prof_slop$t.log
done
find $O -name "*.db" -delete 2>/dev/null
exit 0
