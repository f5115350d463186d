#!/bin/bash
# the slop records of scripts/gpu_full_r2.sh alone (after a change to sa_spans.hip)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O; rm -rf $O/prof_slop
export TMPDIR=/tmp
cd $R
( time timeout 300 python scripts/slop_bench.py ) > $O/slop_bench.log 2>&1
cd /tmp
( timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_slop -- python $R/scripts/slop_bench.py --phrases 16 --cpu-phrases 1 ) > $O/prof_slop.log 2>&1
bash $R/scripts/gpu_slop_prof.sh > $O/slop_heavy.log 2>&1
bash $R/scripts/gpu_slop_pmc.sh > $O/slop_pmc.log 2>&1
find $O -name "*.db" -delete 2>/dev/null
grep "^{" $O/slop_bench.log | cut -c1-400
exit 0
