#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O; rm -rf $O/prof_rank
export TMPDIR=/tmp
cd /tmp
export SA_BENCH_FORCE_COMM=1
( timeout 250 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_rank -- python $R/bench.py --gpus 1 --warmup 5 --docs 1250000 --steps 100 --no-cpu-baseline --no-pmc --corpus-cache /tmp/corpus ) > $O/prof_rank.log 2>&1
find $O -name "*.db" -delete 2>/dev/null
grep '^{' $O/prof_rank.log | cut -c1-300
exit 0
