#!/bin/bash
# tests + sweep + rocprof (csv) on the GPU box
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd $R
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1
( time timeout 900 python scripts/sweep.py --corpus-cache /tmp/corpus $SWEEP_ARGS ) > $O/sweep.log 2>&1
( time timeout 900 python bench.py --steps 10 --warmup 2 --corpus-cache /tmp/corpus $BENCH_ARGS ) > $O/bench.log 2>&1
cd /tmp
( timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --corpus-cache /tmp/corpus $BENCH_ARGS ) > $O/prof_stats.log 2>&1
( timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/prof_pmc_fetch -- python $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --corpus-cache /tmp/corpus $BENCH_ARGS ) > $O/prof_pmc_fetch.log 2>&1
( timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/prof_pmc_write -- python $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --corpus-cache /tmp/corpus $BENCH_ARGS ) > $O/prof_pmc_write.log 2>&1
find $O -name "*.db" -delete 2>/dev/null
find $O -type f -size +8M -delete 2>/dev/null
du -sh $O
tail -n 3 $O/pytest_gpu.log
exit 0
