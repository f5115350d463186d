#!/bin/bash
# Round-1 first GPU pass: smoke, parity tests, bench, tuning sweep, rocprof stats + PMC.
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd $R
rocminfo | grep -E "Marketing Name|gfx" | head -4 > $O/rocminfo.txt 2>&1
nproc > $O/nproc.txt; free -g | head -2 >> $O/nproc.txt
( time timeout 600 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1
( time timeout 1200 python bench.py --steps 10 --warmup 2 --corpus-cache /tmp/corpus ) > $O/bench.log 2>&1
( time timeout 900 python scripts/sweep.py --corpus-cache /tmp/corpus ) > $O/sweep.log 2>&1
cd /tmp
( time timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_stats -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --corpus-cache /tmp/corpus ) > $O/prof_stats.log 2>&1
( time timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/prof_pmc_fetch -- python $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --corpus-cache /tmp/corpus ) > $O/prof_pmc_fetch.log 2>&1
( time timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/prof_pmc_write -- python $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --corpus-cache /tmp/corpus ) > $O/prof_pmc_write.log 2>&1
# keep only the small csv summaries
find $O -name "*.db" -delete 2>/dev/null
find $O -type f -size +20M -delete 2>/dev/null
du -sh $O
tail -3 $O/smoke.log $O/pytest_gpu.log $O/bench.log
