#!/bin/bash
# Full GPU pass: smoke, all gpu tests, bench (N=1), phrase bench, rocprof stats + PMC passes.
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
rm -rf $O/prof_stats $O/prof_pmc_fetch $O/prof_pmc_write $O/prof_phrase $O/prof_slop
mkdir -p $O
export TMPDIR=/tmp
cd $R
( time timeout 600 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1
( time timeout 1800 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1
( time timeout 900 python bench.py --corpus-cache /tmp/corpus ) > $O/bench.log 2>&1
( time timeout 900 python bench.py --corpus-cache /tmp/corpus --k 1000 --no-cpu-baseline --steps 5 ) > $O/bench_k1000.log 2>&1
( time timeout 900 python bench.py --corpus-cache /tmp/corpus --k 100 --no-cpu-baseline --steps 10 ) > $O/bench_k100.log 2>&1
( time timeout 400 python scripts/imp_ab.py --corpus-cache /tmp/corpus ) > $O/imp_ab.log 2>&1
( time timeout 900 python scripts/phrase_bench.py ) > $O/phrase_bench.log 2>&1
( time timeout 900 python scripts/slop_bench.py ) > $O/slop_bench.log 2>&1
( time timeout 600 python scripts/io_bench.py ) > $O/io_bench.log 2>&1
( time timeout 600 python scripts/sim_bench.py ) > $O/sim_bench.log 2>&1
DOCS=1250000 bash scripts/gpu_dist1.sh
cd /tmp
( timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --corpus-cache /tmp/corpus ) > $O/prof_stats.log 2>&1
( timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/prof_pmc_fetch -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --corpus-cache /tmp/corpus ) > $O/prof_pmc_fetch.log 2>&1
( timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/prof_pmc_write -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --corpus-cache /tmp/corpus ) > $O/prof_pmc_write.log 2>&1
( timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_phrase -- python $R/scripts/phrase_bench.py --phrases 16 --cpu-phrases 1 ) > $O/prof_phrase.log 2>&1
( timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_slop -- python $R/scripts/slop_bench.py --phrases 16 --cpu-phrases 1 ) > $O/prof_slop.log 2>&1
find $O -name "*.db" -delete 2>/dev/null
find $O -type f -size +8M -delete 2>/dev/null
grep -E "passed|failed" $O/pytest_gpu.log
exit 0
