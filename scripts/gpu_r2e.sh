#!/bin/bash
# slop / phrase benches + their gpu tests + PMC pass for the span kernels
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd $R
( time timeout 600 python -m pytest tests/test_phrase.py tests/test_fuzz.py tests/test_config_scale.py tests/test_bm25.py -m gpu -q ) > $O/pytest_slop.log 2>&1
( time timeout 300 python scripts/slop_bench.py ) > $O/slop_bench.log 2>&1
( time SA_SPAN_DOCDIR=0 timeout 300 python scripts/slop_bench.py ) > $O/slop_bench_nodd.log 2>&1
( time timeout 300 python scripts/phrase_bench.py ) > $O/phrase_bench.log 2>&1
cd /tmp
rm -rf $O/prof_slop $O/prof_slop_pmc
( timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_slop -- python $R/scripts/slop_bench.py --phrases 16 --cpu-phrases 1 ) > $O/prof_slop.log 2>&1
( timeout 200 rocprofv3 --pmc FETCH_SIZE SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/prof_slop_pmc -- python $R/scripts/slop_bench.py --phrases 4 --cpu-phrases 1 ) > $O/prof_slop_pmc.log 2>&1
find $O -name "*.db" -delete 2>/dev/null
tail -3 $O/pytest_slop.log
exit 0
