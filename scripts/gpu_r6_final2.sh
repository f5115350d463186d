#!/bin/bash
# round 6, second final pass (after the slop batch changes): smoke, all gpu tests, the bench line, phrase / slop benches, rocprofv3 kernel
# stats of the bench and of the phrase / slop legs.  The BM25 sweeps of scripts/gpu_full_r6.sh are not repeated (those kernels did not change).
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
for d in prof_bench prof_slopb prof_slop prof_phrase; do rm -rf $O/$d; done
mkdir -p $O
export TMPDIR=/tmp
C=/tmp/corpus
cd $R
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1
( time timeout 1800 python -m pytest tests -m gpu -q -rxX ) > $O/pytest_gpu.log 2>&1
( time timeout 900 python bench.py --corpus-cache $C ) > $O/bench.log 2> $O/bench.err
( time timeout 300 python scripts/phrase_bench.py ) > $O/phrase_bench.log 2>&1
( time timeout 300 python scripts/slop_bench.py ) > $O/slop_bench.log 2>&1
( time timeout 300 python scripts/fuzz.py --seeds 40 ) > $O/fuzz.log 2>&1
cd /tmp
( timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python $R/bench.py --steps 10 --warmup 2 --repeats 1 --no-cpu-baseline --no-pmc --corpus-cache $C ) > $O/prof_bench.log 2>&1
( timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_slopb -- python $R/scripts/slop_batch_prof.py ) > $O/prof_slopb.log 2>&1
( timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_slop -- python $R/scripts/slop_bench.py --phrases 16 --cpu-phrases 1 ) > $O/prof_slop.log 2>&1
find $O -name "*.db" -delete 2>/dev/null
find $O -name "*hip_api_trace.csv" -delete 2>/dev/null
find $O -name "*kernel_trace.csv" -size +4M -delete 2>/dev/null
find $O -type f -size +8M -delete 2>/dev/null
grep -E "passed|failed" $O/pytest_gpu.log
tail -2 $O/fuzz.log
exit 0
