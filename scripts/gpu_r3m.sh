#!/bin/bash
# doc-parallel slop route: GPU parity + heaviest-query timing with / without it + per-kernel stats
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O; rm -rf $O/prof_slop2 $O/prof_slop3
export TMPDIR=/tmp
cd $R
( timeout 900 python -m pytest tests/test_phrase.py -m gpu -x -q -k "slop or span" ) > $O/slop_tests.log 2>&1
tail -3 $O/slop_tests.log
cd /tmp
echo "SA_SPAN_DOC=0"
( SA_SPAN_DOC=0 timeout 200 python $R/scripts/slop_heavy.py --terms 2,3 --reps 5 ) > $O/slop_doc0.log 2>&1
grep '^{' $O/slop_doc0.log
for cfg in "SA_SPAN_DOC_PA=8" "SA_SPAN_DOC_PA=10" "SA_SPAN_DOC_GRID=2304" "SA_SPAN_DOC_GRID=8192" "SA_SPAN_DOC_PER_BLOCK=2048" "SA_SPAN_DOC_PER_BLOCK=8192" "SA_SPAN_TRACE=2"; do
echo "$cfg"
( env $cfg timeout 200 python $R/scripts/slop_heavy.py --terms 2 --reps 5 ) > $O/slop_doc1.log 2>&1
grep '^{\|bin sizes' $O/slop_doc1.log | tail -2
done
for t in 2 3; do
( timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_slop$t -- python $R/scripts/slop_heavy.py --terms $t --reps 5 ) > $O/prof_slop$t.log 2>&1
grep "^{" $O/prof_slop$t.log
f=$(ls -t $(find $O/prof_slop$t -name "*kernel_stats.csv") | head -1)
[ -n "$f" ] && python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'span' in r['Name']: print("  %-62s calls %3s avg %9.0f ns" % (r['Name'][:62], r['Calls'], float(r['AverageNs'])))
PY
done
find $O -name "*.db" -delete 2>/dev/null
exit 0
