#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
( timeout 900 python -m pytest tests/test_phrase.py -m gpu -x -q -k "slop or span" ) > $O/slop_tests.log 2>&1
tail -3 $O/slop_tests.log
bash scripts/gpu_r3n.sh
cd /tmp
rm -rf $O/prof_slop2
( timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_slop2 -- python $R/scripts/slop_heavy.py --terms 2 --reps 5 ) > $O/prof_slop2.log 2>&1
f=$(ls -t $(find $O/prof_slop2 -name "*kernel_stats.csv") | head -1)
[ -n "$f" ] && python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'span' in r['Name']: print("  %-62s calls %3s avg %9.0f ns" % (r['Name'][:62], r['Calls'], float(r['AverageNs'])))
PY
