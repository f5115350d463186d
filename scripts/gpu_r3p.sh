#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
export TMPDIR=/tmp
cd $R
( timeout 900 python -m pytest tests/test_phrase.py -m gpu -x -q -k "slop or span" ) > $O/slop_tests.log 2>&1
tail -3 $O/slop_tests.log
cd /tmp
for cfg in "SA_SPAN_DOC=1"; do
echo "$cfg"
rm -rf $O/prof_slop2
( env $cfg timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_slop2 -- python $R/scripts/slop_heavy.py --terms 2 --reps 5 ) > $O/prof_slop2.log 2>&1
grep '^{' $O/prof_slop2.log | tail -2
f=$(ls -t $(find $O/prof_slop2 -name "*kernel_stats.csv") | head -1)
[ -n "$f" ] && python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'span' in r['Name']: print("  %-62s calls %3s avg %9.0f ns" % (r['Name'][:62], r['Calls'], float(r['AverageNs'])))
PY
done
rm -rf $O/sq_slop1 $O/sq_slop2
( timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/sq_slop1 -- python $R/scripts/slop_heavy.py --terms 2 --reps 3 ) > $O/sq_slop1.log 2>&1
( timeout 200 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/sq_slop2 -- python $R/scripts/slop_heavy.py --terms 2 --reps 3 ) > $O/sq_slop2.log 2>&1
SQ_PREFIXES=sa_k_span python $R/scripts/sq_summary.py $O/sq_slop1 $O/sq_slop2 > $O/sq_slop.json
cat $O/sq_slop.json
find $O -name "*.db" -delete 2>/dev/null
exit 0
