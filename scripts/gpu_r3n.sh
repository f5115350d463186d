#!/bin/bash
# doc-parallel slop route: SQ counters of its kernels + block-size sweep
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O; rm -rf $O/sq_slop1 $O/sq_slop2
export TMPDIR=/tmp
cd /tmp
for cfg in "SA_SPAN_DOC_GRID=8192" "SA_SPAN_DOC_GRID=16384" "SA_SPAN_DOC_GRID=32768"; do
echo "$cfg"
( env $cfg timeout 200 python $R/scripts/slop_heavy.py --terms 2 --reps 5 ) > $O/slop_doc1.log 2>&1
grep '^{\|doc route:' $O/slop_doc1.log | tail -2
done
( timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/sq_slop1 -- python $R/scripts/slop_heavy.py --terms 2 --reps 3 ) > $O/sq_slop1.log 2>&1
( timeout 200 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/sq_slop2 -- python $R/scripts/slop_heavy.py --terms 2 --reps 3 ) > $O/sq_slop2.log 2>&1
SQ_PREFIXES=sa_k_span python $R/scripts/sq_summary.py $O/sq_slop1 $O/sq_slop2 > $O/sq_slop.json
cat $O/sq_slop.json
find $O -name "*.db" -delete 2>/dev/null
exit 0
