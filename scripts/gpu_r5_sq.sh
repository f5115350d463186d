#!/bin/bash
# round 5: SQ counters of the grouped kernel (env in $1)
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
C=/tmp/corpus
ENVS=${1:-"SA_SPARSE=0,SA_GROUP_FX=1"}
cd /tmp
rm -rf $O/prof_sq1 $O/prof_sq2
SQ1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
SQ2="SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
( timeout 200 rocprofv3 --pmc $SQ1 --kernel-trace --output-format csv -d $O/prof_sq1 -- python $R/scripts/ab.py --corpus-cache $C --ks 10 --steps 2 --envs "$ENVS" ) > $O/prof_sq1.log 2>&1
( timeout 200 rocprofv3 --pmc $SQ2 --kernel-trace --output-format csv -d $O/prof_sq2 -- python $R/scripts/ab.py --corpus-cache $C --ks 10 --steps 2 --envs "$ENVS" ) > $O/prof_sq2.log 2>&1
python $R/scripts/sq_summary.py $O/prof_sq1 $O/prof_sq2 > $O/sq_summary.json
find $O -name "*.db" -delete 2>/dev/null
find $O -name "*kernel_trace.csv" -size +4M -delete 2>/dev/null
find $O -name "*counter_collection.csv" -size +8M -delete 2>/dev/null
exit 0
