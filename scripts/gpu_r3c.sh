#!/bin/bash
# round 3: SQ counters of the scoring kernels of the current build (two passes of 8 counters)
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
rm -rf $O/prof_sq1 $O/prof_sq2
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
( timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/prof_sq1 -- python $R/scripts/ab.py --corpus-cache /tmp/corpus --ks 10 --steps 2 ) > $O/prof_sq1.log 2>&1
( timeout 200 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/prof_sq2 -- python $R/scripts/ab.py --corpus-cache /tmp/corpus --ks 10 --steps 2 ) > $O/prof_sq2.log 2>&1
python $R/scripts/sq_summary.py $O/prof_sq1 $O/prof_sq2 > $O/sq_summary.json
cat $O/sq_summary.json
find $O -name "*.db" -delete 2>/dev/null
find $O -type f -size +8M -delete 2>/dev/null
exit 0
