#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O; rm -rf $O/prof_q $O/prof_q_sq $O/prof_q_sq2
export TMPDIR=/tmp
cd /tmp
( timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_q -- python $R/scripts/group_ab.py --corpus-cache /tmp/corpus --ks ${KS:-10} --only 1 --qsets ${QSETS:-baseline} --steps 5 ) > $O/prof_q.log 2>&1
( timeout 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/prof_q_sq -- python $R/scripts/group_ab.py --corpus-cache /tmp/corpus --ks ${KS:-10} --only 1 --qsets ${QSETS:-baseline} --steps 2 ) > $O/prof_q_sq.log 2>&1
( timeout 120 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/prof_q_sq2 -- python $R/scripts/group_ab.py --corpus-cache /tmp/corpus --ks ${KS:-10} --only 1 --qsets ${QSETS:-baseline} --steps 2 ) > $O/prof_q_sq2.log 2>&1
find $O -name "*.db" -delete 2>/dev/null
grep "^{" $O/prof_q.log
exit 0
