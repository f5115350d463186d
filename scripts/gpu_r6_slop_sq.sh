#!/bin/bash
# round 6: SQ counters of the slop batch's kernel (sa_k_span_doc_fused_multi<2>: bench.py's slop_batch leg alone) -- the evidence behind
# "what bounds the slop batch" (DESIGN 3.4): issue / wait split, instruction mix, LDS conflicts
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O; rm -rf $O/prof_slopb_sq1 $O/prof_slopb_sq2 $O/prof_slopb_sq3
export TMPDIR=/tmp
cd /tmp
SQ1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
SQ2="SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
SQ3="SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE"
( timeout 200 rocprofv3 --pmc $SQ1 --kernel-trace --output-format csv -d $O/prof_slopb_sq1 -- python $R/scripts/slop_batch_prof.py slop ) > $O/prof_slopb_sq1.log 2>&1
( timeout 200 rocprofv3 --pmc $SQ2 --kernel-trace --output-format csv -d $O/prof_slopb_sq2 -- python $R/scripts/slop_batch_prof.py slop ) > $O/prof_slopb_sq2.log 2>&1
( timeout 200 rocprofv3 --pmc $SQ3 --kernel-trace --output-format csv -d $O/prof_slopb_sq3 -- python $R/scripts/slop_batch_prof.py slop ) > $O/prof_slopb_sq3.log 2>&1
SQ_PREFIXES=sa_k_span,sa_k_dense_topk,sa_k_topk python $R/scripts/sq_summary.py $O/prof_slopb_sq1 $O/prof_slopb_sq2 $O/prof_slopb_sq3 > $O/slop_batch_sq_summary.json
find $O -name "*.db" -delete 2>/dev/null
cat $O/slop_batch_sq_summary.json
exit 0
