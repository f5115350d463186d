#!/bin/bash
# grouped kernel: per-kernel times and SQ counters
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rm -rf $O/prof_grp $O/prof_grp_sq $O/prof_grp_sq2
( timeout 90 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_grp -- python $R/scripts/group_ab.py --corpus-cache /tmp/corpus --ks 10 --only 1 --qsets baseline --steps 5 --tiles 2048 ) > $O/prof_grp.log 2>&1
( timeout 90 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/prof_grp_sq -- python $R/scripts/group_ab.py --corpus-cache /tmp/corpus --ks 10 --only 1 --qsets baseline --steps 2 ) > $O/prof_grp_sq.log 2>&1
( timeout 90 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/prof_grp_sq2 -- python $R/scripts/group_ab.py --corpus-cache /tmp/corpus --ks 10 --only 1 --qsets baseline --steps 2 ) > $O/prof_grp_sq2.log 2>&1
cd $R
find $O -name "*.db" -delete 2>/dev/null
find $O -type f -size +8M -delete 2>/dev/null
exit 0
