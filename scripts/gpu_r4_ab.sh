#!/bin/bash
# round 4: quick same-box A/B of the head-group kernel + kernel stats + SQ counters ($1 = "sq" to collect counters)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
C=/tmp/corpus
cd $R
ENVS=${HG_ENVS:-"SA_HG=0;SA_HG=1"}
( timeout 600 python scripts/ab.py --corpus-cache $C --envs "$ENVS" --ks ${HG_KS:-10} --qsets ${HG_QSETS:-baseline} ) > $O/ab_hg.log 2>&1
grep -v "^+" $O/ab_hg.log
cd /tmp
rm -rf $O/prof_hg $O/prof_hg_sq1 $O/prof_hg_sq2
( timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_hg -- python $R/scripts/ab.py --corpus-cache $C --ks 10 --qsets baseline --steps 12 --envs "SA_HG=1" ) > $O/prof_hg.log 2>&1
find $O/prof_hg -name "*kernel_stats.csv" | head -1 | xargs cat | grep "bm25\|topk" | cut -c1-150
if [ "$1" = "sq" ]; then
( timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/prof_hg_sq1 -- python $R/scripts/ab.py --corpus-cache $C --ks 10 --steps 2 --envs "SA_HG=1" ) > $O/prof_hg_sq1.log 2>&1
( timeout 200 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/prof_hg_sq2 -- python $R/scripts/ab.py --corpus-cache $C --ks 10 --steps 2 --envs "SA_HG=1" ) > $O/prof_hg_sq2.log 2>&1
python $R/scripts/sq_summary.py $O/prof_hg_sq1 $O/prof_hg_sq2 > $O/hg_sq_summary.json
cat $O/hg_sq_summary.json
fi
find $O -name "*.db" -delete 2>/dev/null
find $O -type f -size +8M -delete 2>/dev/null
exit 0
