#!/bin/bash
# round 6: SQ counters of the staged-tile kernel (issue / wait / active split, instruction mix) -- rocprofv3 --pmc, kernel trace only
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
C=/tmp/corpus
python $R/scripts/ab.py --corpus-cache $C --ks 10 --qsets baseline --envs "stage=1" --steps 3 > /dev/null 2>&1      # (corpus cache)
i=0
for SET in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_WAVES" \
           "SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf /tmp/sq_$i
  timeout 600 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d /tmp/sq_$i -- python $R/scripts/ab.py --corpus-cache $C --ks 10 --qsets baseline --envs "stage=1" --steps 5 > $O/sq_run_$i.log 2>&1
  f=$(find /tmp/sq_$i -name "*counter_collection.csv" | head -1)
  python - "$f" > $O/sq_counters_$i.txt <<'P'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "bm25_stage" not in k: continue
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"  {c:28s} mean/dispatch {sum(v)/len(v):16.1f}  (n={len(v)})")
P
  cat $O/sq_counters_$i.txt
done
exit 0
