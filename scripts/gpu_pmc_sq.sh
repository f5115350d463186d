#!/bin/bash
# SQ-level counters of the exhaustive tile kernel: where do the wave cycles go?
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf /tmp/pm_$tag
  SA_SPARSE=0 timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pm_$tag -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --corpus-cache /tmp/corpus > /dev/null 2>&1
  f=$(find /tmp/pm_$tag -name "*counter_collection.csv" | head -n 1)
  python - "$f" <<'PY'
import csv,sys
from collections import defaultdict
acc=defaultdict(list)
try:
    for r in csv.DictReader(open(sys.argv[1])):
        if "sa_k_bm25_tiles<" in r["Kernel_Name"] and "list" not in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
except Exception as e:
    print("ERR", e)
for k,v in acc.items(): print(k, len(v), sum(v)/len(v))
PY
done
