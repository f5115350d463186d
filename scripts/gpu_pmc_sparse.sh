#!/bin/bash
# SQ / cache counters of the dynamic-pruning kernels (lead, rest): where do the wave cycles go?
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES SQ_WAVES" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf /tmp/pm_$tag
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pm_$tag -- python $R/bench.py --pruned --steps 2 --warmup 1 --no-cpu-baseline --corpus-cache /tmp/corpus > /tmp/pm_$tag.log 2>&1
  f=$(find /tmp/pm_$tag -name "*counter_collection.csv" | head -n 1)
  python - "$f" <<'PY'
import csv,sys
from collections import defaultdict
acc=defaultdict(list)
try:
    for r in csv.DictReader(open(sys.argv[1])):
        n=r["Kernel_Name"]
        for key in ("sa_k_sparse_lead","sa_k_sparse_rest","sa_k_sparse_score"):
            if key in n: acc[(key,r["Counter_Name"])].append(float(r["Counter_Value"]))
except Exception as e:
    print("ERR", e)
for k,v in sorted(acc.items()): print(k[0], k[1], len(v), sum(v)/len(v))
PY
done 2>&1 | tee $O/pmc_sparse.log
