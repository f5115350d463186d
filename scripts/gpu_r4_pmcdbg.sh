#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/pmcdbg
( time rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmcdbg -- python $R/bench.py --pmc-child fetch --docs 10000000 --vocab 100000 --queries 256 --k 10 --tile 0 --no-cpu-baseline --no-pmc --corpus-cache /tmp/corpus --phrase-docs 1000000 ) > $O/pmcdbg.log 2>&1
tail -5 $O/pmcdbg.log
f=$(find /tmp/pmcdbg -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
rows={}
for r in csv.DictReader(open(sys.argv[1])):
    k=int(r["Dispatch_Id"]); rows.setdefault(k,(r["Kernel_Name"].split("(")[0][:40], float(r["Counter_Value"])))
prev=None;cnt=0
for k in sorted(rows):
    n=rows[k][0]
    if n==prev: cnt+=1
    else:
        if prev: print(prev,cnt)
        prev=n;cnt=1
print(prev,cnt)
PY
cd $R; python -m pytest tests/test_config_10m.py -m gpu -v -x 2>&1 | tail -8
