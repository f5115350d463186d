#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
export TMPDIR=/tmp
cd $R
bash scripts/gpu_slop_pmc.sh > /dev/null 2>&1
python scripts/slop_pmc.py $O > $O/slop_pmc.json
python - <<'PY'
import json
d=json.load(open("/root/repo/gpurun_out/slop_pmc.json"))
for k,v in d.items():
    print(k, {x:v[x] for x in v if x!="kernels"}, list(v["kernels"].keys()))
PY
( timeout 300 python scripts/slop_bench.py ) > $O/slop_bench.log 2>&1
grep "^{" $O/slop_bench.log | tail -1 | cut -c1-1500
