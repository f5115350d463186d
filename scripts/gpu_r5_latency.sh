#!/bin/bash
# round 5: average latency of the grouped kernel's vector-memory, LDS and scalar-memory instructions (rocprofv3's derived
# counters VmemLatency / LdsLatency / SmemLatency: in-flight levels accumulated per cycle / instructions), one pass each
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
C=/tmp/corpus
ENVS=${1:-"SA_SPARSE=0"}
cd /tmp
for c in VmemLatency LdsLatency SmemLatency; do
  rm -rf $O/prof_lat_$c
  ( timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/prof_lat_$c -- python $R/scripts/ab.py --corpus-cache $C --ks 10 --steps 2 --envs "$ENVS" ) > $O/prof_lat_$c.log 2>&1
done
python - <<PY > $O/latency_summary.json
import csv, glob, json, os
out = {}
for c in ("VmemLatency", "LdsLatency", "SmemLatency"):
    fs = sorted(glob.glob("$O/prof_lat_%s/**/*counter_collection.csv" % c, recursive=True), key=os.path.getmtime)
    if not fs:
        continue
    acc = {}
    for r in csv.DictReader(open(fs[-1])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if not k.startswith("sa_k_bm25"):
            continue
        acc.setdefault(k, []).append(float(r["Counter_Value"]))
    out[c] = {k: round(sum(v) / len(v), 1) for k, v in acc.items()}
print(json.dumps(out, indent=1))
PY
cat $O/latency_summary.json
find $O -name "*.db" -delete 2>/dev/null
find $O -name "*kernel_trace.csv" -size +4M -delete 2>/dev/null
exit 0
