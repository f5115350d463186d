#!/bin/bash
# round 4: tiles dealt to the XCDs round-robin (SA_XCD_RANGE=0) vs a range of consecutive tiles per XCD (=1): step time and
# FETCH_SIZE per kernel, BASELINE and pairwise-distinct query sets
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
C=/tmp/corpus
cd $R
ENVS=${XCD_ENVS:-"SA_XCD_RANGE=0;SA_XCD_RANGE=1"}
( timeout 900 python scripts/ab.py --corpus-cache $C --envs "$ENVS" --ks ${XCD_KS:-10} --qsets ${XCD_QSETS:-baseline,distinct} ) > $O/ab_xcd.log 2>&1
grep "^{" $O/ab_xcd.log
cd /tmp
IFS=';' read -ra EV <<< "$ENVS"
i=0
for e in "${EV[@]}"; do
  rm -rf $O/pmc_xcd_$i
  ( timeout 300 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmc_xcd_$i -- python $R/scripts/ab.py --corpus-cache $C --ks 10 --steps 2 --qsets ${XCD_QSETS:-baseline,distinct} --envs "$e" ) > $O/pmc_xcd_$i.log 2>&1
  python - "$O/pmc_xcd_$i" "$e" <<'PY'
import csv, glob, sys, os
from collections import defaultdict
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True))
if not f:
    print("no counters", sys.argv[1]); sys.exit(0)
per = defaultdict(lambda: defaultdict(float)); name = {}
for r in csv.DictReader(open(f[-1])):
    d = int(r["Dispatch_Id"]); name[d] = r["Kernel_Name"].split("(")[0].replace("void ", "")
    per[d][r["Counter_Name"]] += float(r["Counter_Value"])
acc = defaultdict(list)
for d, c in per.items():
    if name[d].startswith("sa_k_bm25"):
        acc[name[d]].append((c.get("FETCH_SIZE", 0), c.get("TCC_HIT_sum", 0), c.get("TCC_MISS_sum", 0)))
for k, v in acc.items():
    big = sorted(v)[len(v) // 2:]            # (the 10 M-doc launches, not the tiny ones)
    fs = sum(x[0] for x in big) / len(big); h = sum(x[1] for x in big); m = sum(x[2] for x in big)
    print(sys.argv[2], k[:60], "dispatches", len(v), "fetch_MB_x2", round(fs * 1024 * 2 / 1e6, 1), "l2_hit", round(h / max(h + m, 1), 3))
PY
  i=$((i+1))
done
find $O -name "*.db" -delete 2>/dev/null
find $O -type f -size +8M -delete 2>/dev/null
exit 0
