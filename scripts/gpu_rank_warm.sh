#!/bin/bash
# rank-sized shard (1.25 M docs) with the exchange forced on one rank: per-step time vs the number of warm-up tiles
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
export SA_BENCH_FORCE_COMM=1
for w in default 2 4 8; do
  if [ $w = default ]; then unset SA_GROUP_WARM; else export SA_GROUP_WARM=$w; fi
  ( timeout 200 python bench.py --gpus 1 --warmup 5 --docs 1250000 --steps 100 --no-cpu-baseline --no-pmc --corpus-cache /tmp/corpus ) > $O/rank_warm_$w.log 2>&1
  echo "warm=$w $(grep '^{' $O/rank_warm_$w.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d.get("dynamic_pruning",{}).get("ms_per_step"))')"
done
exit 0
