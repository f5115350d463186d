#!/bin/bash
# rank-sized shard (1.25 M docs) with the exchange forced on one rank: per-step time vs tile size
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
export SA_BENCH_FORCE_COMM=1
for t in 1024 2048 4096; do
  ( timeout 200 python bench.py --gpus 1 --warmup 5 --docs 1250000 --steps 100 --tile $t --no-cpu-baseline --no-pmc --corpus-cache /tmp/corpus ) > $O/rank_tile_$t.log 2>&1
  echo "tile=$t $(grep '^{' $O/rank_tile_$t.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d.get("dynamic_pruning",{}).get("ms_per_step"))')"
done
exit 0
