#!/bin/bash
# round 6: the fresh-batch stream (bench.py's main region) on shards of 1.25 / 2.5 / 5 / 10 M docs (what a rank of an 8 / 4 / 2 / 1-GPU run holds):
# the staged-tile route against round 5's routes (option stage = 0) -- the shard size from which the route rule takes the staged route
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
C=/tmp/corpus
cd $R
rm -f $O/shard_routes.jsonl
for D in 1250000 2500000 5000000 10000000; do
  for OPT in "stage=1" "stage=0"; do
    ( timeout 400 python bench.py --docs $D --steps 100 --warmup 10 --pipeline 8 --no-cpu-baseline --no-pmc --no-phrase-legs --corpus-cache $C --opt $OPT ) 2> $O/shard_routes.err | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l)
    print(json.dumps({'docs': j['config']['docs'], 'opt': '$OPT', 'route': j['roofline'].get('route'), 'fresh_queries_per_s': j['value'], 'ms_per_step': j['ms_per_step'], 'ms_min_max': [j['repeats']['ms_per_step_min'], j['repeats']['ms_per_step_max']], 'replay_ms_per_step': j['replay']['ms_per_step'], 'kernel_ms': j['roofline']['kernel_ms'], 'batches_in_flight': j['config']['batches_in_flight']}))
" >> $O/shard_routes.jsonl
  done
done
cat $O/shard_routes.jsonl
exit 0
