#!/bin/bash
# round 6: batches in flight of the main region (bench.py --pipeline) at the driver's --steps 20 --warmup 5
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
C=/tmp/corpus
cd $R
rm -f $O/pipeline_sweep.jsonl
for P in 4 6 8 12 16; do
  for S in 20 100; do
    ( timeout 400 python bench.py --steps $S --warmup 5 --pipeline $P --no-cpu-baseline --no-pmc --no-phrase-legs --corpus-cache $C ) 2> $O/pipe.err | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l)
    print(json.dumps({'docs': j['config']['docs'], 'steps': j['steps'], 'route': j['roofline'].get('route'), 'fresh_queries_per_s': j['value'], 'ms_per_step': j['ms_per_step'], 'ms_min_max': [j['repeats']['ms_per_step_min'], j['repeats']['ms_per_step_max']], 'kernel_ms': j['roofline']['kernel_ms'], 'batches_in_flight': j['config']['batches_in_flight'], 'latency_ms': j.get('fresh_batch_latency_ms')}))
" >> $O/pipeline_sweep.jsonl
  done
done
cat $O/pipeline_sweep.jsonl
exit 0
