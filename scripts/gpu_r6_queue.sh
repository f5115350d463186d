#!/bin/bash
# round 6: the fresh-batch stream fed by this process (ring: sa_batch_step + fetch per step from Python) against the library's queue
# (sa_queue_*: a worker thread of the library steps the ring), at 10 M docs and on the rank-sized shard with / without the exchange
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
C=/tmp/corpus
cd $R
rm -f $O/queue_driver.jsonl
run() {
  python bench.py "$@" --no-cpu-baseline --no-pmc --no-phrase-legs --corpus-cache $C 2>/dev/null | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l)
    print(json.dumps({'docs': j['config']['docs'], 'driver': j['config'].get('driver'), 'comm': j['config'].get('collective'), 'batches_in_flight': j['config']['batches_in_flight'], 'steps': j['steps'], 'fresh_queries_per_s': j['value'], 'ms_per_step': j['ms_per_step'], 'ms_min_max': [j['repeats']['ms_per_step_min'], j['repeats']['ms_per_step_max']], 'kernel_ms': j['roofline']['kernel_ms'], 'route': j['roofline'].get('route'), 'parity': j.get('parity_check'), 'fresh_equals_replay': j['replay'].get('fresh_equals_replay')}))
" >> $O/queue_driver.jsonl
}
for D in ring queue ring queue; do
  run --driver $D
  run --driver $D --docs 1250000 --steps 200 --pipeline 8
  RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_PORT=29533 SA_BENCH_FORCE_COMM=1 run --driver $D --docs 1250000 --steps 200 --pipeline 8
done
cat $O/queue_driver.jsonl
exit 0
