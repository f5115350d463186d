#!/bin/bash
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd $R
( RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_PORT=29533 SA_BENCH_FORCE_COMM=1 timeout 300 python bench.py --no-cpu-baseline --no-pmc --docs 1250000 --steps 40 --batch-mult 8 ) > $O/bench_r3k_weak.log 2>&1
( RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_PORT=29533 SA_BENCH_FORCE_COMM=1 timeout 300 python bench.py --no-cpu-baseline --no-pmc --docs 1250000 --steps 100 ) > $O/bench_r3k_strong.log 2>&1
for f in $O/bench_r3k_weak.log $O/bench_r3k_strong.log; do grep "^{" $f | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['scaling'], j['config']['queries_per_step'], j['parity_check'][:40], j.get('fixed_batch'), j.get('scaled_batch'))"; tail -3 $f | cut -c1-300; done
exit 0
