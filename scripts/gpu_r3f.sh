#!/bin/bash
# round 3: pipeline depth of the fresh-batch leg at 10 M docs and on a rank-sized shard
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd $R
( time timeout 600 python -m pytest tests/test_group.py tests/test_reset.py tests/test_sharded.py tests/test_bm25.py -m gpu -q -x ) > $O/pytest_r3f.log 2>&1
tail -3 $O/pytest_r3f.log
for P in 2 4 8; do
( SA_X=1 timeout 600 python bench.py --corpus-cache /tmp/corpus --no-cpu-baseline --no-pmc --no-phrase-legs --pipeline $P --steps 40 ) > $O/bench_r3f_p$P.log 2>&1
( RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_PORT=29533 SA_BENCH_FORCE_COMM=1 timeout 300 python bench.py --no-cpu-baseline --no-pmc --no-phrase-legs --docs 1250000 --steps 100 --pipeline $P ) > $O/bench_r3f_rank_p$P.log 2>&1
( timeout 300 python bench.py --no-cpu-baseline --no-pmc --no-phrase-legs --docs 1250000 --steps 100 --pipeline $P ) > $O/bench_r3f_rank_nocomm_p$P.log 2>&1
done
for f in $O/bench_r3f_*.log; do grep "^{" $f | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print('$f'.split('/')[-1], j['value'], j['ms_per_step'], 'replay', j['replay']['ms_per_step'], j['replay']['fresh_over_replay'], 'kms', j['roofline']['kernel_ms'], j['parity_check'][:40], 'pruned', j['dynamic_pruning']['ms_per_step'], j['dynamic_pruning']['same_results'])"; done
exit 0
