#!/bin/bash
# round 3: per-batch streams + reset folded into the merge -- fresh-batch pipeline at 10 M docs and on a rank-sized shard
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd $R
( time timeout 600 python -m pytest tests/test_group.py tests/test_reset.py tests/test_sharded.py tests/test_bm25.py -m gpu -q -x ) > $O/pytest_r3d.log 2>&1
for S in 1 0; do
( time SA_BATCH_STREAM=$S timeout 600 python bench.py --corpus-cache /tmp/corpus --no-cpu-baseline --no-pmc --no-phrase-legs ) > $O/bench_r3d_s$S.log 2>&1
( time SA_BATCH_STREAM=$S RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_PORT=29533 SA_BENCH_FORCE_COMM=1 timeout 300 python bench.py --no-cpu-baseline --no-pmc --no-phrase-legs --docs 1250000 --steps 50 ) > $O/bench_r3d_rank_s$S.log 2>&1
done
tail -3 $O/pytest_r3d.log
for f in $O/bench_r3d_s1.log $O/bench_r3d_s0.log $O/bench_r3d_rank_s1.log $O/bench_r3d_rank_s0.log; do grep "^{" $f | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print('$f'.split('/')[-1], j['value'], j['ms_per_step'], 'replay', j['replay']['ms_per_step'], j['replay']['fresh_over_replay'], 'kms', j['roofline']['kernel_ms'], j['parity_check'], 'pruned', j['dynamic_pruning']['ms_per_step'], j['dynamic_pruning']['same_results'])"; done
exit 0
