#!/bin/bash
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd $R
rm -f $O/bench_r3g_*.log
for P in 4 6 8; do
( timeout 300 python bench.py --no-cpu-baseline --no-pmc --no-phrase-legs --docs 1250000 --steps 100 --pipeline $P ) > $O/bench_r3g_rank_p$P.log 2>&1
( RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_PORT=29533 SA_BENCH_FORCE_COMM=1 timeout 300 python bench.py --no-cpu-baseline --no-pmc --no-phrase-legs --docs 1250000 --steps 100 --pipeline $P ) > $O/bench_r3g_rankcomm_p$P.log 2>&1
( timeout 600 python bench.py --corpus-cache /tmp/corpus --no-cpu-baseline --no-pmc --no-phrase-legs --pipeline $P --steps 40 ) > $O/bench_r3g_10m_p$P.log 2>&1
done
for f in $O/bench_r3g_*.log; do grep "^{" $f | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print('$f'.split('/')[-1], j['value'], j['ms_per_step'], 'replay', j['replay']['ms_per_step'], j['replay']['fresh_over_replay'], j['parity_check'][:30])"; done
exit 0
