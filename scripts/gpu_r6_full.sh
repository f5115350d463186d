#!/bin/bash
# round 6: smoke + the whole GPU test suite + the bench line (default run) on the final library
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd $R
( time timeout 600 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1
tail -3 $O/smoke.log
( time timeout 2400 python -m pytest tests -m gpu -q -x ) > $O/pytest_gpu.log 2>&1
tail -5 $O/pytest_gpu.log
( time timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_r06.json 2> $O/bench_r06.err
tail -c 3000 $O/bench_r06.err
python - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/bench_r06.json').read().strip().splitlines()[-1])
    print({k: d[k] for k in ('value', 'ms_per_step')}, d['roofline']['route'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['roofline'].get('traffic'), d['roofline'].get('wasted'))
    print('overlay', d['exhaustive_overlay']['value'], d['exhaustive_overlay']['roofline']['kernel_ms'], 'pruning', d['dynamic_pruning']['value'], 'replay', d['replay'], 'distinct', d.get('distinct_terms', {}).get('value'))
    print('cpu', d['cpu_baseline'], d['parity_check'])
except Exception as e:
    print('bench parse failed', e)
PY
exit 0
