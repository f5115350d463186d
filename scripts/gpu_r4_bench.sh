#!/bin/bash
# round 4: the bench line as the driver runs it + the new gpu tests
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd $R
( time timeout 900 python bench.py --corpus-cache /tmp/corpus ) > $O/bench_r4.log 2> $O/bench_r4.err
tail -c 600 $O/bench_r4.err
python - <<'PY'
import json
l=[x for x in open('/root/repo/gpurun_out/bench_r4.log') if x.startswith('{')]
d=json.loads(l[-1])
print({k:d[k] for k in ('value','ms_per_step','repeats','scaling','parity_check')})
print('roofline', {k:d['roofline'].get(k) for k in ('kernel_ms','achieved','frac','frac_basis','traffic','traffic_frac','wasted','fetch_calibration','logical_GBps')})
print('wide', d.get('wide_batch'))
print('cpu', d.get('cpu_baseline',{}).get('value'), d.get('cpu_baseline',{}).get('parity'))
print('cfg', d['config'].get('collective_library'))
print('distinct', d.get('distinct_terms',{}).get('value'), 'pruning', d.get('dynamic_pruning',{}).get('value'))
print('phrase', d.get('phrase_batch',{}).get('value'), 'slop', d.get('slop_batch',{}).get('value'))
PY
( time timeout 1200 python -m pytest tests/test_config_10m.py tests/test_solr.py tests/test_sharded.py tests/test_search_api.py -m gpu -q -x ) 2>&1 | tail -5
exit 0
