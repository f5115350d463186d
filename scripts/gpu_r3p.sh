#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
export TMPDIR=/tmp
cd $R
( timeout 900 python -m pytest tests/test_phrase.py tests/test_config_scale.py -m gpu -x -q ) > $O/slop_tests.log 2>&1
tail -1 $O/slop_tests.log
( SA_SPAN_DOC=0 timeout 900 python -m pytest tests/test_phrase.py -m gpu -x -q -k "slop or span" ) > $O/slop_tests.log 2>&1
tail -1 $O/slop_tests.log
python scripts/slop_routes.py 2>&1 | grep "^{" | head -17 | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print(r['phrase'], r['doc_route_ms'], r['general_ms'])
"
python scripts/slop_heavy.py --terms 2,3 --reps 4 | grep "^{"
python bench.py --no-cpu-baseline --no-pmc --steps 20 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('value', d['value'], 'slop_batch', d.get('slop_batch',{}).get('value'), d.get('slop_batch',{}).get('ms_per_batch'), 'phrase_batch', d.get('phrase_batch',{}).get('value'))
"
