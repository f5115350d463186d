#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
export TMPDIR=/tmp
cd $R
( timeout 900 python -m pytest tests/test_phrase.py -m gpu -x -q -k "slop or span" ) > $O/slop_tests.log 2>&1
tail -1 $O/slop_tests.log
python scripts/slop_routes.py 2>&1 | grep "^{" | head -17 | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print(r['phrase'], r['doc_route_ms'], r['general_ms'])
"
