#!/bin/bash
# round 6: query sets of more than 256 queries on the staged-tile route (slices of 256 device rows) against the overlay route
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd $R
C=/tmp/corpus
( timeout 900 python scripts/ab.py --corpus-cache $C --queries 2048 --ks 10 --qsets baseline --envs "sparse=0,stage=0;stage=1;default=1" ) 2>/dev/null | grep "^{" > $O/wide_set_routes.jsonl
( timeout 900 python scripts/ab.py --corpus-cache $C --queries 1024 --ks 10,100 --qsets baseline --envs "sparse=0,stage=0;stage=1;default=1" ) 2>/dev/null | grep "^{" >> $O/wide_set_routes.jsonl
( timeout 900 python scripts/ab.py --corpus-cache $C --docs 1250000 --queries 2048 --ks 10 --qsets baseline --envs "sparse=0,stage=0;stage=1;default=1" ) 2>/dev/null | grep "^{" >> $O/wide_set_routes.jsonl
cat $O/wide_set_routes.jsonl | cut -c1-300
( timeout 600 python -m pytest tests/test_stage.py tests/test_config_10m.py -m gpu -q -x ) 2>&1 | tail -3
exit 0
