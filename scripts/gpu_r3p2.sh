#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
for cfg in "SA_SPAN_DOC_DBG=0" "SA_SPAN_DOC_DBG=1"; do
echo "$cfg"
( env $cfg timeout 150 python $R/scripts/slop_heavy.py --terms 2 --reps 5 ) 2>&1 | grep "^{"
done
