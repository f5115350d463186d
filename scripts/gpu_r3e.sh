#!/bin/bash
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd $R
( timeout 300 python scripts/host_cost.py --docs 1250000 ) > $O/host_cost.log 2>&1
( timeout 300 python scripts/host_cost.py --docs 1250000 --comm ) >> $O/host_cost.log 2>&1
( timeout 300 python scripts/host_cost.py --docs 200000 ) >> $O/host_cost.log 2>&1
( timeout 300 python scripts/host_cost.py --docs 1250000 --queries 2048 --steps 100 ) >> $O/host_cost.log 2>&1
grep "^{" $O/host_cost.log
exit 0
