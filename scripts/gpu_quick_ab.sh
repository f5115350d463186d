#!/bin/bash
# quick same-box A/B of the exhaustive paths (short timeouts: a faulting kernel must not eat the GPU budget)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
( timeout 120 python scripts/group_ab.py --corpus-cache /tmp/corpus --ks ${KS:-10} --only ${ONLY:-0,1} --qsets ${QSETS:-baseline} --steps 10 --tiles ${TILES:-2048} ) > $O/quick_ab.log 2>&1
grep -v "^    @" $O/quick_ab.log | tail -12
