#!/bin/bash
# round 6: A/B of library builds on the BASELINE batch (scripts/ab.py); LIBS / ENVS / KS / QSETS from the environment
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd $R
C=/tmp/corpus
TAG=${TAG:-ab}
( time timeout 1200 python scripts/ab.py --corpus-cache $C --ks ${KS:-10} --qsets ${QSETS:-baseline} --envs "${ENVS:-stage=1}" --libs ${LIBS:-searcharray_amd/libsearcharray_hip.so} ) > $O/ab_$TAG.log 2>&1
grep -v "^+" $O/ab_$TAG.log | grep -E "lib" | cut -c1-400
exit 0
