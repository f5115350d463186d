#!/bin/bash
# round 6: the staged kernel with a query's bound re-derived every 8 / 16 / 32 (default) candidates, at 10 M docs and on the rank-sized shard
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd $R
C=/tmp/corpus
L="searcharray_amd/libsearcharray_hip.so,build/libsearcharray_hip_ref15.so,build/libsearcharray_hip_ref7.so,searcharray_amd/libsearcharray_hip.so"
( timeout 900 python scripts/ab.py --corpus-cache $C --ks 1,10,32 --qsets baseline --envs "default=1" --libs $L ) 2>&1 | grep "^{" | cut -c1-420 > $O/stage_refresh.jsonl
( timeout 600 python scripts/ab.py --corpus-cache $C --docs 1250000 --ks 10 --qsets baseline --envs "default=1" --libs $L ) 2>&1 | grep "^{" | cut -c1-420 >> $O/stage_refresh.jsonl
cat $O/stage_refresh.jsonl
exit 0
