#!/bin/bash
# round 6: option sweeps of the staged-tile route (probe_div, stage_wgs, stage_docs) and the streaming-hint builds
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd $R
C=/tmp/corpus
( time timeout 900 python scripts/ab.py --corpus-cache $C --ks 10 --qsets baseline --envs "stage=1;stage=1,probe_div=32;stage=1,probe_div=64;stage=1,probe_div=256;stage=1,probe_div=512;stage=1,stage_wgs=1;stage=1,stage_wgs=3;stage=1,stage_docs=768;stage=1,stage_docs=512" --libs searcharray_amd/libsearcharray_hip.so,build/libsearcharray_hip_nt.so,build/libsearcharray_hip_nt2.so ) > $O/ab_sweep1.log 2>&1
grep -v "^+" $O/ab_sweep1.log | grep -E "lib" | cut -c1-330
exit 0
