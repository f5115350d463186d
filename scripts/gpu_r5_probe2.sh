#!/bin/bash
# round 5: cycle sections inside the grouped kernel (-DSA_PROBE build), 10 M docs and the rank-sized shard
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd $R
( timeout 300 python scripts/ab.py --ks 10 --qsets baseline,distinct,hot --libs ${1:-build/libsearcharray_hip_probe.so} --envs "SA_SPARSE=0" ) > $O/probe_sections.log 2>&1
( timeout 300 python scripts/ab.py --ks 10 --docs 1250000 --qsets baseline --libs ${1:-build/libsearcharray_hip_probe.so} --envs "SA_SPARSE=0" ) >> $O/probe_sections.log 2>&1
cat $O/probe_sections.log
exit 0
