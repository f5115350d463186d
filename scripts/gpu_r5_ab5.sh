#!/bin/bash
# round 5: A/B of two builds of the current tree (build/libsearcharray_hip_prev.so = the last commit)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
L=build/libsearcharray_hip_prev.so,searcharray_amd/libsearcharray_hip.so
( timeout 400 python scripts/ab.py --ks 10,100,1000 --qsets baseline --libs $L --envs "SA_SPARSE=0" ) 2>&1 | grep "^{" > $O/ab5.jsonl
( timeout 400 python scripts/ab.py --ks 10 --queries 2048 --steps 10 --qsets baseline --libs $L --envs "SA_SPARSE=0" ) 2>&1 | grep "^{" >> $O/ab5.jsonl
( timeout 300 python scripts/ab.py --ks 10 --docs 1250000 --qsets baseline --libs $L --envs "SA_SPARSE=0" --steps 50 ) 2>&1 | grep "^{" >> $O/ab5.jsonl
cat $O/ab5.jsonl
