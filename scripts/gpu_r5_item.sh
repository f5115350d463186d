#!/bin/bash
# round 5: items of 16 vs 32 queries (one vs two table passes over one base) by shard size
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
rm -f $O/item_sweep.jsonl
for docs in 2500000 5000000 7500000; do
  ( timeout 300 python scripts/ab.py --ks 10 --docs $docs --qsets baseline --libs searcharray_amd/libsearcharray_hip.so --envs "SA_SPARSE=0,group_item=16;SA_SPARSE=0,group_item=32" --steps 40 ) 2>&1 | grep "^{" >> $O/item_sweep.jsonl
done
cat $O/item_sweep.jsonl
