#!/bin/bash
# round 6: timing experiments on the slop batch launch -- builds with a phase REMOVED (results wrong, time only): where a block's time goes
# under the launch's real occupancy (the -DSA_PROBE build's atomics distort it).  VARIANTS = names of build/libsearcharray_hip_<name>.so
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd $R
cp searcharray_amd/libsearcharray_hip.so /tmp/product.so
rm -f $O/slop_batch_exp.jsonl
for V in product $VARIANTS product; do
  if [ $V = product ]; then cp /tmp/product.so searcharray_amd/libsearcharray_hip.so; else cp build/libsearcharray_hip_$V.so searcharray_amd/libsearcharray_hip.so; fi
  timeout 300 python scripts/slop_batch_prof.py slop 2>/dev/null | grep "^{" | sed "s/^{/{\"build\": \"$V\", /" >> $O/slop_batch_exp.jsonl
done
cp /tmp/product.so searcharray_amd/libsearcharray_hip.so
cat $O/slop_batch_exp.jsonl
exit 0
