#!/bin/bash
# round 6: the slop batch's scoring launch with 4 / 3 / 2 / 1 resident blocks per CU (option span_lds_pad: unused dynamic LDS per block)
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd $R
rm -f $O/slop_batch_occupancy.jsonl
for PAD in 0 14000 42000 100000; do
  SA_OPTS="span_lds_pad=$PAD" timeout 300 python scripts/slop_batch_prof.py slop 2>/dev/null | grep "^{" | sed "s/^{/{\"span_lds_pad\": $PAD, /" >> $O/slop_batch_occupancy.jsonl
done
cat $O/slop_batch_occupancy.jsonl
exit 0
