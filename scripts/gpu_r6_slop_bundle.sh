#!/bin/bash
# round 6: the slop batch launch with the phrases of a bundle taking turns in the launch's slots (option span_bundle; 1: a phrase's blocks
# back to back, as before), same box, same library; then the slop tests on the device
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd $R
rm -f $O/slop_batch_bundle.jsonl
for BN in ${BUNDLES:-1 8 16 32 64 128 256 1 32}; do
  SA_OPTS="span_bundle=$BN" timeout 300 python scripts/slop_batch_prof.py slop 2>/dev/null | grep "^{" | sed "s/^{/{\"span_bundle\": $BN, /" >> $O/slop_batch_bundle.jsonl
done
( timeout 900 python -m pytest tests/test_phrase.py tests/test_config_scale.py tests/test_search_api.py -m gpu -q -x 2>&1 | tail -2 ) >> $O/slop_batch_bundle.jsonl
cat $O/slop_batch_bundle.jsonl
exit 0
