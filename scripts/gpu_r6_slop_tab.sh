#!/bin/bash
# round 6: the slop batch launch with two / four of a block's waves holding span tables (option span_tab_waves; unset: the launch rule),
# same box, same library; the slop tests on the device; the single-phrase and light-batch numbers of slop_bench.py
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd $R
rm -f $O/slop_batch_tab_waves.jsonl
for TW in default 4 2 default; do
  if [ $TW = default ]; then OPTS="trace=0"; else OPTS="span_tab_waves=$TW"; fi
  SA_OPTS="$OPTS" timeout 300 python scripts/slop_batch_prof.py slop 2>/dev/null | grep "^{" | sed "s/^{/{\"span_tab_waves\": \"$TW\", /" >> $O/slop_batch_tab_waves.jsonl
done
for TW in default 2; do
  if [ $TW = default ]; then OPTS="trace=0"; else OPTS="span_tab_waves=$TW"; fi
  SA_OPTS="$OPTS" timeout 300 python scripts/slop_bench.py 2>/dev/null | grep "^{" | cut -c1-1500 | sed "s/^{/{\"span_tab_waves\": \"$TW\", /" >> $O/slop_batch_tab_waves.jsonl
done
( timeout 900 python -m pytest tests/test_phrase.py tests/test_config_scale.py tests/test_search_api.py -m gpu -q -x 2>&1 | tail -2 ) >> $O/slop_batch_tab_waves.jsonl
cat $O/slop_batch_tab_waves.jsonl
exit 0
