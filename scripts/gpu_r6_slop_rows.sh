#!/bin/bash
# round 6: the slop batch launch with fewer span-table rows per lane (less LDS per block: more resident blocks per CU); variant builds by
# VARIANT_FILE=sa_spans scripts/build_variant.sh rowsN -DSA_SPAN_FROWS=N.  The box's copy of the product library is swapped per variant.
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd $R
cp searcharray_amd/libsearcharray_hip.so /tmp/product.so
rm -f $O/slop_batch_rows.jsonl
for V in product ${VARIANTS:-rows8 rows10}; do
  if [ $V = product ]; then cp /tmp/product.so searcharray_amd/libsearcharray_hip.so; else cp build/libsearcharray_hip_$V.so searcharray_amd/libsearcharray_hip.so; fi
  timeout 300 python scripts/slop_batch_prof.py slop 2>/dev/null | grep "^{" | sed "s/^{/{\"build\": \"$V\", /" >> $O/slop_batch_rows.jsonl
  timeout 300 python scripts/slop_bench.py 2>/dev/null | grep "^{" | cut -c1-700 | sed "s/^{/{\"build\": \"$V\", /" >> $O/slop_batch_rows.jsonl
  if [ $V != product ]; then ( timeout 600 python -m pytest tests/test_phrase.py tests/test_config_scale.py -m gpu -q -x -k "slop or span" 2>&1 | tail -1 ) >> $O/slop_batch_rows.jsonl; fi
done
cp /tmp/product.so searcharray_amd/libsearcharray_hip.so
cat $O/slop_batch_rows.jsonl
exit 0
