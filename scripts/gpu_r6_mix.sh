#!/bin/bash
# round 6: A/B of kernel builds + the new golden GPU test
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd $R
C=/tmp/corpus
( time timeout 900 python scripts/ab.py --corpus-cache $C --ks 10 --qsets baseline --envs "stage=1;stage=1" --libs searcharray_amd/libsearcharray_hip.so,build/libsearcharray_hip_nt.so,build/libsearcharray_hip_B.so ) > $O/ab_mix.log 2>&1
grep -v "^+" $O/ab_mix.log | grep -E "lib" | cut -c1-300
( time timeout 900 python -m pytest tests/test_config_scale.py tests/test_stage.py -m gpu -q -x ) > $O/pytest_mix.log 2>&1
tail -5 $O/pytest_mix.log
exit 0
