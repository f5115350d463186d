#!/bin/bash
# round 5: first A/B of the fixed-point grouped kernel (sa_k_bm25_group_fx) against the fp32 overlay (SA_GROUP_FX=0) and round 4's library
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd $R
C=/tmp/corpus
( time timeout 900 python -m pytest tests/test_group.py tests/test_config_10m.py tests/test_config_scale.py tests/test_reset.py -m gpu -x -q ) > $O/pytest_fx.log 2>&1
( time timeout 600 python scripts/ab.py --corpus-cache $C --ks 10,100,1000 --qsets baseline,distinct,hot --libs build/libsearcharray_hip_r04.so,searcharray_amd/libsearcharray_hip.so --envs "SA_SPARSE=0,SA_GROUP_FX=0;SA_SPARSE=0,SA_GROUP_FX=1" ) > $O/ab_fx.log 2>&1
( time timeout 300 python scripts/ab.py --docs 1250000 --steps 50 --ks 10 --qsets baseline --libs build/libsearcharray_hip_r04.so,searcharray_amd/libsearcharray_hip.so --envs "SA_SPARSE=0,SA_GROUP_FX=0;SA_SPARSE=0,SA_GROUP_FX=1" ) > $O/ab_fx_rank.log 2>&1
cd /tmp
( timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fx -- python $R/scripts/ab.py --corpus-cache $C --ks 10 --qsets baseline --steps 12 --envs "SA_SPARSE=0,SA_GROUP_FX=1" ) > $O/prof_fx.log 2>&1
find $O -name "*.db" -delete 2>/dev/null
find $O -name "*kernel_trace.csv" -size +4M -delete 2>/dev/null
exit 0
