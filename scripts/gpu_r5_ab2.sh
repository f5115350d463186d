#!/bin/bash
# round 5: A/B of grouped-kernel variants (envs in $1, ks in $2, qsets in $3)
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd $R
C=/tmp/corpus
ENVS=${1:-"SA_SPARSE=0,SA_GROUP_FX=0;SA_SPARSE=0,SA_GROUP_FX=1;SA_SPARSE=0,SA_GROUP_FX=1,SA_GROUP_ST=2"}
( time timeout 600 python scripts/ab.py --corpus-cache $C --ks ${2:-10} --qsets ${3:-baseline,distinct,hot} --envs "$ENVS" ) > $O/ab_fx.log 2>&1
( time timeout 300 python scripts/ab.py --docs 1250000 --steps 50 --ks 10 --qsets baseline --envs "$ENVS" ) > $O/ab_fx_rank.log 2>&1
exit 0
