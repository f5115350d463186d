#!/bin/bash
# round 4, first GPU pass on the head-group kernel: parity tests, same-box A/B against the grouped kernel (SA_HG=0), kernel stats
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
C=/tmp/corpus
cd $R
( time timeout 900 python -m pytest tests/test_headgroup.py tests/test_group.py tests/test_reset.py tests/test_config_scale.py -m gpu -q -x ) > $O/pytest_hg.log 2>&1
tail -5 $O/pytest_hg.log
( time timeout 900 python scripts/ab.py --corpus-cache $C --envs "SA_HG=0;SA_HG=1" --ks 10,1000 --qsets baseline,distinct ) > $O/ab_hg.log 2>&1
cat $O/ab_hg.log | grep -v "^+"
cd /tmp
rm -rf $O/prof_hg
( timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_hg -- python $R/scripts/ab.py --corpus-cache $C --ks 10 --qsets baseline --steps 12 --envs "SA_HG=1" ) > $O/prof_hg.log 2>&1
find $O/prof_hg -name "*kernel_stats.csv" | head -1 | xargs cat | head -12
find $O -name "*.db" -delete 2>/dev/null
find $O -type f -size +8M -delete 2>/dev/null
exit 0
