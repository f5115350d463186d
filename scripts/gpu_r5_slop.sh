#!/bin/bash
# round 5: the slop / phrase batch legs -- timing, kernel stats, HBM traffic (PMC) -- of the library in the tree ($1: tag)
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
T=${1:-new}
mkdir -p $O
export TMPDIR=/tmp
cd $R
( timeout 300 python scripts/slop_batch_prof.py ) > $O/slopb_$T.log 2>&1
( timeout 300 python -m pytest tests/test_phrase.py tests/test_config_scale.py -m gpu -q -x ) > $O/pytest_phrase_$T.log 2>&1
cd /tmp
rm -rf $O/prof_slopb_$T $O/pmc_slopb_f_$T $O/pmc_slopb_w_$T
( timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_slopb_$T -- python $R/scripts/slop_batch_prof.py slop ) > $O/prof_slopb_$T.log 2>&1
( timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_slopb_f_$T -- python $R/scripts/slop_batch_prof.py slop ) > $O/pmc_slopb_f_$T.log 2>&1
( timeout 200 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmc_slopb_w_$T -- python $R/scripts/slop_batch_prof.py slop ) > $O/pmc_slopb_w_$T.log 2>&1
find $O -name "*.db" -delete 2>/dev/null
find $O -name "*kernel_trace.csv" -size +4M -delete 2>/dev/null
exit 0
