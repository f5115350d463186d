#!/bin/bash
# HBM traffic of the slop pipeline on the heaviest 2- and 3-term slop-2 phrases (zipf-1M): two rocprofv3 --pmc passes
# (FETCH_SIZE needs its own pass, MI355X_MICROARCH.md), summed over the kernels of one query by scripts/slop_pmc.py
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O; rm -rf $O/pmc_slop_f $O/pmc_slop_w
export TMPDIR=/tmp
cd /tmp
( timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_slop_f -- python $R/scripts/slop_heavy.py --terms 2,3 --reps 3 ) > $O/pmc_slop_f.log 2>&1
( timeout 200 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmc_slop_w -- python $R/scripts/slop_heavy.py --terms 2,3 --reps 3 ) > $O/pmc_slop_w.log 2>&1
find $O -name "*.db" -delete 2>/dev/null
grep "^{" $O/pmc_slop_f.log
exit 0
