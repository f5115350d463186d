#!/bin/bash
# round 2, first GPU pass: regression tests on the new defaults, the torch-free bench (incl. its own PMC child
# runs and the reference-as-CPU-baseline leg), the communicator path with one rank.
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd $R
( time timeout 600 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > $O/pytest_gpu.log 2>&1
( time timeout 900 python bench.py --corpus-cache /tmp/corpus ) > $O/bench.log 2>&1
( time SA_BENCH_FORCE_COMM=1 timeout 600 python bench.py --corpus-cache /tmp/corpus --no-cpu-baseline --no-pmc ) > $O/bench_comm1.log 2>&1
( time RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_PORT=29511 SA_BENCH_FORCE_COMM=1 timeout 600 python bench.py --corpus-cache /tmp/corpus --no-cpu-baseline --no-pmc --docs 1250000 ) > $O/bench_comm1_rank.log 2>&1
nproc > $O/host.txt; free -g >> $O/host.txt; rocm-smi --showproductname >> $O/host.txt 2>&1
tail -3 $O/pytest_gpu.log
exit 0
