#!/bin/bash
# GPU check of the round's late additions: on-disk loader, device similarities, segmented derivation
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
( timeout 600 python -m pytest tests/test_io.py tests/test_search_api.py tests/test_bm25.py -q -m gpu -k "io or similarit or segmented or file or data_dir" 2>&1 | tail -5 ) > $O/new_tests.log 2>&1
( timeout 600 python scripts/io_bench.py --docs 10000000 ) > $O/io_bench.log 2>&1
( timeout 300 python scripts/sim_bench.py ) > $O/sim_bench.log 2>&1
tail -n 3 $O/new_tests.log; tail -n 2 $O/io_bench.log; tail -n 2 $O/sim_bench.log
exit 0
