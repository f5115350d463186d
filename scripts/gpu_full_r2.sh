#!/bin/bash
# round 2, full GPU pass on the final library: smoke, all gpu tests, bench legs, A/B records, side benches, rocprof
# kernel stats and SQ counters.  Everything is copied into profiles/ by scripts/collect_profiles.py r02.
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
rm -rf $O/prof_stats $O/prof_grp $O/prof_grp_sq $O/prof_grp_sq2 $O/prof_slop $O/prof_phrase
mkdir -p $O
export TMPDIR=/tmp
cd $R
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1
( time timeout 1200 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1
( time timeout 600 python bench.py --corpus-cache /tmp/corpus ) > $O/bench.log 2>&1
( time timeout 300 python bench.py --corpus-cache /tmp/corpus --k 100 --no-cpu-baseline --steps 20 ) > $O/bench_k100.log 2>&1
( time timeout 300 python bench.py --corpus-cache /tmp/corpus --k 1000 --no-cpu-baseline --steps 20 ) > $O/bench_k1000.log 2>&1
( time SA_GROUP=0 timeout 300 python bench.py --corpus-cache /tmp/corpus --no-cpu-baseline ) > $O/bench_nogroup.log 2>&1
( time SA_BENCH_FORCE_COMM=1 timeout 300 python bench.py --corpus-cache /tmp/corpus --no-cpu-baseline --no-pmc ) > $O/bench_comm1.log 2>&1
( time RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_PORT=29533 SA_BENCH_FORCE_COMM=1 timeout 300 python bench.py --corpus-cache /tmp/corpus --no-cpu-baseline --no-pmc --docs 1250000 ) > $O/dist1_rccl.log 2>&1
( time timeout 300 python scripts/group_ab.py --corpus-cache /tmp/corpus --ks 10,100,1000 --only 0,1 ) > $O/group_ab.log 2>&1
( time timeout 300 python scripts/phrase_bench.py ) > $O/phrase_bench.log 2>&1
( time timeout 300 python scripts/slop_bench.py ) > $O/slop_bench.log 2>&1
( time timeout 300 python scripts/io_bench.py ) > $O/io_bench.log 2>&1
( time timeout 300 python scripts/sim_bench.py ) > $O/sim_bench.log 2>&1
cd /tmp
( timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-pmc --corpus-cache /tmp/corpus ) > $O/prof_stats.log 2>&1
( timeout 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/prof_grp_sq -- python $R/scripts/group_ab.py --corpus-cache /tmp/corpus --ks 10 --only 1 --qsets baseline --steps 2 ) > $O/prof_grp_sq.log 2>&1
( timeout 120 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/prof_grp_sq2 -- python $R/scripts/group_ab.py --corpus-cache /tmp/corpus --ks 10 --only 1 --qsets baseline --steps 2 ) > $O/prof_grp_sq2.log 2>&1
( timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_phrase -- python $R/scripts/phrase_bench.py --phrases 16 --cpu-phrases 1 ) > $O/prof_phrase.log 2>&1
( timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_slop -- python $R/scripts/slop_bench.py --phrases 16 --cpu-phrases 1 ) > $O/prof_slop.log 2>&1
bash $R/scripts/gpu_slop_prof.sh > $O/slop_heavy.log 2>&1
bash $R/scripts/gpu_slop_pmc.sh > $O/slop_pmc.log 2>&1
cd /tmp
find $O -name "*.db" -delete 2>/dev/null
find $O -type f -size +8M -delete 2>/dev/null
grep -E "passed|failed" $O/pytest_gpu.log
exit 0
